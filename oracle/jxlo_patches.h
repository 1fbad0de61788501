// TEST INFRASTRUCTURE (oracle): patches, restated from the reference:
//   dictionary  lib/jxl/dec_patch_dictionary.cc:32-175 (PatchDictionary::Decode; contexts patch_dictionary_internal.h:12-24)
//   application lib/jxl/dec_patch_dictionary.cc:317-356 (AddOneRow) with lib/jxl/blending.cc:40-190 (PerformBlending) and
//               lib/jxl/alpha.cc:17-101: every colour mode, and every mode on the alpha channel of an image whose only extra
//               channel is alpha (other extra channels must be left alone: kNone)
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use this file.
#ifndef JXLO_PATCHES_H_
#define JXLO_PATCHES_H_

#include <cstdint>
#include <vector>

#include "jxlo_bits.h"
#include "jxlo_entropy.h"

namespace jxlo {

struct PatchRef {
  uint32_t slot, x0, y0, xsize, ysize;
};
struct PatchPos {
  uint32_t x, y, ref;  // ref: index into refs
  uint32_t mode;       // PatchBlendMode of the colour channels (dec_patch_dictionary.h:32-58): 0 none, 1 replace, 2 add, 3 multiply,
                       // 4 / 5 blend above / below, 6 / 7 alpha-weighted add above / below
  bool clamp;
  uint32_t ec_mode = 0;  // ... of the alpha channel (extra channel 0)
  bool ec_clamp = false;
};
struct Patches {
  std::vector<PatchRef> refs;
  std::vector<PatchPos> pos;
  bool uses_alpha = false;  // some position blends through alpha or changes the alpha channel
};
// A reference frame kept before its colour transform: three XYB planes of w x h samples (and its alpha channel, if any).
struct XybSlot {
  std::vector<float> p[3];
  std::vector<float> alpha;
  size_t w = 0, h = 0;
};

static inline void DecodePatches(BitReader& br, size_t xsize, size_t ysize, size_t num_extra, const XybSlot* slots, Patches* out) {
  EntropyCode code;
  DecodeHistograms(br, 10, &code);
  SymbolReader rd(&code, &br);
  const size_t num_ref = rd.Read(0);
  const size_t max_ref = 1024 + xsize * ysize / 4, max_patches = max_ref * 4;
  JXLO_CHECK(num_ref <= max_ref, "too many patches");
  size_t total = 0;
  for (size_t id = 0; id < num_ref; id++) {
    PatchRef r;
    r.slot = rd.Read(1);
    JXLO_CHECK(r.slot < 4 && slots[r.slot].w != 0, "patches: invalid reference frame");
    r.x0 = rd.Read(3);
    r.y0 = rd.Read(3);
    r.xsize = rd.Read(2) + 1;
    r.ysize = rd.Read(2) + 1;
    JXLO_CHECK(uint64_t(r.x0) + r.xsize <= slots[r.slot].w && uint64_t(r.y0) + r.ysize <= slots[r.slot].h, "patches: rectangle outside the reference frame");
    size_t count = rd.Read(7);
    JXLO_CHECK(count <= max_patches, "too many patches");
    count++;
    total += count;
    JXLO_CHECK(total <= max_patches, "too many patches");
    for (size_t i = 0; i < count; i++) {
      PatchPos p;
      p.ref = uint32_t(out->refs.size());
      if (i == 0) {
        p.x = rd.Read(4);
        p.y = rd.Read(4);
      } else {
        const uint32_t ux = rd.Read(6), uy = rd.Read(6);
        const int64_t dx = (ux & 1) ? -int64_t((uint64_t(ux) + 1) >> 1) : int64_t(ux >> 1);
        const int64_t dy = (uy & 1) ? -int64_t((uint64_t(uy) + 1) >> 1) : int64_t(uy >> 1);
        JXLO_CHECK(int64_t(out->pos.back().x) + dx >= 0 && int64_t(out->pos.back().y) + dy >= 0, "patches: negative coordinate");
        p.x = uint32_t(int64_t(out->pos.back().x) + dx);
        p.y = uint32_t(int64_t(out->pos.back().y) + dy);
      }
      JXLO_CHECK(uint64_t(p.x) + r.xsize <= xsize && uint64_t(p.y) + r.ysize <= ysize, "patches: outside the frame");
      p.mode = 0;
      p.clamp = false;
      for (size_t j = 0; j < num_extra + 1; j++) {  // dec_patch_dictionary.cc:135-163
        const uint32_t mode = rd.Read(5);
        JXLO_CHECK(mode < 8, "invalid patch blend mode");
        if (mode >= 4 && num_extra > 1) JXLO_CHECK(rd.Read(8) < num_extra, "invalid alpha channel for blending");
        bool clamp = false;
        if (mode >= 3) clamp = rd.Read(9) != 0;
        if (j == 0) {
          p.mode = mode;
          p.clamp = clamp;
        } else if (j == 1) {
          p.ec_mode = mode;
          p.ec_clamp = clamp;
        } else {
          JXLO_CHECK(mode == 0, "unsupported: patches that touch extra channels other than alpha");
        }
        if (mode >= 4 || (j > 0 && mode != 0)) {
          JXLO_CHECK(num_extra <= 1, "unsupported: patches that blend through alpha in images with several extra channels");
          out->uses_alpha = true;
        }
      }
      out->pos.push_back(p);
    }
    out->refs.push_back(r);
  }
  JXLO_CHECK(rd.FinalStateOk(), "patches: bad ANS final state");
}

// Every patch onto the three planes (rows of `stride` floats), in dictionary order. alpha: the frame's alpha channel (rows
// of alpha_stride floats) or NULL for an image without one; premultiplied: ExtraChannelInfo::alpha_associated.
// blending.cc:40-190 for one extra channel (alpha, index 0): the alpha channel's own mode first, from the values before
// blending; then the colour channels, whose blend above / below ALSO writes the alpha channel (:127-136: the output layer's
// `a` is the alpha channel's row), whatever its own mode produced. alpha.cc:17-101 for the arithmetic; the special cases of
// a channel that is its own alpha (pointer equality there) are spelled out.
static inline void ApplyPatches(const Patches& P, const XybSlot* slots, float* p0, float* p1, float* p2, size_t stride, float* alpha = nullptr,
                                size_t alpha_stride = 0, bool premultiplied = false) {
  float* planes[3] = {p0, p1, p2};
  auto c01 = [](float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); };
  for (const PatchPos& q : P.pos) {
    const PatchRef& r = P.refs[q.ref];
    const XybSlot& s = slots[r.slot];
    const bool has_alpha = alpha != nullptr;
    JXLO_CHECK(!has_alpha || !P.uses_alpha || s.alpha.size() == s.w * s.h, "patches: the reference frame has no alpha channel");
    for (size_t y = 0; y < r.ysize; y++)
      for (size_t x = 0; x < r.xsize; x++) {
        const size_t si = (r.y0 + y) * s.w + r.x0 + x;
        const float fga = has_alpha && P.uses_alpha ? s.alpha[si] : 1.0f;
        float* ap = has_alpha ? &alpha[(q.y + y) * alpha_stride + q.x + x] : nullptr;
        const float bga = ap ? *ap : 1.0f;
        float a_out = bga;
        if (has_alpha) {
          switch (q.ec_mode) {
            case 1: a_out = fga; break;
            case 2: a_out = bga + fga; break;
            case 3: a_out = bga * (q.ec_clamp ? c01(fga) : fga); break;
            case 4: a_out = 1.0f - (1.0f - (q.ec_clamp ? c01(fga) : fga)) * (1.0f - bga); break;
            case 5: a_out = 1.0f - (1.0f - (q.ec_clamp ? c01(bga) : bga)) * (1.0f - fga); break;
            case 6: a_out = bga; break;  // (PerformAlphaWeightedAdd with fg == fga: the bottom layer unchanged)
            case 7: a_out = fga; break;
            default: break;
          }
        }
        uint32_t mode = q.mode;
        if (!has_alpha && mode >= 4) mode = mode >= 6 ? 2 : 1;  // blending.cc:154-168: without alpha, add / the top layer
        for (int c = 0; c < 3; c++) {
          const float fg = s.p[c][si];
          float& bg = planes[c][(q.y + y) * stride + q.x + x];
          switch (mode) {
            case 1: bg = fg; break;
            case 2: bg = bg + fg; break;
            case 3: bg = bg * (q.clamp ? c01(fg) : fg); break;
            case 4: {  // kBlendAbove: the patch over the frame
              const float fa = q.clamp ? c01(fga) : fga;
              if (premultiplied) {
                bg = fg + bg * (1.0f - fa);
              } else {
                const float new_a = 1.0f - (1.0f - fa) * (1.0f - bga);
                const float rnew_a = new_a > 0 ? 1.0f / new_a : 0.0f;
                bg = (fg * fa + bg * bga * (1.0f - fa)) * rnew_a;
              }
              break;
            }
            case 5: {  // kBlendBelow: the frame over the patch
              const float fa = q.clamp ? c01(bga) : bga;
              if (premultiplied) {
                bg = bg + fg * (1.0f - fa);
              } else {
                const float new_a = 1.0f - (1.0f - fa) * (1.0f - fga);
                const float rnew_a = new_a > 0 ? 1.0f / new_a : 0.0f;
                bg = (bg * fa + fg * fga * (1.0f - fa)) * rnew_a;
              }
              break;
            }
            case 6: bg = bg + fg * (q.clamp ? c01(fga) : fga); break;
            case 7: bg = fg + bg * (q.clamp ? c01(bga) : bga); break;
            default: break;
          }
        }
        if (mode == 4) a_out = 1.0f - (1.0f - (q.clamp ? c01(fga) : fga)) * (1.0f - bga);
        if (mode == 5) a_out = 1.0f - (1.0f - (q.clamp ? c01(bga) : bga)) * (1.0f - fga);
        if (ap) *ap = a_out;
      }
  }
}

}  // namespace jxlo
#endif  // JXLO_PATCHES_H_
