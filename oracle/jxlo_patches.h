// TEST INFRASTRUCTURE (oracle): patches, restated from the reference:
//   dictionary  lib/jxl/dec_patch_dictionary.cc:32-175 (PatchDictionary::Decode; contexts patch_dictionary_internal.h:12-24)
//   application lib/jxl/dec_patch_dictionary.cc:317-356 (AddOneRow) with lib/jxl/blending.cc:40-190 (PerformBlending) for the
//               colour modes that need no alpha (kNone, kReplace, kAdd, kMul); extra channels must be left alone (kNone)
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
  uint32_t mode;       // PatchBlendMode of the colour channels: 0 none, 1 replace, 2 add, 3 multiply
  bool clamp;
};
struct Patches {
  std::vector<PatchRef> refs;
  std::vector<PatchPos> pos;
};
// A reference frame kept before its colour transform: three XYB planes of w x h samples.
struct XybSlot {
  std::vector<float> p[3];
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
      for (size_t j = 0; j < num_extra + 1; j++) {
        const uint32_t mode = rd.Read(5);
        JXLO_CHECK(mode < 8, "invalid patch blend mode");
        JXLO_CHECK(mode < 4, "unsupported: patches blended through an alpha channel");
        bool clamp = false;
        if (mode == 3) clamp = rd.Read(9) != 0;
        if (j == 0) {
          p.mode = mode;
          p.clamp = clamp;
        } else {
          JXLO_CHECK(mode == 0, "unsupported: patches that touch extra channels");
        }
      }
      out->pos.push_back(p);
    }
    out->refs.push_back(r);
  }
  JXLO_CHECK(rd.FinalStateOk(), "patches: bad ANS final state");
}

// Every patch onto the three planes (rows of `stride` floats), in dictionary order.
static inline void ApplyPatches(const Patches& P, const XybSlot* slots, float* p0, float* p1, float* p2, size_t stride) {
  float* planes[3] = {p0, p1, p2};
  for (const PatchPos& q : P.pos) {
    const PatchRef& r = P.refs[q.ref];
    const XybSlot& s = slots[r.slot];
    for (size_t y = 0; y < r.ysize; y++)
      for (size_t x = 0; x < r.xsize; x++)
        for (int c = 0; c < 3; c++) {
          const float fg = s.p[c][(r.y0 + y) * s.w + r.x0 + x];
          float& bg = planes[c][(q.y + y) * stride + q.x + x];
          switch (q.mode) {
            case 1: bg = fg; break;
            case 2: bg = bg + fg; break;
            case 3: bg = bg * (q.clamp ? (fg < 0.0f ? 0.0f : (fg > 1.0f ? 1.0f : fg)) : fg); break;
            default: break;
          }
        }
  }
}

}  // namespace jxlo
#endif  // JXLO_PATCHES_H_
