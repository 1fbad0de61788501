// Patches of a frame (host half): the dictionary is entropy-decoded here (lib/jxl/dec_patch_dictionary.cc:32-175; contexts
// patch_dictionary_internal.h:12-24) and laid out for the device: one record per patch position and, like the splines,
// per-row lists in dictionary order that k_patches_add walks (dec_patch_dictionary.cc:317-356 AddOneRow; every colour mode,
// and every mode on the alpha channel of an image whose only extra channel is alpha: blending.cc:40-190; other extra
// channels must be left alone). The rectangles are checked against the reference frames when the sources are known
// (jxlamd_frame_set_patch_sources).
#ifndef JXH_PATCHES_H_
#define JXH_PATCHES_H_

#include <cstdint>
#include <vector>

#include "jxh_bits.h"
#include "jxh_entropy.h"

namespace jxh {

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

static inline void DecodePatches(BitReader& br, size_t xsize, size_t ysize, size_t num_extra, Patches* out) {
  EntropyCode code;
  DecodeHistograms(br, 10, &code);
  SymbolReader rd(&code, &br);
  const size_t num_ref = rd.Read(0);
  const size_t max_ref = 1024 + xsize * ysize / 4, max_patches = max_ref * 4;
  JXH_CHECK(num_ref <= max_ref, "too many patches");
  size_t total = 0;
  for (size_t id = 0; id < num_ref; id++) {
    PatchRef r;
    r.slot = rd.Read(1);
    JXH_CHECK(r.slot < 4, "patches: invalid reference frame");
    r.x0 = rd.Read(3);
    r.y0 = rd.Read(3);
    r.xsize = rd.Read(2) + 1;
    r.ysize = rd.Read(2) + 1;
    size_t count = rd.Read(7);
    JXH_CHECK(count <= max_patches, "too many patches");
    count++;
    total += count;
    JXH_CHECK(total <= max_patches, "too many patches");
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
        JXH_CHECK(int64_t(out->pos.back().x) + dx >= 0 && int64_t(out->pos.back().y) + dy >= 0, "patches: negative coordinate");
        p.x = uint32_t(int64_t(out->pos.back().x) + dx);
        p.y = uint32_t(int64_t(out->pos.back().y) + dy);
      }
      JXH_CHECK(uint64_t(p.x) + r.xsize <= xsize && uint64_t(p.y) + r.ysize <= ysize, "patches: outside the frame");
      p.mode = 0;
      p.clamp = false;
      for (size_t j = 0; j < num_extra + 1; j++) {  // dec_patch_dictionary.cc:135-163
        const uint32_t mode = rd.Read(5);
        JXH_CHECK(mode < 8, "invalid patch blend mode");
        if (mode >= 4 && num_extra > 1) JXH_CHECK(rd.Read(8) < num_extra, "invalid alpha channel for blending");
        bool clamp = false;
        if (mode >= 3) clamp = rd.Read(9) != 0;
        if (j == 0) {
          p.mode = mode;
          p.clamp = clamp;
        } else if (j == 1) {
          p.ec_mode = mode;
          p.ec_clamp = clamp;
        } else {
          JXH_CHECK(mode == 0, "unsupported: patches that touch extra channels other than alpha");
        }
        if (mode >= 4 || (j > 0 && mode != 0)) {
          JXH_CHECK(num_extra <= 1, "unsupported: patches that blend through alpha in images with several extra channels");
          out->uses_alpha = true;
        }
      }
      out->pos.push_back(p);
    }
    out->refs.push_back(r);
  }
  JXH_CHECK(rd.FinalStateOk(), "patches: bad ANS final state");
}

// Device layout: 8 u32 per position {x, y, xsize, ysize, ref x0, ref y0, slot, mode | clamp << 8 | the alpha channel's
// mode << 16 | its clamp << 24}; row y applies
// positions row_list[row_start[y] .. row_start[y + 1]) in dictionary order.
static inline void BuildPatchRows(const Patches& P, size_t ysize, std::vector<uint32_t>* records, std::vector<uint32_t>* row_start,
                                  std::vector<uint32_t>* row_list) {
  records->clear();
  row_start->assign(ysize + 1, 0);
  uint64_t total = 0;
  for (const PatchPos& q : P.pos) {
    const PatchRef& r = P.refs[q.ref];
    const uint32_t rec[8] = {q.x, q.y, r.xsize, r.ysize, r.x0, r.y0, r.slot,
                             q.mode | (q.clamp ? 256u : 0u) | q.ec_mode << 16 | (q.ec_clamp ? 1u << 24 : 0u)};
    records->insert(records->end(), rec, rec + 8);
    total += r.ysize;
  }
  JXH_CHECK(total < (uint64_t(1) << 26), "patches cover too many rows");
  for (const PatchPos& q : P.pos)
    for (uint32_t y = 0; y < P.refs[q.ref].ysize && q.y + y < ysize; y++) (*row_start)[q.y + y + 1]++;
  for (size_t y = 0; y < ysize; y++) (*row_start)[y + 1] += (*row_start)[y];
  row_list->assign((*row_start)[ysize], 0);
  std::vector<uint32_t> fill(row_start->begin(), row_start->end() - 1);
  for (size_t i = 0; i < P.pos.size(); i++)
    for (uint32_t y = 0; y < P.refs[P.pos[i].ref].ysize && P.pos[i].y + y < ysize; y++) (*row_list)[fill[P.pos[i].y + y]++] = uint32_t(i);
}

}  // namespace jxh
#endif  // JXH_PATCHES_H_
