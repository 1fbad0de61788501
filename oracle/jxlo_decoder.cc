// ORACLE (test infrastructure, not product code): whole-frame scalar decoder with intermediate dumps.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Parity status: the entropy/bitstream/Modular layers are pinned by streams produced by the reference's own
// standalone encoder (oracle/_ref, built from lib/jxl/enc_fast_lossless.cc) and by the 1x1 codestream embedded in
// the reference's decode_test.cc; the VarDCT layer is "parity unpinned" against reference *pixels* (the
// reference decoder cannot be built in this image) and is pinned instead by the closed-form definitions the
// reference's tests use (see tests/ and DESIGN.md).
//
// Follows the frame walk of reference lib/jxl/dec_frame.cc:135-733 (InitFrame, ProcessDCGlobal, ProcessDCGroup,
// FinalizeDC, ProcessACGlobal, ProcessACGroup), lib/jxl/dec_modular.cc:209-562, lib/jxl/dec_group.cc:183-460,
// lib/jxl/dec_cache.cc:117-371 (stage order), lib/jxl/decode.cc:115-150 (signature).
#include <omp.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>

#include "jxlo_bits.h"
#include "jxlo_entropy.h"
#include "jxlo_headers.h"
#include "jxlo_modular.h"
#include "jxlo_patches.h"
#include "jxlo_render.h"
#include "jxlo_splines.h"
#include "jxlo_vardct.h"

namespace jxlo {
static size_t g_flush_prefix = 0;  // jxlo_set_flush_prefix: bytes of the codestream that "have arrived" (0 = all)

struct Decoded {
  ImageHeader ih;
  FrameHeader fh;
  FrameDim dim;
  int out_channels = 3;
  size_t out_xsize = 0, out_ysize = 0;  // the image (= frame size times the frame's upsampling factor, cropped)
  std::vector<uint8_t> rgb8;   // interleaved, out_xsize*out_ysize*out_channels
  std::vector<float> rgbf;     // planar 3 x ysize x xsize (after transfer function, before uint8)
  std::vector<float> alphaf;   // the alpha plane as floats (images with alpha)
  std::vector<float> xyb_save; // frames kept before the colour transform: 3 x out_ysize x out_xsize (dec_cache.cc:225-230)
  // VarDCT intermediates
  std::vector<int32_t> coeffs;     // [group][c][65536], block-contiguous per varblock
  std::vector<int32_t> nzeros;     // [group][c][32*32]
  std::vector<float> xyb_idct;     // 3 planes, ysize_padded x xsize_padded (storage order X, Y, B)
  std::vector<float> xyb_filtered; // 3 planes, ysize x xsize, stride xsize_padded
  std::vector<float> dc;           // 3 planes, ysize_blocks x xsize_blocks (after smoothing)
  std::vector<float> dc_unsmoothed;  // the same before AdaptiveDCSmoothing (kept for the tests' third reading of it)
  float dc_step[3] = {0, 0, 0};      // DC quantisation step per channel (what the smoothing measures its gap in)
  float sigma_params[10] = {0};      // quant_scale, epf_quant_mul, epf_sharp_lut[8] (inputs of ComputeSigma)
  std::vector<uint8_t> acs;        // ysize_blocks x xsize_blocks: (strategy<<1)|is_first
  std::vector<int32_t> quant;      // raw quant field (valid at first blocks)
  std::vector<uint8_t> sharpness;
  std::vector<int8_t> ytox, ytob;
  std::vector<float> inv_sigma;
  std::vector<uint8_t> quant_dc;
  std::vector<int32_t> modular;    // planar decoded integer channels for Modular frames
  uint32_t used_acs = 0;
  uint64_t ac_symbols = 0;
};

struct FrameState {
  Decoded* out;
  const ImageHeader* ih;
  FrameHeader fh;
  FrameDim dim;
  size_t frame_index = 0, nonvisible_index = 0;  // visible frames before this one, invisible ones since (dec_frame.cc:160-168)
  // DC global
  DequantTables dq;
  uint32_t global_scale = 1, quant_dc = 16;
  BlockCtxMap bctx;
  uint32_t color_factor = 84;
  float base_corr_x = 0.0f, base_corr_b = 1.0f;
  int32_t ytox_dc = 0, ytob_dc = 0;
  Splines splines;
  bool has_splines = false;
  Patches patches;
  bool has_patches = false;
  const XybSlot* xyb_slots = nullptr;  // the reference frames kept before their colour transform
  std::vector<float> patch_alpha;      // the alpha channel the patch stage read and wrote (empty: the patches left it alone)
  float noise_lut[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool has_noise = false;
  MGlobal mglobal;
  MImage full;  // global modular image (extra channels, or colour for Modular frames)
  size_t modular_color_channels = 0;
  // AC global
  size_t num_histograms = 1;
  std::vector<std::vector<uint32_t>> orders;  // per pass
  std::vector<EntropyCode> ac_codes;          // per pass
  // planes
  size_t xb, yb;
  Planes3 idct;  // padded
};

static void DecodeDcGlobal(BitReader& br, FrameState* s) {
  const FrameHeader& fh = s->fh;
  if (fh.flags & FrameHeader::kPatches) {  // dec_frame.cc:271-285
    JXLO_CHECK(s->xyb_slots != nullptr, "patches without reference frames");  // (drawn at the frame's resolution, before any upsampling: dec_cache.cc:193-212)
    DecodePatches(br, s->dim.xsize_padded, s->dim.ysize_padded, s->ih->extra.size(), s->xyb_slots, &s->patches);
    s->has_patches = true;
  }
  if (fh.flags & FrameHeader::kSplines) {  // dec_frame.cc:289-293
    DecodeSplines(br, s->dim.xsize * s->dim.ysize, &s->splines);
    s->has_splines = true;
  }
  if (fh.flags & FrameHeader::kNoise) {  // dec_frame.cc:294-296, dec_noise.cc:154-164: eight 10-bit LUT points
    for (float& v : s->noise_lut) v = float(br.Read(10)) / 1024.0f;
    s->has_noise = true;
  }
  // DequantMatrices::DecodeDC
  if (!br.ReadBool()) {
    for (int c = 0; c < 3; c++) {
      s->dq.dc_quant[c] = ReadF16(br) * (1.0f / 128.0f);
      JXLO_CHECK(s->dq.dc_quant[c] >= 1e-8f, "invalid DC quant");
    }
  }
  if (!fh.modular) {
    s->global_scale = ReadU32(br, BitsOffset(11, 1), BitsOffset(11, 2049), BitsOffset(12, 4097), BitsOffset(16, 8193));
    s->quant_dc = ReadU32(br, Val(16), BitsOffset(5, 1), BitsOffset(8, 1), BitsOffset(16, 1));
    ReadBlockCtxMap(br, &s->bctx);
    if (!br.ReadBool()) {  // ColorCorrelation::DecodeDC
      s->color_factor = ReadU32(br, Val(84), Val(256), BitsOffset(8, 2), BitsOffset(16, 258));
      s->base_corr_x = ReadF16(br);
      s->base_corr_b = ReadF16(br);
      JXLO_CHECK(std::fabs(s->base_corr_x) <= 4.0f && std::fabs(s->base_corr_b) <= 4.0f, "CfL base out of range");
      s->ytox_dc = int32_t(br.Read(8)) - 128;
      s->ytob_dc = int32_t(br.Read(8)) - 128;
    }
  }
  // Modular global info
  if (br.ReadBool()) {
    size_t nb = (fh.modular ? (s->ih->gray ? 1 : 3) : 0) + s->ih->extra.size();
    size_t limit = std::min<size_t>(size_t(1) << 22, 1024 + s->dim.xsize * s->dim.ysize * std::max<size_t>(nb, 1) / 16);
    DecodeTree(br, &s->mglobal.tree, limit);
    DecodeHistograms(br, (s->mglobal.tree.size() + 1) / 2, &s->mglobal.code);
    s->mglobal.have = true;
  }
  size_t nb_color = 0;
  if (fh.modular) nb_color = (s->ih->gray && !fh.ycbcr) ? 1 : 3;
  s->modular_color_channels = nb_color;
  s->full.ch.clear();
  s->full.bitdepth = int(s->ih->bits);
  for (size_t c = 0; c < nb_color; c++) s->full.ch.emplace_back(s->dim.xsize, s->dim.ysize);
  for (size_t e = 0; e < s->ih->extra.size(); e++) {
    // dec_modular.cc:262-271: ceil(image / its own upsampling factor) samples, shifted against the frame by the ratio of the
    // two factors (frame_header.cc:272-283: at least the frame's)
    uint32_t ups = s->fh.ec_upsampling.empty() ? 1 : s->fh.ec_upsampling[e];
    JXLO_CHECK(ups >= fh.upsampling, "EC upsampling < color upsampling, which is invalid");
    JXLO_CHECK(!fh.modular || (ups == 1 && fh.upsampling == 1), "unsupported: upsampled Modular frames");
    const int shift = int(CeilLog2(ups)) - int(CeilLog2(fh.upsampling));
    s->full.ch.emplace_back(DivCeil(size_t(fh.xsize), size_t(ups)), DivCeil(size_t(fh.ysize), size_t(ups)), shift, shift);
  }
  // Stream 0: channels no larger than a group are coded here; transforms stay pending on the full image.
  ModularDecode(br, &s->full, 0, &s->mglobal, s->dim.group_dim, /*undo_transforms=*/false);
}

// Decodes the part of the global modular image covered by `rect` whose channels fall in the shift bracket.
static void DecodeModularGroup(BitReader& br, FrameState* s, size_t x0, size_t y0, size_t xs, size_t ys, int min_shift,
                               int max_shift, int stream_id) {
  MImage& full = s->full;
  size_t c = full.nb_meta;
  for (; c < full.ch.size(); c++)
    if (full.ch[c].w > s->dim.group_dim || full.ch[c].h > s->dim.group_dim) break;
  size_t beginc = c;
  MImage gi;
  gi.bitdepth = full.bitdepth;
  std::vector<size_t> map;
  struct R { size_t x0, y0, w, h; };
  std::vector<R> rects;
  for (; c < full.ch.size(); c++) {
    MChannel& fc = full.ch[c];
    int shift = std::min(fc.hshift, fc.vshift);
    if (shift > max_shift || shift < min_shift) continue;
    size_t rx0 = x0 >> fc.hshift, ry0 = y0 >> fc.vshift;
    if (rx0 >= fc.w || ry0 >= fc.h) continue;
    size_t rw = std::min(xs >> fc.hshift, fc.w - rx0), rh = std::min(ys >> fc.vshift, fc.h - ry0);
    if (rw == 0 || rh == 0) continue;
    gi.ch.emplace_back(rw, rh, fc.hshift, fc.vshift);
    map.push_back(c);
    rects.push_back({rx0, ry0, rw, rh});
  }
  (void)beginc;
  if (gi.ch.empty()) return;
  ModularDecode(br, &gi, stream_id, &s->mglobal);
  for (size_t i = 0; i < map.size(); i++) {
    MChannel& fc = full.ch[map[i]];
    for (size_t y = 0; y < rects[i].h; y++)
      memcpy(fc.Row(rects[i].y0 + y) + rects[i].x0, gi.ch[i].Row(y), rects[i].w * sizeof(int32_t));
  }
}

static void DecodeDcGroup(BitReader& br, FrameState* s, size_t g) {
  Decoded* o = s->out;
  const FrameDim& d = s->dim;
  const size_t gx = g % d.xsize_dc_groups, gy = g / d.xsize_dc_groups;
  const size_t bx0 = gx * d.group_dim, by0 = gy * d.group_dim;  // in blocks
  const size_t bw = std::min(d.group_dim, d.xsize_blocks - bx0), bh = std::min(d.group_dim, d.ysize_blocks - by0);
  const size_t ndc = d.num_dc_groups;
  if (!s->fh.modular && !(s->fh.flags & FrameHeader::kUseDcFrame)) {  // (dec_frame.cc:322-326: no DC stream with kUseDcFrame)
    // VarDCT DC: three channels in Y, X, B order (dec_modular.cc:427-465)
    uint32_t extra_precision = uint32_t(br.Read(2));
    float mul = 1.0f / float(1 << extra_precision);
    MImage img;
    img.bitdepth = s->full.bitdepth;
    // (dec_modular.cc:443-452: stream channel 0 is Y, 1 is X / Cb, 2 is B / Cr; a subsampled channel is smaller)
    const FrameHeader& fh = s->fh;
    static const int kStreamChan[3] = {1, 0, 2};
    for (int i = 0; i < 3; i++) img.ch.emplace_back(bw >> fh.hshift[kStreamChan[i]], bh >> fh.vshift[kStreamChan[i]]);
    ModularDecode(br, &img, int(1 + g), &s->mglobal);
    // DequantDC (compressed_dc.cc:201-296)
    const float inv_global_scale = 65536.0f / float(s->global_scale);
    const float inv_quant_dc = inv_global_scale / float(s->quant_dc);
    float fac[3];
    for (int c = 0; c < 3; c++) fac[c] = (inv_quant_dc * s->dq.dc_quant[c]) * mul;
    const float color_scale = 1.0f / float(s->color_factor);
    const float cfl_x = s->base_corr_x + s->ytox_dc * color_scale, cfl_b = s->base_corr_b + s->ytob_dc * color_scale;
    const size_t plane = d.xsize_blocks * d.ysize_blocks;
    if (!fh.Is444()) {
      // compressed_dc.cc:232-250: every channel on its own grid (the top-left part of its plane), no chroma from luma;
      // :253-290: a block's bucket from the samples that cover it
      for (int c = 0; c < 3; c++) {
        const MChannel& ch = img.ch[c < 2 ? c ^ 1 : c];
        const size_t sx0 = bx0 >> fh.hshift[c], sy0 = by0 >> fh.vshift[c];
        for (size_t y = 0; y < ch.h; y++)
          for (size_t x = 0; x < ch.w; x++) o->dc[plane * c + (sy0 + y) * d.xsize_blocks + sx0 + x] = float(ch.Row(y)[x]) * fac[c];
      }
      for (size_t y = 0; y < bh; y++) {
        const int32_t *qx = img.ch[1].Row(y >> fh.vshift[0]), *qy = img.ch[0].Row(y >> fh.vshift[1]), *qb = img.ch[2].Row(y >> fh.vshift[2]);
        for (size_t x = 0; x < bw; x++) {
          uint8_t bucket = 0;
          if (s->bctx.num_dc_ctxs > 1) {
            int bxk = 0, byk = 0, bbk = 0;
            for (int t : s->bctx.dc_thresholds[0]) if (qx[x >> fh.hshift[0]] > t) bxk++;
            for (int t : s->bctx.dc_thresholds[1]) if (qy[x >> fh.hshift[1]] > t) byk++;
            for (int t : s->bctx.dc_thresholds[2]) if (qb[x >> fh.hshift[2]] > t) bbk++;
            int b = bxk;
            b = b * int(s->bctx.dc_thresholds[2].size() + 1) + bbk;
            b = b * int(s->bctx.dc_thresholds[1].size() + 1) + byk;
            bucket = uint8_t(b);
          }
          o->quant_dc[(by0 + y) * d.xsize_blocks + bx0 + x] = bucket;
        }
      }
    } else
    for (size_t y = 0; y < bh; y++) {
      const int32_t *qx = img.ch[1].Row(y), *qy = img.ch[0].Row(y), *qb = img.ch[2].Row(y);
      for (size_t x = 0; x < bw; x++) {
        size_t idx = (by0 + y) * d.xsize_blocks + bx0 + x;
        float in_x = float(qx[x]) * fac[0], in_y = float(qy[x]) * fac[1], in_b = float(qb[x]) * fac[2];
        o->dc[plane * 1 + idx] = in_y;
        o->dc[plane * 0 + idx] = in_y * cfl_x + in_x;
        o->dc[plane * 2 + idx] = in_y * cfl_b + in_b;
        uint8_t bucket = 0;
        if (s->bctx.num_dc_ctxs > 1) {
          int bxk = 0, byk = 0, bbk = 0;
          for (int t : s->bctx.dc_thresholds[0]) if (qx[x] > t) bxk++;
          for (int t : s->bctx.dc_thresholds[1]) if (qy[x] > t) byk++;
          for (int t : s->bctx.dc_thresholds[2]) if (qb[x] > t) bbk++;
          int b = bxk;
          b = b * int(s->bctx.dc_thresholds[2].size() + 1) + bbk;
          b = b * int(s->bctx.dc_thresholds[1].size() + 1) + byk;
          bucket = uint8_t(b);
        }
        o->quant_dc[idx] = bucket;
      }
    }
  }
  // Modular DC: channels of the global image with shift >= 3
  DecodeModularGroup(br, s, bx0 * 8, by0 * 8, d.dc_group_dim, d.dc_group_dim, 3, 1000, int(1 + ndc + g));
  if (!s->fh.modular) {
    // AC metadata (dec_modular.cc:467-562)
    size_t upper = bw * bh;
    size_t count = size_t(br.Read(CeilLog2(upper))) + 1;
    MImage img;
    img.bitdepth = s->full.bitdepth;
    size_t cw = (bw + 7) >> 3, chh = (bh + 7) >> 3;
    img.ch.emplace_back(cw, chh, 3, 3);
    img.ch.emplace_back(cw, chh, 3, 3);
    img.ch.emplace_back(count, 2, 0, 0);
    img.ch.emplace_back(bw, bh, 0, 0);
    ModularDecode(br, &img, int(1 + 2 * ndc + g), &s->mglobal);
    const size_t tiles_x = DivCeil(d.xsize_blocks, 8);
    for (size_t y = 0; y < chh; y++)
      for (size_t x = 0; x < cw; x++) {
        size_t idx = (by0 / 8 + y) * tiles_x + bx0 / 8 + x;
        o->ytox[idx] = int8_t(std::max(-128, std::min(127, img.ch[0].Row(y)[x])));
        o->ytob[idx] = int8_t(std::max(-128, std::min(127, img.ch[1].Row(y)[x])));
      }
    size_t num = 0;
    const int32_t *r1 = img.ch[2].Row(0), *r2 = img.ch[2].Row(1);
    for (size_t iy = 0; iy < bh; iy++) {
      size_t y = by0 + iy;
      for (size_t ix = 0; ix < bw; ix++) {
        size_t x = bx0 + ix;
        int sharp = img.ch[3].Row(iy)[ix];
        JXLO_CHECK(sharp >= 0 && sharp < 8, "corrupted sharpness field");
        o->sharpness[y * d.xsize_blocks + x] = uint8_t(sharp);
        if (o->acs[y * d.xsize_blocks + x] != 0xFF) continue;
        JXLO_CHECK(num < count, "AC metadata: too few strategies");
        int raw = r1[num];
        JXLO_CHECK(raw >= 0 && raw < 27, "invalid AC strategy");
        o->used_acs |= 1u << raw;
        size_t cx = kCoveredX[raw], cy = kCoveredY[raw];
        size_t nx = (x / 32 + 1) * 32, ny = (y / 32 + 1) * 32;
        JXLO_CHECK(cx * cy == 1 || s->fh.Is444(), "AC strategy not compatible with chroma subsampling");  // dec_modular.cc:534-538
        JXLO_CHECK(x + cx <= nx && x + cx <= std::min(d.xsize_blocks, bx0 + bw), "AC strategy x overflow");
        JXLO_CHECK(y + cy <= ny && y + cy <= std::min(d.ysize_blocks, by0 + bh), "AC strategy y overflow");
        for (size_t jy = 0; jy < cy; jy++)
          for (size_t jx = 0; jx < cx; jx++) {
            uint8_t& a = o->acs[(y + jy) * d.xsize_blocks + x + jx];
            JXLO_CHECK(a == 0xFF, "AC strategy overlap");
            a = uint8_t((raw << 1) | ((jx | jy) == 0 ? 1 : 0));
          }
        o->quant[y * d.xsize_blocks + x] = 1 + std::max(0, std::min(255, r2[num]));
        num++;
      }
    }
  }
}

static void FinalizeDc(FrameState* s) {
  Decoded* o = s->out;
  const FrameDim& d = s->dim;
  const size_t xs = d.xsize_blocks, ys = d.ysize_blocks, plane = xs * ys;
  // EPF sigma (epf.cc:39-133); stored as 1/sigma per 8x8 block
  if (s->fh.lf.epf_iters > 0) {
    const float quant_scale = float(s->global_scale) * (1.0f / 65536.0f);
    o->sigma_params[0] = quant_scale;
    o->sigma_params[1] = s->fh.lf.epf_quant_mul;
    for (int i = 0; i < 8; i++) o->sigma_params[2 + i] = s->fh.lf.epf_sharp_lut[i];
    for (size_t by = 0; by < ys; by++)
      for (size_t bx = 0; bx < xs; bx++) {
        uint8_t a = o->acs[by * xs + bx];
        if (!(a & 1)) continue;
        int st = a >> 1;
        float sigma_quant = s->fh.lf.epf_quant_mul / (quant_scale * float(o->quant[by * xs + bx]) * kInvSigmaNum);
        for (size_t iy = 0; iy < kCoveredY[st]; iy++)
          for (size_t ix = 0; ix < kCoveredX[st]; ix++) {
            float sigma = sigma_quant * s->fh.lf.epf_sharp_lut[o->sharpness[(by + iy) * xs + bx + ix]];
            sigma = std::min(-1e-4f, sigma);
            o->inv_sigma[(by + iy) * xs + bx + ix] = 1.0f / sigma;
          }
      }
  }
  // Adaptive DC smoothing (compressed_dc.cc:128-198)
  if (!(s->fh.flags & (FrameHeader::kSkipDcSmoothing | FrameHeader::kUseDcFrame)) && xs > 2 && ys > 2) {  // (dec_frame.cc:347-356)
    const float inv_global_scale = 65536.0f / float(s->global_scale);
    const float inv_quant_dc = inv_global_scale / float(s->quant_dc);
    float dcf[3];
    for (int c = 0; c < 3; c++) dcf[c] = inv_quant_dc * s->dq.dc_quant[c];
    const float w1 = 0.20345139757231578f, w2 = 0.0334829185968739f, w0 = 1.0f - 4.0f * (w1 + w2);
    std::vector<float> sm(o->dc);
    o->dc_unsmoothed = o->dc;
    for (int c = 0; c < 3; c++) o->dc_step[c] = dcf[c];
    for (size_t y = 1; y + 1 < ys; y++)
      for (size_t x = 1; x + 1 < xs; x++) {
        float mc[3], smv[3], gap = 0.5f;
        for (int c = 0; c < 3; c++) {
          const float* p = o->dc.data() + plane * c;
          const float *t = p + (y - 1) * xs, *m = p + y * xs, *b = p + (y + 1) * xs;
          float corner = (t[x - 1] + t[x + 1]) + (b[x - 1] + b[x + 1]);
          float side = (m[x - 1] + m[x + 1]) + (t[x] + b[x]);
          mc[c] = m[x];
          smv[c] = corner * w2 + (side * w1 + mc[c] * w0);
          gap = std::max(gap, std::fabs((mc[c] - smv[c]) / dcf[c]));
        }
        float factor = -4.0f * gap + 3.0f;
        if (factor < 0) factor = 0;
        for (int c = 0; c < 3; c++) sm[plane * c + y * xs + x] = (smv[c] - mc[c]) * factor + mc[c];
      }
    o->dc.swap(sm);
  }
}

static void DecodeAcGlobal(BitReader& br, FrameState* s) {
  if (s->fh.modular) return;
  if (!br.ReadBool()) {
    // RAW tables (dec_modular.cc:795-841): a small Modular image each, stream id 1 + 3 * num_dc_groups + kind, global tree
    const RawTableReader raw = [s](BitReader& r, size_t w, size_t h, int kind, std::vector<int32_t>* out) {
      MImage img;
      for (int c = 0; c < 3; c++) img.ch.emplace_back(w, h);
      ModularDecode(r, &img, int(1 + 3 * s->dim.num_dc_groups + size_t(kind)), &s->mglobal);
      out->resize(3 * w * h);
      for (int c = 0; c < 3; c++)
        for (size_t y = 0; y < h; y++) memcpy(out->data() + (size_t(c) * h + y) * w, img.ch[c].Row(y), w * sizeof(int32_t));
    };
    for (int k = 0; k < 17; k++) {
      ReadQuantEncoding(br, k, &s->dq.enc[k], &raw);
      s->dq.table[k].clear();
    }
  }
  s->num_histograms = 1 + size_t(br.Read(CeilLog2(s->dim.num_groups)));
  s->orders.resize(s->fh.num_passes);
  s->ac_codes.resize(s->fh.num_passes);
  for (uint32_t p = 0; p < s->fh.num_passes; p++) {
    uint32_t used_orders = ReadU32(br, Val(0x5F), Val(0x13), Val(0), Bits(13));
    DecodeCoeffOrders(br, used_orders, s->out->used_acs, &s->orders[p]);
    size_t nctx = s->num_histograms * s->bctx.NumACContexts();
    DecodeHistograms(br, nctx, &s->ac_codes[p]);
    s->ac_codes[p].ctx_map.resize(nctx + 16, 0);  // slack for out-of-range contexts of invalid streams
  }
}

static inline int32_t PredictNz(const int32_t* top, const int32_t* cur, size_t x) {
  if (x == 0) return top ? top[x] : 32;
  if (!top) return cur[x - 1];
  return (top[x] + cur[x - 1] + 1) / 2;
}

// AC group: entropy-decode quantised coefficients (dec_group.cc:469-542, 594-639), one pass.
static void DecodeAcGroupPass(BitReader& br, FrameState* s, size_t g, uint32_t pass, int32_t* coeffs /*[3][65536]*/,
                              int32_t* nz /*[3][1024]*/) {
  Decoded* o = s->out;
  const FrameDim& d = s->dim;
  const size_t gx = g % d.xsize_groups, gy = g / d.xsize_groups;
  const size_t bx0 = gx * 32, by0 = gy * 32;
  const size_t bw = std::min<size_t>(32, d.xsize_blocks - bx0), bh = std::min<size_t>(32, d.ysize_blocks - by0);
  size_t hbits = CeilLog2(s->num_histograms);
  size_t sel = hbits ? size_t(br.Read(hbits)) : 0;
  JXLO_CHECK(sel < s->num_histograms, "invalid histogram selector");
  const size_t ctx_offset = sel * s->bctx.NumACContexts();
  const EntropyCode& code = s->ac_codes[pass];
  uint64_t symbols = 0;  // added to the frame total once per group (groups may run on OpenMP threads)
  SymbolReader rd(&code, &br);
  const std::vector<uint32_t>& orders = s->orders[pass];
  const uint32_t shift = s->fh.pass_shift[pass];
  size_t offset = 0;
  for (size_t fby = 0; fby < bh; fby++) {
    for (size_t fbx = 0; fbx < bw; fbx++) {
      const size_t bx = fbx, by = fby;
      uint8_t a = o->acs[(by0 + by) * d.xsize_blocks + bx0 + bx];
      if (!(a & 1)) continue;
      const int st = a >> 1;
      const size_t cx = kCoveredX[st], cy = kCoveredY[st];
      const size_t log2c = kLog2Covered[st], covered = size_t(1) << log2c, size = covered * 64;
      const int ord = kStrategyOrder[st];
      const uint32_t qf = uint32_t(o->quant[(by0 + by) * d.xsize_blocks + bx0 + bx]);
      const uint8_t qdc = o->quant_dc[(by0 + by) * d.xsize_blocks + bx0 + bx];
      static const int kChanOrder[3] = {1, 0, 2};
      for (int ci = 0; ci < 3; ci++) {
        const int c = kChanOrder[ci];
        // dec_group.cc:568-578, 619-631: a subsampled channel has a block only where the frame's block lands on its grid; its
        // non-zero counts live on that grid
        const size_t hs = s->fh.hshift[c], vs = s->fh.vshift[c];
        const size_t bx = fbx >> hs, by = fby >> vs;
        if ((bx << hs) != fbx || (by << vs) != fby) continue;
        int32_t* nzc = nz + c * 1024;
        const int32_t* top = by ? nzc + (by - 1) * 32 : nullptr;
        int32_t* cur = nzc + by * 32;
        int32_t predicted = PredictNz(top, cur, bx);
        size_t block_ctx = s->bctx.Context(qdc, qf, ord, c);
        size_t nzctx = s->bctx.NonZeroContext(uint32_t(predicted), block_ctx) + ctx_offset;
        size_t nzeros = rd.Read(nzctx);
        JXLO_CHECK(nzeros <= size - covered, "invalid AC: nzeros too large");
        for (size_t y = 0; y < cy; y++)
          for (size_t x = 0; x < cx; x++) cur[bx + x + y * 32] = int32_t((nzeros + covered - 1) >> log2c);
        const size_t histo_offset = ctx_offset + s->bctx.ZeroDensityOffset(block_ctx);
        const uint32_t* order = &orders[CoeffOrderOffset(ord, c)];
        int32_t* block = coeffs + c * 65536 + offset;
        size_t prev = nzeros > size / 16 ? 0 : 1;
        for (size_t k = covered; k < size && nzeros != 0; ++k) {
          size_t ctx = histo_offset + ZeroDensityContext(nzeros, k, covered, log2c, prev);
          uint32_t u = rd.Read(ctx);
          symbols++;
          uint32_t mag = u >> 1, neg = (~u) & 1;
          int32_t coeff = int32_t((mag ^ (neg - 1)) << shift);
          block[order[k]] += coeff;
          prev = u != 0;
          nzeros -= prev;
        }
        JXLO_CHECK(nzeros == 0, "invalid AC: nzeros at end of block != 0");
      }
      offset += size;
    }
  }
  JXLO_CHECK(rd.FinalStateOk(), "AC group: bad ANS final state");
#pragma omp atomic
  o->ac_symbols += symbols;
}

// Dequantise + CfL + LLF + inverse transform for all varblocks of a group (dec_group.cc:115-181, 433-450).
static void ReconstructGroup(FrameState* s, size_t g, const int32_t* coeffs) {
  Decoded* o = s->out;
  const FrameDim& d = s->dim;
  const size_t gx = g % d.xsize_groups, gy = g / d.xsize_groups;
  const size_t bx0 = gx * 32, by0 = gy * 32;
  const size_t bw = std::min<size_t>(32, d.xsize_blocks - bx0), bh = std::min<size_t>(32, d.ysize_blocks - by0);
  const float inv_global_scale = 65536.0f / float(s->global_scale);
  const float x_dm = std::pow(1.25f, 2.0f - float(s->fh.x_qm_scale)), b_dm = std::pow(1.25f, 2.0f - float(s->fh.b_qm_scale));
  const float color_scale = 1.0f / float(s->color_factor);
  const size_t tiles_x = DivCeil(d.xsize_blocks, 8), plane = d.xsize_blocks * d.ysize_blocks;
  std::vector<float> block;
  size_t offset = 0;
  for (size_t by = 0; by < bh; by++)
    for (size_t bx = 0; bx < bw; bx++) {
      const size_t abx = bx0 + bx, aby = by0 + by;
      uint8_t a = o->acs[aby * d.xsize_blocks + abx];
      if (!(a & 1)) continue;
      const int st = a >> 1;
      const size_t size = (size_t(1) << kLog2Covered[st]) * 64;
      block.assign(3 * size, 0.0f);
      const float scaled = inv_global_scale / float(o->quant[aby * d.xsize_blocks + abx]);
      const float sx = scaled * x_dm, sy = scaled, sb = scaled * b_dm;
      const float x_cc = s->base_corr_x + float(o->ytox[(aby / 8) * tiles_x + abx / 8]) * color_scale;
      const float b_cc = s->base_corr_b + float(o->ytob[(aby / 8) * tiles_x + abx / 8]) * color_scale;
      const float *mx = s->dq.Matrix(st, 0), *my = s->dq.Matrix(st, 1), *mb = s->dq.Matrix(st, 2);
      const float* biases = s->ih->quant_bias;
      for (size_t k = 0; k < size; k++) {
        float dx = AdjustQuantBias(0, coeffs[0 * 65536 + offset + k], biases) * (mx[k] * sx);
        float dy = AdjustQuantBias(1, coeffs[1 * 65536 + offset + k], biases) * (my[k] * sy);
        float db = AdjustQuantBias(2, coeffs[2 * 65536 + offset + k], biases) * (mb[k] * sb);
        block[k] = x_cc * dy + dx;
        block[size + k] = dy;
        block[2 * size + k] = b_cc * dy + db;
      }
      for (int c = 0; c < 3; c++) {
        // dec_group.cc:176-179, 443-451: the lowest frequencies from the channel's own DC sample; pixels only for the blocks
        // a subsampled channel has, at its own position (the top-left part of its plane)
        const size_t hs = s->fh.hshift[c], vs = s->fh.vshift[c], sbx = abx >> hs, sby = aby >> vs;
        LowestFrequenciesFromDC(st, o->dc.data() + plane * c + sby * d.xsize_blocks + sbx, d.xsize_blocks,
                                block.data() + c * size);
        if ((sbx << hs) != abx || (sby << vs) != aby) continue;
        TransformToPixels(st, block.data() + c * size, s->idct.p[c].data() + sby * 8 * s->idct.stride + sbx * 8,
                          s->idct.stride);
      }
      offset += size;
    }
}

// The frame's alpha channel as floats for the patch stage, when the patches blend through it or write it (blending.cc:47-56:
// an image whose only extra channel is alpha; the patch stage sees the channel at the frame's resolution).
static void PatchAlpha(FrameState* s, size_t xs, size_t ys) {
  s->patch_alpha.clear();
  const ImageHeader& ih = *s->ih;
  if (!s->patches.uses_alpha || ih.extra.size() != 1 || ih.extra[0].type != 0) return;
  JXLO_CHECK(s->fh.upsampling == 1 && (s->fh.ec_upsampling.empty() || s->fh.ec_upsampling[0] == 1),
             "unsupported: patches that blend through alpha on upsampled frames");
  const MChannel& ch = s->full.ch[s->modular_color_channels];
  JXLO_CHECK(ch.w == xs && ch.h == ys, "alpha channel size");
  const float af = float(1.0 / double((1u << ih.extra[0].bits) - 1));
  s->patch_alpha.resize(xs * ys);
  for (size_t i = 0; i < xs * ys; i++) s->patch_alpha[i] = float(ch.d[i]) * af;
}

// frame_index / nonvisible_index: the visible frames before this one and the invisible ones since (they seed the noise).
// The frame is rendered at its own size; Decode() places it on the canvas (crop origin, blending with a reference slot).
static void DecodeFrame(BitReader& br, const ImageHeader& ih, Decoded* out, bool want_dumps, size_t frame_index = 0,
                        size_t nonvisible_index = 0, const XybSlot* xyb_slots = nullptr) {
  FrameState st;
  FrameState* s = &st;
  s->out = out;
  s->ih = &ih;
  s->frame_index = frame_index;
  s->nonvisible_index = nonvisible_index;
  s->xyb_slots = xyb_slots;
  ReadFrameHeader(br, ih, &s->fh);
  const FrameHeader& fh = s->fh;
  // (frame types: 1 = kDCFrame: rendered like any frame, kept before the colour transform as the DC image of a later
  // frame, dec_cache.cc:221-224; 2 = kReferenceOnly: kept for patches; 3 = kSkipProgressive)
  JXLO_CHECK(fh.frame_type != 1 || ih.extra.empty(), "unsupported: DC frames of images with extra channels");
  JXLO_CHECK(fh.upsampling == 1 || !fh.modular, "unsupported: upsampled Modular frames");
  JXLO_CHECK(!fh.custom_size || fh.upsampling == 1, "unsupported: cropped upsampled frames");
  JXLO_CHECK(!(fh.ycbcr && fh.modular), "unsupported: YCbCr Modular frames");
  // dec_frame.cc:206-212
  JXLO_CHECK(fh.Is444() || (fh.flags & FrameHeader::kSkipDcSmoothing), "chroma subsampling is not allowed when adaptive DC smoothing is enabled");
  JXLO_CHECK(fh.Is444() || fh.frame_type != 1, "unsupported: chroma-subsampled DC frames");
  JXLO_CHECK(!(fh.modular && ih.xyb_encoded && ih.gray), "unsupported: grey XYB Modular frames");
  s->dim = MakeFrameDim(fh);
  const FrameDim& d = s->dim;
  out->fh = fh;
  out->dim = d;
  const size_t np = fh.num_passes;
  const size_t entries = (d.num_groups == 1 && np == 1) ? 1 : 2 + d.num_dc_groups + d.num_groups * np;
  Toc toc;
  ReadToc(br, entries, &toc);
  const size_t base = br.BitPos() / 8;
  JXLO_CHECK(base + toc.total <= br.size(), "truncated frame");
  const uint8_t* data = br.data();
  const size_t xb = d.xsize_blocks, yb = d.ysize_blocks;
  if (!fh.modular) {
    out->dc.assign(3 * xb * yb, 0.0f);
    if (fh.flags & FrameHeader::kUseDcFrame) {  // passes_state.cc:62-77: the DC image is the DC frame of level dc_level + 1
      JXLO_CHECK(fh.dc_level < 4 && xyb_slots, "invalid DC level for kUseDcFrame");
      const XybSlot& dcf = xyb_slots[4 + fh.dc_level];
      JXLO_CHECK(dcf.w == xb && dcf.h == yb, "kUseDcFrame: no DC frame of that level and size was decoded");
      for (int c = 0; c < 3; c++) memcpy(out->dc.data() + c * xb * yb, dcf.p[c].data(), xb * yb * sizeof(float));
    }
    out->acs.assign(xb * yb, 0xFF);
    out->quant.assign(xb * yb, 0);
    out->sharpness.assign(xb * yb, 0);
    out->quant_dc.assign(xb * yb, 0);
    out->ytox.assign(DivCeil(xb, 8) * DivCeil(yb, 8), 0);
    out->ytob.assign(DivCeil(xb, 8) * DivCeil(yb, 8), 0);
    out->inv_sigma.assign(xb * yb, 0.0f);
    out->coeffs.assign(d.num_groups * 3 * 65536, 0);
    out->nzeros.assign(d.num_groups * 3 * 1024, 0);
    s->idct.Alloc(d.xsize, d.ysize, d.xsize_padded);
    for (auto& v : s->idct.p) v.assign(d.xsize_padded * d.ysize_padded, 0.0f);
  }
  auto check_section = [&](BitReader& r, const char* what) {
    JXLO_CHECK(!r.Overread(), std::string("section over-read: ") + what);
  };
  if (entries == 1) {
    BitReader r(data + base + toc.offset[0], toc.size[0]);
    DecodeDcGlobal(r, s);
    DecodeDcGroup(r, s, 0);
    if (!fh.modular) FinalizeDc(s);
    DecodeAcGlobal(r, s);
    if (!fh.modular) DecodeAcGroupPass(r, s, 0, 0, out->coeffs.data(), out->nzeros.data());
    DecodeModularGroup(r, s, 0, 0, d.group_dim, d.group_dim, 0, 2, int(1 + 3 * d.num_dc_groups + 17 + 0));
    check_section(r, "single");
  } else {
    {
      BitReader r(data + base + toc.offset[0], toc.size[0]);
      DecodeDcGlobal(r, s);
      check_section(r, "DC global");
    }
    for (size_t g = 0; g < d.num_dc_groups; g++) {
      BitReader r(data + base + toc.offset[1 + g], toc.size[1 + g]);
      DecodeDcGroup(r, s, g);
      check_section(r, "DC group");
    }
    if (!fh.modular) FinalizeDc(s);
    {
      size_t i = 1 + d.num_dc_groups;
      BitReader r(data + base + toc.offset[i], toc.size[i]);
      DecodeAcGlobal(r, s);
      check_section(r, "AC global");
    }
    // A frame drawn from a prefix of its bytes (FrameDecoder::Flush, dec_frame.cc:735-795 with dec_frame.cc:620-680
    // ProcessSections: a group's passes are decoded in order as their sections arrive, decoded_passes_per_ac_group_, and a
    // flush draws every group from the passes it has, none = the DC image alone; decode.cc:2458-2475 JxlDecoderFlushImage):
    // g_flush_prefix = number of bytes of the codestream that are there; 0 = all.
    std::vector<uint32_t> passes_there(d.num_groups, uint32_t(np));
    if (g_flush_prefix)
      for (size_t g = 0; g < d.num_groups; g++) {
        uint32_t k = 0;
        for (; k < np; k++) {
          const size_t i = 2 + d.num_dc_groups + k * d.num_groups + g;
          if (base + toc.offset[i] + toc.size[i] > g_flush_prefix) break;
        }
        passes_there[g] = k;
      }
    for (size_t p = 0; p < np; p++) {
      // Downsampling bracket (frame_header.h:268-284) for streams without progressive-downsampling info:
      // the last pass carries shifts 0..2, earlier passes carry no Modular data.
      const int min_shift = 0, max_shift = 2;
      // Groups are independent (OpenMP threads when the caller sets OMP_NUM_THREADS > 1: bench.py's cpu_baseline; the
      // tests run it serially or not, the results are identical). Frames with Modular channels keep the serial order.
      std::string group_error;
      const bool parallel = !fh.modular && s->full.ch.empty();
#pragma omp parallel for schedule(dynamic) if (parallel)
      for (size_t g = 0; g < d.num_groups; g++) {
        if (p >= passes_there[g]) continue;
        try {
          size_t i = 2 + d.num_dc_groups + p * d.num_groups + g;
          BitReader r(data + base + toc.offset[i], toc.size[i]);
          if (!fh.modular)
            DecodeAcGroupPass(r, s, g, uint32_t(p), out->coeffs.data() + g * 3 * 65536, out->nzeros.data() + g * 3 * 1024);
          size_t gx = g % d.xsize_groups, gy = g / d.xsize_groups;
          if (np == 1 || p + 1 == np)
            DecodeModularGroup(r, s, gx * d.group_dim, gy * d.group_dim, d.group_dim, d.group_dim, min_shift, max_shift,
                               int(1 + 3 * d.num_dc_groups + 17 + d.num_groups * p + g));
          check_section(r, "AC group");
        } catch (const std::exception& e) {
#pragma omp critical
          group_error = e.what();
        }
      }
      if (!group_error.empty()) throw Error(group_error);
    }
  }
  br.Skip(base * 8 + toc.total * 8 - br.BitPos());
  // Undo the global modular transforms
  for (size_t i = s->full.transforms.size(); i-- > 0;) InverseTransform(&s->full, s->full.transforms[i]);
  s->full.transforms.clear();

  // ---- render (xs, ys: the image; the frame is ceil(image / upsampling))
  const size_t xs = fh.upsampling == 1 ? d.xsize : ih.xsize, ys = fh.upsampling == 1 ? d.ysize : ih.ysize;
  out->out_xsize = xs;
  out->out_ysize = ys;
  const bool has_alpha = !ih.extra.empty() && ih.extra[0].type == 0;
  out->out_channels = has_alpha ? 4 : 3;
  out->rgbf.assign(3 * xs * ys, 0.0f);
  if (!fh.modular) {
    // The dequant matrices are built on first use: build them all before the groups run in parallel.
    // (An invalid table is only an error when a block uses it: dequant_matrices.h EnsureComputed(acs_mask).)
    for (int st = 0; st < 27; st++) {
      try {
        s->dq.Matrix(st, 0);
      } catch (const std::exception&) {
      }
    }
    std::string group_error;
#pragma omp parallel for schedule(dynamic)
    for (size_t g = 0; g < d.num_groups; g++) {
      try {
        ReconstructGroup(s, g, out->coeffs.data() + g * 3 * 65536);
      } catch (const std::exception& e) {
#pragma omp critical
        group_error = e.what();
      }
    }
    if (!group_error.empty()) throw Error(group_error);
    if (!fh.Is444()) {
      // dec_cache.cc:138-150: per channel the horizontal, then the vertical chroma upsampling, in front of every other stage
      // (the dump below then holds what the filters read)
      for (int c = 0; c < 3; c++) ChromaUpsample(&s->idct.p[c], s->idct.stride, d.xsize, d.ysize, d.ysize_padded, fh.hshift[c] != 0, fh.vshift[c] != 0);
    }
    if (want_dumps) {
      out->xyb_idct.resize(3 * d.xsize_padded * d.ysize_padded);
      for (int c = 0; c < 3; c++)
        memcpy(out->xyb_idct.data() + c * d.xsize_padded * d.ysize_padded, s->idct.p[c].data(),
               d.xsize_padded * d.ysize_padded * sizeof(float));
    }
    Planes3 a, b;
    const Planes3* cur = &s->idct;
    if (fh.lf.gab) {
      Gaborish(*cur, fh.lf, &a);
      cur = &a;
    }
    auto run_epf = [&](int stage) {
      Planes3* dst = (cur == &a) ? &b : &a;
      EpfPass(stage, *cur, fh.lf, out->inv_sigma, xb, dst);
      cur = dst;
    };
    if (fh.lf.epf_iters >= 3) run_epf(0);
    if (fh.lf.epf_iters >= 1) run_epf(1);
    if (fh.lf.epf_iters >= 2) run_epf(2);
    if (want_dumps) {
      out->xyb_filtered.resize(3 * d.xsize_padded * d.ysize);
      for (int c = 0; c < 3; c++)
        memcpy(out->xyb_filtered.data() + c * d.xsize_padded * d.ysize, cur->p[c].data(), d.xsize_padded * d.ysize * sizeof(float));
    }
    Planes3 patched;
    if (s->has_patches) {  // dec_cache.cc:193-197: patches, then splines
      patched = *cur;
      PatchAlpha(s, d.xsize, d.ysize);
      ApplyPatches(s->patches, s->xyb_slots, patched.p[0].data(), patched.p[1].data(), patched.p[2].data(), patched.stride,
                   s->patch_alpha.empty() ? nullptr : s->patch_alpha.data(), d.xsize, !ih.extra.empty() && ih.extra[0].alpha_associated);
      cur = &patched;
      if (want_dumps)
        for (int c = 0; c < 3; c++)
          memcpy(out->xyb_filtered.data() + c * d.xsize_padded * d.ysize, cur->p[c].data(), d.xsize_padded * d.ysize * sizeof(float));
    }
    Planes3 splined;
    if (s->has_splines) {  // dec_cache.cc:198-201: after the filters, before upsampling and noise
      InitSplineDrawCache(&s->splines, d.xsize, d.ysize, s->base_corr_x, s->base_corr_b);  // dec_frame.cc:303-308
      splined = *cur;
      DrawSplines(s->splines, splined.p[0].data(), splined.p[1].data(), splined.p[2].data(), splined.stride, d.xsize, d.ysize);
      cur = &splined;
      if (want_dumps)
        for (int c = 0; c < 3; c++)
          memcpy(out->xyb_filtered.data() + c * d.xsize_padded * d.ysize, cur->p[c].data(), d.xsize_padded * d.ysize * sizeof(float));
    }
    Planes3 up;
    if (fh.upsampling != 1) {
      Planes3 crop;  // the filters work on the padded stride; the upsampler mirrors about the frame size
      crop.xs = d.xsize; crop.ys = d.ysize; crop.stride = cur->stride;
      for (int c = 0; c < 3; c++) crop.p[c] = cur->p[c];
      const std::vector<float>& cw = fh.upsampling == 2 ? ih.ups_weights2 : (fh.upsampling == 4 ? ih.ups_weights4 : ih.ups_weights8);
      Upsample(crop, fh.upsampling, cw.empty() ? nullptr : cw.data(), xs, ys, &up);
      cur = &up;
    }
    Planes3 noisy;
    if (s->has_noise) {  // (the reference skips the stage when every LUT point is below 1e-3: noise.h:35-40)
      bool any = false;
      for (float v : s->noise_lut) any = any || std::fabs(v) > 1e-3f;
      if (any) {
        noisy = *cur;
        AddNoise(&noisy, xs, ys, s->noise_lut, s->base_corr_x, s->base_corr_b, uint32_t(s->frame_index), uint32_t(s->nonvisible_index));
        cur = &noisy;
        if (want_dumps && fh.upsampling == 1) {  // (the dump then holds what the colour conversion reads)
          for (int c = 0; c < 3; c++)
            memcpy(out->xyb_filtered.data() + c * d.xsize_padded * d.ysize, cur->p[c].data(), d.xsize_padded * d.ysize * sizeof(float));
        }
      }
    }
    if (fh.frame_type == 2 || fh.frame_type == 1 || fh.save_before_color_transform) {  // what a later frame's patches / DC read
      out->xyb_save.resize(3 * xs * ys);
      for (int c = 0; c < 3; c++)
        for (size_t y = 0; y < ys; y++) memcpy(out->xyb_save.data() + (c * ys + y) * xs, cur->p[c].data() + y * cur->stride, xs * sizeof(float));
    }
    OpsinParams op = MakeOpsinParams(ih);
#pragma omp parallel for schedule(static)
    for (size_t y = 0; y < ys; y++)
      for (size_t x = 0; x < xs; x++) {
        size_t i = y * cur->stride + x;
        float r, g, bb;
        if (!ih.xyb_encoded) {
          // dec_cache.cc:256-263: kYCbCr -> stage_ycbcr.cc:41-60 (full-range BT.601 of JFIF; channels Cb, Y, Cr; fused
          // multiply-adds like the reference's MulAdd), kNone -> nothing; no transfer function follows either
          const float c0 = cur->p[0][i], c1 = cur->p[1][i], c2 = cur->p[2][i];
          if (fh.ycbcr) {
            const float yv = c1 + 128.0f / 255;
            r = std::fma(1.402f, c2, yv);
            g = std::fma(-0.299f * 1.402f / 0.587f, c2, std::fma(-0.114f * 1.772f / 0.587f, c0, yv));
            bb = std::fma(1.772f, c0, yv);
          } else {
            r = c0;
            g = c1;
            bb = c2;
          }
        } else {
        XybToRgb(op, cur->p[0][i], cur->p[1][i], cur->p[2][i], &r, &g, &bb);
        if (!ih.linear_tf) {
          r = LinearToSrgb(r);
          g = LinearToSrgb(g);
          bb = LinearToSrgb(bb);
        }
        }
        out->rgbf[0 * xs * ys + y * xs + x] = r;
        out->rgbf[1 * xs * ys + y * xs + x] = g;
        out->rgbf[2 * xs * ys + y * xs + x] = bb;
      }
  } else {
    const float factor = float(1.0 / double((1u << s->full.bitdepth) - 1));
    const size_t ncol = s->modular_color_channels;
    if (ih.xyb_encoded) {
      // dec_modular.cc:583-631: an XYB Modular frame codes Y, X, B - Y as integers in units of the DC quantisation steps
      const int32_t* cy = s->full.ch[0].d.data();
      const int32_t* cx = s->full.ch[1].d.data();
      const int32_t* cb = s->full.ch[2].d.data();
      for (size_t i = 0; i < xs * ys; i++) {
        out->rgbf[i] = float(cx[i]) * s->dq.dc_quant[0];
        out->rgbf[xs * ys + i] = float(cy[i]) * s->dq.dc_quant[1];
        out->rgbf[2 * xs * ys + i] = float(cb[i] + cy[i]) * s->dq.dc_quant[2];
      }
    } else {
      for (int c = 0; c < 3; c++) {
        const MChannel& ch = s->full.ch[ncol == 1 ? 0 : c];
        for (size_t i = 0; i < xs * ys; i++) out->rgbf[c * xs * ys + i] = float(ch.d[i]) * factor;
      }
    }
    if (s->has_patches) {
      PatchAlpha(s, xs, ys);
      ApplyPatches(s->patches, s->xyb_slots, out->rgbf.data(), out->rgbf.data() + xs * ys, out->rgbf.data() + 2 * xs * ys, xs,
                   s->patch_alpha.empty() ? nullptr : s->patch_alpha.data(), xs, !ih.extra.empty() && ih.extra[0].alpha_associated);
    }
    if (s->has_splines) {  // the same stage on the three colour channels of a Modular frame (default colour correlation: 0, 1)
      InitSplineDrawCache(&s->splines, xs, ys, s->base_corr_x, s->base_corr_b);
      DrawSplines(s->splines, out->rgbf.data(), out->rgbf.data() + xs * ys, out->rgbf.data() + 2 * xs * ys, xs, xs, ys);
    }
    if (fh.frame_type == 2 || fh.frame_type == 1 || fh.save_before_color_transform) out->xyb_save = out->rgbf;
    if (ih.xyb_encoded) {  // then the colour stage of every XYB frame
      OpsinParams op = MakeOpsinParams(ih);
      for (size_t i = 0; i < xs * ys; i++) {
        float r, g, bb;
        XybToRgb(op, out->rgbf[i], out->rgbf[xs * ys + i], out->rgbf[2 * xs * ys + i], &r, &g, &bb);
        if (!ih.linear_tf) {
          r = LinearToSrgb(r);
          g = LinearToSrgb(g);
          bb = LinearToSrgb(bb);
        }
        out->rgbf[i] = r;
        out->rgbf[xs * ys + i] = g;
        out->rgbf[2 * xs * ys + i] = bb;
      }
    }
    if (want_dumps) {
      out->modular.resize(s->full.ch.size() * xs * ys);
      for (size_t c = 0; c < s->full.ch.size(); c++)
        memcpy(out->modular.data() + c * xs * ys, s->full.ch[c].d.data(), xs * ys * sizeof(int32_t));
    }
  }
  const int oc = out->out_channels;
  out->rgb8.resize(xs * ys * oc);
  std::vector<float> alpha;
  if (has_alpha) {
    const MChannel& ch = s->full.ch[s->modular_color_channels];
    const float af = float(1.0 / double((1u << ih.extra[0].bits) - 1));
    const uint32_t ecu = fh.ec_upsampling.empty() ? 1 : fh.ec_upsampling[0];
    if (!s->patch_alpha.empty()) {  // the patches have written the alpha channel too
      alpha = s->patch_alpha;
    } else if (ecu == 1) {
      alpha.resize(xs * ys);
      for (size_t i = 0; i < xs * ys; i++) alpha[i] = float(ch.d[i]) * af;
    } else {
      // dec_cache.cc:172-190, 203-212: the channel as floats (dec_modular.cc:640-700) through the upsampling stage of its own
      // factor, cropped to the image
      Planes3 in, up;
      in.xs = ch.w; in.ys = ch.h; in.stride = ch.w;
      in.p[0].resize(ch.w * ch.h);
      for (size_t i = 0; i < ch.w * ch.h; i++) in.p[0][i] = float(ch.d[i]) * af;
      const std::vector<float>& cw = ecu == 2 ? ih.ups_weights2 : (ecu == 4 ? ih.ups_weights4 : ih.ups_weights8);
      Upsample(in, ecu, cw.empty() ? nullptr : cw.data(), xs, ys, &up, 1);
      alpha = up.p[0];
    }
    out->alphaf = alpha;
  }
#pragma omp parallel for schedule(static)
  for (size_t y = 0; y < ys; y++)
    for (size_t x = 0; x < xs; x++) {
      for (int c = 0; c < 3; c++) out->rgb8[(y * xs + x) * oc + c] = ToU8(out->rgbf[c * xs * ys + y * xs + x], x, y, c);
      if (has_alpha) out->rgb8[(y * xs + x) * oc + 3] = ToU8(alpha[y * xs + x], x, y, 3);
    }
}

static void Decode(const uint8_t* data, size_t size, Decoded* out, bool want_dumps, size_t frame_index = 0, bool want_preview = false) {
  // bare codestream, or a container whose first codestream box is `jxlc`
  static const uint8_t kContainer[12] = {0, 0, 0, 0xC, 'J', 'X', 'L', ' ', 0xD, 0xA, 0x87, 0xA};
  if (size >= 12 && !memcmp(data, kContainer, 12)) {
    size_t pos = 12;
    bool found = false;
    while (pos + 8 <= size) {
      uint64_t bsize = (uint64_t(data[pos]) << 24) | (data[pos + 1] << 16) | (data[pos + 2] << 8) | data[pos + 3];
      size_t hdr = 8;
      if (bsize == 1) {
        JXLO_CHECK(pos + 16 <= size, "truncated box");
        bsize = 0;
        for (int i = 0; i < 8; i++) bsize = (bsize << 8) | data[pos + 8 + i];
        hdr = 16;
      }
      if (bsize == 0) bsize = size - pos;
      JXLO_CHECK(bsize >= hdr && pos + bsize <= size, "bad box size");
      if (!memcmp(data + pos + 4, "jxlc", 4)) {
        data += pos + hdr;
        size = bsize - hdr;
        found = true;
        break;
      }
      JXLO_CHECK(memcmp(data + pos + 4, "jxlp", 4) != 0, "unsupported: jxlp boxes");
      pos += bsize;
    }
    JXLO_CHECK(found, "no codestream box");
  }
  JXLO_CHECK(size >= 2 && data[0] == 0xFF && data[1] == 0x0A, "not a JPEG XL codestream");
  BitReader br(data, size);
  br.Skip(16);
  ReadImageHeader(br, &out->ih);
  // The canvas (blending.cc, render_pipeline/stage_blending.cc, dec_cache.cc:268-290): every frame is blended, in the
  // output colour space, with the reference slot its header names -- outside the frame's rectangle the result is that
  // slot's content (zeros when it was never written) -- and frames that can be referenced store the result in a slot.
  // frame_index counts the frames that are shown (decode.cc:1346-1350: the last one, or one with a duration).
  const ImageHeader ih = out->ih;
  const size_t W = ih.xsize, H = ih.ysize;
  const bool has_alpha = !ih.extra.empty() && ih.extra[0].type == 0;
  struct Slot {
    std::vector<float> p[4];
    bool valid = false;
  } slots[4];
  XybSlot xyb_slots[8];  // 0..3 reference slots, 4..7 the DC frames of level 1..4 (passes_state.h:90)
  size_t visible = 0, nonvisible = 0;
  JXLO_CHECK(!want_preview || ih.have_preview, "the image has no preview");
  if (ih.have_preview) {
    // The codestream's first frame is the preview (decode.cc:1266-1268): a regular frame whose default size is the
    // preview size (frame_header.h:450-463), never blended (frame_header.cc:372-376), counted among the frames that are
    // not shown (dec_frame.cc:160-168).
    ImageHeader pih = ih;
    pih.xsize = ih.preview_xsize;
    pih.ysize = ih.preview_ysize;
    DecodeFrame(br, pih, out, want_dumps && want_preview, visible, nonvisible, xyb_slots);
    JXLO_CHECK(out->fh.frame_type == 0 && !out->fh.custom_size && out->fh.blend.mode == 0, "invalid preview frame");
    if (want_preview) return;
    nonvisible++;
    *out = Decoded();
    out->ih = ih;
  }
  for (;;) {
    DecodeFrame(br, ih, out, want_dumps && visible == frame_index, visible, nonvisible, xyb_slots);
    const FrameHeader fh = out->fh;
    if (fh.frame_type == 2 || (!fh.is_last && fh.save_before_color_transform)) {  // dec_frame.cc FinalizeFrame: kept in XYB
      XybSlot& slot = xyb_slots[fh.save_as_reference];
      slot.w = out->out_xsize;
      slot.h = out->out_ysize;
      for (int c = 0; c < 3; c++) slot.p[c].assign(out->xyb_save.begin() + c * slot.w * slot.h, out->xyb_save.begin() + (c + 1) * slot.w * slot.h);
      slot.alpha = out->alphaf;  // (the frame's extra channels are kept with it: dec_patch_dictionary.cc:342-347)
    }
    if (fh.frame_type == 1) {
      XybSlot& slot = xyb_slots[4 + fh.dc_level - 1];
      slot.w = out->out_xsize;
      slot.h = out->out_ysize;
      for (int c = 0; c < 3; c++) slot.p[c].assign(out->xyb_save.begin() + c * slot.w * slot.h, out->xyb_save.begin() + (c + 1) * slot.w * slot.h);
    }
    if (fh.frame_type == 2 || fh.frame_type == 1) {  // never shown, never blended
      nonvisible++;
      *out = Decoded();
      out->ih = ih;
      continue;
    }
    const bool shown = fh.is_last || fh.duration > 0;
    bool needs_blending = fh.custom_size || fh.blend.mode != 0;
    for (const BlendInfo& e : fh.ec_blend) needs_blending = needs_blending || e.mode != 0;
    const bool can_ref = !fh.is_last && (fh.duration == 0 || fh.save_as_reference != 0);
    if (needs_blending || can_ref || !shown || visible != 0 || nonvisible != 0) {
      JXLO_CHECK(ih.extra.size() <= 1 && (ih.extra.empty() || has_alpha), "unsupported: blending with extra channels other than alpha");
      JXLO_CHECK(!(can_ref && fh.save_before_color_transform), "unsupported: frames saved before the colour transform");
      const size_t fw = out->out_xsize, fhh = out->out_ysize;
      const BlendInfo cb = fh.blend, ab = has_alpha ? fh.ec_blend[0] : BlendInfo();
      const bool premul = has_alpha && ih.extra[0].alpha_associated;
      Slot cur;
      for (auto& v : cur.p) v.assign(W * H, 0.0f);
      const Slot& bgc = slots[cb.source];
      const Slot& bga = slots[ab.source];
      auto clamp01 = [](float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); };
      for (size_t y = 0; y < H; y++)
        for (size_t x = 0; x < W; x++) {
          const size_t i = y * W + x;
          float bg[4] = {0, 0, 0, 0};
          if (bgc.valid)
            for (int c = 0; c < 3; c++) bg[c] = bgc.p[c][i];
          if (bga.valid) bg[3] = bga.p[3][i];
          const long long fx = (long long)x - fh.x0, fy = (long long)y - fh.y0;
          if (fx < 0 || fy < 0 || fx >= (long long)fw || fy >= (long long)fhh) {  // padding: the background shows
            for (int c = 0; c < 4; c++) cur.p[c][i] = bg[c];
            continue;
          }
          const size_t j = size_t(fy) * fw + size_t(fx);
          float fg[4];
          for (int c = 0; c < 3; c++) fg[c] = out->rgbf[c * fw * fhh + j];
          fg[3] = has_alpha ? out->alphaf[j] : 1.0f;
          float o[4];
          // the alpha channel first, from the alpha values before blending (blending.cc:51-112; alpha.cc:44-88)
          if (has_alpha) {
            const float fa = ab.clamp ? clamp01(fg[3]) : fg[3];
            switch (ab.mode) {
              case 1: o[3] = bg[3] + fg[3]; break;
              case 2: o[3] = 1.0f - (1.0f - fa) * (1.0f - bg[3]); break;
              case 3: o[3] = bg[3]; break;
              case 4: o[3] = bg[3] * fa; break;
              default: o[3] = fg[3]; break;
            }
          } else {
            o[3] = 1.0f;
          }
          const float fa = cb.clamp ? clamp01(fg[3]) : fg[3];
          switch (cb.mode) {
            case 1:
              for (int c = 0; c < 3; c++) o[c] = bg[c] + fg[c];
              break;
            case 3:
              for (int c = 0; c < 3; c++) o[c] = has_alpha ? bg[c] + fg[c] * fa : bg[c] + fg[c];
              break;
            case 2:
              if (!has_alpha) {
                for (int c = 0; c < 3; c++) o[c] = fg[c];
              } else if (premul) {
                for (int c = 0; c < 3; c++) o[c] = fg[c] + bg[c] * (1.0f - fa);
                o[3] = 1.0f - (1.0f - fa) * (1.0f - bg[3]);
              } else {
                const float new_a = 1.0f - (1.0f - fa) * (1.0f - bg[3]);
                const float rnew_a = new_a > 0 ? 1.0f / new_a : 0.0f;
                for (int c = 0; c < 3; c++) o[c] = (fg[c] * fa + bg[c] * bg[3] * (1.0f - fa)) * rnew_a;
                o[3] = new_a;
              }
              break;
            case 4:
              for (int c = 0; c < 3; c++) o[c] = bg[c] * (cb.clamp ? clamp01(fg[c]) : fg[c]);
              break;
            default:
              for (int c = 0; c < 3; c++) o[c] = fg[c];
              break;
          }
          for (int c = 0; c < 4; c++) cur.p[c][i] = o[c];
        }
      cur.valid = true;
      if (can_ref) slots[fh.save_as_reference] = cur;
      // the shown image is the canvas
      out->out_xsize = W;
      out->out_ysize = H;
      out->rgbf.resize(3 * W * H);
      for (int c = 0; c < 3; c++) memcpy(out->rgbf.data() + c * W * H, cur.p[c].data(), W * H * sizeof(float));
      if (has_alpha) out->alphaf = cur.p[3];
      const int oc = out->out_channels;
      out->rgb8.resize(W * H * oc);
      for (size_t y = 0; y < H; y++)
        for (size_t x = 0; x < W; x++) {
          for (int c = 0; c < 3; c++) out->rgb8[(y * W + x) * oc + c] = ToU8(out->rgbf[c * W * H + y * W + x], x, y, c);
          if (has_alpha) out->rgb8[(y * W + x) * oc + 3] = ToU8(out->alphaf[y * W + x], x, y, 3);
        }
    }
    if (shown) {
      if (visible == frame_index) break;
      visible++;
      nonvisible = 0;
    } else {
      nonvisible++;
    }
    JXLO_CHECK(!fh.is_last, "no such frame");
    *out = Decoded();  // (frames before the wanted one are decoded and dropped: test sizes)
    out->ih = ih;
  }
}

}  // namespace jxlo

// ------------------------------------------------------------------ C interface (ctypes / tests)
extern "C" {

struct JxloHandle {
  jxlo::Decoded d;
  std::string error;
};

// flags: bit0 = keep intermediate dumps, bits 8.. = index of the frame to decode (animations). Returns a handle (never NULL);
// check jxlo_error().
void jxlo_set_flush_prefix(size_t nbytes) { jxlo::g_flush_prefix = nbytes; }
JxloHandle* jxlo_decode(const uint8_t* data, size_t size, int flags) {
  JxloHandle* h = new JxloHandle;
  try {
    jxlo::Decode(data, size, &h->d, (flags & 1) != 0, size_t(flags) >> 8, (flags & 2) != 0);
  } catch (const std::exception& e) {
    h->error = e.what();
    if (h->error.empty()) h->error = "unknown error";
  }
  return h;
}
const char* jxlo_error(JxloHandle* h) { return h->error.empty() ? nullptr : h->error.c_str(); }
void jxlo_free(JxloHandle* h) { delete h; }
// info[0..15]: xsize, ysize, out_channels, is_modular, xsize_blocks, ysize_blocks, xsize_padded, ysize_padded,
// num_groups, num_dc_groups, epf_iters, gab, num_passes, used_acs, bits, ac_symbols(low 32)
// Threads for the group / row loops (bench.py's cpu_baseline reports this as "cores"); returns the count in effect.
int jxlo_set_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
}
void jxlo_out_size(JxloHandle* h, uint32_t* wh) {
  wh[0] = uint32_t(h->d.out_xsize);
  wh[1] = uint32_t(h->d.out_ysize);
}
void jxlo_info(JxloHandle* h, uint32_t* info) {
  const jxlo::Decoded& d = h->d;
  info[0] = uint32_t(d.dim.xsize);
  info[1] = uint32_t(d.dim.ysize);
  info[2] = uint32_t(d.out_channels);
  info[3] = d.fh.modular;
  info[4] = uint32_t(d.dim.xsize_blocks);
  info[5] = uint32_t(d.dim.ysize_blocks);
  info[6] = uint32_t(d.dim.xsize_padded);
  info[7] = uint32_t(d.dim.ysize_padded);
  info[8] = uint32_t(d.dim.num_groups);
  info[9] = uint32_t(d.dim.num_dc_groups);
  info[10] = d.fh.lf.epf_iters;
  info[11] = d.fh.lf.gab;
  info[12] = d.fh.num_passes;
  info[13] = d.used_acs;
  info[14] = d.ih.bits;
  info[15] = uint32_t(d.ac_symbols);
}
// Animation: {have_animation, tps numerator, tps denominator, loops, duration of the decoded frame, its is_last, its timecode}
void jxlo_animation(JxloHandle* h, uint32_t* a) {
  const jxlo::Decoded& d = h->d;
  a[0] = d.ih.have_animation;
  a[1] = d.ih.anim_tps_num;
  a[2] = d.ih.anim_tps_den;
  a[3] = d.ih.anim_loops;
  a[4] = d.fh.duration;
  a[5] = d.fh.is_last;
  a[6] = d.fh.timecode;
}
// Named buffers: "rgb8","rgbf","coeffs","nzeros","xyb_idct","xyb_filtered","dc","acs","quant","sharpness","ytox",
// "ytob","inv_sigma","quant_dc","modular". Returns pointer and byte size (0 if absent).
const void* jxlo_buffer(JxloHandle* h, const char* name, size_t* nbytes) {
  jxlo::Decoded& d = h->d;
  std::string n(name);
#define JXLO_BUF(field)                                         \
  if (n == #field) {                                            \
    *nbytes = d.field.size() * sizeof(d.field[0]);              \
    return d.field.empty() ? nullptr : (const void*)d.field.data(); \
  }
  JXLO_BUF(rgb8) JXLO_BUF(rgbf) JXLO_BUF(coeffs) JXLO_BUF(nzeros) JXLO_BUF(xyb_idct) JXLO_BUF(xyb_filtered) JXLO_BUF(dc)
  JXLO_BUF(acs) JXLO_BUF(quant) JXLO_BUF(sharpness) JXLO_BUF(ytox) JXLO_BUF(ytob) JXLO_BUF(inv_sigma) JXLO_BUF(quant_dc) JXLO_BUF(dc_unsmoothed)
  JXLO_BUF(modular) JXLO_BUF(alphaf)
#undef JXLO_BUF
  *nbytes = 0;
  return nullptr;
}

// Scalars of the DC path for the tests' float64 reading of it: out[0..2] = DC quantisation steps of X, Y, B,
// out[3] = quant_scale, out[4] = epf_quant_mul, out[5..12] = epf_sharp_lut.
void jxlo_dc_params(JxloHandle* h, float* out) {
  for (int i = 0; i < 3; i++) out[i] = h->d.dc_step[i];
  for (int i = 0; i < 10; i++) out[3 + i] = h->d.sigma_params[i];
}

// Known-answer hook for the colour stage: n XYB triples, planar [3][n], through XybToRgb (+ the sRGB transfer function
// unless linear) with the default opsin parameters; interleaved RGB out (lib/jxl/opsin_image_test.cc's closed forms).
void jxlo_color_kat(const float* xyb, size_t n, int linear, float* rgb) {
  jxlo::ImageHeader ih;
  const jxlo::OpsinParams op = jxlo::MakeOpsinParams(ih);
  for (size_t i = 0; i < n; i++) {
    float r, g, b;
    jxlo::XybToRgb(op, xyb[i], xyb[n + i], xyb[2 * n + i], &r, &g, &b);
    if (!linear) {
      r = jxlo::LinearToSrgb(r);
      g = jxlo::LinearToSrgb(g);
      b = jxlo::LinearToSrgb(b);
    }
    rgb[3 * i] = r;
    rgb[3 * i + 1] = g;
    rgb[3 * i + 2] = b;
  }
}

// Known-answer hook for the patch blending (lib/jxl/alpha_test.cc holds expected values for the blend and multiply arithmetic):
// one pixel, colour bg[3] / fg[3] with alpha bga / fga, through ApplyPatches with the given colour and alpha-channel modes;
// out[0..2] = colour, out[3] = alpha.
void jxlo_patch_blend_kat(const float* bg, float bga, const float* fg, float fga, int mode, int clamp, int ec_mode, int ec_clamp,
                          int premultiplied, int has_alpha, float* out) {
  jxlo::XybSlot slots[4];
  slots[1].w = slots[1].h = 1;
  for (int c = 0; c < 3; c++) slots[1].p[c].assign(1, fg[c]);
  slots[1].alpha.assign(1, fga);
  jxlo::Patches P;
  P.refs.push_back({1, 0, 0, 1, 1});
  jxlo::PatchPos q;
  q.x = q.y = q.ref = 0;
  q.mode = uint32_t(mode);
  q.clamp = clamp != 0;
  q.ec_mode = uint32_t(ec_mode);
  q.ec_clamp = ec_clamp != 0;
  P.pos.push_back(q);
  P.uses_alpha = true;
  float p0 = bg[0], p1 = bg[1], p2 = bg[2], a = bga;
  jxlo::ApplyPatches(P, slots, &p0, &p1, &p2, 1, has_alpha ? &a : nullptr, 1, premultiplied != 0);
  out[0] = p0;
  out[1] = p1;
  out[2] = p2;
  out[3] = a;
}

// Known-answer hook for the chroma upsampling of subsampled YCbCr frames: `plane` holds ceil(xsize / 2) x ceil(ysize / 2) (or
// xsize / ysize in the direction that is not subsampled) samples in its top-left part, rows of `stride` floats, `rows` rows in
// all; upsampled in place (render_pipeline/stage_chroma_upsampling.cc:29-111).
void jxlo_chroma_upsample_kat(float* plane, size_t stride, size_t xsize, size_t ysize, size_t rows, int horizontal, int vertical) {
  std::vector<float> p(plane, plane + stride * rows);
  jxlo::ChromaUpsample(&p, stride, xsize, ysize, rows, horizontal != 0, vertical != 0);
  memcpy(plane, p.data(), p.size() * sizeof(float));
}

// Known-answer hook for the noise generator: `vectors` steps of the single-seed generator, 8 values each
// (lib/jxl/xorshift128plus_test.cc:60-257 holds the expected values for seed 12345).
void jxlo_xorshift_fill(uint64_t seed, uint64_t* out, size_t vectors) {
  jxlo::Xorshift128Plus rng(seed);
  for (size_t i = 0; i < vectors; i++) rng.Fill(out + 8 * i);
}

}  // extern "C"

#ifdef JXLO_MAIN
#include <chrono>
int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s in.jxl [out.ppm] [reps]\n", argv[0]);
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  size_t n;
  while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  fclose(f);
  int reps = argc > 3 ? atoi(argv[3]) : 1;
  for (int r = 0; r < reps; r++) {
    auto t0 = std::chrono::steady_clock::now();
    JxloHandle* h = jxlo_decode(buf.data(), buf.size(), 0);
    auto t1 = std::chrono::steady_clock::now();
    if (jxlo_error(h)) {
      fprintf(stderr, "error: %s\n", jxlo_error(h));
      return 1;
    }
    uint32_t info[16];
    jxlo_info(h, info);
    double sec = std::chrono::duration<double>(t1 - t0).count();
    fprintf(stderr, "%ux%u decoded in %.3f s = %.2f MP/s\n", info[0], info[1], sec, info[0] * double(info[1]) * 1e-6 / sec);
    if (r == reps - 1 && argc > 2 && info[2] == 3) {
      FILE* o = fopen(argv[2], "wb");
      fprintf(o, "P6\n%u %u\n255\n", info[0], info[1]);
      size_t nb;
      const void* p = jxlo_buffer(h, "rgb8", &nb);
      fwrite(p, 1, nb, o);
      fclose(o);
    }
    jxlo_free(h);
  }
  return 0;
}
#endif
