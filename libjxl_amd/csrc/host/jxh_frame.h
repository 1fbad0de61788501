// libjxl_amd host front-end: parses one VarDCT frame up to (not including) the AC group sections and produces a
// FramePlan — the flat, GPU-ready tables the HIP hot path consumes (csrc/hip/). Per-frame work only: headers, TOC,
// DC global, DC groups (Modular decode of DC + AC metadata), adaptive DC smoothing, EPF sigma, AC global
// (dequant tables, coefficient orders, entropy-code tables).
// Follows the frame walk of reference lib/jxl/dec_frame.cc:135-434 and lib/jxl/dec_modular.cc:427-562; this is the
// "prerequisites computed before the hot path" list of SURVEY.md §1. The AC group sections themselves are only
// located here (offset/size); their bytes are decoded on the GPU.
#ifndef JXH_FRAME_H_
#define JXH_FRAME_H_

#include <functional>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "jxh_bits.h"
#include "jxh_entropy.h"
#include "jxh_headers.h"
#include "jxh_modular.h"
#include "jxh_patches.h"
#include "jxh_splines.h"
#include "jxh_vardct.h"

namespace jxh {

// One varblock, in the decode order of its group. Packed for the device.
struct VarBlock {
  uint16_t bx, by;        // absolute block coordinates of the top-left 8x8
  uint8_t strategy;
  uint8_t quant_dc_ctx;   // DC-derived block context bucket
  uint16_t qf;            // raw quant field value 1..256
  uint32_t coef_offset;   // offset of this block's coefficients inside its group's [c][65536] planes
};

struct PassTables {
  bool use_prefix = false;
  bool lz77 = false;
  int log_alpha = 8;
  size_t num_clusters = 0;
  size_t max_num_bits = 0;              // widest value the code can produce (decides int16 vs int32 coefficients)
  std::vector<uint8_t> ctx_map;         // num_histograms * NumACContexts + 16 slack
  std::vector<AliasEntry> alias;        // num_clusters << log_alpha
  std::vector<uint32_t> uint_cfg;       // per cluster: split_exp | msb << 8 | lsb << 16
  std::vector<uint16_t> orders;         // [13 buckets][3 channels] -> order_offset[]; natural or custom
  uint32_t order_offset[39];            // start of (bucket, channel) in `orders` (in entries)
  // prefix-coded streams: one flat lookup table per cluster (JxlHipPassDesc::prefix_table / prefix_offset)
  std::vector<uint32_t> prefix_table, prefix_offset;
  uint32_t lz_min_symbol = 0, lz_min_length = 0, lz_len_cfg = 0, lz_dist_ctx = 0;
};

struct FramePlan {
  ImageHeader ih;
  FrameHeader fh;
  FrameDim dim;
  // quantiser / colour correlation scalars
  uint32_t global_scale = 1, quant_dc = 16;
  float inv_global_scale = 1.0f, x_dm = 1.0f, b_dm = 1.0f;
  float color_scale = 1.0f / 84, base_corr_x = 0.0f, base_corr_b = 1.0f;
  // block-resolution planes (xsize_blocks x ysize_blocks)
  // The two stencils over these planes, adaptive DC smoothing (compressed_dc.cc:130-198) and the EPF's 1 / sigma per
  // block (epf.cc:39-81), run on the device inside the upload (csrc/hip/jxl_hip_dc.h): the plan carries their inputs.
  // DequantDC (compressed_dc.cc:201-296) runs there too: the plan carries the coded integers.
  std::vector<int32_t> dc_q;        // 3 planes X, Y, B of coded DC integers (empty with kUseDcFrame)
  std::vector<uint8_t> dc_extra_precision;  // per DC group: the integers are in units of step / 2^this
  float dc_cfl_x = 0.0f, dc_cfl_b = 1.0f;   // chroma from luma at DC: base correlation + DC factor * colour scale
  bool dc_smoothing = false;        // the frame asks for the smoothing (no kSkipAdaptiveDCSmoothing)
  bool use_dc_frame = false;        // kUseDcFrame: `dc` stays empty, the DC image is `dc_source` (device planes of a DC frame)
  const float* dc_source = nullptr;  // [3][ysize_blocks][xsize_blocks] floats on the device (jxlamd_frame_set_dc_source)
  float dc_step[3] = {1, 1, 1};     // DC quantisation step per channel (the smoothing measures its gap in steps)
  std::vector<uint8_t> acs;         // (strategy << 1) | is_first
  std::vector<uint8_t> sharpness;   // EPF sharpness 0..7 per block
  std::vector<int8_t> ytox, ytob;   // per 64x64 tile
  // varblocks
  std::vector<VarBlock> blocks;             // all groups concatenated, decode order inside each group
  std::vector<uint32_t> group_block_begin;  // num_groups + 1
  uint32_t used_acs = 0;
  // block context map
  BlockCtxMap bctx;
  std::vector<uint8_t> block_ctx_lut;  // [c(3)][ord(13)][qf_idx][dc_ctx] -> block context (as BlockCtxMap::Context)
  // dequant tables for the 17 table kinds, concatenated; dequant_offset[kind] in floats (3 channels contiguous)
  std::vector<float> dequant;
  uint32_t dequant_offset[17];
  uint32_t dequant_size[17];  // per-channel entries
  size_t num_histograms = 1;
  std::vector<PassTables> passes;
  // AC sections: for pass p, group g: index p * num_groups + g
  float noise_lut[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // noise synthesis (frame flag kNoise): strength LUT
  bool has_noise = false;
  Patches patches;  // frame flag kPatches: dictionary (jxh_patches.h) and its device layout
  bool has_patches = false;
  std::vector<uint32_t> patch_records, patch_row_start, patch_row_list;
  const float* patch_src[4] = {nullptr, nullptr, nullptr, nullptr};  // device planes of the reference slots (set_patch_sources)
  const float* patch_src_alpha[4] = {nullptr, nullptr, nullptr, nullptr};  // ... and their alpha planes (set_patch_alpha_sources)
  uint32_t patch_src_w[4] = {0, 0, 0, 0}, patch_src_h[4] = {0, 0, 0, 0};
  Splines splines;  // frame flag kSplines: dictionary + draw cache (jxh_splines.h)
  bool has_splines = false;
  std::vector<uint64_t> section_offset;  // byte offset inside the codestream buffer handed to ParseFrame
  std::vector<uint32_t> section_size;
  // single-section frames: AC data starts mid-byte inside the one section
  uint32_t first_section_bit_offset = 0;
  size_t frame_end = 0;  // byte offset just after the frame
  // A frame parsed from a prefix of its bytes (ParseFrame with allow_partial; dec_frame.cc:735-795 Flush): per group, 1 when
  // one of its AC sections is not complete in the buffer. Such groups are rendered from their DC alone. Empty = whole frame.
  std::vector<uint8_t> group_absent;
  std::vector<uint8_t> passes_there;  // (partial frames) per group: how many of its passes, from the first, are there
  std::vector<uint64_t> section_end;  // (partial frames) [pass * num_groups + group]: byte of the buffer where the section ends
  size_t frame_index = 0, nonvisible_index = 0;  // shown frames before this one, invisible ones since (dec_frame.cc:160-168)
  // Extra channels (alpha, ...): Modular-coded beside the VarDCT colour (dec_frame.cc:511-542, dec_modular.cc:209-425).
  // `extra` is the frame's global Modular image; channels no larger than a group are complete after the DC global
  // section, the others continue in every AC group section BEHIND the coefficient stream, whose end only the entropy
  // decoder knows: FrameParser::FinishExtraChannels() takes the section end positions the GPU stage reports.
  MImage extra;
  bool extra_pending = false;   // group-coded channel data still to be decoded (FinishExtraChannels)
  std::vector<uint64_t> toc_section_offset;  // all TOC sections (for FinishExtraChannels): byte offset in the buffer
  std::vector<uint32_t> toc_section_size;
  MGlobal mglobal;  // the frame's global MA tree + code (Modular streams of the AC groups use it)
};

// Optional parallel-for hook (the JxlParallelRunner of the public API is adapted onto this).
typedef std::function<void(size_t /*count*/, const std::function<void(size_t)>&)> ParallelFor;
static inline void SerialFor(size_t n, const std::function<void(size_t)>& f) {
  for (size_t i = 0; i < n; i++) f(i);
}

class FrameParser {
 public:
  FrameParser(const uint8_t* data, size_t size) : data_(data), size_(size) {}

  // Parses signature + image header; returns the byte position of the first frame.
  size_t ParseImageHeader(ImageHeader* ih) {
    const uint8_t* d = data_;
    size_t n = size_;
    codestream_base_ = 0;
    static const uint8_t kContainer[12] = {0, 0, 0, 0xC, 'J', 'X', 'L', ' ', 0xD, 0xA, 0x87, 0xA};
    if (n >= 12 && !memcmp(d, kContainer, 12)) {
      size_t pos = 12;
      bool found = false;
      while (pos + 8 <= n) {
        uint64_t bsize = (uint64_t(d[pos]) << 24) | (d[pos + 1] << 16) | (d[pos + 2] << 8) | d[pos + 3];
        size_t hdr = 8;
        if (bsize == 1) {
          JXH_CHECK(pos + 16 <= n, "truncated box");
          bsize = 0;
          for (int i = 0; i < 8; i++) bsize = (bsize << 8) | d[pos + 8 + i];
          hdr = 16;
        }
        if (bsize == 0) bsize = n - pos;
        JXH_CHECK(bsize >= hdr && bsize <= n - pos, "bad box size");  // (no addition: a 64-bit size cannot wrap)
        if (!memcmp(d + pos + 4, "jxlc", 4)) {
          codestream_base_ = pos + hdr;
          cs_size_ = bsize - hdr;
          found = true;
          break;
        }
        JXH_CHECK(memcmp(d + pos + 4, "jxlp", 4) != 0, "unsupported: jxlp boxes");
        pos += bsize;
      }
      JXH_CHECK(found, "no codestream box");
    } else {
      cs_size_ = n;
    }
    const uint8_t* cs = data_ + codestream_base_;
    JXH_CHECK(cs_size_ >= 2 && cs[0] == 0xFF && cs[1] == 0x0A, "not a JPEG XL codestream");
    BitReader br(cs, cs_size_);
    br.Skip(16);
    try {
      ReadImageHeader(br, ih);
    } catch (const Error&) {
      // fields read beyond the end of the data are zeros: a check that trips over them is "not enough input", not a bad
      // stream (the reference's wrapper test feeds the first 0..5 bytes: DecoderTest.java:91-100)
      if (br.Overread()) throw Error("truncated image header");
      throw;
    }
    JXH_CHECK(!br.Overread(), "truncated image header");
    return codestream_base_ + br.BitPos() / 8;
  }

  // Parses the frame that starts at byte `pos` of the buffer. Throws jxh::Error for unsupported streams.
  // frame_index / nonvisible_index: the shown frames before this one and the invisible ones since (dec_frame.cc:160-168;
  // they seed the noise). The plan is of the frame at its own size: where it sits on the canvas and how it blends with
  // a reference slot (fh.x0 / y0, fh.blend) is the caller's business (the decoder API composes on a JxlHipCanvas).
  void ParseFrame(size_t pos, const ImageHeader& ih, FramePlan* plan, const ParallelFor& pfor = SerialFor, size_t frame_index = 0,
                  size_t nonvisible_index = 0, bool allow_partial = false) {
    FramePlan& P = *plan;
    P.ih = ih;
    P.frame_index = frame_index;
    P.nonvisible_index = nonvisible_index;
    BitReader br(data_ + pos, codestream_base_ + cs_size_ - pos);
    ReadFrameHeader(br, ih, &P.fh);
    const FrameHeader& fh = P.fh;
    // (frame types: 1 = kDCFrame, decoded like any frame and kept before the colour transform as the DC image of a later
    // frame; 2 = kReferenceOnly: kept for patches; 3 = kSkipProgressive)
    JXH_CHECK(!fh.modular, "unsupported: Modular frames on the GPU path");
    // (a VarDCT frame of an image that is not xyb_encoded: ColorTransform kNone or kYCbCr: the same decode with the
    // quant-matrix scales at 2 and another colour stage; a chroma-subsampled YCbCr frame: see FrameHeader::hshift)
    // dec_frame.cc:206-212
    JXH_CHECK(fh.Is444() || (fh.flags & FrameHeader::kSkipDcSmoothing), "chroma subsampling is not allowed when adaptive DC smoothing is enabled");
    JXH_CHECK(fh.Is444() || fh.frame_type != 1, "unsupported: chroma-subsampled DC frames");
    JXH_CHECK(!fh.custom_size || fh.upsampling == 1, "unsupported: cropped upsampled frames");
    // (extra channels carry an upsampling factor of their own, at least the frame's: frame_header.cc:272-283; the channel is
    // coded at ceil(image / factor) and upsampled by the stage of that factor: dec_cache.cc:172-190, 203-212)
    for (size_t e = 0; e < ih.extra.size(); e++)
      JXH_CHECK(fh.ec_upsampling.empty() || fh.ec_upsampling[e] >= fh.upsampling, "EC upsampling < color upsampling, which is invalid");
    // kUseDcFrame (frame_header.h:348; passes_state.cc:62-77, dec_frame.cc:319-326,347-356): the DC image is the output
    // of the DC frame of level fh.dc_level + 1 decoded earlier; the DC groups then carry no DC stream, the DC-derived block
    // context is 0 everywhere and nothing is smoothed. The planes live on the device (FramePlan::dc_source).
    P.use_dc_frame = (fh.flags & FrameHeader::kUseDcFrame) != 0;
    JXH_CHECK(!P.use_dc_frame || fh.dc_level < 4, "invalid DC level for kUseDcFrame");
    JXH_CHECK(fh.frame_type != 1 || ih.extra.empty(), "unsupported: DC frames of images with extra channels");
    // (patches and splines of an upsampled frame are drawn at the FRAME's resolution, before the upsampling: dec_cache.cc:193-212;
    // its noise is added behind the upsampling, at the image's resolution)
    P.dim = MakeFrameDim(fh);
    const FrameDim& d = P.dim;
    const size_t np = fh.num_passes;
    const size_t entries = (d.num_groups == 1 && np == 1) ? 1 : 2 + d.num_dc_groups + d.num_groups * np;
    Toc toc;
    ReadToc(br, entries, &toc);
    JXH_CHECK(!br.Overread(), "truncated frame header");
    const size_t base = pos + br.BitPos() / 8;
    const size_t have = codestream_base_ + cs_size_;  // bytes of the buffer that belong to the codestream
    bool partial = false;
    if (base + toc.total > have) {
      // A prefix of the frame: usable once the DC image and the AC global section are whole (FrameDecoder::HasDecodedDC,
      // decode.cc:2464-2468). Frames in one section, and frames whose AC sections also carry extra-channel data, wait.
      JXH_CHECK(allow_partial && entries > 1 && ih.extra.empty(), "truncated frame");
      for (size_t i = 0; i < 2 + d.num_dc_groups; i++) JXH_CHECK(base + toc.offset[i] + toc.size[i] <= have, "truncated frame");
      partial = true;
    }
    P.frame_end = base + toc.total;
    const size_t xb = d.xsize_blocks, yb = d.ysize_blocks;
    P.dc_q.assign(3 * xb * yb, 0);
    P.dc_extra_precision.assign(d.num_dc_groups, 0);
    P.acs.assign(xb * yb, 0xFF);
    P.ytox.assign(DivCeil(xb, 8) * DivCeil(yb, 8), 0);
    P.ytob.assign(P.ytox.size(), 0);
    quant_.assign(xb * yb, 0);
    P.sharpness.assign(xb * yb, 0);
    quant_dc_ctx_.assign(xb * yb, 0);

    P.section_offset.assign(d.num_groups * np, 0);
    P.section_size.assign(d.num_groups * np, 0);
    if (entries == 1) {
      BitReader r(data_ + base + toc.offset[0], toc.size[0]);
      DcGlobal(r, &P);
      DcGroup(r, &P, 0);
      FinalizeDc(&P);
      AcGlobal(r, &P);
      JXH_CHECK(!r.Overread(), "section over-read");
      // AC data continues in the same section at an arbitrary bit position
      P.section_offset[0] = base + toc.offset[0] + r.BitPos() / 8;
      P.first_section_bit_offset = uint32_t(r.BitPos() & 7);
      P.section_size[0] = uint32_t(toc.size[0] - r.BitPos() / 8);
    } else {
      const bool timing = getenv("JXLAMD_TIMING") != nullptr;  // debugging aid: stage times of the host front-end
      auto now = [] { return std::chrono::steady_clock::now(); };
      auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
      };
      const auto t0 = now();
      {
        BitReader r(data_ + base + toc.offset[0], toc.size[0]);
        DcGlobal(r, &P);
        JXH_CHECK(!r.Overread(), "DC global over-read");
      }
      const auto t1 = now();
      std::string err;
      std::mutex err_mu;  // DC groups run on the caller's threads: the first failure is kept
      pfor(d.num_dc_groups, [&](size_t g) {
        try {
          BitReader r(data_ + base + toc.offset[1 + g], toc.size[1 + g]);
          DcGroup(r, &P, g);
          JXH_CHECK(!r.Overread(), "DC group over-read");
        } catch (const std::exception& e) {
          std::lock_guard<std::mutex> lk(err_mu);
          if (err.empty()) err = e.what();
        }
      });
      JXH_CHECK(err.empty(), err);
      const auto t2 = now();
      FinalizeDc(&P);
      const auto t3 = now();
      {
        size_t i = 1 + d.num_dc_groups;
        BitReader r(data_ + base + toc.offset[i], toc.size[i]);
        AcGlobal(r, &P);
        JXH_CHECK(!r.Overread(), "AC global over-read");
      }
      if (timing)
        fprintf(stderr, "[host parse] DC global %.2f ms, DC groups %.2f ms, DC smoothing + sigma %.2f ms, AC global %.2f ms\n", ms(t0, t1),
                ms(t1, t2), ms(t2, t3), ms(t3, now()));
      for (size_t p = 0; p < np; p++)
        for (size_t g = 0; g < d.num_groups; g++) {
          size_t i = 2 + d.num_dc_groups + p * d.num_groups + g;
          P.section_offset[p * d.num_groups + g] = base + toc.offset[i];
          P.section_size[p * d.num_groups + g] = toc.size[i];
        }
      if (partial) {
        // A group is drawn from the passes that have arrived, in order (dec_frame.cc:620-680 decodes a group's passes as
        // their sections come, decoded_passes_per_ac_group_; Flush draws with what that left): passes_there[g] = its
        // leading passes whose sections are whole; the later ones get size 0; none at all = absent (the DC image alone).
        P.group_absent.assign(d.num_groups, 0);
        P.passes_there.assign(d.num_groups, uint8_t(np));
        P.section_end.assign(np * d.num_groups, 0);
        for (size_t i = 0; i < np * d.num_groups; i++) P.section_end[i] = base + toc.offset[2 + d.num_dc_groups + i] + toc.size[2 + d.num_dc_groups + i];
        for (size_t g = 0; g < d.num_groups; g++) {
          size_t k = 0;
          for (; k < np; k++) {
            const size_t i = 2 + d.num_dc_groups + k * d.num_groups + g;
            if (base + toc.offset[i] + toc.size[i] > have) break;
          }
          P.passes_there[g] = uint8_t(k);
          P.group_absent[g] = k == 0;
          for (size_t p = k; p < np; p++) {
            P.section_offset[p * d.num_groups + g] = base;  // (never read; kept inside the buffer)
            P.section_size[p * d.num_groups + g] = 0;
          }
        }
      }
    }
    BuildBlockLists(&P);
    if (!P.extra.ch.empty()) {
      P.extra_pending = ExtraHasGroupData(P);
      if (!P.extra_pending) UndoExtraTransforms(&P);
    }
  }

 private:
  void DcGlobal(BitReader& br, FramePlan* P) {
    if (P->fh.flags & FrameHeader::kPatches) {  // dec_frame.cc:271-285
      DecodePatches(br, P->dim.xsize_padded, P->dim.ysize_padded, P->ih.extra.size(), &P->patches);
      BuildPatchRows(P->patches, P->dim.ysize, &P->patch_records, &P->patch_row_start, &P->patch_row_list);
      P->has_patches = true;
    }
    if (P->fh.flags & FrameHeader::kSplines) {  // dec_frame.cc:289-293
      DecodeSplines(br, P->dim.xsize * P->dim.ysize, &P->splines);
      P->has_splines = true;
    }
    if (P->fh.flags & FrameHeader::kNoise) {  // dec_frame.cc:294-296, dec_noise.cc:154-164: eight 10-bit LUT points
      for (float& v : P->noise_lut) v = float(br.Read(10)) / 1024.0f;
      // (the reference skips the stage when every point is below 1e-3: noise.h:35-40)
      for (float v : P->noise_lut) P->has_noise = P->has_noise || std::fabs(v) > 1e-3f;
    }
    if (!br.ReadBool()) {
      for (int c = 0; c < 3; c++) {
        dq_.dc_quant[c] = ReadF16(br) * (1.0f / 128.0f);
        JXH_CHECK(dq_.dc_quant[c] >= 1e-8f, "invalid DC quant");
      }
    }
    P->global_scale = ReadU32(br, BitsOffset(11, 1), BitsOffset(11, 2049), BitsOffset(12, 4097), BitsOffset(16, 8193));
    P->quant_dc = ReadU32(br, Val(16), BitsOffset(5, 1), BitsOffset(8, 1), BitsOffset(16, 1));
    ReadBlockCtxMap(br, &P->bctx);
    uint32_t color_factor = 84;
    if (!br.ReadBool()) {
      color_factor = ReadU32(br, Val(84), Val(256), BitsOffset(8, 2), BitsOffset(16, 258));
      P->base_corr_x = ReadF16(br);
      P->base_corr_b = ReadF16(br);
      JXH_CHECK(std::fabs(P->base_corr_x) <= 4.0f && std::fabs(P->base_corr_b) <= 4.0f, "CfL base out of range");
      ytox_dc_ = int32_t(br.Read(8)) - 128;
      ytob_dc_ = int32_t(br.Read(8)) - 128;
    }
    P->color_scale = 1.0f / float(color_factor);
    // dec_frame.cc:303-308: the draw cache uses the base colour correlation just read
    if (P->has_splines) InitSplineDrawCache(&P->splines, P->dim.xsize, P->dim.ysize, P->base_corr_x, P->base_corr_b);
    P->inv_global_scale = 65536.0f / float(P->global_scale);
    P->x_dm = std::pow(1.25f, 2.0f - float(P->fh.x_qm_scale));
    P->b_dm = std::pow(1.25f, 2.0f - float(P->fh.b_qm_scale));
    MGlobal& mglobal_ = P->mglobal;
    if (br.ReadBool()) {
      const size_t nb = std::max<size_t>(P->ih.extra.size(), 1);
      size_t limit = std::min<size_t>(size_t(1) << 22, 1024 + P->dim.xsize * P->dim.ysize * nb / 16);
      DecodeTree(br, &mglobal_.tree, limit);
      DecodeHistograms(br, (mglobal_.tree.size() + 1) / 2, &mglobal_.code);
      mglobal_.have = true;
    }
    // Global Modular image = the extra channels (a VarDCT frame codes its colour elsewhere). Stream 0 holds the channels
    // no larger than a group; the transforms stay pending until every channel is complete (dec_modular.cc:209-318).
    P->extra.ch.clear();
    P->extra.bitdepth = int(P->ih.bits);
    for (size_t e = 0; e < P->ih.extra.size(); e++) {  // dec_modular.cc:262-271
      const uint32_t ups = P->fh.ec_upsampling.empty() ? 1 : P->fh.ec_upsampling[e];
      const int shift = CeilLog2(ups) - CeilLog2(P->fh.upsampling);
      P->extra.ch.emplace_back(DivCeil(size_t(P->fh.xsize), size_t(ups)), DivCeil(size_t(P->fh.ysize), size_t(ups)), shift, shift);
    }
    ModularDecode(br, &P->extra, 0, &mglobal_, P->dim.group_dim, /*undo_transforms=*/false);
  }

  // Part of the global Modular image covered by a rectangle, for the channels whose shift lies in [min_shift, max_shift]
  // (dec_modular.cc:320-425): DC groups carry the channels squeezed 8x or more, AC groups the rest.
  static void DecodeExtraRect(BitReader& br, FramePlan* P, size_t x0, size_t y0, size_t xs, size_t ys, int min_shift, int max_shift,
                              int stream_id) {
    MImage& full = P->extra;
    size_t c = full.nb_meta;
    const size_t gdim = P->dim.group_dim;
    while (c < full.ch.size() && full.ch[c].w <= gdim && full.ch[c].h <= gdim) c++;  // those came with stream 0
    MImage part;
    part.bitdepth = full.bitdepth;
    struct Place {
      size_t c, x, y;
    };
    std::vector<Place> places;
    for (; c < full.ch.size(); c++) {
      const MChannel& fc = full.ch[c];
      const int shift = std::min(fc.hshift, fc.vshift);
      if (shift < min_shift || shift > max_shift) continue;
      const size_t rx = x0 >> fc.hshift, ry = y0 >> fc.vshift;
      if (rx >= fc.w || ry >= fc.h) continue;
      const size_t rw = std::min(xs >> fc.hshift, fc.w - rx), rh = std::min(ys >> fc.vshift, fc.h - ry);
      if (!rw || !rh) continue;
      part.ch.emplace_back(rw, rh, fc.hshift, fc.vshift);
      places.push_back({c, rx, ry});
    }
    if (part.ch.empty()) return;
    ModularDecode(br, &part, stream_id, &P->mglobal);
    for (size_t i = 0; i < places.size(); i++) {
      MChannel& fc = full.ch[places[i].c];
      for (size_t y = 0; y < part.ch[i].h; y++)
        memcpy(fc.Row(places[i].y + y) + places[i].x, part.ch[i].Row(y), part.ch[i].w * sizeof(int32_t));
    }
  }
  static bool ExtraHasGroupData(const FramePlan& P) {
    for (size_t c = P.extra.nb_meta; c < P.extra.ch.size(); c++)
      if (P.extra.ch[c].w > P.dim.group_dim || P.extra.ch[c].h > P.dim.group_dim) return true;
    return false;
  }
  static void UndoExtraTransforms(FramePlan* P) {
    for (size_t i = P->extra.transforms.size(); i-- > 0;) InverseTransform(&P->extra, P->extra.transforms[i]);
    P->extra.transforms.clear();
    JXH_CHECK(P->extra.ch.size() == P->ih.extra.size(), "extra channels: channel count after the transforms");
  }

 public:
  // Completes the extra channels of a frame whose AC group sections carry Modular data behind the coefficients.
  // sec_end_bit[pass * num_groups + group] = bit position (from the start of the section) where that section's
  // coefficient stream ended, as reported by the entropy stage (jxlhip_get_section_end_bits).
  // The groups are independent Modular streams that fill disjoint rectangles of the channels: they run on the caller's
  // threads like the DC groups (an alpha plane of a 4K frame is 8.3 M samples: ~50 ms on one thread).
  static void FinishExtraChannels(const uint8_t* data, FramePlan* P, const uint32_t* sec_end_bit, const ParallelFor& pfor = SerialFor) {
    if (!P->extra_pending) return;
    const FrameDim& d = P->dim;
    const size_t np = P->fh.num_passes;
    std::string err;
    std::mutex err_mu;
    pfor(d.num_groups, [&](size_t g) {
      try {
        const size_t i = (np - 1) * d.num_groups + g;  // (streams without progressive-downsampling info: last pass only)
        BitReader r(data + P->section_offset[i], P->section_size[i]);
        r.Skip(sec_end_bit[i]);
        const size_t gx = g % d.xsize_groups, gy = g / d.xsize_groups;
        DecodeExtraRect(r, P, gx * d.group_dim, gy * d.group_dim, d.group_dim, d.group_dim, 0, 2,
                        int(1 + 3 * d.num_dc_groups + 17 + d.num_groups * (np - 1) + g));
        JXH_CHECK(!r.Overread(), "AC group: Modular data over-read");
      } catch (const std::exception& e) {
        std::lock_guard<std::mutex> lk(err_mu);
        if (err.empty()) err = e.what();
      }
    });
    JXH_CHECK(err.empty(), err);
    UndoExtraTransforms(P);
    P->extra_pending = false;
  }

 private:

  void DcGroup(BitReader& br, FramePlan* P, size_t g) {
    const FrameDim& d = P->dim;
    const size_t gx = g % d.xsize_dc_groups, gy = g / d.xsize_dc_groups;
    const size_t bx0 = gx * d.group_dim, by0 = gy * d.group_dim;
    const size_t bw = std::min(d.group_dim, d.xsize_blocks - bx0), bh = std::min(d.group_dim, d.ysize_blocks - by0);
    const size_t ndc = d.num_dc_groups, xb = d.xsize_blocks;
    if (!P->use_dc_frame) {
      P->dc_extra_precision[g] = uint8_t(br.Read(2));
      MImage img;
      // (dec_modular.cc:443-452: stream channel 0 is Y, 1 is X / Cb, 2 is B / Cr; a subsampled channel is smaller)
      const FrameHeader& fh = P->fh;
      static const int kStreamChan[3] = {1, 0, 2};
      for (int i = 0; i < 3; i++) img.ch.emplace_back(bw >> fh.hshift[kStreamChan[i]], bh >> fh.vshift[kStreamChan[i]]);
      ModularDecode(br, &img, int(1 + g), &P->mglobal);
      // (the channels are coded in Y, X, B order; dequantisation and chroma from luma happen on the device)
      const size_t plane = xb * d.ysize_blocks;
      const BlockCtxMap& bc = P->bctx;
      if (!fh.Is444()) {
        // compressed_dc.cc:232-290: every channel on its own grid, kept in the top-left part of its plane; a block's bucket
        // from the samples that cover it
        for (int c = 0; c < 3; c++) {
          const MChannel& ch = img.ch[c < 2 ? c ^ 1 : c];
          const size_t sx0 = bx0 >> fh.hshift[c], sy0 = by0 >> fh.vshift[c];
          for (size_t y = 0; y < ch.h; y++) memcpy(&P->dc_q[plane * c + (sy0 + y) * xb + sx0], ch.Row(y), ch.w * sizeof(int32_t));
        }
        for (size_t y = 0; bc.num_dc_ctxs > 1 && y < bh; y++) {
          const int32_t *qx = img.ch[1].Row(y >> fh.vshift[0]), *qy = img.ch[0].Row(y >> fh.vshift[1]), *qb = img.ch[2].Row(y >> fh.vshift[2]);
          for (size_t x = 0; x < bw; x++) {
            int kx = 0, ky = 0, kb = 0;
            for (int t : bc.dc_thresholds[0]) kx += qx[x >> fh.hshift[0]] > t;
            for (int t : bc.dc_thresholds[1]) ky += qy[x >> fh.hshift[1]] > t;
            for (int t : bc.dc_thresholds[2]) kb += qb[x >> fh.hshift[2]] > t;
            int b = kx;
            b = b * int(bc.dc_thresholds[2].size() + 1) + kb;
            b = b * int(bc.dc_thresholds[1].size() + 1) + ky;
            quant_dc_ctx_[(by0 + y) * xb + bx0 + x] = uint8_t(b);
          }
        }
      } else
      for (size_t y = 0; y < bh; y++) {
        const int32_t *qx = img.ch[1].Row(y), *qy = img.ch[0].Row(y), *qb = img.ch[2].Row(y);
        const size_t row = (by0 + y) * xb + bx0;
        memcpy(&P->dc_q[row], qx, bw * sizeof(int32_t));
        memcpy(&P->dc_q[plane + row], qy, bw * sizeof(int32_t));
        memcpy(&P->dc_q[2 * plane + row], qb, bw * sizeof(int32_t));
        if (bc.num_dc_ctxs <= 1) continue;  // (one DC context: the buckets stay 0)
        for (size_t x = 0; x < bw; x++) {
          size_t idx = row + x;
          uint8_t bucket = 0;
          {
            int kx = 0, ky = 0, kb = 0;
            for (int t : bc.dc_thresholds[0]) kx += qx[x] > t;
            for (int t : bc.dc_thresholds[1]) ky += qy[x] > t;
            for (int t : bc.dc_thresholds[2]) kb += qb[x] > t;
            int b = kx;
            b = b * int(bc.dc_thresholds[2].size() + 1) + kb;
            b = b * int(bc.dc_thresholds[1].size() + 1) + ky;
            bucket = uint8_t(b);
          }
          quant_dc_ctx_[idx] = bucket;
        }
      }
    }
    if (!P->extra.ch.empty()) {  // Modular DC group: channels of the global image squeezed 8x or more (stream 1 + ndc + g)
      std::lock_guard<std::mutex> lk(extra_mu_);  // (DC groups run on several threads; the channel list is shared)
      DecodeExtraRect(br, P, bx0 * 8, by0 * 8, d.dc_group_dim, d.dc_group_dim, 3, 1000, int(1 + ndc + g));
    }
    {
      size_t count = size_t(br.Read(CeilLog2(bw * bh))) + 1;
      MImage img;
      size_t cw = (bw + 7) >> 3, ch = (bh + 7) >> 3;
      img.ch.emplace_back(cw, ch, 3, 3);
      img.ch.emplace_back(cw, ch, 3, 3);
      img.ch.emplace_back(count, 2, 0, 0);
      img.ch.emplace_back(bw, bh, 0, 0);
      ModularDecode(br, &img, int(1 + 2 * ndc + g), &P->mglobal);
      const size_t tiles_x = DivCeil(xb, 8);
      for (size_t y = 0; y < ch; y++)
        for (size_t x = 0; x < cw; x++) {
          size_t idx = (by0 / 8 + y) * tiles_x + bx0 / 8 + x;
          P->ytox[idx] = int8_t(std::max(-128, std::min(127, img.ch[0].Row(y)[x])));
          P->ytob[idx] = int8_t(std::max(-128, std::min(127, img.ch[1].Row(y)[x])));
        }
      size_t num = 0;
      const int32_t *r1 = img.ch[2].Row(0), *r2 = img.ch[2].Row(1);
      uint32_t used = 0;
      for (size_t iy = 0; iy < bh; iy++) {
        size_t y = by0 + iy;
        for (size_t ix = 0; ix < bw; ix++) {
          size_t x = bx0 + ix;
          int sh = img.ch[3].Row(iy)[ix];
          JXH_CHECK(sh >= 0 && sh < 8, "corrupted sharpness field");
          P->sharpness[y * xb + x] = uint8_t(sh);
          if (P->acs[y * xb + x] != 0xFF) continue;
          JXH_CHECK(num < count, "AC metadata: too few strategies");
          int raw = r1[num];
          JXH_CHECK(raw >= 0 && raw < 27, "invalid AC strategy");
          used |= 1u << raw;
          size_t cx = kCoveredX[raw], cy = kCoveredY[raw];
          size_t nx = (x / 32 + 1) * 32, ny = (y / 32 + 1) * 32;
          JXH_CHECK(cx * cy == 1 || P->fh.Is444(), "AC strategy not compatible with chroma subsampling");  // dec_modular.cc:534-538
          JXH_CHECK(x + cx <= nx && x + cx <= std::min(d.xsize_blocks, bx0 + bw), "AC strategy x overflow");
          JXH_CHECK(y + cy <= ny && y + cy <= std::min(d.ysize_blocks, by0 + bh), "AC strategy y overflow");
          for (size_t jy = 0; jy < cy; jy++)
            for (size_t jx = 0; jx < cx; jx++) {
              uint8_t& a = P->acs[(y + jy) * xb + x + jx];
              JXH_CHECK(a == 0xFF, "AC strategy overlap");
              a = uint8_t((raw << 1) | ((jx | jy) == 0 ? 1 : 0));
            }
          quant_[y * xb + x] = uint16_t(1 + std::max(0, std::min(255, r2[num])));
          num++;
        }
      }
      used_mutex_or(P, used);
    }
  }

  void used_mutex_or(FramePlan* P, uint32_t used) {
    // DC groups may run on several threads; a plain OR of a 32-bit mask is safe with the atomic builtin.
    __atomic_fetch_or(&P->used_acs, used, __ATOMIC_RELAXED);
  }

  // What the device needs to finish the DC path (smoothing and 1 / sigma are kernels: see FramePlan::dc).
  void FinalizeDc(FramePlan* P) {
    const float inv_quant_dc = P->inv_global_scale / float(P->quant_dc);
    for (int c = 0; c < 3; c++) P->dc_step[c] = inv_quant_dc * dq_.dc_quant[c];
    P->dc_cfl_x = P->base_corr_x + ytox_dc_ * P->color_scale;
    P->dc_cfl_b = P->base_corr_b + ytob_dc_ * P->color_scale;
    if (P->use_dc_frame) P->dc_q.clear();
    P->dc_smoothing = !(P->fh.flags & FrameHeader::kSkipDcSmoothing) && !P->use_dc_frame;
  }

  void AcGlobal(BitReader& br, FramePlan* P) {
    if (!br.ReadBool()) {
      // RAW tables (what JPEG recompression writes) are small Modular images behind the frame's global tree
      const RawTableReader raw = [P](BitReader& r, size_t w, size_t h, int kind, std::vector<int32_t>* out) {
        MImage img;
        for (int c = 0; c < 3; c++) img.ch.emplace_back(w, h);
        ModularDecode(r, &img, int(1 + 3 * P->dim.num_dc_groups + size_t(kind)), &P->mglobal);
        out->resize(3 * w * h);
        for (int c = 0; c < 3; c++)
          for (size_t y = 0; y < h; y++) memcpy(out->data() + (size_t(c) * h + y) * w, img.ch[c].Row(y), w * sizeof(int32_t));
      };
      for (int k = 0; k < 17; k++) {
        ReadQuantEncoding(br, k, &dq_.enc[k], &raw);
        dq_.table[k].clear();
      }
    }
    // dequant tables for the kinds in use
    P->dequant.clear();
    for (int k = 0; k < 17; k++) {
      bool used = false;
      for (int s = 0; s < 27; s++)
        if ((P->used_acs & (1u << s)) && kStrategyQuantTable[s] == k) used = true;
      P->dequant_offset[k] = uint32_t(P->dequant.size());
      P->dequant_size[k] = 0;
      if (!used) continue;
      dq_.Compute(k);
      P->dequant_size[k] = uint32_t(dq_.table[k].size() / 3);
      P->dequant.insert(P->dequant.end(), dq_.table[k].begin(), dq_.table[k].end());
    }
    const FrameDim& d = P->dim;
    P->num_histograms = 1 + size_t(br.Read(CeilLog2(d.num_groups)));
    P->passes.resize(P->fh.num_passes);
    for (uint32_t p = 0; p < P->fh.num_passes; p++) {
      PassTables& T = P->passes[p];
      uint32_t used_orders = ReadU32(br, Val(0x5F), Val(0x13), Val(0), Bits(13));
      std::vector<uint32_t> orders;
      DecodeCoeffOrders(br, used_orders, P->used_acs, &orders);
      // compact the used buckets into u16
      T.orders.clear();
      for (int ord = 0; ord < 13; ord++) {
        int s = OrderBucketStrategy(ord);
        size_t size = size_t(kCoveredX[s]) * kCoveredY[s] * 64;
        bool used = false;
        for (int t = 0; t < 27; t++)
          if ((P->used_acs & (1u << t)) && kStrategyOrder[t] == ord) used = true;
        for (int c = 0; c < 3; c++) {
          T.order_offset[3 * ord + c] = uint32_t(T.orders.size());
          if (!used) continue;
          const uint32_t* src = &orders[CoeffOrderOffset(ord, c)];
          for (size_t k = 0; k < size; k++) T.orders.push_back(uint16_t(src[k]));
        }
      }
      EntropyCode code;
      size_t nctx = P->num_histograms * P->bctx.NumACContexts();
      DecodeHistograms(br, nctx, &code);
      T.use_prefix = code.use_prefix;
      T.lz77 = code.lz77;
      T.log_alpha = code.log_alpha;
      T.num_clusters = code.num_clusters;
      T.max_num_bits = code.max_num_bits;
      T.ctx_map = code.ctx_map;  // (an LZ77 stream's extra distance context sits at the end: kept in lz_dist_ctx)
      T.ctx_map.resize(nctx);
      T.ctx_map.resize(nctx + 16, 0);
      T.alias = code.alias;
      T.uint_cfg.resize(code.num_clusters);
      for (size_t k = 0; k < code.num_clusters; k++)
        T.uint_cfg[k] = code.cfg[k].split_exp | (code.cfg[k].msb << 8) | (code.cfg[k].lsb << 16);
      if (T.use_prefix) {
        T.alias.clear();
        T.prefix_offset.resize(code.num_clusters);
        for (size_t k = 0; k < code.num_clusters; k++) {
          const PrefixCode& pc = code.prefix[k];
          T.prefix_offset[k] = AppendPrefixTables(pc, &T.prefix_table);
        }
      }
      if (T.lz77) {
        T.lz_min_symbol = code.lz_min_symbol;
        T.lz_min_length = code.lz_min_length;
        T.lz_len_cfg = code.lz_len_cfg.split_exp | (code.lz_len_cfg.msb << 8) | (code.lz_len_cfg.lsb << 16);
        T.lz_dist_ctx = code.lz_dist_ctx;
      }
    }
    // block-context LUT: [c][ord][qf_idx][dc_ctx]
    const BlockCtxMap& bc = P->bctx;
    size_t nq = bc.qf_thresholds.size() + 1;
    P->block_ctx_lut.assign(3 * 13 * nq * bc.num_dc_ctxs, 0);
    for (size_t c = 0; c < 3; c++)
      for (size_t ord = 0; ord < 13; ord++)
        for (size_t q = 0; q < nq; q++)
          for (size_t dcx = 0; dcx < bc.num_dc_ctxs; dcx++) {
            size_t idx = c < 2 ? c ^ 1 : 2;
            idx = idx * 13 + ord;
            idx = idx * nq + q;
            idx = idx * bc.num_dc_ctxs + dcx;
            P->block_ctx_lut[((c * 13 + ord) * nq + q) * bc.num_dc_ctxs + dcx] = bc.ctx_map[idx];
          }
  }

  void BuildBlockLists(FramePlan* P) {
    const FrameDim& d = P->dim;
    const size_t xb = d.xsize_blocks;
    P->blocks.clear();
    P->group_block_begin.assign(d.num_groups + 1, 0);
    for (size_t g = 0; g < d.num_groups; g++) {
      P->group_block_begin[g] = uint32_t(P->blocks.size());
      const size_t bx0 = (g % d.xsize_groups) * 32, by0 = (g / d.xsize_groups) * 32;
      const size_t bw = std::min<size_t>(32, d.xsize_blocks - bx0), bh = std::min<size_t>(32, d.ysize_blocks - by0);
      uint32_t offset = 0;
      for (size_t by = 0; by < bh; by++)
        for (size_t bx = 0; bx < bw; bx++) {
          uint8_t a = P->acs[(by0 + by) * xb + bx0 + bx];
          JXH_CHECK(a != 0xFF, "AC strategy map has holes");
          if (!(a & 1)) continue;
          VarBlock v;
          v.bx = uint16_t(bx0 + bx);
          v.by = uint16_t(by0 + by);
          v.strategy = uint8_t(a >> 1);
          v.quant_dc_ctx = quant_dc_ctx_[(by0 + by) * xb + bx0 + bx];
          v.qf = quant_[(by0 + by) * xb + bx0 + bx];
          v.coef_offset = offset;
          offset += uint32_t(64) << kLog2Covered[v.strategy];
          P->blocks.push_back(v);
        }
    }
    P->group_block_begin[d.num_groups] = uint32_t(P->blocks.size());
  }

  const uint8_t* data_;
  size_t size_;
  size_t codestream_base_ = 0, cs_size_ = 0;
  DequantTables dq_;
  std::mutex extra_mu_;
  int32_t ytox_dc_ = 0, ytob_dc_ = 0;
  std::vector<uint16_t> quant_;
  std::vector<uint8_t> quant_dc_ctx_;
};

}  // namespace jxh
#endif  // JXH_FRAME_H_
