// libjxl_amd host front-end (product code; runs on the CPU ahead of the GPU hot path).
// CustomTransformData, FrameHeader (+Passes, BlendingInfo, LoopFilter), TOC.
// Follows: reference lib/jxl/headers.cc:129-203, lib/jxl/image_metadata.cc:26-65,78-230,283-440,
// lib/jxl/color_encoding_internal.cc:94-215, lib/jxl/frame_header.cc:30-439, lib/jxl/frame_header.h:35-50,
// lib/jxl/loop_filter.cc:20-100, lib/jxl/toc.cc:29-115, lib/jxl/toc.h:31-41,
// lib/jxl/frame_dimensions.h:34-59.
// Unsupported features (ICC, preview, animation timing beyond parsing, non-sRGB colour) raise jxh::Error.
#ifndef JXH_HEADERS_H_
#define JXH_HEADERS_H_

#include <cmath>
#include <string>
#include <vector>

#include "jxh_bits.h"
#include "jxh_entropy.h"
#include "jxh_icc.h"

namespace jxh {

struct ExtraChannel {
  uint32_t type = 0;  // 0 = alpha
  uint32_t bits = 8, exp_bits = 0;
  bool floating = false;
  uint32_t dim_shift = 0;
  bool alpha_associated = false;
};

struct ImageHeader {
  uint32_t xsize = 0, ysize = 0;
  uint32_t orientation = 1;
  uint32_t bits = 8, exp_bits = 0;
  bool floating = false;
  bool modular_16bit = true;
  std::vector<ExtraChannel> extra;
  bool xyb_encoded = true;
  bool gray = false;
  // ColorEncoding as coded (enum values of color_encoding.h); custom xy in millionths, gamma in 1e-7
  uint32_t white_point = 1, primaries = 1, transfer_function = 13, rendering_intent = 1, gamma = 0;
  bool have_gamma = false;
  int32_t white_xy[2] = {0, 0}, primaries_xy[6] = {0, 0, 0, 0, 0, 0};
  bool linear_tf = false;  // output transfer function: false = sRGB, true = linear
  bool want_icc = false;   // the original colours are described by an embedded ICC profile ...
  std::vector<uint8_t> icc;  // ... decoded here (jxh_icc.h)
  bool have_preview = false;
  uint32_t preview_xsize = 0, preview_ysize = 0;
  bool have_animation = false, have_timecodes = false;
  uint32_t anim_tps_num = 0, anim_tps_den = 0, anim_loops = 0;  // AnimationHeader (image_metadata.cc:235-250)
  float intensity_target = 255.0f;
  // CustomTransformData / OpsinInverseMatrix
  float inv_opsin[9] = {11.031566901960783f,  -9.866943921568629f, -0.16462299647058826f,
                        -3.254147380392157f,  4.418770392156863f,  -0.16462299647058826f,
                        -3.6588512862745097f, 2.7129230470588235f, 1.9459282392156863f};
  float opsin_bias[3] = {-0.0037930732552754493f, -0.0037930732552754493f, -0.0037930732552754493f};
  float quant_bias[4] = {1.0f - 0.05465007330715401f, 1.0f - 0.07005449891748593f, 1.0f - 0.049935103337343655f,
                         0.145f};
  bool custom_upsampling = false;
  // coded upsampling weights (image_metadata.cc:87-214: upper triangle of the symmetric 5N/2 x 5N/2 matrix); empty = default
  std::vector<float> ups_weights2, ups_weights4, ups_weights8;
};

static inline void ReadBitDepth(BitReader& br, uint32_t* bits, uint32_t* exp_bits, bool* floating) {
  *floating = br.ReadBool();
  if (!*floating) {
    *bits = ReadU32(br, Val(8), Val(10), Val(12), BitsOffset(6, 1));
    *exp_bits = 0;
    JXH_CHECK(*bits <= 31, "invalid bits_per_sample");
  } else {
    *bits = ReadU32(br, Val(32), Val(16), Val(24), BitsOffset(6, 1));
    *exp_bits = uint32_t(br.Read(4)) + 1;
    JXH_CHECK(*exp_bits >= 2 && *exp_bits <= 8, "invalid exponent bits");
    int mant = int(*bits) - int(*exp_bits) - 1;
    JXH_CHECK(mant >= 2 && mant <= 23, "invalid float bits_per_sample");
  }
}

static inline std::string ReadName(BitReader& br) {
  uint32_t len = ReadU32(br, Val(0), Bits(4), BitsOffset(5, 16), BitsOffset(10, 48));
  std::string s(len, ' ');
  for (uint32_t i = 0; i < len; i++) s[i] = char(br.Read(8));
  return s;
}

static inline uint32_t ReadSizeDim(BitReader& br) {
  return ReadU32(br, BitsOffset(9, 1), BitsOffset(13, 1), BitsOffset(18, 1), BitsOffset(30, 1));
}
static inline uint32_t AspectRatioX(uint32_t ysize, uint32_t ratio) {
  static const uint32_t num[8] = {0, 1, 12, 4, 3, 16, 5, 2}, den[8] = {1, 1, 10, 3, 2, 9, 4, 1};
  return uint32_t(uint64_t(ysize) * num[ratio] / den[ratio]);
}

static inline void ReadColorEncoding(BitReader& br, ImageHeader* h) {
  if (br.ReadBool()) return;  // all_default: sRGB
  h->want_icc = br.ReadBool();
  uint32_t cs = ReadEnum(br);  // 0 RGB, 1 Gray, 2 XYB, 3 Unknown
  JXH_CHECK(cs == 0 || cs == 1, "unsupported: colour space");
  h->gray = cs == 1;
  // an embedded ICC profile describes the ORIGINAL colours (it follows the headers: ReadImageHeader); the enum fields are
  // absent then (color_encoding_internal.cc:151-158). XYB images are still decoded to sRGB, the others keep their samples.
  if (h->want_icc) return;
  // color_encoding_internal.cc:144-200: white point, primaries (custom ones as signed millionths), transfer function or
  // gamma, rendering intent. Kept as coded: they describe samples that pass through unchanged when the image is not XYB.
  auto custom_xy = [&](int32_t* xy) {
    for (int i = 0; i < 2; i++) {
      const uint32_t u = ReadU32(br, Bits(19), BitsOffset(19, 524288), BitsOffset(20, 1048576), BitsOffset(21, 2097152));
      xy[i] = (u & 1) ? -int32_t((u + 1) >> 1) : int32_t(u >> 1);
    }
  };
  h->white_point = ReadEnum(br);
  JXH_CHECK(h->white_point == 1 || h->white_point == 2 || h->white_point == 10 || h->white_point == 11, "invalid white point");
  if (h->white_point == 2) custom_xy(h->white_xy);
  if (cs == 0) {
    h->primaries = ReadEnum(br);
    JXH_CHECK(h->primaries == 1 || h->primaries == 2 || h->primaries == 9 || h->primaries == 11, "invalid primaries");
    if (h->primaries == 2)
      for (int c = 0; c < 3; c++) custom_xy(h->primaries_xy + 2 * c);
  }
  h->have_gamma = br.ReadBool();
  if (h->have_gamma) {
    h->gamma = uint32_t(br.Read(24));  // in units of 1e-7 (kGammaMul)
    JXH_CHECK(h->gamma <= 10000000 && uint64_t(h->gamma) * 8192 >= 10000000, "invalid gamma");
  } else {
    h->transfer_function = ReadEnum(br);
    JXH_CHECK(h->transfer_function == 1 || h->transfer_function == 8 || h->transfer_function == 13 || h->transfer_function == 16 ||
            h->transfer_function == 17 || h->transfer_function == 18, "unsupported: transfer function");
  }
  h->rendering_intent = ReadEnum(br);
  JXH_CHECK(h->rendering_intent <= 3, "invalid rendering intent");
  h->linear_tf = !h->have_gamma && h->transfer_function == 8;
  // an XYB image is rendered by the colour stage here, which knows sRGB primaries with the sRGB curve or none
  // (dec_xyb.cc:181-250 does the other enum spaces; no CMS here): anything else is refused rather than mis-rendered
  if (h->xyb_encoded)
    JXH_CHECK(h->white_point == 1 && (cs == 1 || h->primaries == 1) && !h->have_gamma && (h->transfer_function == 13 || h->transfer_function == 8),
            "unsupported: XYB image in a colour space other than (linear) sRGB");
}

static inline void ReadImageHeader(BitReader& br, ImageHeader* h) {
  // SizeHeader
  bool small = br.ReadBool();
  if (small) h->ysize = (uint32_t(br.Read(5)) + 1) * 8;
  else h->ysize = ReadSizeDim(br);
  uint32_t ratio = uint32_t(br.Read(3));
  if (ratio == 0) {
    if (small) h->xsize = (uint32_t(br.Read(5)) + 1) * 8;
    else h->xsize = ReadSizeDim(br);
  } else {
    h->xsize = AspectRatioX(h->ysize, ratio);
  }
  // ImageMetadata
  bool all_default = br.ReadBool();
  if (!all_default) {
    bool extra_fields = br.ReadBool();
    bool have_preview = false;
    if (extra_fields) {
      h->orientation = uint32_t(br.Read(3)) + 1;
      if (br.ReadBool()) {  // intrinsic size (a SizeHeader)
        bool s = br.ReadBool();
        uint32_t ys = s ? (uint32_t(br.Read(5)) + 1) * 8 : ReadSizeDim(br);
        uint32_t r = uint32_t(br.Read(3));
        if (r == 0) {
          if (s) br.Read(5);
          else ReadSizeDim(br);
        }
        (void)ys;
      }
      have_preview = br.ReadBool();
      if (have_preview) {  // headers.cc:155-183 PreviewHeader; the preview is the codestream's first frame, of this size
        const bool div8 = br.ReadBool();
        auto dim = [&]() {
          return div8 ? ReadU32(br, Val(16), Val(32), BitsOffset(5, 1), BitsOffset(9, 33)) * 8
                      : ReadU32(br, BitsOffset(6, 1), BitsOffset(8, 65), BitsOffset(10, 321), BitsOffset(12, 1345));
        };
        h->preview_ysize = dim();
        const uint32_t ratio = uint32_t(br.Read(3));
        h->preview_xsize = ratio == 0 ? dim() : AspectRatioX(h->preview_ysize, ratio);
        h->have_preview = true;
      }
      h->have_animation = br.ReadBool();
      if (h->have_animation) {
        h->anim_tps_num = ReadU32(br, Val(100), Val(1000), BitsOffset(10, 1), BitsOffset(30, 1));
        h->anim_tps_den = ReadU32(br, Val(1), Val(1001), BitsOffset(8, 1), BitsOffset(10, 1));
        h->anim_loops = ReadU32(br, Val(0), Bits(3), Bits(16), Bits(32));
        h->have_timecodes = br.ReadBool();
      }
    }
    ReadBitDepth(br, &h->bits, &h->exp_bits, &h->floating);
    h->modular_16bit = br.ReadBool();
    uint32_t num_extra = ReadU32(br, Val(0), Val(1), BitsOffset(4, 2), BitsOffset(12, 1));
    h->extra.resize(num_extra);
    for (auto& e : h->extra) {
      if (br.ReadBool()) continue;  // all_default: 8-bit alpha
      e.type = ReadEnum(br);
      ReadBitDepth(br, &e.bits, &e.exp_bits, &e.floating);
      e.dim_shift = ReadU32(br, Val(0), Val(3), Val(4), BitsOffset(3, 1));
      JXH_CHECK((1u << e.dim_shift) <= 8, "dim_shift too large");
      ReadName(br);
      if (e.type == 0) e.alpha_associated = br.ReadBool();
      if (e.type == 2) for (int i = 0; i < 4; i++) ReadF16(br);  // spot colour
      if (e.type == 5) ReadU32(br, Val(1), Bits(2), BitsOffset(4, 3), BitsOffset(8, 19));  // CFA
    }
    h->xyb_encoded = br.ReadBool();
    ReadColorEncoding(br, h);
    if (extra_fields) {  // ToneMapping
      if (!br.ReadBool()) {
        h->intensity_target = ReadF16(br);
        JXH_CHECK(h->intensity_target > 0, "invalid intensity target");
        ReadF16(br);   // min_nits
        br.ReadBool();  // relative_to_max_display
        ReadF16(br);   // linear_below
      }
    }
    SkipExtensions(br);
  }
  // CustomTransformData
  if (!br.ReadBool()) {
    if (h->xyb_encoded) {
      if (!br.ReadBool()) {  // OpsinInverseMatrix not default
        for (int i = 0; i < 9; i++) h->inv_opsin[i] = ReadF16(br);
        for (int i = 0; i < 3; i++) h->opsin_bias[i] = ReadF16(br);
        for (int i = 0; i < 4; i++) h->quant_bias[i] = ReadF16(br);
      }
    }
    uint32_t mask = uint32_t(br.Read(3));
    if (mask & 1) for (int i = 0; i < 15; i++) h->ups_weights2.push_back(ReadF16(br));
    if (mask & 2) for (int i = 0; i < 55; i++) h->ups_weights4.push_back(ReadF16(br));
    if (mask & 4) for (int i = 0; i < 210; i++) h->ups_weights8.push_back(ReadF16(br));
    h->custom_upsampling = mask != 0;
  }
  if (h->want_icc) ReadIcc(br, &h->icc);  // decode.cc: after the transform data, before the byte boundary
  br.ToByteBoundary();
}

struct LoopFilter {
  bool gab = true;
  float gab_w[3][2] = {{1.1f * 0.104699568f, 1.1f * 0.055680538f},
                       {1.1f * 0.104699568f, 1.1f * 0.055680538f},
                       {1.1f * 0.104699568f, 1.1f * 0.055680538f}};
  uint32_t epf_iters = 2;
  float epf_sharp_lut[8] = {0, 1.f / 7, 2.f / 7, 3.f / 7, 4.f / 7, 5.f / 7, 6.f / 7, 1};
  float epf_channel_scale[3] = {40.0f, 5.0f, 3.5f};
  float epf_quant_mul = 0.46f, epf_pass0_sigma_scale = 0.9f, epf_pass2_sigma_scale = 6.5f,
        epf_border_sad_mul = 2.0f / 3.0f, epf_sigma_for_modular = 1.0f;
};

// frame_header.cc:72-104 BlendingInfo: how a frame's channels combine with a reference slot (blending.cc, alpha.cc)
struct BlendInfo {
  uint32_t mode = 0;  // 0 replace, 1 add, 2 blend (alpha over), 3 alpha-weighted add, 4 multiply
  uint32_t alpha_channel = 0, source = 0;
  bool clamp = false;
};

struct FrameHeader {
  uint32_t frame_type = 0;  // 0 regular, 1 DC frame, 2 reference only, 3 skip progressive
  uint32_t dc_level = 0;    // 1..4 for a DC frame (it becomes the DC image of level dc_level - 1), else 0
  bool modular = false;
  uint64_t flags = 0;
  bool ycbcr = false;
  // YCbCrChromaSubsampling (frame_header.h:81-166, frame_header.cc:30-31): channel_mode of Cb, Y, Cr = samples per MCU side
  // {1x1, 2x2, 2x1, 1x2}; a channel's shift = the largest log2 factor of the three minus its own. Channel c then has
  // (xsize_blocks >> hshift[c]) x (ysize_blocks >> vshift[c]) blocks.
  uint32_t cs_mode[3] = {0, 0, 0};
  uint32_t hshift[3] = {0, 0, 0}, vshift[3] = {0, 0, 0}, max_hshift = 0, max_vshift = 0;
  bool Is444() const { return !max_hshift && !max_vshift; }
  void SetChannelModes(uint32_t m0, uint32_t m1, uint32_t m2) {
    static const uint32_t kH[4] = {0, 1, 1, 0}, kV[4] = {0, 1, 0, 1};
    cs_mode[0] = m0; cs_mode[1] = m1; cs_mode[2] = m2;
    max_hshift = max_vshift = 0;
    for (uint32_t m : cs_mode) {
      max_hshift = kH[m] > max_hshift ? kH[m] : max_hshift;
      max_vshift = kV[m] > max_vshift ? kV[m] : max_vshift;
    }
    for (int c = 0; c < 3; c++) {
      hshift[c] = max_hshift - kH[cs_mode[c]];
      vshift[c] = max_vshift - kV[cs_mode[c]];
    }
  }
  uint32_t upsampling = 1;
  std::vector<uint32_t> ec_upsampling;
  uint32_t group_size_shift = 1;
  uint32_t x_qm_scale = 3, b_qm_scale = 2;
  uint32_t num_passes = 1;
  uint32_t num_downsample = 0, downsample[4] = {0, 0, 0, 0}, last_pass[4] = {0, 0, 0, 0};  // frame_header.h:299-309 (Passes)
  uint32_t pass_shift[11] = {0};
  bool custom_size = false;
  int32_t x0 = 0, y0 = 0;
  uint32_t xsize = 0, ysize = 0;  // frame dimensions (filled from image if not custom)
  bool is_last = true;
  uint32_t save_as_reference = 0;
  uint32_t duration = 0, timecode = 0;  // AnimationFrame (frame_header.cc:130-150), in ticks of the image's AnimationHeader
  uint32_t blend_mode = 0;              // BlendMode of the colour channels: 0 = replace
  std::string name;                     // frame_header.cc:431 (UTF-8, up to 1071 bytes)
  BlendInfo blend;                      // ... in full, and of every extra channel
  std::vector<BlendInfo> ec_blend;
  bool save_before_color_transform = false;
  LoopFilter lf;
  static const uint64_t kNoise = 1, kPatches = 2, kSplines = 16, kUseDcFrame = 32, kSkipDcSmoothing = 128;
};

static inline void ReadLoopFilter(BitReader& br, bool modular, LoopFilter* lf) {
  if (br.ReadBool()) return;
  lf->gab = br.ReadBool();
  if (lf->gab) {
    if (br.ReadBool()) {
      for (int c = 0; c < 3; c++) {
        lf->gab_w[c][0] = ReadF16(br);
        lf->gab_w[c][1] = ReadF16(br);
        JXH_CHECK(std::fabs(1.0f + (lf->gab_w[c][0] + lf->gab_w[c][1]) * 4) >= 1e-8, "degenerate gaborish");
      }
    }
  }
  lf->epf_iters = uint32_t(br.Read(2));
  if (lf->epf_iters > 0) {
    if (!modular) {
      if (br.ReadBool()) for (int i = 0; i < 8; i++) lf->epf_sharp_lut[i] = ReadF16(br);
    }
    if (br.ReadBool()) {
      for (int i = 0; i < 3; i++) lf->epf_channel_scale[i] = ReadF16(br);
      ReadF16(br);  // pass1 zeroflush (parsed, unused)
      ReadF16(br);  // pass2 zeroflush
    }
    if (br.ReadBool()) {
      if (!modular) lf->epf_quant_mul = ReadF16(br);
      lf->epf_pass0_sigma_scale = ReadF16(br);
      lf->epf_pass2_sigma_scale = ReadF16(br);
      lf->epf_border_sad_mul = ReadF16(br);
    }
    if (modular) {
      lf->epf_sigma_for_modular = ReadF16(br);
      JXH_CHECK(lf->epf_sigma_for_modular >= 1e-8, "EPF sigma for modular too small");
    }
  }
  SkipExtensions(br);
}

static inline void ReadBlendingInfo(BitReader& br, size_t num_extra, bool partial, BlendInfo* b) {
  b->mode = ReadU32(br, Val(0), Val(1), Val(2), BitsOffset(2, 3));
  JXH_CHECK(b->mode <= 4, "invalid blend mode");
  bool alpha_modes = num_extra > 0 && (b->mode == 2 || b->mode == 3);
  if (alpha_modes) {
    b->alpha_channel = ReadU32(br, Val(0), Val(1), Val(2), BitsOffset(3, 3));
    JXH_CHECK(b->alpha_channel < num_extra, "blending refers to a missing alpha channel");
  }
  if (alpha_modes || b->mode == 4) b->clamp = br.ReadBool();
  if (b->mode != 0 || partial) b->source = ReadU32(br, Val(0), Val(1), Val(2), Val(3));
}

static inline void ReadFrameHeader(BitReader& br, const ImageHeader& ih, FrameHeader* f) {
  f->xsize = ih.xsize;
  f->ysize = ih.ysize;
  if (br.ReadBool()) {
    if (!ih.xyb_encoded) f->x_qm_scale = f->b_qm_scale = 2;
    return;
  }
  f->frame_type = uint32_t(br.Read(2));
  f->modular = br.ReadBool();
  f->flags = ReadU64(br);
  if (!ih.xyb_encoded) f->ycbcr = br.ReadBool();
  bool use_dc_frame = (f->flags & FrameHeader::kUseDcFrame) != 0;
  if (f->ycbcr && !use_dc_frame) {
    // YCbCrChromaSubsampling: 3 x 2 bits
    uint32_t m0 = uint32_t(br.Read(2)), m1 = uint32_t(br.Read(2)), m2 = uint32_t(br.Read(2));
    f->SetChannelModes(m0, m1, m2);
  }
  if (!use_dc_frame) {
    f->upsampling = ReadU32(br, Val(1), Val(2), Val(4), Val(8));
    f->ec_upsampling.assign(ih.extra.size(), 1);
    for (size_t i = 0; i < ih.extra.size(); i++) {
      uint32_t u = ReadU32(br, Val(1), Val(2), Val(4), Val(8));
      f->ec_upsampling[i] = u << ih.extra[i].dim_shift;
    }
  }
  if (f->modular) f->group_size_shift = uint32_t(br.Read(2));
  if (!f->modular && ih.xyb_encoded) {
    f->x_qm_scale = uint32_t(br.Read(3));
    f->b_qm_scale = uint32_t(br.Read(3));
  } else {
    f->x_qm_scale = f->b_qm_scale = 2;
  }
  if (f->frame_type != 2) {
    f->num_passes = ReadU32(br, Val(1), Val(2), Val(3), BitsOffset(3, 4));
    if (f->num_passes != 1) {
      uint32_t num_ds = ReadU32(br, Val(0), Val(1), Val(2), BitsOffset(1, 3));
      JXH_CHECK(num_ds <= 4 && num_ds <= f->num_passes, "invalid num_downsample");
      for (uint32_t i = 0; i + 1 < f->num_passes; i++) f->pass_shift[i] = uint32_t(br.Read(2));
      f->pass_shift[f->num_passes - 1] = 0;
      f->num_downsample = num_ds;
      for (uint32_t i = 0; i < num_ds; i++) f->downsample[i] = ReadU32(br, Val(1), Val(2), Val(4), Val(8));
      for (uint32_t i = 0; i < num_ds; i++) f->last_pass[i] = ReadU32(br, Val(0), Val(1), Val(2), Bits(3));
    }
  }
  if (f->frame_type == 1) {  // frame_header.cc:310-318, frame_header.h:470-476: a DC frame has the image's size / 8^level
    f->dc_level = ReadU32(br, Val(1), Val(2), Val(3), Val(4));
    f->xsize = DivCeil(f->xsize, size_t(1) << (3 * f->dc_level));
    f->ysize = DivCeil(f->ysize, size_t(1) << (3 * f->dc_level));
  }
  bool partial = false;
  if (f->frame_type != 1) {
    f->custom_size = br.ReadBool();
    if (f->custom_size) {
      if (f->frame_type == 0 || f->frame_type == 3) {
        uint32_t ux = ReadU32(br, Bits(8), BitsOffset(11, 256), BitsOffset(14, 2304), BitsOffset(30, 18688));
        uint32_t uy = ReadU32(br, Bits(8), BitsOffset(11, 256), BitsOffset(14, 2304), BitsOffset(30, 18688));
        f->x0 = UnpackSigned(ux);
        f->y0 = UnpackSigned(uy);
      }
      f->xsize = ReadU32(br, Bits(8), BitsOffset(11, 256), BitsOffset(14, 2304), BitsOffset(30, 18688));
      f->ysize = ReadU32(br, Bits(8), BitsOffset(11, 256), BitsOffset(14, 2304), BitsOffset(30, 18688));
      JXH_CHECK(f->xsize && f->ysize, "zero-sized frame");
      partial = f->x0 > 0 || f->y0 > 0 || int64_t(f->xsize) + f->x0 < int64_t(ih.xsize) ||
                int64_t(f->ysize) + f->y0 < int64_t(ih.ysize);
    }
  }
  uint32_t blend_mode = 0;
  if (f->frame_type == 0 || f->frame_type == 3) {
    ReadBlendingInfo(br, ih.extra.size(), partial, &f->blend);
    blend_mode = f->blend.mode;
    f->ec_blend.resize(ih.extra.size());
    for (size_t i = 0; i < ih.extra.size(); i++) ReadBlendingInfo(br, ih.extra.size(), partial, &f->ec_blend[i]);
    if (ih.have_animation) {
      f->duration = ReadU32(br, Val(0), Val(1), Bits(8), Bits(32));
      if (ih.have_timecodes) f->timecode = uint32_t(br.Read(32));
    }
    f->blend_mode = blend_mode;
    f->is_last = br.ReadBool();
  } else {
    f->is_last = false;
  }
  if (f->frame_type != 1 && !f->is_last) f->save_as_reference = ReadU32(br, Val(0), Val(1), Val(2), Val(3));
  if (f->frame_type != 1) {
    // frame_header.h:373-379 CanBeReferenced(): !is_last && frame_type != DC && (duration == 0 || save_as_reference != 0)
    const bool can_ref = !f->is_last && (f->duration == 0 || f->save_as_reference != 0);
    if (can_ref && blend_mode == 0 && !partial && (f->frame_type == 0 || f->frame_type == 3)) {
      f->save_before_color_transform = br.ReadBool();
    } else if (f->frame_type == 2) {
      f->save_before_color_transform = br.ReadBool();
    }
  }
  f->name = ReadName(br);
  ReadLoopFilter(br, f->modular, &f->lf);
  SkipExtensions(br);
}

// Geometry derived from the frame header.
struct FrameDim {
  size_t xsize, ysize;            // frame size in pixels (before upsampling)
  size_t xsize_blocks, ysize_blocks;
  size_t xsize_padded, ysize_padded;
  size_t group_dim, dc_group_dim;
  size_t xsize_groups, ysize_groups, xsize_dc_groups, ysize_dc_groups;
  size_t num_groups, num_dc_groups;
};
static inline FrameDim MakeFrameDim(const FrameHeader& f) {
  FrameDim d;
  d.group_dim = size_t(128) << f.group_size_shift;
  d.dc_group_dim = d.group_dim * 8;
  d.xsize = DivCeil(f.xsize, f.upsampling);
  d.ysize = DivCeil(f.ysize, f.upsampling);
  // (frame_dimensions.h:43-44: whole MCUs of a chroma-subsampled frame)
  d.xsize_blocks = DivCeil(d.xsize, size_t(8) << f.max_hshift) << f.max_hshift;
  d.ysize_blocks = DivCeil(d.ysize, size_t(8) << f.max_vshift) << f.max_vshift;
  d.xsize_padded = f.modular ? d.xsize : d.xsize_blocks * 8;
  d.ysize_padded = f.modular ? d.ysize : d.ysize_blocks * 8;
  d.xsize_groups = DivCeil(d.xsize, d.group_dim);
  d.ysize_groups = DivCeil(d.ysize, d.group_dim);
  d.xsize_dc_groups = DivCeil(d.xsize_blocks, d.group_dim);
  d.ysize_dc_groups = DivCeil(d.ysize_blocks, d.group_dim);
  d.num_groups = d.xsize_groups * d.ysize_groups;
  d.num_dc_groups = d.xsize_dc_groups * d.ysize_dc_groups;
  return d;
}

// Lehmer code -> permutation (lib/jxl/lehmer_code.h:60-100): perm[i] = code[i]-th not-yet-used value.
static inline void DecodeLehmer(const std::vector<uint32_t>& code, std::vector<uint32_t>* perm) {
  size_t n = code.size();
  std::vector<uint32_t> avail(n);
  for (size_t i = 0; i < n; i++) avail[i] = uint32_t(i);
  perm->resize(n);
  for (size_t i = 0; i < n; i++) {
    JXH_CHECK(code[i] < avail.size(), "invalid lehmer code");
    (*perm)[i] = avail[code[i]];
    avail.erase(avail.begin() + code[i]);
  }
}

static inline uint32_t PermutationContext(uint32_t v) {
  uint32_t tok = v == 0 ? 0 : uint32_t(FloorLog2(v)) + 1;  // HybridUintConfig(0,0,0) token
  return std::min(tok, 7u);
}

// Reads a permutation of `size` entries whose first `skip` entries are fixed (coeff_order.cc:37-62).
static inline void ReadPermutation(BitReader& br, SymbolReader& rd, size_t skip, size_t size,
                                   std::vector<uint32_t>* perm) {
  std::vector<uint32_t> lehmer(size, 0);
  uint32_t end = rd.Read(PermutationContext(uint32_t(size))) + uint32_t(skip);
  JXH_CHECK(end <= size, "invalid permutation size");
  uint32_t last = 0;
  for (size_t i = skip; i < end; i++) {
    lehmer[i] = rd.Read(PermutationContext(last));
    last = lehmer[i];
    JXH_CHECK(lehmer[i] < size - i, "invalid lehmer code");
  }
  (void)br;
  DecodeLehmer(lehmer, perm);
}

struct Toc {
  std::vector<uint64_t> offset;  // byte offset of each section from the end of the TOC, in logical order
  std::vector<uint32_t> size;
  uint64_t total = 0;
};

static inline void ReadToc(BitReader& br, size_t entries, Toc* toc) {
  JXH_CHECK(entries > 0 && entries <= 65536, "bad TOC entry count");
  std::vector<uint32_t> perm;
  if (br.ReadBool()) {
    EntropyCode code;
    DecodeHistograms(br, 8, &code);
    SymbolReader rd(&code, &br);
    ReadPermutation(br, rd, 0, entries, &perm);
    JXH_CHECK(rd.FinalStateOk(), "TOC permutation: bad ANS final state");
  }
  br.ToByteBoundary();
  std::vector<uint32_t> sizes(entries);
  for (auto& s : sizes)
    s = ReadU32(br, Bits(10), BitsOffset(14, 1024), BitsOffset(22, 17408), BitsOffset(30, 4211712));
  br.ToByteBoundary();
  std::vector<uint64_t> offs(entries);
  uint64_t o = 0;
  for (size_t i = 0; i < entries; i++) {
    offs[i] = o;
    o += sizes[i];
  }
  toc->total = o;
  if (perm.empty()) {
    toc->offset = offs;
    toc->size = sizes;
  } else {
    toc->offset.resize(entries);
    toc->size.resize(entries);
    for (size_t i = 0; i < entries; i++) {
      toc->offset[i] = offs[perm[i]];
      toc->size[i] = sizes[perm[i]];
    }
  }
}

}  // namespace jxh
#endif  // JXH_HEADERS_H_
