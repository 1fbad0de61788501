// libjxl_amd — host side of the drop-in boundary: the JxlDecoder* C API subset (include/jxl/decode.h), the
// JxlThreadParallelRunner / JxlResizableParallelRunner (include/jxl/thread_parallel_runner.h, resizable_parallel_runner.h) and the frame-level
// helpers of include/jxl_amd.h. Mirrors the behaviour of reference lib/jxl/decode.cc (event order, status values,
// sticky errors, caller-owned input/output) for whole-file input; pixels come from the HIP layer only — if no
// MI355X/HIP device is usable every decode fails with JXL_DEC_ERROR (there is deliberately no CPU fallback).
#include <jxl/decode.h>
#include <jxl/resizable_parallel_runner.h>
#include <jxl/thread_parallel_runner.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <algorithm>
#include <vector>

#include "../../../include/jxl_amd.h"
#include "../host/jxh_frame.h"
#include "../host/jxh_modframe.h"

namespace {
thread_local std::string g_last_error;

struct RunnerClosure {
  const std::function<void(size_t)>* fn;
};
int RunnerInit(void*, size_t) { return 0; }
void RunnerFunc(void* opaque, uint32_t value, size_t) { (*static_cast<RunnerClosure*>(opaque)->fn)(value); }

jxh::ParallelFor MakeParallelFor(JxlParallelRunner runner, void* opaque) {
  if (!runner) return jxh::SerialFor;
  return [runner, opaque](size_t n, const std::function<void(size_t)>& f) {
    RunnerClosure c{&f};
    if (n == 0) return;
    if (runner(opaque, &c, RunnerInit, RunnerFunc, 0, uint32_t(n)) != 0) throw jxh::Error("parallel runner failed");
  };
}
}  // namespace

struct JxlAmdFrame {
  jxh::FramePlan plan;
  const uint8_t* data = nullptr;
  size_t size = 0;
  uint32_t coef_bits = 16;
  std::vector<JxlHipPassDesc> pass_desc;
};

extern "C" {

const char* jxlamd_last_error(void) { return g_last_error.c_str(); }

// image_features.splines -> JxlHipSplines (the arrays stay owned by the plan)
static void FillSplines(bool has, const jxh::Splines& s, JxlHipSplines* out) {
  memset(out, 0, sizeof(*out));
  if (!has || s.segments.empty()) return;
  out->num_segments = uint32_t(s.segments.size() / 8);
  out->num_row_segments = uint32_t(s.row_segments.size());
  out->segments = s.segments.data();
  out->row_start = s.row_start.data();
  out->row_segments = s.row_segments.data();
}

int jxlamd_frame_parse(const uint8_t* data, size_t size, JxlParallelRunner runner, void* runner_opaque, JxlAmdFrame** out) {
  return jxlamd_frame_parse_at(data, size, 0, 0, runner, runner_opaque, out);
}

static int ParseFrameAt(const uint8_t* data, size_t size, size_t frame_pos, size_t frame_index, JxlParallelRunner runner,
                        void* runner_opaque, bool allow_partial, JxlAmdFrame** out);
int jxlamd_frame_parse_at(const uint8_t* data, size_t size, size_t frame_pos, size_t frame_index, JxlParallelRunner runner,
                          void* runner_opaque, JxlAmdFrame** out) {
  return ParseFrameAt(data, size, frame_pos, frame_index, runner, runner_opaque, false, out);
}
int jxlamd_frame_parse_partial_at(const uint8_t* data, size_t size, size_t frame_pos, size_t frame_index, JxlParallelRunner runner,
                                  void* runner_opaque, JxlAmdFrame** out, uint32_t* groups_present) {
  const int r = ParseFrameAt(data, size, frame_pos, frame_index, runner, runner_opaque, true, out);
  if (!r && groups_present) {
    const std::vector<uint8_t>& a = (*out)->plan.group_absent;
    uint32_t n = uint32_t((*out)->plan.dim.num_groups);
    for (uint8_t v : a) n -= v ? 1 : 0;
    *groups_present = n;
  }
  return r;
}
int jxlamd_frame_is_partial(const JxlAmdFrame* f) { return f && !f->plan.group_absent.empty() ? 1 : 0; }
uint32_t jxlamd_frame_complete_passes(const JxlAmdFrame* f, size_t have_bytes, uint32_t* info) {
  if (!f) return 0;
  const auto& P = f->plan;
  const uint32_t np = P.fh.num_passes;
  if (info) {
    info[0] = np;
    info[1] = P.fh.num_downsample;
    for (int i = 0; i < 4; i++) {
      info[2 + i] = P.fh.downsample[i];
      info[6 + i] = P.fh.last_pass[i];
    }
  }
  if (P.section_end.empty()) return np;  // a whole frame
  const size_t ng = P.dim.num_groups;
  uint32_t complete = np;
  for (size_t g = 0; g < ng && complete; g++) {
    uint32_t k = 0;
    while (k < complete && P.section_end[size_t(k) * ng + g] <= have_bytes) k++;
    complete = k;
  }
  return complete;
}
static int ParseFrameAt(const uint8_t* data, size_t size, size_t frame_pos, size_t frame_index, JxlParallelRunner runner,
                        void* runner_opaque, bool allow_partial, JxlAmdFrame** out) {
  g_last_error.clear();
  if (!data || !out) {
    g_last_error = "invalid argument";
    return 1;
  }
  *out = nullptr;
  std::unique_ptr<JxlAmdFrame> f(new JxlAmdFrame);
  try {
    jxh::FrameParser parser(data, size);
    jxh::ImageHeader ih;
    size_t pos = parser.ParseImageHeader(&ih);
    const bool first = !frame_pos || frame_pos == pos;
    if (frame_pos) {
      if (frame_pos < pos || frame_pos >= size) throw std::runtime_error("truncated frame");
      pos = frame_pos;
    }
    if (first && ih.have_preview) {  // decode.cc:1266-1268: the first frame is the preview, its default size the preview's
      ih.xsize = ih.preview_xsize;   // (frame_header.h:450-463)
      ih.ysize = ih.preview_ysize;
    }
    parser.ParseFrame(pos, ih, &f->plan, MakeParallelFor(runner, runner_opaque), frame_index, 0, allow_partial);
    if (first && ih.have_preview && (f->plan.fh.frame_type != 0 || f->plan.fh.custom_size || f->plan.fh.blend.mode != 0))
      throw std::runtime_error("invalid preview frame");  // frame_header.cc:224-227, 372-376
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return 2;
  }
  f->data = data;
  f->size = size;
  size_t max_bits = 0;
  for (const auto& p : f->plan.passes) max_bits = std::max(max_bits, p.max_num_bits);
  max_bits += jxh::CeilLog2(f->plan.fh.num_passes);
  f->coef_bits = max_bits < 16 ? 16 : 32;
  *out = f.release();
  return 0;
}

static void FillPlacement(const jxh::ImageHeader& ih, const jxh::FrameHeader& fh, size_t xs, size_t ys, JxlAmdFramePlacement* p) {
  memset(p, 0, sizeof(*p));
  p->x0 = fh.x0;
  p->y0 = fh.y0;
  p->xsize = uint32_t(xs);
  p->ysize = uint32_t(ys);
  p->custom_size = fh.custom_size;
  p->frame_type = fh.frame_type;
  p->mode = fh.blend.mode;
  p->source = fh.blend.source;
  p->clamp = fh.blend.clamp;
  if (!ih.extra.empty() && !fh.ec_blend.empty()) {
    p->alpha_mode = fh.ec_blend[0].mode;
    p->alpha_source = fh.ec_blend[0].source;
    p->alpha_clamp = fh.ec_blend[0].clamp;
  }
  p->duration = fh.duration;
  p->is_last = fh.is_last;
  p->save_as_reference = fh.save_as_reference;
  p->save_before_color_transform = fh.save_before_color_transform;
  p->dc_level = fh.dc_level;
  p->use_dc_frame = (fh.flags & jxh::FrameHeader::kUseDcFrame) ? 1 : 0;
}
void jxlamd_frame_placement(const JxlAmdFrame* f, JxlAmdFramePlacement* p) {
  FillPlacement(f->plan.ih, f->plan.fh, f->plan.dim.xsize * f->plan.fh.upsampling, f->plan.dim.ysize * f->plan.fh.upsampling, p);
  if (f->plan.fh.upsampling != 1) {
    p->xsize = uint32_t(f->plan.ih.xsize);
    p->ysize = uint32_t(f->plan.ih.ysize);
  }
}
int jxlamd_frame_set_dc_source(JxlAmdFrame* f, const float* planes, uint32_t xs, uint32_t ys) {
  g_last_error.clear();
  jxh::FramePlan& P = f->plan;
  if (!P.use_dc_frame || !planes || xs != P.dim.xsize_blocks || ys != P.dim.ysize_blocks) {
    // (passes_state.cc:70-75: "kUseDcFrame specified for dc_level %u, but no frame was decoded with level %u")
    g_last_error = !P.use_dc_frame ? "the frame does not use a DC frame" : "kUseDcFrame: no DC frame of that level and size was decoded";
    return 1;
  }
  P.dc_source = planes;
  return 0;
}

int jxlamd_frame_set_patch_sources(JxlAmdFrame* f, const float* const* planes, const uint32_t* xs, const uint32_t* ys) {
  g_last_error.clear();
  jxh::FramePlan& P = f->plan;
  for (int i = 0; i < 4; i++) {
    P.patch_src[i] = planes[i];
    P.patch_src_w[i] = planes[i] ? xs[i] : 0;
    P.patch_src_h[i] = planes[i] ? ys[i] : 0;
  }
  for (const jxh::PatchRef& r : P.patches.refs) {  // dec_patch_dictionary.cc:63-83
    if (!P.patch_src[r.slot]) {
      g_last_error = "patches: the reference frame is missing";
      return 2;
    }
    if (uint64_t(r.x0) + r.xsize > P.patch_src_w[r.slot] || uint64_t(r.y0) + r.ysize > P.patch_src_h[r.slot]) {
      g_last_error = "patches: rectangle outside the reference frame";
      return 2;
    }
  }
  return 0;
}
// Does the frame's patch stage read and write the alpha channel (blending.cc:47-56: an image with an alpha channel)?
static bool PatchesBlendAlpha(const jxh::FramePlan& P) {
  return P.has_patches && P.patches.uses_alpha && P.ih.extra.size() == 1 && P.ih.extra[0].type == 0;
}
int jxlamd_frame_set_patch_alpha_sources(JxlAmdFrame* f, const float* const* alpha) {
  g_last_error.clear();
  jxh::FramePlan& P = f->plan;
  for (int i = 0; i < 4; i++) P.patch_src_alpha[i] = alpha ? alpha[i] : nullptr;
  if (P.has_patches && P.patches.uses_alpha && !P.ih.extra.empty() && !PatchesBlendAlpha(P)) {
    g_last_error = "unsupported: patches that blend extra channels of an image whose extra channel is not alpha";
    return 2;
  }
  if (!PatchesBlendAlpha(P)) return 0;
  if (P.fh.upsampling != 1 || (!P.fh.ec_upsampling.empty() && P.fh.ec_upsampling[0] != 1)) {
    g_last_error = "unsupported: patches that blend through alpha on upsampled frames";
    return 2;
  }
  for (const jxh::PatchRef& r : P.patches.refs)
    if (!P.patch_src_alpha[r.slot]) {
      g_last_error = "patches: the reference frame kept no alpha channel";
      return 2;
    }
  return 0;
}
void jxlamd_frame_set_indices(JxlAmdFrame* f, uint32_t visible_index, uint32_t nonvisible_index) {
  f->plan.frame_index = visible_index;
  f->plan.nonvisible_index = nonvisible_index;
}
void jxlamd_frame_free(JxlAmdFrame* f) { delete f; }
size_t jxlamd_frame_end(const JxlAmdFrame* f, uint32_t* t) {
  if (t) {
    t[0] = f->plan.fh.duration;
    t[1] = f->plan.fh.is_last;
    t[2] = f->plan.fh.timecode;
  }
  return f->plan.frame_end;
}

void jxlamd_frame_info(const JxlAmdFrame* f, uint32_t* info) {
  const jxh::FramePlan& P = f->plan;
  info[0] = uint32_t(P.dim.xsize);
  info[1] = uint32_t(P.dim.ysize);
  info[2] = uint32_t(P.dim.xsize_blocks);
  info[3] = uint32_t(P.dim.ysize_blocks);
  info[4] = uint32_t(P.dim.num_groups);
  info[5] = uint32_t(P.dim.num_dc_groups);
  info[6] = P.fh.num_passes;
  info[7] = P.used_acs;
  info[8] = P.fh.lf.epf_iters;
  info[9] = P.fh.lf.gab;
  info[10] = f->coef_bits;
  uint64_t total = 0;
  for (uint32_t s : P.section_size) total += s;
  info[11] = uint32_t(total);
  info[12] = P.passes.empty() ? 0 : uint32_t(P.passes[0].log_alpha);
  info[13] = P.passes.empty() ? 0 : uint32_t(P.passes[0].num_clusters);
  info[14] = P.passes.empty() ? 0 : uint32_t(P.passes[0].ctx_map.size());
}

size_t jxlamd_frame_section_sizes(const JxlAmdFrame* f, uint32_t* sizes, size_t n) {
  const std::vector<uint32_t>& s = f->plan.section_size;
  for (size_t i = 0; i < s.size() && i < n; i++) sizes[i] = s[i];
  return s.size();
}

void jxlamd_frame_out_size(const JxlAmdFrame* f, uint32_t* wh) {
  const jxh::FramePlan& P = f->plan;
  wh[0] = uint32_t(P.fh.upsampling == 1 ? P.dim.xsize : P.ih.xsize);
  wh[1] = uint32_t(P.fh.upsampling == 1 ? P.dim.ysize : P.ih.ysize);
}

// The N*N 5x5 upsampling kernels from the upper triangle of the symmetric weight matrix (stage_upsampling.cc:59-84):
// the weights the image header codes (image_metadata.cc:87-214, CustomTransformData) or the default ones (:98-214).
#include "../host/upsampling_weights.inc"
static void UpsamplingKernels(uint32_t N, const jxh::ImageHeader& ih, std::vector<float>* kernel) {
  const std::vector<float>& coded = N == 2 ? ih.ups_weights2 : (N == 4 ? ih.ups_weights4 : ih.ups_weights8);
  const float* weights = !coded.empty() ? coded.data() : (N == 2 ? kUpsamplingWeights2 : (N == 4 ? kUpsamplingWeights4 : kUpsamplingWeights8));
  kernel->assign(size_t(N) * N * 25, 0.0f);
  const size_t H = N / 2;
  for (size_t ky = 0; ky < H; ++ky)
    for (size_t kx = 0; kx < H; ++kx) {
      const size_t o0 = (ky * N + kx) * 25, o1 = (ky * N + (N - 1 - kx)) * 25, o2 = ((N - 1 - ky) * N + kx) * 25,
                   o3 = ((N - 1 - ky) * N + (N - 1 - kx)) * 25;
      for (size_t py = 0; py < 5; ++py)
        for (size_t px = 0; px < 5; ++px) {
          const size_t j = 5 * ky + py, i = 5 * kx + px, my = std::min(i, j), mx = std::max(i, j);
          const float w = weights[5 * H * my - my * (my - 1) / 2 + mx - my];
          (*kernel)[o0 + py * 5 + px] = w;
          (*kernel)[o1 + py * 5 + (4 - px)] = w;
          (*kernel)[o2 + (4 - py) * 5 + px] = w;
          (*kernel)[o3 + (4 - py) * 5 + (4 - px)] = w;
        }
    }
}

int jxlamd_frame_upload(const JxlAmdFrame* f, JxlHipContext* ctx) { return jxlamd_frame_upload_band(f, ctx, 0, 0); }

int jxlamd_frame_upload_band(const JxlAmdFrame* f, JxlHipContext* ctx, uint32_t group_row_begin, uint32_t group_row_end) {
  g_last_error.clear();
  if (!f || !ctx) {
    g_last_error = "invalid argument";
    return 1;
  }
  const jxh::FramePlan& P = f->plan;
  JxlHipFrameDesc d;
  memset(&d, 0, sizeof(d));
  d.xsize = uint32_t(P.dim.xsize);
  d.ysize = uint32_t(P.dim.ysize);
  d.xsize_blocks = uint32_t(P.dim.xsize_blocks);
  d.ysize_blocks = uint32_t(P.dim.ysize_blocks);
  d.xsize_groups = uint32_t(P.dim.xsize_groups);
  d.num_groups = uint32_t(P.dim.num_groups);
  d.num_passes = P.fh.num_passes;
  d.coef_bits = f->coef_bits;
  d.codestream = f->data;
  d.section_offset = P.section_offset.data();
  d.section_size = P.section_size.data();
  d.group_absent = P.group_absent.empty() ? nullptr : P.group_absent.data();
  d.first_section_bit_offset = P.first_section_bit_offset;
  std::vector<JxlHipPassDesc> pd(P.passes.size());
  for (size_t p = 0; p < P.passes.size(); p++) {
    const jxh::PassTables& T = P.passes[p];
    pd[p].log_alpha = uint32_t(T.log_alpha);
    pd[p].num_clusters = uint32_t(T.num_clusters);
    pd[p].ctx_map = T.ctx_map.data();
    pd[p].ctx_map_size = uint32_t(T.ctx_map.size());
    static_assert(sizeof(jxh::AliasEntry) == 8, "alias entry must be 8 bytes");
    pd[p].alias = T.alias.data();
    pd[p].uint_cfg = T.uint_cfg.data();
    pd[p].orders = T.orders.data();
    pd[p].orders_size = uint32_t(T.orders.size());
    memcpy(pd[p].order_offset, T.order_offset, sizeof(T.order_offset));
    pd[p].shift = P.fh.pass_shift[p];
    pd[p].use_prefix = T.use_prefix ? 1 : 0;
    pd[p].prefix_table = T.prefix_table.data();
    pd[p].prefix_table_size = uint32_t(T.prefix_table.size());
    pd[p].prefix_offset = T.prefix_offset.data();
    pd[p].lz77 = T.lz77 ? 1 : 0;
    pd[p].lz_min_symbol = T.lz_min_symbol;
    pd[p].lz_min_length = T.lz_min_length;
    pd[p].lz_len_cfg = T.lz_len_cfg;
    pd[p].lz_dist_ctx = T.lz_dist_ctx;
  }
  d.passes = pd.data();
  static_assert(sizeof(jxh::VarBlock) == sizeof(JxlHipVarBlock), "varblock layout");
  d.blocks = reinterpret_cast<const JxlHipVarBlock*>(P.blocks.data());
  d.num_blocks = uint32_t(P.blocks.size());
  d.group_block_begin = P.group_block_begin.data();
  d.block_ctx_lut = P.block_ctx_lut.data();
  d.block_ctx_lut_size = uint32_t(P.block_ctx_lut.size());
  d.num_block_ctxs = uint32_t(P.bctx.num_ctxs);
  d.num_dc_ctxs = uint32_t(P.bctx.num_dc_ctxs);
  d.num_qf_thresholds = uint32_t(P.bctx.qf_thresholds.size());
  for (size_t i = 0; i < P.bctx.qf_thresholds.size() && i < 16; i++) d.qf_thresholds[i] = P.bctx.qf_thresholds[i];
  d.num_histograms = uint32_t(P.num_histograms);
  d.dequant = P.dequant.data();
  d.dequant_floats = uint32_t(P.dequant.size());
  memcpy(d.dequant_offset, P.dequant_offset, sizeof(d.dequant_offset));
  memcpy(d.dequant_size, P.dequant_size, sizeof(d.dequant_size));
  if (P.use_dc_frame && !P.dc_source) {
    g_last_error = "kUseDcFrame: the DC frame's planes were not set (jxlamd_frame_set_dc_source)";
    return 1;
  }
  d.dc = nullptr;  // DequantDC, smoothing and 1 / sigma run inside the upload (csrc/hip/jxl_hip_dc.h)
  d.dc_quantised = P.use_dc_frame ? nullptr : P.dc_q.data();
  d.dc_extra_precision = P.dc_extra_precision.data();
  d.dc_cfl_x = P.dc_cfl_x;
  d.dc_cfl_b = P.dc_cfl_b;
  d.dc_device = P.use_dc_frame ? P.dc_source : nullptr;
  d.dc_smoothing = P.dc_smoothing ? 1 : 0;  // smoothing and 1 / sigma are computed by the upload (csrc/hip/jxl_hip_dc.h)
  memcpy(d.dc_step, P.dc_step, sizeof(d.dc_step));
  d.inv_sigma = nullptr;
  d.sharpness = P.sharpness.data();
  d.quant_scale = float(P.global_scale) * (1.0f / 65536.0f);
  d.epf_quant_mul = P.fh.lf.epf_quant_mul;
  memcpy(d.epf_sharp_lut, P.fh.lf.epf_sharp_lut, sizeof(d.epf_sharp_lut));
  d.ytox = P.ytox.data();
  d.ytob = P.ytob.data();
  for (int c = 0; c < 3; c++) {
    d.chroma_hshift[c] = uint8_t(P.fh.hshift[c]);
    d.chroma_vshift[c] = uint8_t(P.fh.vshift[c]);
  }
  d.inv_global_scale = P.inv_global_scale;
  d.x_dm = P.x_dm;
  d.b_dm = P.b_dm;
  d.color_scale = P.color_scale;
  d.base_corr_x = P.base_corr_x;
  d.base_corr_b = P.base_corr_b;
  memcpy(d.quant_biases, P.ih.quant_bias, sizeof(d.quant_biases));
  const jxh::LoopFilter& lf = P.fh.lf;
  d.gab = lf.gab;
  for (int c = 0; c < 3; c++) {
    d.gab_w[c * 2] = lf.gab_w[c][0];
    d.gab_w[c * 2 + 1] = lf.gab_w[c][1];
    d.epf_channel_scale[c] = lf.epf_channel_scale[c];
    d.opsin_bias[c] = P.ih.opsin_bias[c];
  }
  d.epf_iters = int32_t(lf.epf_iters);
  d.epf_pass0_sigma_scale = lf.epf_pass0_sigma_scale;
  d.epf_pass2_sigma_scale = lf.epf_pass2_sigma_scale;
  d.epf_border_sad_mul = lf.epf_border_sad_mul;
  for (int i = 0; i < 9; i++) d.opsin_inv[i] = P.ih.inv_opsin[i] * (255.0f / P.ih.intensity_target);
  // (frames of images that are not xyb_encoded: their own colour transform, frame_header.h:176-184, and no transfer function)
  d.linear_output = P.ih.xyb_encoded ? (P.ih.linear_tf ? 1 : 0) : (P.fh.ycbcr ? 2 : 3);
  d.band_group_row_begin = group_row_begin;
  d.band_group_row_end = group_row_end;
  d.has_noise = P.has_noise ? 1 : 0;
  FillSplines(P.has_splines, P.splines, &d.splines);
  memset(&d.patches, 0, sizeof(d.patches));
  if (P.has_patches && !P.patches.pos.empty()) {
    d.patches.num_positions = uint32_t(P.patches.pos.size());
    d.patches.num_row_entries = uint32_t(P.patch_row_list.size());
    d.patches.records = P.patch_records.data();
    d.patches.row_start = P.patch_row_start.data();
    d.patches.row_list = P.patch_row_list.data();
    for (int i = 0; i < 4; i++) {
      d.patches.slot_planes[i] = P.patch_src[i];
      d.patches.slot_w[i] = P.patch_src_w[i];
      d.patches.slot_h[i] = P.patch_src_h[i];
      d.patches.slot_alpha[i] = P.patch_src_alpha[i];
    }
    d.patches.uses_alpha = PatchesBlendAlpha(P) ? 1 : 0;
    d.patches.premultiplied = (d.patches.uses_alpha && P.ih.extra[0].alpha_associated) ? 1 : 0;
  }
  memcpy(d.noise_lut, P.noise_lut, sizeof(d.noise_lut));
  // dec_frame.cc:160-168: the number of visible frames before this one, and of invisible ones since (none are accepted)
  d.noise_frame_index[0] = uint32_t(P.frame_index);
  d.noise_frame_index[1] = uint32_t(P.nonvisible_index);
  std::vector<float> ups_kernel;
  if (P.fh.upsampling != 1) {
    UpsamplingKernels(P.fh.upsampling, P.ih, &ups_kernel);
    d.upsampling = P.fh.upsampling;
    d.out_xsize = uint32_t(P.ih.xsize);
    d.out_ysize = uint32_t(P.ih.ysize);
    d.upsampling_kernel = ups_kernel.data();
  }
  int r = jxlhip_frame_upload(ctx, &d);  // (copies everything it keeps: `pd` and the other locals may go)
  if (r) g_last_error = "jxlhip_frame_upload failed (" + std::to_string(r) + ")";
  return r;
}

// Decodes a coded ICC profile (the byte stream that follows the headers of an image with want_icc: jxh_icc.h). Returns
// 0 and the profile size / the exact number of bits the coded form takes; copies the profile when `out` has room.
int jxlamd_icc_decode(const uint8_t* coded, size_t size, uint8_t* out, size_t out_size, size_t* profile_size, size_t* coded_bits) {
  g_last_error.clear();
  try {
    std::vector<uint8_t> padded(coded, coded + size);
    padded.resize(size + 16, 0);
    jxh::BitReader br(padded.data(), size);
    std::vector<uint8_t> icc;
    jxh::ReadIcc(br, &icc);
    if (profile_size) *profile_size = icc.size();
    if (coded_bits) *coded_bits = br.BitPos();
    if (out && out_size >= icc.size()) memcpy(out, icc.data(), icc.size());
    return 0;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return 1;
  }
}

// ------------------------------------------------------------------------------------------------ Modular frames
}  // extern "C"
struct JxlAmdModFrame {
  jxh::ModFramePlan plan;
  const uint8_t* data = nullptr;
};
extern "C" {
int jxlamd_modframe_parse(const uint8_t* data, size_t size, JxlAmdModFrame** out) {
  return jxlamd_modframe_parse_at(data, size, 0, 0, out);
}
int jxlamd_modframe_set_patch_sources(JxlAmdModFrame* f, const float* const* planes, const uint32_t* xs, const uint32_t* ys) {
  g_last_error.clear();
  jxh::ModFramePlan& P = f->plan;
  for (int i = 0; i < 4; i++) {
    P.patch_src[i] = planes[i];
    P.patch_src_w[i] = planes[i] ? xs[i] : 0;
    P.patch_src_h[i] = planes[i] ? ys[i] : 0;
  }
  for (const jxh::PatchRef& r : P.patches.refs) {  // dec_patch_dictionary.cc:63-83
    if (!P.patch_src[r.slot]) {
      g_last_error = "patches: the reference frame is missing";
      return 2;
    }
    if (uint64_t(r.x0) + r.xsize > P.patch_src_w[r.slot] || uint64_t(r.y0) + r.ysize > P.patch_src_h[r.slot]) {
      g_last_error = "patches: rectangle outside the reference frame";
      return 2;
    }
  }
  return 0;
}

int jxlamd_modframe_parse_at(const uint8_t* data, size_t size, size_t frame_pos, size_t frame_index, JxlAmdModFrame** out) {
  g_last_error.clear();
  if (!data || !out) {
    g_last_error = "invalid argument";
    return 1;
  }
  *out = nullptr;
  std::unique_ptr<JxlAmdModFrame> f(new JxlAmdModFrame);
  try {
    jxh::FrameParser head(data, size);  // (signature / container + image header)
    jxh::ImageHeader ih;
    size_t pos = head.ParseImageHeader(&ih);
    const bool first = !frame_pos || frame_pos == pos;
    if (frame_pos) {
      if (frame_pos < pos || frame_pos >= size) throw std::runtime_error("truncated frame");
      pos = frame_pos;
    }
    if (first && ih.have_preview) {  // (the preview frame: see jxlamd_frame_parse_at)
      ih.xsize = ih.preview_xsize;
      ih.ysize = ih.preview_ysize;
    }
    jxh::ModFrameParser parser(data, size);
    parser.ParseFrame(pos, ih, &f->plan, frame_index);
    if (first && ih.have_preview && (f->plan.fh.frame_type != 0 || f->plan.fh.custom_size || f->plan.fh.blend.mode != 0))
      throw std::runtime_error("invalid preview frame");
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return 2;
  }
  f->data = data;
  *out = f.release();
  return 0;
}
void jxlamd_modframe_placement(const JxlAmdModFrame* f, JxlAmdFramePlacement* p) {
  FillPlacement(f->plan.ih, f->plan.fh, f->plan.dim.xsize, f->plan.dim.ysize, p);
}
// The sized forms: the caller says how large ITS struct is; never more than that is written (a caller built against an
// older header, or a hand-written FFI mirror, gets the leading fields it knows about), and the return value is the size
// of this library's struct so that a binding can tell.
size_t jxlamd_sizeof_frame_placement(void) { return sizeof(JxlAmdFramePlacement); }
size_t jxlamd_frame_placement_sized(const JxlAmdFrame* f, void* out, size_t out_size) {
  JxlAmdFramePlacement p;
  memset(&p, 0, sizeof(p));
  jxlamd_frame_placement(f, &p);
  if (out) memcpy(out, &p, out_size < sizeof(p) ? out_size : sizeof(p));
  return sizeof(p);
}
size_t jxlamd_modframe_placement_sized(const JxlAmdModFrame* f, void* out, size_t out_size) {
  JxlAmdFramePlacement p;
  memset(&p, 0, sizeof(p));
  jxlamd_modframe_placement(f, &p);
  if (out) memcpy(out, &p, out_size < sizeof(p) ? out_size : sizeof(p));
  return sizeof(p);
}
void jxlamd_modframe_free(JxlAmdModFrame* f) { delete f; }
size_t jxlamd_modframe_end(const JxlAmdModFrame* f, uint32_t* t) {
  if (t) {
    t[0] = f->plan.fh.duration;
    t[1] = f->plan.fh.is_last;
    t[2] = f->plan.fh.timecode;
  }
  return f->plan.frame_end;
}
void jxlamd_modframe_info(const JxlAmdModFrame* f, uint32_t* info) {
  const jxh::ModFramePlan& P = f->plan;
  info[0] = uint32_t(P.dim.xsize);
  info[1] = uint32_t(P.dim.ysize);
  info[2] = P.num_color;
  info[3] = P.has_alpha;
  info[4] = P.ih.bits;
  info[5] = uint32_t(P.streams.size());
  info[6] = uint32_t(P.buffers.size());
  info[7] = uint32_t(P.ops.size());
  info[8] = uint32_t(P.ih.extra.size());
  uint64_t total = 0;
  for (uint32_t s : P.section_size) total += s;
  info[9] = uint32_t(total);
  // largest symbol-table set (32-bit words as the stream kernel stages them), LZ77 anywhere, largest tree
  uint32_t words = 0, lz = 0, nodes = 0;
  for (size_t i = 0; i < P.codes.size(); i++) {
    const jxh::EntropyCode& c = P.codes[i];
    if (P.trees[i].empty()) continue;
    std::vector<uint32_t> ptab;
    if (c.use_prefix)
      for (const jxh::PrefixCode& pc : c.prefix) (void)jxh::AppendPrefixTables(pc, &ptab);
    const uint32_t w = c.use_prefix ? uint32_t(ptab.size() + 2 * c.num_clusters) : uint32_t((c.num_clusters << c.log_alpha) * 2 + c.num_clusters);
    words = std::max(words, w);
    lz |= c.lz77 ? 1u : 0u;
    nodes = std::max<uint32_t>(nodes, uint32_t(P.trees[i].size()));
  }
  info[10] = words;
  info[11] = lz;
  info[12] = nodes;
}
uint32_t jxlamd_modframe_extra_buffer(const JxlAmdModFrame* f, uint32_t index) {
  return f && index < f->plan.extra_buffer.size() ? f->plan.extra_buffer[index] : 0xFFFFFFFFu;
}
int jxlamd_modframe_upload(const JxlAmdModFrame* f, JxlHipContext* ctx) {
  g_last_error.clear();
  if (!f || !ctx) {
    g_last_error = "invalid argument";
    return 1;
  }
  const jxh::ModFramePlan& P = f->plan;
  JxlHipModFrameDesc d;
  memset(&d, 0, sizeof(d));
  d.xsize = uint32_t(P.dim.xsize);
  d.ysize = uint32_t(P.dim.ysize);
  d.codestream = f->data;
  d.section_offset = P.section_offset.data();
  d.section_size = P.section_size.data();
  d.num_sections = uint32_t(P.section_size.size());
  // trees
  std::vector<std::vector<JxlHipModTreeNode>> trees(P.trees.size());
  std::vector<const JxlHipModTreeNode*> tree_ptr(P.trees.size());
  std::vector<uint32_t> tree_size(P.trees.size());
  for (size_t i = 0; i < P.trees.size(); i++) {
    for (const jxh::TreeNode& n : P.trees[i]) {
      JxlHipModTreeNode o;
      o.property = n.property;
      o.splitval = n.splitval;
      o.lchild = n.lchild;
      o.rchild = n.rchild;
      o.predictor = n.predictor;
      o.offset = int32_t(n.offset);
      o.multiplier = n.multiplier;
      o.pad = 0;
      trees[i].push_back(o);
    }
    tree_ptr[i] = trees[i].data();
    tree_size[i] = uint32_t(trees[i].size());
  }
  d.trees = tree_ptr.data();
  d.tree_size = tree_size.data();
  d.num_trees = uint32_t(trees.size());
  // codes
  std::vector<JxlHipModCode> codes(P.codes.size());
  std::vector<std::vector<uint32_t>> cfgs(P.codes.size()), ptabs(P.codes.size()), poffs(P.codes.size());
  for (size_t i = 0; i < P.codes.size(); i++) {
    const jxh::EntropyCode& c = P.codes[i];
    JxlHipModCode& o = codes[i];
    memset(&o, 0, sizeof(o));
    o.ctx_map = c.ctx_map.data();
    o.ctx_map_size = uint32_t(c.ctx_map.size());
    o.num_clusters = uint32_t(c.num_clusters);
    o.use_prefix = c.use_prefix ? 1 : 0;
    o.log_alpha = uint32_t(c.log_alpha);
    o.alias = c.alias.data();
    for (const jxh::HybridCfg& h : c.cfg) cfgs[i].push_back(h.split_exp | (h.msb << 8) | (h.lsb << 16));
    o.uint_cfg = cfgs[i].data();
    if (c.use_prefix) {
      for (const jxh::PrefixCode& pc : c.prefix) poffs[i].push_back(jxh::AppendPrefixTables(pc, &ptabs[i]));
      o.prefix_table = ptabs[i].data();
      o.prefix_table_size = uint32_t(ptabs[i].size());
      o.prefix_offset = poffs[i].data();
    }
    o.lz77 = c.lz77 ? 1 : 0;
    o.lz_min_symbol = c.lz_min_symbol;
    o.lz_min_length = c.lz_min_length;
    o.lz_len_cfg = c.lz_len_cfg.split_exp | (c.lz_len_cfg.msb << 8) | (c.lz_len_cfg.lsb << 16);
    o.lz_dist_ctx = c.lz_dist_ctx;
  }
  d.codes = codes.data();
  d.num_codes = uint32_t(codes.size());
  std::vector<JxlHipModBuffer> buffers;
  for (const auto& b : P.buffers) buffers.push_back({b.first, b.second});
  d.buffers = buffers.data();
  d.num_buffers = uint32_t(buffers.size());
  static_assert(sizeof(jxh::ModPlanRect) == sizeof(JxlHipModRect), "rect layout");
  d.rects = reinterpret_cast<const JxlHipModRect*>(P.rects.data());
  d.num_rects = uint32_t(P.rects.size());
  static_assert(sizeof(jxh::ModPlanStream) == sizeof(JxlHipModStream), "stream layout");
  d.streams = reinterpret_cast<const JxlHipModStream*>(P.streams.data());
  d.num_streams = uint32_t(P.streams.size());
  std::vector<JxlHipModOp> ops;
  for (const jxh::ModPlanOp& q : P.ops) {
    JxlHipModOp o;
    memset(&o, 0, sizeof(o));
    o.kind = q.kind;
    memcpy(o.buf, q.buf, sizeof(o.buf));
    o.x0 = q.x0;
    o.y0 = q.y0;
    o.w = q.w;
    o.h = q.h;
    o.param = q.param;
    o.nb = q.nb;
    o.bit_depth = q.bit_depth;
    o.after_stream = 0xFFFFFFFFu;
    ops.push_back(o);
  }
  d.ops = ops.data();
  d.num_ops = uint32_t(ops.size());
  memcpy(d.out_buffer, P.out_buffer, sizeof(d.out_buffer));
  d.num_color = P.num_color;
  d.has_alpha = P.has_alpha;
  d.bits = P.ih.bits | (P.ih.floating ? P.ih.exp_bits << 8 : 0u);
  d.alpha_bits = P.alpha_bits;
  FillSplines(P.has_splines, P.splines, &d.splines);
  memset(&d.patches, 0, sizeof(d.patches));
  if (P.has_patches && !P.patches.pos.empty()) {
    d.patches.num_positions = uint32_t(P.patches.pos.size());
    d.patches.num_row_entries = uint32_t(P.patch_row_list.size());
    d.patches.records = P.patch_records.data();
    d.patches.row_start = P.patch_row_start.data();
    d.patches.row_list = P.patch_row_list.data();
    for (int i = 0; i < 4; i++) {
      d.patches.slot_planes[i] = P.patch_src[i];
      d.patches.slot_w[i] = P.patch_src_w[i];
      d.patches.slot_h[i] = P.patch_src_h[i];
    }
  }
  d.xyb = P.xyb ? 1 : 0;
  for (int c = 0; c < 3; c++) {
    d.xyb_factor[c] = P.dc_quant[c];
    d.opsin_bias[c] = P.ih.opsin_bias[c];
  }
  for (int i = 0; i < 9; i++) d.opsin_inv[i] = P.ih.inv_opsin[i] * (255.0f / P.ih.intensity_target);
  d.linear_output = P.ih.linear_tf ? 1 : 0;
  const int r = jxlhip_modular_upload(ctx, &d);
  if (r) g_last_error = "jxlhip_modular_upload failed (" + std::to_string(r) + ")";
  return r;
}

// ------------------------------------------------------------------------------------------------ JxlDecoder
void jxlamd_frame_set_linear_output(JxlAmdFrame* f, int linear) {
  if (f) f->plan.ih.linear_tf = linear != 0;
}
int jxlamd_frame_extra_pending(const JxlAmdFrame* f) { return f && f->plan.extra_pending ? 1 : 0; }
int jxlamd_frame_finish_extra(JxlAmdFrame* f, JxlHipContext* ctx) { return jxlamd_frame_finish_extra_mt(f, ctx, nullptr, nullptr); }
int jxlamd_frame_finish_extra_mt(JxlAmdFrame* f, JxlHipContext* ctx, JxlParallelRunner runner, void* runner_opaque) {
  g_last_error.clear();
  if (!f || !ctx) return 1;
  if (!f->plan.extra_pending) return 0;
  std::vector<uint32_t> bits(f->plan.section_size.size());
  int r = jxlhip_get_section_end_bits(ctx, bits.data(), bits.size());
  if (r) return r;
  try {
    jxh::FrameParser::FinishExtraChannels(f->data, &f->plan, bits.data(), MakeParallelFor(runner, runner_opaque));
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return 2;
  }
  return 0;
}
const int32_t* jxlamd_frame_extra_plane(const JxlAmdFrame* f, uint32_t index) {
  if (!f || f->plan.extra_pending || index >= f->plan.extra.ch.size()) return nullptr;
  const jxh::MChannel& c = f->plan.extra.ch[index];
  uint32_t whu[3];
  jxlamd_frame_extra_dims(f, index, whu);
  if (c.w != whu[0] || c.h != whu[1]) return nullptr;
  return c.d.data();
}
int jxlamd_frame_extra_dims(const JxlAmdFrame* f, uint32_t index, uint32_t* whu) {
  if (!f || !whu || index >= f->plan.ih.extra.size()) return 1;
  const jxh::FramePlan& P = f->plan;
  const uint32_t ups = P.fh.ec_upsampling.empty() ? 1 : P.fh.ec_upsampling[index];
  whu[0] = uint32_t((size_t(P.fh.xsize) + ups - 1) / ups);
  whu[1] = uint32_t((size_t(P.fh.ysize) + ups - 1) / ups);
  whu[2] = ups;
  return 0;
}
int jxlamd_upsampling_kernels(const JxlAmdFrame* f, uint32_t factor, float* kernels) {
  if (!f || !kernels || (factor != 2 && factor != 4 && factor != 8)) return 1;
  std::vector<float> k;
  UpsamplingKernels(factor, f->plan.ih, &k);
  memcpy(kernels, k.data(), k.size() * sizeof(float));
  return 0;
}
}  // extern "C"

namespace {
// Growable byte buffer on the caller's JxlMemoryManager (memory_manager.h:51-65): the assembled codestream lives here.
struct MmBytes {
  JxlMemoryManager mm{};
  uint8_t* p = nullptr;
  size_t size = 0, cap = 0;
  bool Append(const uint8_t* src, size_t n) {
    if (size + n > cap) {
      size_t want = cap ? cap * 2 : 4096;
      while (want < size + n) want *= 2;
      uint8_t* q = static_cast<uint8_t*>(mm.alloc(mm.opaque, want));
      if (!q) return false;
      if (size) memcpy(q, p, size);
      if (p) mm.free(mm.opaque, p);
      p = q;
      cap = want;
    }
    if (n) memcpy(p + size, src, n);
    size += n;
    return true;
  }
  void Clear() {
    if (p) mm.free(mm.opaque, p);
    p = nullptr;
    size = cap = 0;
  }
};
void* DefaultAlloc(void*, size_t n) { return malloc(n); }
void DefaultFree(void*, void* p) { free(p); }
}  // namespace

struct JxlDecoderStruct {
  JxlMemoryManager mm{};
  JxlParallelRunner runner = nullptr;
  void* runner_opaque = nullptr;
  int events = 0;
  // ---- input (decode.cc:1574-1595): the caller's current buffer; file_pos = file offset of in[0]
  const uint8_t* in = nullptr;
  size_t in_size = 0, in_pos = 0;
  uint64_t file_pos = 0;
  bool input_closed = false;
  // ---- container walk (decode.cc:1639-2157)
  int container = -1;  // -1 unknown, 0 bare codestream, 1 boxes
  enum BoxStage { kHeader, kFtyp, kJxlpIndex, kCodestream, kSkip };
  BoxStage box_stage = kHeader;
  bool box_unbounded = false;
  uint64_t box_remaining = 0, box_size_raw = 0, box_contents_size = 0;
  char box_type[4] = {0, 0, 0, 0}, box_decoded_type[4] = {0, 0, 0, 0};
  size_t box_count = 0;
  bool last_codestream_seen = false, have_box = false;
  uint32_t next_jxlp_index = 0;
  bool decompress_boxes = false;
  uint8_t* box_out = nullptr;
  size_t box_out_size = 0, box_out_pos = 0;
  bool box_out_set = false;
  // ---- codestream
  MmBytes cs;
  bool cs_complete = false;  // no more codestream bytes will come
  int stage = 0;  // 0 start, 1 after basic info, 2 after colour, 3 after frame header, 4 need buffer, 5 image done, 6 end
  bool error = false;
  jxh::ImageHeader ih;
  bool have_ih = false;
  size_t frame_pos = 0, frame_index = 0;  // where the current frame starts in `cs` (0 = behind the image header), its number
  size_t skip_frames = 0;                 // JxlDecoderSkipFrames
  // frames composed on a canvas (several frames, crops, blending): jxlhip_canvas_*; shown / invisible frame counters
  JxlHipCanvas* canvas = nullptr;
  bool canvas_mode = false, frame_shown = true, frame_skipped = false;
  size_t visible_index = 0, nonvisible_index = 0;
  JxlAmdFrame* frame = nullptr;        // a VarDCT frame ...
  JxlAmdModFrame* mframe = nullptr;    // ... or a Modular (lossless) one
  JxlHipContext* ctx = nullptr;
  // ---- settings
  bool keep_orientation = false, unpremul = false;
  bool preview_frame = false;  // the current frame is the image's preview (decode.cc:1266-1268): its own events and buffer
  bool got_preview = false;
  bool frame_partial = false;  // d->frame was parsed from a prefix of its bytes (events only; see JxlDecoderFlushImage)
  // JXL_DEC_FRAME_PROGRESSION (decode.cc:1421-1428,1492-1500): once per frame, when its DC image is there and its AC data is not
  JxlProgressiveDetail prog_detail = kDC;
  bool dc_progression_done = false;
  uint32_t passes_reported = 0;  // complete passes at the last progression step (the next pause point lies above it)
  size_t downsampling_target = 8;  // decode.cc:788
  bool coalescing = true;  // JxlDecoderSetCoalescing: false = every regular frame is delivered by itself, unblended
  int want_linear = -1;  // JxlDecoderSetOutputColorProfile: -1 = as coded
  JxlCmsInterface cms{};
  bool have_cms = false;
  JxlBitDepth bit_depth{JXL_BIT_DEPTH_FROM_PIXEL_FORMAT, 0, 0};
  // ---- output
  JxlPixelFormat fmt{};
  void* out_buf = nullptr;
  size_t out_size = 0;
  JxlImageOutCallback callback = nullptr;
  void* callback_opaque = nullptr;
  JxlImageOutInitCallback mt_init = nullptr;
  JxlImageOutRunCallback mt_run = nullptr;
  JxlImageOutDestroyCallback mt_destroy = nullptr;
  void* mt_init_opaque = nullptr;
  bool have_out = false;
  struct ExtraOut {
    JxlPixelFormat fmt;
    void* buf;
    size_t size;
  };
  std::vector<std::pair<uint32_t, ExtraOut>> extra_out;
};

namespace {
uint16_t F32ToF16(float f) {  // round to nearest even; overflow to infinity, denormals kept
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  u &= 0x7FFFFFFFu;
  if (u >= 0x7F800000u) return uint16_t(sign | 0x7C00u | (u > 0x7F800000u ? 0x200u : 0));
  if (u >= 0x477FF000u) return uint16_t(sign | 0x7C00u);
  if (u < 0x38800000u) {  // denormal half
    if (u < 0x33000000u) return uint16_t(sign);
    const uint32_t shift = 126 - (u >> 23), m = (u & 0x7FFFFFu) | 0x800000u;
    const uint32_t r = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    return uint16_t(sign | (r + ((rem > half || (rem == half && (r & 1))) ? 1 : 0)));
  }
  const uint32_t v = u - 0x38000000u, r = v >> 13, rem = v & 0x1FFFu;
  return uint16_t(sign | (r + ((rem > 0x1000u || (rem == 0x1000u && (r & 1))) ? 1 : 0)));
}
size_t SampleBytes(JxlDataType t) { return t == JXL_TYPE_UINT8 ? 1 : (t == JXL_TYPE_FLOAT ? 4 : 2); }
bool KnownType(JxlDataType t) { return t == JXL_TYPE_UINT8 || t == JXL_TYPE_UINT16 || t == JXL_TYPE_FLOAT16 || t == JXL_TYPE_FLOAT; }
size_t RowStride(const JxlPixelFormat& f, size_t xsize) {
  size_t stride = xsize * f.num_channels * SampleBytes(f.data_type);
  if (f.align > 1) stride = (stride + f.align - 1) / f.align * f.align;
  return stride;
}
bool IsBigEndian(const JxlPixelFormat& f) { return f.endianness == JXL_BIG_ENDIAN; }  // (the hosts of this library are little endian)
void ResetState(JxlDecoder* d) {
  if (d->frame) jxlamd_frame_free(d->frame);
  d->frame = nullptr;
  if (d->mframe) jxlamd_modframe_free(d->mframe);
  d->mframe = nullptr;
  d->events = 0;
  d->in = nullptr;
  d->in_size = d->in_pos = 0;
  d->file_pos = 0;
  d->input_closed = false;
  d->container = -1;
  d->box_stage = JxlDecoder::kHeader;
  d->box_unbounded = false;
  d->box_remaining = 0;
  d->box_count = 0;
  d->last_codestream_seen = d->have_box = false;
  d->next_jxlp_index = 0;
  d->box_out = nullptr;
  d->box_out_set = false;
  d->cs.size = 0;
  d->cs_complete = false;
  d->stage = 0;
  d->frame_pos = d->frame_index = d->skip_frames = 0;
  if (d->canvas) jxlhip_canvas_destroy(d->canvas);
  d->canvas = nullptr;
  d->canvas_mode = false;
  d->frame_shown = true;
  d->frame_skipped = false;
  d->frame_partial = false;
  d->dc_progression_done = false;
  d->passes_reported = 0;
  d->downsampling_target = 8;
  d->preview_frame = d->got_preview = false;
  d->visible_index = d->nonvisible_index = 0;
  d->error = false;
  d->have_ih = false;
  d->have_out = false;
  d->out_buf = nullptr;
  d->callback = nullptr;
  d->mt_run = nullptr;
  d->extra_out.clear();
  d->want_linear = -1;
  d->coalescing = true;  // decode.cc:834
  d->bit_depth = JxlBitDepth{JXL_BIT_DEPTH_FROM_PIXEL_FORMAT, 0, 0};
}
JxlDecoderStatus Fail(JxlDecoder* d, const std::string& why) {
  g_last_error = why;
  d->error = true;
  return JXL_DEC_ERROR;
}
uint32_t Be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

// Input step (container walk). Returns: 0 = consumed something (or changed state), 1 = nothing more can be done with the
// bytes at hand, 2 = event in *ev, 3 = error (sticky).
int StepInput(JxlDecoder* d, JxlDecoderStatus* ev) {
  const uint8_t* in = d->in ? d->in + d->in_pos : nullptr;
  size_t avail = d->in ? d->in_size - d->in_pos : 0;
  auto advance = [&](size_t n) {
    d->in_pos += n;
    in += n;
    avail -= n;
  };
  if (d->container < 0) {
    if (!d->in) return 1;
    const JxlSignature sig = JxlSignatureCheck(in, avail);
    if (sig == JXL_SIG_INVALID) {
      *ev = Fail(d, "invalid signature");
      return 3;
    }
    if (sig == JXL_SIG_NOT_ENOUGH_BYTES) return 1;
    d->container = sig == JXL_SIG_CONTAINER ? 1 : 0;
    return 0;
  }
  if (d->container == 0) {  // bare codestream: every byte belongs to it
    if (d->cs_complete || !avail) return 1;
    if (!d->cs.Append(in, avail)) {
      *ev = Fail(d, "out of memory");
      return 3;
    }
    advance(avail);
    return 0;
  }
  switch (d->box_stage) {
    case JxlDecoder::kHeader: {
      if (avail < 8) return 1;
      uint64_t bsize = Be32(in);
      size_t hdr = 8;
      if (bsize == 1) {
        if (avail < 16) return 1;
        bsize = (uint64_t(Be32(in + 8)) << 32) | Be32(in + 12);
        hdr = 16;
      }
      if (bsize != 0 && bsize < hdr) {
        *ev = Fail(d, "invalid box size");
        return 3;
      }
      const bool brob = !memcmp(in + 4, "brob", 4);
      if (brob && avail < hdr + 4) return 1;
      memcpy(d->box_type, in + 4, 4);
      memcpy(d->box_decoded_type, brob ? in + hdr : in + 4, 4);
      d->box_count++;
      if (d->box_count == 1 && memcmp(d->box_type, "JXL ", 4)) {
        *ev = Fail(d, "the first box must be the signature box");
        return 3;
      }
      if ((d->box_count == 2) != !memcmp(d->box_type, "ftyp", 4)) {
        *ev = Fail(d, "the ftyp box must come second");
        return 3;
      }
      d->box_unbounded = bsize == 0;
      d->box_size_raw = bsize;
      d->box_contents_size = bsize ? bsize - hdr : 0;
      d->box_remaining = d->box_contents_size;
      d->have_box = true;
      d->box_out_set = false;
      d->box_out = nullptr;
      advance(hdr);
      if (!memcmp(d->box_type, "ftyp", 4)) d->box_stage = JxlDecoder::kFtyp;
      else if (!memcmp(d->box_type, "jxlc", 4)) {
        if (d->last_codestream_seen) {
          *ev = Fail(d, "there can only be one jxlc box");
          return 3;
        }
        d->last_codestream_seen = true;
        d->box_stage = JxlDecoder::kCodestream;
      } else if (!memcmp(d->box_type, "jxlp", 4)) {
        if (d->last_codestream_seen) {
          *ev = Fail(d, "cannot have a jxlp box after the last one");
          return 3;
        }
        d->box_stage = JxlDecoder::kJxlpIndex;
      } else {
        d->box_stage = JxlDecoder::kSkip;
      }
      if (d->events & JXL_DEC_BOX) {
        *ev = JXL_DEC_BOX;
        return 2;
      }
      return 0;
    }
    case JxlDecoder::kFtyp: {
      if (d->box_contents_size < 12 && !d->box_unbounded) {
        *ev = Fail(d, "file type box too small");
        return 3;
      }
      if (avail < 8) return 1;
      if (memcmp(in, "jxl ", 4) || Be32(in + 4) > 1) {
        *ev = Fail(d, "unknown file type brand / version");
        return 3;
      }
      d->box_stage = JxlDecoder::kSkip;  // (the 8 bytes stay part of the contents a box buffer receives)
      return 0;
    }
    case JxlDecoder::kJxlpIndex: {
      if (!d->box_unbounded && d->box_contents_size < 4) {
        *ev = Fail(d, "jxlp box too small to contain its index");
        return 3;
      }
      if (avail < 4) return 1;
      const uint32_t index = Be32(in);
      if ((index & 0x7FFFFFFFu) != d->next_jxlp_index) {
        *ev = Fail(d, "unsupported: jxlp boxes out of order");
        return 3;
      }
      d->next_jxlp_index++;
      if (index & 0x80000000u) d->last_codestream_seen = true;
      advance(4);
      if (!d->box_unbounded) d->box_remaining -= 4;
      d->box_stage = JxlDecoder::kCodestream;
      return 0;
    }
    case JxlDecoder::kCodestream:
    case JxlDecoder::kSkip: {
      const bool to_cs = d->box_stage == JxlDecoder::kCodestream;
      if (!d->box_unbounded && d->box_remaining == 0) {
        if (to_cs && d->last_codestream_seen) d->cs_complete = true;
        d->box_stage = JxlDecoder::kHeader;
        d->have_box = false;
        return 0;
      }
      if (!avail) {
        if (d->box_unbounded && d->input_closed) {  // an unbounded box ends with the file
          if (to_cs && d->last_codestream_seen) d->cs_complete = true;
          d->box_unbounded = false;
          d->box_remaining = 0;
          return 0;
        }
        return 1;
      }
      size_t n = d->box_unbounded ? avail : size_t(std::min<uint64_t>(avail, d->box_remaining));
      if (to_cs) {
        if (!d->cs.Append(in, n)) {
          *ev = Fail(d, "out of memory");
          return 3;
        }
      } else if (d->box_out_set) {
        // contents as stored; a Brotli box asked for decompressed has no contents here (decode.h:262-300, no Brotli)
        const bool hidden = d->decompress_boxes && !memcmp(d->box_type, "brob", 4);
        if (!hidden) {
          const size_t room = d->box_out_size - d->box_out_pos;
          if (room == 0) {
            *ev = JXL_DEC_BOX_NEED_MORE_OUTPUT;
            return 2;
          }
          n = std::min(n, room);
          memcpy(d->box_out + d->box_out_pos, in, n);
          d->box_out_pos += n;
        }
      }
      advance(n);
      if (!d->box_unbounded) d->box_remaining -= n;
      return 0;
    }
  }
  return 1;
}

// How the requested pixel format maps on the device writer.
struct OutFormat {
  uint32_t type, nc, bits;
  int big_endian;
};
OutFormat MapFormat(const JxlDecoder* d, const JxlPixelFormat& f) {
  OutFormat o;
  o.type = uint32_t(f.data_type);
  o.nc = f.num_channels;
  o.bits = f.data_type == JXL_TYPE_UINT8 ? 8 : 16;
  if (f.data_type == JXL_TYPE_UINT8 || f.data_type == JXL_TYPE_UINT16) {
    if (d->bit_depth.type == JXL_BIT_DEPTH_FROM_CODESTREAM) o.bits = std::min<uint32_t>(d->ih.bits, o.bits);
    else if (d->bit_depth.type == JXL_BIT_DEPTH_CUSTOM) o.bits = d->bit_depth.bits_per_sample;
  }
  o.big_endian = IsBigEndian(f) ? 1 : 0;
  return o;
}

// Host-side conversion of one Modular-coded extra channel (integer samples) to the caller's sample type
// (dec_frame.cc:511-542 hands these channels to the same stage_write.cc conversions; they never touch the GPU stages).
void StoreExtraRow(const int32_t* src, size_t xs, uint32_t ch_bits, const JxlPixelFormat& f, uint32_t out_bits, uint8_t* dst) {
  const float inv = 1.0f / float((uint64_t(1) << ch_bits) - 1);
  for (size_t x = 0; x < xs; x++) {
    const float v = float(src[x]) * inv;
    if (f.data_type == JXL_TYPE_FLOAT) {
      uint32_t u;
      memcpy(&u, &v, 4);
      if (IsBigEndian(f)) u = __builtin_bswap32(u);
      memcpy(dst + x * 4, &u, 4);
    } else if (f.data_type == JXL_TYPE_FLOAT16) {
      uint16_t h = F32ToF16(v);
      if (IsBigEndian(f)) h = uint16_t((h << 8) | (h >> 8));
      memcpy(dst + x * 2, &h, 2);
    } else {
      const float mul = float((1u << out_bits) - 1u);
      const float t = std::min(std::max(v * mul, 0.0f), mul);
      const uint32_t u = uint32_t(std::nearbyint(t));
      if (f.data_type == JXL_TYPE_UINT8) dst[x] = uint8_t(u);
      else {
        uint16_t h = uint16_t(u);
        if (IsBigEndian(f)) h = uint16_t((h << 8) | (h >> 8));
        memcpy(dst + x * 2, &h, 2);
      }
    }
  }
}

// The orientation the output undoes (decode.cc:2233-2240): 1 when the caller keeps the coded orientation.
uint32_t UndoOrientation(const JxlDecoder* d) { return d->keep_orientation ? 1u : d->ih.orientation; }
size_t OrientedXsize(const JxlDecoder* d) { return UndoOrientation(d) > 4 ? d->ih.ysize : d->ih.xsize; }
size_t OrientedYsize(const JxlDecoder* d) { return UndoOrientation(d) > 4 ? d->ih.xsize : d->ih.ysize; }
void Placement(const JxlDecoder* d, JxlAmdFramePlacement* p);
// decode.cc:980-1001 GetCurrentDimensions: the image, or (coalescing off) the current frame's own upsampled size.
void CurrentDimensions(const JxlDecoder* d, size_t* xs, size_t* ys) {
  *xs = OrientedXsize(d);
  *ys = OrientedYsize(d);
  if ((!d->coalescing || d->preview_frame) && (d->frame || d->mframe)) {
    JxlAmdFramePlacement p;
    Placement(d, &p);
    *xs = p.xsize;
    *ys = p.ysize;
    if (UndoOrientation(d) > 4) std::swap(*xs, *ys);
  }
}

// Extra-channel planes take the same flips and transpose as the colour pixels (stage_write.cc:441-458, 664-699); the
// colour writer does them on the device, these small integer planes are permuted while they are stored.
std::vector<int32_t> OrientPlane(const int32_t* src, size_t xs, size_t ys, uint32_t orientation) {
  static const uint8_t kBits[9] = {0, 0, 1, 3, 2, 4, 6, 7, 5};  // mirror x | mirror y << 1 | transpose << 2
  const uint32_t bits = kBits[orientation];
  std::vector<int32_t> out(xs * ys);
  for (size_t y = 0; y < ys; y++) {
    const size_t oy = (bits & 2) ? ys - 1 - y : y;
    for (size_t x = 0; x < xs; x++) {
      const size_t ox = (bits & 1) ? xs - 1 - x : x;
      out[(bits & 4) ? ox * ys + oy : oy * xs + ox] = src[y * xs + x];
    }
  }
  return out;
}

// `xs`, `ys`: the size of the delivered (oriented) image.
JxlDecoderStatus DeliverPixels(JxlDecoder* d, const OutFormat& of, size_t xs, size_t ys) {
  const size_t bpp = of.nc * SampleBytes(d->fmt.data_type);
  int r;
  if (d->out_buf) {
    r = jxlhip_download_pixels(d->ctx, d->out_buf, RowStride(d->fmt, xs));
    if (r) return Fail(d, "download failed");
  } else {
    std::vector<uint8_t> px(xs * ys * bpp);
    r = jxlhip_download_pixels(d->ctx, px.data(), xs * bpp);
    if (r) return Fail(d, "download failed");
    void* run_opaque = nullptr;
    if (d->mt_run) {
      run_opaque = d->mt_init(d->mt_init_opaque, 1, xs);
      if (!run_opaque) return Fail(d, "image out init callback failed");
    }
    for (size_t y = 0; y < ys; y++) {
      if (d->mt_run) d->mt_run(run_opaque, 0, 0, y, xs, px.data() + y * xs * bpp);
      else d->callback(d->callback_opaque, 0, y, xs, px.data() + y * xs * bpp);
    }
    if (d->mt_run && d->mt_destroy) d->mt_destroy(run_opaque);
  }
  return JXL_DEC_FULL_IMAGE;
}

// A Modular (lossless) frame: every stream, the inverse transforms and the sample conversion run on the device.
JxlDecoderStatus BlendIntoCanvas(JxlDecoder* d);
const OutFormat kCanvasFormat = {0, 4, 0, 0};  // what a frame headed for the canvas is decoded into: f32 x 4, as coded

// JxlDecoderSetUnpremultiplyAlpha acts on an output with alpha of an image whose (first) alpha channel is associated
// (dec_frame.h:205-211, stage_write.cc:359-361).
bool WantsUnpremultiply(const JxlDecoder* d, const OutFormat& of) {
  if (!d->unpremul || !(of.nc == 2 || of.nc == 4)) return false;
  for (const auto& e : d->ih.extra)
    if (e.type == 0) return e.alpha_associated != 0;
  return false;
}

JxlDecoderStatus DecodeModularPixels(JxlDecoder* d, bool to_canvas) {
  const OutFormat of = to_canvas ? kCanvasFormat : MapFormat(d, d->fmt);
  int r = jxlhip_set_output_format(d->ctx, of.type, of.nc, of.bits, of.big_endian);
  if (!r) r = jxlhip_set_output_orientation(d->ctx, to_canvas ? 1 : UndoOrientation(d));
  if (!r) r = jxlhip_set_output_unpremultiply(d->ctx, !to_canvas && WantsUnpremultiply(d, of) ? 1 : 0);
  {
    JxlAmdFramePlacement pl;
    jxlamd_modframe_placement(d->mframe, &pl);
    if (!r) r = jxlhip_set_option(d->ctx, "keep_xyb_planes", (pl.frame_type == 2 || pl.frame_type == 1) ? 1 : 0);
  }
  if (!r && d->mframe->plan.has_patches) {  // the reference frames the patches read: the XYB slots of the canvas
    const float* planes[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t pw[4] = {0, 0, 0, 0}, ph[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < 4 && d->canvas; i++) jxlhip_canvas_xyb_source(d->canvas, i, &planes[i], &pw[i], &ph[i]);
    if (jxlamd_modframe_set_patch_sources(d->mframe, planes, pw, ph)) return Fail(d, g_last_error);
  }
  if (!r) r = jxlamd_modframe_upload(d->mframe, d->ctx);
  if (!r) r = jxlhip_modular_run(d->ctx);
  uint32_t info[16];
  jxlamd_modframe_info(d->mframe, info);
  std::vector<uint32_t> status(info[5] + 1);
  if (!r) r = jxlhip_modular_status(d->ctx, status.data(), nullptr, status.size());
  if (r) return Fail(d, "GPU decode failed (" + std::to_string(r) + ") " + g_last_error);
  if (to_canvas) return BlendIntoCanvas(d);
  const uint32_t orientation = UndoOrientation(d);
  const size_t xs = orientation > 4 ? info[1] : info[0], ys = orientation > 4 ? info[0] : info[1];
  for (const auto& eo : d->extra_out) {
    const uint32_t buffer = jxlamd_modframe_extra_buffer(d->mframe, eo.first);
    std::vector<int32_t> plane(xs * ys);
    if (buffer == 0xFFFFFFFFu || jxlhip_modular_download_buffer(d->ctx, buffer, plane.data(), plane.size())) return Fail(d, "extra channel unavailable");
    if (orientation != 1) plane = OrientPlane(plane.data(), info[0], info[1], orientation);
    const JxlPixelFormat& f = eo.second.fmt;
    const size_t stride = RowStride(f, xs);
    uint32_t bits = f.data_type == JXL_TYPE_UINT8 ? 8 : 16;
    if (d->bit_depth.type == JXL_BIT_DEPTH_FROM_CODESTREAM) bits = std::min(bits, d->ih.extra[eo.first].bits);
    else if (d->bit_depth.type == JXL_BIT_DEPTH_CUSTOM) bits = d->bit_depth.bits_per_sample;
    for (size_t y = 0; y < ys; y++)
      StoreExtraRow(plane.data() + y * xs, xs, d->ih.extra[eo.first].bits, f, bits, static_cast<uint8_t*>(eo.second.buf) + y * stride);
  }
  return DeliverPixels(d, of, xs, ys);
}

JxlDecoderStatus DecodePixels(JxlDecoder* d, bool to_canvas) {
  if (!d->ctx) {
    if (jxlhip_device_count() <= 0) return Fail(d, "no HIP device: libjxl_amd has no CPU decode path");
    const char* dev = getenv("JXLHIP_DEVICE");
    int r = jxlhip_ctx_create(dev ? atoi(dev) : 0, &d->ctx);
    if (r) return Fail(d, "jxlhip_ctx_create failed (" + std::to_string(r) + ")");
  }
  if (d->mframe) return DecodeModularPixels(d, to_canvas);
  const jxh::FramePlan& P = d->frame->plan;
  const OutFormat of = to_canvas ? kCanvasFormat : MapFormat(d, d->fmt);
  jxlamd_frame_set_indices(d->frame, uint32_t(d->visible_index), uint32_t(d->nonvisible_index));
  int alpha_ec = -1;
  for (size_t e = 0; e < d->ih.extra.size(); e++)
    if (d->ih.extra[e].type == 0) {
      alpha_ec = int(e);
      break;
    }
  const bool want_alpha = (of.nc == 2 || of.nc == 4) && alpha_ec >= 0;
  jxlamd_frame_set_linear_output(d->frame, d->want_linear >= 0 ? d->want_linear : (P.ih.linear_tf ? 1 : 0));
  const uint32_t orientation = to_canvas ? 1 : UndoOrientation(d);
  if (to_canvas && d->want_linear >= 0 && d->want_linear != (P.ih.linear_tf ? 1 : 0)) return Fail(d, "unsupported: blending with a changed transfer function");
  int r = jxlhip_set_output_format(d->ctx, of.type, of.nc, of.bits, of.big_endian);
  if (!r) r = jxlhip_set_output_orientation(d->ctx, orientation);
  if (!r) r = jxlhip_set_output_unpremultiply(d->ctx, !to_canvas && WantsUnpremultiply(d, of) ? 1 : 0);
  if (!r) r = jxlhip_set_alpha(d->ctx, nullptr, 0, 0);
  if (!r && P.has_patches) {  // the reference frames the patches read: the XYB slots of the canvas
    const float* planes[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t pw[4] = {0, 0, 0, 0}, ph[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < 4 && d->canvas; i++) jxlhip_canvas_xyb_source(d->canvas, i, &planes[i], &pw[i], &ph[i]);
    if (jxlamd_frame_set_patch_sources(d->frame, planes, pw, ph)) return Fail(d, g_last_error);
    const float* alphas[4] = {nullptr, nullptr, nullptr, nullptr};
    for (uint32_t i = 0; i < 4 && d->canvas; i++) jxlhip_canvas_xyb_alpha(d->canvas, i, &alphas[i]);
    if (jxlamd_frame_set_patch_alpha_sources(d->frame, alphas)) return Fail(d, g_last_error);
  }
  if (!r && P.use_dc_frame) {  // the DC frame decoded earlier: DC slot of the canvas (BlendIntoCanvas put it there)
    const float* planes = nullptr;
    uint32_t w = 0, h = 0;
    if (d->canvas) jxlhip_canvas_xyb_source(d->canvas, 4 + P.fh.dc_level, &planes, &w, &h);
    if (jxlamd_frame_set_dc_source(d->frame, planes, w, h)) return Fail(d, g_last_error);
  }
  if (!r) r = jxlamd_frame_upload(d->frame, d->ctx);
  if (!r) r = jxlhip_run_entropy(d->ctx);
  std::vector<uint32_t> flags(P.dim.num_groups);
  if (!r) r = jxlhip_get_errors(d->ctx, flags.data(), flags.size());
  if (r) return Fail(d, "GPU decode failed (" + std::to_string(r) + ")");
  const bool need_extra = want_alpha || (!to_canvas && !d->extra_out.empty());
  if (need_extra && jxlamd_frame_extra_pending(d->frame)) {
    r = jxlamd_frame_finish_extra_mt(d->frame, d->ctx, d->runner, d->runner_opaque);
    if (r) return Fail(d, "extra channels: " + g_last_error);
  }
  uint32_t out_wh[2];
  jxlamd_frame_out_size(d->frame, out_wh);
  size_t xs = out_wh[0], ys = out_wh[1];
  // An extra channel as floats in [0, 1] at the size it is coded in, and - for a channel with an upsampling factor
  // (frame_header.cc:265-283) - through the upsampling stage on the device (dec_cache.cc:172-190, 203-212), as the alpha plane
  // of the context or back to the host.
  const size_t img_xs = xs, img_ys = ys;  // (xs / ys are swapped below for a transposing orientation)
  auto extra_floats = [&, img_xs, img_ys](uint32_t ec, bool as_alpha, std::vector<float>* host) -> const char* {
    const size_t xs = img_xs, ys = img_ys;
    const int32_t* a = jxlamd_frame_extra_plane(d->frame, ec);
    uint32_t whu[3];
    if (!a || jxlamd_frame_extra_dims(d->frame, ec, whu)) return "extra channel unavailable";
    const size_t n = size_t(whu[0]) * whu[1];
    std::vector<float> af(n);
    const float inv = 1.0f / float((uint64_t(1) << d->ih.extra[ec].bits) - 1);
    for (size_t i = 0; i < n; i++) af[i] = float(a[i]) * inv;
    if (whu[2] == 1) {
      if (as_alpha && jxlhip_set_alpha(d->ctx, af.data(), uint32_t(xs), uint32_t(ys))) return "jxlhip_set_alpha failed";
      if (host) host->swap(af);
      return nullptr;
    }
    std::vector<float> kernels(size_t(whu[2]) * whu[2] * 25);
    if (jxlamd_upsampling_kernels(d->frame, whu[2], kernels.data())) return "upsampling kernels unavailable";
    if (host) host->resize(xs * ys);
    if (jxlhip_upsample_plane(d->ctx, af.data(), whu[0], whu[1], whu[2], kernels.data(), uint32_t(xs), uint32_t(ys), as_alpha ? 1 : 0,
                              host ? host->data() : nullptr))
      return "jxlhip_upsample_plane failed";
    return nullptr;
  };
  if (want_alpha) {
    if (const char* e = extra_floats(uint32_t(alpha_ec), true, nullptr)) return Fail(d, e);
  }
  r = jxlhip_run_transform(d->ctx);
  if (!r) r = jxlhip_run_filter_color(d->ctx);
  if (r) return Fail(d, "GPU decode failed (" + std::to_string(r) + ")");
  if (to_canvas) return BlendIntoCanvas(d);
  const size_t coded_xs = xs, coded_ys = ys;
  if (orientation > 4) std::swap(xs, ys);
  for (const auto& eo : d->extra_out) {
    const int32_t* p = jxlamd_frame_extra_plane(d->frame, eo.first);
    if (!p) return Fail(d, "extra channel unavailable");
    std::vector<int32_t> upsampled;
    uint32_t whu[3] = {0, 0, 1};
    jxlamd_frame_extra_dims(d->frame, eo.first, whu);
    const bool blended = int(eo.first) == alpha_ec && PatchesBlendAlpha(P);  // (the patch stage wrote the alpha channel)
    if (whu[2] != 1 || blended) {  // the channel from the device, back as integers of its bit depth (what the rows below are converted from)
      std::vector<float> f;
      if (blended) {
        f.resize(coded_xs * coded_ys);
        if (jxlhip_download_alpha(d->ctx, f.data())) return Fail(d, "jxlhip_download_alpha failed");
      } else if (const char* e = extra_floats(eo.first, false, &f)) return Fail(d, e);
      const float maxv = float((uint64_t(1) << d->ih.extra[eo.first].bits) - 1);
      upsampled.resize(f.size());
      for (size_t i = 0; i < f.size(); i++) upsampled[i] = int32_t(std::lrint(std::min(1.0f, std::max(0.0f, f[i])) * maxv));
      p = upsampled.data();
    }
    std::vector<int32_t> oriented;
    if (orientation != 1) {
      oriented = OrientPlane(p, coded_xs, coded_ys, orientation);
      p = oriented.data();
    }
    const JxlPixelFormat& f = eo.second.fmt;
    const size_t stride = RowStride(f, xs);
    uint32_t bits = f.data_type == JXL_TYPE_UINT8 ? 8 : 16;
    if (d->bit_depth.type == JXL_BIT_DEPTH_FROM_CODESTREAM) bits = std::min(bits, d->ih.extra[eo.first].bits);
    else if (d->bit_depth.type == JXL_BIT_DEPTH_CUSTOM) bits = d->bit_depth.bits_per_sample;
    for (size_t y = 0; y < ys; y++)
      StoreExtraRow(p + y * xs, xs, d->ih.extra[eo.first].bits, f, bits, static_cast<uint8_t*>(eo.second.buf) + y * stride);
  }
  return DeliverPixels(d, of, xs, ys);
}

void Placement(const JxlDecoder* d, JxlAmdFramePlacement* p) {
  if (d->frame) jxlamd_frame_placement(d->frame, p);
  else jxlamd_modframe_placement(d->mframe, p);
}
// frame_header.h:373-379
bool CanBeReferenced(const JxlAmdFramePlacement& p) { return !p.is_last && (p.duration == 0 || p.save_as_reference != 0); }
// blending.cc:22-40 NeedsBlending, plus: anything another frame may later be blended with
bool NeedsCanvas(const JxlAmdFramePlacement& p) { return p.custom_size || p.mode != 0 || p.alpha_mode != 0 || !p.is_last; }

// The frame the context just decoded (f32 x 4) goes onto the canvas: blending.cc / stage_blending.cc on the device.
JxlDecoderStatus BlendIntoCanvas(JxlDecoder* d) {
  JxlAmdFramePlacement p;
  Placement(d, &p);
  const bool has_alpha = !d->ih.extra.empty() && d->ih.extra[0].type == 0;
  if (d->ih.extra.size() > 1 || (!d->ih.extra.empty() && !has_alpha)) return Fail(d, "unsupported: blending with extra channels other than alpha");
  if (p.frame_type != 2 && CanBeReferenced(p) && p.save_before_color_transform) return Fail(d, "unsupported: regular frames saved before the colour transform");
  int r = 0;
  if (!d->canvas) {
    const char* dev = getenv("JXLHIP_DEVICE");
    r = jxlhip_canvas_create(dev ? atoi(dev) : 0, uint32_t(d->ih.xsize), uint32_t(d->ih.ysize), has_alpha ? 1 : 0,
                             has_alpha && d->ih.extra[0].alpha_associated ? 1 : 0, &d->canvas);
    if (r) return Fail(d, "jxlhip_canvas_create failed (" + std::to_string(r) + ")");
  }
  if (p.frame_type == 1) {  // kDCFrame (dec_cache.cc:221-224): kept before the colour transform as the DC image of level dc_level - 1
    r = jxlhip_canvas_save_xyb(d->canvas, d->ctx, 4 + p.dc_level - 1);
    if (r) return Fail(d, "jxlhip_canvas_save_xyb failed (" + std::to_string(r) + ")");
    return JXL_DEC_SUCCESS;
  }
  if (p.frame_type == 2) {  // kReferenceOnly: kept before the colour transform for the patches of later frames, never blended
    r = jxlhip_canvas_save_xyb(d->canvas, d->ctx, p.save_as_reference);
    if (r) return Fail(d, "jxlhip_canvas_save_xyb failed (" + std::to_string(r) + ")");
    return JXL_DEC_SUCCESS;
  }
  JxlHipBlend b;
  memset(&b, 0, sizeof(b));
  b.x0 = p.x0;
  b.y0 = p.y0;
  b.mode = p.mode;
  b.alpha_mode = p.alpha_mode;
  b.source = p.source;
  b.alpha_source = p.alpha_source;
  b.clamp = p.clamp;
  b.alpha_clamp = p.alpha_clamp;
  b.save_slot = CanBeReferenced(p) ? int32_t(p.save_as_reference) : -1;
  r = jxlhip_canvas_blend(d->canvas, d->ctx, &b);
  if (r) return Fail(d, "jxlhip_canvas_blend failed (" + std::to_string(r) + ")");
  return JXL_DEC_SUCCESS;
}

// The canvas in the caller's format: what DeliverPixels does for a frame decoded straight into that format.
JxlDecoderStatus DeliverCanvas(JxlDecoder* d) {
  const OutFormat of = MapFormat(d, d->fmt);
  if (jxlhip_canvas_set_unpremultiply(d->canvas, WantsUnpremultiply(d, of) ? 1 : 0)) return Fail(d, "canvas output");
  const uint32_t orientation = UndoOrientation(d);
  const size_t xs = OrientedXsize(d), ys = OrientedYsize(d);
  const size_t bpp = of.nc * SampleBytes(d->fmt.data_type);
  for (const auto& eo : d->extra_out) {  // the alpha plane of the canvas, as the integers its bit depth gives
    if (eo.first != 0 || d->ih.extra.empty() || d->ih.extra[0].type != 0) return Fail(d, "extra channel unavailable");
    std::vector<float> a(size_t(d->ih.xsize) * d->ih.ysize);
    if (jxlhip_canvas_download_alpha(d->canvas, a.data(), a.size())) return Fail(d, "download failed");
    const uint32_t ch_bits = d->ih.extra[0].bits;
    const float mul = float((uint64_t(1) << ch_bits) - 1);
    std::vector<int32_t> plane(a.size());
    for (size_t i = 0; i < a.size(); i++) plane[i] = int32_t(std::nearbyint(std::min(1.0f, std::max(0.0f, a[i])) * mul));
    if (orientation != 1) plane = OrientPlane(plane.data(), d->ih.xsize, d->ih.ysize, orientation);
    const JxlPixelFormat& f = eo.second.fmt;
    uint32_t bits = f.data_type == JXL_TYPE_UINT8 ? 8 : 16;
    if (d->bit_depth.type == JXL_BIT_DEPTH_FROM_CODESTREAM) bits = std::min(bits, ch_bits);
    else if (d->bit_depth.type == JXL_BIT_DEPTH_CUSTOM) bits = d->bit_depth.bits_per_sample;
    const size_t stride = RowStride(f, xs);
    for (size_t y = 0; y < ys; y++) StoreExtraRow(plane.data() + y * xs, xs, ch_bits, f, bits, static_cast<uint8_t*>(eo.second.buf) + y * stride);
  }
  if (d->out_buf) {
    if (jxlhip_canvas_download(d->canvas, of.type, of.nc, of.bits, of.big_endian, orientation, d->out_buf, RowStride(d->fmt, xs)))
      return Fail(d, "download failed");
  } else {
    std::vector<uint8_t> px(xs * ys * bpp);
    if (jxlhip_canvas_download(d->canvas, of.type, of.nc, of.bits, of.big_endian, orientation, px.data(), xs * bpp)) return Fail(d, "download failed");
    void* run_opaque = nullptr;
    if (d->mt_run) {
      run_opaque = d->mt_init(d->mt_init_opaque, 1, xs);
      if (!run_opaque) return Fail(d, "image out init callback failed");
    }
    for (size_t y = 0; y < ys; y++) {
      if (d->mt_run) d->mt_run(run_opaque, 0, 0, y, xs, px.data() + y * xs * bpp);
      else d->callback(d->callback_opaque, 0, y, xs, px.data() + y * xs * bpp);
    }
    if (d->mt_run && d->mt_destroy) d->mt_destroy(run_opaque);
  }
  return JXL_DEC_FULL_IMAGE;
}

// Drops the current frame and moves to the one behind it; false (stage 6) when it was the last.
bool NextFrame(JxlDecoder* d) {
  uint32_t t[3];
  const size_t end = d->frame ? jxlamd_frame_end(d->frame, t) : jxlamd_modframe_end(d->mframe, t);
  if (t[1]) {
    d->stage = 6;
    return false;
  }
  if (d->frame_shown) {  // dec_frame.cc:160-168
    d->visible_index++;
    d->nonvisible_index = 0;
  } else {
    d->nonvisible_index++;
  }
  if (d->frame) jxlamd_frame_free(d->frame);
  if (d->mframe) jxlamd_modframe_free(d->mframe);
  d->frame = nullptr;
  d->mframe = nullptr;
  d->frame_partial = false;
  d->dc_progression_done = false;
  d->passes_reported = 0;
  d->frame_pos = end;
  d->frame_index++;
  d->stage = 2;
  return true;
}

// Codestream step. Returns 0 = needs more codestream bytes, 1 = finished, 2 = event / status in *ev.
int StepCodestream(JxlDecoder* d, JxlDecoderStatus* ev) {
  if (d->stage == 0) {
    if (d->cs.size < 2) return 0;
    try {
      jxh::FrameParser parser(d->cs.p, d->cs.size);
      parser.ParseImageHeader(&d->ih);
      d->have_ih = true;
    } catch (const std::exception& e) {
      const std::string w = e.what();
      if (!d->cs_complete && w.find("truncated") != std::string::npos) return 0;
      *ev = Fail(d, w);
      return 2;
    }
    d->stage = 1;
    if (d->events & JXL_DEC_BASIC_INFO) {
      *ev = JXL_DEC_BASIC_INFO;
      return 2;
    }
  }
  if (d->stage == 1) {
    d->stage = 2;
    if (d->events & JXL_DEC_COLOR_ENCODING) {
      *ev = JXL_DEC_COLOR_ENCODING;
      return 2;
    }
  }
  for (;;) {
    if (d->stage == 2) {
      const bool is_preview = d->ih.have_preview && !d->got_preview;  // decode.cc:1266-1268: the codestream's first frame
      if (!(d->events & (JXL_DEC_FRAME | JXL_DEC_FULL_IMAGE | (is_preview ? JXL_DEC_PREVIEW_IMAGE : 0)))) {
        d->stage = 6;
        return 1;
      }
      int r = jxlamd_frame_parse_at(d->cs.p, d->cs.size, d->frame_pos, d->frame_index, d->runner, d->runner_opaque, &d->frame);
      if (r && g_last_error.find("Modular frames") != std::string::npos)  // a lossless frame: the Modular front-end takes it
        r = jxlamd_modframe_parse_at(d->cs.p, d->cs.size, d->frame_pos, d->frame_index, &d->mframe);
      if (r) {
        const std::string w = g_last_error;
        if (!d->cs_complete && w.find("truncated") != std::string::npos) {
          // Not all of the frame is here. Once its DC image is (decode.cc:1431-1530 reaches FrameStage::kFull with a part
          // of the sections), the frame is announced from that prefix: the caller may then ask for what there is with
          // JxlDecoderFlushImage. Frames nobody is shown (layers, skipped frames) wait for all their bytes.
          uint32_t present = 0;
          if (is_preview || !(d->events & JXL_DEC_FULL_IMAGE) || d->skip_frames > 0 ||
              jxlamd_frame_parse_partial_at(d->cs.p, d->cs.size, d->frame_pos, d->frame_index, d->runner, d->runner_opaque, &d->frame, &present))
            return 0;
          JxlAmdFramePlacement pp;
          jxlamd_frame_placement(d->frame, &pp);
          if (!(pp.is_last || pp.duration > 0 || (!d->coalescing && pp.frame_type == 0)) || (d->coalescing && (d->canvas_mode || NeedsCanvas(pp)))) {
            jxlamd_frame_free(d->frame);
            d->frame = nullptr;
            return 0;
          }
          d->frame_partial = true;
        } else {
          *ev = Fail(d, w);
          return 2;
        }
      }
      if (is_preview) {  // decode.cc:1334-1343, 1448-1450, 1554-1559: no FRAME event; its own buffer request and event
        d->frame_shown = false;  // (counted with the frames that are not shown: dec_frame.cc:160-168)
        d->frame_skipped = false;
        if (!(d->events & JXL_DEC_PREVIEW_IMAGE)) {
          d->got_preview = true;
          if (!NextFrame(d)) return 1;
          continue;
        }
        d->preview_frame = true;
        d->stage = 4;
        continue;
      }
      d->stage = 3;
      JxlAmdFramePlacement pl;
      Placement(d, &pl);
      // decode.cc:1346-1354 is_last_of_still: ... or any regular frame when the caller wants the layers themselves
      d->frame_shown = pl.is_last || pl.duration > 0 || (!d->coalescing && pl.frame_type == 0);
      if (d->coalescing && NeedsCanvas(pl)) d->canvas_mode = true;
      d->frame_skipped = false;
      if (d->skip_frames > 0) {  // decode.cc:1359-1408
        d->frame_skipped = true;
        if (d->frame_shown) d->skip_frames--;
      }
      if (!d->frame_shown || d->frame_skipped) {
        // no events for it; its pixels are still needed when a later frame may be blended with them
        // (coalescing off: nothing is blended, but the patches of later frames still read the reference-only frames)
        // (a DC frame is never referenced by blending, but it is the DC image of a frame behind it)
        if ((((d->coalescing ? d->canvas_mode : pl.frame_type == 2) && CanBeReferenced(pl)) || pl.frame_type == 1) && (d->events & JXL_DEC_FULL_IMAGE)) {
          const JxlDecoderStatus st = DecodePixels(d, true);
          if (st != JXL_DEC_SUCCESS) {
            *ev = st;
            return 2;
          }
        }
        if (!NextFrame(d)) return 1;
        continue;
      }
      if (d->events & JXL_DEC_FRAME) {
        *ev = JXL_DEC_FRAME;
        return 2;
      }
    }
    if (d->stage == 3) {
      if (!(d->events & JXL_DEC_FULL_IMAGE)) {  // decode.cc:1431-1437: headers only, the frame's bytes are skipped
        if (!NextFrame(d)) return 1;
        continue;
      }
      d->stage = 4;
    }
    if (d->stage == 4) {
      if (!d->have_out) {
        *ev = d->preview_frame ? JXL_DEC_NEED_PREVIEW_OUT_BUFFER : JXL_DEC_NEED_IMAGE_OUT_BUFFER;
        return 2;
      }
      if (d->preview_frame) {
        *ev = DecodePixels(d, false);
        if (*ev == JXL_DEC_FULL_IMAGE) {
          *ev = JXL_DEC_PREVIEW_IMAGE;
          d->got_preview = true;
          d->preview_frame = false;
          d->events &= ~JXL_DEC_PREVIEW_IMAGE;
          d->have_out = false;
          d->out_buf = nullptr;
          d->callback = nullptr;
          d->mt_run = nullptr;
          if (!NextFrame(d)) d->stage = 6;
        }
        return 2;
      }
      if (d->frame_partial) {  // the frame was announced from a prefix: the pixels need all of it
        uint32_t t[3];
        if (d->cs.size < jxlamd_frame_end(d->frame, t) && !d->cs_complete) {
          // the DC image is decoded and sections are missing: the one progressive step this decoder pauses at (a flush now
          // draws exactly it plus the groups that are whole; steps by pass are not offered: decode.h says the event "is not
          // guaranteed to trigger")
          if ((d->events & JXL_DEC_FRAME_PROGRESSION) && d->prog_detail >= kDC && !d->dc_progression_done && !d->canvas_mode) {
            d->dc_progression_done = true;
            d->downsampling_target = 8;
            *ev = JXL_DEC_FRAME_PROGRESSION;
            return 2;
          }
          // ... and a step by passes (decode.cc:1502-1512, dec_frame.h:144-200): every pass (kPasses) or the last pass of every
          // downsampling level the frame header names (kLastPasses), once every group has it
          if ((d->events & JXL_DEC_FRAME_PROGRESSION) && d->prog_detail >= kLastPasses && !d->canvas_mode) {
            uint32_t info[10];
            const uint32_t complete = jxlamd_frame_complete_passes(d->frame, d->cs.size, info), np = info[0];
            uint32_t next = np;  // the first pause point above the passes already reported
            if (d->prog_detail >= kPasses) {
              next = d->passes_reported + 1;
            } else {
              for (uint32_t i = 0; i < info[1] && i < 4; i++)
                if (info[6 + i] + 1 > d->passes_reported && info[6 + i] + 1 < next) next = info[6 + i] + 1;
            }
            if (complete >= next && complete < np) {
              d->passes_reported = complete;
              uint32_t target = 8;  // frame_header.h:286-295
              for (uint32_t i = 0; i < info[1] && i < 4; i++)
                if (complete > info[6 + i] && info[2 + i] < target) target = info[2 + i];
              d->downsampling_target = target;
              *ev = JXL_DEC_FRAME_PROGRESSION;
              return 2;
            }
          }
          return 0;
        }
        JxlAmdFrame* whole = nullptr;
        if (jxlamd_frame_parse_at(d->cs.p, d->cs.size, d->frame_pos, d->frame_index, d->runner, d->runner_opaque, &whole)) {
          const std::string w = g_last_error;
          if (!d->cs_complete && w.find("truncated") != std::string::npos) return 0;
          *ev = Fail(d, w);
          return 2;
        }
        jxlamd_frame_free(d->frame);
        d->frame = whole;
        d->frame_partial = false;
      }
      if (d->canvas_mode) {
        *ev = DecodePixels(d, true);
        if (*ev == JXL_DEC_SUCCESS) *ev = DeliverCanvas(d);
      } else {
        *ev = DecodePixels(d, false);
      }
      if (*ev == JXL_DEC_FULL_IMAGE) d->stage = 5;
      return 2;
    }
    if (d->stage == 5) {
      // decode.cc:1540-1550: the output buffers belong to one frame; the next one asks again
      d->have_out = false;
      d->out_buf = nullptr;
      d->callback = nullptr;
      d->mt_run = nullptr;
      d->extra_out.clear();
      if (!NextFrame(d)) return 1;
      continue;
    }
    break;
  }
  d->stage = 6;
  return 1;
}
}  // namespace

extern "C" {

uint32_t JxlDecoderVersion(void) { return 0 * 1000000 + 12 * 1000 + 0; }

JxlSignature JxlSignatureCheck(const uint8_t* buf, size_t len) {
  if (len == 0) return JXL_SIG_NOT_ENOUGH_BYTES;
  if (buf[0] == 0xFF) {
    if (len < 2) return JXL_SIG_NOT_ENOUGH_BYTES;
    return buf[1] == 0x0A ? JXL_SIG_CODESTREAM : JXL_SIG_INVALID;
  }
  static const uint8_t kContainer[12] = {0, 0, 0, 0xC, 'J', 'X', 'L', ' ', 0xD, 0xA, 0x87, 0xA};
  size_t n = len < 12 ? len : 12;
  if (memcmp(buf, kContainer, n) != 0) return JXL_SIG_INVALID;
  return len < 12 ? JXL_SIG_NOT_ENOUGH_BYTES : JXL_SIG_CONTAINER;
}

// decode.cc:844-868: the decoder object and its byte buffers come from the caller's memory manager.
JxlDecoder* JxlDecoderCreate(const JxlMemoryManager* memory_manager) {
  JxlMemoryManager mm{nullptr, DefaultAlloc, DefaultFree};
  if (memory_manager) {
    if (!memory_manager->alloc != !memory_manager->free) return nullptr;
    if (memory_manager->alloc) mm = *memory_manager;
  }
  void* mem = mm.alloc(mm.opaque, sizeof(JxlDecoder));
  if (!mem) return nullptr;
  JxlDecoder* d = new (mem) JxlDecoder;
  d->mm = mm;
  d->cs.mm = mm;
  return d;
}
void JxlDecoderReset(JxlDecoder* d) { ResetState(d); }
void JxlDecoderDestroy(JxlDecoder* d) {
  if (!d) return;
  ResetState(d);
  if (d->ctx) jxlhip_ctx_destroy(d->ctx);
  d->cs.Clear();
  const JxlMemoryManager mm = d->mm;
  d->~JxlDecoderStruct();
  mm.free(mm.opaque, d);
}
void JxlDecoderRewind(JxlDecoder* d) {
  const int ev = d->events;
  const JxlParallelRunner r = d->runner;
  void* ro = d->runner_opaque;
  const bool ko = d->keep_orientation, up = d->unpremul, db = d->decompress_boxes, co = d->coalescing;
  ResetState(d);
  d->coalescing = co;
  d->events = ev;
  d->runner = r;
  d->runner_opaque = ro;
  d->keep_orientation = ko;
  d->unpremul = up;
  d->decompress_boxes = db;
}
void JxlDecoderSkipFrames(JxlDecoder* d, size_t amount) { d->skip_frames += amount; }
// decode.cc:904-915: only the current frame is dropped (its bytes are stepped over and the decoder is back before the
// next frame header); a frame that later ones may be blended with still reaches the canvas / its reference slot, like the
// frames StepCodestream skips without events.
JxlDecoderStatus JxlDecoderSkipCurrentFrame(JxlDecoder* d) {
  if (!(d->stage >= 3 && d->stage < 5)) return JXL_DEC_ERROR;
  JxlAmdFramePlacement pl;
  Placement(d, &pl);
  if ((d->coalescing ? d->canvas_mode : pl.frame_type == 2) && CanBeReferenced(pl) && (d->events & JXL_DEC_FULL_IMAGE)) {
    if (DecodePixels(d, true) != JXL_DEC_SUCCESS) return JXL_DEC_ERROR;
  }
  if (d->frame_shown) {  // (is_last_of_still: the output buffers belonged to this frame)
    d->have_out = false;
    d->out_buf = nullptr;
    d->callback = nullptr;
    d->mt_run = nullptr;
    d->extra_out.clear();
  }
  NextFrame(d);  // stage 2 (the next frame's header) or 6 (this was the last frame)
  return JXL_DEC_SUCCESS;
}

JxlDecoderStatus JxlDecoderSetParallelRunner(JxlDecoder* d, JxlParallelRunner runner, void* opaque) {
  if (d->stage != 0) return JXL_DEC_ERROR;
  d->runner = runner;
  d->runner_opaque = opaque;
  return JXL_DEC_SUCCESS;
}
size_t JxlDecoderSizeHintBasicInfo(const JxlDecoder*) { return 98; }
JxlDecoderStatus JxlDecoderSubscribeEvents(JxlDecoder* d, int events) {
  if (d->stage != 0) return JXL_DEC_ERROR;
  if (events & 63) return JXL_DEC_ERROR;
  d->events = events;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetKeepOrientation(JxlDecoder* d, JXL_BOOL keep) {
  if (d->stage != 0) return JXL_DEC_ERROR;
  d->keep_orientation = keep != 0;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetUnpremultiplyAlpha(JxlDecoder* d, JXL_BOOL unpremul) {
  if (d->stage != 0) return JXL_DEC_ERROR;
  d->unpremul = unpremul != 0;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetRenderSpotcolors(JxlDecoder* d, JXL_BOOL) { return d->stage == 0 ? JXL_DEC_SUCCESS : JXL_DEC_ERROR; }
// decode.cc:973-979. Off: every regular frame is its own still (FRAME / NEED_IMAGE_OUT_BUFFER / FULL_IMAGE), delivered at
// its own size without blending; JxlDecoderGetFrameHeader then reports its crop, blend mode and reference slot so that
// the caller can compose (decode.cc:2725-2768).
JxlDecoderStatus JxlDecoderSetCoalescing(JxlDecoder* d, JXL_BOOL coalescing) {
  if (d->stage != 0) return JXL_DEC_ERROR;
  d->coalescing = coalescing != 0;
  return JXL_DEC_SUCCESS;
}

JxlDecoderStatus JxlDecoderSetInput(JxlDecoder* d, const uint8_t* data, size_t size) {
  if (d->in) return JXL_DEC_ERROR;
  d->in = data;
  d->in_size = size;
  d->in_pos = 0;
  return JXL_DEC_SUCCESS;
}
// decode.cc:1586: the unprocessed tail of the buffer (the caller re-supplies it with the next input).
size_t JxlDecoderReleaseInput(JxlDecoder* d) {
  if (!d->in) return 0;
  size_t remaining = d->in_size - d->in_pos;
  if (d->container == 0 && (d->frame || d->mframe) && d->stage >= 5 &&
      (d->frame ? d->frame->plan.fh.is_last : d->mframe->plan.fh.is_last)) {
    // bare codestream: bytes behind the LAST frame were copied along but are not part of it (behind any other frame
    // the next frame follows: those bytes are consumed)
    const uint64_t cs_end = d->frame ? d->frame->plan.frame_end : d->mframe->plan.frame_end;
    const uint64_t given = d->file_pos + d->in_pos;
    if (given > cs_end) remaining = size_t(std::min<uint64_t>(d->in_size, given - cs_end));
  }
  d->file_pos += d->in_size - remaining;
  d->in = nullptr;
  d->in_size = d->in_pos = 0;
  return remaining;
}
void JxlDecoderCloseInput(JxlDecoder* d) { d->input_closed = true; }

// decode.cc:2159: events in stream order; input is consumed box by box, the codestream it carries is assembled and
// decoded as soon as a unit (headers, whole frame) is complete.
JxlDecoderStatus JxlDecoderProcessInput(JxlDecoder* d) {
  if (d->error) return JXL_DEC_ERROR;
  for (;;) {
    JxlDecoderStatus ev = JXL_DEC_SUCCESS;
    bool cs_done = d->stage == 6;
    if (!cs_done && d->cs.size) {
      const int c = StepCodestream(d, &ev);
      if (c == 2) return ev;
      cs_done = c == 1;
    }
    if (cs_done && (d->container == 0 || !(d->events & JXL_DEC_BOX))) return JXL_DEC_SUCCESS;
    const int r = StepInput(d, &ev);
    if (r == 2 || r == 3) return ev;
    if (r == 0) continue;
    // nothing more to do with the bytes at hand
    if (d->container == 0 && d->input_closed && !d->cs_complete && d->cs.size) {
      d->cs_complete = true;  // a bare codestream ends with the input
      continue;
    }
    if (cs_done) {
      // (decode.cc:1770-1790) at a box boundary: success once the caller closed the input, else more boxes may follow
      if (d->input_closed || d->box_stage != JxlDecoder::kHeader) return d->input_closed ? JXL_DEC_SUCCESS : JXL_DEC_NEED_MORE_INPUT;
      return JXL_DEC_NEED_MORE_INPUT;
    }
    if (d->input_closed && (!d->in || d->in_pos == d->in_size)) {
      if (d->cs_complete || d->container < 0 || !d->cs.size) return Fail(d, "truncated input");
      d->cs_complete = true;  // let the codestream step report what is missing
      continue;
    }
    return JXL_DEC_NEED_MORE_INPUT;
  }
}

JxlDecoderStatus JxlDecoderGetBasicInfo(const JxlDecoder* d, JxlBasicInfo* info) {
  if (!d->have_ih) return JXL_DEC_NEED_MORE_INPUT;
  if (info) {
    memset(info, 0, sizeof(*info));
    info->have_container = d->container == 1;
    if (d->ih.have_preview) {  // decode.cc:2231, 2267-2269
      info->have_preview = JXL_TRUE;
      info->preview.xsize = d->ih.preview_xsize;
      info->preview.ysize = d->ih.preview_ysize;
    }
    info->xsize = uint32_t(OrientedXsize(d));
    info->ysize = uint32_t(OrientedYsize(d));
    info->bits_per_sample = d->ih.bits;
    info->exponent_bits_per_sample = d->ih.exp_bits;
    info->intensity_target = d->ih.intensity_target;
    info->uses_original_profile = !d->ih.xyb_encoded;
    info->orientation = d->keep_orientation ? JxlOrientation(d->ih.orientation) : JXL_ORIENT_IDENTITY;
    info->num_color_channels = d->ih.gray ? 1 : 3;
    info->num_extra_channels = uint32_t(d->ih.extra.size());
    for (const auto& e : d->ih.extra)
      if (e.type == 0) {
        info->alpha_bits = e.bits;
        info->alpha_exponent_bits = e.exp_bits;
        info->alpha_premultiplied = e.alpha_associated;
        break;
      }
    if (d->ih.have_animation) {  // decode.cc:2258-2266
      info->have_animation = JXL_TRUE;
      info->animation.tps_numerator = d->ih.anim_tps_num;
      info->animation.tps_denominator = d->ih.anim_tps_den;
      info->animation.num_loops = d->ih.anim_loops;
      info->animation.have_timecodes = d->ih.have_timecodes ? JXL_TRUE : JXL_FALSE;
    }
    info->intrinsic_xsize = info->xsize;
    info->intrinsic_ysize = info->ysize;
  }
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetExtraChannelInfo(const JxlDecoder* d, size_t index, JxlExtraChannelInfo* info) {
  if (!d->have_ih) return JXL_DEC_NEED_MORE_INPUT;
  if (index >= d->ih.extra.size()) return JXL_DEC_ERROR;
  if (info) {
    memset(info, 0, sizeof(*info));
    const jxh::ExtraChannel& e = d->ih.extra[index];
    info->type = JxlExtraChannelType(e.type);
    info->bits_per_sample = e.bits;
    info->exponent_bits_per_sample = e.floating ? e.exp_bits : 0;
    info->dim_shift = e.dim_shift;
    info->name_length = 0;
    info->alpha_premultiplied = e.alpha_associated;
  }
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetExtraChannelName(const JxlDecoder* d, size_t index, char* name, size_t size) {
  if (!d->have_ih) return JXL_DEC_NEED_MORE_INPUT;
  if (index >= d->ih.extra.size() || !name || !size) return JXL_DEC_ERROR;
  name[0] = 0;  // (names are parsed over, not kept)
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetExtraChannelBlendInfo(const JxlDecoder* d, size_t index, JxlBlendInfo* info) {
  if ((!d->frame && !d->mframe) || index >= d->ih.extra.size()) return JXL_DEC_ERROR;
  if (info) {  // decode.cc:2796-2812: the frame header's BlendingInfo of that channel
    memset(info, 0, sizeof(*info));
    const jxh::FrameHeader& fh = d->frame ? d->frame->plan.fh : d->mframe->plan.fh;
    if (index < fh.ec_blend.size()) {
      info->blendmode = JxlBlendMode(fh.ec_blend[index].mode);
      info->source = fh.ec_blend[index].source;
      info->alpha = fh.ec_blend[index].alpha_channel;
      info->clamp = fh.ec_blend[index].clamp ? JXL_TRUE : JXL_FALSE;
    }
  }
  return JXL_DEC_SUCCESS;
}
// The embedded ICC profile (ImageMetadata.color_encoding.want_icc, decoded by the host front-end: jxh_icc.h) answers for
// the ORIGINAL colours, and for the pixels of images that are not XYB-coded (their samples are passed through); the
// pixels of XYB images are sRGB / linear sRGB whatever the original profile was (no CMS is ever called).
static const std::vector<uint8_t>* EmbeddedIcc(const JxlDecoder* d, JxlColorProfileTarget target) {
  if (!d->ih.want_icc) return nullptr;
  if (target == JXL_COLOR_PROFILE_TARGET_ORIGINAL || !d->ih.xyb_encoded) return &d->ih.icc;
  return nullptr;
}
JxlDecoderStatus JxlDecoderGetColorAsEncodedProfile(const JxlDecoder* d, JxlColorProfileTarget target, JxlColorEncoding* ce) {
  if (!d->have_ih) return JXL_DEC_NEED_MORE_INPUT;
  if (EmbeddedIcc(d, target)) return JXL_DEC_ERROR;  // (decode.h:728-730: only the ICC form exists then)
  if (ce) {
    memset(ce, 0, sizeof(*ce));
    // the fields as coded (color_encoding_internal.cc:144-200; enum xy values: color_encoding_cms.h); an XYB image is
    // always (linear) sRGB here, anything else was refused with the headers
    const jxh::ImageHeader& ih = d->ih;
    ce->color_space = ih.gray ? JXL_COLOR_SPACE_GRAY : JXL_COLOR_SPACE_RGB;
    ce->white_point = JxlWhitePoint(ih.white_point);
    switch (ih.white_point) {
      case 2: ce->white_point_xy[0] = ih.white_xy[0] * 1e-6; ce->white_point_xy[1] = ih.white_xy[1] * 1e-6; break;
      case 10: ce->white_point_xy[0] = ce->white_point_xy[1] = 1.0 / 3; break;
      case 11: ce->white_point_xy[0] = 0.314; ce->white_point_xy[1] = 0.351; break;
      default: ce->white_point_xy[0] = 0.3127; ce->white_point_xy[1] = 0.3290; break;
    }
    ce->primaries = JxlPrimaries(ih.primaries);
    static const double kSrgb[6] = {0.639998686, 0.330010138, 0.300003784, 0.600003357, 0.150002046, 0.059997204};
    static const double k2100[6] = {0.708, 0.292, 0.170, 0.797, 0.131, 0.046};
    static const double kP3[6] = {0.680, 0.320, 0.265, 0.690, 0.150, 0.060};
    double xy[6];
    for (int i = 0; i < 6; i++) xy[i] = ih.primaries == 2 ? ih.primaries_xy[i] * 1e-6 : (ih.primaries == 9 ? k2100[i] : (ih.primaries == 11 ? kP3[i] : kSrgb[i]));
    ce->primaries_red_xy[0] = xy[0]; ce->primaries_red_xy[1] = xy[1];
    ce->primaries_green_xy[0] = xy[2]; ce->primaries_green_xy[1] = xy[3];
    ce->primaries_blue_xy[0] = xy[4]; ce->primaries_blue_xy[1] = xy[5];
    // the pixels (target DATA) follow JxlDecoderSetOutputColorProfile; the original profile is what the stream says
    if (target == JXL_COLOR_PROFILE_TARGET_DATA && d->want_linear >= 0) {
      ce->transfer_function = d->want_linear ? JXL_TRANSFER_FUNCTION_LINEAR : JXL_TRANSFER_FUNCTION_SRGB;
    } else if (ih.have_gamma) {
      ce->transfer_function = JXL_TRANSFER_FUNCTION_GAMMA;
      ce->gamma = ih.gamma * 1e-7;
    } else {
      ce->transfer_function = JxlTransferFunction(ih.transfer_function);
    }
    ce->rendering_intent = JxlRenderingIntent(ih.rendering_intent);
  }
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetICCProfileSize(const JxlDecoder* d, JxlColorProfileTarget target, size_t* size) {
  if (!d->have_ih) return JXL_DEC_NEED_MORE_INPUT;
  const std::vector<uint8_t>* icc = EmbeddedIcc(d, target);
  // (without an embedded profile: no ICC synthesis, the encoded profile is the primary representation: size 0 = none)
  if (size) *size = icc ? icc->size() : 0;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetColorAsICCProfile(const JxlDecoder* d, JxlColorProfileTarget target, uint8_t* out, size_t size) {
  if (!d->have_ih) return JXL_DEC_NEED_MORE_INPUT;
  const std::vector<uint8_t>* icc = EmbeddedIcc(d, target);
  if (!icc || !out || size < icc->size()) return JXL_DEC_ERROR;
  memcpy(out, icc->data(), icc->size());
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetPreferredColorProfile(JxlDecoder* d, const JxlColorEncoding* ce) {
  return JxlDecoderSetOutputColorProfile(d, ce, nullptr, 0);
}
JxlDecoderStatus JxlDecoderSetDesiredIntensityTarget(JxlDecoder*, float) { return JXL_DEC_SUCCESS; }
JxlDecoderStatus JxlDecoderSetCms(JxlDecoder* d, JxlCmsInterface cms) {
  d->cms = cms;
  d->have_cms = true;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetOutputColorProfile(JxlDecoder* d, const JxlColorEncoding* ce, const uint8_t* icc, size_t icc_size) {
  // decode.cc:2810: after the colour-encoding event, before the pixels. Only (linear or non-linear) sRGB output exists
  // here; anything else is refused so that the caller notices.
  if (!d->have_ih || d->stage > 4) return JXL_DEC_ERROR;
  if (!ce || icc || icc_size) return JXL_DEC_ERROR;
  if (ce->color_space != (d->ih.gray ? JXL_COLOR_SPACE_GRAY : JXL_COLOR_SPACE_RGB)) return JXL_DEC_ERROR;
  if (ce->white_point != JXL_WHITE_POINT_D65) return JXL_DEC_ERROR;
  if (ce->color_space == JXL_COLOR_SPACE_RGB && ce->primaries != JXL_PRIMARIES_SRGB) return JXL_DEC_ERROR;
  if (ce->transfer_function == JXL_TRANSFER_FUNCTION_LINEAR) d->want_linear = 1;
  else if (ce->transfer_function == JXL_TRANSFER_FUNCTION_SRGB) d->want_linear = 0;
  else return JXL_DEC_ERROR;
  return JXL_DEC_SUCCESS;
}
static const std::string& FrameName(const JxlDecoder* d) { return d->frame ? d->frame->plan.fh.name : d->mframe->plan.fh.name; }
JxlDecoderStatus JxlDecoderGetFrameHeader(const JxlDecoder* d, JxlFrameHeader* h) {
  if (!d->frame && !d->mframe) return JXL_DEC_ERROR;
  if (h) {
    memset(h, 0, sizeof(*h));
    uint32_t t[3];
    if (d->frame) jxlamd_frame_end(d->frame, t);
    else jxlamd_modframe_end(d->mframe, t);
    h->duration = t[0];  // decode.cc:2700-2712
    h->is_last = t[1] ? JXL_TRUE : JXL_FALSE;
    h->timecode = t[2];
    h->name_length = uint32_t(FrameName(d).size());
    size_t xs, ys;
    CurrentDimensions(d, &xs, &ys);  // decode.cc:2714-2722
    h->layer_info.xsize = uint32_t(xs);
    h->layer_info.ysize = uint32_t(ys);
    h->layer_info.blend_info.blendmode = JXL_BLEND_REPLACE;
    if (!d->coalescing) {  // decode.cc:2725-2768: the layer as coded
      JxlAmdFramePlacement p;
      Placement(d, &p);
      if (p.custom_size) {
        int64_t cx = p.x0, cy = p.y0;
        const uint32_t orientation = UndoOrientation(d);
        if (orientation != 1) {  // the crop offset in the oriented image
          const int64_t W = int64_t(OrientedXsize(d)), H = int64_t(OrientedYsize(d));
          if (orientation > 4) std::swap(cx, cy);
          const uint32_t o = (orientation - 1) & 3;
          if (o > 0 && o < 3) cx = W - int64_t(xs) - cx;
          if (o > 1) cy = H - int64_t(ys) - cy;
        }
        h->layer_info.have_crop = JXL_TRUE;
        h->layer_info.crop_x0 = int32_t(cx);
        h->layer_info.crop_y0 = int32_t(cy);
      }
      h->layer_info.blend_info.blendmode = JxlBlendMode(p.mode);
      h->layer_info.blend_info.source = p.source;
      h->layer_info.blend_info.alpha = (d->frame ? d->frame->plan.fh : d->mframe->plan.fh).blend.alpha_channel;
      h->layer_info.blend_info.clamp = p.clamp ? JXL_TRUE : JXL_FALSE;
      h->layer_info.save_as_reference = p.save_as_reference;
    }
  }
  return JXL_DEC_SUCCESS;
}
// decode.cc:2778-2792: the name with its terminating zero; the buffer must hold name_length + 1 bytes.
JxlDecoderStatus JxlDecoderGetFrameName(const JxlDecoder* d, char* name, size_t size) {
  if ((!d->frame && !d->mframe) || !name) return JXL_DEC_ERROR;
  const std::string& n = FrameName(d);
  if (size < n.size() + 1) return JXL_DEC_ERROR;
  memcpy(name, n.c_str(), n.size() + 1);
  return JXL_DEC_SUCCESS;
}
static JxlDecoderStatus CheckFormat(const JxlDecoder* d, const JxlPixelFormat* f);
// decode.cc:2522-2562: the preview frame's own buffer, asked for with JXL_DEC_NEED_PREVIEW_OUT_BUFFER.
JxlDecoderStatus JxlDecoderPreviewOutBufferSize(const JxlDecoder* d, const JxlPixelFormat* f, size_t* size) {
  if (CheckFormat(d, f) != JXL_DEC_SUCCESS || !size || !d->ih.have_preview) return JXL_DEC_ERROR;
  size_t xs = d->ih.preview_xsize, ys = d->ih.preview_ysize;
  if (UndoOrientation(d) > 4) std::swap(xs, ys);
  *size = RowStride(*f, xs) * (ys - 1) + xs * f->num_channels * SampleBytes(f->data_type);
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetPreviewOutBuffer(JxlDecoder* d, const JxlPixelFormat* f, void* buffer, size_t size) {
  size_t need = 0;
  if (!d->preview_frame || d->got_preview) return JXL_DEC_ERROR;  // "No preview out buffer needed at this time"
  if (JxlDecoderPreviewOutBufferSize(d, f, &need) != JXL_DEC_SUCCESS || !buffer || size < need) return JXL_DEC_ERROR;
  d->fmt = *f;
  d->out_buf = buffer;
  d->out_size = size;
  d->callback = nullptr;
  d->mt_run = nullptr;
  d->have_out = true;
  return JXL_DEC_SUCCESS;
}

static JxlDecoderStatus CheckFormat(const JxlDecoder* d, const JxlPixelFormat* f) {
  if (!d->have_ih || !f) return JXL_DEC_ERROR;
  if (f->num_channels < 1 || f->num_channels > 4 || !KnownType(f->data_type)) return JXL_DEC_ERROR;
  if (f->num_channels < 3 && !d->ih.gray) return JXL_DEC_ERROR;  // decode.cc:2512-2520: grayscale output of a colour image
  if (d->ih.gray && !d->mframe) return JXL_DEC_ERROR;            // (grey images: Modular frames only on this path)
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderImageOutBufferSize(const JxlDecoder* d, const JxlPixelFormat* f, size_t* size) {
  if (CheckFormat(d, f) != JXL_DEC_SUCCESS || !size) return JXL_DEC_ERROR;
  if (!d->coalescing && (d->stage < 3 || d->stage > 4)) return JXL_DEC_ERROR;  // decode.cc:2438-2441: frame dimensions unknown
  size_t xs, ys;
  CurrentDimensions(d, &xs, &ys);
  *size = RowStride(*f, xs) * (ys - 1) + xs * f->num_channels * SampleBytes(f->data_type);
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetImageOutBuffer(JxlDecoder* d, const JxlPixelFormat* f, void* buffer, size_t size) {
  size_t need = 0;
  if (JxlDecoderImageOutBufferSize(d, f, &need) != JXL_DEC_SUCCESS) return JXL_DEC_ERROR;
  if (!buffer || size < need) return JXL_DEC_ERROR;
  d->fmt = *f;
  d->out_buf = buffer;
  d->out_size = size;
  d->callback = nullptr;
  d->mt_run = nullptr;
  d->have_out = true;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetImageOutCallback(JxlDecoder* d, const JxlPixelFormat* f, JxlImageOutCallback cb, void* opaque) {
  if (!cb || CheckFormat(d, f) != JXL_DEC_SUCCESS) return JXL_DEC_ERROR;
  d->fmt = *f;
  d->callback = cb;
  d->callback_opaque = opaque;
  d->mt_run = nullptr;
  d->out_buf = nullptr;
  d->have_out = true;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetMultithreadedImageOutCallback(JxlDecoder* d, const JxlPixelFormat* f, JxlImageOutInitCallback init_cb,
                                                            JxlImageOutRunCallback run_cb, JxlImageOutDestroyCallback destroy_cb,
                                                            void* init_opaque) {
  if (!init_cb || !run_cb || CheckFormat(d, f) != JXL_DEC_SUCCESS) return JXL_DEC_ERROR;
  d->fmt = *f;
  d->mt_init = init_cb;
  d->mt_run = run_cb;
  d->mt_destroy = destroy_cb;
  d->mt_init_opaque = init_opaque;
  d->callback = nullptr;
  d->out_buf = nullptr;
  d->have_out = true;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderExtraChannelBufferSize(const JxlDecoder* d, const JxlPixelFormat* f, size_t* size, uint32_t index) {
  if (!d->have_ih || !f || !size || index >= d->ih.extra.size() || !KnownType(f->data_type)) return JXL_DEC_ERROR;
  JxlPixelFormat one = *f;
  one.num_channels = 1;  // decode.cc:2608-2625: the channel count of the format is ignored
  size_t xs, ys;
  CurrentDimensions(d, &xs, &ys);
  *size = RowStride(one, xs) * (ys - 1) + xs * SampleBytes(f->data_type);
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetExtraChannelBuffer(JxlDecoder* d, const JxlPixelFormat* f, void* buffer, size_t size, uint32_t index) {
  size_t need = 0;
  if (JxlDecoderExtraChannelBufferSize(d, f, &need, index) != JXL_DEC_SUCCESS || !buffer || size < need) return JXL_DEC_ERROR;
  JxlPixelFormat one = *f;
  one.num_channels = 1;
  for (auto& eo : d->extra_out)
    if (eo.first == index) {
      eo.second = JxlDecoder::ExtraOut{one, buffer, size};
      return JXL_DEC_SUCCESS;
    }
  d->extra_out.push_back({index, JxlDecoder::ExtraOut{one, buffer, size}});
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetDecompressBoxes(JxlDecoder* d, JXL_BOOL decompress) {
  d->decompress_boxes = decompress != 0;  // (no Brotli here: see JxlDecoderSetBoxBuffer)
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetProgressiveDetail(JxlDecoder* d, JxlProgressiveDetail detail) {  // decode.cc:2963-2973
  if (detail != kDC && detail != kLastPasses && detail != kPasses) return JXL_DEC_ERROR;
  d->prog_detail = detail;
  return JXL_DEC_SUCCESS;
}
size_t JxlDecoderGetIntendedDownsamplingRatio(JxlDecoder* d) { return d->downsampling_target; }
// decode.cc:2458-2475 / dec_frame.cc:735-795: what has arrived of the current frame, drawn into the caller's buffer. The
// groups are decoded from the passes they have (a group's passes are decoded in order as their sections arrive:
// dec_frame.cc:620-680); groups with none come from the DC image alone. Possible between the frame's NEED_IMAGE_OUT_BUFFER and its FULL_IMAGE, once the
// DC image is there (which is also when the frame is announced); JXL_DEC_ERROR = nothing was drawn, and is not fatal.
JxlDecoderStatus JxlDecoderFlushImage(JxlDecoder* d) {
  if (!d->have_out || d->stage != 4 || !d->frame || !d->frame_partial || d->canvas_mode || d->error) return JXL_DEC_ERROR;
  JxlAmdFrame* now = nullptr;
  uint32_t present = 0;
  if (jxlamd_frame_parse_partial_at(d->cs.p, d->cs.size, d->frame_pos, d->frame_index, d->runner, d->runner_opaque, &now, &present))
    return JXL_DEC_ERROR;
  jxlamd_frame_free(d->frame);  // (the earlier prefix: its section pointers may refer to a buffer that has since grown)
  d->frame = now;
  const JxlDecoderStatus st = DecodePixels(d, false);
  if (st != JXL_DEC_FULL_IMAGE) {
    d->error = false;
    return JXL_DEC_ERROR;
  }
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetImageOutBitDepth(JxlDecoder* d, const JxlBitDepth* bd) {
  if (!bd || !d->have_out) return JXL_DEC_ERROR;
  const uint32_t max_bits = d->fmt.data_type == JXL_TYPE_UINT8 ? 8 : 16;
  if (bd->type == JXL_BIT_DEPTH_CUSTOM) {
    if (d->fmt.data_type != JXL_TYPE_UINT8 && d->fmt.data_type != JXL_TYPE_UINT16) return JXL_DEC_ERROR;
    if (bd->bits_per_sample == 0 || bd->bits_per_sample > max_bits) return JXL_DEC_ERROR;
  } else if (bd->type == JXL_BIT_DEPTH_FROM_CODESTREAM) {
    if ((d->fmt.data_type == JXL_TYPE_UINT8 || d->fmt.data_type == JXL_TYPE_UINT16) && d->ih.bits > max_bits) return JXL_DEC_ERROR;
  } else if (bd->type != JXL_BIT_DEPTH_FROM_PIXEL_FORMAT) {
    return JXL_DEC_ERROR;
  }
  d->bit_depth = *bd;
  return JXL_DEC_SUCCESS;
}

// ---- boxes (decode.cc:2852-2990)
JxlDecoderStatus JxlDecoderSetBoxBuffer(JxlDecoder* d, uint8_t* data, size_t size) {
  if (d->box_out_set) return JXL_DEC_ERROR;  // release the previous buffer first
  if (!d->have_box || !(d->events & JXL_DEC_BOX)) return JXL_DEC_ERROR;
  d->box_out = data;
  d->box_out_size = size;
  d->box_out_pos = 0;
  d->box_out_set = true;
  return JXL_DEC_SUCCESS;
}
size_t JxlDecoderReleaseBoxBuffer(JxlDecoder* d) {
  if (!d->box_out_set) return 0;
  const size_t unused = d->box_out_size - d->box_out_pos;
  d->box_out_set = false;
  d->box_out = nullptr;
  return unused;
}
JxlDecoderStatus JxlDecoderGetBoxType(JxlDecoder* d, JxlBoxType type, JXL_BOOL decompressed) {
  if (!d->have_box) return JXL_DEC_ERROR;
  memcpy(type, decompressed ? d->box_decoded_type : d->box_type, 4);
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetBoxSizeRaw(const JxlDecoder* d, uint64_t* size) {
  if (!d->have_box) return JXL_DEC_ERROR;
  if (size) *size = d->box_size_raw;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetBoxSizeContents(const JxlDecoder* d, uint64_t* size) {
  if (!d->have_box) return JXL_DEC_ERROR;
  if (size) *size = d->box_contents_size;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetJPEGBuffer(JxlDecoder*, uint8_t*, size_t) { return JXL_DEC_ERROR; }
size_t JxlDecoderReleaseJPEGBuffer(JxlDecoder*) { return 0; }

// ------------------------------------------------------------------------------------------------ thread runners
}  // extern "C"

namespace {
// Fork-join pool; tasks are handed out with an atomic counter (dynamic self-scheduling), cf. reference
// lib/threads/thread_parallel_runner_internal.cc:68-108.
class Pool {
 public:
  explicit Pool(size_t n) { SetThreads(n); }
  ~Pool() { Stop(); }
  void SetThreads(size_t n) {
    Stop();
    stop_ = false;
    for (size_t i = 0; i < n; i++) workers_.emplace_back([this, i] { Loop(i); });
  }
  int Run(void* opaque, JxlParallelRunInit init, JxlParallelRunFunction func, uint32_t begin, uint32_t end) {
    if (begin > end) return JXL_PARALLEL_RET_RUNNER_ERROR;
    if (begin == end) return 0;
    const size_t nthreads = workers_.empty() ? 1 : workers_.size();
    if (init(opaque, nthreads) != 0) return JXL_PARALLEL_RET_RUNNER_ERROR;
    if (workers_.empty()) {
      for (uint32_t i = begin; i < end; i++) func(opaque, i, 0);
      return 0;
    }
    std::unique_lock<std::mutex> lk(mu_);
    opaque_ = opaque;
    func_ = func;
    next_.store(begin);
    end_ = end;
    pending_ = workers_.size();
    generation_++;
    cv_.notify_all();
    done_cv_.wait(lk, [this] { return pending_ == 0; });
    return 0;
  }

 private:
  void Loop(size_t id) {
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
      if (stop_) return;
      seen = generation_;
      void* opaque = opaque_;
      JxlParallelRunFunction func = func_;
      uint32_t end = end_;
      lk.unlock();
      for (;;) {
        uint32_t i = next_.fetch_add(1);
        if (i >= end) break;
        func(opaque, i, id);
      }
      lk.lock();
      if (--pending_ == 0) done_cv_.notify_all();
    }
  }
  void Stop() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
    workers_.clear();
  }
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  bool stop_ = false;
  uint64_t generation_ = 0;
  size_t pending_ = 0;
  void* opaque_ = nullptr;
  JxlParallelRunFunction func_ = nullptr;
  std::atomic<uint32_t> next_{0};
  uint32_t end_ = 0;
};
}  // namespace

extern "C" {
JxlParallelRetCode JxlThreadParallelRunner(void* runner_opaque, void* jpegxl_opaque, JxlParallelRunInit init,
                                           JxlParallelRunFunction func, uint32_t start_range, uint32_t end_range) {
  if (!runner_opaque) return JXL_PARALLEL_RET_RUNNER_ERROR;
  return static_cast<Pool*>(runner_opaque)->Run(jpegxl_opaque, init, func, start_range, end_range);
}
void* JxlThreadParallelRunnerCreate(const JxlMemoryManager*, size_t num_worker_threads) { return new (std::nothrow) Pool(num_worker_threads); }
void JxlThreadParallelRunnerDestroy(void* runner_opaque) { delete static_cast<Pool*>(runner_opaque); }
size_t JxlThreadParallelRunnerDefaultNumWorkerThreads(void) { return std::thread::hardware_concurrency(); }
JxlParallelRetCode JxlResizableParallelRunner(void* runner_opaque, void* jpegxl_opaque, JxlParallelRunInit init,
                                              JxlParallelRunFunction func, uint32_t start_range, uint32_t end_range) {
  return JxlThreadParallelRunner(runner_opaque, jpegxl_opaque, init, func, start_range, end_range);
}
void* JxlResizableParallelRunnerCreate(const JxlMemoryManager*) { return new (std::nothrow) Pool(0); }
void JxlResizableParallelRunnerSetThreads(void* runner_opaque, size_t num_threads) {
  static_cast<Pool*>(runner_opaque)->SetThreads(num_threads);
}
uint32_t JxlResizableParallelRunnerSuggestThreads(uint64_t xsize, uint64_t ysize) {
  // one thread per 2048x2048 DC group worth of host work (the host only decodes DC groups), at least 1
  uint64_t n = ((xsize + 2047) / 2048) * ((ysize + 2047) / 2048);
  uint64_t hw = std::thread::hardware_concurrency();
  if (n > hw && hw) n = hw;
  return uint32_t(n ? n : 1);
}
void JxlResizableParallelRunnerDestroy(void* runner_opaque) { delete static_cast<Pool*>(runner_opaque); }
}  // extern "C"
