// libjxl_amd — host side of the drop-in boundary: the JxlDecoder* C API subset (include/jxl/decode.h), the
// JxlThreadParallelRunner / JxlResizableParallelRunner (include/jxl/thread_parallel_runner.h) and the frame-level
// helpers of include/jxl_amd.h. Mirrors the behaviour of reference lib/jxl/decode.cc (event order, status values,
// sticky errors, caller-owned input/output) for whole-file input; pixels come from the HIP layer only — if no
// MI355X/HIP device is usable every decode fails with JXL_DEC_ERROR (there is deliberately no CPU fallback).
#include <jxl/decode.h>
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

int jxlamd_frame_parse(const uint8_t* data, size_t size, JxlParallelRunner runner, void* runner_opaque, JxlAmdFrame** out) {
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
    parser.ParseFrame(pos, ih, &f->plan, MakeParallelFor(runner, runner_opaque));
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

void jxlamd_frame_free(JxlAmdFrame* f) { delete f; }

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

void jxlamd_frame_out_size(const JxlAmdFrame* f, uint32_t* wh) {
  const jxh::FramePlan& P = f->plan;
  wh[0] = uint32_t(P.fh.upsampling == 1 ? P.dim.xsize : P.ih.xsize);
  wh[1] = uint32_t(P.fh.upsampling == 1 ? P.dim.ysize : P.ih.ysize);
}

// The N*N 5x5 upsampling kernels from the upper triangle of the symmetric default weight matrix
// (stage_upsampling.cc:59-84; weights image_metadata.cc:98-214).
#include "../host/upsampling_weights.inc"
static void UpsamplingKernels(uint32_t N, std::vector<float>* kernel) {
  const float* weights = N == 2 ? kUpsamplingWeights2 : (N == 4 ? kUpsamplingWeights4 : kUpsamplingWeights8);
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
  d.dc = P.dc.data();
  d.inv_sigma = P.inv_sigma.data();
  d.ytox = P.ytox.data();
  d.ytob = P.ytob.data();
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
  d.linear_output = P.ih.linear_tf;
  d.band_group_row_begin = group_row_begin;
  d.band_group_row_end = group_row_end;
  std::vector<float> ups_kernel;
  if (P.fh.upsampling != 1) {
    UpsamplingKernels(P.fh.upsampling, &ups_kernel);
    d.upsampling = P.fh.upsampling;
    d.out_xsize = uint32_t(P.ih.xsize);
    d.out_ysize = uint32_t(P.ih.ysize);
    d.upsampling_kernel = ups_kernel.data();
  }
  int r = jxlhip_frame_upload(ctx, &d);
  if (r == 0) r = jxlhip_sync(ctx);  // `pd` and the staging copies are locals
  if (r) g_last_error = "jxlhip_frame_upload failed (" + std::to_string(r) + ")";
  return r;
}

// ------------------------------------------------------------------------------------------------ JxlDecoder
}  // extern "C"

struct JxlDecoderStruct {
  JxlMemoryManager mm{};
  JxlParallelRunner runner = nullptr;
  void* runner_opaque = nullptr;
  int events = 0;
  const uint8_t* data = nullptr;
  size_t size = 0;
  bool input_closed = false;
  int stage = 0;  // 0 start, 1 after basic info, 2 after colour, 3 after frame header, 4 need buffer, 5 done image, 6 end
  bool error = false;
  jxh::ImageHeader ih;
  bool have_ih = false;
  size_t frame_pos = 0;
  JxlAmdFrame* frame = nullptr;
  JxlHipContext* ctx = nullptr;
  JxlPixelFormat fmt{};
  void* out_buf = nullptr;
  size_t out_size = 0;
  JxlImageOutCallback callback = nullptr;
  void* callback_opaque = nullptr;
  bool have_out = false;
};

namespace {
size_t RowStride(const JxlPixelFormat& f, size_t xsize) {
  size_t bytes = f.data_type == JXL_TYPE_UINT8 ? 1 : f.data_type == JXL_TYPE_UINT16 || f.data_type == JXL_TYPE_FLOAT16 ? 2 : 4;
  size_t stride = xsize * f.num_channels * bytes;
  if (f.align > 1) stride = (stride + f.align - 1) / f.align * f.align;
  return stride;
}
void ResetState(JxlDecoder* d) {
  if (d->frame) jxlamd_frame_free(d->frame);
  d->frame = nullptr;
  d->events = 0;
  d->data = nullptr;
  d->size = 0;
  d->input_closed = false;
  d->stage = 0;
  d->error = false;
  d->have_ih = false;
  d->have_out = false;
  d->out_buf = nullptr;
  d->callback = nullptr;
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

JxlDecoder* JxlDecoderCreate(const JxlMemoryManager* memory_manager) {
  if (memory_manager && (!memory_manager->alloc != !memory_manager->free)) return nullptr;
  JxlDecoder* d = new (std::nothrow) JxlDecoder;
  if (!d) return nullptr;
  if (memory_manager) d->mm = *memory_manager;
  return d;
}
void JxlDecoderReset(JxlDecoder* d) { ResetState(d); }
void JxlDecoderDestroy(JxlDecoder* d) {
  if (!d) return;
  ResetState(d);
  if (d->ctx) jxlhip_ctx_destroy(d->ctx);
  delete d;
}
void JxlDecoderRewind(JxlDecoder* d) {
  int ev = d->events;
  JxlParallelRunner r = d->runner;
  void* ro = d->runner_opaque;
  ResetState(d);
  d->events = ev;
  d->runner = r;
  d->runner_opaque = ro;
}
void JxlDecoderSkipFrames(JxlDecoder*, size_t) {}
JxlDecoderStatus JxlDecoderSkipCurrentFrame(JxlDecoder* d) { return d->stage >= 3 && d->stage < 5 ? (d->stage = 6, JXL_DEC_SUCCESS) : JXL_DEC_ERROR; }

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
JxlDecoderStatus JxlDecoderSetKeepOrientation(JxlDecoder* d, JXL_BOOL) { return d->stage == 0 ? JXL_DEC_SUCCESS : JXL_DEC_ERROR; }
JxlDecoderStatus JxlDecoderSetUnpremultiplyAlpha(JxlDecoder* d, JXL_BOOL) { return d->stage == 0 ? JXL_DEC_SUCCESS : JXL_DEC_ERROR; }
JxlDecoderStatus JxlDecoderSetRenderSpotcolors(JxlDecoder* d, JXL_BOOL) { return d->stage == 0 ? JXL_DEC_SUCCESS : JXL_DEC_ERROR; }
JxlDecoderStatus JxlDecoderSetCoalescing(JxlDecoder* d, JXL_BOOL) { return d->stage == 0 ? JXL_DEC_SUCCESS : JXL_DEC_ERROR; }

JxlDecoderStatus JxlDecoderSetInput(JxlDecoder* d, const uint8_t* data, size_t size) {
  if (d->data) return JXL_DEC_ERROR;
  d->data = data;
  d->size = size;
  return JXL_DEC_SUCCESS;
}
size_t JxlDecoderReleaseInput(JxlDecoder* d) {
  size_t remaining = 0;
  if (d->data && d->stage < 6) {
    size_t used = d->frame ? d->frame->plan.frame_end : 0;
    remaining = d->stage >= 5 ? d->size - std::min(d->size, used) : d->size;
  }
  d->data = nullptr;
  d->size = 0;
  return remaining;
}
void JxlDecoderCloseInput(JxlDecoder* d) { d->input_closed = true; }

static JxlDecoderStatus Fail(JxlDecoder* d, const std::string& why) {
  g_last_error = why;
  d->error = true;
  return JXL_DEC_ERROR;
}

JxlDecoderStatus JxlDecoderProcessInput(JxlDecoder* d) {
  if (d->error) return JXL_DEC_ERROR;
  if (!d->data) return d->stage == 6 ? JXL_DEC_SUCCESS : JXL_DEC_NEED_MORE_INPUT;
  if (d->stage == 0) {
    JxlSignature sig = JxlSignatureCheck(d->data, d->size);
    if (sig == JXL_SIG_INVALID) return Fail(d, "invalid signature");
    if (sig == JXL_SIG_NOT_ENOUGH_BYTES) return d->input_closed ? Fail(d, "truncated") : JXL_DEC_NEED_MORE_INPUT;
    try {
      jxh::FrameParser parser(d->data, d->size);
      d->frame_pos = parser.ParseImageHeader(&d->ih);
      d->have_ih = true;
    } catch (const std::exception& e) {
      std::string w = e.what();
      if (!d->input_closed && w.find("truncated") != std::string::npos) return JXL_DEC_NEED_MORE_INPUT;
      return Fail(d, w);
    }
    d->stage = 1;
    if (d->events & JXL_DEC_BASIC_INFO) return JXL_DEC_BASIC_INFO;
  }
  if (d->stage == 1) {
    d->stage = 2;
    if (d->events & JXL_DEC_COLOR_ENCODING) return JXL_DEC_COLOR_ENCODING;
  }
  if (d->stage == 2) {
    if (!(d->events & (JXL_DEC_FRAME | JXL_DEC_FULL_IMAGE))) {
      d->stage = 6;
      return JXL_DEC_SUCCESS;
    }
    int r = jxlamd_frame_parse(d->data, d->size, d->runner, d->runner_opaque, &d->frame);
    if (r) {
      std::string w = g_last_error;
      if (!d->input_closed && w.find("truncated") != std::string::npos) return JXL_DEC_NEED_MORE_INPUT;
      return Fail(d, w);
    }
    d->stage = 3;
    if (d->events & JXL_DEC_FRAME) return JXL_DEC_FRAME;
  }
  if (d->stage == 3) {
    if (!(d->events & JXL_DEC_FULL_IMAGE)) {
      d->stage = 6;
      return JXL_DEC_SUCCESS;
    }
    d->stage = 4;
  }
  if (d->stage == 4) {
    if (!d->have_out) return JXL_DEC_NEED_IMAGE_OUT_BUFFER;
    if (!d->ctx) {
      if (jxlhip_device_count() <= 0) return Fail(d, "no HIP device: libjxl_amd has no CPU decode path");
      const char* dev = getenv("JXLHIP_DEVICE");
      int r = jxlhip_ctx_create(dev ? atoi(dev) : 0, &d->ctx);
      if (r) return Fail(d, "jxlhip_ctx_create failed (" + std::to_string(r) + ")");
    }
    int r = jxlamd_frame_upload(d->frame, d->ctx);
    if (!r) r = jxlhip_run_all(d->ctx);
    std::vector<uint32_t> flags(d->frame->plan.dim.num_groups);
    if (!r) r = jxlhip_get_errors(d->ctx, flags.data(), flags.size());
    if (r) return Fail(d, "GPU decode failed (" + std::to_string(r) + ")");
    uint32_t out_wh[2];
    jxlamd_frame_out_size(d->frame, out_wh);
    const size_t xs = out_wh[0], ys = out_wh[1];
    const uint32_t nc = d->fmt.num_channels;
    std::vector<uint8_t> rgb;
    uint8_t* dst = static_cast<uint8_t*>(d->out_buf);
    const size_t stride = RowStride(d->fmt, xs);
    if (nc == 3 && !d->callback) {
      r = jxlhip_download_rgb8(d->ctx, dst, stride);
      if (r) return Fail(d, "download failed");
    } else {
      rgb.resize(xs * ys * 3);
      r = jxlhip_download_rgb8(d->ctx, rgb.data(), xs * 3);
      if (r) return Fail(d, "download failed");
      std::vector<uint8_t> row(xs * nc);
      for (size_t y = 0; y < ys; y++) {
        uint8_t* o = d->callback ? row.data() : dst + y * stride;
        const uint8_t* s = rgb.data() + y * xs * 3;
        for (size_t x = 0; x < xs; x++) {
          if (nc >= 3) {
            o[x * nc] = s[x * 3];
            o[x * nc + 1] = s[x * 3 + 1];
            o[x * nc + 2] = s[x * 3 + 2];
            if (nc == 4) o[x * nc + 3] = 255;
          } else {  // grey output of a colour image: not meaningful; keep the first channel
            o[x * nc] = s[x * 3 + 1];
            if (nc == 2) o[x * nc + 1] = 255;
          }
        }
        if (d->callback) d->callback(d->callback_opaque, 0, y, xs, row.data());
      }
    }
    d->stage = 5;
    return JXL_DEC_FULL_IMAGE;
  }
  d->stage = 6;
  return JXL_DEC_SUCCESS;
}

JxlDecoderStatus JxlDecoderGetBasicInfo(const JxlDecoder* d, JxlBasicInfo* info) {
  if (!d->have_ih) return JXL_DEC_NEED_MORE_INPUT;
  if (info) {
    memset(info, 0, sizeof(*info));
    static const uint8_t kContainer[4] = {0, 0, 0, 0xC};
    info->have_container = d->data && d->size >= 4 && !memcmp(d->data, kContainer, 4);
    info->xsize = d->ih.xsize;
    info->ysize = d->ih.ysize;
    info->bits_per_sample = d->ih.bits;
    info->exponent_bits_per_sample = d->ih.exp_bits;
    info->intensity_target = d->ih.intensity_target;
    info->uses_original_profile = !d->ih.xyb_encoded;
    info->orientation = JxlOrientation(d->ih.orientation);
    info->num_color_channels = d->ih.gray ? 1 : 3;
    info->num_extra_channels = uint32_t(d->ih.extra.size());
    info->intrinsic_xsize = d->ih.xsize;
    info->intrinsic_ysize = d->ih.ysize;
  }
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetExtraChannelInfo(const JxlDecoder*, size_t, JxlExtraChannelInfo*) { return JXL_DEC_ERROR; }
JxlDecoderStatus JxlDecoderGetExtraChannelName(const JxlDecoder*, size_t, char*, size_t) { return JXL_DEC_ERROR; }
JxlDecoderStatus JxlDecoderGetColorAsEncodedProfile(const JxlDecoder* d, JxlColorProfileTarget, JxlColorEncoding* ce) {
  if (!d->have_ih) return JXL_DEC_NEED_MORE_INPUT;
  if (ce) {
    memset(ce, 0, sizeof(*ce));
    ce->color_space = d->ih.gray ? JXL_COLOR_SPACE_GRAY : JXL_COLOR_SPACE_RGB;
    ce->white_point = JXL_WHITE_POINT_D65;
    ce->white_point_xy[0] = 0.3127;
    ce->white_point_xy[1] = 0.3290;
    ce->primaries = JXL_PRIMARIES_SRGB;
    ce->primaries_red_xy[0] = 0.639998686; ce->primaries_red_xy[1] = 0.330010138;
    ce->primaries_green_xy[0] = 0.300003784; ce->primaries_green_xy[1] = 0.600003357;
    ce->primaries_blue_xy[0] = 0.150002046; ce->primaries_blue_xy[1] = 0.059997204;
    ce->transfer_function = d->ih.linear_tf ? JXL_TRANSFER_FUNCTION_LINEAR : JXL_TRANSFER_FUNCTION_SRGB;
    ce->rendering_intent = JXL_RENDERING_INTENT_RELATIVE;
  }
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetICCProfileSize(const JxlDecoder*, JxlColorProfileTarget, size_t* size) {
  if (size) *size = 0;
  return JXL_DEC_ERROR;  // no ICC synthesis: callers fall back to the encoded profile
}
JxlDecoderStatus JxlDecoderGetColorAsICCProfile(const JxlDecoder*, JxlColorProfileTarget, uint8_t*, size_t) { return JXL_DEC_ERROR; }
JxlDecoderStatus JxlDecoderSetPreferredColorProfile(JxlDecoder*, const JxlColorEncoding*) { return JXL_DEC_SUCCESS; }
JxlDecoderStatus JxlDecoderSetDesiredIntensityTarget(JxlDecoder*, float) { return JXL_DEC_SUCCESS; }
JxlDecoderStatus JxlDecoderSetOutputColorProfile(JxlDecoder* d, const JxlColorEncoding* ce, const uint8_t*, size_t) {
  // only (linear or non-linear) sRGB output is implemented; anything else is refused so that the caller notices
  if (!ce) return JXL_DEC_ERROR;
  if (ce->primaries != JXL_PRIMARIES_SRGB || ce->white_point != JXL_WHITE_POINT_D65) return JXL_DEC_ERROR;
  if (ce->transfer_function == JXL_TRANSFER_FUNCTION_LINEAR) d->ih.linear_tf = true;
  else if (ce->transfer_function == JXL_TRANSFER_FUNCTION_SRGB) d->ih.linear_tf = false;
  else return JXL_DEC_ERROR;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetFrameHeader(const JxlDecoder* d, JxlFrameHeader* h) {
  if (!d->frame) return JXL_DEC_ERROR;
  if (h) {
    memset(h, 0, sizeof(*h));
    h->is_last = JXL_TRUE;
    h->layer_info.xsize = d->ih.xsize;
    h->layer_info.ysize = d->ih.ysize;
  }
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetFrameName(const JxlDecoder*, char* name, size_t size) {
  if (name && size) name[0] = 0;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderPreviewOutBufferSize(const JxlDecoder*, const JxlPixelFormat*, size_t*) { return JXL_DEC_ERROR; }
JxlDecoderStatus JxlDecoderSetPreviewOutBuffer(JxlDecoder*, const JxlPixelFormat*, void*, size_t) { return JXL_DEC_ERROR; }
JxlDecoderStatus JxlDecoderImageOutBufferSize(const JxlDecoder* d, const JxlPixelFormat* f, size_t* size) {
  if (!d->have_ih || !f || !size) return JXL_DEC_ERROR;
  if (f->num_channels < 1 || f->num_channels > 4) return JXL_DEC_ERROR;
  if (f->data_type != JXL_TYPE_UINT8) return JXL_DEC_ERROR;  // only 8-bit output is implemented on the GPU path
  size_t stride = RowStride(*f, d->ih.xsize);
  *size = stride * (d->ih.ysize - 1) + size_t(d->ih.xsize) * f->num_channels;
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
  d->have_out = true;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetImageOutCallback(JxlDecoder* d, const JxlPixelFormat* f, JxlImageOutCallback cb, void* opaque) {
  size_t need = 0;
  if (!cb || JxlDecoderImageOutBufferSize(d, f, &need) != JXL_DEC_SUCCESS) return JXL_DEC_ERROR;
  d->fmt = *f;
  d->callback = cb;
  d->callback_opaque = opaque;
  d->out_buf = nullptr;
  d->have_out = true;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderExtraChannelBufferSize(const JxlDecoder*, const JxlPixelFormat*, size_t*, uint32_t) { return JXL_DEC_ERROR; }
JxlDecoderStatus JxlDecoderSetExtraChannelBuffer(JxlDecoder*, const JxlPixelFormat*, void*, size_t, uint32_t) { return JXL_DEC_ERROR; }
JxlDecoderStatus JxlDecoderSetDecompressBoxes(JxlDecoder*, JXL_BOOL) { return JXL_DEC_SUCCESS; }
JxlDecoderStatus JxlDecoderSetProgressiveDetail(JxlDecoder*, JxlProgressiveDetail) { return JXL_DEC_SUCCESS; }
size_t JxlDecoderGetIntendedDownsamplingRatio(JxlDecoder*) { return 1; }
JxlDecoderStatus JxlDecoderFlushImage(JxlDecoder*) { return JXL_DEC_ERROR; }
JxlDecoderStatus JxlDecoderSetImageOutBitDepth(JxlDecoder*, const JxlBitDepth* bd) {
  if (!bd) return JXL_DEC_ERROR;
  if (bd->type == JXL_BIT_DEPTH_CUSTOM && bd->bits_per_sample != 8) return JXL_DEC_ERROR;
  return JXL_DEC_SUCCESS;
}

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
