// libjxl_amd — implementation of the thin extern "C" HIP layer declared in include/jxl_amd_hip.h.
// One context = one HIP stream + the device buffers of one frame. Launch functions never allocate or synchronise
// (jxlhip_frame_upload does the allocation), so a caller may capture jxlhip_run_* into a hipGraph.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <unistd.h>
#include <mutex>
#include <new>
#include <chrono>
#include <string>
#include <vector>

#include "../../../include/jxl_amd_hip.h"
#include "jxl_hip_kernels.h"
#include "jxl_hip_entropy_lanes.h"
#include "jxl_hip_filter_fused.h"
#include "jxl_hip_modular.h"
#include "jxl_hip_enc.h"
#include "jxl_hip_canvas.h"
#include "jxl_hip_dc.h"

namespace {
#include "../host/afv_basis.inc"
#include "../host/dither.inc"

#define HIP_TRY(expr)                          \
  do {                                         \
    hipError_t e_ = (expr);                    \
    if (e_ != hipSuccess) return -int(e_);     \
  } while (0)

// "blocking_sync" option: waits of this library poll + sleep (hipStreamSynchronize / hipEventSynchronize spin).
std::atomic<bool> g_yield_waits{false};
inline hipError_t WaitStream(hipStream_t st) {
  if (!g_yield_waits.load(std::memory_order_relaxed)) return hipStreamSynchronize(st);
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q != hipErrorNotReady) return q;
    usleep(200);
  }
}
inline hipError_t WaitEvent(hipEvent_t ev) {
  if (!g_yield_waits.load(std::memory_order_relaxed)) return hipEventSynchronize(ev);
  for (;;) {
    const hipError_t q = hipEventQuery(ev);
    if (q != hipErrorNotReady) return q;
    usleep(100);
  }
}

// JXLHIP_GUARD=1 (debug aid, read at every allocation): every device buffer gets a guard band either side, filled with a
// pattern that jxlhip_check_guards() verifies: stray WRITES of a kernel next to its buffers show up in a test instead of
// silently landing in a neighbour. Stray READS: the pattern is JXLHIP_GUARD_BYTE (default 0xA5), and a fresh allocation is
// filled with it as a whole; a decode whose result changes with the pattern has read a guard band or memory nothing wrote
// (tests/test_gpu_parity.py::test_no_result_depends_on_bytes_outside_the_buffers runs the kernels under three patterns).
constexpr size_t kGuardBytes = 4096;
inline bool GuardOn() {
  const char* e = getenv("JXLHIP_GUARD");
  return e && *e && atoi(e) != 0;
}
inline int GuardByte() {
  const char* e = getenv("JXLHIP_GUARD_BYTE");
  return e && *e ? (int(strtol(e, nullptr, 0)) & 0xFF) : 0xA5;
}

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  bool view = false;  // p points into another allocation (a frame's table blob): nothing to free
  void* base = nullptr;  // the allocation when it has guard bands (p = base + kGuardBytes), else NULL
  int guard_byte = 0xA5;  // the pattern its bands were filled with
  int Ensure(size_t n) {
    if (view) {
      p = nullptr;
      cap = 0;
      view = false;
    }
    if (n <= cap && p) return 0;
    if (p) {
      hipError_t e = hipFree(base ? base : p);
      p = base = nullptr;
      cap = 0;
      if (e != hipSuccess) return -int(e);
    }
    size_t want = n < 256 ? 256 : n;
    if (GuardOn()) {
      want = (want + 255) & ~size_t(255);
      hipError_t e = hipMalloc(&base, want + 2 * kGuardBytes);
      guard_byte = GuardByte();
      if (e == hipSuccess) e = hipMemset(base, guard_byte, want + 2 * kGuardBytes);  // (bands and body: see GuardByte)
      // (hipMemset returns before the fill has run, and the contexts' non-blocking streams do not wait for the null
      // stream: without this the fill can land on top of a table copy that was enqueued after it)
      if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
      if (e != hipSuccess) {
        if (base) (void)hipFree(base);
        base = nullptr;
        return -int(e);
      }
      p = static_cast<uint8_t*>(base) + kGuardBytes;
      cap = want;
      return 0;
    }
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      return -int(e);
    }
    cap = want;
    return 0;
  }
  void Free() {
    if (p && !view) (void)hipFree(base ? base : p);
    p = base = nullptr;
    cap = 0;
    view = false;
  }
  // 0 = both guard bands intact (or none), 1 = the band before, 2 = the band after, 3 = both were written to
  int GuardsTouched() const {
    if (!base || view) return 0;
    std::vector<uint8_t> h(2 * kGuardBytes);
    if (hipMemcpy(h.data(), base, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    if (hipMemcpy(h.data() + kGuardBytes, static_cast<uint8_t*>(base) + kGuardBytes + cap, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    int r = 0;
    for (size_t i = 0; i < kGuardBytes; i++) {
      if (h[i] != guard_byte) r |= 1;
      if (h[kGuardBytes + i] != guard_byte) r |= 2;
    }
    return r;
  }
  template <typename T>
  T* as() const { return static_cast<T*>(p); }
};

struct PassBufs {
  Buf ctx_map, alias, cfg, orders, ptable, poffset, alias_packed;
};

// One launch of the Modular transform / output kernels over the frames of a set (ModularBuildOps).
struct ModLaunch {
  uint32_t kind;  // 0 RCT, 1 palette, 2 unsqueeze, 3 output
  size_t offset;  // of the first parameter block in the blob
  uint32_t count, gx, gy;
};

// Pinned host staging block of a context: every table of a frame is assembled in it and leaves through asynchronous
// copies on the context's stream, so that an upload neither blocks on pageable-memory copies nor synchronises per table.
// The block is reused by the next upload once `done` (recorded after the upload's last copy) has passed.
struct Stage {
  uint8_t* p = nullptr;
  size_t cap = 0, used = 0;
  hipEvent_t done = nullptr;
  bool recorded = false;
};

constexpr int kEntropyWPG = 4;  // waves (= AC sections) per workgroup of the scalar-form entropy kernel
constexpr int kLanesWPG = 4;    // waves per workgroup of the lane-parallel entropy kernel (one frame per workgroup)
constexpr size_t kLdsBudget = 150 * 1024;

}  // namespace

struct JxlHipContext {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {};
  bool ev_valid[3] = {false, false, false};
  Buf basis;
  Stage stage;
  // jxlhip_frame_upload: every table of the frame lives in one device allocation laid out like the staging block, filled
  // by ONE host-to-device copy at the end of the upload (HIP serialises API calls of all host threads: twenty small
  // copies per frame capped a multi-threaded uploader at ~0.6 ms per frame); the tables' Bufs are views into it.
  Buf frame_blob;
  bool blob_mode = false;
  bool have_frame = false;
  // geometry
  uint32_t xs = 0, ys = 0, xb = 0, yb = 0, xg = 0, ng = 0, np = 0, xp = 0, yp = 0, coef_bits = 16;
  int gab = 0, epf_iters = 0;
  float epf_pass0 = 0.9f, epf_pass2 = 6.5f, epf_border = 2.0f / 3;
  // buffers
  Buf sections, sec_word, sec_size, blocks, gbb, bctx_lut, dequant, dc, inv_sigma, ytox, ytob, passes_dev, coeffs, errors;
  Buf dc_q, dc_ep;    // coded DC integers and the DC groups' extra-precision bytes (DequantDC on the device)
  Buf dc_raw, sharp;  // inputs of the DC-path kernels (jxl_hip_dc.h) when the upload smooths the DC image / computes 1 / sigma
  Buf plane[3], rgb, tlist, scratch, sec_end, lz_window;
  bool generic_codec = false;  // a pass is prefix-coded or uses LZ77 (k_entropy_generic decodes the frame unless lane_prefix)
  bool lane_prefix = false;    // every pass is prefix-coded without LZ77: the lane kernel's prefix form decodes the frame
  // ---- Modular frame (jxlhip_modular_upload): channel pool, tables, stream descriptors, inverse-transform operations
  struct Modular {
    bool have = false;
    Buf pool, sections, blob, streams, rects, status, end_bits, scratch, windows, batch_streams;
    std::vector<size_t> buf_off;            // per channel buffer: first int32 of the pool
    std::vector<uint32_t> buf_w, buf_h;
    std::vector<JxlHipModOp> ops;
    std::vector<jxlhip::ModStream> streams_host;  // device pointers filled in
    std::vector<uint32_t> stream_samples;
    uint32_t out_buffer[4] = {0, 0, 0, 0};
    uint32_t num_color = 3, has_alpha = 0, bits = 8, alpha_bits = 8, xs = 0, ys = 0, nstreams = 0;
    bool xyb = false;  // XYB Modular frame: integer Y, X, B - Y channels, the colour stage's parameters in `xyb_color`
    float xyb_factor[3] = {0, 0, 0};
    jxlhip::FilterParams xyb_color;
    std::vector<const JxlHipContext*> batch_ctxs;
    std::vector<uint64_t> batch_gens;
    uint32_t batch_n = 0, batch_tree_cap = 0, batch_table_cap = 0;
    bool batch_wp = true, batch_refs = true;
    std::vector<uint32_t> code_table_words;  // per entropy code: words of its symbol tables (ModCode::table_words)
    Buf batch_ops;                            // parameter blocks of the set's transform / output launches
    std::vector<ModLaunch> batch_launches;
  } mod;
  size_t plane_bytes = 0;  // bytes of plane[0] the current frame needs
  Buf ep_dev;                         // device copy of `ep` (the entropy kernel reads it through the scalar cache)
  Buf batch_wave_ls;
  size_t batch_off_units = 0, batch_off_queue = 0, batch_off_wave_lanes = 0, batch_units = 0;  // layout of batch_lanes
  Buf batch_params, batch_map, batch_lanes;  // jxlhip_run_entropy_batch: parameter blocks, workgroup map, lane map
  uint32_t batch_wait_shift = 2, batch_lanes_per_wave = 64, batch_wpg = 4;
  bool batch_galias = false;  // the lane kernel reads its alias tables from global memory (PrepareBatch)
  bool batch_a6 = false;      // ... or keeps them in LDS in the six-byte form (jxl_hip_entropy_lanes.h LanesLdsLayout)
  bool batch_prefix = false;  // ... decodes prefix codes
  int batch_kernel = -1;
  std::vector<uint32_t> sec_size_host, sec_sel_host, pass_clusters, pass_log_alpha;
  // Coefficient layout of this frame (see TransformParams::scan_order); scan order is produced by k_entropy_lanes.
  // band decode: groups this context decodes (band + one group row either side), pixel rows it produces
  std::vector<uint32_t> group_list;
  std::vector<uint32_t> absent_groups;  // JxlHipFrameDesc::group_absent: drawn from the DC image alone
  std::vector<uint32_t> absent_blocks;  // per absent group: first block, block count
  std::vector<uint32_t> absent_sections;  // (partial frames) passes of present groups that have not arrived: pass, first block, block count
  uint32_t band_y0 = 0, band_y1 = 0;
  uint32_t ext_y0 = 0, ext_y1 = 0;  // the pixel rows the transforms produce (the band and the group rows decoded around it)
  // upsampled frames: factor (1 = none), image size, kernels
  uint32_t ups = 1, oxs = 0, oys = 0;
  Buf ups_kernel;
  bool scan_order = false;
  // lanes: the frame's entropy stage runs on k_entropy_lanes. Single-pass frames hand the scan-order layout to the
  // transforms (scan_order); progressive frames (lane_multi) decode every pass into its own scan-order buffer and
  // k_merge_passes sums them into the natural layout.
  bool lanes = false, lane_multi = false;
  uint32_t nblocks = 0;
  bool keep_filtered = false;  // jxlhip_set_option("keep_filtered"): also write the filtered XYB planes (tests)
  bool band_halo = false;      // jxlhip_set_option("band_halo"): a band decodes ONLY its own group rows; the rows of the
                               // neighbouring bands that its filters read arrive through jxlhip_halo_unpack
  // output pixel format (jxlhip_set_output_format; JxlDataType numbering): RGB8 by default
  uint32_t out_type = 2, out_nc = 3, out_bits = 8, out_swap = 0;
  // forward (encoder) path, jxlhip_enc_forward: device buffers and the kernel time of the last call
  Buf enc_rgb, enc_planes[3], enc_act, enc_acs, enc_qf, enc_off, enc_dc, enc_coef, enc_lut, enc_dq, enc_ytox, enc_ytob;
  hipEvent_t enc_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // whole sequence; the transform kernel of its last pass
  bool enc_timed = false;
  jxlhip::EncTok enc_tok;      // jxlhip_enc_token_counts -> jxlhip_enc_tokens
  bool enc_tok_ready = false;
  std::vector<uint32_t> enc_tok_totals;
  Buf enc_tok_orders, enc_tok_blk, enc_tok_info, enc_tok_off, enc_tok_nzmap, enc_tok_small, enc_tok_out, enc_tok_base;
  jxlhip::EncFwd enc_last;     // the parameters of the last jxlhip_enc_forward (its input stays resident): jxlhip_enc_forward_rerun
  bool enc_last_gaborish = false;
  uint32_t out_orient = 0;  // jxlhip_set_output_orientation: PixelOut::orient bits (0 = the image as coded)
  bool out_unpremul = false;  // jxlhip_set_output_unpremultiply (PixelOut::orient bit 3 for outputs that carry alpha)
  // splines (JxlHipSplines): the draw cache on the device; for a Modular frame also the float planes they are drawn over
  Buf spl_seg, spl_row_start, spl_row_seg, spl_planes;
  uint32_t spl_segments = 0;
  // patches (JxlHipPatches): records and row lists on the device, the reference planes they read
  Buf pat_rec, pat_row_start, pat_row_list;
  uint32_t pat_positions = 0;
  const float* pat_src[4] = {nullptr, nullptr, nullptr, nullptr};
  const float* pat_src_alpha[4] = {nullptr, nullptr, nullptr, nullptr};
  bool pat_uses_alpha = false, pat_premultiplied = false;
  // the alpha plane the patch stage blended into (a copy: the plane that was set stays as it is, the stages can run again);
  // valid from the filter stage of such a frame on
  Buf alpha_patched;
  bool alpha_patched_valid = false;
  uint32_t pat_src_w[4] = {0, 0, 0, 0}, pat_src_h[4] = {0, 0, 0, 0};
  bool keep_xyb = false;  // option "keep_xyb_planes": a Modular frame also leaves its colour as float planes (spl_planes)
  // noise synthesis (JxlHipFrameDesc::has_noise): raw random planes [3][ys][xs], LUT, seeds, base colour correlation
  Buf noise;
  bool has_noise = false;
  float noise_lut[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t noise_seed[2] = {0, 0};
  float noise_ytox = 0.0f, noise_ytob = 1.0f;
  Buf alpha;                // f32 plane of the image size (jxlhip_set_alpha), used by 2- and 4-channel output
  bool have_alpha = false;
  bool color_out = false;   // the pixels come from k_color_out / k_upsample_color's generic writer (set at upload)
  Buf kend, block_recs, dequant_scan;
  Buf trecs;     // the transform work lists as 16-byte varblock records (TransformParams::trecs)
  Buf ec_stage;  // jxlhip_upsample_plane: a coded extra channel, its kernels and (unless it becomes the alpha plane) the result
  std::vector<JxlHipVarBlock> blocks_host;  // for jxlhip_download("coeffs") of a scan-order frame
  std::vector<uint32_t> gbb_host;
  std::vector<uint16_t> orders_host;
  uint32_t order_offset_host[39] = {};
  std::vector<const JxlHipContext*> batch_ctxs;
  std::vector<uint64_t> batch_gens;
  uint32_t batch_wgs = 0;
  size_t batch_lds = 0;
  hipEvent_t batch_done = nullptr;
  // Set by jxlhip_run_entropy_batch on the contexts whose coefficients a launch on ANOTHER context's stream produces;
  // applied (hipStreamWaitEvent on this context's stream) by the next call that touches those results. Enqueuing the
  // wait only then keeps barrier packets of a long entropy launch out of the hardware queues other streams share.
  hipEvent_t pending_wait = nullptr;
  // jxlhip_share_planes: this context's inverse-transform output lives in `plane_lender`'s plane buffer (NULL: its own).
  // planes_event (on the context that owns the buffer): end of the last filter launch that read the buffer, recorded on
  // planes_stream; a transform launch from another stream waits for it before overwriting the planes.
  // Option "filter_async": batched filter + colour launches from this context go to its SECOND stream, ordered after
  // what the first stream holds at that moment; the first stream is then free for the next entropy launch (which touches
  // neither the planes nor the pixels) while the filter still runs. filter_wait (on every context of such a launch):
  // the launch's end event; waited for by whatever touches the planes or the pixels next (transform, download, sync,
  // upload), never by an entropy launch.
  bool filter_async = false;
  bool entropy_gate = false;  // option "entropy_gate" (see EntropyGate)
  hipEvent_t halo_event = nullptr;  // jxlhip_halo_*_batch: orders the halo copies against the transport's stream
  Buf ups_planes;                   // upsampled X, Y, B planes of a frame with noise ([3][oys][oxs padded to 8])
  bool pooled_once = false;         // the object has been through the context pool (RecycleContext frees it for good)
  bool owns_stream = false;   // `stream` is this context's own (it heads batched launches), not one of the shared pool
  hipStream_t stream2 = nullptr;
  hipStream_t fstream = nullptr;  // stream of the filter launch in progress (set by BeginDownstreamBatch)
  hipEvent_t fork_event = nullptr, filter_done = nullptr, filter_wait = nullptr;
  hipEvent_t transform_done = nullptr;  // filter_async: end of the set's last batched transform launch on `stream`
  bool transform_done_valid = false;
  JxlHipContext* plane_lender = nullptr;
  hipEvent_t planes_event = nullptr;
  hipStream_t planes_stream = nullptr;
  // jxlhip_run_transform_batch / jxlhip_run_filter_color_batch: description of the frame set launched from this context
  struct FilterGroup {
    int key;  // gaborish * 4 + epf iterations
    uint32_t first, count, tiles_x, tiles_y;
    bool u8srgb;  // every frame of the group writes 8-bit sRGB only (k_filter_rows2<true>)
  };
  Buf tb_params, tb_desc, fb_params;
  std::vector<const JxlHipContext*> db_ctxs;
  std::vector<uint64_t> db_gens;
  uint32_t desc_begin[27] = {}, desc_count[27] = {};  // (entry 21: the 8x8 DCT varblocks of chroma-subsampled frames)
  uint32_t cs = 0;  // chroma-subsampled frame: hshift of channel c in bit 2 * c, vshift in bit 2 * c + 1 (0 = 4:4:4)
  std::vector<FilterGroup> fgroups;
  hipEvent_t down_done = nullptr;
  uint64_t generation = 0;            // bumped by every jxlhip_frame_upload
  std::vector<PassBufs> pass_bufs;
  uint32_t list_begin[27] = {}, list_count[27] = {};
  jxlhip::EntropyParams ep;
  jxlhip::TransformParams tp;
  jxlhip::FilterParams fp;
  size_t lds_entropy = 0;
  bool alias_lds = false;
  int final_plane = 0;  // which plane set holds the filtered XYB after jxlhip_run_filter_color
};

// Which entropy kernel: 2 (default) lane-parallel k_entropy_lanes, 1 wave-per-section scalar k_entropy_uni,
// 0 the first k_entropy_ans. Environment override JXLHIP_ENTROPY is for A/B measurements only.
static int EntropyKernelChoice() {
  static const int v = [] {
    const char* e = getenv("JXLHIP_ENTROPY");
    return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 2;
  }();
  return v;
}
static JxlHipContext* PlaneHolder(JxlHipContext* c) { return c->plane_lender ? c->plane_lender : c; }
static const JxlHipContext* PlaneHolder(const JxlHipContext* c) { return c->plane_lender ? c->plane_lender : c; }

static size_t OutSampleBytes(const JxlHipContext* c) { return c->out_type == 2 ? 1 : (c->out_type == 0 ? 4 : 2); }
static size_t OutPixelBytes(const JxlHipContext* c) { return OutSampleBytes(c) * c->out_nc; }
static bool OutIsRgb8(const JxlHipContext* c) { return c->out_type == 2 && c->out_nc == 3 && c->out_bits == 8; }
static bool OutIsRgbF32(const JxlHipContext* c) { return c->out_type == 0 && c->out_nc == 3 && !c->out_swap; }

// Every index PrefixLookup (jxl_hip_kernels.h) can form from a cluster's two-level tables stays inside `table`, and
// every code length it can return is at most 15 bits.
static bool ValidPrefixTables(uint32_t offset_word, const uint32_t* table, size_t table_size) {
  const size_t first = offset_word & 0xFFFFFFu, root_bits = offset_word >> 24;
  if (root_bits > 15 || first + (size_t(1) << root_bits) > table_size) return false;
  for (size_t r = 0; r < (size_t(1) << root_bits); r++) {
    const uint32_t e = table[first + r];
    if (!(e & 0x80u)) {
      if ((e & 0xFFu) > 15) return false;
      continue;
    }
    const size_t sub_bits = e & 0x7Fu, sub = first + (e >> 8);
    if (sub_bits == 0 || root_bits + sub_bits > 15 || sub + (size_t(1) << sub_bits) > table_size) return false;
    for (size_t j = 0; j < (size_t(1) << sub_bits); j++)
      if ((table[sub + j] & 0x80u) || (table[sub + j] & 0xFFu) > 15) return false;
  }
  return true;
}

static int EnvInt(const char* name, int def) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : def;
}

static size_t LanesLdsFor(const JxlHipContext* c, uint32_t lanes = 64) {
  return jxlhip::LanesLdsLayout(1, c->ep.nctx, c->pass_clusters[0], c->pass_log_alpha[0], kLanesWPG, lanes).total;
}

extern "C" {

const char* jxlhip_version(void) { return "libjxl_amd 0.1 (gfx950)"; }

int jxlhip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// Streams. A context does NOT get a HIP stream of its own: with more than ~16 streams alive in a process (used or not) the
// runtime time-slices its hardware queues, and a long kernel is then preempted and resumed several times over (rocprofv3
// counts each resumption as a wave: SQ_WAVES 251 for a 32-wave launch); scripts/r03_streams_probe2.py: the same 8-frame
// entropy launch takes 79 ms with up to 16 streams in the process, 114 ms with 32, 125 ms with 64. Contexts share a small
// pool per device (JXLHIP_STREAMS, default 4); a context that heads batched launches over several contexts gets a
// dedicated stream the first time it does (EnsureOwnStream), so that the batched stages of different frame sets overlap.
static hipStream_t PoolStream(int device) {
  static std::mutex mu;
  static std::vector<hipStream_t> pool[64];
  static size_t next[64] = {};
  std::lock_guard<std::mutex> lock(mu);
  if (device < 0 || device >= 64) return nullptr;
  if (pool[device].empty()) {
    int n = 4;
    if (const char* e = getenv("JXLHIP_STREAMS")) n = atoi(e);
    n = n < 1 ? 1 : (n > 16 ? 16 : n);
    for (int i = 0; i < n; i++) {
      hipStream_t st = nullptr;
      if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
      pool[device].push_back(st);
    }
    if (pool[device].empty()) return nullptr;
  }
  return pool[device][next[device]++ % pool[device].size()];
}

// The head context of a batched launch runs the batch on a stream of its own (created on first use).
static int EnsureOwnStream(JxlHipContext* c);

// Contexts that callers have destroyed are kept, a few per process, with their device buffers, pinned staging block and
// events: a caller that creates a decoder per image (JxlDecoderCreate .. JxlDecoderDestroy, what djxl does) otherwise pays
// for the basis tables, a dozen hipMalloc of up to 100 MB and as many hipFree (each a device synchronisation) on every
// image: 5 - 9 ms of a 27 ms 4K decode. A recycled context is a NEW JxlHipContext object (every option and per-frame field
// at its default) that takes over the old one's allocations. Off with JXLHIP_CTX_POOL=0 and in guard mode (whose tests
// want fresh, pattern-filled allocations).
static std::mutex g_ctx_pool_mu;
static std::vector<JxlHipContext*> g_ctx_pool;
constexpr size_t kCtxPoolMax = 4, kCtxPoolBytes = size_t(1) << 30;
static std::vector<Buf*> AllBufs(JxlHipContext* c);
static JxlHipContext* RecycleContext(int device) {
  if (GuardOn() || !EnvInt("JXLHIP_CTX_POOL", 1)) return nullptr;
  JxlHipContext* o = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_ctx_pool_mu);
    for (size_t i = 0; i < g_ctx_pool.size(); i++)
      if (g_ctx_pool[i]->device == device) {
        o = g_ctx_pool[i];
        g_ctx_pool.erase(g_ctx_pool.begin() + long(i));
        break;
      }
  }
  if (!o) return nullptr;
  JxlHipContext* c = new (std::nothrow) JxlHipContext;
  if (!c) {
    jxlhip_ctx_destroy(o);
    return nullptr;
  }
  c->device = device;
  c->stream = PoolStream(device);
  std::vector<Buf*> to = AllBufs(c), from = AllBufs(o);  // (the new object has no per-pass buffers yet: the common prefix)
  for (size_t i = 0; i < to.size(); i++) {
    *to[i] = *from[i];
    if (to[i]->view) *to[i] = Buf();  // (a view into the old frame's table blob means nothing here)
    *from[i] = Buf();
  }
  c->pass_bufs = std::move(o->pass_bufs);
  o->pass_bufs.clear();
  for (size_t i = 0; i < sizeof(c->ev) / sizeof(c->ev[0]); i++) std::swap(c->ev[i], o->ev[i]);
  for (size_t i = 0; i < sizeof(c->enc_ev) / sizeof(c->enc_ev[0]); i++) std::swap(c->enc_ev[i], o->enc_ev[i]);
  std::swap(c->stage, o->stage);
  jxlhip_ctx_destroy(o);  // (what is left of it: lazily created events, a stream of its own)
  return c;
}

int jxlhip_ctx_create(int device, JxlHipContext** out) {
  if (!out) return JXLHIP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  HIP_TRY(hipSetDevice(device));
  if (JxlHipContext* recycled = RecycleContext(device)) {
    *out = recycled;
    return 0;
  }
  JxlHipContext* c = new (std::nothrow) JxlHipContext;
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  c->device = device;
  hipError_t e = hipSuccess;
  c->stream = PoolStream(device);
  if (!c->stream) {
    delete c;
    return JXLHIP_ERR_INVALID_ARGUMENT;
  }
  for (auto& ev : c->ev) {
    e = hipEventCreate(&ev);
    if (e != hipSuccess) {
      delete c;
      return -int(e);
    }
  }
  // static tables
  constexpr size_t kBasisFloats = 87381;
  std::vector<float> bt(2 * kBasisFloats + 4);  // [k][n] matrices, then (16-byte aligned) the transposed [n][k] ones
  for (int l = 0; l <= 8; l++) {
    const int N = 1 << l;
    float* dst = bt.data() + (size_t(N) * N - 1) / 3;
    float* dst_t = dst + kBasisFloats + 3;
    for (int k = 0; k < N; k++)
      for (int n = 0; n < N; n++) {
        const float v = float((k ? std::sqrt(2.0) : 1.0) * std::cos((n + 0.5) * k * M_PI / N));
        dst[size_t(k) * N + n] = v;
        dst_t[size_t(n) * N + k] = v;
      }
  }
  int r = c->basis.Ensure(bt.size() * sizeof(float));
  if (r) {
    jxlhip_ctx_destroy(c);
    return r;
  }
  float resample[63];
  for (int n = 1; n <= 32; n *= 2)
    for (int i = 0; i < n; i++) {
      const double N = 8.0 * n;
      resample[n - 1 + i] = float(1.0 / (std::cos(i / (2 * N) * M_PI) * std::cos(i / N * M_PI) * std::cos(i / (N / 2) * M_PI)));
    }
  e = hipMemcpy(c->basis.p, bt.data(), bt.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(jxlhip::c_afv_basis), kAfvBasis, sizeof(kAfvBasis));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(jxlhip::c_dither), kDither32, sizeof(kDither32));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(jxlhip::c_resample), resample, sizeof(resample));
  if (e != hipSuccess) {
    jxlhip_ctx_destroy(c);
    return -int(e);
  }
  *out = c;
  return 0;
}

static std::vector<Buf*> AllBufs(JxlHipContext* c) {
  std::vector<Buf*> all = {&c->basis, &c->sections, &c->sec_word, &c->sec_size, &c->blocks, &c->gbb, &c->bctx_lut, &c->dequant, &c->dc, &c->dc_raw, &c->dc_q, &c->dc_ep, &c->sharp,
                &c->inv_sigma, &c->ytox, &c->ytob, &c->passes_dev, &c->coeffs, &c->errors, &c->plane[0], &c->plane[1],
                &c->plane[2], &c->rgb, &c->tlist, &c->scratch, &c->ep_dev, &c->batch_params, &c->batch_map, &c->batch_lanes, &c->batch_wave_ls, &c->ups_kernel, &c->kend, &c->block_recs, &c->dequant_scan, &c->ec_stage, &c->alpha_patched, &c->trecs, &c->enc_tok_orders, &c->enc_tok_blk, &c->enc_tok_info,
                &c->enc_tok_off, &c->enc_tok_nzmap, &c->enc_tok_small, &c->enc_tok_out, &c->enc_tok_base, &c->tb_params, &c->tb_desc, &c->fb_params, &c->alpha, &c->sec_end, &c->lz_window, &c->mod.pool, &c->mod.sections, &c->mod.blob, &c->mod.streams,
                &c->mod.rects, &c->mod.status, &c->mod.end_bits, &c->mod.scratch, &c->mod.windows, &c->mod.batch_streams, &c->mod.batch_ops, &c->frame_blob, &c->noise, &c->spl_seg, &c->spl_row_start, &c->spl_row_seg, &c->spl_planes, &c->pat_rec, &c->pat_row_start, &c->pat_row_list,
                &c->enc_rgb, &c->enc_planes[0], &c->enc_planes[1], &c->enc_planes[2], &c->enc_act, &c->enc_acs, &c->enc_qf, &c->enc_off, &c->enc_dc, &c->enc_coef, &c->enc_lut, &c->enc_dq, &c->enc_ytox, &c->enc_ytob, &c->ups_planes};
  for (auto& pb : c->pass_bufs)
    for (Buf* b : {&pb.ctx_map, &pb.alias, &pb.cfg, &pb.orders, &pb.ptable, &pb.poffset, &pb.alias_packed}) all.push_back(b);
  return all;
}

void jxlhip_ctx_destroy(JxlHipContext* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  if (c->basis.p && !c->pooled_once && !GuardOn() && EnvInt("JXLHIP_CTX_POOL", 1)) {  // keep it for the next jxlhip_ctx_create (see RecycleContext)
    size_t bytes = 0;
    for (Buf* b : AllBufs(c))
      if (!b->view) bytes += b->cap;
    std::lock_guard<std::mutex> lk(g_ctx_pool_mu);
    if (bytes <= kCtxPoolBytes && g_ctx_pool.size() < kCtxPoolMax) {
      c->pooled_once = true;  // (a context leaves the pool through RecycleContext, which destroys the shell for good)
      g_ctx_pool.push_back(c);
      return;
    }
  }
  std::vector<Buf*> all = AllBufs(c);
  for (Buf* b : all) b->Free();
  for (auto& ev : c->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : c->enc_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (c->stage.done) (void)hipEventDestroy(c->stage.done);
  if (c->stage.p) (void)hipHostFree(c->stage.p);
  if (c->batch_done) (void)hipEventDestroy(c->batch_done);
  if (c->down_done) (void)hipEventDestroy(c->down_done);
  if (c->fork_event) (void)hipEventDestroy(c->fork_event);
  if (c->transform_done) (void)hipEventDestroy(c->transform_done);
  if (c->filter_done) (void)hipEventDestroy(c->filter_done);
  if (c->halo_event) (void)hipEventDestroy(c->halo_event);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream && c->owns_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

static int ApplyPendingWait(JxlHipContext* c) {
  if (c->pending_wait) {
    HIP_TRY(hipStreamWaitEvent(c->stream, c->pending_wait, 0));
    c->pending_wait = nullptr;
  }
  if (c->filter_wait) {
    HIP_TRY(hipStreamWaitEvent(c->stream, c->filter_wait, 0));
    c->filter_wait = nullptr;
  }
  return 0;
}

// Table uploads go through one high-priority stream per device, not through the contexts' own streams: streams share a
// few hardware queues, and a copy queued behind another context's 100 ms entropy launch on the same queue waits for it
// (6.6 ms per upload in the end-to-end pipeline against 0.7 ms alone). A priority stream gets a queue of its own.
static hipStream_t CopyStream(int device) {
  static std::mutex mu;
  static hipStream_t streams[64] = {};
  std::lock_guard<std::mutex> lock(mu);
  if (device < 0 || device >= 64) return nullptr;
  if (!streams[device]) {
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return nullptr;
    if (hipStreamCreateWithPriority(&streams[device], hipStreamNonBlocking, hi) != hipSuccess) streams[device] = nullptr;
  }
  return streams[device];
}

// Start of an upload: the previous upload's copies have left the staging block.
static int StageReset(JxlHipContext* c) {
  Stage& s = c->stage;
  if (!s.done) HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  if (s.recorded) HIP_TRY(hipEventSynchronize(s.done));
  else if (s.used) HIP_TRY(hipStreamSynchronize(c->stream));  // (an upload that failed half way)
  s.recorded = false;
  s.used = 0;
  return 0;
}
static int StageAlloc(JxlHipContext* c, size_t bytes, void** out) {
  Stage& s = c->stage;
  const size_t need = (bytes + 63) & ~size_t(63);
  if (s.used + need > s.cap) {  // a larger block: copies out of the current one may still be in flight
    if (c->blob_mode) return JXLHIP_ERR_INVALID_ARGUMENT;  // (the blob's bound was wrong: the views would dangle)
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (hipStream_t cs = CopyStream(c->device)) HIP_TRY(hipStreamSynchronize(cs));
    size_t want = s.cap * 2 > need ? s.cap * 2 : need;
    if (want < (size_t(1) << 22)) want = size_t(1) << 22;
    if (s.p) HIP_TRY(hipHostFree(s.p));
    s.p = nullptr;
    s.cap = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.p), want, hipHostMallocDefault));
    s.cap = want;
    s.used = 0;
  }
  *out = s.p + s.used;
  s.used += need;
  return 0;
}
// Copies `bytes` already assembled in the staging block (`staged`, from StageAlloc) to `b`.
static int UploadStaged(JxlHipContext* c, Buf& b, const void* staged, size_t bytes) {
  if (c->blob_mode) {
    const size_t at = size_t(static_cast<const uint8_t*>(staged) - c->stage.p);
    if (!staged || at + (bytes ? bytes : 16) > c->frame_blob.cap) return JXLHIP_ERR_INVALID_ARGUMENT;  // (the bound was wrong)
    if (!b.view) b.Free();
    b.p = c->frame_blob.as<uint8_t>() + at;
    b.cap = bytes ? bytes : 16;
    b.view = true;
    return 0;
  }
  int r = b.Ensure(bytes ? bytes : 16);
  if (r) return r;
  if (bytes) HIP_TRY(hipMemcpyAsync(b.p, staged, bytes, hipMemcpyHostToDevice, c->stream));
  return 0;
}
static int UploadStaged(JxlHipContext* c, Buf& b, const void* staged, size_t bytes);
static int UploadSplines(JxlHipContext* c, const JxlHipSplines& sp, uint32_t ysize);
static int Upload(JxlHipContext* c, Buf& b, const void* src, size_t bytes) {
  void* st = nullptr;
  int r = StageAlloc(c, bytes ? bytes : 16, &st);  // (blob mode: an empty table still gets a valid address)
  if (r) return r;
  if (bytes) memcpy(st, src, bytes);
  return UploadStaged(c, b, st, bytes);
}
// Small host-to-device copies of launch descriptions (parameter blocks, work lists): through the context's staging block
// and the copy stream like the tables, complete when End() returns.
struct ParamCopy {
  JxlHipContext* c = nullptr;
  hipStream_t cs = nullptr;
  int Begin(JxlHipContext* ctx) {
    c = ctx;
    cs = CopyStream(c->device);
    if (!cs) cs = c->stream;
    return StageReset(c);
  }
  int Add(void* dst, const void* src, size_t bytes) {
    if (!bytes) return 0;
    void* st = nullptr;
    int r = StageAlloc(c, bytes, &st);
    if (r) return r;
    memcpy(st, src, bytes);
    HIP_TRY(hipMemcpyAsync(dst, st, bytes, hipMemcpyHostToDevice, cs));
    return 0;
  }
  int End() {
    HIP_TRY(hipEventRecord(c->stage.done, cs));
    c->stage.recorded = true;
    HIP_TRY(WaitEvent(c->stage.done));
    return 0;
  }
};

// Blob mode: sizes the staging block and the device blob for `bound` bytes of tables; until BlobEnd, Upload / UploadStaged
// only place tables (the Bufs become views at the staging offsets) and BlobEnd copies everything at once.
static int BlobBegin(JxlHipContext* c, size_t bound) {
  Stage& s = c->stage;
  if (s.cap < bound) {
    void* dummy = nullptr;
    int r = StageAlloc(c, bound, &dummy);  // grows the block (the stream is idle here: StageReset ran)
    if (r) return r;
    s.used = 0;
  }
  int r = c->frame_blob.Ensure(s.cap);
  if (r) return r;
  c->blob_mode = true;
  return 0;
}
// `smooth` / `sigma`: the DC-path kernels of this frame (NULL = not asked for), launched on the copy stream behind the copy
// they read, so that the event the upload waits for covers them too.
static int BlobEnd(JxlHipContext* c, const jxlhip::DcSmoothParams* smooth = nullptr, const jxlhip::SigmaParams* sigma = nullptr,
                   const jxlhip::DcDequantParams* dequant = nullptr) {
  c->blob_mode = false;
  hipStream_t cs = CopyStream(c->device);
  if (!cs) cs = c->stream;
  if (c->stage.used) HIP_TRY(hipMemcpyAsync(c->frame_blob.p, c->stage.p, c->stage.used, hipMemcpyHostToDevice, cs));
  if (dequant) {
    hipLaunchKernelGGL(jxlhip::k_dc_dequant, dim3((dequant->xs + 63) / 64, (dequant->ys + 3) / 4), dim3(256), 0, cs, *dequant);
    HIP_TRY(hipGetLastError());
  }
  if (smooth) {
    hipLaunchKernelGGL(jxlhip::k_dc_smooth, dim3((smooth->xs + 63) / 64, (smooth->ys + 3) / 4), dim3(256), 0, cs, *smooth);
    HIP_TRY(hipGetLastError());
  }
  if (sigma && sigma->num_blocks) {
    hipLaunchKernelGGL(jxlhip::k_epf_sigma, dim3((sigma->num_blocks + 255) / 256), dim3(256), 0, cs, *sigma);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(c->stage.done, cs));
  c->stage.recorded = true;
  return 0;
}

// The patch dictionary of a frame to the device: validated like every table a kernel indexes with (rectangles inside the
// frame and inside their reference frame, row lists consistent). ysize rows; positions may reach xlimit x ylimit.
static int UploadPatches(JxlHipContext* c, const JxlHipPatches& pt, uint32_t ysize, uint32_t xlimit, uint32_t ylimit) {
  c->pat_positions = 0;
  c->pat_uses_alpha = false;
  c->alpha_patched_valid = false;
  if (!pt.num_positions) return 0;
  int r;
  if (!pt.records || !pt.row_start || !pt.row_list || pt.num_positions > (1u << 24) || pt.num_row_entries > (1u << 26)) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (pt.row_start[0] != 0 || pt.row_start[ysize] != pt.num_row_entries) return JXLHIP_ERR_INVALID_ARGUMENT;
  for (uint32_t y = 0; y < ysize; y++)
    if (pt.row_start[y] > pt.row_start[y + 1]) return JXLHIP_ERR_INVALID_ARGUMENT;
  for (uint32_t i = 0; i < pt.num_positions; i++) {
    const uint32_t* q = pt.records + size_t(i) * 8;
    const uint32_t slot = q[6];
    if (slot > 3 || !pt.slot_planes[slot] || (q[7] & 255) > 7 || ((q[7] >> 16) & 255) > 7 || !q[2] || !q[3]) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (pt.uses_alpha && !pt.slot_alpha[slot]) return JXLHIP_ERR_INVALID_ARGUMENT;  // (the reference frame kept no alpha plane)
    if (uint64_t(q[4]) + q[2] > pt.slot_w[slot] || uint64_t(q[5]) + q[3] > pt.slot_h[slot]) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (uint64_t(q[0]) + q[2] > xlimit || uint64_t(q[1]) + q[3] > ylimit) return JXLHIP_ERR_INVALID_ARGUMENT;
  }
  for (uint32_t y = 0; y < ysize; y++)
    for (uint32_t i = pt.row_start[y]; i < pt.row_start[y + 1]; i++) {
      if (pt.row_list[i] >= pt.num_positions) return JXLHIP_ERR_INVALID_ARGUMENT;
      const uint32_t* q = pt.records + size_t(pt.row_list[i]) * 8;
      if (y < q[1] || y >= q[1] + q[3]) return JXLHIP_ERR_INVALID_ARGUMENT;  // (the row is one of the patch's rows)
    }
  if ((r = Upload(c, c->pat_rec, pt.records, size_t(pt.num_positions) * 32))) return r;
  if ((r = Upload(c, c->pat_row_start, pt.row_start, (size_t(ysize) + 1) * 4))) return r;
  if ((r = Upload(c, c->pat_row_list, pt.row_list, std::max<size_t>(4, size_t(pt.num_row_entries) * 4)))) return r;
  c->pat_uses_alpha = pt.uses_alpha != 0;
  c->pat_premultiplied = pt.premultiplied != 0;
  for (int i = 0; i < 4; i++) {
    c->pat_src_alpha[i] = pt.slot_alpha[i];
    c->pat_src[i] = pt.slot_planes[i];
    c->pat_src_w[i] = pt.slot_w[i];
    c->pat_src_h[i] = pt.slot_h[i];
  }
  c->pat_positions = pt.num_positions;
  return 0;
}

int jxlhip_frame_upload(JxlHipContext* c, const JxlHipFrameDesc* d) {
  if (!c || !d) return JXLHIP_ERR_INVALID_ARGUMENT;
  // measurement aid: JXLHIP_UPLOAD_PROF=1 prints where the host time of an upload goes (microseconds per phase)
  const bool prof = EnvInt("JXLHIP_UPLOAD_PROF", 0) != 0;
  std::chrono::steady_clock::time_point tp0 = std::chrono::steady_clock::now();
  std::string prof_line;
  auto lap = [&](const char* what) {
    if (!prof) return;
    const auto now = std::chrono::steady_clock::now();
    prof_line += std::string(what) + " " + std::to_string(std::chrono::duration_cast<std::chrono::microseconds>(now - tp0).count()) + "  ";
    tp0 = now;
  };
  if (!d->xsize || !d->ysize || !d->num_groups || !d->num_passes || d->num_passes > 11) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (d->coef_bits != 16 && d->coef_bits != 32) return JXLHIP_ERR_INVALID_ARGUMENT;
  uint32_t cs = 0, cs_maxh = 0, cs_maxv = 0;
  for (int ch = 0; ch < 3; ch++) {
    if (d->chroma_hshift[ch] > 1 || d->chroma_vshift[ch] > 1) return JXLHIP_ERR_INVALID_ARGUMENT;
    cs |= uint32_t(d->chroma_hshift[ch]) << (2 * ch) | uint32_t(d->chroma_vshift[ch]) << (2 * ch + 1);
    cs_maxh |= d->chroma_hshift[ch];
    cs_maxv |= d->chroma_vshift[ch];
  }
  // (frame_dimensions.h:43-44: whole MCUs of a chroma-subsampled frame)
  if (d->xsize_blocks != ((d->xsize + (8u << cs_maxh) - 1) / (8u << cs_maxh)) << cs_maxh ||
      d->ysize_blocks != ((d->ysize + (8u << cs_maxv) - 1) / (8u << cs_maxv)) << cs_maxv)
    return JXLHIP_ERR_INVALID_ARGUMENT;
  if (cs) {
    // no adaptive DC smoothing (dec_frame.cc:206-212), no DC frame behind it, every varblock one block (checked below);
    // a band whose neighbours' rows arrive as halos cannot upsample its edge rows: the other band form works
    if (d->dc_smoothing || d->dc_device || d->linear_output != 2) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (c->band_halo && (d->band_group_row_begin | d->band_group_row_end)) return JXLHIP_ERR_UNSUPPORTED;
  }
  if (d->num_groups != d->xsize_groups * ((d->ysize + 255) / 256) || d->xsize_groups != (d->xsize + 255) / 256)
    return JXLHIP_ERR_INVALID_ARGUMENT;
  if (d->num_qf_thresholds > 15 || d->num_block_ctxs == 0 || d->num_block_ctxs > 16 || d->num_dc_ctxs == 0)
    return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
    if ((pw = StageReset(c))) return pw;
    // the tables are overwritten from the copy stream: whatever still reads the old ones has to be finished
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->stream2) HIP_TRY(hipStreamSynchronize(c->stream2));
  }
  c->have_frame = false;
  c->cs = cs;
  c->xs = d->xsize; c->ys = d->ysize; c->xb = d->xsize_blocks; c->yb = d->ysize_blocks;
  c->xg = d->xsize_groups; c->ng = d->num_groups; c->np = d->num_passes;
  c->nblocks = d->num_blocks;
  c->xp = c->xb * 8; c->yp = c->yb * 8;
  c->ups = d->upsampling > 1 ? d->upsampling : 1;
  c->oxs = c->ups == 1 ? d->xsize : d->out_xsize;
  c->oys = c->ups == 1 ? d->ysize : d->out_ysize;
  if (c->ups != 1) {
    if ((c->ups != 2 && c->ups != 4 && c->ups != 8) || !d->upsampling_kernel || (d->band_group_row_begin | d->band_group_row_end) ||
        (c->oxs + c->ups - 1) / c->ups != d->xsize || (c->oys + c->ups - 1) / c->ups != d->ysize)
      return JXLHIP_ERR_INVALID_ARGUMENT;
  }
  uint32_t ext_row0 = 0, ext_row1 = d->num_groups / d->xsize_groups;  // group rows to entropy-decode and transform
  {
    const uint32_t yg = d->num_groups / d->xsize_groups;
    uint32_t rb = d->band_group_row_begin, re = d->band_group_row_end;
    if (rb == 0 && re == 0) re = yg;
    if (rb >= re || re > yg) return JXLHIP_ERR_INVALID_ARGUMENT;
    ext_row0 = (rb && !c->band_halo) ? rb - 1 : rb;
    ext_row1 = (re < yg && !c->band_halo) ? re + 1 : re;
    c->band_y0 = rb * 256;
    c->ext_y0 = ext_row0 * 256;
    c->ext_y1 = ext_row1 * 256 < d->ysize ? ext_row1 * 256 : d->ysize;
    c->band_y1 = re * 256 < d->ysize ? re * 256 : d->ysize;
    c->group_list.clear();
    c->absent_groups.clear();
    c->absent_sections.clear();
    for (uint32_t g = ext_row0 * d->xsize_groups; g < ext_row1 * d->xsize_groups; g++) {
      if (d->group_absent && d->group_absent[g]) {
        for (uint32_t p = 0; p < d->num_passes; p++)
          if (d->section_size[size_t(p) * d->num_groups + g]) return JXLHIP_ERR_INVALID_ARGUMENT;
        c->absent_groups.push_back(g);
      } else {
        c->group_list.push_back(g);
        // a partial frame's group is drawn from its leading passes (dec_frame.cc:620-680): the missing ones have size 0,
        // and once one is missing all later ones are
        bool missing = false;
        for (uint32_t p = 1; d->group_absent && p < d->num_passes; p++) {
          const bool empty = d->section_size[size_t(p) * d->num_groups + g] == 0;
          if (missing && !empty) return JXLHIP_ERR_INVALID_ARGUMENT;
          missing = missing || empty;
          if (empty) {
            c->absent_sections.push_back(p);
            c->absent_sections.push_back(d->group_block_begin[g]);
            c->absent_sections.push_back(d->group_block_begin[g + 1] - d->group_block_begin[g]);
          }
        }
      }
    }
  }
  c->coef_bits = d->coef_bits;
  c->gab = d->gab; c->epf_iters = d->epf_iters;
  // ---- validate varblocks against the geometry (the kernels index with these)
  {
    uint32_t prev_end = 0;
    for (uint32_t g = 0; g < d->num_groups; g++) {
      const uint32_t b0 = d->group_block_begin[g], b1 = d->group_block_begin[g + 1];
      if (b0 != prev_end || b1 < b0 || b1 > d->num_blocks) return JXLHIP_ERR_INVALID_ARGUMENT;
      prev_end = b1;
      uint32_t off = 0;
      const uint32_t gx0 = (g % d->xsize_groups) * 32, gy0 = (g / d->xsize_groups) * 32;
      for (uint32_t i = b0; i < b1; i++) {
        const JxlHipVarBlock& v = d->blocks[i];
        if (v.strategy >= 27 || v.qf == 0 || v.qf > 256 || v.quant_dc_ctx >= d->num_dc_ctxs) return JXLHIP_ERR_INVALID_ARGUMENT;
        static const uint8_t cx[27] = {1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32};
        static const uint8_t cy[27] = {1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16};
        if (v.bx < gx0 || v.by < gy0 || v.bx + cx[v.strategy] > gx0 + 32 || v.by + cy[v.strategy] > gy0 + 32 ||
            v.bx + cx[v.strategy] > d->xsize_blocks || v.by + cy[v.strategy] > d->ysize_blocks)
          return JXLHIP_ERR_INVALID_ARGUMENT;
        if (v.coef_offset != off) return JXLHIP_ERR_INVALID_ARGUMENT;
        if (cs && cx[v.strategy] * cy[v.strategy] != 1) return JXLHIP_ERR_INVALID_ARGUMENT;  // (dec_modular.cc:534-538)
        off += 64u * cx[v.strategy] * cy[v.strategy];
        if (off > 65536) return JXLHIP_ERR_INVALID_ARGUMENT;
      }
    }
    if (prev_end != d->num_blocks) return JXLHIP_ERR_INVALID_ARGUMENT;
  }
  lap("validate");
  // ---- pack the AC sections at 16-byte aligned offsets
  const size_t nsec = size_t(d->num_groups) * d->num_passes;
  std::vector<uint32_t> sec_word(nsec), sec_size(nsec);
  size_t total = 0;
  for (size_t i = 0; i < nsec; i++) {
    sec_word[i] = uint32_t(total / 4);
    sec_size[i] = d->section_size[i];
    total += (size_t(d->section_size[i]) + 15 + 4) & ~size_t(15);
  }
  int r;
  {
    // every table this upload places (each rounded up to the staging granule), generously
    size_t bound = 65536;
    auto add = [&](size_t bytes) { bound += (bytes + 16 + 127) & ~size_t(63); };
    const size_t nblk_b = size_t(d->xsize_blocks) * d->ysize_blocks, ntiles_b = size_t((d->xsize_blocks + 7) / 8) * ((d->ysize_blocks + 7) / 8);
    add(total + 256); add(nsec * 4); add(nsec * 4);
    add(size_t(d->num_blocks) * sizeof(JxlHipVarBlock)); add((size_t(d->num_groups) + 1) * 4); add(d->block_ctx_lut_size);
    add(size_t(d->dequant_floats) * 4); add(nblk_b * 12); add(nblk_b * 4); add(nblk_b); add(ntiles_b); add(ntiles_b); add(ntiles_b + 64);
    for (uint32_t p = 0; p < d->num_passes; p++) {
      const JxlHipPassDesc& q = d->passes[p];
      if (q.num_clusters > 256 || (!q.use_prefix && q.log_alpha > 8)) return JXLHIP_ERR_INVALID_ARGUMENT;
      add(q.ctx_map_size); add(q.use_prefix ? 0 : (size_t(q.num_clusters) << q.log_alpha) * 8); add(size_t(q.num_clusters) * 4);
      add(q.use_prefix ? 0 : (size_t(q.num_clusters) << q.log_alpha) * 8);  // (the lane kernel's packed form)
      add(size_t(q.orders_size) * 2); add(size_t(q.prefix_table_size) * 4); add(size_t(q.num_clusters) * 4);
    }
    add(size_t(d->num_passes) * sizeof(jxlhip::PassDev)); add(64 * 25 * 4); add((size_t(d->num_blocks) + 1) * 4);
    add((size_t(d->num_blocks) + 1) * 16);
    add((size_t(d->num_blocks) + 16) * 4); add(sizeof(jxlhip::EntropyParams)); add(size_t(d->dequant_floats) * 4);
    add(size_t(d->splines.num_segments) * 32); add((size_t(d->ysize) + 1) * 4); add(size_t(d->splines.num_row_segments) * 4 + 16);
    add(size_t(d->patches.num_positions) * 32); add((size_t(d->ysize) + 1) * 4); add(size_t(d->patches.num_row_entries) * 4 + 16);
    if ((r = BlobBegin(c, bound))) return r;
  }
  struct BlobGuard {  // whatever the outcome, later uploads of other kinds see the plain mode
    JxlHipContext* c;
    ~BlobGuard() { c->blob_mode = false; }
  } blob_guard{c};
  {
    uint8_t* packed = nullptr;  // (+ 256: the lane kernel prefetches up to 64 bytes past a section)
    if ((r = StageAlloc(c, total + 256, reinterpret_cast<void**>(&packed)))) return r;
    size_t end = 0;
    for (size_t i = 0; i < nsec; i++) {
      const size_t at = size_t(sec_word[i]) * 4;
      memset(packed + end, 0, at - end);
      memcpy(packed + at, d->codestream + d->section_offset[i], d->section_size[i]);
      end = at + d->section_size[i];
    }
    memset(packed + end, 0, total + 256 - end);
    if ((r = UploadStaged(c, c->sections, packed, total + 256))) return r;
  }
  if ((r = Upload(c, c->sec_word, sec_word.data(), nsec * 4))) return r;
  if ((r = Upload(c, c->sec_size, sec_size.data(), nsec * 4))) return r;
  c->sec_size_host = sec_size;
  {
    // histogram selector of every first-pass section: its first ceil(log2(num_histograms)) bits (dec_group.cc:594-610)
    uint32_t hb = 0;
    while ((1u << hb) < d->num_histograms) hb++;
    c->sec_sel_host.assign(nsec, 0);
    for (size_t g = 0; hb && g < nsec; g++) {  // [pass * num_groups + group]
      const uint8_t* p = d->codestream + d->section_offset[g];
      const uint32_t bit0 = g == 0 ? d->first_section_bit_offset : 0;
      uint64_t v = 0;
      for (uint32_t b = 0; b < 8 && b < d->section_size[g]; b++) v |= uint64_t(p[b]) << (8 * b);
      c->sec_sel_host[g] = uint32_t((v >> bit0) & ((1u << hb) - 1));
    }
  }
  c->pass_clusters.assign(d->num_passes, 0);
  c->pass_log_alpha.assign(d->num_passes, 0);
  lap("sections");
  if ((r = Upload(c, c->blocks, d->blocks, size_t(d->num_blocks) * sizeof(JxlHipVarBlock)))) return r;
  if ((r = Upload(c, c->gbb, d->group_block_begin, (size_t(d->num_groups) + 1) * 4))) return r;
  if ((r = Upload(c, c->bctx_lut, d->block_ctx_lut, d->block_ctx_lut_size))) return r;
  if ((r = Upload(c, c->dequant, d->dequant, size_t(d->dequant_floats) * 4))) return r;
  const size_t nblk = size_t(c->xb) * c->yb;
  // DC image and 1 / sigma: final as handed over, or finished on the device (jxl_hip_dc.h) behind the table copy
  jxlhip::DcSmoothParams smooth_p;
  jxlhip::SigmaParams sigma_p;
  const bool dev_smooth = d->dc_smoothing != 0 && c->xb > 2 && c->yb > 2;  // (compressed_dc.cc:134: smaller images are left alone)
  const bool dev_sigma = !d->inv_sigma && d->epf_iters > 0;
  jxlhip::DcDequantParams dequant_p;
  const bool dev_dequant = d->dc_quantised != nullptr && !d->dc_device;
  if ((!d->dc && !d->dc_device && !d->dc_quantised) || (dev_sigma && !d->sharpness)) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (dev_dequant) {
    for (int ch = 0; ch < 3; ch++)
      if (!(d->dc_step[ch] > 0.0f)) return JXLHIP_ERR_INVALID_ARGUMENT;
    const uint32_t xdg = (c->xb + 255) / 256, ydg = (c->yb + 255) / 256;
    if ((r = Upload(c, c->dc_q, d->dc_quantised, nblk * 3 * 4))) return r;
    if (d->dc_extra_precision && (r = Upload(c, c->dc_ep, d->dc_extra_precision, size_t(xdg) * ydg))) return r;
    dequant_p.q = c->dc_q.as<int32_t>();
    dequant_p.xs = c->xb;
    dequant_p.ys = c->yb;
    dequant_p.xgroups = xdg;
    dequant_p.extra_precision = d->dc_extra_precision ? c->dc_ep.as<uint8_t>() : nullptr;
    for (int ch = 0; ch < 3; ch++) dequant_p.step[ch] = d->dc_step[ch];
    dequant_p.cfl_x = d->dc_cfl_x;
    dequant_p.cfl_b = d->dc_cfl_b;
    dequant_p.cs = cs;
  }
  if (d->dc_device) {  // the planes of a DC frame decoded earlier (kUseDcFrame): used where they are, never smoothed
    if (!c->dc.view) c->dc.Free();
    c->dc.p = const_cast<float*>(d->dc_device);
    c->dc.cap = nblk * 3 * 4;
    c->dc.view = true;
  } else if (dev_smooth) {
    for (int ch = 0; ch < 3; ch++)
      if (!(d->dc_step[ch] > 0.0f)) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (dev_dequant) {  // (the dequantised planes are a kernel's output: an allocation of their own, not a view of the blob)
      if ((r = c->dc_raw.Ensure(nblk * 3 * 4))) return r;
      dequant_p.out = c->dc_raw.as<float>();
    } else if ((r = Upload(c, c->dc_raw, d->dc, nblk * 3 * 4))) {
      return r;
    }
    if ((r = c->dc.Ensure(nblk * 3 * 4))) return r;
    smooth_p.in = c->dc_raw.as<float>();
    smooth_p.out = c->dc.as<float>();
    smooth_p.xs = c->xb;
    smooth_p.ys = c->yb;
    for (int ch = 0; ch < 3; ch++) smooth_p.step[ch] = d->dc_step[ch];
  } else if (dev_dequant) {
    if ((r = c->dc.Ensure(nblk * 3 * 4))) return r;
    dequant_p.out = c->dc.as<float>();
  } else if ((r = Upload(c, c->dc, d->dc, nblk * 3 * 4))) {
    return r;
  }
  if (dev_sigma) {
    if (!(d->quant_scale > 0.0f)) return JXLHIP_ERR_INVALID_ARGUMENT;
    if ((r = Upload(c, c->sharp, d->sharpness, nblk))) return r;
    if ((r = c->inv_sigma.Ensure(nblk * 4))) return r;
    sigma_p.blocks = c->blocks.as<JxlHipVarBlock>();
    sigma_p.num_blocks = d->num_blocks;
    sigma_p.xb = c->xb;
    sigma_p.sharpness = c->sharp.as<uint8_t>();
    sigma_p.inv_sigma = c->inv_sigma.as<float>();
    sigma_p.quant_scale = d->quant_scale;
    sigma_p.epf_quant_mul = d->epf_quant_mul;
    memcpy(sigma_p.sharp_lut, d->epf_sharp_lut, sizeof(sigma_p.sharp_lut));
  } else if (d->inv_sigma) {
    if ((r = Upload(c, c->inv_sigma, d->inv_sigma, nblk * 4))) return r;
  } else if ((r = c->inv_sigma.Ensure(nblk * 4))) {  // (no EPF: never read)
    return r;
  }
  const size_t ntiles = size_t((c->xb + 7) / 8) * ((c->yb + 7) / 8);
  if ((r = Upload(c, c->ytox, d->ytox, ntiles))) return r;
  if ((r = Upload(c, c->ytob, d->ytob, ntiles))) return r;
  lap("planes");
  // ---- per-pass tables
  if (c->pass_bufs.size() < d->num_passes) c->pass_bufs.resize(d->num_passes);
  std::vector<jxlhip::PassDev> pd(d->num_passes);
  const uint32_t nctx = d->num_block_ctxs * 495;
  size_t alias_bytes_max = 0;
  bool generic = false, any_lz77 = false;
  for (uint32_t p = 0; p < d->num_passes; p++) {
    const JxlHipPassDesc& s = d->passes[p];
    if (s.num_clusters == 0 || s.num_clusters > 256) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (!s.use_prefix && (s.log_alpha < 5 || s.log_alpha > 8)) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (s.use_prefix) {  // every table index the kernel can form must be inside the table
      if (!s.prefix_offset || !s.prefix_table) return JXLHIP_ERR_INVALID_ARGUMENT;
      for (uint32_t i = 0; i < s.num_clusters; i++) {
        if (!ValidPrefixTables(s.prefix_offset[i], s.prefix_table, s.prefix_table_size)) return JXLHIP_ERR_INVALID_ARGUMENT;
      }
    }
    if (s.lz77 && (s.lz_dist_ctx >= s.num_clusters || s.lz_min_length == 0)) return JXLHIP_ERR_INVALID_ARGUMENT;
    generic = generic || s.use_prefix || s.lz77;
    if (s.ctx_map_size < size_t(d->num_histograms) * nctx + 16) return JXLHIP_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < s.ctx_map_size; i++)
      if (s.ctx_map[i] >= s.num_clusters) return JXLHIP_ERR_INVALID_ARGUMENT;
    PassBufs& pb = c->pass_bufs[p];
    const size_t alias_bytes = s.use_prefix ? 0 : (size_t(s.num_clusters) << s.log_alpha) * 8;
    alias_bytes_max = alias_bytes > alias_bytes_max ? alias_bytes : alias_bytes_max;
    if ((r = Upload(c, pb.ctx_map, s.ctx_map, s.ctx_map_size))) return r;
    if ((r = Upload(c, pb.alias, s.alias, alias_bytes))) return r;
    {
      // alias entry {cutoff u8, right u8, freq0 u16 | offsets1 u16, freq1 u16} -> the lane kernel's form
      //   x = (freq0 - 1) & 0xFFF | uint config << 12 | cutoff << 24   taken when pos <  cutoff: symbol = slot, offset = pos
      //   y = (freq1 - 1) & 0xFFF | offsets1 << 12 | right << 24        taken when pos >= cutoff
      // (uint config of the entry's cluster: split_exponent | msb_in_token << 4 | lsb_in_token << 8, validated below)
      void* st = nullptr;
      if ((r = StageAlloc(c, alias_bytes ? alias_bytes : 16, &st))) return r;
      const uint32_t* src = reinterpret_cast<const uint32_t*>(s.alias);
      uint32_t* dst = static_cast<uint32_t*>(st);
      for (size_t i = 0; i < alias_bytes / 8; i++) {
        const uint32_t ex = src[2 * i], ey = src[2 * i + 1];
        const uint32_t cutoff = ex & 0xFF, right = (ex >> 8) & 0xFF, freq0 = ex >> 16, offs1 = ey & 0xFFFF, freq1 = ey >> 16;
        const uint32_t uc = s.uint_cfg[i >> s.log_alpha];
        const uint32_t cfg12 = (uc & 15u) | (((uc >> 8) & 15u) << 4) | (((uc >> 16) & 15u) << 8);
        dst[2 * i] = ((freq0 - 1) & 0xFFFu) | (cfg12 << 12) | (cutoff << 24);
        dst[2 * i + 1] = ((freq1 - 1) & 0xFFFu) | ((offs1 & 0xFFFu) << 12) | (right << 24);
      }
      if ((r = UploadStaged(c, pb.alias_packed, st, alias_bytes))) return r;
    }
    if ((r = Upload(c, pb.cfg, s.uint_cfg, size_t(s.num_clusters) * 4))) return r;
    if ((r = Upload(c, pb.orders, s.orders, size_t(s.orders_size) * 2))) return r;
    pd[p].ctx_map = pb.ctx_map.as<uint8_t>();
    pd[p].alias = pb.alias.as<uint2>();
    pd[p].alias_packed = pb.alias_packed.as<uint2>();
    pd[p].cfg = pb.cfg.as<uint32_t>();
    pd[p].orders = pb.orders.as<uint16_t>();
    memcpy(pd[p].order_offset, s.order_offset, sizeof(s.order_offset));
    for (uint32_t i = 0; i < s.num_clusters; i++) {  // hybrid-uint configs: split_exponent, msb_in_token, lsb_in_token
      const uint32_t se = s.uint_cfg[i] & 0xFF, msb = (s.uint_cfg[i] >> 8) & 0xFF, lsb = (s.uint_cfg[i] >> 16) & 0xFF;
      if (se > (s.use_prefix ? 15u : s.log_alpha) || msb > se || lsb > se - msb) return JXLHIP_ERR_INVALID_ARGUMENT;
    }
    if ((r = Upload(c, pb.ptable, s.prefix_table, s.use_prefix ? size_t(s.prefix_table_size) * 4 : 0))) return r;
    if ((r = Upload(c, pb.poffset, s.prefix_offset, s.use_prefix ? size_t(s.num_clusters) * 4 : 0))) return r;
    pd[p].use_prefix = s.use_prefix ? 1 : 0;
    pd[p].lz77 = s.lz77 ? 1 : 0;
    pd[p].lz_min_symbol = s.lz_min_symbol;
    pd[p].lz_min_length = s.lz_min_length;
    pd[p].lz_len_cfg = s.lz_len_cfg;
    pd[p].lz_dist_ctx = s.lz_dist_ctx;
    pd[p].prefix_table = pb.ptable.as<uint32_t>();
    pd[p].prefix_offset = pb.poffset.as<uint32_t>();
    if (s.lz77) any_lz77 = true;
    pd[p].log_alpha = s.log_alpha;
    c->pass_clusters[p] = s.num_clusters;
    c->pass_log_alpha[p] = s.log_alpha;
    pd[p].num_clusters = s.num_clusters;
    pd[p].shift = s.shift;
    pd[p].alias_lds = 0;
  }
  if ((r = Upload(c, c->passes_dev, pd.data(), pd.size() * sizeof(jxlhip::PassDev)))) return r;
  lap("tables");
  // ---- work buffers
  // (progressive frames: the natural-layout buffer + one scan-order buffer per pass for the lane kernel)
  const size_t coef_bytes = size_t(d->num_groups) * 3 * 65536 * (d->coef_bits / 8);
  // (+ 64 bytes: a lane of the entropy kernel that decodes a CORRUPT section may write up to three coefficients past the
  // last scan position of a block before it is stopped, jxl_hip_lanes_trip.inc: past the buffer for the very last block)
  if ((r = c->coeffs.Ensure(coef_bytes * (d->num_passes > 1 ? d->num_passes + 1 : 1) + 64))) return r;
  if ((r = c->errors.Ensure(size_t(d->num_groups) * 4))) return r;
  HIP_TRY(hipMemsetAsync(c->errors.p, 0, size_t(d->num_groups) * 4, c->stream));  // groups outside a band stay clean
  const size_t plane_bytes = size_t(c->xp) * c->yp * 3 * 4;
  c->plane_bytes = plane_bytes;
  if (!c->plane_lender && (r = c->plane[0].Ensure(plane_bytes))) return r;
  // pixels: RGB8 from every filter kernel, RGB f32 from the row-streaming one; any other format from the generic writer
  // (k_color_out on the filtered planes, or k_upsample_color)
  c->color_out = !OutIsRgb8(c) && !(OutIsRgbF32(c) && c->ups == 1 && (c->epf_iters == 1 || c->epf_iters == 2) &&
                                    !EnvInt("JXLHIP_FILTER_TILES", 0) && !EnvInt("JXLHIP_FILTER_ROWS1", 0));
  if (c->out_orient) c->color_out = true;  // (the oriented layout is written by the generic writer)
  // frames of images that are not XYB encoded (linear_output 2 = YCbCr, 3 = no colour transform): the colour stage is
  // XybToRgb's other branches, which only the generic writers take
  if (d->linear_output < 0 || d->linear_output > 3) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (d->linear_output >= 2) c->color_out = true;
  // noise is added to the filtered planes between the filter launch and the colour conversion
  c->has_noise = d->has_noise != 0;
  if (c->has_noise) {
    c->color_out = true;
    memcpy(c->noise_lut, d->noise_lut, sizeof(c->noise_lut));
    c->noise_seed[0] = d->noise_frame_index[0];
    c->noise_seed[1] = d->noise_frame_index[1];
    c->noise_ytox = d->base_corr_x;
    c->noise_ytob = d->base_corr_b;
    // (an upsampled frame's noise is generated and added at the IMAGE's resolution, behind the upsampling:
    // dec_cache.cc:206-216, dec_group.cc PrepareNoiseInput: one generator per 256 x 256 square of the image)
    const size_t nxs = c->ups == 1 ? c->xs : c->oxs, nys = c->ups == 1 ? c->ys : c->oys;
    if ((r = c->noise.Ensure(nxs * nys * 3 * 4))) return r;
    if (c->ups != 1 && (r = c->ups_planes.Ensure(size_t((c->oxs + 7) & ~7u) * c->oys * 3 * 4))) return r;
  }
  // splines are drawn over the filtered planes before noise and the colour conversion
  // (an upsampled frame's splines and patches are drawn at its own resolution, on the planes the upsampling kernel reads:
  // dec_cache.cc:193-212)
  if (d->splines.num_segments) c->color_out = true;
  if ((r = UploadSplines(c, d->splines, c->ys))) return r;
  // patches: validated like every table a kernel indexes with (rectangles inside the frame and inside their reference)
  if ((r = UploadPatches(c, d->patches, c->ys, c->xp, c->yp))) return r;
  if (c->pat_positions) c->color_out = true;
  if (c->pat_uses_alpha && c->ups != 1) return JXLHIP_ERR_UNSUPPORTED;
  if (c->have_alpha && c->alpha.cap < size_t(c->oxs) * c->oys * 4) return JXLHIP_ERR_INVALID_ARGUMENT;
  if ((c->keep_filtered || c->ups != 1 || c->color_out) && (r = c->plane[1].Ensure(plane_bytes))) return r;
  if ((r = c->rgb.Ensure(size_t(c->oxs) * c->oys * OutPixelBytes(c)))) return r;
  if (c->ups != 1) {
    if ((r = Upload(c, c->ups_kernel, d->upsampling_kernel, size_t(c->ups) * c->ups * 25 * 4))) return r;
  }
  lap("buffers");
  // ---- transform work lists (block indices bucketed by strategy)
  {
    std::vector<uint32_t> count(27, 0);
    auto in_band = [&](uint32_t i) { return uint32_t(d->blocks[i].by >> 5) >= ext_row0 && uint32_t(d->blocks[i].by >> 5) < ext_row1; };
    for (uint32_t i = 0; i < d->num_blocks; i++)
      if (in_band(i)) count[d->blocks[i].strategy]++;
    uint32_t acc = 0;
    for (int s = 0; s < 27; s++) {
      c->list_begin[s] = acc;
      c->list_count[s] = count[s];
      acc += count[s];
    }
    std::vector<uint32_t> list(d->num_blocks ? d->num_blocks : 1), fill(27, 0);
    for (uint32_t i = 0; i < d->num_blocks; i++) {
      if (!in_band(i)) continue;
      const int s = d->blocks[i].strategy;
      list[c->list_begin[s] + fill[s]++] = i;
    }
    if ((r = Upload(c, c->tlist, list.data(), list.size() * 4))) return r;
    {
      // ... and, in the same order, everything k_idct_fast needs to know of a varblock in ONE 16-byte load (the kernel's
      // waves live for a handful of dependent memory round trips: list entry -> block -> coefficients was three of them):
      // {bx | by << 16, raw quant field, element offset of its coefficients (group * 3 * 65536 + offset in the group's planes),
      // index of the varblock (for its coefficient counts)}
      void* st = nullptr;
      if ((r = StageAlloc(c, list.size() * 16, &st))) return r;
      uint32_t* rec = static_cast<uint32_t*>(st);
      for (size_t j = 0; j < list.size(); j++) {
        const uint32_t i = list[j];
        if (i >= d->num_blocks || (j >= acc)) {
          rec[4 * j] = rec[4 * j + 1] = rec[4 * j + 2] = rec[4 * j + 3] = 0;
          continue;
        }
        const JxlHipVarBlock& v = d->blocks[i];
        const uint32_t g = uint32_t(v.by >> 5) * d->xsize_groups + (v.bx >> 5);
        rec[4 * j] = uint32_t(v.bx) | uint32_t(v.by) << 16;
        rec[4 * j + 1] = v.qf;
        rec[4 * j + 2] = g * (3u * 65536u) + v.coef_offset;
        rec[4 * j + 3] = i;
      }
      if ((r = UploadStaged(c, c->trecs, st, list.size() * 16))) return r;
    }
    uint32_t big = 0;
    for (int s = 21; s < 27; s++) big = big > count[s] ? big : count[s];
    if (big) {
      const uint32_t chunk = big < 32 ? big : 32;
      if ((r = c->scratch.Ensure(size_t(chunk) * 3 * 2 * 65536 * 4))) return r;
    }
  }
  lap("lists");
  // ---- kernel parameter blocks
  jxlhip::EntropyParams& ep = c->ep;
  memset(&ep, 0, sizeof(ep));
  ep.sections = c->sections.as<uint32_t>();
  ep.sec_word = c->sec_word.as<uint32_t>();
  ep.sec_size = c->sec_size.as<uint32_t>();
  ep.first_bit_offset = d->first_section_bit_offset;
  ep.blocks = c->blocks.as<JxlHipVarBlock>();
  ep.gbb = c->gbb.as<uint32_t>();
  ep.bctx_lut = c->bctx_lut.as<uint8_t>();
  ep.nq = d->num_qf_thresholds + 1;
  ep.ndc = d->num_dc_ctxs;
  ep.num_bctx = d->num_block_ctxs;
  for (uint32_t i = 0; i < d->num_qf_thresholds; i++) ep.qf_thr[i] = d->qf_thresholds[i];
  ep.num_hist = d->num_histograms;
  ep.nctx = nctx;
  ep.cs = cs;
  ep.passes = c->passes_dev.as<jxlhip::PassDev>();
  ep.num_passes = d->num_passes;
  ep.num_groups = d->num_groups;
  ep.coeffs = c->coeffs.p;
  ep.errors = c->errors.as<uint32_t>();
  c->generic_codec = generic;
  ep.lz_window = nullptr;
  if (any_lz77) {
    if ((r = c->lz_window.Ensure(size_t(d->num_groups) * jxlhip::kLzWindow * 4))) return r;
    ep.lz_window = c->lz_window.as<uint32_t>();
  }
  if ((r = c->sec_end.Ensure(nsec * 4))) return r;
  HIP_TRY(hipMemsetAsync(c->sec_end.p, 0, nsec * 4, c->stream));
  ep.sec_end_bits = c->sec_end.as<uint32_t>();
  ep.lds_ctx_bytes = (nctx + 16 + 15) & ~15u;
  const size_t lds_budget = 150 * 1024;
  c->alias_lds = ep.lds_ctx_bytes + alias_bytes_max + 1024 + 4 * 5120 <= lds_budget;
  ep.lds_alias_bytes = c->alias_lds ? uint32_t((alias_bytes_max + 15) & ~size_t(15)) : 0;
  c->lds_entropy = ep.lds_ctx_bytes + ep.lds_alias_bytes + 3072;
  if (size_t(d->block_ctx_lut_size) < size_t(3) * 13 * ep.nq * ep.ndc) return JXLHIP_ERR_INVALID_ARGUMENT;
  // single-pass frames whose tables fit LDS are decoded by the lane-parallel kernel into scan order
  // (its packed block records hold the block contexts in 4 bits each: the codestream allows at most 16)
  bool all_prefix = true;
  for (uint32_t p = 0; p < d->num_passes; p++) all_prefix = all_prefix && d->passes[p].use_prefix;
  c->lane_prefix = all_prefix && !any_lz77 && !EnvInt("JXLHIP_NO_LANE_PREFIX", 0);
  c->lanes = EntropyKernelChoice() == 2 && ep.num_bctx <= 16 && d->num_passes <= 8 && (!generic || c->lane_prefix);
  for (uint32_t p = 0; c->lanes && p < d->num_passes; p++)  // (alias tables that do not fit LDS are read in place)
    c->lanes = jxlhip::LanesLdsLayout(1, ep.nctx, c->pass_clusters[p], c->pass_log_alpha[p], kLanesWPG, 64, false, c->lane_prefix).total <= kLdsBudget;
  if (!c->lanes) c->lane_prefix = false;
  // (sections that have not arrived: the lane kernel takes a list of the ones that have; the section-per-workgroup kernels
  // step over a section of size 0 and leave the group's coefficients zeroed)
  c->scan_order = c->lanes && d->num_passes == 1;
  c->lane_multi = c->lanes && d->num_passes > 1;
  const size_t kend_per_pass = size_t(d->num_blocks ? d->num_blocks : 1) * 3;
  if ((r = c->kend.Ensure(kend_per_pass * 4 * d->num_passes))) return r;
  ep.kend = c->kend.as<uint32_t>();
  ep.coef_pass_base = c->lane_multi ? uint64_t(d->num_groups) * 3 * 65536 : 0;
  ep.coef_pass_stride = uint64_t(d->num_groups) * 3 * 65536;
  ep.kend_pass_stride = uint32_t(kend_per_pass);
  if (c->lane_multi) {  // k_merge_passes scatters through every pass's orders: validate them here
    static const uint8_t kind_of[27] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};
    static const uint8_t bucket_of[27] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
    for (uint32_t p = 0; p < d->num_passes; p++)
      for (int st = 0; st < 27; st++) {
        if (!c->list_count[st]) continue;
        const uint32_t size = d->dequant_size[kind_of[st]];
        for (int ch = 0; ch < 3; ch++) {
          const uint32_t oo = d->passes[p].order_offset[bucket_of[st] * 3 + ch];
          if (size_t(oo) + size > d->passes[p].orders_size) return JXLHIP_ERR_INVALID_ARGUMENT;
          for (uint32_t k = 0; k < size; k++)
            if (d->passes[p].orders[oo + k] >= size) return JXLHIP_ERR_INVALID_ARGUMENT;
        }
      }
  }
  {
    // Per varblock, everything the lane kernel's block transition needs in one word, so that it does no dependent
    // table lookups: column in the group (5 bits) | not in the group's first row (1) | log2 covered_x (3) |
    // log2 covered_y (3) | block context of X, Y, B (4 bits each: ac_context.h:101-143 BlockCtxMap::Context, from the
    // strategy's order bucket, the quant-field bucket and the DC bucket).
    static const uint8_t kOrderBucket[27] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
    static const uint8_t kLog2Cx[27] = {0, 0, 0, 0, 1, 2, 0, 1, 0, 2, 1, 2, 0, 0, 0, 0, 0, 0, 3, 2, 3, 4, 3, 4, 5, 4, 5};
    static const uint8_t kLog2Cy[27] = {0, 0, 0, 0, 1, 2, 1, 0, 2, 0, 2, 1, 0, 0, 0, 0, 0, 0, 3, 3, 2, 4, 4, 3, 5, 5, 4};
    std::vector<uint32_t> recs(size_t(d->num_blocks) + 16, 0);
    for (uint32_t i = 0; i < d->num_blocks; i++) {
      const JxlHipVarBlock& v = d->blocks[i];
      uint32_t qfi = 0;
      for (uint32_t t = 0; t < d->num_qf_thresholds; t++) qfi += v.qf > d->qf_thresholds[t];
      uint32_t rec = (v.bx & 31u) | ((v.by & 31u) ? 32u : 0u) | uint32_t(kLog2Cx[v.strategy]) << 6 | uint32_t(kLog2Cy[v.strategy]) << 9;
      for (uint32_t ch = 0; ch < 3; ch++) {
        const uint32_t bctx = d->block_ctx_lut[((ch * 13 + kOrderBucket[v.strategy]) * ep.nq + qfi) * ep.ndc + v.quant_dc_ctx];
        rec |= (bctx & 15u) << (12 + 4 * ch);
        // (chroma-subsampled frames: bit 24 + channel = the block is off the channel's grid and carries nothing for it,
        // bit 27 + channel = the channel's columns are half the frame's)
        const uint32_t hs = (cs >> (2 * ch)) & 1u, vs = (cs >> (2 * ch + 1)) & 1u;
        if ((v.bx & hs) | (v.by & vs)) rec |= 1u << (24 + ch);
        rec |= hs << (27 + ch);
      }
      recs[i] = rec;
    }
    if ((r = Upload(c, c->block_recs, recs.data(), recs.size() * 4))) return r;
    ep.block_recs = c->block_recs.as<uint32_t>();
  }
  lap("records");
  c->absent_blocks.clear();
  for (uint32_t g : c->absent_groups) {
    c->absent_blocks.push_back(d->group_block_begin[g]);
    c->absent_blocks.push_back(d->group_block_begin[g + 1] - d->group_block_begin[g]);
  }
  if (c->scan_order) {
    c->blocks_host.assign(d->blocks, d->blocks + d->num_blocks);
    c->gbb_host.assign(d->group_block_begin, d->group_block_begin + d->num_groups + 1);
    c->orders_host.assign(d->passes[0].orders, d->passes[0].orders + d->passes[0].orders_size);
    memcpy(c->order_offset_host, d->passes[0].order_offset, sizeof(c->order_offset_host));
  }
  if ((r = Upload(c, c->ep_dev, &ep, sizeof(ep)))) return r;
  c->generation++;

  jxlhip::TransformParams& tp = c->tp;
  memset(&tp, 0, sizeof(tp));
  tp.coeffs = c->coeffs.p;
  tp.coef_bits = d->coef_bits;
  tp.blocks = c->blocks.as<JxlHipVarBlock>();
  tp.dequant = c->dequant.as<float>();
  memcpy(tp.dq_offset, d->dequant_offset, sizeof(tp.dq_offset));
  memcpy(tp.dq_size, d->dequant_size, sizeof(tp.dq_size));
  for (int s = 0; s < 27; s++) {
    static const uint8_t qt[27] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};
    static const uint16_t areas[27] = {1, 1, 1, 1, 4, 16, 2, 2, 4, 4, 8, 8, 1, 1, 1, 1, 1, 1, 64, 32, 32, 256, 128, 128, 1024, 512, 512};
    if (c->list_count[s] == 0) continue;
    const uint32_t k = qt[s];
    if (d->dequant_size[k] != 64u * areas[s] || size_t(d->dequant_offset[k]) + 3 * size_t(d->dequant_size[k]) > d->dequant_floats)
      return JXLHIP_ERR_INVALID_ARGUMENT;
  }
  tp.cs = cs;
  tp.cs_out = c->plane[1].as<float>();  // (a YCbCr frame's colour stage is the generic writer: the second plane set exists)
  tp.dc = c->dc.as<float>();
  tp.ytox = c->ytox.as<int8_t>();
  tp.ytob = c->ytob.as<int8_t>();
  tp.basis_t = c->basis.as<float>();
  tp.basis_n = c->basis.as<float>() + 87381 + 3;
  tp.inv_global_scale = d->inv_global_scale;
  tp.x_dm = d->x_dm;
  tp.b_dm = d->b_dm;
  tp.color_scale = d->color_scale;
  tp.base_x = d->base_corr_x;
  tp.base_b = d->base_corr_b;
  memcpy(tp.biases, d->quant_biases, sizeof(tp.biases));
  tp.xb = c->xb; tp.yb = c->yb; tp.xg = c->xg; tp.xp = c->xp; tp.yp = c->yp;
  tp.out = PlaneHolder(c)->plane[0].as<float>();  // (a lender's buffer is checked again at launch: PrepareDownstream)
  tp.scratch = c->scratch.as<float>();
  tp.tlist = c->tlist.as<uint32_t>();
  tp.trecs = c->trecs.as<uint4>();
  memcpy(tp.list_begin, c->list_begin, sizeof(tp.list_begin));
  memcpy(tp.list_count, c->list_count, sizeof(tp.list_count));
  tp.scan_order = c->scan_order ? 1 : 0;
  tp.kend = c->kend.as<uint32_t>();
  tp.orders = c->pass_bufs[0].orders.as<uint16_t>();
  tp.dequant_scan = nullptr;
  if (c->scan_order) {
    // dequant tables in scan order for the kinds in use
    static const uint8_t kind_of[27] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};
    static const uint8_t bucket_of[27] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
    std::vector<float> scan(d->dequant_floats, 0.0f);
    for (int st = 0; st < 27; st++) {
      if (!c->list_count[st]) continue;
      const uint32_t kind = kind_of[st], size = d->dequant_size[kind];
      for (int ch = 0; ch < 3; ch++) {
        const uint32_t oo = d->passes[0].order_offset[bucket_of[st] * 3 + ch];
        if (size_t(oo) + size > d->passes[0].orders_size) return JXLHIP_ERR_INVALID_ARGUMENT;
        const uint16_t* order = d->passes[0].orders + oo;
        const float* m = d->dequant + d->dequant_offset[kind] + size_t(ch) * size;
        float* o = scan.data() + d->dequant_offset[kind] + size_t(ch) * size;
        for (uint32_t k = 0; k < size; k++) {
          if (order[k] >= size) return JXLHIP_ERR_INVALID_ARGUMENT;
          o[k] = m[order[k]];
        }
      }
    }
    if ((r = Upload(c, c->dequant_scan, scan.data(), scan.size() * 4))) return r;
    tp.dequant_scan = c->dequant_scan.as<float>();
  }
  memcpy(tp.order_offset, d->passes[0].order_offset, sizeof(tp.order_offset));

  jxlhip::FilterParams& fp = c->fp;
  memset(&fp, 0, sizeof(fp));
  fp.xs = c->xs; fp.ys = c->ys; fp.xp = c->xp; fp.yp = c->yp; fp.xb = c->xb;
  fp.y_begin = c->band_y0; fp.y_end = c->band_y1;
  fp.inv_sigma = c->inv_sigma.as<float>();
  for (int ch = 0; ch < 3; ch++) {
    const float w1 = d->gab_w[ch * 2], w2 = d->gab_w[ch * 2 + 1];
    const float mul = 1.0f / (1.0f + 4 * (w1 + w2));
    fp.gab_w[ch * 3] = mul;
    fp.gab_w[ch * 3 + 1] = w1 * mul;
    fp.gab_w[ch * 3 + 2] = w2 * mul;
    fp.ch_scale[ch] = d->epf_channel_scale[ch];
    fp.opsin_bias[ch] = d->opsin_bias[ch];
    fp.opsin_bias_cbrt[ch] = cbrtf(d->opsin_bias[ch]);
  }
  memcpy(fp.opsin_inv, d->opsin_inv, sizeof(fp.opsin_inv));
  fp.linear_output = d->linear_output;
  fp.rgb = OutIsRgb8(c) ? c->rgb.as<uint8_t>() : nullptr;
  fp.rgbf = !OutIsRgb8(c) && !c->color_out ? c->rgb.as<float>() : nullptr;
  c->epf_pass0 = d->epf_pass0_sigma_scale;
  c->epf_pass2 = d->epf_pass2_sigma_scale;
  c->epf_border = d->epf_border_sad_mul;
  c->ev_valid[0] = c->ev_valid[1] = c->ev_valid[2] = false;
  if ((r = BlobEnd(c, dev_smooth && !d->dc_device ? &smooth_p : nullptr, dev_sigma ? &sigma_p : nullptr, dev_dequant ? &dequant_p : nullptr))) return r;  // (records the staging block's event on the copy stream)
  // The tables are resident when the call returns (batch launches over this context run on other contexts' streams).
  // (Leaving the copies of many contexts in flight at once instead, ordered by events, made every later kernel of the
  // process ~1.35x slower on this runtime: scripts/async_probe.py.)
  HIP_TRY(WaitEvent(c->stage.done));
  lap("rest");
  if (prof) fprintf(stderr, "[upload] %s\n", prof_line.c_str());
  c->have_frame = true;
  c->mod.have = false;  // (the context now holds a VarDCT frame, not the Modular one it may have held before)
  return 0;
}

}  // extern "C"

template <typename CoefT>
static int LaunchEntropy(JxlHipContext* c) {
  const dim3 grid(c->ng), block(64);
  const bool use_uni = EntropyKernelChoice() != 0;
  if (c->generic_codec) {
    hipLaunchKernelGGL(jxlhip::k_entropy_generic<CoefT>, grid, block, 3072, c->stream, c->ep);
  } else if (c->alias_lds && use_uni) {
    // scalar-form kernel: alias tables, context map and uint configs shared in LDS by the waves of a workgroup
    const size_t shared = size_t(c->ep.lds_ctx_bytes) + c->ep.lds_alias_bytes + 1024;
    jxlhip::EntropyBatch b{c->ep_dev.as<jxlhip::EntropyParams>(), nullptr};
    if (c->ep.num_hist == 1) {
      constexpr int WPG = kEntropyWPG;
      auto k = jxlhip::k_entropy_uni<CoefT, WPG>;
      const size_t lds = shared + WPG * 5120;
      if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
      hipLaunchKernelGGL(k, dim3((c->ng + WPG - 1) / WPG), dim3(64 * WPG), lds, c->stream, b);
    } else {
      auto k = jxlhip::k_entropy_uni<CoefT, 1>;
      const size_t lds = shared + 5120;
      if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
      hipLaunchKernelGGL(k, grid, block, lds, c->stream, b);
    }
  } else if (c->alias_lds) {
    auto k = jxlhip::k_entropy_ans<CoefT, true>;
    if (c->lds_entropy > 48 * 1024)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(c->lds_entropy)));
    hipLaunchKernelGGL(k, grid, block, c->lds_entropy, c->stream, c->ep);
  } else {
    auto k = jxlhip::k_entropy_ans<CoefT, false>;
    hipLaunchKernelGGL(k, grid, block, c->lds_entropy, c->stream, c->ep);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

static int RunEntropySingle(JxlHipContext* c) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return JXLHIP_ERR_NO_FRAME;
  HIP_TRY(hipSetDevice(c->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
  }
  HIP_TRY(hipEventRecord(c->ev[0], c->stream));
  HIP_TRY(hipMemsetAsync(c->errors.p, 0, size_t(c->ng) * 4, c->stream));
  int r = c->coef_bits == 16 ? LaunchEntropy<int16_t>(c) : LaunchEntropy<int32_t>(c);
  if (r) return r;
  HIP_TRY(hipEventRecord(c->ev[1], c->stream));
  c->ev_valid[0] = true;
  return 0;
}

// ---- transform + filter launches, batched over the frames of a set ---------------------------------------------------
// JXLHIP_IDCT_MATRIX=1 selects the matrix-form column kernel (k_idct_cols) instead of the fast in-register form.
static bool IdctMatrixForm() { return EnvInt("JXLHIP_IDCT_MATRIX", 0) != 0; }
// Varblocks a workgroup of the strategy's kernel handles.
static uint32_t BlocksPerWG(int s) {
  static const uint8_t cx[27] = {1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32};
  static const uint8_t cy[27] = {1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16};
  if (s == 0 || (s >= 4 && s <= 11)) {
    if (IdctMatrixForm()) return jxlhip::kIdctColsThreads / (cx[s] * 8);  // k_idct_cols: 8 * covered_x threads per varblock
    return jxlhip::IdctFastThreads(cx[s], cy[s]) / ((cx[s] > cy[s] ? cx[s] : cy[s]) * 8);  // k_idct_fast: max(rows, columns) threads
  }
  if (s >= 18 && s <= 20) return 1;                               // k_dct 64-class
  return 4;                                                       // k_special
}

template <typename CoefT, int CX, int CY, bool CS = false>
static int LaunchIdctFast(JxlHipContext* c0, int s) {
  constexpr int C = CX * 8, R = CY * 8, TB = R > C ? R : C, GROUPS = jxlhip::IdctFastThreads(CX, CY) / TB;
  constexpr size_t lds = size_t(GROUPS) * R * (C + 1) * sizeof(float);
  const int li = CS ? 21 : s;  // (the work list; the strategy stays s)
  hipLaunchKernelGGL((jxlhip::k_idct_fast<CoefT, CX, CY, CS>), dim3(c0->desc_count[li]), dim3(jxlhip::IdctFastThreads(CX, CY)), lds, c0->stream,
                     c0->tb_params.as<jxlhip::TransformParams>(), c0->tb_desc.as<uint2>() + c0->desc_begin[li], uint32_t(s));
  return 0;
}

template <typename CoefT, int CX, int CY>
static int LaunchIdctCols(JxlHipContext* c0, int s) {
  if (!IdctMatrixForm()) return LaunchIdctFast<CoefT, CX, CY>(c0, s);
  constexpr int C = CX * 8, R = CY * 8, SIZE = CX * CY * 64, GROUPS = jxlhip::kIdctColsThreads / C;
  constexpr size_t lds = (size_t(GROUPS) * (2 * SIZE + 4) + R * R) * sizeof(float);
  auto k = jxlhip::k_idct_cols<CoefT, CX, CY>;
  if (lds > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
  hipLaunchKernelGGL(k, dim3(c0->desc_count[s]), dim3(jxlhip::kIdctColsThreads), lds, c0->stream,
                     c0->tb_params.as<jxlhip::TransformParams>(), c0->tb_desc.as<uint2>() + c0->desc_begin[s], uint32_t(s));
  return 0;
}

template <typename CoefT, int CX, int CY>
static int LaunchDct(JxlHipContext* c0, int s) {
  constexpr int SIZE = CX * CY * 64;
  constexpr size_t lds = size_t(3) * SIZE * sizeof(float);
  auto k = jxlhip::k_dct<CoefT, CX, CY>;
  if (lds > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
  hipLaunchKernelGGL(k, dim3(c0->desc_count[s]), dim3(256), lds, c0->stream, c0->tb_params.as<jxlhip::TransformParams>(),
                     c0->tb_desc.as<uint2>() + c0->desc_begin[s], uint32_t(s));
  return 0;
}

template <typename CoefT>
static int LaunchTransforms(JxlHipContext* c0) {
  for (int s = 0; s < 21; s++) {
    if (!c0->desc_count[s]) continue;
    int e = 0;
    switch (s) {
      case 0: e = LaunchIdctCols<CoefT, 1, 1>(c0, s); break;
      case 4: e = LaunchIdctCols<CoefT, 2, 2>(c0, s); break;
      case 5: e = LaunchIdctCols<CoefT, 4, 4>(c0, s); break;
      case 6: e = LaunchIdctCols<CoefT, 1, 2>(c0, s); break;
      case 7: e = LaunchIdctCols<CoefT, 2, 1>(c0, s); break;
      case 8: e = LaunchIdctCols<CoefT, 1, 4>(c0, s); break;
      case 9: e = LaunchIdctCols<CoefT, 4, 1>(c0, s); break;
      case 10: e = LaunchIdctCols<CoefT, 2, 4>(c0, s); break;
      case 11: e = LaunchIdctCols<CoefT, 4, 2>(c0, s); break;
      // the 64-class: the fast (recursive even / odd) form too, one varblock per wave; JXLHIP_IDCT_MATRIX keeps the
      // matrix-form kernel (the plain statement of the transform) reachable
      case 18: e = IdctMatrixForm() ? LaunchDct<CoefT, 8, 8>(c0, s) : LaunchIdctFast<CoefT, 8, 8>(c0, s); break;
      case 19: e = IdctMatrixForm() ? LaunchDct<CoefT, 4, 8>(c0, s) : LaunchIdctFast<CoefT, 4, 8>(c0, s); break;
      case 20: e = IdctMatrixForm() ? LaunchDct<CoefT, 8, 4>(c0, s) : LaunchIdctFast<CoefT, 8, 4>(c0, s); break;
      default:
        hipLaunchKernelGGL((jxlhip::k_special<CoefT>), dim3(c0->desc_count[s]), dim3(256), 0, c0->stream,
                           c0->tb_params.as<jxlhip::TransformParams>(), c0->tb_desc.as<uint2>() + c0->desc_begin[s], uint32_t(s));
    }
    if (e) return e;
    HIP_TRY(hipGetLastError());
  }
  if (c0->desc_count[21]) {  // the 8x8 DCT varblocks of chroma-subsampled frames
    int e = LaunchIdctFast<CoefT, 1, 1, true>(c0, 0);
    if (e) return e;
    HIP_TRY(hipGetLastError());
  }
  // ... whose subsampled channels then go back to the frame's resolution, in front of the filters
  for (const JxlHipContext* c : c0->db_ctxs) {
    if (!c->cs) continue;
    for (uint32_t ch = 0; ch < 3; ch++) {
      jxlhip::ChromaUpParams up;
      up.hs = (c->cs >> (2 * ch)) & 1u;
      up.vs = (c->cs >> (2 * ch + 1)) & 1u;
      if (!(up.hs | up.vs)) continue;
      const size_t plane = size_t(c->xp) * c->yp;
      up.src = c->plane[1].as<float>() + ch * plane;
      up.dst = PlaneHolder(c)->plane[0].as<float>() + ch * plane;
      up.xs = c->xs;
      up.ys = c->ys;
      up.xp = c->xp;
      up.y0 = c->ext_y0;
      up.y1 = c->ext_y1;
      if (up.y1 <= up.y0) continue;
      hipLaunchKernelGGL(jxlhip::k_chroma_upsample, dim3((up.xs + 63) / 64, (up.y1 - up.y0 + 3) / 4), dim3(256), 0, c0->stream, up);
    }
    HIP_TRY(hipGetLastError());
  }
  // 128/256-class transforms go through per-frame global scratch: one frame at a time (rare)
  for (const JxlHipContext* c : c0->db_ctxs)
    for (int s = 21; s < 27; s++) {
      const uint32_t n = c->list_count[s];
      if (!n) continue;
      const uint32_t* list = c->tlist.as<uint32_t>() + c->list_begin[s];
      for (uint32_t i = 0; i < n; i += 32) {
        const uint32_t m = n - i < 32 ? n - i : 32;
        hipLaunchKernelGGL((jxlhip::k_dct_big<CoefT>), dim3(m * 3), dim3(256), 0, c0->stream, c->tp, list + i, m, uint32_t(s));
      }
      HIP_TRY(hipGetLastError());
    }
  return 0;
}

static void FillFusedParams(const JxlHipContext* c, jxlhip::FusedFilterParams* p) {
  p->debug = uint32_t(EnvInt("JXLHIP_FILTER_DEBUG", 0));
  p->f = c->fp;
  p->f.in = PlaneHolder(c)->plane[0].as<float>();
  p->f.out = nullptr;
  for (int stage = 0; stage < 3; stage++) {
    const float scale = stage == 0 ? c->epf_pass0 : stage == 2 ? c->epf_pass2 : 1.0f;
    p->sm[stage] = stage == 1 ? 1.65f : float(scale * 1.65);
    p->bsm[stage] = p->sm[stage] * c->epf_border;
  }
  p->filtered = (c->keep_filtered || c->ups != 1 || c->color_out) ? c->plane[1].as<float>() : nullptr;
  if (c->ups != 1 || c->color_out) {
    p->f.rgb = nullptr;
    p->f.rgbf = nullptr;
  }  // the upsampling / colour kernel produces the pixels
}
static int FilterKey(const JxlHipContext* c) {
  const int epf = c->epf_iters < 0 ? 0 : (c->epf_iters > 3 ? 3 : c->epf_iters);
  return (c->gab ? 4 : 0) + epf;
}
// Builds (or re-uses) the description of a set of frames for the batched transform and filter launches.
static int PrepareDownstream(JxlHipContext* c0, JxlHipContext* const* ctxs, size_t n) {
  bool same = c0->db_ctxs.size() == n;
  for (size_t i = 0; same && i < n; i++) same = c0->db_ctxs[i] == ctxs[i] && c0->db_gens[i] == ctxs[i]->generation;
  if (same) return 0;
  std::vector<jxlhip::TransformParams> tparams(n);
  for (size_t i = 0; i < n; i++) {
    tparams[i] = ctxs[i]->tp;
    const JxlHipContext* h = PlaneHolder(ctxs[i]);
    if (!h->plane[0].p || h->plane[0].cap < ctxs[i]->plane_bytes) return JXLHIP_ERR_INVALID_ARGUMENT;  // lender too small
    tparams[i].out = h->plane[0].as<float>();
    if (ctxs[i]->cs && (!ctxs[i]->plane[1].p || ctxs[i]->plane[1].cap < ctxs[i]->plane_bytes)) return JXLHIP_ERR_INVALID_ARGUMENT;
    tparams[i].cs_out = ctxs[i]->plane[1].as<float>();
  }
  std::vector<uint2> desc;
  for (int s = 0; s < 22; s++) {  // (21: the 8x8 DCT varblocks of chroma-subsampled frames, which list 0 leaves out)
    c0->desc_begin[s] = uint32_t(desc.size());
    const int st = s == 21 ? 0 : s;
    const uint32_t bpw = BlocksPerWG(st);
    for (size_t i = 0; i < n; i++) {
      if (st == 0 && (ctxs[i]->cs != 0) != (s == 21)) continue;
      for (uint32_t j = 0; j < ctxs[i]->list_count[st]; j += bpw) desc.push_back(make_uint2(uint32_t(i), j));
    }
    c0->desc_count[s] = uint32_t(desc.size()) - c0->desc_begin[s];
  }
  // filter launches: frames grouped by (gaborish, epf iterations), one grid z slice per frame
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return FilterKey(ctxs[a]) < FilterKey(ctxs[b]); });
  std::vector<jxlhip::FusedFilterParams> fparams(n);
  c0->fgroups.clear();
  for (size_t j = 0; j < n; j++) {
    const JxlHipContext* c = ctxs[order[j]];
    FillFusedParams(c, &fparams[j]);
    const int key = FilterKey(c);
    const uint32_t tx = (c->xs + jxlhip::kFusedTW - 1) / jxlhip::kFusedTW;
    const uint32_t ty = (c->band_y1 - c->band_y0 + jxlhip::kFusedTH - 1) / jxlhip::kFusedTH;
    if (c0->fgroups.empty() || c0->fgroups.back().key != key) c0->fgroups.push_back({key, uint32_t(j), 0, 0, 0, true});
    JxlHipContext::FilterGroup& g = c0->fgroups.back();
    g.count++;
    g.u8srgb = g.u8srgb && fparams[j].f.rgb && !fparams[j].f.rgbf && !fparams[j].filtered && !fparams[j].f.linear_output;
    g.tiles_x = tx > g.tiles_x ? tx : g.tiles_x;
    g.tiles_y = ty > g.tiles_y ? ty : g.tiles_y;
  }
  int r;
  if ((r = c0->tb_params.Ensure(n * sizeof(tparams[0])))) return r;
  if ((r = c0->tb_desc.Ensure((desc.size() + 1) * sizeof(uint2)))) return r;
  if ((r = c0->fb_params.Ensure(n * sizeof(fparams[0])))) return r;
  {
    ParamCopy pc;
    if ((r = pc.Begin(c0))) return r;
    if ((r = pc.Add(c0->tb_params.p, tparams.data(), n * sizeof(tparams[0])))) return r;
    if ((r = pc.Add(c0->tb_desc.p, desc.data(), desc.size() * sizeof(uint2)))) return r;
    if ((r = pc.Add(c0->fb_params.p, fparams.data(), n * sizeof(fparams[0])))) return r;
    if ((r = pc.End())) return r;
  }
  c0->db_ctxs.assign(ctxs, ctxs + n);
  c0->db_gens.resize(n);
  for (size_t i = 0; i < n; i++) c0->db_gens[i] = ctxs[i]->generation;
  return 0;
}

template <bool GAB, int EPF>
static int LaunchFused(JxlHipContext* c0, const JxlHipContext::FilterGroup& g) {
  auto k = jxlhip::k_filter_fused<GAB, EPF>;
  constexpr size_t lds = jxlhip::FusedLdsBytes(GAB, EPF);
  if (lds > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
  for (uint32_t z = 0; z < g.count; z += 65535) {  // grid z limit
    const uint32_t zn = g.count - z < 65535 ? g.count - z : 65535;
    hipLaunchKernelGGL(k, dim3(g.tiles_x, g.tiles_y, zn), dim3(jxlhip::kFusedThreads), lds, c0->fstream,
                       c0->fb_params.as<jxlhip::FusedFilterParams>() + g.first + z);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// Gaborish + EPF1 (the d1.0 configuration) or + EPF1 + EPF2 (`epf` = 2): the row-streaming kernel, no LDS.
static int LaunchFilterRows(JxlHipContext* c0, const JxlHipContext::FilterGroup& g, int epf = 1, bool gab = true) {
  const uint32_t cols = g.tiles_x * jxlhip::kFusedTW, rows = g.tiles_y * jxlhip::kFusedTH;  // upper bounds of the group
  const bool one_px = epf == 1 && gab && EnvInt("JXLHIP_FILTER_ROWS1", 0) != 0;  // measurement aid: the one-column-per-lane form
  const uint32_t per_wave = one_px ? jxlhip::kRowsLanes : jxlhip::kRows2Cols;
  const uint32_t gx = ((cols + per_wave - 1) / per_wave + jxlhip::kRowsWaves - 1) / jxlhip::kRowsWaves;
  // strip height: every strip re-reads and re-filters 6 halo rows, so the taller the better as long as the launch still
  // has several waves for each of the chip's 1024 SIMDs
  uint32_t strip = jxlhip::kRowsStrip;
  if (!one_px) {
    const uint32_t want = uint32_t(EnvInt("JXLHIP_FILTER_STRIP", 0));
    for (uint32_t s2 = want ? 256 : 128; s2 > strip; s2 >>= 1)  // (256 measured no better than 64: 0.0374 / 0.0361 / 0.0374 ms per 4K frame)
      if (want ? s2 == want : uint64_t(gx) * jxlhip::kRowsWaves * ((rows + s2 - 1) / s2) * g.count >= 6 * 1024) {
        strip = s2;
        break;
      }
  }
  const uint32_t gy = (rows + strip - 1) / strip;
  for (uint32_t z = 0; z < g.count; z += 65535) {  // grid z limit
    const uint32_t zn = g.count - z < 65535 ? g.count - z : 65535;
    const jxlhip::FusedFilterParams* fp = c0->fb_params.as<jxlhip::FusedFilterParams>() + g.first + z;
    if (one_px)
      hipLaunchKernelGGL(jxlhip::k_filter_rows, dim3(gx, gy, zn), dim3(64 * jxlhip::kRowsWaves), 0, c0->fstream, fp);
    else {
      typedef void (*RowsKernel)(const jxlhip::FusedFilterParams*, int);
      const RowsKernel k1t = jxlhip::k_filter_rows2<true, 1>, k1f = jxlhip::k_filter_rows2<false, 1>;
      const RowsKernel k2t = jxlhip::k_filter_rows2<true, 2>, k2f = jxlhip::k_filter_rows2<false, 2>;
      const RowsKernel n1t = jxlhip::k_filter_rows2<true, 1, false>, n1f = jxlhip::k_filter_rows2<false, 1, false>;
      const RowsKernel n2t = jxlhip::k_filter_rows2<true, 2, false>, n2f = jxlhip::k_filter_rows2<false, 2, false>;
      const RowsKernel k = gab ? (epf == 2 ? (g.u8srgb ? k2t : k2f) : (g.u8srgb ? k1t : k1f))
                               : (epf == 2 ? (g.u8srgb ? n2t : n2f) : (g.u8srgb ? n1t : n1f));
      hipLaunchKernelGGL(k, dim3(gx, gy, zn), dim3(64 * jxlhip::kRowsWaves), 0, c0->fstream, fp, int(strip));
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// Validates a set for a batched downstream call and orders the launch stream after everything its frames wait for.
static int EnsureOwnStream(JxlHipContext* c) {
  if (c->owns_stream) return 0;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));  // (what the context queued on the shared stream so far)
  hipStream_t st = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  c->stream = st;
  c->owns_stream = true;
  return 0;
}

// A batched entropy launch must get the machine FIRST. Its workgroups are few (one per frame), long-lived and large in LDS
// (37 - 54 KB: 640 of them take 84 % of the chip's LDS); when a transform launch of another frame set is already streaming
// its small workgroups through the CUs, the entropy workgroups find no CU with that much LDS free and trickle in only as the
// other launch drains: 133 ms per launch instead of 62 ms (scripts/r03_interference.py; 80 ms behind a filter launch,
// which holds registers, not LDS). Arriving first it loses nothing to the same neighbours (62 ms). So: every workgroup of a
// batched entropy launch counts itself in when it starts, and the batched transform / filter launches enqueued after it,
// on whatever stream, wait (hipStreamWaitValue64, a wait packet in their own queue) until that count says the whole
// entropy launch is resident. The counter only grows: a launch's target is the running total after it.
struct EntropyGate {
  unsigned long long* counter = nullptr;  // device memory (hipMallocSignalMemory)
  unsigned long long total = 0;           // workgroups of every launch so far
  unsigned long long target = 0;          // count at which the latest launch is resident
  unsigned cus = 256;                     // compute units and LDS bytes per unit of the device (read with the counter)
  size_t lds_per_cu = 160 * 1024;
  bool tried = false;
};
static std::mutex g_gate_mu;
static EntropyGate g_gate[16];
// True when this process's kernels may be run one at a time (runtime debug switches, a single hardware queue, counter
// collection by a profiler): a device-side wait for ANOTHER launch's workgroups could then wait for a kernel that is not
// allowed to start, so the gate is never used there. Read once.
static bool RuntimeMaySerialise() {
  static const bool v = [] {
    auto on = [](const char* n) {
      const char* e = getenv(n);
      return e && *e && !(e[0] == '0' && !e[1]);
    };
    if (on("AMD_SERIALIZE_KERNEL") || on("AMD_SERIALIZE_COPY") || on("HIP_LAUNCH_BLOCKING") || on("ROCPROF_COUNTER_COLLECTION") ||
        on("HSA_ENABLE_DEBUG") || on("ROCM_DEBUG_AGENT") || on("HIP_ENABLE_DEFERRED_LOADING_DEBUG"))
      return true;
    if (const char* q = getenv("GPU_MAX_HW_QUEUES"))
      if (*q && atoi(q) < 2) return true;
    if (const char* t = getenv("HSA_TOOLS_LIB"))  // a debugger / tracer intercepting the queues (rocgdb, roctracer)
      if (*t && (strstr(t, "debug") || strstr(t, "rocgdb"))) return true;
    return false;
  }();
  return v;
}
// the counter the next batched entropy launch of `device` counts into (NULL: gating unavailable, not asked for by the
// caller -- option "entropy_gate" of the launch's first context, off by default -- or unsafe in this process)
static unsigned long long* GateCounter(const JxlHipContext* c0) {
  const int device = c0->device;
  if (device < 0 || device >= 16 || !c0->entropy_gate || !EnvInt("JXLHIP_ENTROPY_GATE", 1) || RuntimeMaySerialise()) return nullptr;
  std::lock_guard<std::mutex> lk(g_gate_mu);
  EntropyGate& g = g_gate[device];
  if (!g.tried) {
    g.tried = true;
    int can = 0;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device) == hipSuccess && can) {
      void* p = nullptr;
      if (hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) == hipSuccess && hipMemset(p, 0, 8) == hipSuccess)
        g.counter = static_cast<unsigned long long*>(p);
    }
    (void)hipGetLastError();
    int cus = 0, lds = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) g.cus = unsigned(cus);
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) == hipSuccess && lds > 0) g.lds_per_cu = size_t(lds);
    (void)hipGetLastError();
  }
  return g.counter;
}
static void GateAdvance(int device, unsigned long long workgroups, unsigned long long resident_cap) {
  std::lock_guard<std::mutex> lk(g_gate_mu);
  EntropyGate& g = g_gate[device];
  g.target = g.total + (workgroups < resident_cap ? workgroups : resident_cap);
  g.total += workgroups;
}
static int GateWait(int device, hipStream_t stream) {
  if (device < 0 || device >= 16) return 0;
  unsigned long long* counter;
  unsigned long long target;
  {
    std::lock_guard<std::mutex> lk(g_gate_mu);
    counter = g_gate[device].counter;
    target = g_gate[device].target;
  }
  if (!counter || !target) return 0;
  HIP_TRY(hipStreamWaitValue64(stream, counter, target, hipStreamWaitValueGte, ~0ull));
  return 0;
}

static int BeginDownstreamBatch(JxlHipContext* const* ctxs, size_t n, bool filter_stage = false) {
  if (!ctxs || !n) return JXLHIP_ERR_INVALID_ARGUMENT;
  JxlHipContext* c0 = ctxs[0];
  for (size_t i = 0; i < n; i++) {
    if (!ctxs[i]) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (!ctxs[i]->have_frame) return JXLHIP_ERR_NO_FRAME;
    if (ctxs[i]->device != c0->device || ctxs[i]->coef_bits != c0->coef_bits) return JXLHIP_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(c0->device));
  if (n > 1) {
    const int r = EnsureOwnStream(c0);
    if (r) return r;
  }
  hipStream_t ls = c0->stream;  // launch stream
  if (filter_stage && c0->filter_async) {
    if (!c0->stream2) {
      HIP_TRY(hipStreamCreateWithFlags(&c0->stream2, hipStreamNonBlocking));
      if (!c0->fork_event) HIP_TRY(hipEventCreateWithFlags(&c0->fork_event, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c0->filter_done, hipEventDisableTiming));
    }
    ls = c0->stream2;
    if (c0->transform_done_valid) {
      // after the set's transform launch, NOT after whatever the first stream has been given since: the caller may
      // already have queued the set's next entropy launch there (which must reach the machine before this launch does:
      // see EntropyGate), and the filter stage has nothing to do with it
      HIP_TRY(hipStreamWaitEvent(ls, c0->transform_done, 0));
    } else {
      HIP_TRY(hipEventRecord(c0->fork_event, c0->stream));  // after everything the first stream holds
      HIP_TRY(hipStreamWaitEvent(ls, c0->fork_event, 0));
    }
  }
  c0->fstream = ls;
  if (n > 1 && c0->entropy_gate && !RuntimeMaySerialise()) {  // (see EntropyGate)
    const int r = GateWait(c0->device, ls);
    if (r) return r;
  }
  hipEvent_t waited = nullptr;
  for (size_t i = 0; i < n; i++)
    if (ctxs[i]->pending_wait) {
      if (ctxs[i]->pending_wait != waited) HIP_TRY(hipStreamWaitEvent(ls, ctxs[i]->pending_wait, 0));
      waited = ctxs[i]->pending_wait;
      ctxs[i]->pending_wait = nullptr;
    }
  // planes / pixels still in use by an asynchronous filter launch: wait for it, once per event
  waited = nullptr;
  for (size_t i = 0; i < n; i++)
    if (ctxs[i]->filter_wait) {
      if (ctxs[i]->filter_wait != waited) HIP_TRY(hipStreamWaitEvent(ls, ctxs[i]->filter_wait, 0));
      waited = ctxs[i]->filter_wait;
      ctxs[i]->filter_wait = nullptr;
    }
  // plane buffers last read by a filter launch on another stream (shared planes): wait for that launch, once per event
  waited = nullptr;
  for (size_t i = 0; i < n; i++) {
    JxlHipContext* h = PlaneHolder(ctxs[i]);
    if (h->planes_event && h->planes_stream != ls && h->planes_event != waited) {
      HIP_TRY(hipStreamWaitEvent(ls, h->planes_event, 0));
      waited = h->planes_event;
    }
  }
  return PrepareDownstream(c0, ctxs, n);
}
static int EndDownstreamBatch(JxlHipContext* const* ctxs, size_t n, bool filter_stage = false) {
  JxlHipContext* c0 = ctxs[0];
  bool shared = false;
  for (size_t i = 0; i < n && !shared; i++) shared = ctxs[i]->plane_lender != nullptr || ctxs[i]->planes_event != nullptr;
  hipEvent_t done = nullptr;
  if (filter_stage && c0->fstream != c0->stream) {  // asynchronous filter launch: every context of it remembers its end
    HIP_TRY(hipEventRecord(c0->filter_done, c0->fstream));
    done = c0->filter_done;
    for (size_t i = 0; i < n; i++) ctxs[i]->filter_wait = done;
  } else if (n > 1 || shared) {
    if (!c0->down_done) HIP_TRY(hipEventCreateWithFlags(&c0->down_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c0->down_done, c0->stream));
    done = c0->down_done;
    for (size_t i = 1; i < n; i++) ctxs[i]->pending_wait = done;
  }
  if (!filter_stage && c0->filter_async) {
    if (!c0->transform_done) HIP_TRY(hipEventCreateWithFlags(&c0->transform_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c0->transform_done, c0->stream));
    c0->transform_done_valid = true;
  }
  if (shared && filter_stage)
    for (size_t i = 0; i < n; i++) {
      JxlHipContext* h = PlaneHolder(ctxs[i]);
      h->planes_event = done;
      h->planes_stream = c0->fstream;
    }
  return 0;
}

template <typename CoefT, int WPG, bool AIDS, bool GALIAS, bool PREFIX = false, bool ASMT = false, bool A6 = false>
static int LaunchEntropyLanesW(JxlHipContext* c0);
template <typename CoefT, bool GALIAS, bool ASMT>
static int LaunchEntropyLanesA(JxlHipContext* c0) {
  if (EnvInt("JXLHIP_LANES_DEBUG", 0) || EnvInt("JXLHIP_LANES_PROF", 0))  // measurement aids: instrumented build of the kernel
    return c0->batch_wpg == 1 ? LaunchEntropyLanesW<CoefT, 1, true, GALIAS, false, ASMT>(c0)
                              : (c0->batch_wpg == 2 ? LaunchEntropyLanesW<CoefT, 2, true, GALIAS, false, ASMT>(c0)
                                                    : LaunchEntropyLanesW<CoefT, 4, true, GALIAS, false, ASMT>(c0));
  return c0->batch_wpg == 1 ? LaunchEntropyLanesW<CoefT, 1, false, GALIAS, false, ASMT>(c0)
                            : (c0->batch_wpg == 2 ? LaunchEntropyLanesW<CoefT, 2, false, GALIAS, false, ASMT>(c0)
                                                  : LaunchEntropyLanesW<CoefT, 4, false, GALIAS, false, ASMT>(c0));
}
template <typename CoefT, bool GALIAS>
static int LaunchEntropyLanesG(JxlHipContext* c0) {
  // the hand-written trip (jxl_hip_lanes_trip.inc) serves LDS alias tables with int16 coefficients; JXLHIP_LANES_CPP=1 keeps
  // the C++ trip for that form too (the two are held against each other by tests/test_gpu_parity.py)
  if constexpr (!GALIAS && sizeof(CoefT) == 2) {
    if (!EnvInt("JXLHIP_LANES_CPP", 0)) return LaunchEntropyLanesA<CoefT, GALIAS, true>(c0);
  }
  return LaunchEntropyLanesA<CoefT, GALIAS, false>(c0);
}
template <typename CoefT>
static int LaunchEntropyLanes(JxlHipContext* c0) {
  if (c0->batch_prefix)  // (no instrumented build of the prefix form)
    return c0->batch_wpg == 1 ? LaunchEntropyLanesW<CoefT, 1, false, true, true>(c0)
                              : (c0->batch_wpg == 2 ? LaunchEntropyLanesW<CoefT, 2, false, true, true>(c0)
                                                    : LaunchEntropyLanesW<CoefT, 4, false, true, true>(c0));
  if constexpr (sizeof(CoefT) == 2) {
    if (c0->batch_a6)  // (PrepareBatch: one wave per workgroup, int16 coefficients; no instrumented build)
      return EnvInt("JXLHIP_LANES_CPP", 0) ? LaunchEntropyLanesW<CoefT, 1, false, false, false, false, true>(c0)
                                           : LaunchEntropyLanesW<CoefT, 1, false, false, false, true, true>(c0);
  }
  return c0->batch_galias ? LaunchEntropyLanesG<CoefT, true>(c0) : LaunchEntropyLanesG<CoefT, false>(c0);
}
template <typename CoefT, int WPG, bool AIDS, bool GALIAS, bool PREFIX, bool ASMT, bool A6>
static int LaunchEntropyLanesW(JxlHipContext* c0) {
  auto k = jxlhip::k_entropy_lanes<CoefT, WPG, AIDS, GALIAS, PREFIX, ASMT, A6>;
  if (c0->batch_lds > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(c0->batch_lds)));
  jxlhip::EntropyLaneBatch b;
  b.params = c0->batch_params.as<jxlhip::EntropyParams>();
  uint8_t* blob = c0->batch_lanes.as<uint8_t>();
  b.wg_unit = c0->batch_map.as<uint32_t>();
  b.list = reinterpret_cast<const uint32_t*>(blob);
  b.units = reinterpret_cast<const uint4*>(blob + c0->batch_off_units);
  b.queue = reinterpret_cast<uint32_t*>(blob + c0->batch_off_queue);
  b.wave_lanes = blob + c0->batch_off_wave_lanes;
  HIP_TRY(hipMemsetAsync(b.queue, 0, c0->batch_units * 4, c0->stream));
  b.wait_shift = c0->batch_wait_shift;
  {
    // rounds (hot trips + transition pass) per refill round; the hand-written loop runs several groups of trips per round
    int every = EnvInt("JXLHIP_REFILL_EVERY", ASMT ? 1 : int(jxlhip::kLanesRefillEvery));
    if (every < 1 || (every & (every - 1))) every = int(jxlhip::kLanesRefillEvery);
    b.refill_mask = uint32_t(every - 1);
  }
  b.prio = uint32_t(EnvInt("JXLHIP_LANES_PRIO", 0));
  b.extra_pass_min = uint32_t(EnvInt("JXLHIP_EXTRA_PASS_MIN", 8));
  if (b.extra_pass_min < 1) b.extra_pass_min = 1;
  b.wave_log_ls = c0->batch_wave_ls.as<uint8_t>();
  b.debug = uint32_t(EnvInt("JXLHIP_LANES_DEBUG", 0));
  b.prof = nullptr;
  b.started = c0->batch_ctxs.size() > 1 ? GateCounter(c0) : nullptr;
  const bool prof = EnvInt("JXLHIP_LANES_PROF", 0) != 0;  // debugging aid: per-wave cycle split, printed to stderr
  const size_t nwaves = size_t(c0->batch_wgs) * WPG;
  if (prof) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b.prof), nwaves * 64));
    HIP_TRY(hipMemsetAsync(b.prof, 0, nwaves * 64, c0->stream));
  }
  hipLaunchKernelGGL(k, dim3(c0->batch_wgs), dim3(64 * WPG), c0->batch_lds, c0->stream, b);
  HIP_TRY(hipGetLastError());
  if (b.started) {
    // "resident" = as many of its workgroups as the chip's LDS holds at once, less a margin (the rest start as the first
    // ones end, and are not waited for)
    unsigned long long cus, lds_cu;
    {
      std::lock_guard<std::mutex> lk(g_gate_mu);
      cus = g_gate[c0->device].cus;
      lds_cu = g_gate[c0->device].lds_per_cu;
    }
    const unsigned long long per_cu = c0->batch_lds ? lds_cu / c0->batch_lds : 8, cap = cus * (per_cu ? per_cu : 1) * 3 / 4;
    GateAdvance(c0->device, c0->batch_wgs, cap);
  }
  if (prof) {
    std::vector<unsigned long long> h(nwaves * 8);
    HIP_TRY(hipStreamSynchronize(c0->stream));
    HIP_TRY(hipMemcpy(h.data(), b.prof, nwaves * 64, hipMemcpyDeviceToHost));
    (void)hipFree(b.prof);
    // per wave: {cycles total, cycles in service, service calls, hot trips, lane-trips taken, service: waiting for the
    // previous phase's DMA / flush + DMA requests / transitions}; one JSON line: the longest wave and the mean over waves
    unsigned long long mx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t used = 0;
    for (size_t w = 0; w < nwaves; w++) {
      if (!h[w * 8 + 3] && !h[w * 8 + 2]) continue;
      used++;
      if (h[w * 8] > mx[0]) for (int j = 0; j < 8; j++) mx[j] = h[w * 8 + j];
      for (int j = 0; j < 8; j++) sum[j] += double(h[w * 8 + j]);
    }
    const char* names[8] = {"cycles", "refill_cycles", "passes", "trips", "lane_trips", "trip_cycles", "pass_lanes_served", "pass_cycles"};
    std::string line = "{\"waves\": " + std::to_string(used) + ", \"longest\": {";
    for (int j = 0; j < 8; j++) line += std::string(j ? ", " : "") + "\"" + names[j] + "\": " + std::to_string(mx[j]);
    line += "}, \"mean\": {";
    for (int j = 0; j < 8; j++) line += std::string(j ? ", " : "") + "\"" + names[j] + "\": " + std::to_string((unsigned long long)(sum[j] / (used ? used : 1)));
    line += "}}";
    fprintf(stderr, "[lanes prof] %s\n", line.c_str());
    if (b.debug & 16)  // (measurement aid: per wave: cycles, trips, HW_ID fields wave / simd / cu / sh / se, XCC)
      for (size_t w = 0; w < nwaves && w < 64; w++) {
        const unsigned long long hw = h[w * 8 + 6];
        fprintf(stderr, "[lanes wave] %zu cycles %llu trips %llu trip_cycles %llu pass_cycles %llu passes %llu refill_cycles %llu simd %llu cu %llu sh %llu se %llu xcc %llu\n", w, h[w * 8], h[w * 8 + 3], h[w * 8 + 5], h[w * 8 + 7], h[w * 8 + 2], h[w * 8 + 1], (hw >> 4) & 3,
                (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 32) & 15);
      }
  }
  return 0;
}

template <typename CoefT>
static int LaunchEntropyUniBatch(JxlHipContext* c0) {
  auto k = jxlhip::k_entropy_uni<CoefT, kEntropyWPG>;
  if (c0->batch_lds > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(c0->batch_lds)));
  jxlhip::EntropyBatch b{c0->batch_params.as<jxlhip::EntropyParams>(), c0->batch_map.as<uint32_t>()};
  hipLaunchKernelGGL(k, dim3(c0->batch_wgs), dim3(64 * kEntropyWPG), c0->batch_lds, c0->stream, b);
  HIP_TRY(hipGetLastError());
  return 0;
}

// Builds (or re-uses) the description of a batch: parameter blocks, workgroup -> frame map and, for the lane-parallel
// kernel, the lane -> section assignment (sections sorted by compressed size inside each frame so that the lanes of a
// wave finish together; `lanes_per_wave` populated lanes per wave so that small batches still spread over all SIMDs).
static int PrepareBatch(JxlHipContext* c0, JxlHipContext* const* ctxs, size_t n, int kernel) {
  bool same = c0->batch_ctxs.size() == n && c0->batch_kernel == kernel;
  for (size_t i = 0; same && i < n; i++) same = c0->batch_ctxs[i] == ctxs[i] && c0->batch_gens[i] == ctxs[i]->generation;
  if (same) return 0;
  std::vector<jxlhip::EntropyParams> params(n);
  std::vector<uint32_t> map, list, unit_desc;
  std::vector<uint8_t> wave_lanes;
  std::vector<uint8_t> wave_ls;
  size_t lds = 0;
  for (size_t i = 0; i < n; i++) params[i] = ctxs[i]->ep;
  if (kernel == 2) {
    // unit = the sections of one frame that share a histogram set (a workgroup stages ONE set's slice of the context
    // map; the selector is read from the first bits of every section at upload). The lanes of a unit take its sections
    // from a queue, largest first, so the split by lanes below only fixes HOW MANY lanes serve a unit:
    //   * a launch lasts as long as its busiest lane, which cannot beat the unit's longest section, so a unit gets
    //     about (its total cost) / (cost of its longest section) lanes: more would idle (typical 4K d1.0 frame: 135
    //     sections of 12k-100k tokens, 60 lanes). The cost of a section is estimated as bytes + kSectionBytes (token
    //     counts are unknown before decoding; a constant term dominates in sparse sections);
    //   * small batches that leave SIMDs empty spread over up to one lane per section;
    //   * the lanes of a unit are split evenly over the waves of its workgroups, and a wave's LDS rows are strided by
    //     its lane count rounded up to a power of two.
    const uint32_t kSectionBytes = uint32_t(EnvInt("JXLHIP_SECTION_BYTES", 4000));
    const size_t target_waves = size_t(EnvInt("JXLHIP_TARGET_WAVES", 1024));
    const int forced = EnvInt("JXLHIP_LANES", 0);  // measurement aid: lanes per wave
    const int spread = EnvInt("JXLHIP_SPREAD", 100);  // percent of the minimum lane count (>= 100)
    c0->batch_wait_shift = uint32_t(EnvInt("JXLHIP_WAIT_SHIFT", 1));
    // waves per workgroup: ONE. The waves of a workgroup slow each other down (a lone 4K frame, one lane per wave: 832
    // cycles per trip with four waves per workgroup, 517 with two, 417 with one; 29.6 / 24.1 / 19.6 ms for the frame:
    // profiles/r04_single_frame.txt), and a workgroup of its own per wave costs only another copy of the tables in LDS.
    // JXLHIP_WPG = 2 / 4 keeps the shared-table forms reachable.
    uint32_t wpg = uint32_t(EnvInt("JXLHIP_WPG", 1));
    if (wpg != 1 && wpg != 2) wpg = 4;
    c0->batch_wpg = wpg;
    struct Unit {
      uint32_t frame, sel, pass, begin, count, min_lanes, lanes;
    };
    std::vector<Unit> units;
    std::vector<uint32_t> order;
    size_t min_total = 0, total_sections = 0;
    for (size_t i = 0; i < n; i++) {
      const JxlHipContext* c = ctxs[i];
      for (uint32_t pass = 0; pass < c->np; pass++)
      for (uint32_t sel = 0; sel < c->ep.num_hist; sel++) {
        const uint32_t* sz = c->sec_size_host.data() + size_t(pass) * c->ng;    // [pass * num_groups + group]
        const uint32_t* ssel = c->sec_sel_host.data() + size_t(pass) * c->ng;
        order.clear();
        for (uint32_t g : c->group_list) {
          if (pass && sz[g] == 0 && !c->absent_sections.empty()) continue;  // (a partial frame: this pass of the group has not arrived)
          if (ssel[g] == sel || (sel == 0 && ssel[g] >= c->ep.num_hist)) order.push_back(g);
        }
        if (order.empty()) continue;
        std::stable_sort(order.begin(), order.end(), [sz](uint32_t a, uint32_t b) { return sz[a] > sz[b]; });
        uint64_t total = 0;
        for (uint32_t g : order) total += uint64_t(sz[g]) + kSectionBytes;
        const uint64_t longest = uint64_t(sz[order[0]]) + kSectionBytes;
        Unit u;
        u.frame = uint32_t(i);
        u.sel = sel;
        u.pass = pass;
        u.begin = uint32_t(list.size());
        u.count = uint32_t(order.size());
        u.min_lanes = uint32_t((total * uint64_t(spread < 100 ? 100 : spread) / 100 + longest - 1) / longest);
        u.min_lanes = u.min_lanes > u.count ? u.count : (u.min_lanes ? u.min_lanes : 1);
        u.lanes = u.min_lanes;
        list.insert(list.end(), order.begin(), order.end());
        min_total += u.min_lanes;
        total_sections += u.count;
        units.push_back(u);
      }
    }
    // lanes per wave: the fewest that fit all lanes into `target_waves` waves (a trip costs ~900 + 20 * lanes cycles)
    uint32_t lanes_per_wave = 1;
    while (lanes_per_wave < 64 && (min_total + lanes_per_wave - 1) / lanes_per_wave > target_waves) lanes_per_wave *= 2;
    if (forced >= 1 && forced <= 64 && (forced & (forced - 1)) == 0) lanes_per_wave = uint32_t(forced);
    if (!EnvInt("JXLHIP_WPG", 0) && lanes_per_wave >= 32) {  // dense regime: one wave per workgroup, all of a unit's lanes
      wpg = 1;                                                // in it (a trip costs about the same for 14 or 55 lanes)
      lanes_per_wave = 64;
      for (Unit& u : units)  // the wave's spare lanes are free: shorter lists per lane
        if (u.lanes < 64) u.lanes = u.count < 64 ? u.count : 64;
    }
    c0->batch_wpg = wpg;
    if (lanes_per_wave == 1 && min_total < target_waves) {  // room to spare: more lanes per unit, up to one per section
      const double grow = double(target_waves) / double(min_total);
      for (Unit& u : units) {
        const uint32_t want = uint32_t(double(u.min_lanes) * grow);
        u.lanes = want > u.count ? u.count : (want < u.min_lanes ? u.min_lanes : want);
      }
    }
    // Dense regime: a frame's few largest sections decide how long its launch lasts (the lane that decodes one of them
    // runs one token per trip whatever the other lanes do, and every transition pass of the wave delays it), so they get a
    // wave of their own with few lanes (few transitions to serve) beside the wave for all the other sections; the two
    // share one set of LDS rows (EntropyLaneBatch::wave_log_ls).
    const uint32_t split = (wpg == 1 && lanes_per_wave == 64) ? uint32_t(EnvInt("JXLHIP_SPLIT", 0)) : 0u;  // (0: measured slower on the benchmark frames)
    std::vector<uint8_t> wg_shared;  // per workgroup: its waves share one row set
    if (split && split < 64) {
      wpg = 2;
      c0->batch_wpg = 2;
    }
    for (const Unit& u : units) {
      if (split && split < 64) {
        const uint32_t heavy = u.count >= 64 ? split : 0u;  // (a small unit: one wave)
        const uint32_t first_index = uint32_t(unit_desc.size() / 4);
        const uint32_t parts[2][2] = {{u.begin, heavy}, {u.begin + heavy, u.count - heavy}};
        for (int part = 0; part < 2; part++) {
          unit_desc.push_back(u.frame | u.sel << 16);
          unit_desc.push_back(parts[part][0]);
          unit_desc.push_back(parts[part][1]);
          unit_desc.push_back(u.pass);
        }
        const uint32_t light_lanes = std::min<uint32_t>(64 - heavy, u.count - heavy);
        map.push_back(first_index);
        map.push_back(first_index + 1);
        wave_lanes.push_back(uint8_t(heavy));
        wave_lanes.push_back(uint8_t(light_lanes));
        wave_ls.push_back(uint8_t(0x80 | (64 - heavy) % 64));  // the heavy wave's lanes use the top columns
        wave_ls.push_back(uint8_t(0x80));
        wg_shared.push_back(1);
        continue;
      }
      const uint32_t per_wg = lanes_per_wave * wpg;
      const uint32_t wgs = (u.lanes + per_wg - 1) / per_wg;
      const uint32_t waves = wgs * wpg;
      const uint32_t unit_index = uint32_t(unit_desc.size() / 4);
      unit_desc.push_back(u.frame | u.sel << 16);
      unit_desc.push_back(u.begin);
      unit_desc.push_back(u.count);
      unit_desc.push_back(u.pass);
      for (uint32_t j = 0; j < wgs; j++) wg_shared.push_back(0);
      for (uint32_t w = 0; w < waves; w++) {  // even split of the unit's lanes over its waves
        const uint32_t cnt = u.lanes / waves + (w < u.lanes % waves ? 1u : 0u);
        map.push_back(unit_index);
        wave_ls.push_back(uint8_t(0));  // (rows of its own, 64 lanes wide: jxl_hip_entropy_lanes.h)
        wave_lanes.push_back(uint8_t(cnt));
      }
    }
    // LDS of the launch = the largest workgroup. With the alias tables in LDS a CU holds 160 KB / that many workgroups;
    // when that leaves part of the launch waiting for a second round (libjxl-sized tables: 128 clusters x 2^6 slots are
    // 64 KB), the tables stay in global memory instead: a cached global round trip on every token's serial chain, but
    // every frame resident (JXLHIP_GALIAS = 0 / 1 forces either form: measurement aid)
    // Between the two, for tables of up to 128 clusters and 8192 slots (128 x 2^6: what libjxl writes for large frames): the six-byte
    // form of the tables in LDS (70 KB per frame: two frames per CU instead of one), when THAT leaves the launch resident.
    size_t lds_by_form[3] = {0, 0, 0};
    const size_t num_wgs = map.size() / wpg;  // (`map` holds one unit per wave)
    bool a6_ok = wpg == 1 && !c0->lane_prefix && c0->coef_bits == 16 && EnvInt("JXLHIP_A6", 1) != 0;
    for (int form = 0; form < 3; form++)
      for (size_t wg = 0; wg < num_wgs; wg++) {
        const JxlHipContext* c = ctxs[unit_desc[size_t(map[wg * wpg]) * 4] & 0xFFFF];
        const uint32_t up = unit_desc[size_t(map[wg * wpg]) * 4 + 3];  // the unit's pass
        if (c->pass_log_alpha[up] > 7 || c->pass_clusters[up] > jxlhip::kLanesA6Clusters ||
            (size_t(c->pass_clusters[up]) << c->pass_log_alpha[up]) > jxlhip::kLanesA6Slots)
          a6_ok = false;
        size_t l = jxlhip::LanesLdsLayout(1, c->ep.nctx, c->pass_clusters[up], c->pass_log_alpha[up], 0, 0, form == 0 && !c0->lane_prefix,
                                          c0->lane_prefix, form == 2).wave0;
        l += size_t(jxlhip::kLanesPerLaneBytes) * 64 * (wg_shared[wg] ? 1 : wpg);
        lds_by_form[form] = l > lds_by_form[form] ? l : lds_by_form[form];
      }
    int dev_cus = 256;
    (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c0->device);
    const size_t resident = lds_by_form[0] ? size_t(dev_cus) * ((160 * 1024) / lds_by_form[0]) : num_wgs;
    bool galias = lds_by_form[0] > kLdsBudget || resident < num_wgs;
    const int forced_form = EnvInt("JXLHIP_GALIAS", -1);
    if (forced_form == 0 && lds_by_form[0] <= kLdsBudget) galias = false;
    if (forced_form == 1) galias = true;
    if (c0->lane_prefix) galias = true;  // (prefix codes have no alias tables: the two forms are the same)
    const size_t resident6 = lds_by_form[2] ? size_t(dev_cus) * ((160 * 1024) / lds_by_form[2]) : num_wgs;
    const bool a6 = a6_ok && lds_by_form[2] <= kLdsBudget &&
                    ((galias && forced_form != 1 && resident6 >= num_wgs) || EnvInt("JXLHIP_A6", 1) == 2);  // (2: whenever eligible; tests)
    if (a6) galias = false;
    c0->batch_a6 = a6;
    c0->batch_galias = galias;
    c0->batch_prefix = c0->lane_prefix;
    lds = lds_by_form[a6 ? 2 : (galias ? 1 : 0)] + size_t(EnvInt("JXLHIP_LDS_PAD", 0));  // (measurement aid: fewer workgroups per CU)
    if (EnvInt("JXLHIP_PACK_DEBUG", 0))
      fprintf(stderr, "[pack] units %zu sections %zu lanes(min) %zu lanes/wave %u waves/wg %u workgroups %zu lds %zu tables %s\n", units.size(),
              total_sections, min_total, lanes_per_wave, wpg, map.size() / wpg, lds,
              c0->lane_prefix ? "prefix" : (a6 ? "lds6" : (galias ? "global" : "lds8")));
  } else {
    for (size_t i = 0; i < n; i++) {
      const uint32_t wgs = (ctxs[i]->ng + kEntropyWPG - 1) / kEntropyWPG;
      for (uint32_t j = 0; j < wgs; j++) map.push_back(uint32_t(i) << 16 | j);
      const size_t l = size_t(ctxs[i]->ep.lds_ctx_bytes) + ctxs[i]->ep.lds_alias_bytes + 1024 + kEntropyWPG * 5120;
      lds = l > lds ? l : lds;
    }
  }
  int r;
  if ((r = c0->batch_params.Ensure(params.size() * sizeof(params[0])))) return r;
  if ((r = c0->batch_map.Ensure(map.size() * 4))) return r;
  ParamCopy pc;
  if ((r = pc.Begin(c0))) return r;
  if ((r = pc.Add(c0->batch_params.p, params.data(), params.size() * sizeof(params[0])))) return r;
  if ((r = pc.Add(c0->batch_map.p, map.data(), map.size() * 4))) return r;
  if (!wave_ls.empty()) {
    if ((r = c0->batch_wave_ls.Ensure(wave_ls.size()))) return r;
    if ((r = pc.Add(c0->batch_wave_ls.p, wave_ls.data(), wave_ls.size()))) return r;
  }
  if (!list.empty()) {
    // one buffer: section list | unit descriptors (16-byte aligned) | queue counters | populated lanes per wave
    const size_t o_units = (list.size() * 4 + 15) & ~size_t(15), o_queue = o_units + unit_desc.size() * 4;
    const size_t o_wl = o_queue + unit_desc.size(), total = o_wl + wave_lanes.size();
    std::vector<uint8_t> blob(total, 0);
    memcpy(blob.data(), list.data(), list.size() * 4);
    memcpy(blob.data() + o_units, unit_desc.data(), unit_desc.size() * 4);
    memcpy(blob.data() + o_wl, wave_lanes.data(), wave_lanes.size());
    if ((r = c0->batch_lanes.Ensure(total))) return r;
    if ((r = pc.Add(c0->batch_lanes.p, blob.data(), total))) return r;
    c0->batch_off_units = o_units;
    c0->batch_off_queue = o_queue;
    c0->batch_off_wave_lanes = o_wl;
    c0->batch_units = unit_desc.size() / 4;
  }
  if ((r = pc.End())) return r;
  c0->batch_ctxs.assign(ctxs, ctxs + n);
  c0->batch_gens.resize(n);
  for (size_t i = 0; i < n; i++) c0->batch_gens[i] = ctxs[i]->generation;
  c0->batch_wgs = uint32_t(kernel == 2 ? map.size() / c0->batch_wpg : map.size());
  c0->batch_lds = lds;
  c0->batch_kernel = kernel;
  return 0;
}

extern "C" int jxlhip_run_entropy_batch(JxlHipContext* const* ctxs, size_t n) {
  if (!ctxs || !n || n > 0xFFFF) return JXLHIP_ERR_INVALID_ARGUMENT;
  JxlHipContext* c0 = ctxs[0];
  // One launch needs one kernel: every frame in scan-order layout (lane kernel), or every frame in natural layout with
  // tables that the scalar-form kernel can share in LDS; anything else is decoded frame by frame.
  bool all_scan = true, all_uni = true;
  for (size_t i = 0; i < n; i++) {
    const JxlHipContext* c = ctxs[i];
    if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (!c->have_frame) return JXLHIP_ERR_NO_FRAME;
    if (c->device != c0->device) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (n > 1 && !c->absent_groups.empty()) return JXLHIP_ERR_INVALID_ARGUMENT;  // (partial frames go one at a time: jxlhip_run_entropy)
    if (!c->lanes || c->coef_bits != c0->coef_bits || c->lane_prefix != c0->lane_prefix) all_scan = false;
    if (c->lanes || c->generic_codec || !c->alias_lds || c->ep.num_hist != 1 || c->np != 1 || c->coef_bits != c0->coef_bits ||
        c->ng > 0xFFFF * kEntropyWPG || EntropyKernelChoice() == 0)
      all_uni = false;
  }
  if (!all_scan && !all_uni) {
    for (size_t i = 0; i < n; i++) {
      int r = ctxs[i]->lanes ? jxlhip_run_entropy_batch(&ctxs[i], 1) : RunEntropySingle(ctxs[i]);
      if (r) return r;
    }
    return 0;
  }
  const int kernel = all_scan ? 2 : 1;
  HIP_TRY(hipSetDevice(c0->device));
  if (n > 1) {
    const int r = EnsureOwnStream(c0);
    if (r) return r;
  }
  // measurement aid: JXLHIP_BATCH_PROF=1 prints the host time of the phases of this call (microseconds)
  const bool prof = EnvInt("JXLHIP_BATCH_PROF", 0) != 0;
  std::chrono::steady_clock::time_point tp0 = std::chrono::steady_clock::now();
  std::string prof_line;
  auto lap = [&](const char* what) {
    if (!prof) return;
    const auto now = std::chrono::steady_clock::now();
    prof_line += std::string(what) + " " + std::to_string(std::chrono::duration_cast<std::chrono::microseconds>(now - tp0).count()) + "  ";
    tp0 = now;
  };
  if (!c0->batch_done) HIP_TRY(hipEventCreateWithFlags(&c0->batch_done, hipEventDisableTiming));
  int r = PrepareBatch(c0, ctxs, n, kernel);
  if (r) return r;
  lap("prepare");
  // the batch kernel runs on the first context's stream, after whatever the other contexts still have in flight
  for (size_t i = 1; i < n; i++) {
    if (ctxs[i]->ev_valid[2]) HIP_TRY(hipStreamWaitEvent(c0->stream, ctxs[i]->ev[5], 0));
  }
  hipEvent_t waited = nullptr;
  for (size_t i = 0; i < n; i++)
    if (ctxs[i]->pending_wait) {  // results of an earlier batch launch that nobody consumed: order after that launch
      if (ctxs[i]->pending_wait != waited) HIP_TRY(hipStreamWaitEvent(c0->stream, ctxs[i]->pending_wait, 0));
      waited = ctxs[i]->pending_wait;
      ctxs[i]->pending_wait = nullptr;
    }
  // Measurement aid (JXLHIP_SERIAL_ENTROPY=1): batched entropy launches of one device one after the other, whatever
  // streams they are on.
  static std::mutex serial_mu;
  static hipEvent_t serial_tail[16] = {};
  // MEASURED WORSE, hence off: alone beside the other sets' transform / filter launches for its whole length, a launch
  // takes 112 - 126 ms (77 - 82 ms when two entropy launches overlap part of the time; 62 ms with nothing beside it):
  // the lone waves lose far more to issue-bound neighbours on their SIMDs than to each other. s_setprio does not help.
  const bool serial = n > 1 && c0->device >= 0 && c0->device < 16 && EnvInt("JXLHIP_SERIAL_ENTROPY", 0) != 0;
  std::unique_lock<std::mutex> serial_lock(serial_mu, std::defer_lock);  // (held from the wait to the record: host threads)
  if (serial) {
    serial_lock.lock();
    hipEvent_t& tail = serial_tail[c0->device];
    if (!tail) HIP_TRY(hipEventCreateWithFlags(&tail, hipEventDisableTiming));
    else HIP_TRY(hipStreamWaitEvent(c0->stream, tail, 0));
  }
  lap("waits");
  HIP_TRY(hipEventRecord(c0->ev[0], c0->stream));
  for (size_t i = 0; i < n; i++)  // (the lane kernel writes every section's flag word itself, unless passes share it)
    if (kernel != 2 || ctxs[i]->np > 1) HIP_TRY(hipMemsetAsync(ctxs[i]->errors.p, 0, size_t(ctxs[i]->ng) * 4, c0->stream));
  if (kernel == 2) r = c0->coef_bits == 16 ? LaunchEntropyLanes<int16_t>(c0) : LaunchEntropyLanes<int32_t>(c0);
  else r = c0->coef_bits == 16 ? LaunchEntropyUniBatch<int16_t>(c0) : LaunchEntropyUniBatch<int32_t>(c0);
  if (r) return r;
  if (kernel == 2)
    for (size_t i = 0; i < n; i++) {  // progressive frames: sum the passes into the natural layout
      const JxlHipContext* c = ctxs[i];
      if (!c->lane_multi) continue;
      jxlhip::MergeParams m;
      m.coeffs = c->coeffs.p;
      m.blocks = c->blocks.as<JxlHipVarBlock>();
      m.passes = c->passes_dev.as<jxlhip::PassDev>();
      m.kend = c->kend.as<uint32_t>();
      m.pass_stride = c->ep.coef_pass_stride;
      m.num_blocks = c->nblocks;
      m.num_passes = c->np;
      m.xg = c->xg;
      m.kend_stride = c->ep.kend_pass_stride;
      if (!m.num_blocks) continue;
      if (c->coef_bits == 16) hipLaunchKernelGGL(jxlhip::k_merge_passes<int16_t>, dim3(m.num_blocks), dim3(64), 0, c0->stream, m);
      else hipLaunchKernelGGL(jxlhip::k_merge_passes<int32_t>, dim3(m.num_blocks), dim3(64), 0, c0->stream, m);
      HIP_TRY(hipGetLastError());
    }
  lap("launch");
  if (prof) fprintf(stderr, "[entropy batch] %s\n", prof_line.c_str());
  HIP_TRY(hipEventRecord(c0->ev[1], c0->stream));
  c0->ev_valid[0] = true;
  if (serial) HIP_TRY(hipEventRecord(serial_tail[c0->device], c0->stream));
  if (n > 1) {
    HIP_TRY(hipEventRecord(c0->batch_done, c0->stream));
    for (size_t i = 1; i < n; i++) {
      ctxs[i]->pending_wait = c0->batch_done;
      ctxs[i]->ev_valid[0] = false;
    }
  }
  return 0;
}

// Groups whose AC sections have not arrived (JxlHipFrameDesc::group_absent): no coefficients at all, so that the transform
// stage draws their blocks from the lowest frequencies (the DC image) alone. Queued in front of the entropy launch.
static int ZeroAbsentGroups(JxlHipContext* c) {
  if (c->absent_groups.empty() && c->absent_sections.empty()) return 0;
  HIP_TRY(hipSetDevice(c->device));
  const size_t coef_group_bytes = size_t(3) * 65536 * (c->coef_bits / 8);
  for (size_t i = 0; i < c->absent_groups.size(); i++) {
    const uint32_t g = c->absent_groups[i], b0 = c->absent_blocks[2 * i], nb = c->absent_blocks[2 * i + 1];
    HIP_TRY(hipMemsetAsync(c->coeffs.as<uint8_t>() + size_t(g) * coef_group_bytes, 0, coef_group_bytes, c->stream));
    for (uint32_t p = 0; p < c->np && nb && c->lanes; p++)  // (the scan-order layout's counts; the other kernels zero-fill natural layout)
      HIP_TRY(hipMemsetAsync(c->kend.as<uint32_t>() + size_t(p) * c->ep.kend_pass_stride + size_t(b0) * 3, 0, size_t(nb) * 12, c->stream));
  }
  for (size_t i = 0; c->lanes && i + 2 < c->absent_sections.size(); i += 3) {  // passes that have not arrived add nothing (k_merge_passes)
    const uint32_t p = c->absent_sections[i], b0 = c->absent_sections[i + 1], nb = c->absent_sections[i + 2];
    if (nb) HIP_TRY(hipMemsetAsync(c->kend.as<uint32_t>() + size_t(p) * c->ep.kend_pass_stride + size_t(b0) * 3, 0, size_t(nb) * 12, c->stream));
  }
  return 0;
}

extern "C" int jxlhip_run_entropy(JxlHipContext* c) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return JXLHIP_ERR_NO_FRAME;
  const int r = ZeroAbsentGroups(c);
  if (r) return r;
  if (c->group_list.empty()) {  // (nothing but the DC image has arrived)
    HIP_TRY(hipMemsetAsync(c->errors.p, 0, size_t(c->ng) * 4, c->stream));
    return 0;
  }
  return c->lanes ? jxlhip_run_entropy_batch(&c, 1) : RunEntropySingle(c);
}


// The draw cache of a frame's splines to the device (validated: every index a kernel forms from it is in range).
static int UploadSplines(JxlHipContext* c, const JxlHipSplines& sp, uint32_t ysize) {
  c->spl_segments = 0;
  if (!sp.num_segments) return 0;
  if (!sp.segments || !sp.row_start || !sp.row_segments || sp.num_segments > (1u << 26) || sp.num_row_segments > (1u << 30))
    return JXLHIP_ERR_INVALID_ARGUMENT;
  if (sp.row_start[0] != 0 || sp.row_start[ysize] != sp.num_row_segments) return JXLHIP_ERR_INVALID_ARGUMENT;
  for (uint32_t y = 0; y < ysize; y++)
    if (sp.row_start[y] > sp.row_start[y + 1]) return JXLHIP_ERR_INVALID_ARGUMENT;
  for (uint32_t i = 0; i < sp.num_row_segments; i++)
    if (sp.row_segments[i] >= sp.num_segments) return JXLHIP_ERR_INVALID_ARGUMENT;
  int r;
  if ((r = Upload(c, c->spl_seg, sp.segments, size_t(sp.num_segments) * 32))) return r;
  if ((r = Upload(c, c->spl_row_start, sp.row_start, (size_t(ysize) + 1) * 4))) return r;
  if ((r = Upload(c, c->spl_row_seg, sp.row_segments, std::max<size_t>(4, size_t(sp.num_row_segments) * 4)))) return r;
  c->spl_segments = sp.num_segments;
  return 0;
}

// ------------------------------------------------------------------------------------------------ Modular frames
extern "C" int jxlhip_modular_upload(JxlHipContext* c, const JxlHipModFrameDesc* d) {
  if (!c || !d || !d->xsize || !d->ysize || !d->num_sections || !d->num_trees || !d->num_codes || d->num_trees != d->num_codes)
    return JXLHIP_ERR_INVALID_ARGUMENT;
  if (d->num_color != 1 && d->num_color != 3) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
    if ((pw = StageReset(c))) return pw;
  }
  JxlHipContext::Modular& M = c->mod;
  M.have = false;
  c->have_frame = false;
  int r;
  // ---- channel pool
  M.buf_off.assign(d->num_buffers, 0);
  M.buf_w.assign(d->num_buffers, 0);
  M.buf_h.assign(d->num_buffers, 0);
  size_t pool = 0;
  for (uint32_t i = 0; i < d->num_buffers; i++) {
    M.buf_off[i] = pool;
    M.buf_w[i] = d->buffers[i].w;
    M.buf_h[i] = d->buffers[i].h;
    pool += (size_t(d->buffers[i].w) * d->buffers[i].h + 3) & ~size_t(3);
  }
  if ((r = M.pool.Ensure((pool + 4) * 4))) return r;
  // (a channel buffer that no stream covers must read as zeros, not as the previous frame's samples)
  HIP_TRY(hipMemsetAsync(M.pool.p, 0, (pool + 4) * 4, c->stream));
  // ---- sections, 4-byte aligned, zero padded
  std::vector<size_t> sec_off(d->num_sections);
  size_t total = 0;
  for (uint32_t i = 0; i < d->num_sections; i++) {
    sec_off[i] = total;
    total += (size_t(d->section_size[i]) + 3 + 8) & ~size_t(3);
  }
  std::vector<uint8_t> packed(total + 16, 0);
  for (uint32_t i = 0; i < d->num_sections; i++) memcpy(packed.data() + sec_off[i], d->codestream + d->section_offset[i], d->section_size[i]);
  if ((r = Upload(c, M.sections, packed.data(), packed.size()))) return r;
  // ---- trees and codes: one blob, device pointers patched in
  std::vector<uint8_t> blob;
  auto put = [&](const void* src, size_t bytes) {
    const size_t at = (blob.size() + 15) & ~size_t(15);
    blob.resize(at + bytes);
    if (bytes) memcpy(blob.data() + at, src, bytes);
    return at;
  };
  std::vector<size_t> tree_at(d->num_trees), code_at(d->num_codes);
  std::vector<std::vector<size_t>> code_parts(d->num_codes);  // ctx_map, alias, cfg, prefix_table, prefix_offset
  for (uint32_t i = 0; i < d->num_trees; i++) {
    // the kernel's 16-byte nodes: leaves carry -1 - predictor, the offset, the histogram their context maps to, the multiplier
    std::vector<jxlhip::ModTreeNode> packed_tree(d->tree_size[i]);
    for (uint32_t k = 0; k < d->tree_size[i]; k++) {  // every index the walk can follow must exist
      const JxlHipModTreeNode& nd = d->trees[i][k];
      if (nd.property >= int32_t(jxlhip::kModMaxProps)) return JXLHIP_ERR_UNSUPPORTED;
      // children behind their parent (the breadth-first order dec_ma.cc:107-159 produces): a back edge would make the
      // kernel's tree walk spin forever
      if (nd.property >= 0 && (nd.lchild >= d->tree_size[i] || nd.rchild >= d->tree_size[i] || nd.lchild <= k || nd.rchild <= k))
        return JXLHIP_ERR_INVALID_ARGUMENT;
      if (nd.property < 0 && (nd.predictor > 13 || nd.lchild >= d->codes[i].ctx_map_size || !d->codes[i].ctx_map)) return JXLHIP_ERR_INVALID_ARGUMENT;
      jxlhip::ModTreeNode& o = packed_tree[k];
      if (nd.property >= 0) {
        o.property = nd.property;
        o.splitval = nd.splitval;
        o.lchild = nd.lchild;
        o.rchild = nd.rchild;
      } else {
        o.property = -1 - int32_t(nd.predictor);
        o.splitval = nd.offset;
        o.lchild = d->codes[i].ctx_map[nd.lchild];
        o.rchild = nd.multiplier;
      }
    }
    tree_at[i] = put(packed_tree.data(), packed_tree.size() * sizeof(jxlhip::ModTreeNode));
  }
  for (uint32_t i = 0; i < d->num_codes; i++) {
    const JxlHipModCode& k = d->codes[i];
    if (!d->tree_size[i]) {  // (an absent global tree)
      code_parts[i] = {0, 0, 0, 0, 0};
      continue;
    }
    if (!k.num_clusters || k.num_clusters > 256 || !k.ctx_map_size) return JXLHIP_ERR_INVALID_ARGUMENT;
    for (uint32_t j = 0; j < k.ctx_map_size; j++)
      if (k.ctx_map[j] >= k.num_clusters) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (!k.use_prefix && (k.log_alpha < 5 || k.log_alpha > 8)) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (k.use_prefix)
      for (uint32_t j = 0; j < k.num_clusters; j++) {
        if (!ValidPrefixTables(k.prefix_offset[j], k.prefix_table, k.prefix_table_size)) return JXLHIP_ERR_INVALID_ARGUMENT;
      }
    if (k.lz77 && (k.lz_dist_ctx >= k.num_clusters || !k.lz_min_length)) return JXLHIP_ERR_INVALID_ARGUMENT;
    code_parts[i] = {put(k.ctx_map, k.ctx_map_size), put(k.alias, k.use_prefix ? 0 : (size_t(k.num_clusters) << k.log_alpha) * 8),
                     put(k.uint_cfg, size_t(k.num_clusters) * 4), put(k.prefix_table, k.use_prefix ? size_t(k.prefix_table_size) * 4 : 0),
                     put(k.prefix_offset, k.use_prefix ? size_t(k.num_clusters) * 4 : 0)};
  }
  const size_t codes_at = (blob.size() + 15) & ~size_t(15);
  blob.resize(codes_at + size_t(d->num_codes) * sizeof(jxlhip::ModCode));
  if ((r = M.blob.Ensure(blob.size() + 16))) return r;
  uint8_t* dev = M.blob.as<uint8_t>();
  M.code_table_words.assign(d->num_codes, 0);
  for (uint32_t i = 0; i < d->num_codes; i++) {
    const JxlHipModCode& k = d->codes[i];
    jxlhip::ModCode mc;
    memset(&mc, 0, sizeof(mc));
    mc.ctx_map = dev + code_parts[i][0];
    mc.alias = reinterpret_cast<const uint2*>(dev + code_parts[i][1]);
    mc.cfg = reinterpret_cast<const uint32_t*>(dev + code_parts[i][2]);
    mc.prefix_table = reinterpret_cast<const uint32_t*>(dev + code_parts[i][3]);
    mc.prefix_offset = reinterpret_cast<const uint32_t*>(dev + code_parts[i][4]);
    mc.log_alpha = k.log_alpha;
    mc.use_prefix = k.use_prefix;
    mc.lz77 = k.lz77;
    mc.lz_min_symbol = k.lz_min_symbol;
    mc.lz_min_length = k.lz_min_length;
    mc.lz_len_cfg = k.lz_len_cfg;
    mc.lz_dist_ctx = k.lz_dist_ctx;
    mc.num_clusters = k.num_clusters;
    if (d->tree_size[i])
      mc.table_words = k.use_prefix ? k.prefix_table_size + 2 * k.num_clusters : ((k.num_clusters << k.log_alpha) * 2 + k.num_clusters);
    M.code_table_words[i] = mc.table_words;
    memcpy(blob.data() + codes_at + size_t(i) * sizeof(mc), &mc, sizeof(mc));
  }
  HIP_TRY(hipMemcpyAsync(M.blob.p, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream));
  // ---- rectangles and streams
  std::vector<jxlhip::ModChannel> rects(d->num_rects ? d->num_rects : 1);
  for (uint32_t i = 0; i < d->num_rects; i++) {
    const JxlHipModRect& q = d->rects[i];
    if (q.buffer >= d->num_buffers || uint64_t(q.x0) + q.w > M.buf_w[q.buffer] || uint64_t(q.y0) + q.h > M.buf_h[q.buffer])
      return JXLHIP_ERR_INVALID_ARGUMENT;
    rects[i].data = M.pool.as<int32_t>() + M.buf_off[q.buffer] + size_t(q.y0) * M.buf_w[q.buffer] + q.x0;
    rects[i].stride = M.buf_w[q.buffer];
    rects[i].w = q.w;
    rects[i].h = q.h;
    rects[i].sig = q.sig;
  }
  if ((r = Upload(c, M.rects, rects.data(), rects.size() * sizeof(rects[0])))) return r;
  M.nstreams = d->num_streams;
  if ((r = M.status.Ensure(size_t(d->num_streams + 1) * 4))) return r;
  if ((r = M.end_bits.Ensure(size_t(d->num_streams + 1) * 4))) return r;
  size_t scratch_ints = 0, window_words = 0;
  std::vector<size_t> scratch_at(d->num_streams, 0), window_at(d->num_streams, 0);
  std::vector<uint32_t> window_mask(d->num_streams, 0);
  for (uint32_t i = 0; i < d->num_streams; i++) {
    const JxlHipModStream& q = d->streams[i];
    if (q.section >= d->num_sections || q.tree >= d->num_trees || q.code >= d->num_codes || !d->tree_size[q.tree] ||
        uint64_t(q.first_rect) + q.num_rects > d->num_rects || q.bit_offset > uint64_t(d->section_size[q.section]) * 8 ||
        q.num_props < 16 || q.num_props > uint32_t(jxlhip::kModMaxProps))
      return JXLHIP_ERR_INVALID_ARGUMENT;
    if (q.uses_wp) {
      scratch_at[i] = scratch_ints;
      scratch_ints += size_t(q.max_width + 2) * 2 * jxlhip::kModWpEntry;
    }
    if (d->codes[q.code].lz77) {
      uint32_t w = 256;
      while (w < q.num_samples && w < (1u << 20)) w <<= 1;  // as large as the stream can fill, at most the format's 2^20
      window_at[i] = window_words;
      window_mask[i] = w - 1;
      window_words += w;
    }
  }
  if ((r = M.scratch.Ensure((scratch_ints + 4) * 4))) return r;
  if ((r = M.windows.Ensure((window_words + 4) * 4))) return r;
  M.streams_host.assign(d->num_streams, jxlhip::ModStream());
  M.stream_samples.assign(d->num_streams, 0);
  for (uint32_t i = 0; i < d->num_streams; i++) {
    const JxlHipModStream& q = d->streams[i];
    jxlhip::ModStream& o = M.streams_host[i];
    memset(&o, 0, sizeof(o));
    o.words = reinterpret_cast<const uint32_t*>(M.sections.as<uint8_t>() + sec_off[q.section]);
    o.bit_offset = q.bit_offset;
    o.size_bytes = d->section_size[q.section];
    o.tree = reinterpret_cast<const jxlhip::ModTreeNode*>(dev + tree_at[q.tree]);
    o.tree_nodes = d->tree_size[q.tree];
    o.code = reinterpret_cast<const jxlhip::ModCode*>(dev + codes_at) + q.code;
    o.channels = M.rects.as<jxlhip::ModChannel>() + q.first_rect;
    o.num_channels = q.num_rects;
    o.stream_id = q.stream_id;
    o.first_channel_index = q.first_channel_index;
    memcpy(o.wp, q.wp, sizeof(o.wp));
    o.uses_wp = q.uses_wp;
    o.num_props = q.num_props;
    o.dist_multiplier = q.dist_multiplier;
    o.wp_scratch = q.uses_wp ? M.scratch.as<int32_t>() + scratch_at[i] : nullptr;
    o.lz_window = d->codes[q.code].lz77 ? M.windows.as<uint32_t>() + window_at[i] : nullptr;
    o.lz_window_mask = window_mask[i];
    o.status = M.status.as<uint32_t>() + i;
    o.end_bit = M.end_bits.as<uint32_t>() + i;
    M.stream_samples[i] = q.num_samples;
  }
  if ((r = Upload(c, M.streams, M.streams_host.data(), M.streams_host.size() * sizeof(jxlhip::ModStream)))) return r;
  // ---- operations and output
  M.ops.assign(d->ops, d->ops + d->num_ops);
  for (const JxlHipModOp& op : M.ops) {
    const uint32_t nbuf = op.kind == 0 ? 3 : (op.kind == 1 ? 2 + op.nb : 3);
    if (op.kind > 3 || nbuf > 6) return JXLHIP_ERR_INVALID_ARGUMENT;
    for (uint32_t j = 0; j < nbuf; j++)
      if (op.buf[j] >= d->num_buffers) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (op.kind == 0) {
      for (int j = 0; j < 3; j++)
        if (uint64_t(op.x0) + op.w > M.buf_w[op.buf[j]] || uint64_t(op.y0) + op.h > M.buf_h[op.buf[j]]) return JXLHIP_ERR_INVALID_ARGUMENT;
      if (op.param >= 42) return JXLHIP_ERR_INVALID_ARGUMENT;
    } else if (op.kind == 1) {
      if (op.nb < 1 || op.nb > 4 || M.buf_h[op.buf[0]] < op.nb || op.param != M.buf_w[op.buf[0]]) return JXLHIP_ERR_INVALID_ARGUMENT;
      for (uint32_t j = 1; j < 2 + op.nb; j++)
        if (M.buf_w[op.buf[j]] != op.w || M.buf_h[op.buf[j]] != op.h) return JXLHIP_ERR_INVALID_ARGUMENT;
    } else {
      const uint32_t a = op.buf[0], q = op.buf[1], o = op.buf[2];
      const bool hz = op.kind == 2;
      const uint32_t na = hz ? M.buf_w[a] : M.buf_h[a], nr = hz ? M.buf_w[q] : M.buf_h[q], no = hz ? M.buf_w[o] : M.buf_h[o];
      const uint32_t la = hz ? M.buf_h[a] : M.buf_w[a], lr = hz ? M.buf_h[q] : M.buf_w[q], lo = hz ? M.buf_h[o] : M.buf_w[o];
      if (na + nr != no || (nr != na && nr + 1 != na) || la != lr || la != lo) return JXLHIP_ERR_INVALID_ARGUMENT;
    }
  }
  for (uint32_t j = 0; j < d->num_color + (d->has_alpha ? 1 : 0); j++) {
    if (d->out_buffer[j] >= d->num_buffers || M.buf_w[d->out_buffer[j]] != d->xsize || M.buf_h[d->out_buffer[j]] != d->ysize)
      return JXLHIP_ERR_INVALID_ARGUMENT;
    M.out_buffer[j] = d->out_buffer[j];
  }
  M.num_color = d->num_color;
  M.has_alpha = d->has_alpha;
  M.bits = d->bits;
  M.alpha_bits = d->alpha_bits;
  {  // sample depths: low byte = bits, bits 8..15 = exponent bits of a float type (image_metadata.cc BitDepth: 2..8, mantissa 2..23)
    auto depth_ok = [](uint32_t d) {
      const uint32_t bits = d & 0xFF, e = (d >> 8) & 0xFF;
      if (d >> 16) return false;
      if (e == 0) return bits >= 1 && bits <= 31;
      return e >= 2 && e <= 8 && bits >= e + 3 && bits <= e + 24 && bits <= 32;
    };
    if (!depth_ok(M.bits) || !depth_ok(M.alpha_bits) || (M.alpha_bits >> 8) || (M.alpha_bits & 0xFF) > 24) return JXLHIP_ERR_INVALID_ARGUMENT;
  }
  M.xs = d->xsize;
  M.ys = d->ysize;
  c->oxs = c->xs = d->xsize;
  c->oys = c->ys = d->ysize;
  if ((r = c->rgb.Ensure(size_t(c->oxs) * c->oys * OutPixelBytes(c)))) return r;
  HIP_TRY(hipStreamSynchronize(c->stream));  // the staging vectors are locals
  if (d->splines.num_segments && d->num_color != 3) return JXLHIP_ERR_UNSUPPORTED;
  M.xyb = d->xyb != 0;
  if (M.xyb) {
    if (d->num_color != 3) return JXLHIP_ERR_UNSUPPORTED;
    memset(&M.xyb_color, 0, sizeof(M.xyb_color));
    for (int ch = 0; ch < 3; ch++) {
      M.xyb_factor[ch] = d->xyb_factor[ch];
      M.xyb_color.opsin_bias[ch] = d->opsin_bias[ch];
      M.xyb_color.opsin_bias_cbrt[ch] = cbrtf(d->opsin_bias[ch]);
    }
    memcpy(M.xyb_color.opsin_inv, d->opsin_inv, sizeof(M.xyb_color.opsin_inv));
    M.xyb_color.linear_output = d->linear_output;
  }
  if ((r = UploadSplines(c, d->splines, d->ysize))) return r;
  if (d->patches.num_positions && d->patches.uses_alpha) return JXLHIP_ERR_UNSUPPORTED;  // (the frame's alpha is an integer channel here)
  if ((r = UploadPatches(c, d->patches, d->ysize, d->xsize, d->ysize))) return r;
  const bool float_planes = c->spl_segments || c->keep_xyb || c->pat_positions;
  if (float_planes && d->num_color != 3) return JXLHIP_ERR_UNSUPPORTED;
  if (float_planes && (r = c->spl_planes.Ensure(size_t(d->xsize) * d->ysize * 3 * 4))) return r;
  M.have = true;
  M.batch_ctxs.clear();
  c->generation++;
  return 0;
}

// The inverse transforms and the pixel writer of a set of frames, as launches over parameter-block arrays: launch k does
// the k-th transform of every frame that has one (one launch per kind), so a frame's steps stay in order on the stream
// and a step of all frames shares the device. Built once per set (cached with the stream list), blocks in `out` (host),
// launches in `launches`.
static void ModularBuildOps(JxlHipContext* const* ctxs, size_t n, std::vector<uint8_t>* blob, std::vector<ModLaunch>* launches) {
  blob->clear();
  launches->clear();
  size_t levels = 0;
  for (size_t i = 0; i < n; i++) levels = std::max(levels, ctxs[i]->mod.ops.size());
  auto append = [&](const void* p, size_t bytes) {
    const size_t at = blob->size();
    blob->resize(at + bytes);
    memcpy(blob->data() + at, p, bytes);
  };
  auto align = [&]() { blob->resize((blob->size() + 15) & ~size_t(15)); };
  for (size_t level = 0; level < levels; level++) {
    for (uint32_t kind = 0; kind < 3; kind++) {
      align();
      ModLaunch L{kind, blob->size(), 0, 0, 0};
      for (size_t i = 0; i < n; i++) {
        const JxlHipContext::Modular& M = ctxs[i]->mod;
        if (level >= M.ops.size()) continue;
        const JxlHipModOp& op = M.ops[level];
        int32_t* pool = M.pool.as<int32_t>();
        if (kind == 0 && op.kind == 0) {
          jxlhip::ModRct p;
          memset(&p, 0, sizeof(p));
          for (int j = 0; j < 3; j++) {
            p.stride[j] = M.buf_w[op.buf[j]];
            p.c[j] = pool + M.buf_off[op.buf[j]] + size_t(op.y0) * p.stride[j] + op.x0;
          }
          p.w = op.w;
          p.h = op.h;
          p.type = op.param;
          if (!op.w || !op.h) continue;
          append(&p, sizeof(p));
          L.count++;
          L.gx = std::max(L.gx, (op.w + 255) / 256);
          L.gy = std::max(L.gy, op.h);
        } else if (kind == 1 && op.kind == 1) {
          jxlhip::ModPalette p;
          memset(&p, 0, sizeof(p));
          p.palette = pool + M.buf_off[op.buf[0]];
          p.index = pool + M.buf_off[op.buf[1]];
          for (uint32_t j = 0; j < op.nb; j++) p.out[j] = pool + M.buf_off[op.buf[2 + j]];
          p.palette_w = op.param;
          p.nb = op.nb;
          p.w = op.w;
          p.h = op.h;
          p.bit_depth = op.bit_depth;
          p.index_stride = op.w;
          p.out_stride = op.w;
          if (!op.w || !op.h) continue;
          append(&p, sizeof(p));
          L.count++;
          L.gx = std::max(L.gx, (op.w + 255) / 256);
          L.gy = std::max(L.gy, op.h);
        } else if (kind == 2 && op.kind >= 2) {
          const uint32_t a = op.buf[0], q = op.buf[1], o = op.buf[2];
          const bool hz = op.kind == 2;
          jxlhip::ModUnsqueeze p;
          memset(&p, 0, sizeof(p));
          p.avg = pool + M.buf_off[a];
          p.res = pool + M.buf_off[q];
          p.out = pool + M.buf_off[o];
          p.lines = hz ? M.buf_h[a] : M.buf_w[a];
          p.na = hz ? M.buf_w[a] : M.buf_h[a];
          p.nr = hz ? M.buf_w[q] : M.buf_h[q];
          p.avg_line = hz ? M.buf_w[a] : 1;
          p.avg_step = hz ? 1 : M.buf_w[a];
          p.res_line = hz ? M.buf_w[q] : 1;
          p.res_step = hz ? 1 : M.buf_w[q];
          p.out_line = hz ? M.buf_w[o] : 1;
          p.out_step = hz ? 1 : M.buf_w[o];
          if (!p.lines) continue;
          append(&p, sizeof(p));
          L.count++;
          L.gx = std::max(L.gx, (p.lines + 63) / 64);
          L.gy = 1;
        }
      }
      if (L.count) launches->push_back(L);
    }
  }
  // frames with patches / splines: colour samples to float planes (kind 4), the patches (kind 6), then the splines (kind 5)
  // over them (dec_cache.cc:193-201), then the output reads them
  for (uint32_t pass : {4u, 6u, 5u}) {
    align();
    ModLaunch S{pass, blob->size(), 0, 0, 0};
    for (size_t i = 0; i < n; i++) {
      const JxlHipContext* c = ctxs[i];
      const JxlHipContext::Modular& M = c->mod;
      if (pass == 4 ? !(c->spl_segments || c->keep_xyb || c->pat_positions) : (pass == 6 ? !c->pat_positions : !c->spl_segments)) continue;
      if (pass == 6) {
        jxlhip::PatchParams pp;
        memset(&pp, 0, sizeof(pp));
        pp.planes = c->spl_planes.as<float>();
        pp.records = c->pat_rec.as<uint32_t>();
        pp.row_start = c->pat_row_start.as<uint32_t>();
        pp.row_list = c->pat_row_list.as<uint32_t>();
        for (int k = 0; k < 4; k++) {
          pp.slot_planes[k] = c->pat_src[k];
          pp.slot_w[k] = c->pat_src_w[k];
          pp.slot_h[k] = c->pat_src_h[k];
        }
        pp.stride = M.xs;
        pp.plane_stride = size_t(M.xs) * M.ys;
        pp.xsize = M.xs;
        pp.y_begin = 0;
        pp.y_end = M.ys;
        append(&pp, sizeof(pp));
        S.gx = 1;
      } else if (pass == 4) {
        jxlhip::ModOutput o;
        memset(&o, 0, sizeof(o));
        for (uint32_t j = 0; j < 3; j++) {
          o.ch[j] = M.pool.as<int32_t>() + M.buf_off[M.out_buffer[j]];
          o.stride[j] = M.xs;
        }
        o.num_color = 3;
        o.bits = M.bits;
        o.w = M.xs;
        o.h = M.ys;
        o.fplanes = c->spl_planes.as<float>();
        o.fmode = 1;
        o.xyb = M.xyb ? 1 : 0;  // (the planes then hold XYB; the final pass converts what it reads back)
        memcpy(o.xyb_factor, M.xyb_factor, sizeof(o.xyb_factor));
        append(&o, sizeof(o));
        S.gx = std::max(S.gx, (M.xs + 255) / 256);
      } else {
        jxlhip::SplineParams sp;
        memset(&sp, 0, sizeof(sp));
        sp.planes = c->spl_planes.as<float>();
        sp.segments = c->spl_seg.as<float>();
        sp.row_start = c->spl_row_start.as<uint32_t>();
        sp.row_segments = c->spl_row_seg.as<uint32_t>();
        sp.stride = M.xs;
        sp.plane_stride = size_t(M.xs) * M.ys;
        sp.xsize = M.xs;
        sp.y_begin = 0;
        sp.y_end = M.ys;
        append(&sp, sizeof(sp));
        S.gx = 1;
      }
      S.count++;
      S.gy = std::max(S.gy, M.ys);
    }
    if (S.count) launches->push_back(S);
  }
  align();
  ModLaunch L{3, blob->size(), 0, 0, 0};
  for (size_t i = 0; i < n; i++) {
    const JxlHipContext* c = ctxs[i];
    const JxlHipContext::Modular& M = c->mod;
    int32_t* pool = M.pool.as<int32_t>();
    jxlhip::ModOutput o;
    memset(&o, 0, sizeof(o));
    if (c->spl_segments || c->keep_xyb || c->pat_positions) {
      o.fplanes = c->spl_planes.as<float>();
      o.fmode = 2;
    }
    o.xyb = M.xyb ? 1 : 0;
    memcpy(o.xyb_factor, M.xyb_factor, sizeof(o.xyb_factor));
    o.color = M.xyb_color;
    for (uint32_t j = 0; j < M.num_color + (M.has_alpha ? 1 : 0); j++) {
      o.ch[j] = pool + M.buf_off[M.out_buffer[j]];
      o.stride[j] = M.xs;
    }
    o.num_color = M.num_color;
    o.has_alpha = M.has_alpha;
    o.bits = M.bits;
    o.alpha_bits = M.alpha_bits;
    o.w = M.xs;
    o.h = M.ys;
    o.po.dst = c->rgb.p;
    o.po.alpha = nullptr;
    o.po.xsize = M.xs;
    o.po.ysize = M.ys;
    o.po.orient = c->out_orient | (c->out_unpremul ? 8u : 0u);
    o.po.type = c->out_type;
    o.po.nc = c->out_nc;
    o.po.bits = c->out_bits;
    o.po.swap = c->out_swap;
    append(&o, sizeof(o));
    L.count++;
    L.gx = std::max(L.gx, (M.xs + 255) / 256);
    L.gy = std::max(L.gy, M.ys);
  }
  launches->push_back(L);
}
static int ModularLaunchOps(const std::vector<ModLaunch>& launches, const uint8_t* dev, hipStream_t st) {
  for (const ModLaunch& L : launches)
    for (uint32_t z = 0; z < L.count; z += 65535) {  // grid z / y limit
      const uint32_t zn = std::min<uint32_t>(L.count - z, 65535);
      const uint32_t gy = std::min<uint32_t>(L.gy, 4096);  // (the row loops stride by the grid)
      if (L.kind == 0)
        hipLaunchKernelGGL(jxlhip::k_modular_rct, dim3(L.gx, gy, zn), dim3(256), 0, st, reinterpret_cast<const jxlhip::ModRct*>(dev + L.offset) + z);
      else if (L.kind == 1)
        hipLaunchKernelGGL(jxlhip::k_modular_palette, dim3(L.gx, gy, zn), dim3(256), 0, st, reinterpret_cast<const jxlhip::ModPalette*>(dev + L.offset) + z);
      else if (L.kind == 2)
        hipLaunchKernelGGL(jxlhip::k_modular_unsqueeze, dim3(L.gx, zn), dim3(64), 0, st, reinterpret_cast<const jxlhip::ModUnsqueeze*>(dev + L.offset) + z);
      else if (L.kind == 5)  // (one workgroup per row: the full height, not the strided grid of the sample kernels)
        hipLaunchKernelGGL(jxlhip::k_splines_add_batch, dim3(L.gy, zn), dim3(256), 0, st, reinterpret_cast<const jxlhip::SplineParams*>(dev + L.offset) + z);
      else if (L.kind == 6)
        hipLaunchKernelGGL(jxlhip::k_patches_add_batch, dim3(L.gy, zn), dim3(256), 0, st, reinterpret_cast<const jxlhip::PatchParams*>(dev + L.offset) + z);
      else
        hipLaunchKernelGGL(jxlhip::k_modular_output, dim3(L.gx, gy, zn), dim3(256), 0, st, reinterpret_cast<const jxlhip::ModOutput*>(dev + L.offset) + z);
      HIP_TRY(hipGetLastError());
    }
  return 0;
}

extern "C" int jxlhip_modular_run_batch(JxlHipContext* const* ctxs, size_t n) {
  if (!ctxs || !n) return JXLHIP_ERR_INVALID_ARGUMENT;
  JxlHipContext* c0 = ctxs[0];
  for (size_t i = 0; i < n; i++) {
    if (!ctxs[i] || ctxs[i]->device != c0->device) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (!ctxs[i]->mod.have) return JXLHIP_ERR_NO_FRAME;
    if (size_t(ctxs[i]->oxs) * ctxs[i]->oys * OutPixelBytes(ctxs[i]) > ctxs[i]->rgb.cap) return JXLHIP_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(c0->device));
  if (n > 1) {
    const int r = EnsureOwnStream(c0);
    if (r) return r;
  }
  JxlHipContext::Modular& M0 = c0->mod;
  bool same = M0.batch_ctxs.size() == n;
  for (size_t i = 0; same && i < n; i++) same = M0.batch_ctxs[i] == ctxs[i] && M0.batch_gens[i] == ctxs[i]->generation;
  if (!same) {
    // every stream of every frame, largest first: the lanes of a wave then carry streams of similar length
    struct Ref {
      uint32_t samples;
      const jxlhip::ModStream* s;
    };
    std::vector<Ref> all;
    for (size_t i = 0; i < n; i++)
      for (size_t j = 0; j < ctxs[i]->mod.streams_host.size(); j++) all.push_back({ctxs[i]->mod.stream_samples[j], &ctxs[i]->mod.streams_host[j]});
    std::stable_sort(all.begin(), all.end(), [](const Ref& a, const Ref& b) { return a.samples > b.samples; });
    std::vector<jxlhip::ModStream> flat(all.size() ? all.size() : 1);
    for (size_t i = 0; i < all.size(); i++) flat[i] = *all[i].s;
    int r = M0.batch_streams.Ensure(flat.size() * sizeof(jxlhip::ModStream));
    if (r) return r;
    HIP_TRY(hipMemcpy(M0.batch_streams.p, flat.data(), flat.size() * sizeof(jxlhip::ModStream), hipMemcpyHostToDevice));
    M0.batch_ctxs.assign(ctxs, ctxs + n);
    M0.batch_gens.resize(n);
    for (size_t i = 0; i < n; i++) M0.batch_gens[i] = ctxs[i]->generation;
    M0.batch_n = uint32_t(all.size());
    uint32_t cap = 0;  // LDS room for the largest tree of the batch that fits (waves whose streams share a tree stage it)
    for (const Ref& q : all)
      if (q.s->tree_nodes <= jxlhip::kModTreeLdsNodes && q.s->tree_nodes > cap) cap = q.s->tree_nodes;
    M0.batch_tree_cap = cap;
    M0.batch_wp = M0.batch_refs = false;  // the kernel form: what some stream of the launch needs
    for (const Ref& q : all) {
      M0.batch_wp = M0.batch_wp || q.s->uses_wp != 0;
      M0.batch_refs = M0.batch_refs || q.s->num_props > 16;
    }
    uint32_t tcap = 0;  // likewise for the largest set of symbol tables that fits
    const uint32_t table_limit = uint32_t(EnvInt("JXLHIP_MOD_TABLE_WORDS", int(jxlhip::kModTableLdsWords)));  // (measurement aid)
    for (size_t i = 0; i < n; i++)
      for (uint32_t w : ctxs[i]->mod.code_table_words)
        if (w <= table_limit && w > tcap) tcap = w;
    M0.batch_table_cap = tcap;
    std::vector<uint8_t> ops_blob;
    ModularBuildOps(ctxs, n, &ops_blob, &M0.batch_launches);
    r = M0.batch_ops.Ensure(ops_blob.size() + 16);
    if (r) return r;
    if (!ops_blob.empty()) HIP_TRY(hipMemcpy(M0.batch_ops.p, ops_blob.data(), ops_blob.size(), hipMemcpyHostToDevice));
  }
  for (size_t i = 0; i < n; i++) {
    int pw = ApplyPendingWait(ctxs[i]);
    if (pw) return pw;
  }
  for (size_t i = 1; i < n; i++) {  // the launch runs on the first context's stream, after what the others still have in flight
    if (!ctxs[i]->down_done) HIP_TRY(hipEventCreateWithFlags(&ctxs[i]->down_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ctxs[i]->down_done, ctxs[i]->stream));
    HIP_TRY(hipStreamWaitEvent(c0->stream, ctxs[i]->down_done, 0));
  }
  HIP_TRY(hipEventRecord(c0->ev[0], c0->stream));
  if (M0.batch_n) {
    // A stream is a chain of dependent latencies: as few streams per wave as still leaves every SIMD of the device a
    // few waves (JXLHIP_MOD_LANES overrides: measurement aid). 384 lossless 4K frames = 53 760 streams: 32 lanes per wave
    // 2 219 MP/s, 16 lanes 2 360, 8 lanes 1 864, 4 lanes 1 166: up to four waves per SIMD.
    uint32_t lanes = 1;
    while (lanes < 64 && M0.batch_n / lanes > 4096) lanes *= 2;
    const int forced = EnvInt("JXLHIP_MOD_LANES", 0);
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16 || forced == 32 || forced == 64) lanes = uint32_t(forced);
    const uint32_t lds = jxlhip::ModLdsBytes(lanes, M0.batch_tree_cap, M0.batch_table_cap);
    auto kernel = M0.batch_wp ? (M0.batch_refs ? jxlhip::k_modular_streams<true, true> : jxlhip::k_modular_streams<true, false>)
                              : (M0.batch_refs ? jxlhip::k_modular_streams<false, true> : jxlhip::k_modular_streams<false, false>);
    if (lds > 48 * 1024)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(kernel, dim3((M0.batch_n + lanes - 1) / lanes), dim3(64), lds, c0->stream,
                       M0.batch_streams.as<jxlhip::ModStream>(), M0.batch_n, lanes, M0.batch_tree_cap, M0.batch_table_cap);
    HIP_TRY(hipGetLastError());
  }
  {
    int r = ModularLaunchOps(M0.batch_launches, M0.batch_ops.as<uint8_t>(), c0->stream);
    if (r) return r;
  }
  HIP_TRY(hipEventRecord(c0->ev[1], c0->stream));
  c0->ev_valid[0] = true;
  if (n > 1) {
    if (!c0->batch_done) HIP_TRY(hipEventCreateWithFlags(&c0->batch_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c0->batch_done, c0->stream));
    for (size_t i = 1; i < n; i++) ctxs[i]->pending_wait = c0->batch_done;
  }
  for (size_t i = 0; i < n; i++) ctxs[i]->have_frame = true;  // (the pixel download entry points need it)
  return 0;
}
extern "C" int jxlhip_modular_run(JxlHipContext* c) { return jxlhip_modular_run_batch(&c, 1); }

extern "C" int jxlhip_modular_status(JxlHipContext* c, uint32_t* status, uint32_t* end_bits, size_t n) {
  if (!c || !status) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->mod.have) return JXLHIP_ERR_NO_FRAME;
  if (n < c->mod.nstreams) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->mod.nstreams) {
    HIP_TRY(hipMemcpy(status, c->mod.status.p, size_t(c->mod.nstreams) * 4, hipMemcpyDeviceToHost));
    if (end_bits) HIP_TRY(hipMemcpy(end_bits, c->mod.end_bits.p, size_t(c->mod.nstreams) * 4, hipMemcpyDeviceToHost));
  }
  for (uint32_t i = 0; i < c->mod.nstreams; i++)
    if (status[i]) return JXLHIP_ERR_STREAM;
  return 0;
}

extern "C" int jxlhip_modular_download_buffer(JxlHipContext* c, uint32_t buffer, int32_t* dst, size_t n) {
  if (!c || !dst) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->mod.have) return JXLHIP_ERR_NO_FRAME;
  if (buffer >= c->mod.buf_off.size() || n < size_t(c->mod.buf_w[buffer]) * c->mod.buf_h[buffer]) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(dst, c->mod.pool.as<int32_t>() + c->mod.buf_off[buffer], size_t(c->mod.buf_w[buffer]) * c->mod.buf_h[buffer] * 4,
                    hipMemcpyDeviceToHost));
  return 0;
}

extern "C" {

int jxlhip_run_transform_batch(JxlHipContext* const* ctxs, size_t n) {
  int r = BeginDownstreamBatch(ctxs, n);
  if (r) return r;
  JxlHipContext* c0 = ctxs[0];
  HIP_TRY(hipEventRecord(c0->ev[2], c0->stream));
  r = c0->coef_bits == 16 ? LaunchTransforms<int16_t>(c0) : LaunchTransforms<int32_t>(c0);
  if (r) return r;
  HIP_TRY(hipEventRecord(c0->ev[3], c0->stream));
  c0->ev_valid[1] = true;
  for (size_t i = 1; i < n; i++) ctxs[i]->ev_valid[1] = false;
  return EndDownstreamBatch(ctxs, n);
}

int jxlhip_run_filter_color_batch(JxlHipContext* const* ctxs, size_t n) {
  int r = BeginDownstreamBatch(ctxs, n, true);
  if (r) return r;
  JxlHipContext* c0 = ctxs[0];
  const hipStream_t ls = c0->fstream;
  HIP_TRY(hipEventRecord(c0->ev[4], ls));
  for (const JxlHipContext::FilterGroup& g : c0->fgroups) {
    switch (g.key) {
      case 0: r = LaunchFused<false, 0>(c0, g); break;
      case 1: r = EnvInt("JXLHIP_FILTER_TILES", 0) ? LaunchFused<false, 1>(c0, g) : LaunchFilterRows(c0, g, 1, false); break;
      case 2: r = EnvInt("JXLHIP_FILTER_TILES", 0) ? LaunchFused<false, 2>(c0, g) : LaunchFilterRows(c0, g, 2, false); break;
      case 3: r = LaunchFused<false, 3>(c0, g); break;
      case 4: r = LaunchFused<true, 0>(c0, g); break;
      case 5: r = EnvInt("JXLHIP_FILTER_TILES", 0) ? LaunchFused<true, 1>(c0, g) : LaunchFilterRows(c0, g); break;
      case 6: r = EnvInt("JXLHIP_FILTER_TILES", 0) ? LaunchFused<true, 2>(c0, g) : LaunchFilterRows(c0, g, 2); break;
      default: r = LaunchFused<true, 3>(c0, g); break;
    }
    if (r) return r;
  }
  for (size_t i = 0; i < n; i++) {
    JxlHipContext* c = ctxs[i];
    c->alpha_patched_valid = false;
    if (!c->pat_positions) continue;
    jxlhip::PatchParams pp;
    memset(&pp, 0, sizeof(pp));
    if (c->pat_uses_alpha) {  // the frame's alpha plane must be there by now (jxlhip_set_alpha); blended into a copy of it
      const size_t bytes = size_t(c->xs) * c->ys * 4;
      if (!c->have_alpha || c->alpha.cap < bytes) return JXLHIP_ERR_INVALID_ARGUMENT;
      int ar = c->alpha_patched.Ensure(bytes);
      if (ar) return ar;
      HIP_TRY(hipMemcpyAsync(c->alpha_patched.p, c->alpha.p, bytes, hipMemcpyDeviceToDevice, ls));
      pp.alpha = c->alpha_patched.as<float>();
      pp.alpha_stride = c->xs;
      pp.premultiplied = c->pat_premultiplied ? 1 : 0;
      for (int k = 0; k < 4; k++) pp.slot_alpha[k] = c->pat_src_alpha[k];
      c->alpha_patched_valid = true;
    }
    pp.planes = c->plane[1].as<float>();
    pp.records = c->pat_rec.as<uint32_t>();
    pp.row_start = c->pat_row_start.as<uint32_t>();
    pp.row_list = c->pat_row_list.as<uint32_t>();
    for (int k = 0; k < 4; k++) {
      pp.slot_planes[k] = c->pat_src[k];
      pp.slot_w[k] = c->pat_src_w[k];
      pp.slot_h[k] = c->pat_src_h[k];
    }
    pp.stride = c->xp;
    pp.plane_stride = size_t(c->xp) * c->yp;
    pp.xsize = c->xs;
    pp.y_begin = c->band_y0;
    pp.y_end = c->band_y1;
    hipLaunchKernelGGL(jxlhip::k_patches_add, dim3(c->band_y1 - c->band_y0), dim3(256), 0, ls, pp);
    HIP_TRY(hipGetLastError());
  }
  for (size_t i = 0; i < n; i++) {
    const JxlHipContext* c = ctxs[i];
    if (!c->spl_segments) continue;
    jxlhip::SplineParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.planes = c->plane[1].as<float>();
    sp.segments = c->spl_seg.as<float>();
    sp.row_start = c->spl_row_start.as<uint32_t>();
    sp.row_segments = c->spl_row_seg.as<uint32_t>();
    sp.stride = c->xp;
    sp.plane_stride = size_t(c->xp) * c->yp;
    sp.xsize = c->xs;
    sp.y_begin = c->band_y0;
    sp.y_end = c->band_y1;
    hipLaunchKernelGGL(jxlhip::k_splines_add, dim3(c->band_y1 - c->band_y0), dim3(256), 0, ls, sp);
    HIP_TRY(hipGetLastError());
  }
  for (size_t i = 0; i < n; i++) {
    const JxlHipContext* c = ctxs[i];
    if (!c->has_noise || c->ups != 1) continue;  // (upsampled frames: behind the upsampling, below)
    jxlhip::NoiseParams np;
    memset(&np, 0, sizeof(np));
    np.raw = c->noise.as<float>();
    np.planes = c->plane[1].as<float>();
    np.xsize = c->xs;
    np.ysize = c->ys;
    np.xp = c->xp;
    np.yp = c->yp;
    np.xgroups = c->xg;
    np.ngroups = c->ng;
    np.seed[0] = c->noise_seed[0];
    np.seed[1] = c->noise_seed[1];
    memcpy(np.lut, c->noise_lut, sizeof(np.lut));
    np.ytox = c->noise_ytox;
    np.ytob = c->noise_ytob;
    np.y_begin = c->band_y0;
    np.y_end = c->band_y1;
    hipLaunchKernelGGL(jxlhip::k_noise_random, dim3((c->ng * 8 + 63) / 64), dim3(64), 0, ls, np);
    hipLaunchKernelGGL(jxlhip::k_noise_add, dim3((c->xs + 63) / 64, (c->band_y1 - c->band_y0 + 3) / 4), dim3(256), 0, ls, np);
    HIP_TRY(hipGetLastError());
  }
  for (size_t i = 0; i < n; i++) {
    const JxlHipContext* c = ctxs[i];
    if (c->ups == 1 && !c->color_out) continue;
    jxlhip::PixelOut po;
    memset(&po, 0, sizeof(po));
    if (c->color_out) {
      po.dst = c->rgb.p;
      po.alpha = c->have_alpha ? (c->alpha_patched_valid ? c->alpha_patched.as<float>() : c->alpha.as<float>()) : nullptr;
      po.xsize = c->oxs;
      po.ysize = c->oys;
      po.orient = c->out_orient | (c->out_unpremul ? 8u : 0u);
      po.type = c->out_type;
      po.nc = c->out_nc;
      po.bits = c->out_bits;
      po.swap = c->out_swap;
    }
    if (c->ups == 1) {
      jxlhip::ColorOutParams cp;
      cp.f = c->fp;
      cp.f.in = c->plane[1].as<float>();
      cp.po = po;
      hipLaunchKernelGGL(jxlhip::k_color_out, dim3((c->xs + 63) / 64, (c->band_y1 - c->band_y0 + 3) / 4), dim3(256), 0, ls, cp);
      HIP_TRY(hipGetLastError());
      continue;
    }
    jxlhip::UpsampleParams up;
    up.f = c->fp;
    up.f.in = c->plane[1].as<float>();
    up.f.rgb = c->rgb.as<uint8_t>();
    up.kernel = c->ups_kernel.as<float>();
    up.n = c->ups;
    up.oxs = c->oxs;
    up.oys = c->oys;
    up.po = po;
    up.xyb_out = nullptr;
    up.oxp = 0;
    if (c->has_noise) {  // upsample to planes, add the noise at the image's resolution, then the colour stage
      up.xyb_out = c->ups_planes.as<float>();
      up.oxp = (c->oxs + 7) & ~7u;
    }
    hipLaunchKernelGGL(jxlhip::k_upsample_color, dim3((c->xs + 63) / 64, (c->ys + 3) / 4), dim3(256), 0, ls, up);
    HIP_TRY(hipGetLastError());
    if (c->has_noise) {
      jxlhip::NoiseParams np;
      memset(&np, 0, sizeof(np));
      np.raw = c->noise.as<float>();
      np.planes = up.xyb_out;
      np.xsize = c->oxs;
      np.ysize = c->oys;
      np.xp = up.oxp;
      np.yp = c->oys;
      np.xgroups = (c->oxs + 255) / 256;
      np.ngroups = np.xgroups * ((c->oys + 255) / 256);
      np.seed[0] = c->noise_seed[0];
      np.seed[1] = c->noise_seed[1];
      memcpy(np.lut, c->noise_lut, sizeof(np.lut));
      np.ytox = c->noise_ytox;
      np.ytob = c->noise_ytob;
      np.y_begin = 0;
      np.y_end = c->oys;
      hipLaunchKernelGGL(jxlhip::k_noise_random, dim3((np.ngroups * 8 + 63) / 64), dim3(64), 0, ls, np);
      hipLaunchKernelGGL(jxlhip::k_noise_add, dim3((c->oxs + 63) / 64, (c->oys + 3) / 4), dim3(256), 0, ls, np);
      jxlhip::ColorOutParams cp;
      cp.f = c->fp;
      cp.f.in = up.xyb_out;
      cp.f.xs = c->oxs;
      cp.f.ys = c->oys;
      cp.f.xp = up.oxp;
      cp.f.yp = c->oys;
      cp.f.y_begin = 0;
      cp.f.y_end = c->oys;
      cp.po = po;
      hipLaunchKernelGGL(jxlhip::k_color_out, dim3((c->oxs + 63) / 64, (c->oys + 3) / 4), dim3(256), 0, ls, cp);
      HIP_TRY(hipGetLastError());
    }
  }
  HIP_TRY(hipEventRecord(c0->ev[5], ls));
  c0->ev_valid[2] = true;
  for (size_t i = 0; i < n; i++) ctxs[i]->final_plane = 1;
  for (size_t i = 1; i < n; i++) ctxs[i]->ev_valid[2] = false;
  return EndDownstreamBatch(ctxs, n, true);
}

int jxlhip_run_transform(JxlHipContext* c) { return jxlhip_run_transform_batch(&c, 1); }
int jxlhip_run_filter_color(JxlHipContext* c) { return jxlhip_run_filter_color_batch(&c, 1); }

int jxlhip_share_planes(JxlHipContext* c, JxlHipContext* lender) {
  if (!c || c == lender || (lender && (lender->plane_lender || lender->device != c->device))) return JXLHIP_ERR_INVALID_ARGUMENT;
  c->plane_lender = lender;
  if (lender && !lender->planes_event) {  // marks the lender as shared from now on (its launches record the event)
    HIP_TRY(hipSetDevice(lender->device));
    if (!lender->down_done) HIP_TRY(hipEventCreateWithFlags(&lender->down_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(lender->down_done, lender->stream));
    lender->planes_event = lender->down_done;
    lender->planes_stream = lender->stream;
  }
  c->generation++;  // cached batch descriptions hold plane pointers
  return 0;
}

// ---- halo exchange between the bands of one frame: the filters of a band read up to `rows` rows of the
// inverse-transform output above and below it; with "band_halo" those rows are not decoded a second time but copied from
// the neighbouring band's planes (another context: another GPU's, through any device-to-device transport).
int jxlhip_halo_rows(JxlHipContext* c, uint32_t* rows) {
  if (!c || !rows) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return JXLHIP_ERR_NO_FRAME;
  *rows = uint32_t(jxlhip::FusedHalo(c->gab != 0, int(c->epf_iters)));
  return 0;
}
namespace {
// rows [y0, y0 + rows) of the three planes <-> a dense [3][rows][xp] f32 block, enqueued on `st`
int HaloCopy(JxlHipContext* c, uint32_t y0, uint32_t rows, void* block, bool to_block, hipStream_t st) {
  const size_t row_bytes = size_t(c->xp) * 4, plane = size_t(c->xp) * c->yp;
  float* planes = PlaneHolder(c)->plane[0].as<float>();
  for (int ch = 0; ch < 3; ch++) {
    float* in_plane = planes + ch * plane + size_t(y0) * c->xp;
    float* in_block = static_cast<float*>(block) + size_t(ch) * rows * c->xp;
    HIP_TRY(hipMemcpyAsync(to_block ? in_block : in_plane, to_block ? in_plane : in_block, row_bytes * rows, hipMemcpyDeviceToDevice, st));
  }
  return 0;
}
int HaloCheck(JxlHipContext* c, int side, const void* p, size_t bytes, bool pack, uint32_t* rows) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  int r = jxlhip_halo_rows(c, rows);
  if (r) return r;
  if (!p || (side != 0 && side != 1) || bytes < size_t(3) * *rows * c->xp * 4) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (pack ? c->band_y1 - c->band_y0 < *rows : (side == 0 ? c->band_y0 < *rows : c->band_y1 + *rows > c->ys)) return JXLHIP_ERR_INVALID_ARGUMENT;
  return 0;
}
// the stream on which the halo copies of a frame set led by c0 run: the one its batched transform launch ran on
hipStream_t HaloStream(JxlHipContext* c0) { return c0->stream; }
}  // namespace
// side 0: the first rows of this band (what the band ABOVE needs), side 1: its last rows (what the band BELOW needs);
// `dst` = device memory of at least 3 * rows * xp floats. Runs behind the context's transform; complete on return.
int jxlhip_halo_pack(JxlHipContext* c, int side, void* dst, size_t dst_bytes) {
  uint32_t rows = 0;
  int r = HaloCheck(c, side, dst, dst_bytes, true, &rows);
  if (r) return r;
  HIP_TRY(hipSetDevice(c->device));
  if (!rows) return 0;
  if ((r = ApplyPendingWait(c))) return r;  // (behind the batched transform launch that produced the planes)
  if ((r = HaloCopy(c, side == 0 ? c->band_y0 : c->band_y1 - rows, rows, dst, true, c->stream))) return r;
  HIP_TRY(hipStreamSynchronize(c->stream));  // the block is handed to a transport this library knows nothing about
  return 0;
}
// side 0: the rows just above this band (the band above packed them with side 1), side 1: the rows just below it.
int jxlhip_halo_unpack(JxlHipContext* c, int side, const void* src, size_t src_bytes) {
  uint32_t rows = 0;
  int r = HaloCheck(c, side, src, src_bytes, false, &rows);
  if (r) return r;
  HIP_TRY(hipSetDevice(c->device));
  if (!rows) return 0;
  if ((r = ApplyPendingWait(c))) return r;
  if ((r = HaloCopy(c, side == 0 ? c->band_y0 - rows : c->band_y1, rows, const_cast<void*>(src), false, c->stream))) return r;
  HIP_TRY(hipStreamSynchronize(c->stream));  // the planes go to a filter launch on another stream
  return 0;
}
// The set forms: every frame of a set (the contexts of one batched transform launch, same order) in one call, block i at
// `dst + i * block_bytes`, ordered against the TRANSPORT's stream by events instead of host synchronisation.
int jxlhip_halo_pack_batch(JxlHipContext* const* ctxs, size_t n, int side, void* dst, size_t block_bytes, void* transport_stream) {
  if (!ctxs || !n || !ctxs[0]) return JXLHIP_ERR_INVALID_ARGUMENT;
  JxlHipContext* c0 = ctxs[0];
  HIP_TRY(hipSetDevice(c0->device));
  hipStream_t st = HaloStream(c0);
  uint32_t rows = 0;
  for (size_t i = 0; i < n; i++) {
    int r = HaloCheck(ctxs[i], side, dst, block_bytes, true, &rows);
    if (r) return r;
    if (ctxs[i]->device != c0->device) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (!rows) continue;
    // (the set's transform launch ran on this very stream: a frame's `pending_wait`, which names that launch, stays for its filter stage)
    if ((r = HaloCopy(ctxs[i], side == 0 ? ctxs[i]->band_y0 : ctxs[i]->band_y1 - rows, rows, static_cast<uint8_t*>(dst) + i * block_bytes, true, st))) return r;
  }
  if (!c0->halo_event) HIP_TRY(hipEventCreateWithFlags(&c0->halo_event, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(c0->halo_event, st));
  HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(transport_stream), c0->halo_event, 0));  // the transport reads the blocks after the copies
  return 0;
}
int jxlhip_halo_unpack_batch(JxlHipContext* const* ctxs, size_t n, int side, const void* src, size_t block_bytes, void* transport_stream) {
  if (!ctxs || !n || !ctxs[0]) return JXLHIP_ERR_INVALID_ARGUMENT;
  JxlHipContext* c0 = ctxs[0];
  HIP_TRY(hipSetDevice(c0->device));
  hipStream_t st = HaloStream(c0);
  // the copies read the blocks after whatever the transport's stream holds NOW (the receive the caller enqueued there)
  if (!c0->halo_event) HIP_TRY(hipEventCreateWithFlags(&c0->halo_event, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(c0->halo_event, static_cast<hipStream_t>(transport_stream)));
  HIP_TRY(hipStreamWaitEvent(st, c0->halo_event, 0));
  uint32_t rows = 0;
  for (size_t i = 0; i < n; i++) {
    int r = HaloCheck(ctxs[i], side, src, block_bytes, false, &rows);
    if (r) return r;
    if (ctxs[i]->device != c0->device) return JXLHIP_ERR_INVALID_ARGUMENT;
    if (!rows) continue;
    if ((r = HaloCopy(ctxs[i], side == 0 ? ctxs[i]->band_y0 - rows : ctxs[i]->band_y1, rows,
                      const_cast<uint8_t*>(static_cast<const uint8_t*>(src)) + i * block_bytes, false, st)))
      return r;
  }
  // the set's filter launch follows the copies: on this stream by stream order; on the second stream (option
  // "filter_async") it waits for `transform_done`, which is moved behind the copies here. The transport's stream must not
  // recycle the blocks before the copies have read them either.
  if (c0->filter_async && c0->transform_done_valid) HIP_TRY(hipEventRecord(c0->transform_done, st));
  HIP_TRY(hipEventRecord(c0->halo_event, st));
  HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(transport_stream), c0->halo_event, 0));
  return 0;
}

int jxlhip_set_option(JxlHipContext* c, const char* name, int value) {
  if (!c || !name) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (std::string(name) == "keep_filtered") {
    c->keep_filtered = value != 0;
    return 0;
  }
  if (std::string(name) == "band_halo") {  // (takes effect at the next upload of a band)
    c->band_halo = value != 0;
    return 0;
  }
  if (std::string(name) == "filter_async") {
    c->filter_async = value != 0;
    return 0;
  }
  if (std::string(name) == "entropy_gate") {
    // Set on the FIRST context of a frame set by callers that keep several batched launches in flight (bench.py's
    // pipeline): the set's batched entropy launches then count their workgroups in, and batched transform / filter
    // launches enqueued after one wait (a device-side wait packet) until it is resident: see EntropyGate. Off by default:
    // the wait has no bound, so it is only for callers that know every launch they enqueue can start.
    c->entropy_gate = value != 0;
    return 0;
  }
  if (std::string(name) == "keep_xyb_planes") {  // (takes effect at the next Modular upload)
    c->keep_xyb = value != 0;
    c->generation++;
    return 0;
  }
  if (std::string(name) == "blocking_sync") {
    // Host threads waiting for the device in this library's calls poll and sleep instead of spinning inside the runtime
    // (the whole process, not only this context): for hosts that pipeline many frames over more threads than they have
    // CPUs to spare.
    g_yield_waits.store(value != 0);
    return 0;
  }
  return JXLHIP_ERR_INVALID_ARGUMENT;
}

int jxlhip_set_output_format(JxlHipContext* c, uint32_t data_type, uint32_t num_channels, uint32_t bits_per_sample, int big_endian) {
  if (!c || num_channels < 1 || num_channels > 4) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (data_type != 0 && data_type != 2 && data_type != 3 && data_type != 5) return JXLHIP_ERR_INVALID_ARGUMENT;
  const uint32_t max_bits = data_type == 2 ? 8 : 16;
  if (bits_per_sample == 0) bits_per_sample = max_bits;
  if ((data_type == 2 || data_type == 3) && bits_per_sample > max_bits) return JXLHIP_ERR_INVALID_ARGUMENT;
  c->out_type = data_type;
  c->out_nc = num_channels;
  c->out_bits = bits_per_sample;
  c->out_swap = big_endian && data_type != 2 ? 1 : 0;
  c->generation++;  // cached batch descriptions hold the pixel pointers
  return 0;
}

int jxlhip_set_output_orientation(JxlHipContext* c, uint32_t orientation) {
  if (!c || orientation < 1 || orientation > 8) return JXLHIP_ERR_INVALID_ARGUMENT;
  // EXIF numbering -> mirror x (1) | mirror y (2) | transpose (4): stage_write.cc:441-458
  static const uint8_t kBits[9] = {0, 0, 1, 3, 2, 4, 6, 7, 5};
  c->out_orient = kBits[orientation];
  c->generation++;
  return 0;
}

int jxlhip_set_output_unpremultiply(JxlHipContext* c, int on) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  c->out_unpremul = on != 0;
  c->generation++;
  return 0;
}

int jxlhip_set_alpha(JxlHipContext* c, const float* alpha, uint32_t xsize, uint32_t ysize) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  c->alpha_patched_valid = false;
  if (!alpha) {
    c->have_alpha = false;
    return 0;
  }
  HIP_TRY(hipSetDevice(c->device));
  const size_t bytes = size_t(xsize) * ysize * 4;
  int r = c->alpha.Ensure(bytes);
  if (r) return r;
  HIP_TRY(hipMemcpy(c->alpha.p, alpha, bytes, hipMemcpyHostToDevice));
  c->have_alpha = true;
  return 0;
}

int jxlhip_download_alpha(JxlHipContext* c, float* dst) {
  if (!c || !dst) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_alpha) return JXLHIP_ERR_NO_FRAME;
  HIP_TRY(hipSetDevice(c->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
  }
  const Buf& a = c->alpha_patched_valid ? c->alpha_patched : c->alpha;
  HIP_TRY(hipMemcpyAsync(dst, a.p, size_t(c->oxs) * c->oys * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int jxlhip_upsample_plane(JxlHipContext* c, const float* plane, uint32_t xsize, uint32_t ysize, uint32_t factor, const float* kernels,
                          uint32_t out_xsize, uint32_t out_ysize, int as_alpha, float* host_out) {
  if (!c || !plane || !kernels || !xsize || !ysize || (factor != 2 && factor != 4 && factor != 8)) return JXLHIP_ERR_INVALID_ARGUMENT;
  if ((uint64_t(out_xsize) + factor - 1) / factor != xsize || (uint64_t(out_ysize) + factor - 1) / factor != ysize || !out_xsize || !out_ysize)
    return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
  }
  const size_t in_bytes = size_t(xsize) * ysize * 4, k_bytes = size_t(factor) * factor * 25 * 4, out_bytes = size_t(out_xsize) * out_ysize * 4;
  int r;
  // (the staging buffer: the coded plane, then the kernels; the result goes to the alpha plane or behind them)
  const size_t k_at = (in_bytes + 255) & ~size_t(255), out_at = (k_at + k_bytes + 255) & ~size_t(255);
  if ((r = c->ec_stage.Ensure(out_at + (as_alpha ? 0 : out_bytes)))) return r;
  if (as_alpha && (r = c->alpha.Ensure(out_bytes))) return r;
  uint8_t* base = static_cast<uint8_t*>(c->ec_stage.p);
  HIP_TRY(hipMemcpyAsync(base, plane, in_bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(base + k_at, kernels, k_bytes, hipMemcpyHostToDevice, c->stream));
  jxlhip::UpsamplePlaneParams up;
  up.in = reinterpret_cast<const float*>(base);
  up.kernel = reinterpret_cast<const float*>(base + k_at);
  up.out = as_alpha ? c->alpha.as<float>() : reinterpret_cast<float*>(base + out_at);
  up.xs = xsize;
  up.ys = ysize;
  up.n = factor;
  up.oxs = out_xsize;
  up.oys = out_ysize;
  hipLaunchKernelGGL(jxlhip::k_upsample_plane, dim3((xsize + 63) / 64, (ysize + 3) / 4), dim3(256), 0, c->stream, up);
  HIP_TRY(hipGetLastError());
  if (host_out) HIP_TRY(hipMemcpyAsync(host_out, up.out, out_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // (the host buffers are the caller's; pageable copies are staged by the runtime)
  if (as_alpha) c->have_alpha = true;
  return 0;
}

int jxlhip_download_pixels(JxlHipContext* c, void* dst, size_t stride) {
  if (!c || !dst) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return JXLHIP_ERR_NO_FRAME;
  const bool transposed = (c->out_orient & 4) != 0;  // rows of the output are columns of the image
  const size_t row = size_t(transposed ? c->oys : c->oxs) * OutPixelBytes(c), rows = transposed ? c->oxs : c->oys;
  if (stride < row) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
  }
  HIP_TRY(hipMemcpy2DAsync(dst, stride, c->rgb.p, row, row, rows, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int jxlhip_run_all(JxlHipContext* c) {
  int r = jxlhip_run_entropy(c);
  if (!r) r = jxlhip_run_transform(c);
  if (!r) r = jxlhip_run_filter_color(c);
  return r;
}

// Test entry: the colour stage alone (k_color_out: XYB -> linear RGB -> transfer function) on `n` caller-supplied XYB
// triples, planar [3][n], with the opsin parameters of the frame the context last uploaded; f32 RGB out, interleaved.
// For the closed-form colour tests of the reference (opsin_image_test.cc) against the kernel itself.
int jxlhip_debug_color(JxlHipContext* c, const float* xyb, size_t n, int linear_output, float* rgb) {
  if (!c || !xyb || !rgb || !n || n > (1u << 24)) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return JXLHIP_ERR_NO_FRAME;
  HIP_TRY(hipSetDevice(c->device));
  Buf in, out;
  int r;
  if ((r = in.Ensure(n * 12)) || (r = out.Ensure(n * 12))) {
    in.Free();
    out.Free();
    return r;
  }
  hipError_t e = hipMemcpy(in.p, xyb, n * 12, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    jxlhip::ColorOutParams cp;
    memset(&cp, 0, sizeof(cp));
    cp.f = c->fp;
    cp.f.in = in.as<float>();
    cp.f.xs = cp.f.xp = uint32_t(n);
    cp.f.ys = cp.f.yp = 1;
    cp.f.y_begin = 0;
    cp.f.y_end = 1;
    cp.f.linear_output = linear_output;
    cp.po.dst = out.p;
    cp.po.xsize = uint32_t(n);
    cp.po.ysize = 1;
    cp.po.type = 0;
    cp.po.nc = 3;
    cp.po.bits = 32;
    hipLaunchKernelGGL(jxlhip::k_color_out, dim3(uint32_t((n + 63) / 64), 1), dim3(256), 0, c->stream, cp);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(rgb, out.p, n * 12, hipMemcpyDeviceToHost);
  }
  in.Free();
  out.Free();
  return e == hipSuccess ? 0 : -int(e);
}

int jxlhip_check_guards(JxlHipContext* c, uint32_t* touched) {
  if (!c || !touched) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  *touched = 0;
  uint32_t index = 0;
  for (Buf* b : AllBufs(c)) {
    index++;
    const int t = b->GuardsTouched();
    if (t && !*touched) *touched = index << 2 | uint32_t(t);  // which buffer (1-based position in AllBufs), which side
  }
  return 0;
}

int jxlhip_sync(JxlHipContext* c) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
  }
  HIP_TRY(WaitStream(c->stream));
  return 0;
}

int jxlhip_download_rgb8(JxlHipContext* c, uint8_t* dst, size_t stride) {
  if (!c || !dst) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return JXLHIP_ERR_NO_FRAME;
  if (stride < size_t(c->oxs) * 3 || !OutIsRgb8(c)) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
  }
  HIP_TRY(hipMemcpy2DAsync(dst, stride, c->rgb.p, size_t(c->oxs) * 3, size_t(c->oxs) * 3, c->oys, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(WaitStream(c->stream));
  return 0;
}

int jxlhip_download_rgb8_rows(JxlHipContext* c, uint8_t* dst, size_t stride, uint32_t y_begin, uint32_t y_end) {
  if (!c || !dst) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return JXLHIP_ERR_NO_FRAME;
  if (stride < size_t(c->oxs) * 3 || y_begin >= y_end || y_end > c->oys || !OutIsRgb8(c)) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (c->out_orient & 4) return JXLHIP_ERR_INVALID_ARGUMENT;  // a transposed output has other rows: jxlhip_download_pixels
  HIP_TRY(hipSetDevice(c->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
  }
  HIP_TRY(hipMemcpy2DAsync(dst, stride, c->rgb.as<uint8_t>() + size_t(y_begin) * c->oxs * 3, size_t(c->oxs) * 3, size_t(c->oxs) * 3,
                           y_end - y_begin, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

const uint8_t* jxlhip_rgb8_device_ptr(JxlHipContext* c) { return c && c->have_frame ? c->rgb.as<uint8_t>() : nullptr; }

int jxlhip_get_errors(JxlHipContext* c, uint32_t* flags, size_t n) {
  if (!c || !flags) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return JXLHIP_ERR_NO_FRAME;
  if (n < c->ng) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
  }
  HIP_TRY(hipMemcpyAsync(flags, c->errors.p, size_t(c->ng) * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (uint32_t g = 0; g < c->ng; g++)
    if (flags[g]) return JXLHIP_ERR_STREAM;
  return 0;
}

int jxlhip_get_section_end_bits(JxlHipContext* c, uint32_t* bits, size_t n) {
  if (!c || !bits) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return JXLHIP_ERR_NO_FRAME;
  if (n < size_t(c->ng) * c->np) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
  }
  HIP_TRY(hipMemcpyAsync(bits, c->sec_end.p, size_t(c->ng) * c->np * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int jxlhip_download(JxlHipContext* c, const char* name, void* dst, size_t dst_size, size_t* needed) {
  if (!c || !name) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return JXLHIP_ERR_NO_FRAME;
  const void* src = nullptr;
  size_t bytes = 0;
  const std::string n(name);
  const size_t plane_bytes = size_t(c->xp) * c->yp * 3 * 4;
  if (n == "coeffs") {
    src = c->coeffs.p;
    bytes = size_t(c->ng) * 3 * 65536 * (c->coef_bits / 8);
  } else if (n == "xyb_idct") {
    src = PlaneHolder(c)->plane[0].p;
    bytes = plane_bytes;
  } else if (n == "dc") {  // the DC image the transform stage reads (after smoothing): [3][yb][xb]
    src = c->dc.p;
    bytes = size_t(c->xb) * c->yb * 12;
  } else if (n == "inv_sigma") {
    src = c->inv_sigma.p;
    bytes = size_t(c->xb) * c->yb * 4;
  } else if (n == "xyb_filtered") {
    if (!c->keep_filtered || !c->plane[1].p) return JXLHIP_ERR_INVALID_ARGUMENT;  // needs jxlhip_set_option("keep_filtered", 1)
    src = c->plane[1].p;
    bytes = plane_bytes;
  } else {
    return JXLHIP_ERR_INVALID_ARGUMENT;
  }
  if (needed) *needed = bytes;
  if (!dst) return 0;
  if (dst_size < bytes) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
  }
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (n == "coeffs" && c->scan_order) {
    // the device holds scan-order entries [covered, kend) per (block, channel): present them in the natural layout
    std::vector<uint32_t> kend(c->blocks_host.size() * 3);
    HIP_TRY(hipMemcpy(kend.data(), c->kend.p, kend.size() * 4, hipMemcpyDeviceToHost));
    static const uint8_t log2c[27] = {0, 0, 0, 0, 2, 4, 1, 1, 2, 2, 3, 3, 0, 0, 0, 0, 0, 0, 6, 5, 5, 8, 7, 7, 10, 9, 9};
    static const uint8_t bucket[27] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
    const size_t esz = c->coef_bits / 8;
    std::vector<uint8_t> tmp(65536 * esz);
    for (uint32_t g = 0; g < c->ng; g++)
      for (uint32_t b = c->gbb_host[g]; b < c->gbb_host[g + 1]; b++) {
        const JxlHipVarBlock& v = c->blocks_host[b];
        const uint32_t covered = 1u << log2c[v.strategy], size = covered * 64;
        for (int ch = 0; ch < 3; ch++) {
          uint8_t* base = static_cast<uint8_t*>(dst) + ((size_t(g) * 3 + ch) * 65536 + v.coef_offset) * esz;
          const uint16_t* order = c->orders_host.data() + c->order_offset_host[bucket[v.strategy] * 3 + ch];
          uint32_t ke = kend[size_t(b) * 3 + ch];
          ke = ke > size ? size : ke;
          memcpy(tmp.data(), base, size * esz);
          memset(base, 0, size * esz);
          for (uint32_t k = covered; k < ke; k++) memcpy(base + size_t(order[k]) * esz, tmp.data() + size_t(k) * esz, esz);
        }
      }
  }
  return 0;
}

// ---- canvas (jxl_hip_canvas.h)
struct JxlHipCanvas {
  int device = 0;
  uint32_t xs = 0, ys = 0;
  bool has_alpha = false, premultiplied = false, unpremul_out = false;
  Buf cur, slot[4], pixels;
  bool slot_valid[4] = {false, false, false, false};
  // frames kept before the colour transform, [3][xyb_h][xyb_w]: 0..3 the reference slots (patch sources), 4..7 the DC
  // frames of level 1..4 (passes_state.h:90 dc_frames: the DC image of a later frame with kUseDcFrame)
  Buf xyb[8];
  uint32_t xyb_w[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xyb_h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool xyb_alpha[8] = {false, false, false, false, false, false, false, false};  // a fourth plane: the frame's alpha
  hipStream_t last_stream = nullptr;  // the stream of the last blend: later work on the canvas is ordered behind it
};

int jxlhip_canvas_set_unpremultiply(JxlHipCanvas* v, int on) {
  if (!v) return JXLHIP_ERR_INVALID_ARGUMENT;
  v->unpremul_out = on != 0;
  return 0;
}

int jxlhip_canvas_create(int device, uint32_t xsize, uint32_t ysize, uint32_t has_alpha, uint32_t premultiplied, JxlHipCanvas** out) {
  if (!out || !xsize || !ysize || xsize > (1u << 18) || ysize > (1u << 18)) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(device));
  JxlHipCanvas* v = new (std::nothrow) JxlHipCanvas;
  if (!v) return JXLHIP_ERR_INVALID_ARGUMENT;
  v->device = device;
  v->xs = xsize;
  v->ys = ysize;
  v->has_alpha = has_alpha != 0;
  v->premultiplied = premultiplied != 0;
  int r = v->cur.Ensure(size_t(xsize) * ysize * 16);
  if (r) {
    delete v;
    return r;
  }
  *out = v;
  return 0;
}

void jxlhip_canvas_destroy(JxlHipCanvas* v) {
  if (!v) return;
  (void)hipSetDevice(v->device);
  if (v->last_stream) (void)hipStreamSynchronize(v->last_stream);
  v->cur.Free();
  v->pixels.Free();
  for (Buf& b : v->slot) b.Free();
  for (Buf& b : v->xyb) b.Free();
  delete v;
}

int jxlhip_canvas_save_xyb(JxlHipCanvas* v, JxlHipContext* c, uint32_t slot) {
  if (!v || !c || slot > 7 || c->device != v->device) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(v->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
  }
  const float* src;
  size_t src_stride, src_plane;
  uint32_t w, h;
  if (c->mod.have) {  // a Modular frame (a later VarDCT upload clears the flag): the float planes of the "keep_xyb_planes" pass
    if (!c->keep_xyb || !c->spl_planes.p) return JXLHIP_ERR_INVALID_ARGUMENT;
    w = c->mod.xs;
    h = c->mod.ys;
    src = c->spl_planes.as<float>();
    src_stride = w;
    src_plane = size_t(w) * h;
  } else if (c->have_frame) {  // a VarDCT frame: the filtered planes (patches, splines, noise already on them)
    if (c->ups != 1 || !c->color_out || !c->plane[1].p) return JXLHIP_ERR_UNSUPPORTED;
    w = c->xs;
    h = c->ys;
    src = c->plane[1].as<float>();
    src_stride = c->xp;
    src_plane = size_t(c->xp) * c->yp;
  } else {
    return JXLHIP_ERR_NO_FRAME;
  }
  // (the frame's extra channels are kept with it, for patches that blend through alpha: dec_patch_dictionary.cc:342-347)
  const bool mod_alpha = c->mod.have && c->mod.has_alpha;
  const bool with_alpha = mod_alpha || (!c->mod.have && c->have_alpha && c->alpha.cap >= size_t(w) * h * 4);
  int r = v->xyb[slot].Ensure(size_t(w) * h * (with_alpha ? 16 : 12));
  if (r) return r;
  if (v->last_stream && v->last_stream != c->stream) HIP_TRY(hipStreamSynchronize(v->last_stream));
  hipLaunchKernelGGL(jxlhip::k_copy_planes, dim3((w + 255) / 256, h, 3), dim3(256), 0, c->stream, src, src_stride, src_plane,
                     v->xyb[slot].as<float>(), w, h);
  HIP_TRY(hipGetLastError());
  if (mod_alpha) {
    const JxlHipContext::Modular& M = c->mod;
    const size_t n = size_t(w) * h;
    hipLaunchKernelGGL(jxlhip::k_int_plane_to_float, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, c->stream,
                       M.pool.as<int32_t>() + M.buf_off[M.out_buffer[M.num_color]], v->xyb[slot].as<float>() + n * 3, n,
                       1.0f / float((uint64_t(1) << M.alpha_bits) - 1));
    HIP_TRY(hipGetLastError());
  } else if (with_alpha)
    HIP_TRY(hipMemcpyAsync(v->xyb[slot].as<float>() + size_t(w) * h * 3, (c->alpha_patched_valid ? c->alpha_patched : c->alpha).p,
                           size_t(w) * h * 4, hipMemcpyDeviceToDevice, c->stream));
  v->xyb_alpha[slot] = with_alpha;
  HIP_TRY(hipStreamSynchronize(c->stream));  // (later frames read the slot from other streams of other contexts)
  v->xyb_w[slot] = w;
  v->xyb_h[slot] = h;
  if (slot < 4) v->slot_valid[slot] = false;  // (the slot now holds a frame saved BEFORE the colour transform: not a blending source)
  v->last_stream = c->stream;
  return 0;
}

int jxlhip_canvas_xyb_source(JxlHipCanvas* v, uint32_t slot, const float** planes, uint32_t* xsize, uint32_t* ysize) {
  if (!v || slot > 7 || !planes || !xsize || !ysize) return JXLHIP_ERR_INVALID_ARGUMENT;
  *planes = v->xyb_w[slot] ? v->xyb[slot].as<float>() : nullptr;
  *xsize = v->xyb_w[slot];
  *ysize = v->xyb_h[slot];
  return 0;
}

int jxlhip_canvas_xyb_alpha(JxlHipCanvas* v, uint32_t slot, const float** alpha) {
  if (!v || slot > 7 || !alpha) return JXLHIP_ERR_INVALID_ARGUMENT;
  *alpha = (v->xyb_w[slot] && v->xyb_alpha[slot]) ? v->xyb[slot].as<float>() + size_t(v->xyb_w[slot]) * v->xyb_h[slot] * 3 : nullptr;
  return 0;
}

int jxlhip_canvas_blend(JxlHipCanvas* v, JxlHipContext* c, const JxlHipBlend* b) {
  if (!v || !c || !b || c->device != v->device) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (b->mode > 4 || b->alpha_mode > 4 || b->source > 3 || b->alpha_source > 3 || b->save_slot > 3) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame && !c->mod.have) return JXLHIP_ERR_NO_FRAME;
  if (c->out_type != 0 || c->out_nc != 4 || c->out_swap || c->out_orient) return JXLHIP_ERR_INVALID_ARGUMENT;  // f32 x 4 as coded
  if (size_t(c->oxs) * c->oys * 16 > c->rgb.cap) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(v->device));
  {
    int pw = ApplyPendingWait(c);
    if (pw) return pw;
  }
  const size_t bytes = size_t(v->xs) * v->ys * 16;
  jxlhip::BlendParams P;
  memset(&P, 0, sizeof(P));
  P.out = v->cur.as<float>();
  P.bg_color = v->slot_valid[b->source] ? v->slot[b->source].as<float>() : nullptr;
  P.bg_alpha = v->slot_valid[b->alpha_source] ? v->slot[b->alpha_source].as<float>() : nullptr;
  P.fg = c->rgb.as<float>();
  P.w = v->xs;
  P.h = v->ys;
  P.fw = c->oxs;
  P.fh = c->oys;
  P.x0 = b->x0;
  P.y0 = b->y0;
  P.mode = b->mode;
  P.alpha_mode = b->alpha_mode;
  P.clamp = b->clamp;
  P.alpha_clamp = b->alpha_clamp;
  P.has_alpha = v->has_alpha;
  P.premultiplied = v->premultiplied;
  if (v->last_stream && v->last_stream != c->stream) HIP_TRY(hipStreamSynchronize(v->last_stream));
  hipLaunchKernelGGL(jxlhip::k_canvas_blend, dim3((v->xs + 255) / 256, v->ys), dim3(256), 0, c->stream, P);
  HIP_TRY(hipGetLastError());
  if (b->save_slot >= 0) {
    int r = v->slot[b->save_slot].Ensure(bytes);
    if (r) return r;
    HIP_TRY(hipMemcpyAsync(v->slot[b->save_slot].p, v->cur.p, bytes, hipMemcpyDeviceToDevice, c->stream));
    v->slot_valid[b->save_slot] = true;
    // the slot now holds a frame saved AFTER the colour transform: patches cannot take it (dec_patch_dictionary.cc:66-73)
    v->xyb_w[b->save_slot] = v->xyb_h[b->save_slot] = 0;
  }
  v->last_stream = c->stream;
  return 0;
}

int jxlhip_debug_blend(int device, const float* bg, const float* fg, size_t n, const JxlHipBlend* b, uint32_t has_alpha, uint32_t premultiplied,
                       float* out) {
  if (!bg || !fg || !b || !out || !n || n > (1u << 20) || b->mode > 4 || b->alpha_mode > 4) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(device));
  Buf dbg, dfg, dout;
  int r;
  if ((r = dbg.Ensure(n * 16)) || (r = dfg.Ensure(n * 16)) || (r = dout.Ensure(n * 16))) {
    dbg.Free();
    dfg.Free();
    dout.Free();
    return r;
  }
  hipError_t e = hipMemcpy(dbg.p, bg, n * 16, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dfg.p, fg, n * 16, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    jxlhip::BlendParams P;
    memset(&P, 0, sizeof(P));
    P.out = dout.as<float>();
    P.bg_color = dbg.as<float>();
    P.bg_alpha = dbg.as<float>();
    P.fg = dfg.as<float>();
    P.w = P.fw = uint32_t(n);
    P.h = P.fh = 1;
    P.mode = b->mode;
    P.alpha_mode = b->alpha_mode;
    P.clamp = b->clamp;
    P.alpha_clamp = b->alpha_clamp;
    P.has_alpha = has_alpha;
    P.premultiplied = premultiplied;
    hipLaunchKernelGGL(jxlhip::k_canvas_blend, dim3((uint32_t(n) + 255) / 256, 1), dim3(256), 0, nullptr, P);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, dout.p, n * 16, hipMemcpyDeviceToHost);
  }
  dbg.Free();
  dfg.Free();
  dout.Free();
  return e == hipSuccess ? 0 : -int(e);
}

int jxlhip_canvas_download(JxlHipCanvas* v, uint32_t data_type, uint32_t num_channels, uint32_t bits, int big_endian, uint32_t orientation,
                           void* dst, size_t stride) {
  if (!v || !dst || num_channels < 1 || num_channels > 4 || orientation < 1 || orientation > 8) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (data_type != 0 && data_type != 2 && data_type != 3 && data_type != 5) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(v->device));
  static const uint8_t kBits[9] = {0, 0, 1, 3, 2, 4, 6, 7, 5};
  const uint32_t sample = data_type == 2 ? 1 : (data_type == 0 ? 4 : 2), full = sample * 8;
  jxlhip::PixelOut po;
  memset(&po, 0, sizeof(po));
  int r = v->pixels.Ensure(size_t(v->xs) * v->ys * num_channels * sample);
  if (r) return r;
  po.dst = v->pixels.p;
  po.alpha = v->has_alpha ? v->cur.as<float>() + size_t(3) * v->xs * v->ys : nullptr;
  po.xsize = v->xs;
  po.ysize = v->ys;
  po.orient = kBits[orientation] | ((v->unpremul_out && v->has_alpha) ? 8u : 0u);
  po.type = data_type;
  po.nc = num_channels;
  po.bits = (data_type == 2 || data_type == 3) ? (bits && bits < full ? bits : full) : 0;
  po.swap = big_endian && sample > 1;
  hipStream_t st = v->last_stream;
  hipLaunchKernelGGL(jxlhip::k_canvas_out, dim3((v->xs + 255) / 256, v->ys), dim3(256), 0, st, static_cast<const float*>(v->cur.as<float>()), po);
  HIP_TRY(hipGetLastError());
  const bool transposed = (po.orient & 4) != 0;
  const size_t row = size_t(transposed ? v->ys : v->xs) * num_channels * sample, rows = transposed ? v->xs : v->ys;
  if (stride < row) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipMemcpy2DAsync(dst, stride, v->pixels.p, row, row, rows, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return 0;
}

int jxlhip_canvas_download_alpha(JxlHipCanvas* v, float* dst, size_t n) {
  if (!v || !dst || n < size_t(v->xs) * v->ys) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(v->device));
  HIP_TRY(hipMemcpyAsync(dst, v->cur.as<float>() + size_t(3) * v->xs * v->ys, size_t(v->xs) * v->ys * 4, hipMemcpyDeviceToHost, v->last_stream));
  HIP_TRY(hipStreamSynchronize(v->last_stream));
  return 0;
}

// ---- forward path: see jxl_hip_enc.h
// The kernel sequence of the forward path on the context's stream (P.planes is set on the way).
static int EncLaunch(JxlHipContext* c, jxlhip::EncFwd& P, bool gaborish) {
  const size_t nb = size_t(P.xb) * P.yb, ng = size_t(P.xg) * P.yg;
  // colour: into planes[0] when there is no sharpening, else into the `orig` set
  float* sets[3] = {c->enc_planes[0].as<float>(), c->enc_planes[1].as<float>(), c->enc_planes[2].as<float>()};
  const dim3 px_grid((P.xp + 255) / 256, P.yp);
  P.planes = gaborish ? sets[2] : sets[0];
  hipLaunchKernelGGL(jxlhip::k_enc_xyb, px_grid, dim3(256), 0, c->stream, P);
  if (gaborish) {
    if (getenv("JXLHIP_ENC_SHARPEN_ROUNDS")) {  // the one-round-per-launch form: orig = sets[2]; 2 -> 0 -> 1 -> 0 -> 1
      const dim3 g3(px_grid.x, px_grid.y, 3);
      const float* in = sets[2];
      float* out = sets[0];
      for (int it = 0; it < 4; it++) {
        hipLaunchKernelGGL(jxlhip::k_enc_sharpen, g3, dim3(256), 0, c->stream, static_cast<const float*>(sets[2]), in, out, P.xp, P.yp);
        in = out;
        out = out == sets[0] ? sets[1] : sets[0];
      }
      P.planes = const_cast<float*>(in);
    } else if (getenv("JXLHIP_ENC_SHARPEN_TILE")) {  // the LDS-tile form of the four rounds (the row form is held against it bit for bit)
      hipLaunchKernelGGL(jxlhip::k_enc_sharpen4, dim3((P.xp + 63) / 64, (P.yp + 31) / 32, 3), dim3(256), 0, c->stream,
                         static_cast<const float*>(sets[2]), sets[0], P.xp, P.yp);
      P.planes = sets[0];
    } else {
      const uint32_t strips = (P.xp + jxlhip::kSharpenCols - 1) / jxlhip::kSharpenCols;
      hipLaunchKernelGGL(jxlhip::k_enc_sharpen_rows, dim3((strips + 3) / 4, (P.yp + jxlhip::kSharpenRows - 1) / jxlhip::kSharpenRows, 3), dim3(256), 0,
                         c->stream, static_cast<const float*>(sets[2]), sets[0], P.xp, P.yp);
      P.planes = sets[0];
    }
  }
  hipLaunchKernelGGL(jxlhip::k_enc_activity, dim3((P.xb + 7) / 8, P.yb), dim3(64), 0, c->stream, P);
  const uint32_t tiles = ((P.xb + 7) / 8) * ((P.yb + 7) / 8);
  hipLaunchKernelGGL(jxlhip::k_enc_select, dim3(tiles), dim3(64), 0, c->stream, P);
  hipLaunchKernelGGL(jxlhip::k_enc_offsets, dim3(uint32_t(ng)), dim3(64), 0, c->stream, P);
  HIP_TRY(hipMemsetAsync(c->enc_coef.p, 0, ng * 3 * 65536 * 4, c->stream));
  HIP_TRY(hipEventRecord(c->enc_ev[2], c->stream));
  if (getenv("JXLHIP_ENC_BLOCK_KERNEL"))  // the one-workgroup-per-transform form (kept as the simple statement of the algorithm)
    hipLaunchKernelGGL(jxlhip::k_enc_transform<256>, dim3(uint32_t(nb)), dim3(256), 0, c->stream, P);
  else
    hipLaunchKernelGGL(jxlhip::k_enc_transform_tile, dim3(tiles), dim3(256), 0, c->stream, P);
  HIP_TRY(hipEventRecord(c->enc_ev[3], c->stream));
  HIP_TRY(hipGetLastError());
  return 0;
}

int jxlhip_enc_forward(JxlHipContext* c, const uint8_t* rgb, size_t stride, const JxlHipEncDesc* d, uint8_t* acs, int32_t* qf, int32_t* dc,
                       int32_t* coeffs) {
  if (!c || !rgb || !d || !acs || !qf || !dc) return JXLHIP_ERR_INVALID_ARGUMENT;  // (coeffs may be NULL: they stay on the device)
  if (!d->xsize || !d->ysize || d->xsize > (1u << 18) || d->ysize > (1u << 18) || stride < size_t(d->xsize) * 3) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!(d->distance > 0) || !d->global_scale || !d->quant_dc || !d->dequant || d->strategy_mode > 1) return JXLHIP_ERR_INVALID_ARGUMENT;
  for (int k = 0; k < 17; k++)
    if (size_t(d->dequant_offset[k]) + 3 * size_t(d->dequant_size[k]) > d->dequant_floats) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  jxlhip::EncFwd P;
  memset(&P, 0, sizeof(P));
  P.xs = d->xsize;
  P.ys = d->ysize;
  P.xb = (P.xs + 7) / 8;
  P.yb = (P.ys + 7) / 8;
  P.xp = P.xb * 8;
  P.yp = P.yb * 8;
  P.xg = (P.xs + 255) / 256;
  P.yg = (P.ys + 255) / 256;
  const size_t nb = size_t(P.xb) * P.yb, plane = size_t(P.xp) * P.yp, ng = size_t(P.xg) * P.yg;
  int r;
  if ((r = c->enc_rgb.Ensure(stride * P.ys)) || (r = c->enc_act.Ensure(nb * 4)) || (r = c->enc_acs.Ensure(nb)) || (r = c->enc_qf.Ensure(nb * 4)) ||
      (r = c->enc_off.Ensure(nb * 4)) || (r = c->enc_dc.Ensure(3 * nb * 4)) || (r = c->enc_coef.Ensure(ng * 3 * 65536 * 4)) ||
      (r = c->enc_lut.Ensure(256 * 4)) || (r = c->enc_dq.Ensure(size_t(d->dequant_floats) * 4)))
    return r;
  const size_t ntiles = size_t((P.xb + 7) / 8) * ((P.yb + 7) / 8);
  if ((r = c->enc_ytox.Ensure(ntiles)) || (r = c->enc_ytob.Ensure(ntiles))) return r;
  if (d->cfl_fit && getenv("JXLHIP_ENC_BLOCK_KERNEL")) return JXLHIP_ERR_UNSUPPORTED;  // (the tile kernel makes the fit)
  for (auto& b : c->enc_planes)
    if ((r = b.Ensure(3 * plane * 4))) return r;
  for (auto& ev : c->enc_ev)
    if (!ev) HIP_TRY(hipEventCreate(&ev));
  float lut[256];
  for (int i = 0; i < 256; i++) {
    const float v = float(i) / 255.0f;
    lut[i] = v <= 0.04045f ? v / 12.92f : std::pow((v + 0.055f) / 1.055f, 2.4f);  // sRGB EOTF (transfer_functions-inl.h TF_SRGB)
  }
  HIP_TRY(hipMemcpyAsync(c->enc_lut.p, lut, sizeof(lut), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->enc_dq.p, d->dequant, size_t(d->dequant_floats) * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->enc_rgb.p, rgb, stride * P.ys, hipMemcpyHostToDevice, c->stream));
  P.rgb = c->enc_rgb.as<uint8_t>();
  P.rgb_stride = stride;
  P.srgb_lut = c->enc_lut.as<float>();
  P.act = c->enc_act.as<float>();
  P.acs = c->enc_acs.as<uint8_t>();
  P.qf = c->enc_qf.as<int32_t>();
  P.coef_off = c->enc_off.as<uint32_t>();
  constexpr size_t kBasisFloats = 87381;
  P.basis_t = c->basis.as<float>() + kBasisFloats + 3;
  P.dequant = c->enc_dq.as<float>();
  memcpy(P.dq_offset, d->dequant_offset, sizeof(P.dq_offset));
  memcpy(P.dq_size, d->dequant_size, sizeof(P.dq_size));
  P.dc = c->enc_dc.as<int32_t>();
  P.coeffs = c->enc_coef.as<int32_t>();
  P.distance = d->distance;
  P.quant_ac = d->quant_ac;
  P.inv_gs = 65536.0f / float(d->global_scale);
  P.x_dm = std::pow(1.25f, 2.0f - 3.0f);  // x_qm_scale 3, b_qm_scale 2 (dec_cache.h:161-162)
  P.b_dm = std::pow(1.25f, 2.0f - 2.0f);
  const float inv_quant_dc = P.inv_gs / float(d->quant_dc);
  P.dc_step[0] = inv_quant_dc / 4096.0f;
  P.dc_step[1] = inv_quant_dc / 512.0f;
  P.dc_step[2] = inv_quant_dc / 256.0f;
  P.strategy_mode = d->strategy_mode;
  P.cfl_fit = d->cfl_fit ? 1 : 0;
  P.scale = float(d->global_scale) / 65536.0f;
  P.ytox = c->enc_ytox.as<int8_t>();
  P.ytob = c->enc_ytob.as<int8_t>();
  HIP_TRY(hipMemsetAsync(c->enc_ytox.p, 0, ntiles, c->stream));
  HIP_TRY(hipMemsetAsync(c->enc_ytob.p, 0, ntiles, c->stream));
  HIP_TRY(hipEventRecord(c->enc_ev[0], c->stream));
  if ((r = EncLaunch(c, P, d->gaborish != 0))) return r;
  c->enc_last = P;
  c->enc_last_gaborish = d->gaborish != 0;
  HIP_TRY(hipEventRecord(c->enc_ev[1], c->stream));
  c->enc_timed = true;
  if (d->ytox) HIP_TRY(hipMemcpyAsync(d->ytox, c->enc_ytox.p, ntiles, hipMemcpyDeviceToHost, c->stream));
  if (d->ytob) HIP_TRY(hipMemcpyAsync(d->ytob, c->enc_ytob.p, ntiles, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(acs, c->enc_acs.p, nb, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(qf, c->enc_qf.p, nb * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(dc, c->enc_dc.p, 3 * nb * 4, hipMemcpyDeviceToHost, c->stream));
  if (coeffs) HIP_TRY(hipMemcpyAsync(coeffs, c->enc_coef.p, ng * 3 * 65536 * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->enc_tok_ready = false;
  return 0;
}

// ---- tokenisation of the last forward call's coefficients (jxl_hip_enc.h)
int jxlhip_enc_token_counts(JxlHipContext* c, const JxlHipEncTokDesc* d, uint32_t* totals) {
  if (!c || !d || !totals || !d->orders || !d->num_ctxs || !d->num_hist || d->num_ctxs > 16) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->enc_timed) return JXLHIP_ERR_NO_FRAME;
  // every index the kernels form from the tables must stay inside them: the order buckets of the transforms the forward
  // path selects (up to 64x64), complete and with entries below their size; block contexts below num_ctxs
  static const uint32_t kBucketSize[9] = {64, 64, 256, 1024, 128, 256, 512, 4096, 2048};
  for (int b = 0; b < 9; b++) {
    if (uint64_t(d->order_offset[b]) + kBucketSize[b] > d->orders_size) return JXLHIP_ERR_INVALID_ARGUMENT;
    for (uint32_t k = 0; k < kBucketSize[b]; k++)
      if (d->orders[d->order_offset[b] + k] >= kBucketSize[b]) return JXLHIP_ERR_INVALID_ARGUMENT;
  }
  for (int i = 0; i < 39; i++)
    if (d->ctx_map[i] >= d->num_ctxs) return JXLHIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  const jxlhip::EncFwd& F = c->enc_last;
  const size_t ng = size_t(F.xg) * F.yg;
  int r;
  if ((r = c->enc_tok_orders.Ensure(size_t(d->orders_size) * 2 + 16)) || (r = c->enc_tok_blk.Ensure(ng * 1024 * 2)) ||
      (r = c->enc_tok_info.Ensure(ng * jxlhip::kTokPerGroup * 4)) || (r = c->enc_tok_off.Ensure(ng * jxlhip::kTokPerGroup * 4)) ||
      (r = c->enc_tok_nzmap.Ensure(ng * 3 * 1024)) || (r = c->enc_tok_small.Ensure(ng * 12)))
    return r;
  HIP_TRY(hipMemcpyAsync(c->enc_tok_orders.p, d->orders, size_t(d->orders_size) * 2, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(c->enc_tok_nzmap.p, 0, ng * 3 * 1024, c->stream));
  jxlhip::EncTok& T = c->enc_tok;
  memset(&T, 0, sizeof(T));
  T.acs = F.acs;
  T.coef_off = F.coef_off;
  T.coeffs = F.coeffs;
  T.xb = F.xb;
  T.yb = F.yb;
  T.xg = F.xg;
  T.orders = c->enc_tok_orders.as<uint16_t>();
  memcpy(T.order_offset, d->order_offset, sizeof(T.order_offset));
  memcpy(T.ctx_map, d->ctx_map, sizeof(T.ctx_map));
  T.num_ctxs = d->num_ctxs;
  T.num_hist = d->num_hist;
  T.nctx = d->num_ctxs * 495;
  T.blk = c->enc_tok_blk.as<uint16_t>();
  T.info = c->enc_tok_info.as<uint32_t>();
  T.off = c->enc_tok_off.as<uint32_t>();
  T.nzmap = c->enc_tok_nzmap.as<uint8_t>();
  T.nblk = c->enc_tok_small.as<uint32_t>();
  T.total = c->enc_tok_small.as<uint32_t>() + ng;
  hipLaunchKernelGGL(jxlhip::k_enc_tok_count, dim3(uint32_t(ng)), dim3(jxlhip::kTokThreads), 0, c->stream, T);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(totals, T.total, ng * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->enc_tok_totals.assign(totals, totals + ng);
  c->enc_tok_ready = true;
  return 0;
}

int jxlhip_enc_tokens(JxlHipContext* c, const uint32_t* bases, uint32_t* tokens, size_t capacity) {
  if (!c || !bases || !tokens) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->enc_tok_ready) return JXLHIP_ERR_NO_FRAME;
  const size_t ng = c->enc_tok_totals.size();
  uint64_t need = 0;
  for (size_t g = 0; g < ng; g++) {  // every group's range inside the caller's buffer (ranges may not overlap: they are a prefix sum)
    if (uint64_t(bases[g]) + c->enc_tok_totals[g] > capacity) return JXLHIP_ERR_INVALID_ARGUMENT;
    need = std::max<uint64_t>(need, uint64_t(bases[g]) + c->enc_tok_totals[g]);
  }
  HIP_TRY(hipSetDevice(c->device));
  int r;
  if ((r = c->enc_tok_out.Ensure(size_t(need ? need : 1) * 8)) || (r = c->enc_tok_base.Ensure(ng * 4))) return r;
  HIP_TRY(hipMemcpyAsync(c->enc_tok_base.p, bases, ng * 4, hipMemcpyHostToDevice, c->stream));
  jxlhip::EncTok T = c->enc_tok;
  T.base = c->enc_tok_base.as<uint32_t>();
  T.tokens = c->enc_tok_out.as<uint2>();
  hipLaunchKernelGGL(jxlhip::k_enc_tok_emit, dim3(uint32_t(ng)), dim3(jxlhip::kTokThreads), 0, c->stream, T);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(tokens, c->enc_tok_out.p, size_t(need) * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int jxlhip_enc_forward_rerun(JxlHipContext* c, uint32_t times) {
  if (!c || !times || times > 4096) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->enc_timed) return JXLHIP_ERR_NO_FRAME;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventRecord(c->enc_ev[0], c->stream));
  for (uint32_t i = 0; i < times; i++) {
    int r = EncLaunch(c, c->enc_last, c->enc_last_gaborish);
    if (r) return r;
  }
  HIP_TRY(hipEventRecord(c->enc_ev[1], c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int jxlhip_enc_last_ms(JxlHipContext* c, float* ms) {
  if (!c || !ms) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->enc_timed) return JXLHIP_ERR_NO_FRAME;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventSynchronize(c->enc_ev[1]));
  HIP_TRY(hipEventElapsedTime(ms, c->enc_ev[0], c->enc_ev[1]));
  return 0;
}

int jxlhip_enc_last_transform_ms(JxlHipContext* c, float* ms) {
  if (!c || !ms) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->enc_timed) return JXLHIP_ERR_NO_FRAME;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventSynchronize(c->enc_ev[3]));
  HIP_TRY(hipEventElapsedTime(ms, c->enc_ev[2], c->enc_ev[3]));
  return 0;
}

int jxlhip_last_stage_ms(JxlHipContext* c, int which, float* ms) {
  if (!c || !ms || which < 0 || which > 2) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->ev_valid[which]) return JXLHIP_ERR_NO_FRAME;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventSynchronize(c->ev[which * 2 + 1]));
  HIP_TRY(hipEventElapsedTime(ms, c->ev[which * 2], c->ev[which * 2 + 1]));
  return 0;
}

}  // extern "C"
