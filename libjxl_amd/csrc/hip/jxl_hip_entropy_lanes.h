// libjxl_amd — lane-parallel AC entropy decode for gfx950 (MI355X): the throughput form of the entropy stage.
//
// Replaces (like k_entropy_uni) lib/jxl/dec_group.cc:469-542,594-639 (DecodeACVarBlock / GetBlockFromBitstream),
// dec_ans.h:170-257 (rANS symbol + hybrid uint), ans_common.h:102-142 (alias lookup), ac_context.h:63-143.
//
// One AC section (256x256 group of one pass) is an inherently serial adaptive-context rANS stream, so the only
// parallelism is ACROSS sections. Here every LANE of a wave decodes its own section: the decoder state (rANS state,
// 64-bit bit window, block/coefficient cursors) lives in VGPRs, the frame's entropy tables are shared in LDS by the
// workgroup, and one trip of the main loop decodes one symbol for every runnable lane. A wave-per-section scalar decoder
// (k_entropy_uni) is bound by the scalar ALU, which the four SIMDs of a CU share; the vector ALUs give 64 decoders per
// wave for the same issue slots. The host packs sections of similar compressed size into the same wave (longest first)
// and chooses how many lanes per wave are populated, so that small batches still spread over all SIMDs.
//
// Control flow is a per-lane state machine kept convergent on the hot part:
//   mode 2 (COEF)  the pending symbol is a coefficient token        \  one shared symbol-decode sequence per trip
//   mode 1 (NZ)    the pending symbol is a block's non-zero count    /
//   mode 0 (NEXT)  the lane finished a (block, channel) and waits for the block-transition code, which runs for all
//                  waiting lanes at once when enough of them wait (so its cost is amortised over many lanes)
//   mode 3 (DONE)  section finished, lane idle, or a stream error was flagged
#ifndef JXL_HIP_ENTROPY_LANES_H_
#define JXL_HIP_ENTROPY_LANES_H_

#include "jxl_hip_kernels.h"

namespace jxlhip {

struct EntropyLaneBatch {
  const EntropyParams* params;  // one per frame of the batch (device memory)
  const uint32_t* wg_frame;     // per workgroup: index into params
  const uint32_t* lane_group;   // per lane of every wave: group (AC section) index in its frame, 0xFFFFFFFF = idle lane
  uint32_t pass;                // which pass's sections this launch decodes
  uint32_t wait_shift;          // block transitions run once (waiting lanes << wait_shift) >= runnable lanes
};

// LDS layout of one workgroup (sizes in bytes, every region 16-byte aligned); must match LanesLdsBytes() on the host.
struct LanesLds {
  uint32_t ctx, alias, lut, ctx2, nz, total;
};
__host__ __device__ inline LanesLds LanesLdsLayout(uint32_t num_hist, uint32_t nctx, uint32_t num_clusters, uint32_t log_alpha,
                                                   uint32_t lut_bytes, uint32_t waves) {
  LanesLds l;
  l.ctx = 0;
  l.alias = ((num_hist * nctx + 16) * 4 + 15) & ~15u;
  l.lut = l.alias + (num_clusters << log_alpha) * 8;
  l.ctx2 = l.lut + ((lut_bytes + 15) & ~15u);
  l.nz = l.ctx2 + 64 * 64 * 2;
  l.total = l.nz + waves * 96 * 64;
  return l;
}

template <typename CoefT, int WPG>
__global__ __launch_bounds__(64 * WPG) void k_entropy_lanes(EntropyLaneBatch B) {
  extern __shared__ __align__(16) uint8_t lds_raw[];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const EntropyParams& P = B.params[B.wg_frame[blockIdx.x]];
  const uint32_t pass = B.pass;
  const PassDev& T = P.passes[pass];
  const uint32_t log_alpha = T.log_alpha, log_entry = 12 - log_alpha, shift = T.shift, nclusters = T.num_clusters;
  const uint32_t nq = P.nq, ndc = P.ndc, num_bctx = P.num_bctx, nctx = P.nctx, num_hist = P.num_hist;
  const uint32_t lut_bytes = 39 * nq * ndc;
  const LanesLds L = LanesLdsLayout(num_hist, nctx, nclusters, log_alpha, lut_bytes, WPG);
  uint32_t* l_ctx = reinterpret_cast<uint32_t*>(lds_raw + L.ctx);     // per context: uint config | cluster << 24
  uint2* l_alias = reinterpret_cast<uint2*>(lds_raw + L.alias);        // repacked alias entries, see below
  uint8_t* l_lut = lds_raw + L.lut;                                    // block context LUT
  uint16_t* l_ctx2 = reinterpret_cast<uint16_t*>(lds_raw + L.ctx2);    // [nzeros_left_bucket_input][k bucket input]
  uint8_t* l_nz = lds_raw + L.nz + wave * (96 * 64);                   // nzeros line buffer [channel * 32 + column][lane]

  // ---- stage the frame's tables (whole workgroup)
  {
    const uint32_t n_ctx = num_hist * nctx + 16;
    for (uint32_t i = tid; i < n_ctx; i += 64 * WPG) {
      uint32_t cl = T.ctx_map[i];
      cl = cl < nclusters ? cl : nclusters - 1;
      l_ctx[i] = (T.cfg[cl] & 0xFFFFFFu) | (cl << 24);
    }
    // alias entry {cutoff u8, right u8, freq0 u16 | offsets1 u16, freq1 u16} ->
    //   x = (freq0 - 1) & 0xFFF | cutoff << 24            (taken when pos <  cutoff: symbol = slot, offset = pos)
    //   y = (freq1 - 1) & 0xFFF | offsets1 << 12 | right << 24   (taken when pos >= cutoff)
    const uint32_t n_alias = nclusters << log_alpha;
    for (uint32_t i = tid; i < n_alias; i += 64 * WPG) {
      const uint2 e = T.alias[i];
      const uint32_t cutoff = e.x & 0xFF, right = (e.x >> 8) & 0xFF, freq0 = e.x >> 16, offs1 = e.y & 0xFFFF, freq1 = e.y >> 16;
      l_alias[i] = make_uint2(((freq0 - 1) & 0xFFFu) | (cutoff << 24), ((freq1 - 1) & 0xFFFu) | ((offs1 & 0xFFFu) << 12) | (right << 24));
    }
    for (uint32_t i = tid; i < lut_bytes; i += 64 * WPG) l_lut[i] = P.bctx_lut[i];
    for (uint32_t i = tid; i < 64 * 64; i += 64 * WPG)
      l_ctx2[i] = uint16_t((uint32_t(c_coeff_nnz_ctx[i >> 6]) + c_coeff_freq_ctx[i & 63]) * 2);
    uint32_t* z = reinterpret_cast<uint32_t*>(l_nz);
    for (uint32_t i = lane; i < 96 * 64 / 4; i += 64) z[i] = 0;
  }
  __syncthreads();

  // ---- per-lane section setup
  const uint32_t g = B.lane_group[(blockIdx.x * WPG + wave) * 64 + lane];
  uint32_t mode = g == 0xFFFFFFFFu ? 3u : 0u;
  uint32_t err = 0;
  uint32_t b1 = 0, bi = 0, ci = 2;
  const uint32_t* stream = P.sections;
  uint32_t nwords = 0, sec_size = 0, idx = 0, nxt = 0, bits = 0, state = 0, ctx_base = 0;
  uint64_t buf = 0;
#define LJ_REFILL()                                   \
  if (bits <= 32) {                                   \
    buf |= uint64_t(nxt) << bits;                     \
    bits += 32;                                       \
    idx++;                                            \
    nxt = stream[idx < nwords ? idx : nwords];        \
  }
#define LJ_READ(n_, out_)                             \
  {                                                   \
    const uint32_t nn_ = (n_);                        \
    out_ = uint32_t(buf) & ((1u << nn_) - 1);         \
    buf >>= nn_;                                      \
    bits -= nn_;                                      \
  }
  if (mode == 0) {
    bi = P.gbb[g] - 1;  // the first transition advances to the group's first block
    b1 = P.gbb[g + 1];
    const uint32_t sec = pass * P.num_groups + g;
    stream = P.sections + P.sec_word[sec];
    sec_size = P.sec_size[sec];
    nwords = (sec_size + 3) / 4;  // word `nwords` is zero padding (jxlhip_frame_upload)
    buf = uint64_t(stream[0]) | (uint64_t(stream[nwords < 1 ? nwords : 1]) << 32);
    bits = 64;
    idx = 2;
    nxt = stream[nwords < 2 ? nwords : 2];
    uint32_t tmp;
    if (sec == 0 && P.first_bit_offset) LJ_READ(P.first_bit_offset, tmp);
    uint32_t hb = 0;
    while ((1u << hb) < num_hist) hb++;
    uint32_t sel = 0;
    LJ_REFILL();
    if (hb) LJ_READ(hb, sel);
    if (sel >= num_hist) {
      err = kErrSelector;
      sel = 0;
    }
    ctx_base = sel * nctx;
    LJ_REFILL();
    uint32_t lo16, hi16;
    LJ_READ(16, lo16);
    LJ_READ(16, hi16);
    state = lo16 | (hi16 << 16);
    LJ_REFILL();
  }
  // block / channel cursor
  uint32_t info = 0, lbx = 0, lby = 0, qfi = 0, dcctx = 0, coef_offset = 0, bctx = 0, c = 0;
  // coefficient cursor
  uint32_t nzeros = 0, k = 0, size = 0, log2c = 0, hoff = 0, prev = 0, ctxe = 0;
  const uint16_t* order = T.orders;
  CoefT* dst = static_cast<CoefT*>(P.coeffs);
  const uint32_t* blk = reinterpret_cast<const uint32_t*>(P.blocks);

  for (;;) {
    const uint64_t waiting = __ballot(mode == 0), runnable = __ballot(mode == 1 || mode == 2);
    if (!(waiting | runnable)) break;
    // ------------------------------------------------------------------ block / channel transition (batched)
    if (waiting && (!runnable || (uint32_t(__popcll(waiting)) << B.wait_shift) >= uint32_t(__popcll(runnable)))) {
      if (mode == 0) {
        ci++;
        if (ci >= 3) {
          ci = 0;
          bi++;
        }
        if (bi >= b1) {  // section complete
          if (state != (0x13u << 16)) err |= kErrFinalState;
          const uint64_t consumed = uint64_t(idx) * 32 - uint64_t(bits);
          if (consumed > uint64_t(sec_size) * 8) err |= kErrOverread;
          mode = 3;
        } else {
          if (ci == 0) {
            const uint32_t w0 = blk[bi * 3], w1 = blk[bi * 3 + 1];
            coef_offset = blk[bi * 3 + 2];
            const uint32_t qf = w1 >> 16;
            dcctx = (w1 >> 8) & 0xFF;
            info = c_strategy_info[w1 & 0xFF];
            lbx = w0 & 31;
            lby = (w0 >> 16) & 31;
            qfi = 0;
            for (uint32_t t = 0; t + 1 < nq; t++) qfi += qf > P.qf_thr[t];
          }
          c = ci == 0 ? 1u : (ci == 1 ? 0u : 2u);
          const uint32_t ord = info >> 24;
          const uint8_t* line = l_nz + (c * 32) * 64 + lane;
          uint32_t pred;
          if (lbx == 0) pred = lby ? line[0] : 32;
          else if (lby == 0) pred = line[(lbx - 1) * 64];
          else pred = (uint32_t(line[lbx * 64]) + line[(lbx - 1) * 64] + 1) >> 1;
          bctx = l_lut[((c * 13 + ord) * nq + qfi) * ndc + dcctx];
          uint32_t nzb = pred >= 64 ? 64 : pred;
          nzb = nzb < 8 ? nzb : 4 + nzb / 2;
          ctxe = l_ctx[ctx_base + nzb * num_bctx + bctx];
          mode = 1;
        }
      }
    }
    // ------------------------------------------------------------------ one symbol for every runnable lane
    if (mode == 1 || mode == 2) {
      const uint32_t res = state & 0xFFFu, slot = res >> log_entry, pos = res & ((1u << log_entry) - 1);
      const uint2 e = l_alias[((ctxe >> 24) << log_alpha) + slot];
      const bool gt = pos >= (e.x >> 24);
      const uint32_t x = gt ? e.y : e.x;
      uint32_t tok = gt ? (x >> 24) : slot;
      const uint32_t hi = state >> 12;
      state = (x & 0xFFFu) * hi + hi + ((x >> 12) & 0xFFFu) + pos;
      {
        const bool need = state < (1u << 16);
        const uint32_t sh = need ? 16u : 0u;
        state = need ? ((state << 16) | (uint32_t(buf) & 0xFFFFu)) : state;
        buf >>= sh;
        bits -= sh;
      }
      LJ_REFILL();
      const uint32_t se = ctxe & 0xFF;
      if (tok >= (1u << se)) {  // hybrid uint: extra bits
        const uint32_t msb = (ctxe >> 8) & 0xFF, lsb = (ctxe >> 16) & 0xFF;
        const uint32_t nb = (se - (msb + lsb) + ((tok - (1u << se)) >> (msb + lsb))) & 31u;
        const uint32_t low = tok & ((1u << lsb) - 1), top = tok >> lsb;
        uint32_t xb;
        LJ_READ(nb, xb);
        tok = (((((1u << msb) | (top & ((1u << msb) - 1))) << nb) | xb) << lsb) | low;
        LJ_REFILL();
      }
      if (mode == 2) {
        // ---------------------------------------------------------------- coefficient token
        if (tok) {
          const uint32_t mag = tok >> 1, neg = (~tok) & 1;
          const int32_t coeff = int32_t((mag ^ (neg - 1)) << shift);
          const uint32_t p = order[k];
          if (pass == 0) dst[p] = CoefT(coeff);
          else dst[p] = CoefT(dst[p] + coeff);
          nzeros--;
          prev = 1;
        } else {
          prev = 0;
        }
        k++;
        if (nzeros == 0) {
          mode = 0;
        } else if (k >= size) {
          err |= kErrNzeros;
          mode = 3;
        } else {
          const uint32_t covered = 1u << log2c;
          const uint32_t a = ((nzeros + covered - 1) >> log2c) & 63, b = (k >> log2c) & 63;
          ctxe = l_ctx[ctx_base + hoff + l_ctx2[a * 64 + b] + prev];
        }
      } else {
        // ---------------------------------------------------------------- non-zero count of (block, channel)
        const uint32_t cx = info & 0xFF;
        log2c = (info >> 16) & 0xFF;
        const uint32_t covered = 1u << log2c;
        size = covered * 64;
        nzeros = tok;
        if (nzeros > size - covered) {
          err |= kErrNzeros;
          mode = 3;
        } else {
          const uint8_t nzv = uint8_t((nzeros + covered - 1) >> log2c);
          uint8_t* line = l_nz + (c * 32 + lbx) * 64 + lane;
          for (uint32_t i = 0; i < cx; i++) line[i * 64] = nzv;
          if (nzeros == 0) {
            mode = 0;
          } else {
            const uint32_t ord = info >> 24;
            hoff = num_bctx * 37 + 458 * bctx;
            order = T.orders + T.order_offset[ord * 3 + c];
            dst = static_cast<CoefT*>(P.coeffs) + (size_t(g) * 3 + c) * 65536 + coef_offset;
            prev = nzeros > size / 16 ? 0 : 1;
            k = covered;
            const uint32_t a = ((nzeros + covered - 1) >> log2c) & 63, b = (k >> log2c) & 63;
            ctxe = l_ctx[ctx_base + hoff + l_ctx2[a * 64 + b] + prev];
            mode = 2;
          }
        }
      }
    }
  }
#undef LJ_REFILL
#undef LJ_READ
  if (err) atomicOr(&P.errors[g], err);
}

}  // namespace jxlhip
#endif  // JXL_HIP_ENTROPY_LANES_H_
