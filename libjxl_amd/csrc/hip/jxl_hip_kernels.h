// libjxl_amd — HIP kernels of the VarDCT decode hot path for gfx950 (MI355X). Device code only; the C ABI that
// launches it is in jxl_hip_api.hip. No CPU implementation of these stages exists in the product.
//
// Kernels and the reference code they replace:
//   k_entropy_lanes (jxl_hip_entropy_lanes.h), k_entropy_uni, k_entropy_ans
//                   lib/jxl/dec_group.cc:469-542,594-639 + dec_ans.h:170-257 + ans_common.h:102-142
//   k_idct_cols<CX,CY>, k_dct<CX,CY>
//                   lib/jxl/dec_group.cc:115-181 (dequant, CfL), dec_transforms-inl.h:691-818 (LLF from DC),
//                   dct-inl.h:376-397 (scaled IDCT, evaluated here in its separable matrix form)
//   k_special       dec_transforms-inl.h:66-93,95-454,463-568 (IDENTITY, DCT2X2, DCT4X4, DCT4X8, DCT8X4, AFV0-3)
//   k_dct_big       same as k_dct for 128/256-class transforms (global scratch instead of LDS)
//   k_filter_fused (jxl_hip_filter_fused.h)
//                   render_pipeline/stage_gaborish.cc:56-100, stage_epf.cc:82-494, stage_xyb.cc:80-92,
//                   dec_xyb-inl.h:38-86, stage_from_linear.cc:114-144, cms/transfer_functions-inl.h:245-268,
//                   stage_write.cc:266-286,548-590
#ifndef JXL_HIP_KERNELS_H_
#define JXL_HIP_KERNELS_H_

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../../include/jxl_amd_hip.h"

namespace jxlhip {

// ---------------------------------------------------------------------------------------------- constants
__constant__ uint8_t c_covered_x[27] = {1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32};
__constant__ uint8_t c_covered_y[27] = {1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16};
__constant__ uint8_t c_log2_covered[27] = {0, 0, 0, 0, 2, 4, 1, 1, 2, 2, 3, 3, 0, 0, 0, 0, 0, 0, 6, 5, 5, 8, 7, 7, 10, 9, 9};
__constant__ uint8_t c_strategy_order[27] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
__constant__ uint8_t c_strategy_qtable[27] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};
__constant__ uint16_t c_coeff_freq_ctx[64] = {
    0,  0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 15, 16, 16, 17, 17,
    18, 18, 19, 19, 20, 20, 21, 21, 22, 22, 23, 23, 23, 23, 24, 24, 24, 24, 25, 25, 25, 25,
    26, 26, 26, 26, 27, 27, 27, 27, 28, 28, 28, 28, 29, 29, 29, 29, 30, 30, 30, 30};
__constant__ uint16_t c_coeff_nnz_ctx[64] = {
    0,   0,   31,  62,  62,  93,  93,  93,  93,  123, 123, 123, 123, 152, 152, 152, 152, 152, 152, 152, 152, 180,
    180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206,
    206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206};
// AFV basis and dither tables are uploaded at context creation (see jxl_hip_api.hip).
__constant__ float c_afv_basis[16 * 16];
__constant__ float c_dither[32 * 32];
// Resample scales mapping coefficient i of an n-point DCT to an 8n-point DCT; entry [n - 1 + i], n in {1,2,4,8,16,32}.
__constant__ float c_resample[63];

// Copies a per-frame parameter block into a local object with loads through the constant address space: they become
// scalar loads that the compiler may hoist and re-materialise freely (a plain global reference has to be re-read after
// every store the compiler cannot prove distinct).
template <typename T>
__device__ __forceinline__ void LoadParams(T& dst, const T* src) {
  static_assert(sizeof(T) % 4 == 0, "parameter blocks are dword multiples");
  const uint32_t __attribute__((address_space(4)))* s = (const uint32_t __attribute__((address_space(4)))*)(uintptr_t)src;
  uint32_t* d = reinterpret_cast<uint32_t*>(&dst);
#pragma unroll
  for (uint32_t i = 0; i < sizeof(T) / 4; i++) d[i] = s[i];
}

static const uint32_t kErrNzeros = 1, kErrFinalState = 2, kErrOverread = 4, kErrSelector = 8;

struct PassDev {
  const uint8_t* ctx_map;
  const uint2* alias;
  const uint32_t* cfg;
  const uint16_t* orders;
  uint32_t order_offset[39];
  uint32_t log_alpha, num_clusters, shift, alias_lds;
  // prefix-coded and / or LZ77 streams (k_entropy_generic): see JxlHipPassDesc
  uint32_t use_prefix, lz77, lz_min_symbol, lz_min_length, lz_len_cfg, lz_dist_ctx;
  const uint32_t* prefix_table;
  const uint32_t* prefix_offset;
  // the alias entries as the lane kernel reads them (x = (freq0 - 1) & 0xFFF | cutoff << 24, y = (freq1 - 1) & 0xFFF |
  // offsets1 << 12 | right << 24): staged into LDS, or read in place when the tables are too large to keep the frames
  // of a launch resident
  const uint2* alias_packed;
};

struct EntropyParams {
  const uint32_t* sections;   // all AC sections, each starting at a 16-byte aligned offset
  const uint32_t* sec_word;   // [pass * num_groups + group] start, in 32-bit words
  const uint32_t* sec_size;   // bytes
  uint32_t first_bit_offset;
  const JxlHipVarBlock* blocks;
  const uint32_t* gbb;
  const uint8_t* bctx_lut;
  uint32_t nq, ndc, num_bctx;
  uint32_t qf_thr[16];
  uint32_t num_hist, nctx;  // nctx = num_bctx * 495
  uint32_t cs;              // chroma-subsampled frames: hshift of channel c in bit 2 * c, vshift in bit 2 * c + 1 (0 = 4:4:4)
  const PassDev* passes;
  uint32_t num_passes, num_groups;
  void* coeffs;
  uint32_t* errors;
  uint32_t lds_ctx_bytes, lds_alias_bytes;
  uint32_t* kend;  // [pass][block * 3 + channel]: number of valid scan-order entries (k_entropy_lanes only)
  // k_entropy_lanes, multi-pass frames: pass p writes its scan-order coefficients at element offset coef_pass_base +
  // p * coef_pass_stride of `coeffs` and its kend at p * kend_pass_stride (single pass: all three are 0 / unused);
  // k_merge_passes then sums the passes into the natural-layout buffer at offset 0
  uint64_t coef_pass_base, coef_pass_stride;
  uint32_t kend_pass_stride;
  // per block, for k_entropy_lanes: lbx | lby << 5 | strategy << 10 | qf bucket << 15 | dc bucket << 19 (16-byte aligned, padded)
  const uint32_t* block_recs;
  // [pass * num_groups + group]: bit position (from the section's first byte) where the coefficient stream ended. The
  // host needs it for sections that carry Modular data of extra channels behind the coefficients (dec_frame.cc:511-542).
  uint32_t* sec_end_bits;
  // LZ77 streams (k_entropy_generic): window of the values decoded so far, kLzWindow entries per section of one pass
  uint32_t* lz_window;
};
constexpr uint32_t kLzWindow = 1u << 18;  // a section decodes fewer than 3 * (65536 + 1024) values: never wraps

struct BitReader {
  const uint32_t* p;
  uint32_t idx, nwords;
  uint64_t buf;
  int bits;
};
__device__ __forceinline__ void BrRefill(BitReader& b) {
  if (b.bits < 32) {
    uint32_t w = b.idx < b.nwords ? b.p[b.idx] : 0u;
    b.idx++;
    b.buf |= uint64_t(w) << b.bits;
    b.bits += 32;
  }
}
__device__ __forceinline__ uint32_t BrRead(BitReader& b, uint32_t n) {  // n <= 32, caller refilled
  uint32_t v = uint32_t(b.buf & ((uint64_t(1) << n) - 1));
  b.buf >>= n;
  b.bits -= int(n);
  return v;
}

// One rANS symbol + hybrid-uint extra bits from `cluster`.
template <bool ALIAS_LDS>
__device__ __forceinline__ uint32_t ReadHybrid(BitReader& br, uint32_t& state, uint32_t cluster, const uint2* alias_g,
                                               const uint2* alias_l, const uint32_t* cfg, uint32_t log_alpha) {
  const uint32_t log_entry = 12 - log_alpha;
  const uint32_t res = state & 0xFFFu;
  const uint32_t i = res >> log_entry;
  const uint32_t pos = res & ((1u << log_entry) - 1);
  const uint2 e = ALIAS_LDS ? alias_l[(cluster << log_alpha) + i] : alias_g[(cluster << log_alpha) + i];
  const uint32_t cutoff = e.x & 0xFF, right = (e.x >> 8) & 0xFF, freq0 = e.x >> 16;
  const uint32_t offsets1 = e.y & 0xFFFF, freq1 = e.y >> 16;
  const bool greater = pos >= cutoff;
  const uint32_t token = greater ? right : i;
  const uint32_t off = (greater ? offsets1 : 0u) + pos;
  const uint32_t freq = greater ? freq1 : freq0;
  state = freq * (state >> 12) + off;
  BrRefill(br);
  if (state < (1u << 16)) state = (state << 16) | BrRead(br, 16);
  const uint32_t c = cfg[cluster];
  const uint32_t split_exp = c & 0xFF, msb = (c >> 8) & 0xFF, lsb = (c >> 16) & 0xFF;
  const uint32_t split_token = 1u << split_exp;
  if (token < split_token) return token;
  uint32_t nbits = split_exp - (msb + lsb) + ((token - split_token) >> (msb + lsb));
  nbits &= 31u;
  const uint32_t low = token & ((1u << lsb) - 1);
  const uint32_t hi = token >> lsb;
  BrRefill(br);
  const uint32_t bits = BrRead(br, nbits);
  return (((((1u << msb) | (hi & ((1u << msb) - 1))) << nbits) | bits) << lsb) | low;
}

// One 64-lane workgroup per 256x256 group. All lanes zero the group's coefficient planes and stage the entropy
// tables in LDS; lane 0 then walks the (inherently serial) adaptive-context rANS stream.
template <typename CoefT, bool ALIAS_LDS>
__global__ __launch_bounds__(64) void k_entropy_ans(EntropyParams P) {
  extern __shared__ __align__(16) uint8_t lds_raw[];
  const uint32_t g = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  uint8_t* l_ctx = lds_raw;
  uint2* l_alias = reinterpret_cast<uint2*>(lds_raw + P.lds_ctx_bytes);
  uint8_t* l_nz = lds_raw + P.lds_ctx_bytes + P.lds_alias_bytes;  // 3 * 1024
  const uint32_t b0 = P.gbb[g], b1 = P.gbb[g + 1];
  CoefT* gco = static_cast<CoefT*>(P.coeffs) + size_t(g) * 3 * 65536;
  // total coefficients per channel in this group
  uint32_t total = 0;
  if (b1 > b0) {
    const JxlHipVarBlock last = P.blocks[b1 - 1];
    total = last.coef_offset + (64u << c_log2_covered[last.strategy]);
  }
  for (uint32_t pass = 0; pass < P.num_passes; pass++) {
    const PassDev& T = P.passes[pass];
    const uint32_t sec = pass * P.num_groups + g;
    if (pass && P.sec_size[sec] == 0) break;  // (a frame drawn from a prefix of its bytes: this pass of the group has not arrived)
    __syncthreads();
    if (pass == 0) {
      // zero fill (16-byte stores)
      const uint32_t n16 = (total * uint32_t(sizeof(CoefT))) / 16;
      for (int c = 0; c < 3; c++) {
        uint4* dst = reinterpret_cast<uint4*>(gco + size_t(c) * 65536);
        for (uint32_t i = lane; i < n16; i += 64) dst[i] = make_uint4(0, 0, 0, 0);
      }
      if (P.sec_size[sec] == 0) break;  // (none of the group's passes has arrived: the DC image alone)
    }
    for (uint32_t i = lane; i < 3 * 1024; i += 64) l_nz[i] = 0;
    // histogram selector is read by lane 0 first; all lanes then stage that slice of the context map
    __shared__ uint32_t s_sel, s_state;
    __shared__ BitReader s_br;
    if (lane == 0) {
      BitReader br;
      br.p = P.sections + P.sec_word[sec];
      br.nwords = (P.sec_size[sec] + 3) / 4;
      br.idx = 0;
      br.buf = 0;
      br.bits = 0;
      BrRefill(br);
      if (sec == 0 && P.first_bit_offset) BrRead(br, P.first_bit_offset);
      uint32_t hb = 0;
      while ((1u << hb) < P.num_hist) hb++;
      BrRefill(br);
      uint32_t sel = hb ? BrRead(br, hb) : 0;
      if (sel >= P.num_hist) {
        atomicOr(&P.errors[g], kErrSelector);
        sel = 0;
      }
      BrRefill(br);
      uint32_t st = BrRead(br, 16);
      BrRefill(br);
      st |= BrRead(br, 16) << 16;
      s_sel = sel;
      s_state = st;
      s_br = br;
    }
    __syncthreads();
    {
      const uint8_t* src = T.ctx_map + size_t(s_sel) * P.nctx;
      for (uint32_t i = lane; i < P.nctx + 16; i += 64) l_ctx[i] = src[i];
      if (ALIAS_LDS) {
        const uint32_t n = T.num_clusters << T.log_alpha;
        for (uint32_t i = lane; i < n; i += 64) l_alias[i] = T.alias[i];
      }
    }
    __threadfence_block();
    __syncthreads();
    if (lane != 0) continue;

    BitReader br = s_br;
    uint32_t state = s_state;
    const uint32_t log_alpha = T.log_alpha;
    const uint32_t shift = T.shift;
    uint32_t err = 0;
    for (uint32_t bi = b0; bi < b1 && !err; bi++) {
      const JxlHipVarBlock vb = P.blocks[bi];
      const uint32_t st = vb.strategy;
      const uint32_t cx = c_covered_x[st], cy = c_covered_y[st], log2c = c_log2_covered[st];
      const uint32_t covered = 1u << log2c, size = covered * 64;
      const uint32_t ord = c_strategy_order[st];
      uint32_t qfi = 0;
      for (uint32_t t = 0; t + 1 < P.nq; t++) qfi += vb.qf > P.qf_thr[t];
      const uint32_t fbx = vb.bx & 31, fby = vb.by & 31;
#pragma unroll 1
      for (int ci = 0; ci < 3 && !err; ci++) {
        const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
        // (a subsampled channel, EntropyParams::cs: only the blocks on its own grid carry it, and its counts live there)
        const uint32_t hs = (P.cs >> (2 * c)) & 1u, vs = (P.cs >> (2 * c + 1)) & 1u;
        if ((fbx & hs) | (fby & vs)) continue;
        const uint32_t lbx = fbx >> hs, lby = fby >> vs;
        uint8_t* nzc = l_nz + c * 1024;
        uint32_t pred;
        if (lbx == 0) pred = lby ? nzc[(lby - 1) * 32] : 32;
        else if (lby == 0) pred = nzc[lbx - 1];
        else pred = (uint32_t(nzc[(lby - 1) * 32 + lbx]) + nzc[lby * 32 + lbx - 1] + 1) >> 1;
        const uint32_t bctx = P.bctx_lut[((c * 13 + ord) * P.nq + qfi) * P.ndc + vb.quant_dc_ctx];
        uint32_t nzb = pred >= 64 ? 64 : pred;
        nzb = nzb < 8 ? nzb : 4 + nzb / 2;
        uint32_t nzeros = ReadHybrid<ALIAS_LDS>(br, state, l_ctx[nzb * P.num_bctx + bctx], T.alias, l_alias, T.cfg, log_alpha);
        if (nzeros > size - covered) {
          err = kErrNzeros;
          break;
        }
        const uint8_t nzv = uint8_t((nzeros + covered - 1) >> log2c);
        for (uint32_t y = 0; y < cy; y++)
          for (uint32_t x = 0; x < cx; x++) nzc[(lby + y) * 32 + lbx + x] = nzv;
        const uint32_t hoff = P.num_bctx * 37 + 458 * bctx;
        const uint16_t* order = T.orders + T.order_offset[ord * 3 + c];
        CoefT* dst = gco + size_t(c) * 65536 + vb.coef_offset;
        uint32_t prev = nzeros > size / 16 ? 0 : 1;
        for (uint32_t k = covered; k < size && nzeros != 0; ++k) {
          const uint32_t nzl = (nzeros + covered - 1) >> log2c;
          const uint32_t ctx = hoff + (uint32_t(c_coeff_nnz_ctx[nzl & 63]) + c_coeff_freq_ctx[(k >> log2c) & 63]) * 2 + prev;
          const uint32_t u = ReadHybrid<ALIAS_LDS>(br, state, l_ctx[ctx], T.alias, l_alias, T.cfg, log_alpha);
          const uint32_t mag = u >> 1, neg = (~u) & 1;
          const int32_t coeff = int32_t((mag ^ (neg - 1)) << shift);
          if (u) {
            const uint32_t pos = order[k];
            if (pass == 0) dst[pos] = CoefT(coeff);
            else dst[pos] = CoefT(dst[pos] + coeff);
          }
          prev = u != 0;
          nzeros -= prev;
        }
        if (nzeros != 0) err = kErrNzeros;
      }
    }
    if (!err && state != (0x13u << 16)) err |= kErrFinalState;
    {
      const uint64_t consumed = uint64_t(br.idx) * 32 - uint64_t(br.bits);
      if (consumed > uint64_t(P.sec_size[sec]) * 8) err |= kErrOverread;
      P.sec_end_bits[sec] = uint32_t(consumed);
    }
    if (err) atomicOr(&P.errors[g], err);
  }
}

// ---------------------------------------------------------------------------------------------- entropy, every code
// The streams libjxl's fastest and slowest efforts write: prefix codes instead of rANS (dec_huffman.h:28-41, dec_ans.h:
// 170-197) and / or LZ77 copies of earlier values (dec_ans.h:288-353). Same walk as k_entropy_ans, tables in global
// memory, one lane per section: the correctness path for these streams, not a tuned one.
// Prefix-code lookup in the two-level tables the host builds (jxh_entropy.h AppendPrefixTables): `tab` = the cluster's
// root, `root_bits` its index width, `bits` = the next >= 15 bits of the stream. Returns symbol << 8 | code length.
__device__ __forceinline__ uint32_t PrefixLookup(const uint32_t* tab, uint32_t root_bits, uint32_t bits) {
  uint32_t e = tab[bits & ((1u << root_bits) - 1)];
  if (e & 0x80u) e = tab[(e >> 8) + ((bits >> root_bits) & ((1u << (e & 0x7Fu)) - 1))];
  return e;
}

struct GenericReader {
  BitReader br;
  uint32_t state;
  const PassDev* T;
  uint32_t* window;
  uint32_t num_decoded, num_to_copy, copy_pos;
  uint32_t err;
};
__device__ __forceinline__ uint32_t GenericSymbol(GenericReader& r, uint32_t cluster) {
  const PassDev& T = *r.T;
  BrRefill(r.br);
  if (T.use_prefix) {
    const uint32_t po = T.prefix_offset[cluster];
    const uint32_t e = PrefixLookup(T.prefix_table + (po & 0xFFFFFFu), po >> 24, uint32_t(r.br.buf));
    BrRead(r.br, e & 0xFF);
    return e >> 8;
  }
  const uint32_t log_entry = 12 - T.log_alpha;
  const uint32_t res = r.state & 0xFFFu, i = res >> log_entry, pos = res & ((1u << log_entry) - 1);
  const uint2 e = T.alias[(cluster << T.log_alpha) + i];
  const uint32_t cutoff = e.x & 0xFF, right = (e.x >> 8) & 0xFF, freq0 = e.x >> 16, offsets1 = e.y & 0xFFFF, freq1 = e.y >> 16;
  const bool greater = pos >= cutoff;
  r.state = (greater ? freq1 : freq0) * (r.state >> 12) + (greater ? offsets1 : 0u) + pos;
  if (r.state < (1u << 16)) r.state = (r.state << 16) | BrRead(r.br, 16);
  return greater ? right : i;
}
__device__ __forceinline__ uint32_t GenericUint(GenericReader& r, uint32_t cfg, uint32_t token) {
  const uint32_t split_exp = cfg & 0xFF, msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
  if (token < (1u << split_exp)) return token;
  const uint32_t nbits = (split_exp - (msb + lsb) + ((token - (1u << split_exp)) >> (msb + lsb))) & 31u;
  const uint32_t low = token & ((1u << lsb) - 1), hi = token >> lsb;
  BrRefill(r.br);
  const uint32_t bits = BrRead(r.br, nbits);
  return (((((1u << msb) | (hi & ((1u << msb) - 1))) << nbits) | bits) << lsb) | low;
}
__device__ __forceinline__ uint32_t GenericRead(GenericReader& r, uint32_t cluster) {
  const PassDev& T = *r.T;
  if (T.lz77 && r.num_to_copy > 0) {
    // (copy_pos == num_decoded only for a copy at the very start of a stream, distance 0: zeros, dec_ans.h:320-327)
    const uint32_t v = r.copy_pos >= r.num_decoded ? 0u : r.window[r.copy_pos & (kLzWindow - 1)];
    r.copy_pos++;
    r.num_to_copy--;
    r.window[(r.num_decoded++) & (kLzWindow - 1)] = v;
    return v;
  }
  const uint32_t token = GenericSymbol(r, cluster);
  if (T.lz77 && token >= T.lz_min_symbol) {
    r.num_to_copy = GenericUint(r, T.lz_len_cfg, token - T.lz_min_symbol) + T.lz_min_length;
    const uint32_t dtok = GenericSymbol(r, T.lz_dist_ctx);
    uint32_t distance = GenericUint(r, T.cfg[T.lz_dist_ctx], dtok) + 1;  // (no special distances without a multiplier)
    if (distance > r.num_decoded) distance = r.num_decoded;
    if (distance > kLzWindow) distance = kLzWindow;
    r.copy_pos = r.num_decoded - distance;
    if (r.num_to_copy < T.lz_min_length || r.num_decoded + r.num_to_copy > kLzWindow) {  // length overflow / more than a section holds
      r.err |= kErrNzeros;
      r.num_to_copy = 0;
      return 0;
    }
    const uint32_t v = distance == 0 ? 0u : r.window[r.copy_pos & (kLzWindow - 1)];
    r.copy_pos++;
    r.num_to_copy--;
    r.window[(r.num_decoded++) & (kLzWindow - 1)] = v;
    return v;
  }
  const uint32_t v = GenericUint(r, T.cfg[cluster], token);
  if (T.lz77) r.window[(r.num_decoded++) & (kLzWindow - 1)] = v;
  return v;
}

template <typename CoefT>
__global__ __launch_bounds__(64) void k_entropy_generic(EntropyParams P) {
  extern __shared__ __align__(16) uint8_t lds_raw[];
  const uint32_t g = blockIdx.x, lane = threadIdx.x;
  uint8_t* l_nz = lds_raw;  // 3 * 1024
  const uint32_t b0 = P.gbb[g], b1 = P.gbb[g + 1];
  CoefT* gco = static_cast<CoefT*>(P.coeffs) + size_t(g) * 3 * 65536;
  uint32_t total = 0;
  if (b1 > b0) {
    const JxlHipVarBlock last = P.blocks[b1 - 1];
    total = last.coef_offset + (64u << c_log2_covered[last.strategy]);
  }
  {
    const uint32_t n16 = (total * uint32_t(sizeof(CoefT))) / 16;
    for (int c = 0; c < 3; c++) {
      uint4* dst = reinterpret_cast<uint4*>(gco + size_t(c) * 65536);
      for (uint32_t i = lane; i < n16; i += 64) dst[i] = make_uint4(0, 0, 0, 0);
    }
  }
  for (uint32_t pass = 0; pass < P.num_passes; pass++) {
    const PassDev& T = P.passes[pass];
    const uint32_t sec = pass * P.num_groups + g;
    if (P.sec_size[sec] == 0) break;  // (this pass of the group has not arrived: a frame drawn from a prefix; the group is zeroed above)
    __syncthreads();
    for (uint32_t i = lane; i < 3 * 1024; i += 64) l_nz[i] = 0;
    __threadfence_block();
    __syncthreads();
    if (lane != 0) continue;
    GenericReader r;
    r.T = &T;
    r.window = P.lz_window ? P.lz_window + size_t(g) * kLzWindow : nullptr;  // (passes run one after the other)
    r.num_decoded = r.num_to_copy = r.copy_pos = 0;
    r.err = 0;
    r.br.p = P.sections + P.sec_word[sec];
    r.br.nwords = (P.sec_size[sec] + 3) / 4;
    r.br.idx = 0;
    r.br.buf = 0;
    r.br.bits = 0;
    BrRefill(r.br);
    if (sec == 0 && P.first_bit_offset) BrRead(r.br, P.first_bit_offset);
    uint32_t hb = 0;
    while ((1u << hb) < P.num_hist) hb++;
    BrRefill(r.br);
    uint32_t sel = hb ? BrRead(r.br, hb) : 0;
    if (sel >= P.num_hist) {
      r.err |= kErrSelector;
      sel = 0;
    }
    r.state = 0x13u << 16;
    if (!T.use_prefix) {  // dec_ans.cc:402: only an ANS stream starts with its state
      BrRefill(r.br);
      r.state = BrRead(r.br, 16);
      BrRefill(r.br);
      r.state |= BrRead(r.br, 16) << 16;
    }
    const uint8_t* ctx_map = T.ctx_map + size_t(sel) * P.nctx;
    const uint32_t shift = T.shift;
    for (uint32_t bi = b0; bi < b1 && !r.err; bi++) {
      const JxlHipVarBlock vb = P.blocks[bi];
      const uint32_t st = vb.strategy;
      const uint32_t cx = c_covered_x[st], cy = c_covered_y[st], log2c = c_log2_covered[st];
      const uint32_t covered = 1u << log2c, size = covered * 64;
      const uint32_t ord = c_strategy_order[st];
      uint32_t qfi = 0;
      for (uint32_t t = 0; t + 1 < P.nq; t++) qfi += vb.qf > P.qf_thr[t];
      const uint32_t fbx = vb.bx & 31, fby = vb.by & 31;
#pragma unroll 1
      for (int ci = 0; ci < 3 && !r.err; ci++) {
        const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
        const uint32_t hs = (P.cs >> (2 * c)) & 1u, vs = (P.cs >> (2 * c + 1)) & 1u;  // (subsampled channels: see k_entropy_ans)
        if ((fbx & hs) | (fby & vs)) continue;
        const uint32_t lbx = fbx >> hs, lby = fby >> vs;
        uint8_t* nzc = l_nz + c * 1024;
        uint32_t pred;
        if (lbx == 0) pred = lby ? nzc[(lby - 1) * 32] : 32;
        else if (lby == 0) pred = nzc[lbx - 1];
        else pred = (uint32_t(nzc[(lby - 1) * 32 + lbx]) + nzc[lby * 32 + lbx - 1] + 1) >> 1;
        const uint32_t bctx = P.bctx_lut[((c * 13 + ord) * P.nq + qfi) * P.ndc + vb.quant_dc_ctx];
        uint32_t nzb = pred >= 64 ? 64 : pred;
        nzb = nzb < 8 ? nzb : 4 + nzb / 2;
        uint32_t nzeros = GenericRead(r, ctx_map[nzb * P.num_bctx + bctx]);
        if (nzeros > size - covered) {
          r.err |= kErrNzeros;
          break;
        }
        const uint8_t nzv = uint8_t((nzeros + covered - 1) >> log2c);
        for (uint32_t y = 0; y < cy; y++)
          for (uint32_t x = 0; x < cx; x++) nzc[(lby + y) * 32 + lbx + x] = nzv;
        const uint32_t hoff = P.num_bctx * 37 + 458 * bctx;
        const uint16_t* order = T.orders + T.order_offset[ord * 3 + c];
        CoefT* dst = gco + size_t(c) * 65536 + vb.coef_offset;
        uint32_t prev = nzeros > size / 16 ? 0 : 1;
        for (uint32_t k = covered; k < size && nzeros != 0 && !r.err; ++k) {
          const uint32_t nzl = (nzeros + covered - 1) >> log2c;
          const uint32_t ctx = hoff + (uint32_t(c_coeff_nnz_ctx[nzl & 63]) + c_coeff_freq_ctx[(k >> log2c) & 63]) * 2 + prev;
          const uint32_t u = GenericRead(r, ctx_map[ctx]);
          const uint32_t mag = u >> 1, neg = (~u) & 1;
          const int32_t coeff = int32_t((mag ^ (neg - 1)) << shift);
          if (u) {
            const uint32_t pos = order[k];
            dst[pos] = CoefT(dst[pos] + coeff);
          }
          prev = u != 0;
          nzeros -= prev;
        }
        if (nzeros != 0) r.err |= kErrNzeros;
      }
    }
    uint32_t err = r.err;
    if (!err && r.state != (0x13u << 16)) err |= kErrFinalState;
    {
      const uint64_t consumed = uint64_t(r.br.idx) * 32 - uint64_t(r.br.bits);
      if (consumed > uint64_t(P.sec_size[sec]) * 8) err |= kErrOverread;
      P.sec_end_bits[sec] = uint32_t(consumed);
    }
    if (err) atomicOr(&P.errors[g], err);
  }
}

// ---------------------------------------------------------------------------------------------- entropy, scalar form
// Same decode as k_entropy_ans, restructured so that the serial chain runs on the scalar unit: every lane of the
// wave executes the identical (wave-uniform) control flow, all decoder state lives in SGPRs, table reads that have a
// uniform address go through the scalar cache (constant address space) or LDS + readfirstlane, and only the
// coefficient stores are vector instructions (lane 0). The context-map / uint-config lookups of the NEXT coefficient
// are issued speculatively for both outcomes (zero / non-zero) before the current symbol is resolved, which takes
// them off the dependency chain: per symbol the chain is one LDS alias-table read + ~30 scalar ALU ops.
typedef const uint32_t __attribute__((address_space(4)))* CU32;
__device__ __forceinline__ CU32 AsConst(const void* p) { return (CU32)(uintptr_t)p; }
__device__ __forceinline__ uint32_t Uni(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }

__constant__ uint32_t c_strategy_info[27] = {  // covered_x | covered_y << 8 | log2_covered << 16 | order bucket << 24
    0x00000101, 0x01000101, 0x01000101, 0x01000101, 0x02020202, 0x03040404, 0x04010201, 0x04010102, 0x05020401,
    0x05020104, 0x06030402, 0x06030204, 0x01000101, 0x01000101, 0x01000101, 0x01000101, 0x01000101, 0x01000101,
    0x07060808, 0x08050804, 0x08050408, 0x09081010, 0x0A071008, 0x0A070810, 0x0B0A2020, 0x0C092010, 0x0C091020};

struct SBits {
  CU32 p;
  uint32_t idx, nwords;  // idx = number of words already moved into buf
  uint64_t buf;
  uint32_t bits;
  uint32_t next;         // word idx, preloaded
};
__device__ __forceinline__ void SRefill(SBits& b) {
  if (b.bits < 32) {
    b.buf |= uint64_t(b.next) << b.bits;
    b.bits += 32;
    b.idx++;
    const uint32_t j = b.idx < b.nwords ? b.idx : b.nwords;  // word `nwords` is zero padding
    b.next = b.p[j];
  }
}
__device__ __forceinline__ uint32_t SRead(SBits& b, uint32_t n) {
  const uint32_t v = uint32_t(b.buf) & ((n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1));
  b.buf >>= n;
  b.bits -= n;
  return v;
}
__device__ __forceinline__ uint32_t FreqCtx(uint32_t k) {  // kCoeffFreqContext, arithmetic form (k in 1..63)
  return k < 16 ? k - 1 : (k < 32 ? 15 + ((k - 16) >> 1) : 23 + ((k - 32) >> 2));
}

// Decodes one symbol from `cluster` (uniform); `cfg` = packed uint config of that cluster.
__device__ __forceinline__ uint32_t SReadHybrid(SBits& br, uint32_t& state, uint32_t cluster, uint32_t cfg, const uint2* l_alias,
                                                uint32_t log_alpha) {
  const uint32_t log_entry = 12 - log_alpha;
  const uint32_t res = state & 0xFFFu;
  const uint32_t i = res >> log_entry;
  const uint32_t pos = res & ((1u << log_entry) - 1);
  const uint2 ev = l_alias[(cluster << log_alpha) + i];
  const uint32_t e0 = Uni(ev.x), e1 = Uni(ev.y);
  const uint32_t cutoff = e0 & 0xFF;
  const bool greater = pos >= cutoff;
  const uint32_t token = greater ? ((e0 >> 8) & 0xFF) : i;
  const uint32_t off = (greater ? (e1 & 0xFFFF) : 0u) + pos;
  const uint32_t freq = greater ? (e1 >> 16) : (e0 >> 16);
  state = freq * (state >> 12) + off;
  SRefill(br);
  if (state < (1u << 16)) state = (state << 16) | SRead(br, 16);
  const uint32_t split_exp = cfg & 0xFF, msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
  const uint32_t split_token = 1u << split_exp;
  if (token < split_token) return token;
  uint32_t nbits = split_exp - (msb + lsb) + ((token - split_token) >> (msb + lsb));
  nbits &= 31u;
  const uint32_t low = token & ((1u << lsb) - 1);
  const uint32_t hi = token >> lsb;
  SRefill(br);
  const uint32_t bits = SRead(br, nbits);
  return (((((1u << msb) | (hi & ((1u << msb) - 1))) << nbits) | bits) << lsb) | low;
}

__device__ __forceinline__ uint32_t NnzCtx(uint32_t n) {  // kCoeffNumNonzeroContext, arithmetic form (n in 1..63)
  const uint32_t b = (n > 1) + (n > 2) + (n > 4) + (n > 8) + (n > 12) + (n > 20) + (n > 32);
  return uint32_t((0xCEB4987B5D3E1F00ull >> (8 * b)) & 0xFF);
}

// WPG waves per workgroup, one 256x256 group (AC section) per wave; the waves of a workgroup share one LDS copy of
// the context map, alias tables and uint configs. Each wave owns an nzeros map and a small (k, value) list in LDS:
// the serial scalar loop only appends non-zero coefficients to the list; all 64 lanes then scatter the list through
// the coefficient-order table into HBM (so neither the order lookup nor the stores sit on the serial chain).
struct EntropyBatch {
  const EntropyParams* params;  // one per frame (device memory)
  const uint32_t* wg_map;       // per workgroup: frame << 16 | workgroup index inside the frame; NULL = single frame
};

template <typename CoefT, int WPG>
__global__ __launch_bounds__(64 * WPG) void k_entropy_uni(EntropyBatch B) {
  extern __shared__ __align__(16) uint8_t lds_raw[];
  constexpr uint32_t kList = 256;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = Uni(threadIdx.x >> 6);
  const uint32_t wg = B.wg_map ? AsConst(B.wg_map)[blockIdx.x] : blockIdx.x;
  // the frame's parameter block is read through the constant address space (scalar loads, never clobbered)
  const EntropyParams __attribute__((address_space(4)))* Pp =
      (const EntropyParams __attribute__((address_space(4)))*)(uintptr_t)(B.params + (wg >> 16));
#define P (*Pp)
  const uint32_t g = (wg & 0xFFFF) * WPG + wave;
  const bool have_group = g < P.num_groups;
  uint8_t* l_ctx = lds_raw;
  uint2* l_alias = reinterpret_cast<uint2*>(lds_raw + P.lds_ctx_bytes);
  uint32_t* l_cfg = reinterpret_cast<uint32_t*>(lds_raw + P.lds_ctx_bytes + P.lds_alias_bytes);  // 256
  uint8_t* l_priv = reinterpret_cast<uint8_t*>(l_cfg + 256) + wave * (3072 + kList * 8);
  uint8_t* l_nz = l_priv;                                         // 3 * 1024
  uint2* l_list = reinterpret_cast<uint2*>(l_priv + 3072);        // kList x (k, value)
  const CU32 gbb = AsConst(P.gbb);
  const CU32 blk = AsConst(P.blocks);
  uint32_t b0 = 0, b1 = 0;
  if (have_group) {
    b0 = gbb[g];
    b1 = gbb[g + 1];
  }
  CoefT* gco = static_cast<CoefT*>(P.coeffs) + size_t(have_group ? g : 0) * 3 * 65536;
  uint32_t total = 0;
  if (b1 > b0) {
    const uint32_t w1 = blk[(b1 - 1) * 3 + 1], w2 = blk[(b1 - 1) * 3 + 2];
    total = w2 + (64u << ((c_strategy_info[w1 & 0xFF] >> 16) & 0xFF));
  }
  const CU32 lut = AsConst(P.bctx_lut);
  for (uint32_t pass = 0; pass < P.num_passes; pass++) {
    const PassDev& T = P.passes[pass];
    const uint32_t log_alpha = Uni(T.log_alpha), shift = Uni(T.shift), nclusters = Uni(T.num_clusters);
    __syncthreads();
    {
      // shared tables (selector 0 slice; launches with several histogram sets use WPG = 1 and re-stage below)
      const uint8_t* src = T.ctx_map;
      for (uint32_t i = threadIdx.x; i < P.nctx + 16; i += 64 * WPG) l_ctx[i] = src[i];
      const uint32_t n = nclusters << log_alpha;
      for (uint32_t i = threadIdx.x; i < n; i += 64 * WPG) l_alias[i] = T.alias[i];
      for (uint32_t i = threadIdx.x; i < nclusters; i += 64 * WPG) l_cfg[i] = T.cfg[i];
    }
    if (have_group && pass == 0) {
      const uint32_t n16 = (total * uint32_t(sizeof(CoefT))) / 16;
      for (int c = 0; c < 3; c++) {
        uint4* dst = reinterpret_cast<uint4*>(gco + size_t(c) * 65536);
        for (uint32_t i = lane; i < n16; i += 64) dst[i] = make_uint4(0, 0, 0, 0);
      }
    }
    for (uint32_t i = lane; i < 3 * 1024; i += 64) l_nz[i] = 0;
    __threadfence_block();
    __syncthreads();
    if (!have_group) continue;
    const uint32_t sec = pass * P.num_groups + g;
    // bit reader: the stream word for the NEXT refill is fetched with a vector load (tracked by vmcnt, so LDS waits
    // in the loop never wait for it) and moved to an SGPR only when it is consumed.
    const uint32_t sec_size = Uni(P.sec_size[sec]);
    if (sec_size == 0) continue;  // (this pass of the group has not arrived: a frame drawn from a prefix; the group was zeroed above)
    const uint32_t* stream = P.sections + Uni(P.sec_word[sec]);
    const uint32_t nwords = (sec_size + 3) / 4;
    uint32_t idx = 0;  // words moved into buf
    uint64_t buf = 0;
    uint32_t bits = 0;
    uint32_t next_v = stream[0];
#define JXL_REFILL()                                                       \
  if (bits < 32) {                                                         \
    buf |= uint64_t(Uni(next_v)) << bits;                                  \
    bits += 32;                                                            \
    idx++;                                                                 \
    next_v = stream[idx < nwords ? idx : nwords]; /* word nwords = 0 pad */ \
  }
#define JXL_READ(n, out)                                                            \
  {                                                                                 \
    const uint32_t n_ = (n);                                                        \
    out = uint32_t(buf) & ((n_ >= 32) ? 0xFFFFFFFFu : ((1u << n_) - 1));            \
    buf >>= n_;                                                                     \
    bits -= n_;                                                                     \
  }
    JXL_REFILL();
    uint32_t tmp;
    if (sec == 0 && P.first_bit_offset) JXL_READ(P.first_bit_offset, tmp);
    uint32_t hb = 0;
    while ((1u << hb) < P.num_hist) hb++;
    JXL_REFILL();
    uint32_t sel = 0;
    if (hb) JXL_READ(hb, sel);
    uint32_t err = 0;
    if (sel >= P.num_hist) {
      err = kErrSelector;
      sel = 0;
    }
    if (WPG == 1 && sel != 0) {  // re-stage the selected slice of the context map (single-wave workgroup)
      const uint8_t* src = T.ctx_map + size_t(sel) * P.nctx;
      for (uint32_t i = lane; i < P.nctx + 16; i += 64) l_ctx[i] = src[i];
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
    }
    uint32_t state, hi16;
    JXL_REFILL();
    JXL_READ(16, state);
    JXL_REFILL();
    JXL_READ(16, hi16);
    state |= hi16 << 16;
    const uint32_t log_entry = 12 - log_alpha;
    // One symbol from cluster `cl_` with uint config `cf_` into `out_` (all scalar).
#define JXL_SYMBOL(cl_, cf_, out_)                                                                         \
  {                                                                                                        \
    const uint32_t res_ = state & 0xFFFu;                                                                  \
    const uint32_t i_ = res_ >> log_entry, pos_ = res_ & ((1u << log_entry) - 1);                          \
    const uint2 ev_ = l_alias[((cl_) << log_alpha) + i_];                                                  \
    const uint32_t e0_ = Uni(ev_.x), e1_ = Uni(ev_.y);                                                     \
    const bool gt_ = pos_ >= (e0_ & 0xFF);                                                                 \
    uint32_t tok_ = gt_ ? ((e0_ >> 8) & 0xFF) : i_;                                                        \
    const uint32_t off_ = (gt_ ? (e1_ & 0xFFFF) : 0u) + pos_;                                              \
    const uint32_t freq_ = gt_ ? (e1_ >> 16) : (e0_ >> 16);                                                \
    state = freq_ * (state >> 12) + off_;                                                                  \
    JXL_REFILL();                                                                                          \
    if (state < (1u << 16)) {                                                                              \
      uint32_t lo_;                                                                                        \
      JXL_READ(16, lo_);                                                                                   \
      state = (state << 16) | lo_;                                                                         \
    }                                                                                                      \
    const uint32_t se_ = (cf_) & 0xFF, msb_ = ((cf_) >> 8) & 0xFF, lsb_ = ((cf_) >> 16) & 0xFF;            \
    const uint32_t st_ = 1u << se_;                                                                        \
    if (tok_ >= st_) {                                                                                     \
      const uint32_t nb_ = (se_ - (msb_ + lsb_) + ((tok_ - st_) >> (msb_ + lsb_))) & 31u;                  \
      const uint32_t low_ = tok_ & ((1u << lsb_) - 1);                                                     \
      const uint32_t hi_ = tok_ >> lsb_;                                                                   \
      JXL_REFILL();                                                                                        \
      uint32_t xb_;                                                                                        \
      JXL_READ(nb_, xb_);                                                                                  \
      tok_ = (((((1u << msb_) | (hi_ & ((1u << msb_) - 1))) << nb_) | xb_) << lsb_) | low_;                \
    }                                                                                                      \
    out_ = tok_;                                                                                           \
  }
    const CU32 ooff = AsConst(&T.order_offset[0]);
    for (uint32_t bi = b0; bi < b1 && !err; bi++) {
      const uint32_t w0 = blk[bi * 3], w1 = blk[bi * 3 + 1], coef_offset = blk[bi * 3 + 2];
      const uint32_t st = w1 & 0xFF, dcctx = (w1 >> 8) & 0xFF, qf = w1 >> 16;
      const uint32_t info = c_strategy_info[st];
      const uint32_t cx = info & 0xFF, cy = (info >> 8) & 0xFF, log2c = (info >> 16) & 0xFF, ord = info >> 24;
      const uint32_t covered = 1u << log2c, size = covered * 64;
      uint32_t qfi = 0;
      for (uint32_t t = 0; t + 1 < P.nq; t++) qfi += qf > P.qf_thr[t];
      const uint32_t fbx = w0 & 31, fby = (w0 >> 16) & 31;
#pragma unroll 1
      for (int ci = 0; ci < 3 && !err; ci++) {
        const uint32_t c = ci == 0 ? 1u : (ci == 1 ? 0u : 2u);
        const uint32_t hs = (P.cs >> (2 * c)) & 1u, vs = (P.cs >> (2 * c + 1)) & 1u;  // (subsampled channels: see k_entropy_ans)
        if ((fbx & hs) | (fby & vs)) continue;
        const uint32_t lbx = fbx >> hs, lby = fby >> vs;
        uint8_t* nzc = l_nz + c * 1024;
        uint32_t pred;
        if (lbx == 0) pred = lby ? Uni(nzc[(lby - 1) * 32]) : 32;
        else if (lby == 0) pred = Uni(nzc[lbx - 1]);
        else pred = (Uni(nzc[(lby - 1) * 32 + lbx]) + Uni(nzc[lby * 32 + lbx - 1]) + 1) >> 1;
        const uint32_t li = ((c * 13 + ord) * P.nq + qfi) * P.ndc + dcctx;
        const uint32_t bctx = (lut[li >> 2] >> ((li & 3) * 8)) & 0xFF;
        uint32_t nzb = pred >= 64 ? 64 : pred;
        nzb = nzb < 8 ? nzb : 4 + nzb / 2;
        uint32_t cl = Uni(l_ctx[nzb * P.num_bctx + bctx]);
        uint32_t cf = Uni(l_cfg[cl]);
        uint32_t nzeros;
        JXL_SYMBOL(cl, cf, nzeros);
        if (nzeros > size - covered) {
          err = kErrNzeros;
          break;
        }
        {
          const uint8_t nzv = uint8_t((nzeros + covered - 1) >> log2c);
          for (uint32_t i = lane; i < cx * cy; i += 64) nzc[(lby + i / cx) * 32 + lbx + i % cx] = nzv;
        }
        if (nzeros == 0) continue;
        const uint32_t hoff = P.num_bctx * 37 + 458 * bctx;
        const uint16_t* order = T.orders + ooff[ord * 3 + c];
        CoefT* dst = gco + size_t(c) * 65536 + coef_offset;
        uint32_t prev = nzeros > size / 16 ? 0 : 1;
        uint32_t k = covered;
        uint32_t nnz_cur = NnzCtx(((nzeros + covered - 1) >> log2c) & 63);
        uint32_t nnz_nxt = NnzCtx(((nzeros - 1 + covered - 1) >> log2c) & 63);
        cl = Uni(l_ctx[hoff + (nnz_cur + FreqCtx((k >> log2c) & 63)) * 2 + prev]);
        cf = Uni(l_cfg[cl]);
        uint32_t count = 0;
        for (;;) {
          // speculative lookups for coefficient k+1: A = this one is zero, B = this one is non-zero
          const uint32_t kn = k + 1;
          const uint32_t fq = FreqCtx((kn >> log2c) & 63);
          const uint8_t rawA = l_ctx[hoff + (nnz_cur + fq) * 2];
          const uint8_t rawB = l_ctx[hoff + (nnz_nxt + fq) * 2 + 1];
          const uint32_t cfA_v = l_cfg[rawA], cfB_v = l_cfg[rawB];
          uint32_t u;
          JXL_SYMBOL(cl, cf, u);
          if (u) {
            const uint32_t mag = u >> 1, neg = (~u) & 1;
            const uint32_t coeff = (mag ^ (neg - 1)) << shift;
            if (lane == 0) l_list[count] = make_uint2(k, coeff);
            count++;
            nzeros--;
            if (nzeros == 0) break;
            if (count == kList) {
              __builtin_amdgcn_s_waitcnt(0xC07F);
              __builtin_amdgcn_wave_barrier();
              for (uint32_t i = lane; i < kList; i += 64) {
                const uint2 e = l_list[i];
                const uint32_t pos = order[e.x];
                if (pass == 0) dst[pos] = CoefT(int32_t(e.y));
                else dst[pos] = CoefT(dst[pos] + int32_t(e.y));
              }
              count = 0;
            }
            nnz_cur = nnz_nxt;
            nnz_nxt = NnzCtx(((nzeros - 1 + covered - 1) >> log2c) & 63);
            cl = Uni(rawB);
            cf = Uni(cfB_v);
          } else {
            cl = Uni(rawA);
            cf = Uni(cfA_v);
          }
          k = kn;
          if (k >= size) break;
        }
        if (nzeros != 0) err = kErrNzeros;
        // scatter what is left in the list (all lanes)
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < count; i += 64) {
          const uint2 e = l_list[i];
          const uint32_t pos = order[e.x];
          if (pass == 0) dst[pos] = CoefT(int32_t(e.y));
          else dst[pos] = CoefT(dst[pos] + int32_t(e.y));
        }
      }
    }
#undef JXL_SYMBOL
#undef JXL_READ
#undef JXL_REFILL
    if (!err && state != (0x13u << 16)) err |= kErrFinalState;
    {
      const uint64_t consumed = uint64_t(idx) * 32 - uint64_t(bits);
      if (consumed > uint64_t(sec_size) * 8) err |= kErrOverread;
      if (lane == 0) P.sec_end_bits[sec] = uint32_t(consumed);
    }
    if (err && lane == 0) atomicOr(&P.errors[g], err);
  }
#undef P
}

// ---------------------------------------------------------------------------------------------- multi-pass merge
// Progressive frames on the lane kernel: every pass was decoded into its own scan-order buffer (its own coefficient
// order, dec_group.cc:335-338 adds the passes' coefficients); this kernel zero-fills the natural-layout buffer the
// transform kernels read (TransformParams::scan_order = 0) and adds each pass's valid scan prefix at its positions.
// One 64-thread workgroup per varblock; passes are separated by barriers (positions are unique within a pass only).
struct MergeParams {
  void* coeffs;  // natural-layout result at element offset 0, pass p's scan-order buffer at (1 + p) * pass_stride
  const JxlHipVarBlock* blocks;
  const PassDev* passes;
  const uint32_t* kend;  // [pass][block * 3 + channel]
  uint64_t pass_stride;
  uint32_t num_blocks, num_passes, xg, kend_stride;
};
template <typename CoefT>
__global__ __launch_bounds__(64) void k_merge_passes(MergeParams M) {
  const uint32_t b = blockIdx.x, t = threadIdx.x;
  if (b >= M.num_blocks) return;
  const JxlHipVarBlock vb = M.blocks[b];
  const uint32_t st = vb.strategy, covered = 1u << c_log2_covered[st], size = covered * 64, ord = c_strategy_order[st];
  const uint32_t g = (vb.by >> 5) * M.xg + (vb.bx >> 5);
  CoefT* const base = static_cast<CoefT*>(M.coeffs);
  for (uint32_t c = 0; c < 3; c++) {
    const size_t off = (size_t(g) * 3 + c) * 65536 + vb.coef_offset;
    CoefT* out = base + off;
    for (uint32_t i = t; i < size; i += 64) out[i] = CoefT(0);
    __syncthreads();
    for (uint32_t p = 0; p < M.num_passes; p++) {
      const CoefT* src = base + (1 + p) * M.pass_stride + off;
      const uint16_t* order = M.passes[p].orders + M.passes[p].order_offset[ord * 3 + c];
      const uint32_t ke0 = M.kend[size_t(p) * M.kend_stride + b * 3 + c];
      const uint32_t ke = ke0 < size ? ke0 : size;
      for (uint32_t k = covered + t; k < ke; k += 64) {
        const CoefT q = src[k];
        if (q) out[order[k]] = CoefT(out[order[k]] + q);
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------- transforms
struct TransformParams {
  const void* coeffs;
  uint32_t coef_bits;
  const JxlHipVarBlock* blocks;
  const float* dequant;
  uint32_t dq_offset[17], dq_size[17];
  const float* dc;  // 3 planes
  const int8_t* ytox;
  const int8_t* ytob;
  const float* basis_t;  // BT_N[k * N + n], N = 1..256, offset (N*N-1)/3
  const float* basis_n;  // the same matrices transposed: [n * N + k]
  float inv_global_scale, x_dm, b_dm, color_scale, base_x, base_b;
  float biases[4];
  uint32_t xb, yb, xg, xp, yp;
  float* out;  // 3 planes of xp * yp
  float* scratch;
  // Coefficient layout. scan_order = 0: natural (coefficient-layout raster) positions, whole block valid.
  // scan_order = 1: entry k of a (block, channel) is the coefficient at position orders[...][k] and only entries
  // [covered, kend) were written by the entropy stage (everything else is zero).
  uint32_t scan_order;
  const uint32_t* kend;    // [block * 3 + channel]
  const uint16_t* orders;  // pass 0 coefficient orders
  uint32_t order_offset[39];
  const float* dequant_scan;  // the dequant tables permuted into scan order: [dq_offset[kind] + c * dq_size[kind] + k] =
                              // dequant[... + orders[k]] (scan_order frames; lets the staging loop run without a dependent gather)
  // transform work lists: block indices bucketed by strategy (tlist[list_begin[s] .. + list_count[s]))
  const uint32_t* tlist;
  uint32_t list_begin[27], list_count[27];
  // the same lists as records (k_idct_fast: one load instead of list entry -> JxlHipVarBlock): x = bx | by << 16, y = raw quant
  // field, z = element offset of the varblock's coefficients in `coeffs` (channel 0; the channels are 65536 apart), w = the
  // varblock's index (kend)
  const uint4* trecs;
  // Chroma-subsampled YCbCr frames (JxlHipFrameDesc::chroma_hshift / _vshift; dec_group.cc:443-451): cs = hshift of channel c
  // in bit 2 * c, vshift in bit 2 * c + 1 (0 = 4:4:4). A varblock carries channel c only when it lies on the channel's grid;
  // the pixels then go to block (bx >> hshift, by >> vshift) of the channel's plane in `cs_out` (same geometry as `out`: the
  // channel fills the top-left part), for k_chroma_upsample to spread into `out`; channels with no shift go to `out` directly.
  // The lowest frequencies come from the channel's own DC sample, at the same shifted position of its DC plane.
  uint32_t cs;
  float* cs_out;
};

// Batched launches: one workgroup descriptor per workgroup = {frame index into the parameter array, index of the
// workgroup's first varblock in that frame's list of the launched strategy}.
#define JXL_TRANSFORM_PREAMBLE()                                   \
  const uint2 wgd = desc[blockIdx.x];                              \
  /* the frame's parameter block through the constant address space: scalar loads the compiler may hoist */ \
  const TransformParams __attribute__((address_space(4)))& P =                                                \
      *(const TransformParams __attribute__((address_space(4)))*)(uintptr_t)(params + wgd.x);                 \
  const uint32_t n = P.list_count[strategy];                       \
  const uint32_t* list = P.tlist + P.list_begin[strategy];

template <typename BiasPtr>
__device__ __forceinline__ float QuantBias(int c, int q, BiasPtr b) {
  if (q == 0) return 0.0f;
  if (q == 1) return b[c];
  if (q == -1) return -b[c];
  const float qf = float(q);
  return qf - b[3] * (1.0f / qf);
}

// The same for q != 0 as selects, with the hardware reciprocal (1 ulp; the result is a dequantised coefficient compared at
// 2e-5 absolute after the transform). q == 0 gives an unspecified value: the caller does not store it.
template <typename BiasPtr>
__device__ __forceinline__ float QuantBiasNoBranch(int c, int q, BiasPtr b) {
  const float qf = float(q);
  const float big = qf - b[3] * __builtin_amdgcn_rcpf(qf);
  const float one = __builtin_copysignf(b[c], qf);
  return (q == 1 || q == -1) ? one : big;
}

// Stages the dequantised coefficients of channel c of one varblock into LDS/scratch `l` at their natural positions
// (lib/jxl/dec_group.cc:115-181: dequantisation with AdjustQuantBias, then chroma-from-luma for X and B from the
// already staged dequantised Y in `l_y`). `gq` = the block's coefficients of channel c, `m` = channel c's dequant
// table, `mul` = inv_global_scale / qf (times the channel's x_dm / b_dm multiplier), `cc` = CfL factor (0 for Y).
template <typename CoefT, typename PT>
__device__ __forceinline__ void StageChannel(const PT& P, const CoefT* gq, uint32_t block_index, int c, uint32_t ord,
                                             uint32_t size, uint32_t covered, const float* m, float mul, float cc, const float* l_y,
                                             float* l, uint32_t t, uint32_t nthreads) {
  if (P.scan_order) {
    const uint16_t* order = P.orders + P.order_offset[ord * 3 + c];
    const uint32_t kend = P.kend[block_index * 3 + c];
    for (uint32_t k = t; k < size; k += nthreads) {
      const uint32_t pos = order[k];
      const int q = (k >= covered && k < kend) ? int(gq[k]) : 0;
      float v = QuantBias(c, q, P.biases) * (m[pos] * mul);
      if (c != 1) v += cc * l_y[pos];
      l[pos] = v;
    }
  } else {
    for (uint32_t k = t; k < size; k += nthreads) {
      float v = QuantBias(c, int(gq[k]), P.biases) * (m[k] * mul);
      if (c != 1) v += cc * l_y[k];
      l[k] = v;
    }
  }
}

__device__ __forceinline__ uint32_t BasisOffset(uint32_t n) { return (n * n - 1) / 3; }

// Lowest-frequency coefficient (ky, kx) of a CY x CX varblock from the DC image (scaled forward DCT of the
// covered DC samples times the resample scales).
template <int CX, int CY, typename PT>
__device__ __forceinline__ float LlfFromDc(const PT& P, const float* dc, int ky, int kx) {
  const float* bty = P.basis_t + BasisOffset(CY);
  const float* btx = P.basis_t + BasisOffset(CX);
  float s = 0.0f;
  for (int y = 0; y < CY; y++) {
    float r = 0.0f;
    for (int x = 0; x < CX; x++) r += dc[y * P.xb + x] * btx[kx * CX + x];
    s += r * bty[ky * CY + y];
  }
  s *= 1.0f / float(CX * CY);
  return s * c_resample[CY - 1 + ky] * c_resample[CX - 1 + kx];
}

// DCT-family strategies up to 64x64: per channel, dequantised coefficients are staged in LDS and the separable
// inverse transform is evaluated as two matrix passes (LDS-resident tile, coalesced plane writes).
template <typename CoefT, int CX, int CY>
__global__ __launch_bounds__(256) void k_dct(const TransformParams* params, const uint2* desc, uint32_t strategy) {
  JXL_TRANSFORM_PREAMBLE();
  constexpr int R = CY * 8, C = CX * 8, SIZE = R * C;
  constexpr int TPB = SIZE >= 256 ? 256 : SIZE;
  constexpr int BPW = 256 / TPB;
  extern __shared__ __align__(16) float lds_f[];
  const int sub = threadIdx.x / TPB, t = threadIdx.x % TPB;
  float* l_y = lds_f + sub * 3 * SIZE;  // dequantised Y stays resident for the chroma-from-luma of X and B
  float* l_xb = l_y + SIZE;
  float* l_tmp = l_xb + SIZE;
  const uint32_t li = wgd.y + sub;
  const bool active = li < n;
  JxlHipVarBlock vb;
  const CoefT* gq = nullptr;
  const float* m = nullptr;
  uint32_t msize = 0, bidx = 0;
  float sc = 0, x_cc = 0, b_cc = 0;
  if (active) {
    bidx = list[li];
    vb = P.blocks[bidx];
    const uint32_t g = (vb.by >> 5) * P.xg + (vb.bx >> 5);
    gq = static_cast<const CoefT*>(P.coeffs) + size_t(g) * 3 * 65536 + vb.coef_offset;
    const uint32_t kind = c_strategy_qtable[strategy];
    m = P.dequant + P.dq_offset[kind];
    msize = P.dq_size[kind];
    sc = P.inv_global_scale / float(vb.qf);
    const uint32_t tiles_x = (P.xb + 7) / 8;
    const uint32_t tile = (vb.by / 8) * tiles_x + vb.bx / 8;
    x_cc = P.base_x + float(P.ytox[tile]) * P.color_scale;
    b_cc = P.base_b + float(P.ytob[tile]) * P.color_scale;
  }
  const float* btc = P.basis_t + BasisOffset(C);
  const float* btr = P.basis_t + BasisOffset(R);
  for (int ci = 0; ci < 3; ci++) {
    const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
    float* l_coef = c == 1 ? l_y : l_xb;
    if (active) {
      const float mul = c == 1 ? sc : sc * (c == 0 ? P.x_dm : P.b_dm);
      StageChannel<CoefT>(P, gq + size_t(c) * 65536, bidx, c, c_strategy_order[strategy], SIZE, CX * CY, m + size_t(c) * msize, mul,
                          c == 0 ? x_cc : b_cc, l_y, l_coef, t, TPB);
    }
    __syncthreads();
    if (active && t < CX * CY) {
      const float* dc = P.dc + size_t(c) * P.xb * P.yb + size_t(vb.by) * P.xb + vb.bx;
      // LLF corner: rows = short side
      constexpr int LC = CX > CY ? CX : CY;  // long side
      const int row = t / LC, col = t % LC;
      int ky, kx;
      if (CY < CX) { ky = row; kx = col; } else { kx = row; ky = col; }
      l_coef[row * (LC * 8) + col] = LlfFromDc<CX, CY>(P, dc, ky, kx);
    }
    __syncthreads();
    if (active) {
      // pass A: tmp[ky][x] = sum_kx coef(ky,kx) * B_C[x][kx]
      for (int i = t; i < SIZE; i += TPB) {
        const int ky = i / C, x = i % C;
        float s = 0.0f;
        if (R < C) {
#pragma unroll 8
          for (int kx = 0; kx < C; kx++) s += l_coef[ky * C + kx] * btc[kx * C + x];
        } else {
#pragma unroll 8
          for (int kx = 0; kx < C; kx++) s += l_coef[kx * R + ky] * btc[kx * C + x];
        }
        l_tmp[i] = s;
      }
    }
    __syncthreads();
    if (active) {
      float* out = P.out + size_t(c) * P.xp * P.yp + size_t(vb.by) * 8 * P.xp + size_t(vb.bx) * 8;
      for (int i = t; i < SIZE; i += TPB) {
        const int y = i / C, x = i % C;
        float s = 0.0f;
#pragma unroll 8
        for (int ky = 0; ky < R; ky++) s += l_tmp[ky * C + x] * btr[ky * R + y];
        out[size_t(y) * P.xp + x] = s;
      }
    }
    __syncthreads();
  }
}

// DCT-family strategies up to 32x32, column-thread form. C = CX * 8 threads own one varblock, thread x owns column x:
//   * dequantised coefficients are staged in LDS as [ky][kx] (zero / chroma-from-luma fill, then only the entropy
//     stage's valid scan-order prefix is scattered through the coefficient order);
//   * row pass:    tmp[ky] = sum_kx coef[ky][kx] * B_C[kx][x]   -- the thread keeps its basis column B_C[:][x] in
//     registers and reads coefficient rows as LDS broadcasts (one ds_read_b128 per 4 FMAs, same address for the group);
//   * column pass: out[y][x] = sum_ky tmp[ky] * B_R[ky][y]      -- tmp[] never leaves registers, the basis rows are LDS
//     broadcasts as well (scalar-cache loads were measured to serialise on their latency), no second barrier;
//   * row y of the block is written by the C threads of the group as one contiguous segment.
typedef const float __attribute__((address_space(4)))* CF32;

constexpr int kIdctColsThreads = 128;  // small workgroups: their LDS has to fit beside the resident entropy workgroups

template <typename CoefT, int CX, int CY>
__global__ __launch_bounds__(kIdctColsThreads) void k_idct_cols(const TransformParams* params, const uint2* desc, uint32_t strategy) {
  JXL_TRANSFORM_PREAMBLE();
  constexpr int R = CY * 8, C = CX * 8, SIZE = R * C, GSTRIDE = 2 * SIZE + 4;
  extern __shared__ __align__(16) float lds_f[];
  const int grp = threadIdx.x / C, x = threadIdx.x % C;
  float* l_bn = lds_f;                          // column-pass basis [y][ky], read as broadcasts
  float* l_y = lds_f + R * R + grp * GSTRIDE;  // dequantised Y stays resident for the chroma-from-luma of X and B
  float* l_xb = l_y + SIZE;
  const uint32_t li = wgd.y + grp;
  const bool active = li < n;
  JxlHipVarBlock vb;
  const CoefT* gq = nullptr;
  const float* m = nullptr;
  uint32_t msize = 0, bidx = 0;
  float sc = 0, x_cc = 0, b_cc = 0;
  if (active) {
    bidx = list[li];
    vb = P.blocks[bidx];
    const uint32_t g = (vb.by >> 5) * P.xg + (vb.bx >> 5);
    gq = static_cast<const CoefT*>(P.coeffs) + size_t(g) * 3 * 65536 + vb.coef_offset;
    const uint32_t kind = c_strategy_qtable[strategy];
    m = P.dequant + P.dq_offset[kind];
    msize = P.dq_size[kind];
    sc = P.inv_global_scale / float(vb.qf);
    const uint32_t tiles_x = (P.xb + 7) / 8;
    const uint32_t tile = (vb.by / 8) * tiles_x + vb.bx / 8;
    x_cc = P.base_x + float(P.ytox[tile]) * P.color_scale;
    b_cc = P.base_b + float(P.ytob[tile]) * P.color_scale;
  }
  float breg[C];
  {
    const float* bt = P.basis_t + BasisOffset(C);
#pragma unroll
    for (int kx = 0; kx < C; kx++) breg[kx] = bt[kx * C + x];
  }
  {
    const float* bn = P.basis_n + BasisOffset(R);  // [y * R + ky]
    for (int i = threadIdx.x; i < R * R; i += kIdctColsThreads) l_bn[i] = bn[i];
  }
  const uint32_t ord = c_strategy_order[strategy];
  for (int ci = 0; ci < 3; ci++) {
    const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
    float* l = c == 1 ? l_y : l_xb;
    if (active) {
      if (c == 1) {
        for (int i = x * 4; i < SIZE; i += C * 4) *reinterpret_cast<float4*>(l + i) = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const float cc = c == 0 ? x_cc : b_cc;
        for (int i = x * 4; i < SIZE; i += C * 4) {
          const float4 y4 = *reinterpret_cast<const float4*>(l_y + i);
          *reinterpret_cast<float4*>(l + i) = make_float4(cc * y4.x, cc * y4.y, cc * y4.z, cc * y4.w);
        }
      }
    }
    __syncthreads();
    if (active) {
      const float mul = c == 1 ? sc : sc * (c == 0 ? P.x_dm : P.b_dm);
      const CoefT* gqc = gq + size_t(c) * 65536;
      const float* mc = m + size_t(c) * msize;
      if (P.scan_order) {
        // entry k: coefficient, its position and its dequant weight are three independent coalesced loads
        const uint16_t* order = P.orders + P.order_offset[ord * 3 + c];
        const float* ms = P.dequant_scan + (mc - P.dequant);
        const uint32_t ke = P.kend[bidx * 3 + c];
        const uint32_t k1 = ke < uint32_t(SIZE) ? ke : uint32_t(SIZE);
#pragma unroll 4
        for (uint32_t k = CX * CY + x; k < k1; k += C) {
          const int q = int(gqc[k]);
          const uint32_t pos = order[k];
          const float w = ms[k];
          if (q) {
            const uint32_t idx = R < C ? pos : (pos % R) * C + pos / R;  // natural layout keeps the short side as rows
            l[idx] += QuantBias(c, q, P.biases) * (w * mul);
          }
        }
      } else {
        for (uint32_t k = x; k < uint32_t(SIZE); k += C) {
          const int q = int(gqc[k]);
          if (q) {
            const uint32_t idx = R < C ? k : (k % R) * C + k / R;
            l[idx] += QuantBias(c, q, P.biases) * (mc[k] * mul);
          }
        }
      }
      if (x < CX * CY) {  // lowest frequencies from the DC image
        const float* dc = P.dc + size_t(c) * P.xb * P.yb + size_t(vb.by) * P.xb + vb.bx;
        const int ky = x / CX, kx = x % CX;
        l[ky * C + kx] = LlfFromDc<CX, CY>(P, dc, ky, kx);
      }
    }
    __syncthreads();
    if (active) {
      float tmp[R];
#pragma unroll
      for (int ky = 0; ky < R; ky++) {
        float acc = 0.0f;
#pragma unroll
        for (int k4 = 0; k4 < C; k4 += 4) {
          const float4 v = *reinterpret_cast<const float4*>(l + ky * C + k4);
          acc += v.x * breg[k4];
          acc += v.y * breg[k4 + 1];
          acc += v.z * breg[k4 + 2];
          acc += v.w * breg[k4 + 3];
        }
        tmp[ky] = acc;
      }
      float* out = P.out + size_t(c) * P.xp * P.yp + size_t(vb.by) * 8 * P.xp + size_t(vb.bx) * 8 + x;
#pragma unroll
      for (int y = 0; y < R; y++) {
        float acc = 0.0f;
#pragma unroll
        for (int k4 = 0; k4 < R; k4 += 4) {
          const float4 v = *reinterpret_cast<const float4*>(l_bn + y * R + k4);
          acc += tmp[k4] * v.x;
          acc += tmp[k4 + 1] * v.y;
          acc += tmp[k4 + 2] * v.z;
          acc += tmp[k4 + 3] * v.w;
        }
        out[size_t(y) * P.xp] = acc;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// DCT-family strategies up to 32x32, fast form: the separable inverse DCT as two passes of the recursive even / odd
// decomposition (the algorithm of lib/jxl/dct-inl.h:191-232, IDCT1DImpl<N>: N/2-point IDCT of the even coefficients;
// N/2-point IDCT of the odd ones after d[j] = c[2j+1] + c[2j-1], d[0] = sqrt(2) c[1]; odd half scaled by
// WcMultipliers<N>[n] = 1 / (2 cos((n + 1/2) pi / N)), dct_scales.h:234-236; out[n], out[N-1-n] = even +- odd), each
// 1-D transform entirely in the registers of one thread: ~9 flops per sample and pass instead of N FMAs, and two plain
// LDS accesses per sample and pass instead of N/4 broadcast reads.
//   * max(R, C) threads own one varblock; its dequantised coefficients are staged in ONE LDS tile [ky][kx] (row stride
//     C + 1: both the per-row and the per-column accesses are conflict-free);
//   * pass 1: thread ky transforms row ky in place; pass 2: thread x transforms column x and writes it to the plane,
//     row segments contiguous across the threads;
//   * chroma from luma is applied to the PIXELS (the transform is linear): the thread keeps its column of Y output and
//     adds cc * Y to the X / B column. The reference adds cc * Y to the coefficients except the lowest-frequency corner
//     (which comes from the DC image, dec_group.cc:115-181); that corner is therefore staged as LLF_c - cc * LLF_Y. This
//     is what lets one tile per varblock do: the dequantised Y coefficients need not stay resident.
template <int N>
struct WcTable;
template <>
struct WcTable<2> {
  static constexpr float v[1] = {7.071067812e-01f};
};
template <>
struct WcTable<4> {
  static constexpr float v[2] = {5.411961001e-01f, 1.306562965e+00f};
};
template <>
struct WcTable<8> {
  static constexpr float v[4] = {5.097955791e-01f, 6.013448869e-01f, 8.999762231e-01f, 2.562915448e+00f};
};
template <>
struct WcTable<16> {
  static constexpr float v[8] = {5.024192862e-01f, 5.224986149e-01f, 5.669440348e-01f, 6.468217834e-01f, 7.881546235e-01f, 1.060677686e+00f, 1.722447098e+00f, 5.101148619e+00f};
};
template <>
struct WcTable<32> {
  static constexpr float v[16] = {5.006029982e-01f, 5.054709599e-01f, 5.154473099e-01f, 5.310425911e-01f, 5.531038960e-01f, 5.829349682e-01f, 6.225041230e-01f, 6.748083415e-01f, 7.445362710e-01f, 8.393496454e-01f, 9.725682379e-01f, 1.169439933e+00f, 1.484164616e+00f, 2.057781010e+00f, 3.407608418e+00f, 1.019000812e+01f};
};

template <>
struct WcTable<64> {  // 1 / (2 cos((n + 1/2) pi / 64)), n = 0..31 (dct_scales.h:234-236 WcMultipliers<64>)
  static constexpr float v[32] = {5.001506360e-01f, 5.013584524e-01f, 5.037887257e-01f, 5.074711721e-01f, 5.124514794e-01f, 5.187927131e-01f, 5.265773152e-01f, 5.359098169e-01f, 5.469204380e-01f, 5.597698129e-01f, 5.746551840e-01f, 5.918185359e-01f, 6.115573479e-01f, 6.342389367e-01f, 6.603198078e-01f, 6.903721282e-01f, 7.251205224e-01f, 7.654941650e-01f, 8.127020908e-01f, 8.683447152e-01f, 9.345835970e-01f, 1.014408265e+00f, 1.112071621e+00f, 1.233832738e+00f, 1.389293959e+00f, 1.593972283e+00f, 1.874675980e+00f, 2.282050068e+00f, 2.924628428e+00f, 4.084611078e+00f, 6.796750712e+00f, 2.037387817e+01f};
};

template <int N>
__device__ __forceinline__ void FastIdct(float (&v)[N]) {
  if constexpr (N == 2) {
    const float a = v[0] + v[1], b = v[0] - v[1];
    v[0] = a;
    v[1] = b;
  } else if constexpr (N > 2) {
    float e[N / 2], o[N / 2];
#pragma unroll
    for (int j = 0; j < N / 2; j++) {
      e[j] = v[2 * j];
      o[j] = v[2 * j + 1];
    }
#pragma unroll
    for (int j = N / 2 - 1; j > 0; j--) o[j] += o[j - 1];
    o[0] *= 1.41421356237309504880f;
    FastIdct<N / 2>(e);
    FastIdct<N / 2>(o);
#pragma unroll
    for (int n = 0; n < N / 2; n++) {
      const float t = o[n] * WcTable<N>::v[n];
      v[n] = e[n] + t;
      v[N - 1 - n] = e[n] - t;
    }
  }
}

// Tile hand-over inside a ONE-WAVE workgroup: the LDS operations of a wave execute in order, so a later read sees an
// earlier write of any lane without a wait; only the compiler must not reorder them. __syncthreads() here would be a
// workgroup-scope fence over global memory too: s_waitcnt vmcnt(0), i.e. every channel would wait for the previous
// channel's row stores to be acknowledged before it may start (the transform kernels sat in such waits for 82 % of their
// wave cycles, profiles/r02_sq_counters_isolated.txt).
__device__ __forceinline__ void WaveLdsSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// threads per workgroup: one wave (e.g. two 32x32 varblocks, 8.4 KB of LDS; eight 8x8 ones), so that a workgroup fits into
// the LDS the resident entropy workgroups leave free and a barrier stalls one wave only
__host__ __device__ constexpr int IdctFastThreads(int cx, int cy) { return (void(cx), void(cy), 64); }

// CS: the frame is chroma subsampled (TransformParams::cs; the 8x8 class only: such frames have no larger varblocks).
template <typename CoefT, int CX, int CY, bool CS = false>
// (the 32-point instantiations need ~150 registers for a column of Y output plus the transform's temporaries: at four
// waves per SIMD they spilled 8-21 of them to scratch; the 64-point ones, one varblock per wave, take two waves' share)
__global__ __launch_bounds__(IdctFastThreads(CX, CY)) __attribute__((amdgpu_waves_per_eu((CX == 8 || CY == 8) ? 2 : ((CX == 4 || CY == 4) ? 3 : 4), 8))) void k_idct_fast(const TransformParams* params, const uint2* desc, uint32_t strategy) {
  JXL_TRANSFORM_PREAMBLE();
  static_assert(!CS || (CX == 1 && CY == 1), "chroma-subsampled frames: varblocks of one block only");
  static_assert(IdctFastThreads(CX, CY) == 64, "WaveLdsSync: the workgroup is one wave");
  constexpr int R = CY * 8, C = CX * 8, SIZE = R * C, TB = R > C ? R : C, S = C + 1, TILE = R * S;
  constexpr int LOGC = CX == 1 ? 3 : (CX == 2 ? 4 : (CX == 4 ? 5 : 6));
  extern __shared__ __align__(16) float lds_f[];
  const int grp = threadIdx.x / TB, t = threadIdx.x % TB;
  float* l = lds_f + grp * TILE;
  const uint32_t li = wgd.y + grp;
  const bool active = li < n;
  struct {
    uint32_t bx, by, qf;
  } vb = {0, 0, 1};
  const CoefT* gq = nullptr;
  const float* m = nullptr;
  uint32_t msize = 0, bidx = 0;
  float sc = 0, x_cc = 0, b_cc = 0;
  if (active) {
    const uint4 rec = P.trecs[P.list_begin[strategy] + li];
    vb.bx = rec.x & 0xFFFFu;
    vb.by = rec.x >> 16;
    vb.qf = rec.y;
    bidx = rec.w;
    gq = static_cast<const CoefT*>(P.coeffs) + rec.z;
    const uint32_t kind = c_strategy_qtable[strategy];
    m = P.dequant + P.dq_offset[kind];
    msize = P.dq_size[kind];
    sc = P.inv_global_scale / float(vb.qf);
    const uint32_t tiles_x = (P.xb + 7) / 8;
    const uint32_t tile = (vb.by / 8) * tiles_x + vb.bx / 8;
    x_cc = P.base_x + float(P.ytox[tile]) * P.color_scale;
    b_cc = P.base_b + float(P.ytob[tile]) * P.color_scale;
  }
  const uint32_t ord = c_strategy_order[strategy];
  float llf_y = 0.0f;  // this thread's lowest-frequency coefficient of Y (threads t < CX * CY)
  float yout[R];  // this thread's column of the Y output (threads t < C)
#pragma unroll
  for (int y = 0; y < R; y++) yout[y] = 0.0f;
  // Scan-order input: everything a channel's staging reads from memory (four coefficients, their positions and their
  // weights per round) is requested a whole channel ahead, all rounds at once, and the three channels' coefficient counts
  // with the block record: the kernel sat in memory waits for 82 % of its wave cycles (profiles/r04_sq_counters_*.txt)
  // with one dependent round trip after the other (list -> block -> count -> coefficients, the last two per channel).
  struct alignas(sizeof(CoefT) * 4) Coef4 {
    CoefT v[4];
  };
  constexpr int ROUNDS = SIZE / (TB * 4);     // rounds of four scan positions per thread: 2 (8x8) .. 16 (64x64)
  constexpr bool PF = ROUNDS <= 4;            // (the larger classes keep their loads inside the staging loop)
  constexpr int PR = PF ? ROUNDS : 1;
  // (the coefficients are read exactly once: with the non-temporal hint they do not push the tables and the other stages'
  // lines out of the L2: all transform launches 24.6 -> 24.0 ms alone, the pipelined step 59.6 -> 58.6 ms)
  auto load_coef4 = [](const CoefT* p) -> Coef4 {
    typedef uint32_t nt_u2 __attribute__((ext_vector_type(2)));
    typedef uint32_t nt_u4 __attribute__((ext_vector_type(4)));
    if constexpr (sizeof(CoefT) == 2) return __builtin_bit_cast(Coef4, __builtin_nontemporal_load(reinterpret_cast<const nt_u2*>(p)));
    else return __builtin_bit_cast(Coef4, __builtin_nontemporal_load(reinterpret_cast<const nt_u4*>(p)));
  };
  Coef4 pq[PR];
  ushort4 pp[PR];
  float4 pw[PR];
  uint32_t ke3[3] = {0, 0, 0};
  auto prefetch = [&](int c) {
    const CoefT* gqc = gq + size_t(c) * 65536;
    const uint16_t* order = P.orders + P.order_offset[ord * 3 + c];
    const float* ms = P.dequant_scan + ((m + size_t(c) * msize) - P.dequant);
#pragma unroll
    for (int r = 0; r < PR; r++) {
      const uint32_t k4 = uint32_t(t) * 4 + uint32_t(r) * TB * 4;
      pq[r] = load_coef4(gqc + k4);
      pp[r] = *reinterpret_cast<const ushort4*>(order + k4);
      pw[r] = *reinterpret_cast<const float4*>(ms + k4);
    }
  };
  const bool pf = PF && P.scan_order != 0;
  if (pf && active) {
#pragma unroll
    for (int c = 0; c < 3; c++) ke3[c] = P.kend[bidx * 3 + c];
    prefetch(1);
  }
#pragma unroll
  for (int ci = 0; ci < 3; ci++) {
    const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
    const float cc = c == 1 ? 0.0f : (c == 0 ? x_cc : b_cc);
    // (CS) does this varblock carry channel c, and where do the channel's DC sample and pixels sit
    const uint32_t hs = CS ? (P.cs >> (2 * c)) & 1u : 0u, vs = CS ? (P.cs >> (2 * c + 1)) & 1u : 0u;
    const bool act_c = active && (!CS || ((uint32_t(vb.bx) & hs) | (uint32_t(vb.by) & vs)) == 0);
    const uint32_t obx = active ? uint32_t(vb.bx) >> hs : 0u, oby = active ? uint32_t(vb.by) >> vs : 0u;
    if (!pf) {
      if (active)
        for (int i = t * 4; i < TILE; i += TB * 4) *reinterpret_cast<float4*>(l + i) = make_float4(0.f, 0.f, 0.f, 0.f);
      WaveLdsSync();
    }
    if (active) {
      const float mul = c == 1 ? sc : sc * (c == 0 ? P.x_dm : P.b_dm);
      const CoefT* gqc = gq + size_t(c) * 65536;
      const float* mc = m + size_t(c) * msize;
      if (pf) {
        // this channel's rounds are in registers; the next channel's leave now
        Coef4 cq[PR];
        ushort4 cp[PR];
        float4 cw[PR];
#pragma unroll
        for (int r = 0; r < PR; r++) {
          cq[r] = pq[r];
          cp[r] = pp[r];
          cw[r] = pw[r];
        }
        if (ci < 2) prefetch(ci == 0 ? 0 : 2);
        const uint32_t ke = ke3[c];
        const uint32_t k1 = ke < uint32_t(SIZE) ? ke : uint32_t(SIZE);
#pragma unroll
        for (int r = 0; r < PR; r++) {
          const uint32_t k4 = uint32_t(t) * 4 + uint32_t(r) * TB * 4;
          const uint32_t pos4[4] = {cp[r].x, cp[r].y, cp[r].z, cp[r].w};
          const float wv[4] = {cw[r].x, cw[r].y, cw[r].z, cw[r].w};
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const uint32_t k = k4 + j;
            const int q = int(cq[r].v[j]);
            const uint32_t pos = pos4[j];
            const uint32_t idx = R < C ? pos : (pos % R) * C + pos / R;  // natural layout keeps the short side as rows
            const float val = QuantBiasNoBranch(c, q, P.biases) * (wv[j] * mul);
            // every position of the tile is written exactly once by these rounds, so nothing zeroes it first and no store
            // is predicated (24 EXEC regions with their branches per thread otherwise): zero where there is no coefficient
            // (beyond the count, or a zero), and at the lowest-frequency corner, which the wave overwrites below (its LDS
            // operations execute in order)
            l[(idx >> LOGC) * S + (idx & (C - 1))] = (k >= uint32_t(CX * CY) && k < k1 && q) ? val : 0.0f;
          }
        }
      } else if (P.scan_order) {
        // entry k: coefficient, its position and its dequant weight are three independent coalesced loads
        const uint16_t* order = P.orders + P.order_offset[ord * 3 + c];
        const float* ms = P.dequant_scan + (mc - P.dequant);
        const uint32_t ke = P.kend[bidx * 3 + c];
        const uint32_t k1 = ke < uint32_t(SIZE) ? ke : uint32_t(SIZE);
        // four scan positions per thread and round: one 8-byte (int16) coefficient load, one 8-byte load of their positions
        // and one 16-byte load of their weights instead of four rounds of 2 + 2 + 4 bytes (the kernel is bound by the number
        // of its small memory operations, not by their bytes); all three tables are 16-byte aligned per (block, channel)
        for (uint32_t k4 = uint32_t(t) * 4; k4 < k1; k4 += TB * 4) {
          const Coef4 q4 = load_coef4(gqc + k4);
          const ushort4 p4 = *reinterpret_cast<const ushort4*>(order + k4);
          const float4 w4 = *reinterpret_cast<const float4*>(ms + k4);
          const uint32_t pos4[4] = {p4.x, p4.y, p4.z, p4.w};
          const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const uint32_t k = k4 + j;
            const int q = int(q4.v[j]);
            // value and address for every entry, without branches; only the store is conditional (a plain store: the tile
            // was zeroed and every position has one coefficient; entries below the lowest-frequency corner / beyond kend
            // are unspecified and skipped)
            const uint32_t pos = pos4[j];
            const uint32_t idx = R < C ? pos : (pos % R) * C + pos / R;  // natural layout keeps the short side as rows
            const float val = QuantBiasNoBranch(c, q, P.biases) * (wv[j] * mul);
            if (k >= uint32_t(CX * CY) && k < k1 && q) l[(idx >> LOGC) * S + (idx & (C - 1))] = val;
          }
        }
      } else {
        for (uint32_t k = t; k < uint32_t(SIZE); k += TB) {
          const int q = int(gqc[k]);
          if (q) {
            const uint32_t idx = R < C ? k : (k % R) * C + k / R;
            l[(idx >> LOGC) * S + (idx & (C - 1))] = QuantBias(c, q, P.biases) * (mc[k] * mul);
          }
        }
      }
      if (t < CX * CY) {  // lowest frequencies from the DC image (minus the part the pixel-domain chroma from luma adds back)
        const float* dc = P.dc + size_t(c) * P.xb * P.yb + size_t(oby) * P.xb + obx;
        const int ky = t / CX, kx = t % CX;
        float v = LlfFromDc<CX, CY>(P, dc, ky, kx);
        if (c == 1) llf_y = act_c ? v : 0.0f;  // (Y comes first)
        else v -= cc * llf_y;
        l[ky * S + kx] = v;
      }
    }
    WaveLdsSync();
    if (act_c && t < R) {  // pass 1: row ky = t
      float v[C];
#pragma unroll
      for (int kx = 0; kx < C; kx++) v[kx] = l[t * S + kx];
      FastIdct<C>(v);
#pragma unroll
      for (int x = 0; x < C; x++) l[t * S + x] = v[x];
    }
    WaveLdsSync();
    if (act_c && t < C) {  // pass 2: column x = t
      float v[R];
#pragma unroll
      for (int ky = 0; ky < R; ky++) v[ky] = l[ky * S + t];
      FastIdct<R>(v);
      // rows through a raw buffer: one 32-bit lane offset for the column, the row's offset in a scalar register (24 64-bit
      // address computations per thread otherwise); three planes of at most 1 GiB: the offsets fit 32 bits.
      // (Measured and not kept, round 4: rows in the last pass, so that a thread stores C adjacent pixels as 16-byte pieces
      // - a quarter of the store instructions, but 64 scattered pieces each instead of whole 32-byte-and-longer row segments:
      // all transform launches 25.6 -> 34.2 ms; a wave walking several sets of varblocks: no faster; the staging loads through
      // raw buffers with two register sets instead of copies: 62 vector instructions fewer per wave of the 8x8 class, 12 more
      // registers, 24.6 -> 26.0 ms.)
      // The 8x8 class stores with the non-temporal hint (cache policy bit `nt`): its 32-byte row segments cost 13 % less
      // that way (8.72 -> 7.59 ms per 640 frames alone; the stores are 44 % of this launch: 4.9 ms without them), while the
      // 128-byte rows of the larger classes cost MORE with it (32x32: 8.40 -> 9.78 ms) and keep the default policy; sc0 and
      // sc0 + nt are worse everywhere (profiles/r04_transform_store_policy.txt).
      constexpr int kStoreAux = (CX == 1 && CY == 1) ? 2 : 0;
      const __amdgpu_buffer_rsrc_t out_buf = __builtin_amdgcn_make_buffer_rsrc((CS && (hs | vs)) ? P.cs_out : P.out, 0, 0xFFFFFFFFu, 0x00020000);
      const uint32_t voff = (uint32_t(c) * P.xp * P.yp + oby * 8 * P.xp + obx * 8 + uint32_t(t)) * 4u;
      const uint32_t row_bytes = P.xp * 4u;
#pragma unroll
      for (int y = 0; y < R; y++) {
        float r = v[y];
        if (c == 1) yout[y] = r;
        else r += cc * yout[y];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, r), out_buf, voff, uint32_t(y) * row_bytes, kStoreAux);
      }
    }
    WaveLdsSync();
  }
}

// 128/256-class transforms: one workgroup per (varblock, channel), intermediates in global scratch.
template <typename CoefT>
__global__ __launch_bounds__(256) void k_dct_big(TransformParams P, const uint32_t* list, uint32_t n, uint32_t strategy) {
  const uint32_t li = blockIdx.x / 3;
  const int c = blockIdx.x % 3;
  if (li >= n) return;
  const uint32_t CX = c_covered_x[strategy], CY = c_covered_y[strategy];
  const uint32_t R = CY * 8, C = CX * 8, SIZE = R * C;
  const JxlHipVarBlock vb = P.blocks[list[li]];
  const uint32_t g = (vb.by >> 5) * P.xg + (vb.bx >> 5);
  const CoefT* gq = static_cast<const CoefT*>(P.coeffs) + size_t(g) * 3 * 65536 + vb.coef_offset;
  const uint32_t kind = c_strategy_qtable[strategy];
  const float* m = P.dequant + P.dq_offset[kind];
  const uint32_t msize = P.dq_size[kind];
  const float sc = P.inv_global_scale / float(vb.qf);
  const uint32_t tiles_x = (P.xb + 7) / 8;
  const uint32_t tile = (vb.by / 8) * tiles_x + vb.bx / 8;
  const float x_cc = P.base_x + float(P.ytox[tile]) * P.color_scale;
  const float b_cc = P.base_b + float(P.ytob[tile]) * P.color_scale;
  float* s_coef = P.scratch + size_t(blockIdx.x) * 2 * 65536;
  float* s_tmp = s_coef + 65536;
  const uint32_t ord = c_strategy_order[strategy];
  if (c != 1) {  // dequantised Y first (into s_tmp, free until the first transform pass) for the chroma-from-luma term
    StageChannel<CoefT>(P, gq + 65536, list[li], 1, ord, SIZE, CX * CY, m + msize, sc, 0.0f, s_tmp, s_tmp, threadIdx.x, 256);
    __threadfence_block();
    __syncthreads();
  }
  StageChannel<CoefT>(P, gq + size_t(c) * 65536, list[li], c, ord, SIZE, CX * CY, m + size_t(c) * msize,
                      c == 1 ? sc : sc * (c == 0 ? P.x_dm : P.b_dm), c == 0 ? x_cc : b_cc, s_tmp, s_coef, threadIdx.x, 256);
  __threadfence_block();
  __syncthreads();
  {
    const float* dc = P.dc + size_t(c) * P.xb * P.yb + size_t(vb.by) * P.xb + vb.bx;
    const float* bty = P.basis_t + BasisOffset(CY);
    const float* btx = P.basis_t + BasisOffset(CX);
    const uint32_t LC = CX > CY ? CX : CY;
    for (uint32_t i = threadIdx.x; i < CX * CY; i += 256) {
      const uint32_t row = i / LC, col = i % LC;
      uint32_t ky, kx;
      if (CY < CX) { ky = row; kx = col; } else { kx = row; ky = col; }
      float s = 0.0f;
      for (uint32_t y = 0; y < CY; y++) {
        float r = 0.0f;
        for (uint32_t x = 0; x < CX; x++) r += dc[y * P.xb + x] * btx[kx * CX + x];
        s += r * bty[ky * CY + y];
      }
      s *= 1.0f / float(CX * CY);
      s_coef[row * (LC * 8) + col] = s * c_resample[CY - 1 + ky] * c_resample[CX - 1 + kx];
    }
  }
  __threadfence_block();
  __syncthreads();
  const float* btc = P.basis_t + BasisOffset(C);
  const float* btr = P.basis_t + BasisOffset(R);
  for (uint32_t i = threadIdx.x; i < SIZE; i += 256) {
    const uint32_t ky = i / C, x = i % C;
    float s = 0.0f;
    if (R < C) {
      for (uint32_t kx = 0; kx < C; kx++) s += s_coef[ky * C + kx] * btc[kx * C + x];
    } else {
      for (uint32_t kx = 0; kx < C; kx++) s += s_coef[kx * R + ky] * btc[kx * C + x];
    }
    s_tmp[i] = s;
  }
  __threadfence_block();
  __syncthreads();
  float* out = P.out + size_t(c) * P.xp * P.yp + size_t(vb.by) * 8 * P.xp + size_t(vb.bx) * 8;
  for (uint32_t i = threadIdx.x; i < SIZE; i += 256) {
    const uint32_t y = i / C, x = i % C;
    float s = 0.0f;
    for (uint32_t ky = 0; ky < R; ky++) s += s_tmp[ky * C + x] * btr[ky * R + y];
    out[size_t(y) * P.xp + x] = s;
  }
}

// 4-point / 8-point scaled IDCT basis value C_N(n, k).
__device__ __forceinline__ float B4(const float* bt4, int n, int k) { return bt4[k * 4 + n]; }
__device__ __forceinline__ float B8(const float* bt8, int n, int k) { return bt8[k * 8 + n]; }

// 8x8-coverage special transforms; 64 threads per varblock (one per pixel), 4 varblocks per workgroup.
template <typename CoefT>
__global__ __launch_bounds__(256) void k_special(const TransformParams* params, const uint2* desc, uint32_t strategy) {
  JXL_TRANSFORM_PREAMBLE();
  __shared__ float l_all[4][4][64];
  const int sub = threadIdx.x >> 6, t = threadIdx.x & 63;
  float* l_y = l_all[sub][0];  // dequantised Y stays resident for the chroma-from-luma of X and B
  float* l_xb = l_all[sub][1];
  float* buf = l_all[sub][2];
  float* buf2 = l_all[sub][3];
  const uint32_t li = wgd.y + sub;
  const bool active = li < n;
  JxlHipVarBlock vb;
  const CoefT* gq = nullptr;
  const float* m = nullptr;
  uint32_t msize = 0, bidx = 0;
  float sc = 0, x_cc = 0, b_cc = 0;
  if (active) {
    bidx = list[li];
    vb = P.blocks[bidx];
    const uint32_t g = (vb.by >> 5) * P.xg + (vb.bx >> 5);
    gq = static_cast<const CoefT*>(P.coeffs) + size_t(g) * 3 * 65536 + vb.coef_offset;
    const uint32_t kind = c_strategy_qtable[strategy];
    m = P.dequant + P.dq_offset[kind];
    msize = P.dq_size[kind];
    sc = P.inv_global_scale / float(vb.qf);
    const uint32_t tiles_x = (P.xb + 7) / 8;
    const uint32_t tile = (vb.by / 8) * tiles_x + vb.bx / 8;
    x_cc = P.base_x + float(P.ytox[tile]) * P.color_scale;
    b_cc = P.base_b + float(P.ytob[tile]) * P.color_scale;
  }
  const float* bt4 = P.basis_t + BasisOffset(4);
  const float* bt8 = P.basis_t + BasisOffset(8);
  const int py = t >> 3, px = t & 7;
  for (int ci = 0; ci < 3; ci++) {
    const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
    float* co = c == 1 ? l_y : l_xb;
    if (active) {
      StageChannel<CoefT>(P, gq + size_t(c) * 65536, bidx, c, 1, 64, 1, m + size_t(c) * msize,
                          c == 1 ? sc : sc * (c == 0 ? P.x_dm : P.b_dm), c == 0 ? x_cc : b_cc, l_y, co, t, 64);
    }
    __syncthreads();
    // (chroma-subsampled frames, TransformParams::cs: the channel's own grid; a varblock off that grid carries nothing for it,
    // its staged coefficients are all zero then)
    const uint32_t hs = (P.cs >> (2 * c)) & 1u, vs = (P.cs >> (2 * c + 1)) & 1u;
    const bool act_c = active && ((uint32_t(vb.bx) & hs) | (uint32_t(vb.by) & vs)) == 0;
    const uint32_t obx = active ? uint32_t(vb.bx) >> hs : 0u, oby = active ? uint32_t(vb.by) >> vs : 0u;
    if (active && t == 0) co[0] = P.dc[size_t(c) * P.xb * P.yb + size_t(oby) * P.xb + obx];
    __syncthreads();
    float result = 0.0f;
    if (strategy == 1) {  // IDENTITY
      if (active) {
        const int y = py >> 2, x = px >> 2, iy = py & 3, ix = px & 3;
        const float b00 = co[0], b01 = co[1], b10 = co[8], b11 = co[9];
        const float sy = y ? -1.0f : 1.0f, sx = x ? -1.0f : 1.0f;
        // dcs[y*2+x] = b00 + sy*b01 + sx*b10 + sy*sx*b11
        const float block_dc = b00 + sy * b01 + sx * b10 + sx * sy * b11;
        float rs = 0.0f;
        for (int jy = 0; jy < 4; jy++)
          for (int jx = 0; jx < 4; jx++)
            if (jx || jy) rs += co[(y + jy * 2) * 8 + x + jx * 2];
        const float base = block_dc - rs * (1.0f / 16);
        if (iy == 1 && ix == 1) result = base;
        else if (iy == 0 && ix == 0) result = co[(y + 2) * 8 + x + 2] + base;
        else result = co[(y + iy * 2) * 8 + x + ix * 2] + base;
      }
    } else if (strategy == 2) {  // DCT2X2: three 2x2 Hadamard stages
      float* src = co;
      float* dst = buf;
      for (int S = 2; S <= 8; S *= 2) {
        const int h = S / 2;
        float v = 0.0f;
        if (active) {
          v = src[t];
          if (py < S && px < S) {
            const int y = py >> 1, x = px >> 1;
            const float c00 = src[y * 8 + x], c01 = src[y * 8 + h + x], c10 = src[(y + h) * 8 + x], c11 = src[(y + h) * 8 + h + x];
            // output parity (py&1, px&1): (0,0) + + + +, (0,1) + + - -, (1,0) + - + -, (1,1) + - - +
            const float s01 = (py & 1) ? -1.0f : 1.0f, s10 = (px & 1) ? -1.0f : 1.0f;
            v = c00 + s01 * c01 + s10 * c10 + (s01 * s10) * c11;
          }
          dst[t] = v;
        }
        __syncthreads();
        src = dst;
        dst = dst == buf ? buf2 : buf;  // never back into `co`: l_y must survive
      }
      if (active) result = src[t];
    } else if (strategy == 3) {  // DCT4X4
      if (active) {
        const int y = py >> 2, x = px >> 2, iy = py & 3, ix = px & 3;
        const float b00 = co[0], b01 = co[1], b10 = co[8], b11 = co[9];
        const float sy = y ? -1.0f : 1.0f, sx = x ? -1.0f : 1.0f;
        const float dcq = b00 + sy * b01 + sx * b10 + sx * sy * b11;
        float s = 0.0f;
        for (int ky = 0; ky < 4; ky++)
          for (int kx = 0; kx < 4; kx++) {
            // sub-block element [a*4+b] = co[(y + a*2)*8 + x + b*2]; layout of a 4x4 IDCT input is [kx*4+ky]
            const float cf = (kx == 0 && ky == 0) ? dcq : co[(y + kx * 2) * 8 + x + ky * 2];
            s += cf * B4(bt4, iy, ky) * B4(bt4, ix, kx);
          }
        result = s;
      }
    } else if (strategy == 12) {  // DCT4X8: two 4-row x 8-col halves stacked
      if (active) {
        const int y = py >> 2, iy = py & 3;
        const float b0 = co[0], b1 = co[8];
        const float dcq = y ? b0 - b1 : b0 + b1;
        float s = 0.0f;
        for (int ky = 0; ky < 4; ky++)
          for (int kx = 0; kx < 8; kx++) {
            const float cf = (kx == 0 && ky == 0) ? dcq : co[(y + ky * 2) * 8 + kx];
            s += cf * B4(bt4, iy, ky) * B8(bt8, px, kx);
          }
        result = s;
      }
    } else if (strategy == 13) {  // DCT8X4: two 8-row x 4-col halves side by side
      if (active) {
        const int x = px >> 2, ix = px & 3;
        const float b0 = co[0], b1 = co[8];
        const float dcq = x ? b0 - b1 : b0 + b1;
        float s = 0.0f;
        for (int kx = 0; kx < 4; kx++)
          for (int ky = 0; ky < 8; ky++) {
            // sub-block [a*8+b] = co[(x + a*2)*8 + b]; an 8x4 IDCT reads its input as [kx*8+ky]
            const float cf = (kx == 0 && ky == 0) ? dcq : co[(x + kx * 2) * 8 + ky];
            s += cf * B8(bt8, py, ky) * B4(bt4, ix, kx);
          }
        result = s;
      }
    } else {  // AFV0..3
      if (active) {
        const int kind = int(strategy) - 14;
        const int afv_x = kind & 1, afv_y = kind >> 1;
        const float b00 = co[0], b01 = co[1], b10 = co[8];
        const float dcs0 = (b00 + b10 + b01) * 4.0f, dcs1 = b00 + b10 - b01, dcs2 = b00 - b10;
        const int qy = py >> 2, qx = px >> 2, iy = py & 3, ix = px & 3;
        if (qy == afv_y && qx == afv_x) {
          const int by_ = afv_y ? 3 - iy : iy, bx_ = afv_x ? 3 - ix : ix;
          const int pi = by_ * 4 + bx_;
          float s = 0.0f;
          for (int j = 0; j < 16; j++) {
            const float cf = j == 0 ? dcs0 : co[(j >> 2) * 2 * 8 + (j & 3) * 2];
            s += cf * c_afv_basis[j * 16 + pi];
          }
          result = s;
        } else if (qy == afv_y) {  // 4x4 DCT next to the AFV corner
          float s = 0.0f;
          for (int ky = 0; ky < 4; ky++)
            for (int kx = 0; kx < 4; kx++) {
              const float cf = (kx == 0 && ky == 0) ? dcs1 : co[kx * 2 * 8 + ky * 2 + 1];
              s += cf * B4(bt4, iy, ky) * B4(bt4, ix, kx);
            }
          result = s;
        } else {  // 4x8 DCT in the other half
          float s = 0.0f;
          for (int ky = 0; ky < 4; ky++)
            for (int kx = 0; kx < 8; kx++) {
              const float cf = (kx == 0 && ky == 0) ? dcs2 : co[(1 + ky * 2) * 8 + kx];
              s += cf * B4(bt4, iy, ky) * B8(bt8, px, kx);
            }
          result = s;
        }
      }
    }
    if (act_c) ((hs | vs) ? P.cs_out : P.out)[size_t(c) * P.xp * P.yp + (size_t(oby) * 8 + py) * P.xp + size_t(obx) * 8 + px] = result;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- chroma upsampling
// The subsampled channels of a YCbCr frame back at the frame's resolution, in front of the loop filters
// (render_pipeline/stage_chroma_upsampling.cc:29-111; dec_cache.cc:138-150: per channel the horizontal stage, then the
// vertical one). Each stage turns a sample into two: 3/4 of itself + 1/4 of its neighbour on that side, computed as one
// multiply and one fused multiply-add like the reference's Mul / MulAdd, and mirrors about the channel's own image size,
// ceil(xs / 2) columns / ceil(ys / 2) rows (low_memory_render_pipeline.cc:348-355, 668-683). One thread per output pixel;
// the horizontal stage is evaluated for the two input rows the vertical one reads.
struct ChromaUpParams {
  const float* src;  // the channel as the transforms wrote it: the top-left part of a plane with row stride xp
  float* dst;        // the channel's plane of the frame, row stride xp
  uint32_t xs, ys, xp;
  uint32_t y0, y1;   // output rows to produce
  uint32_t hs, vs;   // the channel's shifts (0 or 1, not both 0)
};
__device__ __forceinline__ uint32_t ChromaMirror(int32_t i, uint32_t n) { return i < 0 ? 0u : (uint32_t(i) >= n ? n - 1 : uint32_t(i)); }
__global__ __launch_bounds__(256) void k_chroma_upsample(ChromaUpParams P) {
#pragma clang fp contract(off)
  const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = P.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= P.xs || y >= P.y1) return;
  const uint32_t ws = (P.xs + 1) / 2, hh = (P.ys + 1) / 2;
  auto row_value = [&](uint32_t r) -> float {  // the horizontal stage's output at column x of (subsampled) row r
    const float* in = P.src + size_t(r) * P.xp;
    if (!P.hs) return in[x];
    const uint32_t sx = x >> 1;
    // (the neighbour one step outside the ws samples is the edge sample itself: Mirror(-1) = 0, Mirror(ws) = ws - 1)
    const float nb = in[ChromaMirror((x & 1) ? int32_t(sx) + 1 : int32_t(sx) - 1, ws)];
    return __builtin_fmaf(0.25f, nb, in[sx] * 0.75f);
  };
  float v;
  if (P.vs) {
    const uint32_t sy = y >> 1;
    const float mid = row_value(sy) * 0.75f;
    const float other = row_value(ChromaMirror((y & 1) ? int32_t(sy) + 1 : int32_t(sy) - 1, hh));
    v = __builtin_fmaf(other, 0.25f, mid);
  } else {
    v = row_value(y);
  }
  P.dst[size_t(y) * P.xp + x] = v;
}

// ---------------------------------------------------------------------------------------------- filters + colour
struct FilterParams {
  const float* in;   // 3 planes, stride xp, plane size xp * yp
  float* out;
  uint32_t xs, ys, xp, yp, xb;
  uint32_t y_begin, y_end;  // pixel rows to produce (band decode; whole frame: 0, ys); y_begin is a multiple of 256
  const float* inv_sigma;
  float gab_w[9];  // normalised {w0,w1,w2} per channel
  float ch_scale[3];
  // colour
  float opsin_inv[9], opsin_bias[3], opsin_bias_cbrt[3];
  int32_t linear_output;
  uint8_t* rgb;  // interleaved RGB8 (xs * 3 bytes per row), or NULL
  float* rgbf;   // interleaved RGB f32 (stage_write.cc:334-370 StoreFloatRow), or NULL; honoured by k_filter_rows2 only
};

// Every output format of the boundary (JxlPixelFormat; stage_write.cc:266-286,334-370,548-590): written by k_color_out
// and k_upsample_color. `type` uses the JxlDataType values (0 f32, 2 u8, 3 u16, 5 f16).
struct PixelOut {
  void* dst;           // interleaved, tightly packed rows of samples
  const float* alpha;  // plane of the image size in [0, 1], or NULL (opaque)
  uint32_t xsize, type, nc, bits, swap;  // bits: sample depth of the unsigned types; swap: byte-swapped (big endian) samples
  // Undoing the image's orientation (stage_write.cc:292-306,341-343,441-458,664-699): bit 0 = mirror x, bit 1 = mirror y,
  // bit 2 = transpose (rows of the output are columns of the image: ysize samples long). 0 = rows of xsize pixels.
  // Bit 3: the colour samples of a pixel with alpha are divided by max(alpha, 2^-26) on the way out (JxlDecoderSetUnpremultiplyAlpha
  // on an image whose alpha is associated: stage_write.cc:359-361,460-482, after the transfer function as there).
  uint32_t orient, ysize;
};
// Where pixel (x, y) of the decoded image goes and which dither cell it takes (the reference dithers after the flips).
__device__ __forceinline__ size_t PixelOutIndex(const PixelOut& o, int x, int y, int* dx, int* dy) {
  const int ox = (o.orient & 1) ? int(o.xsize) - 1 - x : x, oy = (o.orient & 2) ? int(o.ysize) - 1 - y : y;
  *dx = ox;
  *dy = oy;
  return (o.orient & 4) ? size_t(ox) * o.ysize + oy : size_t(oy) * o.xsize + ox;
}

__device__ __forceinline__ int MirrorI(int x, int n) {
  while (x < 0 || x >= n) x = x < 0 ? -x - 1 : 2 * n - 1 - x;
  return x;
}
__device__ __forceinline__ float LinearToSrgb(float v) {
  const float a = fabsf(v);
  float r;
  if (a > 0.0031308f) {
    const float s = __builtin_amdgcn_sqrtf(a);  // 1 ulp hardware square root (a > 0.0031308: no denormals)
    float yp = 7.352629620e-01f * s + 1.474205315e+00f;
    yp = yp * s + 3.903842876e-01f;
    yp = yp * s + 5.287254571e-03f;
    yp = yp * s + -5.135152395e-04f;
    float yq = 2.424867759e-02f * s + 9.258482155e-01f;
    yq = yq * s + 1.340816930e+00f;
    yq = yq * s + 3.036675394e-01f;
    yq = yq * s + 1.004519624e-02f;
    r = yp * __builtin_amdgcn_rcpf(yq);  // yq in [0.01, 3.6]: 1 ulp hardware reciprocal
  } else {
    r = a * 12.92f;
  }
  return copysignf(r, v);
}
__device__ __forceinline__ uint8_t ToU8D(float v, float dither) {  // dither = c_dither[(y + c * 13) & 31][(x + c * 23) & 31]
  v = __builtin_amdgcn_fmed3f(v * 255.0f + dither, 0.0f, 255.0f);  // clamp (a NaN becomes 0 as before)
  return uint8_t(__float2int_rn(v));
}
__device__ __forceinline__ uint8_t ToU8(float v, int x, int y, int c) {
  return ToU8D(v, c_dither[((y + c * 13) & 31) * 32 + ((x + c * 23) & 31)]);
}

// One pixel in any output format. Colour images hand out R (and alpha) when fewer than three channels are asked for,
// like the reference (stage_write.cc:334-370: num_color_ = 1); 8-bit samples are dithered with the channel's interleave
// index (stage_write.cc:266-286), wider ones are not.
constexpr float kSmallAlphaOut = 1.0f / float(1u << 26);  // alpha.h:22 kSmallAlpha
__device__ __forceinline__ void StorePixel(const PixelOut& o, int x, int y, float r, float g, float b) {
  const uint32_t nc = o.nc, ncol = nc < 3 ? 1u : 3u;
  float v[4] = {r, g, b, 1.0f};
  if (nc == 2 || nc == 4) {
    const float a = o.alpha ? o.alpha[size_t(y) * o.xsize + x] : 1.0f;
    v[ncol] = a;
    if (o.orient & 8) {
      const float m = 1.0f / fmaxf(kSmallAlphaOut, a);
      for (uint32_t c = 0; c < ncol; c++) v[c] *= m;
    }
  }
  int dx, dy;
  const size_t base = PixelOutIndex(o, x, y, &dx, &dy) * nc;
  for (uint32_t c = 0; c < nc; c++) {
    const float f = v[c];
    if (o.type == 2) {
      const float mul = float((1u << o.bits) - 1u);
      const float t = __builtin_amdgcn_fmed3f(f * mul + c_dither[((dy + int(c) * 13) & 31) * 32 + ((dx + int(c) * 23) & 31)], 0.0f, mul);
      static_cast<uint8_t*>(o.dst)[base + c] = uint8_t(__float2int_rn(t));
    } else if (o.type == 3) {
      const float mul = float((1u << o.bits) - 1u);
      uint32_t u = uint32_t(__float2int_rn(__builtin_amdgcn_fmed3f(f * mul, 0.0f, mul)));
      if (o.swap) u = ((u & 0xFF) << 8) | (u >> 8);
      static_cast<uint16_t*>(o.dst)[base + c] = uint16_t(u);
    } else if (o.type == 5) {
      uint32_t u = __half_as_ushort(__float2half_rn(f));
      if (o.swap) u = ((u & 0xFF) << 8) | (u >> 8);
      static_cast<uint16_t*>(o.dst)[base + c] = uint16_t(u);
    } else {
      uint32_t u = __float_as_uint(f);
      if (o.swap) u = __builtin_bswap32(u);
      static_cast<uint32_t*>(o.dst)[base + c] = u;
    }
  }
}

}  // namespace jxlhip
#endif  // JXL_HIP_KERNELS_H_
