// libjxl_amd — Modular (lossless / MA-tree) decode on gfx950: SURVEY.md §8 row f1, and the machine the VarDCT DC path
// (row a12) shares.
//
// Replaces lib/jxl/modular/encoding/encoding.cc:148-506 (DecodeModularChannelMAANS: properties, MA-tree walk,
// predictors, residual + multiplier), context_predict.h:66-218 (self-correcting weighted predictor), :372-560
// (properties / predictors), dec_ans.h:170-353 (rANS / prefix symbol, hybrid uint, LZ77 with special distances),
// modular/transform/rct.cc:97-147, squeeze.cc:128-330, palette.cc:26-202 (inverse transforms), and the int -> sample
// conversion of the Modular frame path (dec_modular.cc:564-793 + render_pipeline/stage_write.cc).
//
// A Modular stream (one group x a subset of channels) is a pixel-serial adaptive decode: every sample's context comes
// from the MA tree evaluated on already-decoded neighbours. The parallelism is ACROSS streams, exactly as for the AC
// sections of VarDCT: k_modular_streams gives every LANE its own stream (a 4K lossless frame has 135 group streams,
// a batch of frames fills the machine). A wave carries only a few streams (the host picks 1..64 per wave so that all SIMDs
// hold waves): the decode is a chain of dependent latencies, not issue bound, and lanes of one wave wait for each other's
// divergent tree walks. Per sample the chain is: MA-tree walk (16-byte nodes, from LDS when the wave's streams share a
// tree that fits, else global; the properties sit in an LDS array indexed by the node), one alias-table / prefix-table
// lookup in global memory, the predictor. Nothing else on the chain touches memory: the neighbourhood (W, WW, NW, N, NE,
// NEE, NN) and the weighted predictor's error rows are sliding windows in registers fed by loads issued one sample
// ahead, the bit reader keeps its next word loaded ahead, and the code's tables are addressed through a register copy of
// the descriptor.
// The host front-end (csrc/host/jxh_modframe.h) parses every stream's group header (transforms, optional local tree and
// histograms) and hands the kernel flat descriptors; samples are written straight into the frame's channel buffers.
#ifndef JXL_HIP_MODULAR_H_
#define JXL_HIP_MODULAR_H_

#include "jxl_hip_kernels.h"

namespace jxlhip {

struct ModTreeNode {  // 16 bytes (jxlhip_modular_upload packs the boundary's nodes): one load per step of the walk
  int32_t property;   // decision: the property tested (>= 0); leaf: -1 - predictor
  int32_t splitval;   // leaf: offset
  uint32_t lchild;    // leaf: histogram (the context already mapped through the code's context map)
  uint32_t rchild;    // leaf: multiplier
};

struct ModCode {  // one entropy code (global or local to a stream); pointers into device memory
  const uint8_t* ctx_map;   // context -> cluster
  const uint2* alias;       // cluster << log_alpha (ANS)
  const uint32_t* cfg;      // per cluster: split_exp | msb << 8 | lsb << 16
  const uint32_t* prefix_table;
  const uint32_t* prefix_offset;
  uint32_t log_alpha, use_prefix, lz77, lz_min_symbol, lz_min_length, lz_len_cfg, lz_dist_ctx;
  uint32_t table_words;     // 32-bit words of the symbol tables a wave may stage in LDS: the alias table (ANS) or the prefix
                            // table, then one word per cluster (cfg), then for prefix codes one more per cluster (offsets)
  uint32_t num_clusters, pad[3];
};

struct ModChannel {  // one channel of a stream: a rectangle of a frame channel buffer
  int32_t* data;     // top-left sample of the rectangle
  uint32_t stride;   // samples per row of the underlying buffer
  uint32_t w, h;
  uint32_t sig;      // channels of a stream with equal `sig` have identical size and shifts ("previous channel" properties)
};

struct ModStream {
  const uint32_t* words;   // section bytes (4-byte aligned start)
  uint32_t bit_offset;     // where the stream's sample data begins
  uint32_t size_bytes;     // section size: reads beyond it are zero, consuming beyond it is an error
  const ModTreeNode* tree;
  uint32_t tree_nodes;
  const ModCode* code;
  const ModChannel* channels;
  uint32_t num_channels;
  uint32_t stream_id;      // MA property 1
  uint32_t first_channel_index;  // MA property 0 of channels[0] (the following count up)
  int32_t wp[11];          // weighted predictor header: p1C p2C p3Ca..p3Ce, w0..w3
  uint32_t uses_wp;        // the tree tests property 15 or predicts with predictor 6
  uint32_t num_props;      // 16 + 4 * referenced previous channels
  uint32_t dist_multiplier;
  int32_t* wp_scratch;     // 5 arrays of 2 * (max width + 2) ints (uses_wp only)
  uint32_t* lz_window;     // lz_window_mask + 1 entries (code->lz77 only)
  uint32_t lz_window_mask;
  uint32_t* status;        // out: 0 ok, bit 0 = ANS final state, bit 1 = over-read, bit 2 = LZ77 length overflow
  uint32_t* end_bit;       // out: bit position after the last sample
};

__constant__ int8_t c_special_distances[120][2] = {
    {0, 1},  {1, 0},  {1, 1},  {-1, 1}, {0, 2},  {2, 0},  {1, 2},  {-1, 2}, {2, 1},  {-2, 1}, {2, 2},  {-2, 2},
    {0, 3},  {3, 0},  {1, 3},  {-1, 3}, {3, 1},  {-3, 1}, {2, 3},  {-2, 3}, {3, 2},  {-3, 2}, {0, 4},  {4, 0},
    {1, 4},  {-1, 4}, {4, 1},  {-4, 1}, {3, 3},  {-3, 3}, {2, 4},  {-2, 4}, {4, 2},  {-4, 2}, {0, 5},  {3, 4},
    {-3, 4}, {4, 3},  {-4, 3}, {5, 0},  {1, 5},  {-1, 5}, {5, 1},  {-5, 1}, {2, 5},  {-2, 5}, {5, 2},  {-5, 2},
    {4, 4},  {-4, 4}, {3, 5},  {-3, 5}, {5, 3},  {-5, 3}, {0, 6},  {6, 0},  {1, 6},  {-1, 6}, {6, 1},  {-6, 1},
    {2, 6},  {-2, 6}, {6, 2},  {-6, 2}, {4, 5},  {-4, 5}, {5, 4},  {-5, 4}, {3, 6},  {-3, 6}, {6, 3},  {-6, 3},
    {0, 7},  {7, 0},  {1, 7},  {-1, 7}, {5, 5},  {-5, 5}, {7, 1},  {-7, 1}, {4, 6},  {-4, 6}, {6, 4},  {-6, 4},
    {2, 7},  {-2, 7}, {7, 2},  {-7, 2}, {3, 7},  {-3, 7}, {7, 3},  {-7, 3}, {5, 6},  {-5, 6}, {6, 5},  {-6, 5},
    {8, 0},  {4, 7},  {-4, 7}, {7, 4},  {-7, 4}, {8, 1},  {8, 2},  {6, 6},  {-6, 6}, {8, 3},  {5, 7},  {-5, 7},
    {7, 5},  {-7, 5}, {8, 4},  {6, 7},  {-6, 7}, {7, 6},  {-7, 6}, {8, 5},  {7, 7},  {-7, 7}, {8, 6},  {8, 7}};

// Four consecutive samples of a channel row in one 16-byte access (rows are only 4-byte aligned). A lane's accesses go to
// cache lines of its own, and the memory pipeline takes such a wave-instruction one lane at a time: the stream kernel is
// bound by the NUMBER of its global accesses per sample (~70 G lane-accesses/s for the whole device), so samples move
// four at a time.
typedef int32_t ModI4 __attribute__((ext_vector_type(4), aligned(4)));
// Samples [pos, pos + 4) of `row`, zeros outside [0, w) or when the row does not exist.
__device__ __forceinline__ ModI4 ModLoadChunk(const int32_t* row, uint32_t pos, uint32_t w, bool row_exists) {
  ModI4 v = {0, 0, 0, 0};
  if (!row_exists || pos >= w) return v;
  if (pos + 3 < w) return *reinterpret_cast<const ModI4*>(row + pos);
  v.x = row[pos];
  if (pos + 1 < w) v.y = row[pos + 1];
  if (pos + 2 < w) v.z = row[pos + 2];
  return v;
}
__device__ __forceinline__ int32_t ModPick(const ModI4& v, uint32_t k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }

// Bit reader with the next word loaded ahead (the refill that consumes it issues the following load).
struct ModBits {
  const uint32_t* p;
  uint32_t idx, nwords, next;
  uint64_t buf;
  int bits;
};
__device__ __forceinline__ void MbInit(ModBits& b, const uint32_t* words, uint32_t nwords, uint32_t first_word) {
  b.p = words;
  b.nwords = nwords;
  b.idx = first_word;
  b.buf = 0;
  b.bits = 0;
  b.next = b.idx < b.nwords ? b.p[b.idx] : 0u;
}
__device__ __forceinline__ void MbRefill(ModBits& b) {
  if (b.bits < 32) {
    b.buf |= uint64_t(b.next) << b.bits;
    b.bits += 32;
    b.idx++;
    b.next = b.idx < b.nwords ? b.p[b.idx] : 0u;
  }
}
__device__ __forceinline__ uint32_t MbRead(ModBits& b, uint32_t n) {  // n <= 32, caller refilled
  const uint32_t v = uint32_t(b.buf & ((uint64_t(1) << n) - 1));
  b.buf >>= n;
  b.bits -= int(n);
  return v;
}

struct ModReader {
  ModBits br;
  uint32_t state;
  ModCode T;  // register copy of the stream's code descriptor
  const uint32_t* ltab;  // the code's tables in LDS (alias or prefix table | cfg | prefix offsets), or NULL
  uint32_t* window;
  uint32_t mask, num_decoded, num_to_copy, copy_pos, num_special, dist_mult, err;
  ModI4 wlast;  // the last four window values (LZ77): the window itself is written four values at a time
};
// LZ77 window (dec_ans.h:288-353): value `v` becomes entry num_decoded; entries reach memory as aligned 16-byte chunks.
__device__ __forceinline__ void ModWindowPush(ModReader& r, uint32_t v) {
  r.wlast.x = r.wlast.y;
  r.wlast.y = r.wlast.z;
  r.wlast.z = r.wlast.w;
  r.wlast.w = int32_t(v);
  r.num_decoded++;
  if ((r.num_decoded & 3u) == 0) *reinterpret_cast<ModI4*>(r.window + ((r.num_decoded - 4) & r.mask)) = r.wlast;
}
__device__ __forceinline__ uint32_t ModWindowGet(const ModReader& r, uint32_t pos) {  // pos < num_decoded
  if (pos + 4 >= r.num_decoded) return uint32_t(ModPick(r.wlast, pos + 4 - r.num_decoded));
  return r.window[pos & r.mask];
}
__device__ __forceinline__ uint32_t ModSymbol(ModReader& r, uint32_t cluster) {
  const ModCode& T = r.T;
  MbRefill(r.br);
  if (T.use_prefix) {
    const uint32_t po = r.ltab ? r.ltab[T.table_words - T.num_clusters + cluster] : T.prefix_offset[cluster];
    const uint32_t e = PrefixLookup((r.ltab ? r.ltab : T.prefix_table) + (po & 0xFFFFFFu), po >> 24, uint32_t(r.br.buf));
    MbRead(r.br, e & 0xFF);
    return e >> 8;
  }
  const uint32_t log_entry = 12 - T.log_alpha;
  const uint32_t res = r.state & 0xFFFu, i = res >> log_entry, pos = res & ((1u << log_entry) - 1);
  const uint32_t slot = (cluster << T.log_alpha) + i;
  const uint2 e = r.ltab ? *reinterpret_cast<const uint2*>(r.ltab + 2 * slot) : T.alias[slot];
  const uint32_t cutoff = e.x & 0xFF, right = (e.x >> 8) & 0xFF, freq0 = e.x >> 16, offsets1 = e.y & 0xFFFF, freq1 = e.y >> 16;
  const bool greater = pos >= cutoff;
  r.state = (greater ? freq1 : freq0) * (r.state >> 12) + (greater ? offsets1 : 0u) + pos;
  if (r.state < (1u << 16)) r.state = (r.state << 16) | MbRead(r.br, 16);
  return greater ? right : i;
}
__device__ __forceinline__ uint32_t ModUint(ModReader& r, uint32_t cfg, uint32_t token) {
  const uint32_t split_exp = cfg & 0xFF, msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
  if (token < (1u << split_exp)) return token;
  const uint32_t nbits = (split_exp - (msb + lsb) + ((token - (1u << split_exp)) >> (msb + lsb))) & 31u;
  const uint32_t low = token & ((1u << lsb) - 1), hi = token >> lsb;
  MbRefill(r.br);
  const uint32_t bits = MbRead(r.br, nbits);
  return (((((1u << msb) | (hi & ((1u << msb) - 1))) << nbits) | bits) << lsb) | low;
}
// dec_ans.h:288-353 (ReadHybridUintClusteredInlined with LZ77)
__device__ __forceinline__ uint32_t ModRead(ModReader& r, uint32_t cluster) {
  const ModCode& T = r.T;
  if (T.lz77 && r.num_to_copy > 0) {
    // (copy_pos == num_decoded only for a copy at the very start of a stream, distance 0: zeros, dec_ans.h:320-327)
    const uint32_t v = r.copy_pos >= r.num_decoded ? 0u : ModWindowGet(r, r.copy_pos);
    r.copy_pos++;
    r.num_to_copy--;
    ModWindowPush(r, v);
    return v;
  }
  const uint32_t cfg_at = T.use_prefix ? T.table_words - 2 * T.num_clusters : T.table_words - T.num_clusters;
  const uint32_t cfg = r.ltab ? r.ltab[cfg_at + cluster] : T.cfg[cluster];  // (issued with the symbol's table lookup: one round trip for both)
  const uint32_t token = ModSymbol(r, cluster);
  if (T.lz77 && token >= T.lz_min_symbol) {
    r.num_to_copy = ModUint(r, T.lz_len_cfg, token - T.lz_min_symbol) + T.lz_min_length;
    const uint32_t dtok = ModSymbol(r, T.lz_dist_ctx);
    uint32_t distance = ModUint(r, r.ltab ? r.ltab[cfg_at + T.lz_dist_ctx] : T.cfg[T.lz_dist_ctx], dtok);
    if (distance < r.num_special) {
      const int d = int(c_special_distances[distance][0]) + int(r.dist_mult) * int(c_special_distances[distance][1]);
      distance = d > 1 ? uint32_t(d) : 1u;
    } else {
      distance = distance + 1 - r.num_special;
    }
    if (distance > r.num_decoded) distance = r.num_decoded;
    if (distance > r.mask + 1) distance = r.mask + 1;
    r.copy_pos = r.num_decoded - distance;
    if (r.num_to_copy < T.lz_min_length) {  // wrapped
      r.err |= 4;
      r.num_to_copy = 0;
      return 0;
    }
    // (distance 0 only at the very start of a stream: the reference zero-fills the window then)
    const uint32_t v = distance == 0 ? 0u : ModWindowGet(r, r.copy_pos);
    r.copy_pos++;
    r.num_to_copy--;
    ModWindowPush(r, v);
    return v;
  }
  const uint32_t v = ModUint(r, cfg, token);
  if (T.lz77) ModWindowPush(r, v);
  return v;
}

__device__ __forceinline__ int32_t ModClampedGradient(int32_t n, int32_t w, int32_t l) {
  const int32_t m = n < w ? n : w, M = n < w ? w : n;
  const int32_t grad = int32_t(uint32_t(n) + uint32_t(w) - uint32_t(l));
  const int32_t gc = (l < m) ? M : grad;
  return (l > M) ? m : gc;
}
__device__ __forceinline__ int64_t ModAbs64(int64_t v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int ModFloorLog2(uint64_t v) { return 63 - __clzll(static_cast<long long>(v)); }

// Self-correcting weighted predictor (context_predict.h:66-218). The reference keeps five arrays of two rows (the true
// errors and the four sub-predictors' errors, where an error is also added to the entry above and to the right); here a
// row entry is {sub-predictor errors 0..3, true error, pad x 3} (32 bytes, two rows of width + 2 entries in the stream's
// scratch), and everything a sample reads lives in registers: the entries above (N, NW, NE) slide along the row, the
// "+= above right" lands in the register that becomes N of the next sample (it is never read from memory again: the row
// above is overwritten two rows later), the entry two to the right is loaded one sample ahead, and a sample's own entry
// is a fire-and-forget store for the row below.
struct ModWp {
  uint32_t peN[4], peNW[4], peNE[4];
  int32_t teW, teN, teNW, teNE;
  int64_t prediction[4];
  int64_t pred;
};
constexpr uint32_t kModWpEntry = 8;  // ints per row entry
// `ldiv` = (1 << 24) / (i + 1) for i in 0..63 in LDS (the reference's divlookup, context_predict.h:79-90): both
// divisions of the predictor have divisors of at most 64, and an integer division is ~30 vector instructions here.
__device__ __forceinline__ uint32_t ModErrorWeight(uint32_t x, uint32_t maxweight, const uint32_t* ldiv) {
  int shift = 58 - int(__clzll(static_cast<long long>(uint64_t(x) + 1)));  // floor(log2(x + 1)) - 5, x any 32-bit value
  if (shift < 0) shift = 0;
  return uint32_t(4 + ((uint64_t(maxweight) * ldiv[x >> shift]) >> shift));  // (x >> shift is at most 63)
}
__device__ __forceinline__ int64_t ModWpPredict(ModWp& s, const int32_t* hd, int64_t N, int64_t W, int64_t NE, int64_t NW, int64_t NN,
                                               int32_t* max_err_prop, const uint32_t* ldiv) {
  uint32_t weights[4];
#pragma unroll
  for (int i = 0; i < 4; i++) weights[i] = ModErrorWeight(s.peN[i] + s.peNE[i] + s.peNW[i], uint32_t(hd[7 + i]), ldiv);
  N *= 8; W *= 8; NE *= 8; NW *= 8; NN *= 8;
  const int64_t teW = s.teW, teN = s.teN, teNW = s.teNW, teNE = s.teNE;
  const int64_t sumWN = teN + teW;
  {
    int64_t p = teW;
    if (ModAbs64(teN) > ModAbs64(p)) p = teN;
    if (ModAbs64(teNW) > ModAbs64(p)) p = teNW;
    if (ModAbs64(teNE) > ModAbs64(p)) p = teNE;
    *max_err_prop = int32_t(p);
  }
  s.prediction[0] = W + NE - N;
  s.prediction[1] = N - (((sumWN + teNE) * hd[0]) >> 5);
  s.prediction[2] = W - (((sumWN + teNW) * hd[1]) >> 5);
  s.prediction[3] = N - ((teNW * hd[2] + teN * hd[3] + teNE * hd[4] + (NN - N) * hd[5] + (NW - W) * hd[6]) >> 5);
  {  // WeightedAverage
    uint32_t ws = weights[0] + weights[1] + weights[2] + weights[3];
    const int lw = ModFloorLog2(ws);
    ws = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      weights[i] >>= lw - 4;
      ws += weights[i];
    }
    int64_t sum = (ws >> 1) - 1;
#pragma unroll
    for (int i = 0; i < 4; i++) sum += s.prediction[i] * int64_t(weights[i]);
    s.pred = (sum * int64_t(ldiv[(ws - 1) & 63])) >> 24;  // (ws is in 13..31)
  }
  if (((teN ^ teW) | (teN ^ teNW)) > 0) return (s.pred + 3) >> 3;
  const int64_t mx = W > NE ? (W > N ? W : N) : (NE > N ? NE : N), mn = W < NE ? (W < N ? W : N) : (NE < N ? NE : N);
  s.pred = s.pred < mn ? mn : (s.pred > mx ? mx : s.pred);
  return (s.pred + 3) >> 3;
}
// After sample x of a row `w` wide got value `val`: stores the sample's entry into `cur_entry` (row y, for row y + 1) and
// slides the window to sample x + 1; `ahead` is the entry of the row above at x + 2 (zeros in the first row).
__device__ __forceinline__ void ModWpUpdate(ModWp& s, int64_t val, int32_t* cur_entry, const int4& ahead_pe, int32_t ahead_te, bool ahead_valid) {
  val *= 8;
  const int32_t te = int32_t(s.pred - val);
  uint32_t err[4];
#pragma unroll
  for (int i = 0; i < 4; i++) err[i] = uint32_t((ModAbs64(s.prediction[i] - val) + 3) >> 3);
  *reinterpret_cast<int4*>(cur_entry) = make_int4(int(err[0]), int(err[1]), int(err[2]), int(err[3]));
  cur_entry[4] = te;
  const uint32_t ahead[4] = {uint32_t(ahead_pe.x), uint32_t(ahead_pe.y), uint32_t(ahead_pe.z), uint32_t(ahead_pe.w)};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    s.peNW[i] = s.peN[i];
    s.peN[i] = s.peNE[i] + err[i];  // the reference's "+= above right"
    s.peNE[i] = ahead_valid ? ahead[i] : s.peN[i];  // (last sample of a row: its NE is its N)
  }
  s.teNW = s.teN;
  s.teN = s.teNE;
  s.teNE = ahead_valid ? ahead_te : s.teN;
  s.teW = te;
}

__device__ __forceinline__ int64_t ModPredict(uint32_t p, int64_t left, int64_t top, int64_t toptop, int64_t topleft, int64_t topright,
                                             int64_t leftleft, int64_t toprightright, int64_t wp) {
  switch (p) {
    case 1: return left;
    case 2: return top;
    case 3: return (left + top) / 2;
    case 4: {
      const int64_t pp = left + top - topleft;
      return ModAbs64(pp - left) < ModAbs64(pp - top) ? left : top;
    }
    case 5: return ModClampedGradient(int32_t(left), int32_t(top), int32_t(topleft));
    case 6: return wp;
    case 7: return topright;
    case 8: return topleft;
    case 9: return leftleft;
    case 10: return (left + topleft) / 2;
    case 11: return (topleft + top) / 2;
    case 12: return (top + topright) / 2;
    case 13: return (6 * top - 2 * toptop + 7 * left + leftleft + toprightright + 3 * topright + 8) / 16;
    default: return 0;
  }
}

constexpr int kModMaxProps = 16 + 4 * 4;  // property 15 + up to four referenced previous channels (host refuses trees beyond)
constexpr uint32_t kModTreeLdsNodes = 2048;  // largest tree kept in LDS (32 KB)

constexpr uint32_t kModTableLdsWords = 8192;  // largest symbol-table set kept in LDS (32 KB)

// Dynamic LDS of a workgroup (one wave, `lanes` streams): the properties [property][lane], then `tree_cap` tree nodes,
// then `table_cap` words of symbol tables.
__host__ __device__ inline uint32_t ModLdsBytes(uint32_t lanes, uint32_t tree_cap, uint32_t table_cap) {
  return kModMaxProps * lanes * 4 + tree_cap * 16 + table_cap * 4 + 64 * 4;  // (+ the weighted predictor's division table)
}

// One lane per stream, `lanes` (a power of two, <= 64) streams per workgroup; `streams` holds `n` descriptors.
// WP / REFS = false compile the weighted predictor / the previous-channel properties out (the host picks the form no
// stream of the launch needs more than): half the registers, twice the waves per SIMD to hide the chain's latencies.
template <bool WP, bool REFS>
__global__ __launch_bounds__(64) void k_modular_streams(const ModStream* streams, uint32_t n, uint32_t lanes, uint32_t tree_cap, uint32_t table_cap) {
  extern __shared__ __align__(16) uint8_t mod_lds[];
  const uint32_t lane = threadIdx.x;
  const uint32_t si = blockIdx.x * lanes + lane;
  const bool active = lane < lanes && si < n;
  int32_t* const lprops = reinterpret_cast<int32_t*>(mod_lds) + lane;  // property q at lprops[q * lanes]
  int4* const ltree = reinterpret_cast<int4*>(mod_lds + kModMaxProps * lanes * 4);
  ModStream S;
  if (active) S = streams[si];
  else memset(&S, 0, sizeof(S));
  // ---- the tree goes to LDS when every stream of the wave walks the same one and it fits
  const uint64_t tree_bits = reinterpret_cast<uint64_t>(S.tree);
  const uint32_t first_lo = __builtin_amdgcn_readfirstlane(uint32_t(tree_bits)), first_hi = __builtin_amdgcn_readfirstlane(uint32_t(tree_bits >> 32));
  const uint32_t first_nodes = __builtin_amdgcn_readfirstlane(S.tree_nodes);
  const bool same = !active || (uint32_t(tree_bits) == first_lo && uint32_t(tree_bits >> 32) == first_hi);
  const bool tree_lds = __builtin_amdgcn_ballot_w64(same) == ~0ull && first_nodes <= tree_cap && first_nodes != 0;
  if (tree_lds) {
    const int4* src = reinterpret_cast<const int4*>((uint64_t(first_hi) << 32) | first_lo);
    for (uint32_t i = lane; i < first_nodes; i += 64) ltree[i] = src[i];
  }
  // ---- likewise the symbol tables of the streams' entropy code
  ModReader r;
  if (active) r.T = *S.code;
  else memset(&r.T, 0, sizeof(r.T));
  uint32_t* const ltab = reinterpret_cast<uint32_t*>(mod_lds + kModMaxProps * lanes * 4 + tree_cap * 16);
  uint32_t* const ldiv = ltab + table_cap;
  if (WP) ldiv[lane] = (1u << 24) / (lane + 1);
  {
    const uint64_t code_bits = reinterpret_cast<uint64_t>(S.code);
    const uint32_t c_lo = __builtin_amdgcn_readfirstlane(uint32_t(code_bits)), c_hi = __builtin_amdgcn_readfirstlane(uint32_t(code_bits >> 32));
    const uint32_t words = __builtin_amdgcn_readfirstlane(r.T.table_words);
    const bool same_code = !active || (uint32_t(code_bits) == c_lo && uint32_t(code_bits >> 32) == c_hi);
    const bool code_lds = __builtin_amdgcn_ballot_w64(same_code) == ~0ull && words <= table_cap && words != 0;
    r.ltab = code_lds ? ltab : nullptr;
    if (code_lds) {
      // (lane 0 is active: its register copy of the descriptor names the tables)
      const uint64_t t0 = reinterpret_cast<uint64_t>(r.T.use_prefix ? static_cast<const void*>(r.T.prefix_table) : static_cast<const void*>(r.T.alias));
      const uint64_t t1 = reinterpret_cast<uint64_t>(r.T.cfg), t2 = reinterpret_cast<uint64_t>(r.T.prefix_offset);
      // (the builtin returns int: both halves go through uint32_t, or a low half with its top bit set sign-extends)
      auto uniform = [](uint64_t v) {
        const uint32_t lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v)))), hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v >> 32))));
        return (uint64_t(hi) << 32) | uint64_t(lo);
      };
      const uint32_t* src0 = reinterpret_cast<const uint32_t*>(uniform(t0));
      const uint32_t* src1 = reinterpret_cast<const uint32_t*>(uniform(t1));
      const uint32_t* src2 = reinterpret_cast<const uint32_t*>(uniform(t2));
      const uint32_t ncl = __builtin_amdgcn_readfirstlane(r.T.num_clusters), pfx = __builtin_amdgcn_readfirstlane(r.T.use_prefix);
      const uint32_t main_words = words - ncl * (pfx ? 2 : 1);
      for (uint32_t i = lane; i < main_words; i += 64) ltab[i] = src0[i];
      for (uint32_t i = lane; i < ncl; i += 64) ltab[main_words + i] = src1[i];
      if (pfx)
        for (uint32_t i = lane; i < ncl; i += 64) ltab[main_words + ncl + i] = src2[i];
    }
  }
  __syncthreads();
  if (!active) return;
  r.window = S.lz_window;
  r.mask = S.lz_window_mask;
  r.num_decoded = r.num_to_copy = r.copy_pos = 0;
  r.wlast = ModI4{0, 0, 0, 0};
  r.dist_mult = S.dist_multiplier;
  r.num_special = S.dist_multiplier ? 120u : 0u;
  r.err = 0;
  MbInit(r.br, S.words, (S.size_bytes + 3) / 4, S.bit_offset >> 5);
  MbRefill(r.br);
  if (S.bit_offset & 31) MbRead(r.br, S.bit_offset & 31);
  r.state = 0x13u << 16;
  if (!r.T.use_prefix) {
    MbRefill(r.br);
    r.state = MbRead(r.br, 16);
    MbRefill(r.br);
    r.state |= MbRead(r.br, 16) << 16;
  }
  const int4* const gtree = reinterpret_cast<const int4*>(S.tree);
  auto node = [&](uint32_t pos) -> int4 { return tree_lds ? ltree[pos] : gtree[pos]; };
#pragma unroll
  for (int i = 0; i < kModMaxProps; i++) lprops[i * lanes] = 0;
  const bool wp_on = WP && S.uses_wp != 0;
  for (uint32_t ci = 0; ci < S.num_channels && !r.err; ci++) {
    const ModChannel ch = S.channels[ci];
    if (!ch.w || !ch.h) continue;
    const uint32_t w = ch.w;
    // previous channels of the same shape, nearest first (context_predict.h:419-451)
    const int32_t* ref_data[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t ref_stride[4] = {0, 0, 0, 0};
    uint32_t nrefs = 0;
    for (int j = int(ci) - 1; REFS && j >= 0 && nrefs * 4 < S.num_props - 16 && nrefs < 4; j--) {
      const ModChannel o = S.channels[j];
      if (o.sig == ch.sig && o.w == ch.w && o.h == ch.h) {
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (uint32_t(q) == nrefs) {
            ref_data[q] = o.data;
            ref_stride[q] = o.stride;
          }
        nrefs++;
      }
    }
    const ptrdiff_t stride = ptrdiff_t(ch.stride);
#pragma unroll
    for (int i = 16; i < kModMaxProps; i++) lprops[i * lanes] = 0;  // slots of previous channels this channel does not have
    lprops[0] = int32_t(S.first_channel_index + ci);
    lprops[1 * lanes] = int32_t(S.stream_id);
    // properties 0 and 1 are constant over the channel: their decisions at the top of the tree are taken once
    uint32_t root = 0;
    for (;;) {
      const int4 nd = node(root);
      if (nd.x != 0 && nd.x != 1) break;
      const int32_t v = nd.x == 0 ? int32_t(S.first_channel_index + ci) : int32_t(S.stream_id);
      root = v > nd.y ? uint32_t(nd.z) : uint32_t(nd.w);
    }
    ModWp wp;
    for (uint32_t y = 0; y < ch.h && !r.err; y++) {
      int32_t* const p = ch.data + size_t(y) * ch.stride;
      const int32_t* const pN = p - stride;       // read only when y > 0
      const int32_t* const pNN = p - 2 * stride;  // read only when y > 1
      lprops[2 * lanes] = int32_t(y);
      int32_t prev9 = 0;
      // the row above at x, x - 1, x + 1, x + 2 and two rows above at x: a sliding window (raw samples; the edge rules
      // are applied where the neighbours are formed)
      // (fed by 16-byte chunks: cN / cNN hold the chunk the next look-ahead values come from, nN / nNN the one after it,
      // loaded four samples ahead of its first use)
      ModI4 cN = ModLoadChunk(pN, 0, w, y > 0), nN = ModLoadChunk(pN, 4, w, y > 0);
      ModI4 cNN = ModLoadChunk(pNN, 0, w, y > 1), nNN = ModLoadChunk(pNN, 4, w, y > 1);
      int32_t rN = cN.x, rNW = 0, rNE = cN.y, rNEE = cN.z;
      int32_t rNN = cNN.x;
      int32_t rW = 0, rWW = 0;
      ModI4 outv = {0, 0, 0, 0};  // the last samples of this row, stored four at a time
      // previous channels: this row and the one above at x and x - 1
      int32_t rv[4] = {0, 0, 0, 0}, rvl[4] = {0, 0, 0, 0}, rvt[4] = {0, 0, 0, 0}, rvtl[4] = {0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (REFS && uint32_t(j) < nrefs) {
          rv[j] = ref_data[j][size_t(y) * ref_stride[j]];
          rvt[j] = y ? ref_data[j][size_t(y - 1) * ref_stride[j]] : 0;
        }
      int32_t* wcur = nullptr;
      const int32_t* wprev = nullptr;
      if (wp_on) {
        wcur = S.wp_scratch + size_t((y & 1) ? 0 : (w + 2)) * kModWpEntry;
        wprev = S.wp_scratch + size_t((y & 1) ? (w + 2) : 0) * kModWpEntry;
        int4 n0 = make_int4(0, 0, 0, 0), n1 = make_int4(0, 0, 0, 0);
        int32_t t0 = 0, t1 = 0;
        if (y) {
          n0 = *reinterpret_cast<const int4*>(wprev);
          t0 = wprev[4];
          if (w > 1) {
            n1 = *reinterpret_cast<const int4*>(wprev + kModWpEntry);
            t1 = wprev[kModWpEntry + 4];
          } else {
            n1 = n0;
            t1 = t0;
          }
        }
        wp.peN[0] = wp.peNW[0] = uint32_t(n0.x); wp.peN[1] = wp.peNW[1] = uint32_t(n0.y);
        wp.peN[2] = wp.peNW[2] = uint32_t(n0.z); wp.peN[3] = wp.peNW[3] = uint32_t(n0.w);
        wp.peNE[0] = uint32_t(n1.x); wp.peNE[1] = uint32_t(n1.y); wp.peNE[2] = uint32_t(n1.z); wp.peNE[3] = uint32_t(n1.w);
        wp.teN = wp.teNW = t0;
        wp.teNE = t1;
        wp.teW = 0;
      }
      for (uint32_t x = 0; x < w; x++) {
        // ---- loads for the next sample first (nothing below waits for them)
        // pN[x + 3] and pNN[x + 1] out of the chunks (zeros beyond the row: the edge rules never use them)
        if (((x + 3) & 3) == 0) {
          cN = nN;
          nN = ModLoadChunk(pN, x + 7, w, y > 0);
        }
        if (((x + 1) & 3) == 0) {
          cNN = nNN;
          nNN = ModLoadChunk(pNN, x + 5, w, y > 1);
        }
        const int32_t aNEE = ModPick(cN, (x + 3) & 3);
        const int32_t aNN = ModPick(cNN, (x + 1) & 3);
        int4 a_pe = make_int4(0, 0, 0, 0);
        int32_t a_te = 0;
        const bool a_valid = x + 2 < w;
        if (wp_on && y && a_valid) {
          a_pe = *reinterpret_cast<const int4*>(wprev + size_t(x + 2) * kModWpEntry);
          a_te = wprev[size_t(x + 2) * kModWpEntry + 4];
        }
        int32_t av[4] = {0, 0, 0, 0}, avt[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (REFS && uint32_t(j) < nrefs && x + 1 < w) {
            av[j] = ref_data[j][size_t(y) * ref_stride[j] + x + 1];
            avt[j] = y ? ref_data[j][size_t(y - 1) * ref_stride[j] + x + 1] : 0;
          }
        // ---- neighbours (encoding.cc / context_predict.h edge rules)
        const int64_t left = x ? rW : (y ? rN : 0);
        const int64_t top = y ? rN : left;
        const int64_t topleft = (x && y) ? rNW : left;
        const int64_t topright = (x + 1 < w && y) ? rNE : top;
        const int64_t leftleft = x > 1 ? rWW : left;
        const int64_t toptop = y > 1 ? rNN : top;
        const int64_t toprightright = (x + 2 < w && y) ? rNEE : topright;
        const int32_t p9 = int32_t(left + top - topleft);
        lprops[3 * lanes] = int32_t(x);
        lprops[4 * lanes] = int32_t(top > 0 ? top : -top);
        lprops[5 * lanes] = int32_t(left > 0 ? left : -left);
        lprops[6 * lanes] = int32_t(top);
        lprops[7 * lanes] = int32_t(left);
        lprops[8 * lanes] = int32_t(left - prev9);  // uses the previous pixel's property 9
        lprops[9 * lanes] = p9;
        prev9 = p9;
        lprops[10 * lanes] = int32_t(left - topleft);
        lprops[11 * lanes] = int32_t(topleft - top);
        lprops[12 * lanes] = int32_t(top - topright);
        lprops[13 * lanes] = int32_t(top - toptop);
        lprops[14 * lanes] = int32_t(left - leftleft);
        int64_t wp_pred = 0;
        if (wp_on) {
          int32_t max_err = 0;
          wp_pred = ModWpPredict(wp, S.wp, top, left, topright, topleft, toptop, &max_err, ldiv);
          lprops[15 * lanes] = max_err;
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (REFS && uint32_t(j) < nrefs) {
            const int64_t v = rv[j];
            const int64_t vl = x ? rvl[j] : 0;
            const int64_t vt = y ? rvt[j] : vl;
            const int64_t vtl = (x && y) ? rvtl[j] : vl;
            const int64_t vp = ModClampedGradient(int32_t(vl), int32_t(vt), int32_t(vtl));
            lprops[(16 + 4 * j) * lanes] = int32_t(ModAbs64(v));
            lprops[(17 + 4 * j) * lanes] = int32_t(v);
            lprops[(18 + 4 * j) * lanes] = int32_t(ModAbs64(v - vp));
            lprops[(19 + 4 * j) * lanes] = int32_t(v - vp);
          }
        // ---- MA tree walk
        int4 nd = node(root);
        while (nd.x >= 0) {
          const uint32_t pos = lprops[uint32_t(nd.x) * lanes] > nd.y ? uint32_t(nd.z) : uint32_t(nd.w);
          nd = node(pos);
        }
        const int64_t guess = int64_t(nd.y) + ModPredict(uint32_t(-1 - nd.x), left, top, toptop, topleft, topright, leftleft, toprightright, wp_pred);
        const uint32_t v = ModRead(r, uint32_t(nd.z));
        const int64_t val = int64_t(int32_t((v >> 1) ^ (0u - (v & 1)))) * int64_t(uint32_t(nd.w)) + guess;
        const int32_t out = int32_t(val);
        outv.x = outv.y;
        outv.y = outv.z;
        outv.z = outv.w;
        outv.w = out;
        if ((x & 3) == 3) {
          *reinterpret_cast<ModI4*>(p + (x - 3)) = outv;
        } else if (x + 1 == w) {  // the row's last one to three samples
          const uint32_t part = (x & 3) + 1;
          if (part == 3) p[x - 2] = outv.y;
          if (part >= 2) p[x - 1] = outv.z;
          p[x] = outv.w;
        }
        if (wp_on) ModWpUpdate(wp, out, wcur + size_t(x) * kModWpEntry, a_pe, a_te, a_valid);
        // ---- slide the windows
        rWW = rW;
        rW = out;
        rNW = rN;
        rN = rNE;
        rNE = rNEE;
        rNEE = aNEE;
        rNN = aNN;
#pragma unroll
        for (int j = 0; REFS && j < 4; j++) {
          rvl[j] = rv[j];
          rvtl[j] = rvt[j];
          rv[j] = av[j];
          rvt[j] = avt[j];
        }
      }
    }
  }
  uint32_t err = r.err;
  if (r.state != (0x13u << 16)) err |= 1;
  const uint64_t consumed = uint64_t(r.br.idx) * 32 - uint64_t(r.br.bits);
  if (consumed > uint64_t(S.size_bytes) * 8) err |= 2;
  *S.status = err;
  *S.end_bit = uint32_t(consumed);
}

// ---- inverse transforms on whole channels (one launch per transform step)
struct ModRct {
  int32_t* c[3];  // the three channels in stored order
  uint32_t stride[3];
  uint32_t w, h, type;
};
// Every transform kernel takes an array of parameter blocks, one per blockIdx.z: a launch does the same step of many
// frames (their k-th inverse transform), grid x / y sized for the largest.
__global__ __launch_bounds__(256) void k_modular_rct(const ModRct* ops) {  // rct.cc:97-147
  const ModRct& P = ops[blockIdx.z];
  const uint32_t x = blockIdx.x * 256 + threadIdx.x;
  if (x >= P.w) return;
  for (uint32_t y = blockIdx.y; y < P.h; y += gridDim.y) {
  const uint32_t perm = P.type / 7, custom = P.type % 7, second = custom >> 1, third = custom & 1;
  int32_t x0 = P.c[0][size_t(y) * P.stride[0] + x], x1 = P.c[1][size_t(y) * P.stride[1] + x], x2 = P.c[2][size_t(y) * P.stride[2] + x];
  int32_t o0, o1, o2;
  if (custom == 6) {  // YCoCg
    const int32_t tmp = int32_t(uint32_t(x0) - uint32_t(x2 >> 1));
    const int32_t G = int32_t(uint32_t(x2) + uint32_t(tmp));
    const int32_t B = int32_t(uint32_t(tmp) - uint32_t(x1 >> 1));
    o0 = int32_t(uint32_t(B) + uint32_t(x1));
    o1 = G;
    o2 = B;
  } else {
    if (third) x2 = int32_t(uint32_t(x2) + uint32_t(x0));
    if (second == 1) x1 = int32_t(uint32_t(x1) + uint32_t(x0));
    else if (second == 2) x1 = int32_t(uint32_t(x1) + uint32_t(int32_t(uint32_t(x0) + uint32_t(x2)) >> 1));
    o0 = x0;
    o1 = x1;
    o2 = x2;
  }
  P.c[perm % 3][size_t(y) * P.stride[perm % 3] + x] = o0;
  P.c[(perm + 1 + perm / 3) % 3][size_t(y) * P.stride[(perm + 1 + perm / 3) % 3] + x] = o1;
  P.c[(perm + 2 - perm / 3) % 3][size_t(y) * P.stride[(perm + 2 - perm / 3) % 3] + x] = o2;
  }
}

__device__ __forceinline__ int64_t ModSmoothTendency(int64_t before, int64_t avg, int64_t next) {  // squeeze.h:54-77
  const bool falling = before >= avg && avg >= next, rising = before <= avg && avg <= next;
  if (!falling && !rising) return 0;
  int64_t t = (4 * before - 3 * next - avg + (falling ? 6 : -6)) / 12;
  const int64_t odd = t & 1, lim_a = 2 * (before - avg), lim_b = 2 * (avg - next);
  if (falling) {
    if (t - odd > lim_a) t = lim_a + 1;
    if (t + (t & 1) > lim_b) t = lim_b;
  } else {
    if (t + odd < lim_a) t = lim_a - 1;
    if (t - (t & 1) < lim_b) t = lim_b;
  }
  return t;
}
struct ModUnsqueeze {
  const int32_t* avg;
  const int32_t* res;
  int32_t* out;
  uint32_t lines, na, nr;             // lines to do; averages / residuals per line
  uint32_t avg_line, avg_step, res_line, res_step, out_line, out_step;  // strides between lines / between samples of a line
};
// One thread per line (a row for a horizontal step, a column for a vertical one): squeeze.cc:128-330. A line is a serial
// chain through the smooth-tendency term; its loads are issued four samples ahead of that chain.
__global__ __launch_bounds__(64) void k_modular_unsqueeze(const ModUnsqueeze* ops) {
  const ModUnsqueeze& P = ops[blockIdx.y];
  const uint32_t l = blockIdx.x * 64 + threadIdx.x;
  if (l >= P.lines) return;
  const int32_t* avg = P.avg + size_t(l) * P.avg_line;
  const int32_t* res = P.res + size_t(l) * P.res_line;
  int32_t* out = P.out + size_t(l) * P.out_line;
  const uint32_t nr = P.nr, na = P.na;
  const size_t as = P.avg_step, rs = P.res_step, os = P.out_step;
  int64_t a = nr ? avg[0] : 0, before = a;
  for (uint32_t i = 0; i < nr; i += 4) {
    int32_t an[4], rr[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      an[k] = i + k + 1 < na ? avg[size_t(i + k + 1) * as] : 0;
      rr[k] = i + k < nr ? res[size_t(i + k) * rs] : 0;
    }
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      if (i + k >= nr) break;
      const int64_t next = i + k + 1 < na ? int64_t(an[k]) : a;
      const int64_t diff = int64_t(rr[k]) + ModSmoothTendency(before, a, next);
      const int64_t first = a + diff / 2, second = first - diff;
      out[size_t(2 * (i + k)) * os] = int32_t(first);
      out[size_t(2 * (i + k) + 1) * os] = int32_t(second);
      before = second;
      a = next;
    }
  }
  if (na > nr) out[size_t(2 * nr) * os] = avg[size_t(na - 1) * as];
}

struct ModPalette {  // palette.cc:26-202, the form without delta entries and predictor (nb_deltas == 0, predictor 0)
  const int32_t* palette;  // [nb_channels][palette_w]
  const int32_t* index;    // the index channel
  int32_t* out[4];
  uint32_t palette_w, nb, w, h, bit_depth, index_stride, out_stride;
};
__constant__ int16_t c_palette_delta[72][3] = {
    {0, 0, 0},       {4, 4, 4},       {11, 0, 0},      {0, 0, -13},     {0, -12, 0},     {-10, -10, -10},
    {-18, -18, -18}, {-27, -27, -27}, {-18, -18, 0},   {0, 0, -32},     {-32, 0, 0},     {-37, -37, -37},
    {0, -32, -32},   {24, 24, 45},    {50, 50, 50},    {-45, -24, -24}, {-24, -45, -45}, {0, -24, -24},
    {-34, -34, 0},   {-24, 0, -24},   {-45, -45, -24}, {64, 64, 64},    {-32, 0, -32},   {0, -32, 0},
    {-32, 0, 32},    {-24, -45, -24}, {45, 24, 45},    {24, -24, -45},  {-45, -24, 24},  {80, 80, 80},
    {64, 0, 0},      {0, 0, -64},     {0, -64, -64},   {-24, -24, 45},  {96, 96, 96},    {64, 64, 0},
    {45, -24, -24},  {34, -34, 0},    {112, 112, 112}, {24, -45, -45},  {45, 45, -24},   {0, -32, 32},
    {24, -24, 45},   {0, 96, 96},     {45, -24, 24},   {24, -45, -24},  {-24, -45, 24},  {0, -64, 0},
    {96, 0, 0},      {128, 128, 128}, {64, 0, 64},     {144, 144, 144}, {96, 96, 0},     {-36, -36, 36},
    {45, -24, -45},  {45, -45, -24},  {0, 0, -96},     {0, 128, 128},   {0, 96, 0},      {45, 24, -45},
    {-128, 0, 0},    {24, -45, 24},   {-45, 24, -45},  {64, 0, -64},    {64, -64, -64},  {96, 0, 96},
    {45, -45, 24},   {24, 45, -45},   {64, 64, -64},   {128, 128, 0},   {0, 0, -128},    {-24, 45, -45}};
__device__ __forceinline__ int32_t ModPaletteValue(const ModPalette& P, int index, uint32_t c) {  // palette.h:25-140
  const int palette_size = int(P.palette_w);
  if (index < 0) {
    if (c >= 3) return 0;
    index = -(index + 1);
    index %= 1 + 2 * (72 - 1);
    int32_t r = int32_t(c_palette_delta[(index + 1) >> 1][c]) * ((index & 1) ? 1 : -1);
    if (P.bit_depth > 8) r *= int32_t(1) << (P.bit_depth - 8);
    return r;
  } else if (palette_size <= index && index < palette_size + 64) {
    if (c >= 3) return 0;
    index -= palette_size;
    index >>= c * 2;
    return int32_t((uint64_t(index % 4) * ((uint64_t(1) << P.bit_depth) - 1)) >> 2) + (1 << (P.bit_depth > 3 ? P.bit_depth - 3 : 0));
  } else if (palette_size + 64 <= index) {
    if (c >= 3) return 0;
    index -= palette_size + 64;
    if (c == 1) index /= 5;
    if (c == 2) index /= 25;
    return int32_t((uint64_t(index % 5) * ((uint64_t(1) << P.bit_depth) - 1)) >> 2);
  }
  return P.palette[size_t(c) * P.palette_w + index];
}
__global__ __launch_bounds__(256) void k_modular_palette(const ModPalette* ops) {
  const ModPalette& P = ops[blockIdx.z];
  const uint32_t x = blockIdx.x * 256 + threadIdx.x;
  if (x >= P.w) return;
  for (uint32_t y = blockIdx.y; y < P.h; y += gridDim.y) {
    int index = P.index[size_t(y) * P.index_stride + x];
    if (P.nb == 1) index = index < 0 ? 0 : (index >= int(P.palette_w) ? int(P.palette_w) - 1 : index);
    int32_t v[4];
    for (uint32_t c = 0; c < P.nb; c++) v[c] = ModPaletteValue(P, index, c);
    for (uint32_t c = 0; c < P.nb; c++) P.out[c][size_t(y) * P.out_stride + x] = v[c];  // (out[0] may alias the index channel)
  }
}

// ---- integer channels -> interleaved output samples (non-XYB Modular frames: dec_modular.cc:564-793 converts the
// integers to float with 1 / (2^bits - 1), the write stage makes the samples of the caller's format)
struct ModOutput {
  const int32_t* ch[4];  // colour channels (1 or 3) then alpha (or NULL)
  uint32_t stride[4];
  uint32_t num_color, has_alpha, bits, alpha_bits, w, h;  // bits / alpha_bits: ModSampleToFloat's `depth`
  PixelOut po;           // po.nc: 1 / 2 (grey, grey + alpha) or 3 / 4
  // frames with splines: fmode 1 = only write the colour samples as floats to fplanes ([3][h][w]; the splines are then
  // drawn over them), fmode 2 = take the colour from fplanes instead of the integer channels; 0 = neither
  float* fplanes;
  uint32_t fmode;
  // XYB Modular frames: ch[0..2] = Y, X, B - Y in units of xyb_factor (X, Y, B); color = the colour stage's parameters
  uint32_t xyb;
  float xyb_factor[3];
  FilterParams color;
};
// One decoded integer -> the float the render pipeline starts from. `depth`: low byte = bits per sample, bits 8..15 = exponent
// bits of a floating-point sample type (0 = integer samples).
//  * integers (dec_modular.cc:655-690): v * 1 / (2^bits - 1), in single precision below 23 bits, in double precision from there;
//  * floats (dec_modular.cc:128-185, int_to_float): the integer IS the sample's bit pattern in a [bits]-bit float with
//    [exp] exponent bits: binary32 as it is; narrower types widened exactly (subnormals normalised when exp < 8, NaN /
//    infinity keep their mantissa bits).
__device__ __forceinline__ float ModSampleToFloat(int32_t v, uint32_t depth) {
  const int bits = int(depth & 0xFF), exp_bits = int((depth >> 8) & 0xFF);
  if (exp_bits == 0) {
    if (bits < 23) return float(v) * (1.0f / float((uint64_t(1) << bits) - 1));
    return float(double(v) * (1.0 / double((uint64_t(1) << bits) - 1)));
  }
  uint32_t f = uint32_t(v);
  if (bits == 32) return __uint_as_float(f);
  const int exp_bias = (1 << (exp_bits - 1)) - 1, sign_shift = bits - 1, mant_bits = bits - exp_bits - 1, mant_shift = 23 - mant_bits;
  const uint32_t sign = ((f >> sign_shift) & 1u) ? 0x80000000u : 0u;
  f &= (1u << sign_shift) - 1;
  if (f == 0) return __uint_as_float(sign);
  int exp = int(f >> mant_bits);
  uint32_t mantissa = f & ((1u << mant_bits) - 1);
  if (exp == (1 << exp_bits) - 1) return __uint_as_float(sign | (0xFFu << 23) | (mantissa << mant_shift));
  mantissa <<= mant_shift;
  if (exp == 0 && exp_bits < 8) {  // subnormal of the narrow type: a normal binary32 (mantissa != 0 here, so the shift ends)
    const int up = __clz(int(mantissa)) - 8;  // places until bit 23 is set
    mantissa <<= up;
    exp = 1 - up;
    mantissa &= 0x7FFFFF;
  }
  exp += 127 - exp_bias;
  return __uint_as_float(sign | (uint32_t(exp) << 23) | mantissa);
}
__global__ __launch_bounds__(256) void k_modular_output(const ModOutput* ops) {
  const ModOutput& P = ops[blockIdx.z];
  const uint32_t x = blockIdx.x * 256 + threadIdx.x;
  if (x >= P.w) return;
  for (uint32_t y = blockIdx.y; y < P.h; y += gridDim.y) {
  float v[4];
  if (P.xyb) {  // dec_modular.cc:583-631 (MultiplySum for B)
    const int32_t iy = P.ch[0][size_t(y) * P.stride[0] + x], ix = P.ch[1][size_t(y) * P.stride[1] + x], ib = P.ch[2][size_t(y) * P.stride[2] + x];
    v[0] = float(ix) * P.xyb_factor[0];
    v[1] = float(iy) * P.xyb_factor[1];
    v[2] = float(ib + iy) * P.xyb_factor[2];
  } else {
    for (uint32_t c = 0; c < P.num_color; c++) v[c] = ModSampleToFloat(P.ch[c][size_t(y) * P.stride[c] + x], P.bits);
  }
  if (P.fmode == 1) {
    for (uint32_t c = 0; c < 3; c++) P.fplanes[(size_t(c) * P.h + y) * P.w + x] = v[c];
    continue;
  }
  if (P.fmode == 2)
    for (uint32_t c = 0; c < 3; c++) v[c] = P.fplanes[(size_t(c) * P.h + y) * P.w + x];
  if (P.xyb) XybToRgb(P.color, v[0], v[1], v[2], &v[0], &v[1], &v[2]);  // (after the splines, like every XYB frame)
  const float a = P.has_alpha ? ModSampleToFloat(P.ch[P.num_color][size_t(y) * P.stride[P.num_color] + x], P.alpha_bits) : 1.0f;
  const uint32_t nc = P.po.nc, ncol = nc < 3 ? 1u : 3u;
  float s[4];
  for (uint32_t c = 0; c < ncol; c++) s[c] = v[P.num_color == 1 ? 0 : c];  // grey images replicate into RGB output
  if (nc == 2 || nc == 4) {
    s[ncol] = a;
    if (P.po.orient & 8) {  // un-premultiplied on the way out (PixelOut::orient)
      const float m = 1.0f / fmaxf(kSmallAlphaOut, a);
      for (uint32_t c = 0; c < ncol; c++) s[c] *= m;
    }
  }
  int dx, dy;
  const size_t base = PixelOutIndex(P.po, int(x), int(y), &dx, &dy) * nc;
  for (uint32_t c = 0; c < nc; c++) {
    const float f = s[c];
    if (P.po.type == 2) {
      const float m = float((1u << P.po.bits) - 1u);
      const float t = __builtin_amdgcn_fmed3f(f * m + c_dither[((uint32_t(dy) + c * 13) & 31) * 32 + ((uint32_t(dx) + c * 23) & 31)], 0.0f, m);
      static_cast<uint8_t*>(P.po.dst)[base + c] = uint8_t(__float2int_rn(t));
    } else if (P.po.type == 3) {
      const float m = float((1u << P.po.bits) - 1u);
      uint32_t u = uint32_t(__float2int_rn(__builtin_amdgcn_fmed3f(f * m, 0.0f, m)));
      if (P.po.swap) u = ((u & 0xFF) << 8) | (u >> 8);
      static_cast<uint16_t*>(P.po.dst)[base + c] = uint16_t(u);
    } else if (P.po.type == 5) {
      uint32_t u = __half_as_ushort(__float2half_rn(f));
      if (P.po.swap) u = ((u & 0xFF) << 8) | (u >> 8);
      static_cast<uint16_t*>(P.po.dst)[base + c] = uint16_t(u);
    } else {
      uint32_t u = __float_as_uint(f);
      if (P.po.swap) u = __builtin_bswap32(u);
      static_cast<uint32_t*>(P.po.dst)[base + c] = u;
    }
  }
  }
}

}  // namespace jxlhip
#endif  // JXL_HIP_MODULAR_H_
