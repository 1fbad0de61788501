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
// a batch of frames fills the machine). Tables (tree nodes, alias / prefix tables, context maps) stay in global memory:
// they differ per stream when the encoder used local trees, and the walk is bound by its dependent loads either way.
// The host front-end (csrc/host/jxh_modframe.h) parses every stream's group header (transforms, optional local tree and
// histograms) and hands the kernel flat descriptors; samples are written straight into the frame's channel buffers.
#ifndef JXL_HIP_MODULAR_H_
#define JXL_HIP_MODULAR_H_

#include "jxl_hip_kernels.h"

namespace jxlhip {

struct ModTreeNode {  // 32 bytes
  int32_t property;   // -1 = leaf
  int32_t splitval;
  uint32_t lchild, rchild;  // leaves: lchild = context id
  uint32_t predictor;
  int32_t offset;
  uint32_t multiplier;
  uint32_t pad;
};

struct ModCode {  // one entropy code (global or local to a stream); pointers into device memory
  const uint8_t* ctx_map;   // context -> cluster
  const uint2* alias;       // cluster << log_alpha (ANS)
  const uint32_t* cfg;      // per cluster: split_exp | msb << 8 | lsb << 16
  const uint32_t* prefix_table;
  const uint32_t* prefix_offset;
  uint32_t log_alpha, use_prefix, lz77, lz_min_symbol, lz_min_length, lz_len_cfg, lz_dist_ctx, pad;
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

struct ModReader {
  BitReader br;
  uint32_t state;
  const ModCode* code;
  uint32_t* window;
  uint32_t mask, num_decoded, num_to_copy, copy_pos, num_special, dist_mult, err;
};
__device__ __forceinline__ uint32_t ModSymbol(ModReader& r, uint32_t cluster) {
  const ModCode& T = *r.code;
  BrRefill(r.br);
  if (T.use_prefix) {
    const uint32_t po = T.prefix_offset[cluster], max_len = po >> 24;
    if (max_len == 0) return T.prefix_table[po & 0xFFFFFFu] >> 8;
    const uint32_t e = T.prefix_table[(po & 0xFFFFFFu) + uint32_t(r.br.buf & ((1u << max_len) - 1))];
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
__device__ __forceinline__ uint32_t ModUint(ModReader& r, uint32_t cfg, uint32_t token) {
  const uint32_t split_exp = cfg & 0xFF, msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
  if (token < (1u << split_exp)) return token;
  const uint32_t nbits = (split_exp - (msb + lsb) + ((token - (1u << split_exp)) >> (msb + lsb))) & 31u;
  const uint32_t low = token & ((1u << lsb) - 1), hi = token >> lsb;
  BrRefill(r.br);
  const uint32_t bits = BrRead(r.br, nbits);
  return (((((1u << msb) | (hi & ((1u << msb) - 1))) << nbits) | bits) << lsb) | low;
}
// dec_ans.h:288-353 (ReadHybridUintClusteredInlined with LZ77)
__device__ __forceinline__ uint32_t ModRead(ModReader& r, uint32_t cluster) {
  const ModCode& T = *r.code;
  if (T.lz77 && r.num_to_copy > 0) {
    // (copy_pos == num_decoded only for a copy at the very start of a stream, distance 0: zeros, dec_ans.h:320-327)
    const uint32_t v = r.copy_pos >= r.num_decoded ? 0u : r.window[r.copy_pos & r.mask];
    r.copy_pos++;
    r.num_to_copy--;
    r.window[(r.num_decoded++) & r.mask] = v;
    return v;
  }
  const uint32_t token = ModSymbol(r, cluster);
  if (T.lz77 && token >= T.lz_min_symbol) {
    r.num_to_copy = ModUint(r, T.lz_len_cfg, token - T.lz_min_symbol) + T.lz_min_length;
    const uint32_t dtok = ModSymbol(r, T.lz_dist_ctx);
    uint32_t distance = ModUint(r, T.cfg[T.lz_dist_ctx], dtok);
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
    const uint32_t v = distance == 0 ? 0u : r.window[(r.copy_pos++) & r.mask];
    if (distance == 0) r.copy_pos++;
    r.num_to_copy--;
    r.window[(r.num_decoded++) & r.mask] = v;
    return v;
  }
  const uint32_t v = ModUint(r, T.cfg[cluster], token);
  if (T.lz77) r.window[(r.num_decoded++) & r.mask] = v;
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

// Self-correcting weighted predictor (context_predict.h:66-218). State: per row-parity arrays of width + 2 entries:
// error[2][w + 2], pred_errors[4][2][w + 2] in the stream's scratch.
struct ModWp {
  int32_t* error;        // [2 * (w + 2)]
  uint32_t* pred_err[4];  // each [2 * (w + 2)]
  int64_t prediction[4];
  int64_t pred;
  const int32_t* hd;     // p1C p2C p3Ca p3Cb p3Cc p3Cd p3Ce w0 w1 w2 w3
};
__device__ __forceinline__ uint32_t ModErrorWeight(uint64_t x, uint32_t maxweight) {
  int shift = ModFloorLog2(x + 1) - 5;
  if (shift < 0) shift = 0;
  const uint32_t div = (1u << 24) / (uint32_t(x >> shift) + 1u);
  return uint32_t(4 + ((uint64_t(maxweight) * div) >> shift));
}
__device__ __forceinline__ int64_t ModWpPredict(ModWp& s, uint32_t x, uint32_t y, uint32_t xsize, int64_t N, int64_t W, int64_t NE, int64_t NW,
                                               int64_t NN, int32_t* max_err_prop) {
  const uint32_t cur = (y & 1) ? 0 : (xsize + 2), prev = (y & 1) ? (xsize + 2) : 0;
  const uint32_t pos_N = prev + x;
  const uint32_t pos_NE = x < xsize - 1 ? pos_N + 1 : pos_N;
  const uint32_t pos_NW = x > 0 ? pos_N - 1 : pos_N;
  uint32_t weights[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t e = s.pred_err[i][pos_N] + s.pred_err[i][pos_NE] + s.pred_err[i][pos_NW];
    weights[i] = ModErrorWeight(e, uint32_t(s.hd[7 + i]));
  }
  N *= 8; W *= 8; NE *= 8; NW *= 8; NN *= 8;
  const int64_t teW = x == 0 ? 0 : s.error[cur + x - 1];
  const int64_t teN = s.error[pos_N], teNW = s.error[pos_NW], teNE = s.error[pos_NE];
  const int64_t sumWN = teN + teW;
  {
    int64_t p = teW;
    if (ModAbs64(teN) > ModAbs64(p)) p = teN;
    if (ModAbs64(teNW) > ModAbs64(p)) p = teNW;
    if (ModAbs64(teNE) > ModAbs64(p)) p = teNE;
    *max_err_prop = int32_t(p);
  }
  s.prediction[0] = W + NE - N;
  s.prediction[1] = N - (((sumWN + teNE) * s.hd[0]) >> 5);
  s.prediction[2] = W - (((sumWN + teNW) * s.hd[1]) >> 5);
  s.prediction[3] = N - ((teNW * s.hd[2] + teN * s.hd[3] + teNE * s.hd[4] + (NN - N) * s.hd[5] + (NW - W) * s.hd[6]) >> 5);
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
    s.pred = (sum * int64_t((1u << 24) / ws)) >> 24;
  }
  if (((teN ^ teW) | (teN ^ teNW)) > 0) return (s.pred + 3) >> 3;
  const int64_t mx = W > NE ? (W > N ? W : N) : (NE > N ? NE : N), mn = W < NE ? (W < N ? W : N) : (NE < N ? NE : N);
  s.pred = s.pred < mn ? mn : (s.pred > mx ? mx : s.pred);
  return (s.pred + 3) >> 3;
}
__device__ __forceinline__ void ModWpUpdate(ModWp& s, int64_t val, uint32_t x, uint32_t y, uint32_t xsize) {
  const uint32_t cur = (y & 1) ? 0 : (xsize + 2), prev = (y & 1) ? (xsize + 2) : 0;
  val *= 8;
  s.error[cur + x] = int32_t(s.pred - val);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int64_t err = (ModAbs64(s.prediction[i] - val) + 3) >> 3;
    s.pred_err[i][cur + x] = uint32_t(err);
    s.pred_err[i][prev + x + 1] += uint32_t(err);
  }
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

// One lane per stream. `streams` holds `n` descriptors.
__global__ __launch_bounds__(64) void k_modular_streams(const ModStream* streams, uint32_t n) {
  const uint32_t si = blockIdx.x * 64 + threadIdx.x;
  if (si >= n) return;
  const ModStream& S = streams[si];
  ModReader r;
  r.code = S.code;
  r.window = S.lz_window;
  r.mask = S.lz_window_mask;
  r.num_decoded = r.num_to_copy = r.copy_pos = 0;
  r.dist_mult = S.dist_multiplier;
  r.num_special = S.dist_multiplier ? 120u : 0u;
  r.err = 0;
  r.br.p = S.words;
  r.br.nwords = (S.size_bytes + 3) / 4;
  r.br.idx = S.bit_offset >> 5;
  r.br.buf = 0;
  r.br.bits = 0;
  BrRefill(r.br);
  if (S.bit_offset & 31) BrRead(r.br, S.bit_offset & 31);
  r.state = 0x13u << 16;
  if (!S.code->use_prefix) {
    BrRefill(r.br);
    r.state = BrRead(r.br, 16);
    BrRefill(r.br);
    r.state |= BrRead(r.br, 16) << 16;
  }
  const ModTreeNode* tree = S.tree;
  int32_t props[kModMaxProps];
#pragma unroll
  for (int i = 0; i < kModMaxProps; i++) props[i] = 0;
  for (uint32_t ci = 0; ci < S.num_channels && !r.err; ci++) {
    const ModChannel ch = S.channels[ci];
    if (!ch.w || !ch.h) continue;
    // previous channels of the same shape, nearest first (context_predict.h:419-451)
    const ModChannel* refs[4];
    uint32_t nrefs = 0;
    for (int j = int(ci) - 1; j >= 0 && nrefs * 4 < S.num_props - 16 && nrefs < 4; j--)
      if (S.channels[j].sig == ch.sig && S.channels[j].w == ch.w && S.channels[j].h == ch.h) refs[nrefs++] = &S.channels[j];
    ModWp wp;
    wp.hd = S.wp;
    if (S.uses_wp) {
      const uint32_t span = 2 * (ch.w + 2);
      wp.error = S.wp_scratch;
      for (int i = 0; i < 4; i++) wp.pred_err[i] = reinterpret_cast<uint32_t*>(S.wp_scratch) + span * (1 + i);
      for (uint32_t i = 0; i < span * 5; i++) S.wp_scratch[i] = 0;
    }
    const ptrdiff_t stride = ptrdiff_t(ch.stride);
#pragma unroll
    for (int i = 16; i < kModMaxProps; i++) props[i] = 0;  // slots of previous channels this channel does not have
    props[0] = int32_t(S.first_channel_index + ci);
    props[1] = int32_t(S.stream_id);
    for (uint32_t y = 0; y < ch.h && !r.err; y++) {
      int32_t* p = ch.data + size_t(y) * ch.stride;
      props[2] = int32_t(y);
      props[9] = 0;
      for (uint32_t x = 0; x < ch.w; x++) {
        const int32_t* pp = p + x;
        const int64_t left = x ? pp[-1] : (y ? pp[-stride] : 0);
        const int64_t top = y ? pp[-stride] : left;
        const int64_t topleft = (x && y) ? pp[-1 - stride] : left;
        const int64_t topright = (x + 1 < ch.w && y) ? pp[1 - stride] : top;
        const int64_t leftleft = x > 1 ? pp[-2] : left;
        const int64_t toptop = y > 1 ? pp[-2 * stride] : top;
        const int64_t toprightright = (x + 2 < ch.w && y) ? pp[2 - stride] : topright;
        props[3] = int32_t(x);
        props[4] = int32_t(top > 0 ? top : -top);
        props[5] = int32_t(left > 0 ? left : -left);
        props[6] = int32_t(top);
        props[7] = int32_t(left);
        props[8] = int32_t(left - props[9]);  // uses the previous pixel's property 9
        props[9] = int32_t(left + top - topleft);
        props[10] = int32_t(left - topleft);
        props[11] = int32_t(topleft - top);
        props[12] = int32_t(top - topright);
        props[13] = int32_t(top - toptop);
        props[14] = int32_t(left - leftleft);
        int64_t wp_pred = 0;
        if (S.uses_wp) wp_pred = ModWpPredict(wp, x, y, ch.w, top, left, topright, topleft, toptop, &props[15]);
        for (uint32_t j = 0; j < nrefs; j++) {
          const int32_t* rp = refs[j]->data + size_t(y) * refs[j]->stride;
          const int32_t* rprev = refs[j]->data + size_t(y ? y - 1 : 0) * refs[j]->stride;
          const int64_t v = rp[x];
          const int64_t vl = x ? rp[x - 1] : 0;
          const int64_t vt = y ? rprev[x] : vl;
          const int64_t vtl = (x && y) ? rprev[x - 1] : vl;
          const int64_t vp = ModClampedGradient(int32_t(vl), int32_t(vt), int32_t(vtl));
          props[16 + 4 * j] = int32_t(ModAbs64(v));
          props[17 + 4 * j] = int32_t(v);
          props[18 + 4 * j] = int32_t(ModAbs64(v - vp));
          props[19 + 4 * j] = int32_t(v - vp);
        }
        uint32_t pos = 0;
        ModTreeNode nd = tree[0];
        while (nd.property >= 0) {
          pos = props[nd.property < kModMaxProps ? nd.property : 0] > nd.splitval ? nd.lchild : nd.rchild;
          nd = tree[pos];
        }
        const int64_t guess = int64_t(nd.offset) + ModPredict(nd.predictor, left, top, toptop, topleft, topright, leftleft, toprightright, wp_pred);
        const uint32_t v = ModRead(r, S.code->ctx_map[nd.lchild]);
        const int64_t val = int64_t(int32_t((v >> 1) ^ (0u - (v & 1)))) * int64_t(nd.multiplier) + guess;
        p[x] = int32_t(val);
        if (S.uses_wp) ModWpUpdate(wp, p[x], x, y, ch.w);
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
__global__ __launch_bounds__(256) void k_modular_rct(ModRct P) {  // rct.cc:97-147
  const uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= P.w || y >= P.h) return;
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
// One thread per line (a row for a horizontal step, a column for a vertical one): squeeze.cc:128-330.
__global__ __launch_bounds__(64) void k_modular_unsqueeze(ModUnsqueeze P) {
  const uint32_t l = blockIdx.x * 64 + threadIdx.x;
  if (l >= P.lines) return;
  const int32_t* avg = P.avg + size_t(l) * P.avg_line;
  const int32_t* res = P.res + size_t(l) * P.res_line;
  int32_t* out = P.out + size_t(l) * P.out_line;
  int64_t before = 0;
  for (uint32_t i = 0; i < P.nr; i++) {
    const int64_t a = avg[size_t(i) * P.avg_step], next = i + 1 < P.na ? avg[size_t(i + 1) * P.avg_step] : a;
    if (i == 0) before = a;
    const int64_t diff = int64_t(res[size_t(i) * P.res_step]) + ModSmoothTendency(before, a, next);
    const int64_t first = a + diff / 2, second = first - diff;
    out[size_t(2 * i) * P.out_step] = int32_t(first);
    out[size_t(2 * i + 1) * P.out_step] = int32_t(second);
    before = second;
  }
  if (P.na > P.nr) out[size_t(2 * P.nr) * P.out_step] = avg[size_t(P.na - 1) * P.avg_step];
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
__global__ __launch_bounds__(256) void k_modular_palette(ModPalette P) {
  const uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= P.w || y >= P.h) return;
  int index = P.index[size_t(y) * P.index_stride + x];
  if (P.nb == 1) index = index < 0 ? 0 : (index >= int(P.palette_w) ? int(P.palette_w) - 1 : index);
  int32_t v[4];
  for (uint32_t c = 0; c < P.nb; c++) v[c] = ModPaletteValue(P, index, c);
  for (uint32_t c = 0; c < P.nb; c++) P.out[c][size_t(y) * P.out_stride + x] = v[c];  // (out[0] may alias the index channel)
}

// ---- integer channels -> interleaved output samples (non-XYB Modular frames: dec_modular.cc:564-793 converts the
// integers to float with 1 / (2^bits - 1), the write stage makes the samples of the caller's format)
struct ModOutput {
  const int32_t* ch[4];  // colour channels (1 or 3) then alpha (or NULL)
  uint32_t stride[4];
  uint32_t num_color, has_alpha, bits, alpha_bits, w, h;
  PixelOut po;           // po.nc: 1 / 2 (grey, grey + alpha) or 3 / 4
};
__global__ __launch_bounds__(256) void k_modular_output(ModOutput P) {
  const uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= P.w || y >= P.h) return;
  const float mul = 1.0f / float((uint64_t(1) << P.bits) - 1);
  float v[4];
  for (uint32_t c = 0; c < P.num_color; c++) v[c] = float(P.ch[c][size_t(y) * P.stride[c] + x]) * mul;
  const float a = P.has_alpha ? float(P.ch[P.num_color][size_t(y) * P.stride[P.num_color] + x]) * (1.0f / float((uint64_t(1) << P.alpha_bits) - 1)) : 1.0f;
  const uint32_t nc = P.po.nc, ncol = nc < 3 ? 1u : 3u;
  float s[4];
  for (uint32_t c = 0; c < ncol; c++) s[c] = v[P.num_color == 1 ? 0 : c];  // grey images replicate into RGB output
  if (nc == 2 || nc == 4) s[ncol] = a;
  const size_t base = (size_t(y) * P.po.xsize + x) * nc;
  for (uint32_t c = 0; c < nc; c++) {
    const float f = s[c];
    if (P.po.type == 2) {
      const float m = float((1u << P.po.bits) - 1u);
      const float t = __builtin_amdgcn_fmed3f(f * m + c_dither[((y + c * 13) & 31) * 32 + ((x + c * 23) & 31)], 0.0f, m);
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

}  // namespace jxlhip
#endif  // JXL_HIP_MODULAR_H_
