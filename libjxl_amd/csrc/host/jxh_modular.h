// libjxl_amd host front-end (product code; runs on the CPU ahead of the GPU hot path): Modular sub-bitstreams — MA tree,
// predictors (incl. the self-correcting weighted predictor), per-channel pixel decode, inverse RCT / Palette / Squeeze.
// Follows: reference lib/jxl/modular/encoding/dec_ma.cc:107-182 (tree), encoding.cc:148-506 (channel
// decode; this file restates the generic path, the reference's specialised fast paths are equivalent),
// encoding.cc:554-724 (stream layout), context_predict.h:38-330 (weighted predictor), :372-560
// (properties, predictors), modular/transform/transform.cc:36-100, rct.cc:30-147, palette.cc:26-202,
// squeeze.cc:128-517 + squeeze.h:54-77.
#ifndef JXH_MODULAR_H_
#define JXH_MODULAR_H_

#include <array>
#include <cstdlib>
#include <vector>

#include "jxh_entropy.h"

namespace jxh {

struct MChannel {
  size_t w = 0, h = 0;
  int hshift = 0, vshift = 0;
  std::vector<int32_t> d;
  MChannel() {}
  MChannel(size_t w_, size_t h_, int hs = 0, int vs = 0) : w(w_), h(h_), hshift(hs), vshift(vs), d(w_ * h_, 0) {}
  int32_t* Row(size_t y) { return d.data() + y * w; }
  const int32_t* Row(size_t y) const { return d.data() + y * w; }
};

struct SqueezeStep {  // one entry of a Squeeze transform (modular/transform/squeeze_params.cc:14-24)
  bool horizontal = false, in_place = false;
  uint32_t begin_c = 0, num_c = 2;
};
struct MTransform {
  uint32_t id = 0;  // 0 RCT, 1 Palette, 2 Squeeze
  uint32_t begin_c = 0, rct_type = 6;
  uint32_t num_c = 3, nb_colors = 256, nb_deltas = 0, predictor = 0;
  std::vector<SqueezeStep> steps;  // Squeeze; empty in the stream = the default sequence (filled in by SqueezeMeta)
};

struct WpHeader {
  int32_t p1C = 16, p2C = 10, p3Ca = 7, p3Cb = 7, p3Cc = 7, p3Cd = 0, p3Ce = 0;
  uint32_t w[4] = {0xd, 0xc, 0xc, 0xc};
};

struct MImage {
  std::vector<MChannel> ch;
  size_t nb_meta = 0;
  int bitdepth = 8;
  std::vector<MTransform> transforms;
  WpHeader wp;
};

struct TreeNode {
  int property;  // -1 = leaf
  int32_t splitval;
  uint32_t lchild, rchild;  // for leaves: lchild = leaf (context) id
  uint32_t predictor;
  int64_t offset;
  uint32_t multiplier;
};
typedef std::vector<TreeNode> MTree;

struct MGlobal {  // global tree + code shared by all streams of a frame
  bool have = false;
  MTree tree;
  EntropyCode code;
};

static inline void DecodeTree(BitReader& br, MTree* tree, size_t limit) {
  EntropyCode code;
  DecodeHistograms(br, 6, &code);
  SymbolReader rd(&code, &br);
  size_t leaf_id = 0, to_decode = 1;
  tree->clear();
  limit = std::min<size_t>(limit, size_t(1) << 22);
  while (to_decode > 0) {
    JXH_CHECK(!br.Overread(), "tree: out of data");
    JXH_CHECK(tree->size() <= limit, "tree too large");
    to_decode--;
    uint32_t prop1 = rd.Read(1);
    JXH_CHECK(prop1 <= 256, "invalid tree property");
    int property = int(prop1) - 1;
    if (property == -1) {
      uint32_t predictor = rd.Read(2);
      JXH_CHECK(predictor < 14, "invalid predictor");
      int64_t off = UnpackSigned(rd.Read(3));
      uint32_t mul_log = rd.Read(4);
      JXH_CHECK(mul_log < 31, "invalid multiplier log");
      uint32_t mul_bits = rd.Read(5);
      JXH_CHECK(mul_bits < (1u << (31u - mul_log)) - 1u, "invalid multiplier");
      tree->push_back({-1, 0, uint32_t(leaf_id++), 0, predictor, off, (mul_bits + 1u) << mul_log});
      continue;
    }
    int32_t splitval = UnpackSigned(rd.Read(0));
    uint32_t l = uint32_t(tree->size() + to_decode + 1), r = uint32_t(tree->size() + to_decode + 2);
    tree->push_back({property, splitval, l, r, 0, 0, 1});
    to_decode += 2;
  }
  JXH_CHECK(rd.FinalStateOk(), "tree: bad ANS final state");
}

// ---- weighted predictor state (context_predict.h:66-218)
struct WpState {
  static const int kExtra = 3;
  static const int64_t kRound = ((1 << kExtra) >> 1) - 1;
  int64_t prediction[4] = {0, 0, 0, 0};
  int64_t pred = 0;
  std::vector<uint32_t> pred_errors[4];
  std::vector<int32_t> error;
  const WpHeader& hd;
  uint32_t divlookup[64];
  WpState(const WpHeader& h, size_t xsize) : hd(h) {
    for (auto& p : pred_errors) p.assign((xsize + 2) * 2, 0);
    error.assign((xsize + 2) * 2, 0);
    for (int i = 0; i < 64; i++) divlookup[i] = (1u << 24) / (i + 1);
  }
  uint32_t ErrorWeight(uint64_t x, uint32_t maxweight) const {
    int shift = FloorLog2(x + 1) - 5;
    if (shift < 0) shift = 0;
    return uint32_t(4 + ((uint64_t(maxweight) * divlookup[x >> shift]) >> shift));
  }
  int64_t WeightedAverage(const int64_t* p, std::array<uint32_t, 4> w) const {
    uint32_t ws = 0;
    for (int i = 0; i < 4; i++) ws += w[i];
    int lw = FloorLog2(ws);
    ws = 0;
    for (int i = 0; i < 4; i++) {
      w[i] >>= lw - 4;
      ws += w[i];
    }
    int64_t sum = (ws >> 1) - 1;
    for (int i = 0; i < 4; i++) sum += p[i] * int64_t(w[i]);
    return (sum * int64_t(divlookup[ws - 1])) >> 24;
  }
  // Returns prediction; *max_err_prop receives the WP property value.
  int64_t Predict(size_t x, size_t y, size_t xsize, int64_t N, int64_t W, int64_t NE, int64_t NW, int64_t NN,
                  int32_t* max_err_prop) {
    size_t cur = (y & 1) ? 0 : (xsize + 2), prev = (y & 1) ? (xsize + 2) : 0;
    size_t pos_N = prev + x;
    size_t pos_NE = x < xsize - 1 ? pos_N + 1 : pos_N;
    size_t pos_NW = x > 0 ? pos_N - 1 : pos_N;
    std::array<uint32_t, 4> weights;
    for (int i = 0; i < 4; i++) {
      uint32_t e = pred_errors[i][pos_N] + pred_errors[i][pos_NE] + pred_errors[i][pos_NW];
      weights[i] = ErrorWeight(e, hd.w[i]);
    }
    N *= 8; W *= 8; NE *= 8; NW *= 8; NN *= 8;
    int64_t teW = x == 0 ? 0 : error[cur + x - 1];
    int64_t teN = error[pos_N], teNW = error[pos_NW], teNE = error[pos_NE];
    int64_t sumWN = teN + teW;
    if (max_err_prop) {
      int64_t p = teW;
      if (std::llabs(teN) > std::llabs(p)) p = teN;
      if (std::llabs(teNW) > std::llabs(p)) p = teNW;
      if (std::llabs(teNE) > std::llabs(p)) p = teNE;
      *max_err_prop = int32_t(p);
    }
    prediction[0] = W + NE - N;
    prediction[1] = N - (((sumWN + teNE) * hd.p1C) >> 5);
    prediction[2] = W - (((sumWN + teNW) * hd.p2C) >> 5);
    prediction[3] = N - ((teNW * hd.p3Ca + teN * hd.p3Cb + teNE * hd.p3Cc + (NN - N) * hd.p3Cd + (NW - W) * hd.p3Ce) >> 5);
    pred = WeightedAverage(prediction, weights);
    if (((teN ^ teW) | (teN ^ teNW)) > 0) return (pred + kRound) >> kExtra;
    int64_t mx = std::max(W, std::max(NE, N)), mn = std::min(W, std::min(NE, N));
    pred = std::max(mn, std::min(mx, pred));
    return (pred + kRound) >> kExtra;
  }
  void Update(int64_t val, size_t x, size_t y, size_t xsize) {
    size_t cur = (y & 1) ? 0 : (xsize + 2), prev = (y & 1) ? (xsize + 2) : 0;
    val *= 8;
    error[cur + x] = int32_t(pred - val);
    for (int i = 0; i < 4; i++) {
      int64_t err = (std::llabs(prediction[i] - val) + kRound) >> kExtra;
      pred_errors[i][cur + x] = uint32_t(err);
      pred_errors[i][prev + x + 1] += uint32_t(err);
    }
  }
};

static inline int32_t ClampedGradient(int32_t n, int32_t w, int32_t l) {
  int32_t m = std::min(n, w), M = std::max(n, w);
  int32_t grad = int32_t(uint32_t(n) + uint32_t(w) - uint32_t(l));
  int32_t gc = (l < m) ? M : grad;
  return (l > M) ? m : gc;
}

static inline int64_t PredictOne(uint32_t p, int64_t left, int64_t top, int64_t toptop, int64_t topleft,
                                 int64_t topright, int64_t leftleft, int64_t toprightright, int64_t wp) {
  switch (p) {
    case 0: return 0;
    case 1: return left;
    case 2: return top;
    case 3: return (left + top) / 2;
    case 4: {
      int64_t pp = left + top - topleft;
      return std::llabs(pp - left) < std::llabs(pp - top) ? left : top;
    }
    case 5: return ClampedGradient(int32_t(left), int32_t(top), int32_t(topleft));
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

// A channel whose part of the tree is one leaf (no decision left once the channel index and the stream id are known:
// what simple encoders and libjxl's fast efforts emit): no properties, one histogram, the predictor a compile-time
// constant for the common ones (P < 0: `leaf.predictor` at run time).
template <int P>
static inline void DecodeLeafChannel(SymbolReader& rd, size_t cluster, const TreeNode& leaf, MChannel& c) {
  const ptrdiff_t stride = ptrdiff_t(c.w);
  const uint32_t predictor = P < 0 ? leaf.predictor : uint32_t(P);
  const int64_t offset = leaf.offset, mul = int64_t(leaf.multiplier);
  for (size_t y = 0; y < c.h; y++) {
    int32_t* p = c.Row(y);
    rd.ReadRun(cluster, c.w, [&](size_t x, uint32_t v) {
      const int32_t* pp = p + x;
      const int64_t left = x ? pp[-1] : (y ? pp[-stride] : 0);
      const int64_t top = y ? pp[-stride] : left;
      const int64_t topleft = (x && y) ? pp[-1 - stride] : left;
      const int64_t topright = (x + 1 < c.w && y) ? pp[1 - stride] : top;
      const int64_t leftleft = x > 1 ? pp[-2] : left;
      const int64_t toptop = y > 1 ? pp[-2 * stride] : top;
      const int64_t toprightright = (x + 2 < c.w && y) ? pp[2 - stride] : topright;
      const int64_t guess = offset + PredictOne(predictor, left, top, toptop, topleft, topright, leftleft, toprightright, 0);
      p[x] = int32_t(int64_t(UnpackSigned(v)) * mul + guess);
    });
  }
}

// Decodes one channel (encoding.cc:148-506).
static inline void DecodeChannel(BitReader& br, SymbolReader& rd, const EntropyCode& code, const MTree& tree,
                                 const WpHeader& wph, int chan, int stream_id, MImage* img) {
  (void)br;
  MChannel& c = img->ch[chan];
  if (c.w == 0 || c.h == 0) return;
  // decisions on the channel index and the stream id are the same for every sample of the channel: taken once
  size_t root = 0;
  while (tree[root].property == 0 || tree[root].property == 1) {
    const TreeNode& n = tree[root];
    root = (n.property == 0 ? chan : stream_id) > n.splitval ? n.lchild : n.rchild;
  }
  // what the rest of the tree looks at
  int max_prop = 15;
  bool uses_wp = false;
  {
    std::vector<uint32_t> todo{uint32_t(root)};
    while (!todo.empty()) {
      const TreeNode& n = tree[todo.back()];
      todo.pop_back();
      if (n.property >= 0) {
        max_prop = std::max(max_prop, n.property);
        if (n.property == 15) uses_wp = true;
        todo.push_back(n.lchild);
        todo.push_back(n.rchild);
      } else if (n.predictor == 6) {
        uses_wp = true;
      }
    }
  }
  if (tree[root].property < 0 && !uses_wp) {
    const TreeNode& leaf = tree[root];
    const size_t cluster = code.ctx_map[leaf.lchild];
    switch (leaf.predictor) {
      case 0: DecodeLeafChannel<0>(rd, cluster, leaf, c); break;
      case 1: DecodeLeafChannel<1>(rd, cluster, leaf, c); break;
      case 2: DecodeLeafChannel<2>(rd, cluster, leaf, c); break;
      case 5: DecodeLeafChannel<5>(rd, cluster, leaf, c); break;
      default: DecodeLeafChannel<-1>(rd, cluster, leaf, c); break;
    }
    return;
  }
  size_t num_props = 16;
  if (max_prop >= 16) num_props = 16 + DivCeil(size_t(max_prop - 16 + 1), 4) * 4;
  // reference channels for the "previous channel" properties (context_predict.h:419-451)
  std::vector<int> refs;
  for (int j = chan - 1; j >= 0 && refs.size() * 4 < num_props - 16; j--) {
    const MChannel& r = img->ch[j];
    if (r.w != c.w || r.h != c.h || r.hshift != c.hshift || r.vshift != c.vshift) continue;
    refs.push_back(j);
  }
  std::vector<int32_t> props(num_props, 0);
  WpState wp(wph, uses_wp ? c.w : 0);  // (two rows of state per sub-predictor: only when the tree asks for it)
  const ptrdiff_t stride = ptrdiff_t(c.w);
  for (size_t y = 0; y < c.h; y++) {
    int32_t* p = c.Row(y);
    props[0] = chan;
    props[1] = stream_id;
    props[2] = int32_t(y);
    props[9] = 0;
    for (size_t x = 0; x < c.w; x++) {
      const int32_t* pp = p + x;
      int64_t left = x ? pp[-1] : (y ? pp[-stride] : 0);
      int64_t top = y ? pp[-stride] : left;
      int64_t topleft = (x && y) ? pp[-1 - stride] : left;
      int64_t topright = (x + 1 < c.w && y) ? pp[1 - stride] : top;
      int64_t leftleft = x > 1 ? pp[-2] : left;
      int64_t toptop = y > 1 ? pp[-2 * stride] : top;
      int64_t toprightright = (x + 2 < c.w && y) ? pp[2 - stride] : topright;
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
      if (uses_wp) wp_pred = wp.Predict(x, y, c.w, top, left, topright, topleft, toptop, &props[15]);
      size_t off = 16;
      for (int j : refs) {
        const MChannel& r = img->ch[j];
        const int32_t* rp = r.Row(y);
        const int32_t* rprev = r.Row(y ? y - 1 : 0);
        int64_t v = rp[x];
        int64_t vl = x ? rp[x - 1] : 0;
        int64_t vt = y ? rprev[x] : vl;
        int64_t vtl = (x && y) ? rprev[x - 1] : vl;
        int64_t vp = ClampedGradient(int32_t(vl), int32_t(vt), int32_t(vtl));
        props[off++] = int32_t(std::llabs(v));
        props[off++] = int32_t(v);
        props[off++] = int32_t(std::llabs(v - vp));
        props[off++] = int32_t(v - vp);
      }
      // tree walk
      size_t pos = root;
      while (tree[pos].property >= 0) {
        const TreeNode& n = tree[pos];
        pos = props[n.property] > n.splitval ? n.lchild : n.rchild;
      }
      const TreeNode& leaf = tree[pos];
      int64_t guess = leaf.offset + PredictOne(leaf.predictor, left, top, toptop, topleft, topright, leftleft,
                                               toprightright, wp_pred);
      uint32_t v = rd.ReadClustered(code.ctx_map[leaf.lchild]);
      int64_t val = int64_t(UnpackSigned(v)) * int64_t(leaf.multiplier) + guess;
      p[x] = int32_t(val);
      if (uses_wp) wp.Update(p[x], x, y, c.w);
    }
  }
}

static inline void ReadTransform(BitReader& br, MTransform* t) {
  t->id = ReadU32(br, Val(0), Val(1), Val(2), Val(3));
  JXH_CHECK(t->id != 3, "invalid transform id");
  if (t->id == 0 || t->id == 1) t->begin_c = ReadU32(br, Bits(3), BitsOffset(6, 8), BitsOffset(10, 72), BitsOffset(13, 1096));
  if (t->id == 0) {
    t->rct_type = ReadU32(br, Val(6), Bits(2), BitsOffset(4, 2), BitsOffset(6, 10));
    JXH_CHECK(t->rct_type < 42, "invalid RCT type");
  }
  if (t->id == 1) {
    t->num_c = ReadU32(br, Val(1), Val(3), Val(4), BitsOffset(13, 1));
    t->nb_colors = ReadU32(br, BitsOffset(8, 0), BitsOffset(10, 256), BitsOffset(12, 1280), BitsOffset(16, 5376));
    t->nb_deltas = ReadU32(br, Val(0), BitsOffset(8, 1), BitsOffset(10, 257), BitsOffset(16, 1281));
    t->predictor = uint32_t(br.Read(4));
    JXH_CHECK(t->predictor < 14, "invalid palette predictor");
  }
  if (t->id == 2) {  // transform.cc:77-87
    t->steps.resize(ReadU32(br, Val(0), BitsOffset(4, 1), BitsOffset(6, 9), BitsOffset(8, 41)));
    for (SqueezeStep& q : t->steps) {
      q.horizontal = br.ReadBool();
      q.in_place = br.ReadBool();
      q.begin_c = ReadU32(br, Bits(3), BitsOffset(6, 8), BitsOffset(10, 72), BitsOffset(13, 1096));
      q.num_c = ReadU32(br, Val(1), Val(2), Val(3), BitsOffset(4, 4));
    }
  }
}

// ---- Squeeze (a Haar-like split of a channel into pair averages and residuals, squeeze.cc). One strided line routine
// serves both directions: `avg` holds na averages, `res` nr residuals (nr = na or na - 1), `out` receives na + nr samples.
// Residual i is (first - second of pair i) minus the tendency predicted from the sample before the pair, this pair's
// average and the next one (squeeze.h:54-77); the average rounds towards the first sample.
static inline int64_t SqueezeTendency(int64_t before, int64_t avg, int64_t next) {
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
static inline void UnsqueezeLine(const int32_t* avg, ptrdiff_t sa, size_t na, const int32_t* res, ptrdiff_t sr, size_t nr, int32_t* out,
                                 ptrdiff_t so) {
  int64_t before = 0;
  for (size_t i = 0; i < nr; i++) {
    const int64_t a = avg[ptrdiff_t(i) * sa], next = i + 1 < na ? avg[ptrdiff_t(i + 1) * sa] : a;
    if (i == 0) before = a;
    const int64_t diff = int64_t(res[ptrdiff_t(i) * sr]) + SqueezeTendency(before, a, next);
    const int64_t first = a + diff / 2, second = first - diff;
    out[ptrdiff_t(2 * i) * so] = int32_t(first);
    out[ptrdiff_t(2 * i + 1) * so] = int32_t(second);
    before = second;
  }
  if (na > nr) out[ptrdiff_t(2 * nr) * so] = avg[ptrdiff_t(na - 1) * sa];
}
// Default step list (squeeze.cc:387-444): chroma-like channel pairs first, then alternate directions down to 8x8.
static inline void DefaultSqueeze(const std::vector<MChannel>& ch, size_t nb_meta, std::vector<SqueezeStep>* steps) {
  const size_t first = nb_meta, count = ch.size() - nb_meta;
  size_t w = ch[first].w, h = ch[first].h;
  steps->clear();
  auto push = [&](bool horizontal, bool in_place, size_t begin, size_t num) {
    SqueezeStep q;
    q.horizontal = horizontal;
    q.in_place = in_place;
    q.begin_c = uint32_t(begin);
    q.num_c = uint32_t(num);
    steps->push_back(q);
  };
  if (count > 2 && ch[first + 1].w == w && ch[first + 1].h == h) {
    push(true, false, first + 1, 2);
    push(false, false, first + 1, 2);
  }
  if (w <= h && h > 8) {  // tall (or square) images start with a vertical step
    push(false, true, first, count);
    h = (h + 1) / 2;
  }
  while (w > 8 || h > 8) {
    if (w > 8) {
      push(true, true, first, count);
      w = (w + 1) / 2;
    }
    if (h > 8) {
      push(false, true, first, count);
      h = (h + 1) / 2;
    }
  }
}


static inline void InvRct(MImage* img, const MTransform& t) {
  size_t m = t.begin_c;
  JXH_CHECK(m + 2 < img->ch.size(), "RCT: channel range");
  MChannel &a = img->ch[m], &b = img->ch[m + 1], &c = img->ch[m + 2];
  JXH_CHECK(a.w == b.w && a.w == c.w && a.h == b.h && a.h == c.h, "RCT: channel sizes differ");
  if (t.rct_type == 0) return;
  int perm = t.rct_type / 7, custom = t.rct_type % 7;
  int second = custom >> 1, third = custom & 1;
  size_t n = a.w * a.h;
  std::vector<int32_t> o0(n), o1(n), o2(n);
  for (size_t i = 0; i < n; i++) {
    int32_t x0 = a.d[i], x1 = b.d[i], x2 = c.d[i];
    if (custom == 6) {
      int32_t tmp = int32_t(uint32_t(x0) - uint32_t(x2 >> 1));
      int32_t G = int32_t(uint32_t(x2) + uint32_t(tmp));
      int32_t B = int32_t(uint32_t(tmp) - uint32_t(x1 >> 1));
      int32_t R = int32_t(uint32_t(B) + uint32_t(x1));
      o0[i] = R; o1[i] = G; o2[i] = B;
    } else {
      if (third) x2 = int32_t(uint32_t(x2) + uint32_t(x0));
      if (second == 1) x1 = int32_t(uint32_t(x1) + uint32_t(x0));
      else if (second == 2) x1 = int32_t(uint32_t(x1) + uint32_t(int32_t(uint32_t(x0) + uint32_t(x2)) >> 1));
      o0[i] = x0; o1[i] = x1; o2[i] = x2;
    }
  }
  img->ch[m + (perm % 3)].d = o0;
  img->ch[m + ((perm + 1 + perm / 3) % 3)].d = o1;
  img->ch[m + ((perm + 2 - perm / 3) % 3)].d = o2;
}

static inline void CheckEqualChannels(const MImage& img, size_t c1, size_t c2) {
  JXH_CHECK(c1 <= c2 && c2 < img.ch.size(), "transform: invalid channel range");
  JXH_CHECK(!(c1 < img.nb_meta && c2 >= img.nb_meta), "transform: range spans meta and non-meta channels");
  for (size_t c = c1 + 1; c <= c2; c++)
    JXH_CHECK(img.ch[c].w == img.ch[c1].w && img.ch[c].h == img.ch[c1].h && img.ch[c].hshift == img.ch[c1].hshift &&
                   img.ch[c].vshift == img.ch[c1].vshift,
               "transform: channels differ in size");
}

// Palette (modular/transform/palette.cc:26-202, palette.h:25-140)
static inline void MetaPalette(MImage* img, const MTransform& t) {
  size_t begin_c = t.begin_c, end_c = t.begin_c + t.num_c - 1, nb = t.num_c;
  CheckEqualChannels(*img, begin_c, end_c);
  if (begin_c >= img->nb_meta) {
    img->nb_meta++;
  } else {
    JXH_CHECK(end_c < img->nb_meta, "palette: bad meta channel range");
    img->nb_meta += 2 - nb;
  }
  img->ch.erase(img->ch.begin() + begin_c + 1, img->ch.begin() + end_c + 1);
  MChannel pch(t.nb_colors + t.nb_deltas, nb, -1, -1);
  img->ch.insert(img->ch.begin(), pch);
}

static inline int32_t PaletteValue(const MChannel& pal, int index, size_t c, int bit_depth) {
  static const int16_t kDelta[72][3] = {
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
  const int palette_size = int(pal.w);
  if (index < 0) {
    if (c >= 3) return 0;
    index = -(index + 1);
    index %= 1 + 2 * (72 - 1);
    int32_t r = kDelta[(index + 1) >> 1][c] * ((index & 1) ? 1 : -1);
    if (bit_depth > 8) r *= int32_t(1) << (bit_depth - 8);
    return r;
  } else if (palette_size <= index && index < palette_size + 64) {
    if (c >= 3) return 0;
    index -= palette_size;
    index >>= c * 2;
    return int32_t((uint64_t(index % 4) * ((uint64_t(1) << bit_depth) - 1)) >> 2) + (1 << std::max(0, bit_depth - 3));
  } else if (palette_size + 64 <= index) {
    if (c >= 3) return 0;
    index -= palette_size + 64;
    if (c == 1) index /= 5;
    if (c == 2) index /= 25;
    return int32_t((uint64_t(index % 5) * ((uint64_t(1) << bit_depth) - 1)) >> 2);
  }
  return pal.Row(c)[index];
}

static inline void InvPalette(MImage* img, const MTransform& t) {
  JXH_CHECK(img->nb_meta >= 1, "palette transform without palette");
  const int nb = int(img->ch[0].h);
  const size_t c0 = t.begin_c + 1;
  JXH_CHECK(c0 < img->ch.size() && nb >= 1, "palette: channel out of range");
  const size_t w = img->ch[c0].w, h = img->ch[c0].h;
  for (int i = 1; i < nb; i++) img->ch.insert(img->ch.begin() + c0 + 1, MChannel(w, h, img->ch[c0].hshift, img->ch[c0].vshift));
  const MChannel pal = img->ch[0];
  const int bit_depth = std::min(img->bitdepth, 24);
  if (w != 0) {
    if (t.nb_deltas == 0 && t.predictor == 0) {
      const std::vector<int32_t> idx = img->ch[c0].d;
      for (int c = 0; c < nb; c++) {
        int32_t* out = img->ch[c0 + c].d.data();
        for (size_t i = 0; i < w * h; i++) {
          int index = idx[i];
          if (nb == 1) index = std::max(0, std::min(index, int(pal.w) - 1));
          out[i] = PaletteValue(pal, index, c, bit_depth);
        }
      }
    } else {
      const std::vector<int32_t> idx = img->ch[c0].d;
      for (int c = 0; c < nb; c++) {
        MChannel& ch = img->ch[c0 + c];
        WpState wp(img->wp, ch.w);
        const ptrdiff_t stride = ptrdiff_t(ch.w);
        for (size_t y = 0; y < h; y++) {
          int32_t* p = ch.Row(y);
          for (size_t x = 0; x < w; x++) {
            int index = idx[y * w + x];
            int32_t entry = PaletteValue(pal, index, c, bit_depth);
            const int32_t* pp = p + x;
            int64_t left = x ? pp[-1] : (y ? pp[-stride] : 0);
            int64_t top = y ? pp[-stride] : left;
            int64_t topleft = (x && y) ? pp[-1 - stride] : left;
            int64_t topright = (x + 1 < w && y) ? pp[1 - stride] : top;
            int64_t leftleft = x > 1 ? pp[-2] : left;
            int64_t toptop = y > 1 ? pp[-2 * stride] : top;
            int64_t toprightright = (x + 2 < w && y) ? pp[2 - stride] : topright;
            int64_t wp_pred = 0;
            if (t.predictor == 6) wp_pred = wp.Predict(x, y, w, top, left, topright, topleft, toptop, nullptr);
            int64_t val = entry;
            if (index < int(t.nb_deltas))
              val += PredictOne(t.predictor, left, top, toptop, topleft, topright, leftleft, toprightright, wp_pred);
            p[x] = int32_t(val);
            if (t.predictor == 6) wp.Update(p[x], x, y, w);
          }
        }
      }
    }
  }
  if (c0 >= img->nb_meta) {
    img->nb_meta--;
  } else {
    img->nb_meta -= 2 - nb;
  }
  img->ch.erase(img->ch.begin());
}

// Channel list after the squeeze steps (squeeze.cc:456-517): every squeezed channel halves along the step's direction
// and gains a residual channel, right behind the range (in place) or at the end of the list.
static inline void SqueezeMeta(MImage* img, MTransform* t) {
  if (t->steps.empty()) DefaultSqueeze(img->ch, img->nb_meta, &t->steps);
  for (const SqueezeStep& q : t->steps) {
    const size_t n = img->ch.size(), b = q.begin_c, e = size_t(q.begin_c) + q.num_c;  // [b, e)
    JXH_CHECK(q.num_c >= 1 && b < n && e <= n, "squeeze: invalid channel range");
    if (b < img->nb_meta) {
      JXH_CHECK(e <= img->nb_meta && q.in_place, "squeeze: bad meta channel range");
      img->nb_meta += q.num_c;
    }
    size_t at = q.in_place ? e : n;
    for (size_t c = b; c < e; c++, at++) {
      MChannel& src = img->ch[c];
      JXH_CHECK(src.w && src.h, "squeeze of an empty channel");
      JXH_CHECK(src.hshift <= 30 && src.vshift <= 30, "squeeze: too many steps");
      size_t rw = src.w, rh = src.h;
      if (q.horizontal) {
        src.w = (src.w + 1) / 2;
        rw -= src.w;
        if (src.hshift >= 0) src.hshift++;
      } else {
        src.h = (src.h + 1) / 2;
        rh -= src.h;
        if (src.vshift >= 0) src.vshift++;
      }
      src.d.assign(src.w * src.h, 0);
      MChannel residual(rw, rh, src.hshift, src.vshift);
      img->ch.insert(img->ch.begin() + at, residual);
    }
  }
}
static inline void SqueezeInverse(MImage* img, const MTransform& t) {
  for (size_t k = t.steps.size(); k-- > 0;) {
    const SqueezeStep& q = t.steps[k];
    const size_t b = q.begin_c, e = size_t(q.begin_c) + q.num_c;
    JXH_CHECK(e <= img->ch.size(), "squeeze: invalid channel range");
    const size_t first_res = q.in_place ? e : img->ch.size() + b - e;
    JXH_CHECK(first_res >= e && first_res + q.num_c <= img->ch.size(), "squeeze: residual channels missing");
    if (b < img->nb_meta) {
      JXH_CHECK(img->nb_meta >= q.num_c, "squeeze: meta channel bookkeeping");
      img->nb_meta -= q.num_c;
    }
    for (size_t c = b; c < e; c++) {
      MChannel& a = img->ch[c];
      const MChannel& r = img->ch[first_res + (c - b)];
      if (q.horizontal) {
        JXH_CHECK(a.w == (a.w + r.w + 1) / 2 && a.h == r.h, "squeeze: channel sizes do not match");
        MChannel out(a.w + r.w, a.h, a.hshift - 1, a.vshift);
        if (r.w == 0) out.d = a.d;
        else
          for (size_t y = 0; y < a.h; y++) UnsqueezeLine(a.Row(y), 1, a.w, r.Row(y), 1, r.w, out.Row(y), 1);
        img->ch[c] = out;
      } else {
        JXH_CHECK(a.h == (a.h + r.h + 1) / 2 && a.w == r.w, "squeeze: channel sizes do not match");
        MChannel out(a.w, a.h + r.h, a.hshift, a.vshift - 1);
        if (r.h == 0) out.d = a.d;
        else
          for (size_t x = 0; x < a.w; x++)
            UnsqueezeLine(a.d.data() + x, ptrdiff_t(a.w), a.h, r.d.data() + x, ptrdiff_t(r.w), r.h, out.d.data() + x, ptrdiff_t(out.w));
        img->ch[c] = out;
      }
    }
    img->ch.erase(img->ch.begin() + first_res, img->ch.begin() + first_res + q.num_c);
  }
}

static inline void MetaApply(MImage* img, MTransform& t) {
  if (t.id == 0) CheckEqualChannels(*img, t.begin_c, t.begin_c + 2);
  else if (t.id == 1) MetaPalette(img, t);
  else SqueezeMeta(img, &t);
}
static inline void InverseTransform(MImage* img, const MTransform& t) {
  if (t.id == 0) InvRct(img, t);
  else if (t.id == 1) InvPalette(img, t);
  else SqueezeInverse(img, t);
}

// Decodes one Modular stream into `img` (whose channels are pre-sized) and undoes its transforms.
// max_chan_size: channels larger than this (non-meta) are left for later streams.
static inline void ModularDecode(BitReader& br, MImage* img, int stream_id, const MGlobal* global,
                                 size_t max_chan_size = size_t(1) << 30, bool undo_transforms = true) {
  if (img->ch.empty()) return;
  // GroupHeader
  bool use_global_tree = br.ReadBool();
  if (!br.ReadBool()) {  // weighted predictor header not default
    img->wp.p1C = int32_t(br.Read(5));
    img->wp.p2C = int32_t(br.Read(5));
    img->wp.p3Ca = int32_t(br.Read(5));
    img->wp.p3Cb = int32_t(br.Read(5));
    img->wp.p3Cc = int32_t(br.Read(5));
    img->wp.p3Cd = int32_t(br.Read(5));
    img->wp.p3Ce = int32_t(br.Read(5));
    for (int i = 0; i < 4; i++) img->wp.w[i] = uint32_t(br.Read(4));
  }
  uint32_t nt = ReadU32(br, Val(0), Val(1), BitsOffset(4, 2), BitsOffset(8, 18));
  img->transforms.resize(nt);
  for (auto& t : img->transforms) ReadTransform(br, &t);
  for (auto& t : img->transforms) MetaApply(img, t);
  size_t nch = img->ch.size();
  size_t dist_mult = 0, num = 0;
  for (size_t i = 0; i < nch; i++) {
    MChannel& c = img->ch[i];
    if (i >= img->nb_meta && (c.w > max_chan_size || c.h > max_chan_size)) break;
    if (!c.w || !c.h) continue;
    dist_mult = std::max(dist_mult, c.w);
    num++;
  }
  if (num == 0) return;
  MTree local_tree;
  EntropyCode local_code;
  const MTree* tree;
  const EntropyCode* code;
  if (!use_global_tree) {
    uint64_t limit = 1024;
    for (size_t i = 0; i < nch; i++) {
      MChannel& c = img->ch[i];
      if (i >= img->nb_meta && (c.w > max_chan_size || c.h > max_chan_size)) break;
      limit += uint64_t(c.w) * c.h;
    }
    DecodeTree(br, &local_tree, std::min<uint64_t>(limit, 1 << 20));
    DecodeHistograms(br, (local_tree.size() + 1) / 2, &local_code);
    tree = &local_tree;
    code = &local_code;
  } else {
    JXH_CHECK(global && global->have && !global->tree.empty(), "global tree requested but absent");
    tree = &global->tree;
    code = &global->code;
  }
  SymbolReader rd(code, &br, dist_mult);
  for (size_t i = 0; i < nch; i++) {
    MChannel& c = img->ch[i];
    if (i >= img->nb_meta && (c.w > max_chan_size || c.h > max_chan_size)) break;
    if (!c.w || !c.h) continue;
    DecodeChannel(br, rd, *code, *tree, img->wp, int(i), stream_id, img);
    JXH_CHECK(!br.Overread(), "modular stream truncated");
  }
  JXH_CHECK(rd.FinalStateOk(), "modular stream: bad ANS final state");
  if (undo_transforms) {
    for (size_t i = img->transforms.size(); i-- > 0;) InverseTransform(img, img->transforms[i]);
    img->transforms.clear();
  }
}

}  // namespace jxh
#endif  // JXH_MODULAR_H_
