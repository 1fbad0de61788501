// libjxl_amd synthetic-stream encoder: produces valid JPEG XL VarDCT codestreams for tests and benchmarks.
//
// There are no .jxl fixtures in the reference tree and the reference encoder cannot be built in this image, so the
// decode path is exercised with streams written here. Two modes:
//   * image mode  : RGB8 -> XYB -> AC-strategy choice (DCT family 8..64) -> quantisation at a butteraugli-style
//                   distance (same initial parameters as the reference: kAcQuant/kDcQuant, enc_adaptive_quantization.cc
//                   :835-837,1250-1262; quantizer.cc:45-76) -> rANS-coded sections.
//   * random mode : random varblock tilings over all 27 strategies with random quantised coefficients, DC, quant
//                   field, colour-correlation map and sharpness (exercises every decoder branch; not a real image).
// The bitstream layout written here mirrors what the decoder reads: reference lib/jxl/headers.cc:129-152,
// image_metadata.cc:283-356, frame_header.cc:215-439, loop_filter.cc:20-100, toc.cc:29-73, dec_frame.cc:269-434,
// dec_modular.cc:427-562, dec_group.cc:469-639, dec_ans.cc:58-376, dec_context_map.cc:48-95, dec_ma.cc:107-159.
// This is a growth seed for the "VarDCT encoder forward path" row of SURVEY.md §8f, not a quality-tuned encoder.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/jxl_amd_hip.h"  // JxlHipEncDesc: the forward path's descriptor (no link dependency)

#include "../host/jxh_bits.h"
#include "../host/jxh_entropy.h"
#include "../host/jxh_modular.h"
#include "../host/jxh_vardct.h"

namespace jxe {
using jxh::BitWriter;
using jxh::CeilLog2;
using jxh::DivCeil;
using jxh::FloorLog2;

struct Rng {  // xorshift128+
  uint64_t s[2];
  explicit Rng(uint64_t seed) {
    s[0] = seed * 0x9E3779B97F4A7C15ull + 1;
    s[1] = (seed ^ 0xD1B54A32D192ED03ull) * 0xBF58476D1CE4E5B9ull + 7;
    for (int i = 0; i < 8; i++) Next();
  }
  uint64_t Next() {
    uint64_t a = s[0], b = s[1];
    s[0] = b;
    a ^= a << 23;
    s[1] = a ^ b ^ (a >> 17) ^ (b >> 26);
    return s[1] + b;
  }
  uint32_t Below(uint32_t n) { return uint32_t((Next() >> 11) % n); }
  float Uniform() { return float((Next() >> 40) * (1.0 / 16777216.0)); }
};

struct Token {
  uint32_t ctx;
  uint32_t value;
};

static inline uint32_t PackSigned(int32_t v) { return v >= 0 ? uint32_t(v) * 2 : uint32_t(-(v + 1)) * 2 + 1; }

static inline void HybridEncode(const jxh::HybridCfg& c, uint32_t value, uint32_t* token, uint32_t* nbits, uint32_t* bits) {
  if (value < c.split_token) {
    *token = value;
    *nbits = 0;
    *bits = 0;
    return;
  }
  uint32_t n = uint32_t(FloorLog2(value));
  uint32_t m = value - (1u << n);
  *token = c.split_token + ((n - c.split_exp) << (c.msb + c.lsb)) + ((m >> (n - c.msb)) << c.lsb) + (m & ((1u << c.lsb) - 1));
  *nbits = n - c.msb - c.lsb;
  *bits = (value >> c.lsb) & ((1u << *nbits) - 1);
}

static inline void Symbolize(const jxh::HybridCfg& cfg, const jxh::HybridCfg& len_cfg, uint32_t value, uint32_t* token, uint32_t* nbits,
                             uint32_t* bits) {
  if (value & 0x80000000u) {  // LZ77 length token (kLzLenFlag)
    HybridEncode(len_cfg, value & 0x7FFFFFFFu, token, nbits, bits);
    *token += 224;  // kLzMinSymbol
    return;
  }
  HybridEncode(cfg, value, token, nbits, bits);
}

static void WriteVarLenU8(BitWriter& bw, uint32_t n) {
  if (n == 0) {
    bw.Write(1, 0);
    return;
  }
  bw.Write(1, 1);
  uint32_t nb = uint32_t(FloorLog2(n));
  bw.Write(3, nb);
  bw.Write(nb, n - (1u << nb));
}

// Writes a histogram that sums to 4096 (dec_ans.cc:58-191 reads it back).
static void WriteHistogram(BitWriter& bw, const std::vector<int32_t>& counts) {
  std::vector<int> nz;
  for (size_t i = 0; i < counts.size(); i++)
    if (counts[i]) nz.push_back(int(i));
  if (nz.size() <= 2) {
    bw.Write(1, 1);
    bw.Write(1, nz.size() == 2 ? 1 : 0);
    if (nz.empty()) {
      WriteVarLenU8(bw, 0);
      return;
    }
    for (int s : nz) WriteVarLenU8(bw, s);
    if (nz.size() == 2) bw.Write(12, counts[nz[0]]);
    return;
  }
  bw.Write(1, 0);  // not simple
  bw.Write(1, 0);  // not flat
  bw.Write(3, 7);  // shift code: three 1s ...
  bw.Write(3, 6);  // ... then (shift+1) - 8 = 6  => shift = 13 (exact counts)
  size_t length = std::max<size_t>(3, size_t(nz.back()) + 1);
  WriteVarLenU8(bw, uint32_t(length - 3));
  static const uint8_t kLC[14][3] = {{0, 3, 10}, {1, 7, 12}, {2, 3, 7},  {3, 4, 3},  {4, 3, 6},   {5, 3, 8},  {6, 3, 9},
                                     {7, 4, 5},  {9, 4, 4},  {11, 4, 1}, {15, 4, 2}, {17, 5, 0},  {33, 6, 11}, {65, 7, 13}};
  std::vector<int> logc(length, -1);
  int omit_log = -1, omit_pos = -1;
  for (size_t i = 0; i < length; i++) {
    int32_t c = i < counts.size() ? counts[i] : 0;
    int val = c == 0 ? 0 : FloorLog2(uint32_t(c)) + 1;
    logc[i] = val - 1;
    for (int k = 0; k < 14; k++)
      if (kLC[k][2] == val) bw.Write(kLC[k][1], kLC[k][0]);
    if (logc[i] > omit_log) {
      omit_log = logc[i];
      omit_pos = int(i);
    }
  }
  for (size_t i = 0; i < length; i++) {
    if (int(i) == omit_pos || logc[i] <= 0) continue;
    bw.Write(unsigned(logc[i]), uint32_t(counts[i]) - (1u << logc[i]));
  }
}

static void WriteHybridCfg(BitWriter& bw, const jxh::HybridCfg& c, int log_alpha) {
  bw.Write(CeilLog2(log_alpha + 1), c.split_exp);
  if (int(c.split_exp) != log_alpha) {
    bw.Write(CeilLog2(c.split_exp + 1), c.msb);
    bw.Write(CeilLog2(c.split_exp - c.msb + 1), c.lsb);
  }
}

// Normalises counts to sum 4096 with every used symbol >= 1.
static std::vector<int32_t> Normalize(const std::vector<uint32_t>& h) {
  uint64_t total = 0;
  for (uint32_t c : h) total += c;
  std::vector<int32_t> n(h.size(), 0);
  if (total == 0) {
    n.assign(1, 4096);
    return n;
  }
  int64_t sum = 0;
  size_t big = 0;
  for (size_t i = 0; i < h.size(); i++) {
    if (!h[i]) continue;
    int64_t v = int64_t(double(h[i]) * 4096.0 / double(total) + 0.5);
    if (v < 1) v = 1;
    n[i] = int32_t(v);
    sum += v;
    if (h[i] > h[big] || !h[big]) big = i;
  }
  int64_t diff = 4096 - sum;
  while (diff != 0) {
    // adjust the largest entries, keeping everything >= 1
    size_t best = big;
    for (size_t i = 0; i < n.size(); i++)
      if (n[i] > n[best]) best = i;
    int64_t step = diff > 0 ? diff : std::max<int64_t>(diff, -(int64_t(n[best]) - 1));
    if (step == 0) break;
    n[best] += int32_t(step);
    diff -= step;
  }
  while (!n.empty() && n.back() == 0) n.pop_back();
  return n;
}

// Special tokens of an LZ77-enabled stream (dec_ans.h:288-353): a copy is a length token in the context of the value it
// replaces (value = kLzLenFlag | (length - min_length)) followed by a distance token in the extra last context
// (value = distance - 1; no special distances: AC streams have no distance multiplier).
static const uint32_t kLzLenFlag = 0x80000000u;
static const uint32_t kLzMinSymbol = 224, kLzMinLength = 3;

struct EncCode {
  size_t num_ctx = 0;  // (including the distance context of an LZ77 stream)
  std::vector<uint8_t> ctx_map;
  size_t num_clusters = 1;
  jxh::HybridCfg cfg;
  int log_alpha = 5;
  bool use_prefix = false, lz77 = false;
  jxh::HybridCfg lz_len_cfg;
  std::vector<std::vector<uint8_t>> plen;          // prefix codes: [cluster][symbol] code length (0 = unused)
  std::vector<std::vector<uint16_t>> pbits;        // [cluster][symbol] code, bit-reversed for LSB-first writing
  std::vector<std::vector<int32_t>> norm;          // [cluster][symbol]
  std::vector<std::vector<uint32_t>> rev_start;    // [cluster][symbol] -> start in rev
  std::vector<std::vector<uint16_t>> rev;          // [cluster] slot of (symbol, offset)
};

static void BuildReverseMaps(EncCode* code) {
  code->rev.assign(code->num_clusters, {});
  code->rev_start.assign(code->num_clusters, {});
  std::vector<jxh::AliasEntry> alias(size_t(1) << code->log_alpha);
  const int log_entry = 12 - code->log_alpha;
  for (size_t c = 0; c < code->num_clusters; c++) {
    jxh::InitAliasTable(code->norm[c], code->log_alpha, alias.data());
    std::vector<int32_t> dist = code->norm[c];
    if (dist.empty()) dist.push_back(4096);
    std::vector<uint32_t>& start = code->rev_start[c];
    start.assign(dist.size() + 1, 0);
    for (size_t s = 0; s < dist.size(); s++) start[s + 1] = start[s] + uint32_t(dist[s]);
    code->rev[c].assign(4096, 0);
    for (uint32_t slot = 0; slot < 4096; slot++) {
      const jxh::AliasEntry& e = alias[slot >> log_entry];
      uint32_t pos = slot & ((1u << log_entry) - 1);
      uint32_t sym, off;
      if (pos >= e.cutoff) {
        sym = e.right_value;
        off = e.offsets1 + pos;
      } else {
        sym = slot >> log_entry;
        off = pos;
      }
      code->rev[c][start[sym] + off] = uint16_t(slot);
    }
  }
}

// OpenMP threads of the writer's loops: a container's CPU share can be far below the CPUs it sees (a 16-CPU quota of 256:
// 256 threads then spend their time in the barriers). The smaller of the visible CPUs, the cgroup quota and 64, unless
// OMP_NUM_THREADS or jxlenc_set_threads says otherwise; applied on entry by every encode call (the setting is per thread).
static int g_enc_threads = 0;
static int DefaultThreads() {
  int n = omp_get_num_procs();
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    long long quota = 0, period = 0;
    if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) n = std::min<long long>(n, (quota + period - 1) / period);
    fclose(f);
  } else if (FILE* q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // (cgroup v1)
    long long quota = 0, period = 100000;
    const bool ok = fscanf(q, "%lld", &quota) == 1;
    fclose(q);
    if (FILE* pf = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (fscanf(pf, "%lld", &period) != 1) period = 100000;
      fclose(pf);
    }
    if (ok && quota > 0 && period > 0) n = std::min<long long>(n, (quota + period - 1) / period);
  }
  return std::max(1, std::min(n, 64));
}
static void UseThreads() {
  static const int def = getenv("OMP_NUM_THREADS") ? 0 : DefaultThreads();
  const int n = g_enc_threads > 0 ? g_enc_threads : def;
  if (n > 0) omp_set_num_threads(n);
}

static double NowSeconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Builds clustered, normalised histograms for a token set over `num_ctx` contexts.
static void BuildPrefixCodes(EncCode* code, const std::vector<std::vector<double>>& csum);

// mode: bit 0 = prefix codes instead of ANS, bit 1 = the streams carry LZ77 tokens (num_ctx then includes the distance context)
static void BuildCode(const std::vector<const std::vector<Token>*>& streams, size_t num_ctx, size_t max_clusters,
                      jxh::HybridCfg cfg, EncCode* code, int mode = 0) {
  code->num_ctx = num_ctx;
  code->cfg = cfg;
  code->use_prefix = (mode & 1) != 0;
  code->lz77 = (mode & 2) != 0;
  code->lz_len_cfg.split_exp = 4;
  code->lz_len_cfg.split_token = 16;
  code->lz_len_cfg.msb = code->lz_len_cfg.lsb = 0;
  uint32_t max_tok = 0;
  std::vector<std::vector<uint32_t>> hist(num_ctx);
  auto count = [&](const std::vector<Token>& st, std::vector<std::vector<uint32_t>>& into, uint32_t* mx) {
    for (const Token& t : st) {
      uint32_t tok, nb, bits;
      Symbolize(cfg, code->lz_len_cfg, t.value, &tok, &nb, &bits);
      *mx = std::max(*mx, tok);
      auto& h = into[t.ctx];
      if (h.size() <= tok) h.resize(tok + 1, 0);
      h[tok]++;
    }
  };
  size_t total_tokens = 0;
  for (const auto* st : streams) total_tokens += st->size();
  if (streams.size() >= 8 && total_tokens >= (size_t(1) << 18)) {  // a large frame's AC tokens: counted per thread, then added up
#pragma omp parallel
    {
      std::vector<std::vector<uint32_t>> mine(num_ctx);
      uint32_t mx = 0;
#pragma omp for schedule(dynamic, 4) nowait
      for (size_t i = 0; i < streams.size(); i++) count(*streams[i], mine, &mx);
#pragma omp critical
      {
        max_tok = std::max(max_tok, mx);
        for (size_t c = 0; c < num_ctx; c++) {
          if (hist[c].size() < mine[c].size()) hist[c].resize(mine[c].size(), 0);
          for (size_t k = 0; k < mine[c].size(); k++) hist[c][k] += mine[c][k];
        }
      }
    }
  } else {
    for (const auto* st : streams) count(*st, hist, &max_tok);
  }
  code->log_alpha = code->use_prefix ? 15 : std::max(5, CeilLog2(max_tok + 1));
  if (!code->use_prefix && code->log_alpha > 8) abort();
  const size_t A = max_tok + 1;
  // greedy clustering, largest contexts first
  std::vector<size_t> order;
  std::vector<uint64_t> total(num_ctx, 0);
  for (size_t c = 0; c < num_ctx; c++) {
    for (uint32_t v : hist[c]) total[c] += v;
    if (total[c]) order.push_back(c);
  }
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return total[a] != total[b] ? total[a] > total[b] : a < b; });
  std::vector<std::vector<double>> csum;  // cluster histograms
  std::vector<double> ctot;
  code->ctx_map.assign(num_ctx, 0);
  auto cost_in = [&](const std::vector<uint32_t>& h, const std::vector<double>& cs, double ct) {
    double bits = 0;
    for (size_t s = 0; s < h.size(); s++)
      if (h[s]) bits -= h[s] * std::log2((cs[s] + 0.5) / (ct + 0.5 * A));
    return bits;
  };
  for (size_t c : order) {
    std::vector<double> self(A, 0.0);
    for (size_t s = 0; s < hist[c].size(); s++) self[s] = hist[c][s];
    double self_cost = cost_in(hist[c], self, double(total[c]));
    int best = -1;
    double best_cost = 1e300;
    for (size_t k = 0; k < csum.size(); k++) {
      double cst = cost_in(hist[c], csum[k], ctot[k]);
      if (cst < best_cost) {
        best_cost = cst;
        best = int(k);
      }
    }
    bool merge = best >= 0 && (csum.size() >= max_clusters || best_cost < self_cost * 1.08 + 64.0);
    if (!merge) {
      csum.push_back(self);
      ctot.push_back(double(total[c]));
      code->ctx_map[c] = uint8_t(csum.size() - 1);
    } else {
      for (size_t s = 0; s < hist[c].size(); s++) csum[best][s] += hist[c][s];
      ctot[best] += double(total[c]);
      code->ctx_map[c] = uint8_t(best);
    }
  }
  if (csum.empty()) {
    csum.push_back(std::vector<double>(A, 0.0));
    ctot.push_back(0);
  }
  code->num_clusters = csum.size();
  if (code->use_prefix) {
    BuildPrefixCodes(code, csum);
    return;
  }
  code->norm.resize(code->num_clusters);
  for (size_t k = 0; k < code->num_clusters; k++) {
    std::vector<uint32_t> h(A);
    for (size_t s = 0; s < A; s++) h[s] = uint32_t(csum[k][s]);
    code->norm[k] = Normalize(h);
  }
  BuildReverseMaps(code);
}

// rANS-encodes tokens (all contexts via code->ctx_map); writes the 32-bit initial state first.
// Length-limited Huffman code lengths (limit 15: dec_huffman.cc), then canonical codes (huffman_table.cc:27-161).
static void HuffmanLengths(std::vector<double> counts, int limit, std::vector<uint8_t>* lens) {
  const size_t n = counts.size();
  for (;;) {
    lens->assign(n, 0);
    struct Node { double w; int l, r; };
    std::vector<Node> nodes;
    std::vector<int> live;
    for (size_t i = 0; i < n; i++)
      if (counts[i] > 0) {
        nodes.push_back({counts[i], -1, int(i)});
        live.push_back(int(nodes.size()) - 1);
      }
    if (live.size() <= 1) {
      if (live.size() == 1) (*lens)[size_t(nodes[0].r)] = 1;  // a single symbol: zero-bit code (any non-zero length marks it)
      return;
    }
    while (live.size() > 1) {
      std::sort(live.begin(), live.end(), [&](int a, int b) { return nodes[a].w != nodes[b].w ? nodes[a].w > nodes[b].w : a < b; });
      const int a = live.back();
      live.pop_back();
      const int b = live.back();
      live.pop_back();
      nodes.push_back({nodes[a].w + nodes[b].w, a, b});
      live.push_back(int(nodes.size()) - 1);
    }
    int max_len = 0;
    std::vector<std::pair<int, int>> stack{{live[0], 0}};
    while (!stack.empty()) {
      const auto [id, depth] = stack.back();
      stack.pop_back();
      if (nodes[id].l < 0) {
        (*lens)[size_t(nodes[id].r)] = uint8_t(depth);
        max_len = std::max(max_len, depth);
      } else {
        stack.push_back({nodes[id].l, depth + 1});
        stack.push_back({nodes[id].r, depth + 1});
      }
    }
    if (max_len <= limit) return;
    for (double& c : counts)  // flatten and retry
      if (c > 0) c = std::floor(c / 2) + 1;
  }
}
static void CanonicalCodes(const std::vector<uint8_t>& lens, std::vector<uint16_t>* rev_bits) {
  rev_bits->assign(lens.size(), 0);
  uint32_t code = 0;
  for (int l = 1; l <= 15; l++) {
    for (size_t s = 0; s < lens.size(); s++) {
      if (lens[s] != l) continue;
      uint32_t rev = 0;
      for (int b = 0; b < l; b++) rev |= ((code >> b) & 1) << (l - 1 - b);
      (*rev_bits)[s] = uint16_t(rev);
      code++;
    }
    code <<= 1;
  }
}
static void BuildPrefixCodes(EncCode* code, const std::vector<std::vector<double>>& csum) {
  code->plen.resize(code->num_clusters);
  code->pbits.resize(code->num_clusters);
  for (size_t k = 0; k < code->num_clusters; k++) {
    std::vector<double> c = csum[k];
    while (!c.empty() && c.back() == 0) c.pop_back();
    if (c.empty()) c.push_back(1);
    HuffmanLengths(c, 15, &code->plen[k]);
    CanonicalCodes(code->plen[k], &code->pbits[k]);
  }
}
// One prefix code as HuffmanDecodingData::ReadFromBitStream expects it (dec_huffman.cc:179-245; the "complex" form with
// a code-length code, or the 1-symbol simple form).
static void WritePrefixCode(BitWriter& bw, const std::vector<uint8_t>& lens) {
  const size_t alphabet = lens.size();
  std::vector<size_t> used;
  for (size_t i = 0; i < alphabet; i++)
    if (lens[i]) used.push_back(i);
  if (used.size() == 1) {
    bw.Write(2, 1);  // simple code
    bw.Write(2, 0);  // one symbol
    bw.Write(alphabet > 1 ? FloorLog2(uint32_t(alphabet - 1)) + 1 : 0, uint32_t(used[0]));
    return;
  }
  bw.Write(2, 0);  // complex code, no skipped code-length-code entries
  const size_t last = used.back();
  std::vector<double> clh(18, 0.0);
  for (size_t i = 0; i <= last; i++) clh[lens[i]] += 1;
  std::vector<uint8_t> cl_lens;
  HuffmanLengths(clh, 5, &cl_lens);
  std::vector<uint16_t> cl_bits;
  CanonicalCodes(cl_lens, &cl_bits);
  static const uint8_t kOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
  // fixed code of the code-length-code lengths: value -> (bit pattern LSB first, length)
  static const uint8_t kPat[6][2] = {{0, 2}, {7, 4}, {3, 3}, {2, 2}, {1, 2}, {15, 4}};
  int space = 32, num_codes = 0;
  for (size_t i = 0; i < 18 && space > 0; i++) {
    const int v = cl_lens[kOrder[i]];
    bw.Write(kPat[v][1], kPat[v][0]);
    if (v) {
      space -= 32 >> v;
      num_codes++;
    }
  }
  if (!(num_codes == 1 || space == 0)) abort();
  for (size_t i = 0; i <= last; i++)
    if (num_codes > 1) bw.Write(cl_lens[lens[i]], cl_bits[lens[i]]);
}

static void WriteTokens(BitWriter& bw, const Token* tk, size_t n, const EncCode& code) {
  if (code.use_prefix) {  // forward order, no coder state (dec_ans.h:170-197 ReadSymbolHuffWithoutRefill)
    for (size_t i = 0; i < n; i++) {
      uint32_t tok, nbits, bits;
      Symbolize(code.cfg, code.lz_len_cfg, tk[i].value, &tok, &nbits, &bits);
      const size_t k = code.ctx_map[tk[i].ctx];
      if (tok >= code.plen[k].size() || code.plen[k][tok] == 0) abort();
      size_t used = 0;
      for (uint8_t l : code.plen[k]) used += l != 0;
      if (used > 1) bw.Write(code.plen[k][tok], code.pbits[k][tok]);
      bw.Write(nbits, bits);
    }
    return;
  }
  struct Out { uint32_t tok, nbits, bits; uint16_t chunk; bool has_chunk; uint8_t cluster; };
  std::vector<Out> o(n);
  for (size_t i = 0; i < n; i++) {
    Symbolize(code.cfg, code.lz_len_cfg, tk[i].value, &o[i].tok, &o[i].nbits, &o[i].bits);
    o[i].cluster = code.ctx_map[tk[i].ctx];
    o[i].has_chunk = false;
  }
  uint32_t state = jxh::kAnsSignature << 16;
  for (size_t i = n; i-- > 0;) {
    const auto& norm = code.norm[o[i].cluster];
    uint32_t freq = o[i].tok < norm.size() ? uint32_t(norm[o[i].tok]) : 0;
    if (freq == 0) abort();  // token without probability mass: internal error
    if ((state >> (32 - 12)) >= freq) {
      o[i].chunk = uint16_t(state & 0xFFFF);
      o[i].has_chunk = true;
      state >>= 16;
    }
    uint32_t slot = code.rev[o[i].cluster][code.rev_start[o[i].cluster][o[i].tok] + state % freq];
    state = ((state / freq) << 12) + slot;
  }
  bw.Write(32, state);
  for (size_t i = 0; i < n; i++) {
    if (o[i].has_chunk) bw.Write(16, o[i].chunk);
    bw.Write(o[i].nbits, o[i].bits);
  }
}

static void WriteCodeHeader(BitWriter& bw, const EncCode& code);

static void WriteContextMap(BitWriter& bw, const EncCode& code) {
  if (code.num_clusters <= 8) {
    int bits = CeilLog2(code.num_clusters);
    bw.Write(1, 1);
    bw.Write(2, bits);
    if (bits)
      for (uint8_t e : code.ctx_map) bw.Write(bits, e);
    return;
  }
  bw.Write(1, 0);  // not simple
  bw.Write(1, 0);  // no MTF
  std::vector<Token> t(code.ctx_map.size());
  for (size_t i = 0; i < t.size(); i++) t[i] = {0, code.ctx_map[i]};
  EncCode inner;
  jxh::HybridCfg cfg;
  cfg.split_exp = 4; cfg.split_token = 16; cfg.msb = 2; cfg.lsb = 0;
  BuildCode({&t}, 1, 1, cfg, &inner);
  WriteCodeHeader(bw, inner);
  WriteTokens(bw, t.data(), t.size(), inner);
}

// Everything DecodeHistograms() reads (dec_ans.cc:341-376).
static void WriteVarLenU16(BitWriter& bw, uint32_t n) {
  if (n == 0) {
    bw.Write(1, 0);
    return;
  }
  bw.Write(1, 1);
  uint32_t nb = uint32_t(FloorLog2(n));
  bw.Write(4, nb);
  bw.Write(nb, n - (1u << nb));
}
static void WriteCodeHeader(BitWriter& bw, const EncCode& code) {
  if (!code.lz77) {
    bw.Write(1, 0);  // lz77 disabled
  } else {
    bw.Write(1, 1);
    bw.Write(2, 0);  // min_symbol 224
    bw.Write(2, 0);  // min_length 3
    WriteHybridCfg(bw, code.lz_len_cfg, 8);
  }
  if (code.num_ctx > 1) WriteContextMap(bw, code);
  if (code.use_prefix) {
    bw.Write(1, 1);
    for (size_t k = 0; k < code.num_clusters; k++) WriteHybridCfg(bw, code.cfg, 15);
    for (size_t k = 0; k < code.num_clusters; k++) WriteVarLenU16(bw, uint32_t(code.plen[k].size() - 1));
    for (size_t k = 0; k < code.num_clusters; k++)
      if (code.plen[k].size() > 1) WritePrefixCode(bw, code.plen[k]);
    return;
  }
  bw.Write(1, 0);  // ANS, not prefix codes
  bw.Write(2, code.log_alpha - 5);
  for (size_t k = 0; k < code.num_clusters; k++) WriteHybridCfg(bw, code.cfg, code.log_alpha);
  for (size_t k = 0; k < code.num_clusters; k++) WriteHistogram(bw, code.norm[k]);
}

// Replaces runs that repeat the values `distance` tokens back (distance 1, 2 or 3: zero runs and short periodic
// patterns) by LZ77 copies. The copy's length token takes the context of the first value it replaces.
// num_special: 120 in streams with a distance multiplier (Modular: the first 120 distance values are the 2-D "special
// distances", plain distances follow, dec_ans.h:122-145,308-316), 0 in AC coefficient streams.
static void Lz77Pass(std::vector<Token>* tokens, uint32_t dist_ctx, uint32_t num_special = 0) {
  const std::vector<Token>& in = *tokens;
  std::vector<Token> out;
  out.reserve(in.size());
  for (size_t i = 0; i < in.size();) {
    size_t best_len = 0, best_d = 0;
    for (size_t d = 1; d <= 3 && d <= i; d++) {
      size_t l = 0;
      while (i + l < in.size() && in[i + l].value == in[i + l - d].value) l++;
      if (l > best_len) {
        best_len = l;
        best_d = d;
      }
    }
    if (best_len >= 6) {
      out.push_back({in[i].ctx, kLzLenFlag | uint32_t(best_len - kLzMinLength)});
      out.push_back({dist_ctx, uint32_t(best_d - 1) + num_special});
      i += best_len;
    } else {
      out.push_back(in[i++]);
    }
  }
  tokens->swap(out);
}

// IEEE half (fields.cc:550-575): the value must be a normal half-precision number or zero.
static void WriteF16(BitWriter& bw, float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  const uint32_t sign = u >> 31, e32 = (u >> 23) & 0xFF, m32 = u & 0x7FFFFF;
  uint32_t h = 0;
  if (e32 != 0) {
    const int e = int(e32) - 127 + 15;
    if (e <= 0 || e >= 31 || (m32 & 0x1FFF)) abort();  // only exactly representable normal values are used here
    h = (sign << 15) | (uint32_t(e) << 10) | (m32 >> 13);
  }
  bw.Write(16, h);
}

static void WriteU32Sel(BitWriter& bw, uint32_t v, const uint32_t bits[4], const uint32_t offs[4]) {
  for (int s = 0; s < 4; s++) {
    if (bits[s] == 0) {
      if (v == offs[s]) {
        bw.Write(2, s);
        return;
      }
    } else if (v >= offs[s] && uint64_t(v - offs[s]) < (uint64_t(1) << bits[s])) {
      bw.Write(2, s);
      bw.Write(bits[s], v - offs[s]);
      return;
    }
  }
  abort();
}

// ---------------------------------------------------------------- Modular (DC + AC metadata) tokeniser
struct TreeSpec {
  struct Node { int prop; int32_t split; int l, r; uint32_t predictor; };
  std::vector<Node> nodes;  // BFS order
  std::vector<int> leaf_ctx;
};

// Fixed tree: splits on stream kind (property 1) and channel (property 0); see header comment.
static TreeSpec MakeTree(size_t ndc) {
  TreeSpec t;
  // explicit BFS layout
  // 0: p1 > 2*ndc ? 1 : 2
  // 1 (AC meta): p0 > 1 ? 3 : 4(leaf cmap)
  // 2 (DC): p0 > 0 ? 5 : 6(leaf Y)
  // 3: p0 > 2 ? 7(leaf sharpness) : 8(leaf acs/qf)
  // 5: p0 > 1 ? 9(leaf B) : 10(leaf X)
  t.nodes = {{1, int32_t(2 * ndc), 1, 2, 0}, {0, 1, 3, 4, 0}, {0, 0, 5, 6, 0}, {0, 2, 7, 8, 0}, {-1, 0, 0, 0, 5},
             {0, 1, 9, 10, 0},              {-1, 0, 0, 0, 5}, {-1, 0, 0, 0, 5}, {-1, 0, 0, 0, 0}, {-1, 0, 0, 0, 5},
             {-1, 0, 0, 0, 5}};
  t.leaf_ctx.assign(t.nodes.size(), -1);
  int leaf = 0;
  for (size_t i = 0; i < t.nodes.size(); i++)
    if (t.nodes[i].prop < 0) t.leaf_ctx[i] = leaf++;
  return t;
}
static void TreeLookup(const TreeSpec& t, int chan, int stream, int* ctx, uint32_t* predictor) {
  int pos = 0;
  while (t.nodes[pos].prop >= 0) {
    int v = t.nodes[pos].prop == 0 ? chan : stream;
    pos = v > t.nodes[pos].split ? t.nodes[pos].l : t.nodes[pos].r;
  }
  *ctx = t.leaf_ctx[pos];
  *predictor = t.nodes[pos].predictor;
}
static void TreeTokens(const TreeSpec& t, std::vector<Token>* out) {
  for (const auto& n : t.nodes) {
    if (n.prop < 0) {
      out->push_back({1, 0});            // property+1 = 0 => leaf
      out->push_back({2, n.predictor});  // predictor
      out->push_back({3, 0});            // offset
      out->push_back({4, 0});            // multiplier log
      out->push_back({5, 0});            // multiplier bits
    } else {
      out->push_back({1, uint32_t(n.prop + 1)});
      out->push_back({0, PackSigned(n.split)});
    }
  }
}

static inline int32_t ClampedGradient(int32_t n, int32_t w, int32_t l) {
  int32_t m = std::min(n, w), M = std::max(n, w);
  int32_t grad = n + w - l;
  return l > M ? m : (l < m ? M : grad);
}

static void ModularTokens(const TreeSpec& tree, const int32_t* px, size_t w, size_t h, int chan, int stream,
                          std::vector<Token>* out) {
  int ctx;
  uint32_t predictor;
  TreeLookup(tree, chan, stream, &ctx, &predictor);
  for (size_t y = 0; y < h; y++)
    for (size_t x = 0; x < w; x++) {
      const int32_t* p = px + y * w + x;
      int32_t left = x ? p[-1] : (y ? p[-ptrdiff_t(w)] : 0);
      int32_t top = y ? p[-ptrdiff_t(w)] : left;
      int32_t topleft = (x && y) ? p[-1 - ptrdiff_t(w)] : left;
      int32_t pred = predictor == 5 ? ClampedGradient(left, top, topleft) : 0;
      out->push_back({uint32_t(ctx), PackSigned(*p - pred)});
    }
}

// ---------------------------------------------------------------- colour + transforms
static inline float SrgbToLinear(float v) {
  return v <= 0.04045f ? v / 12.92f : std::pow((v + 0.055f) / 1.055f, 2.4f);
}
static void RgbToXyb(const uint8_t* rgb, size_t xs, size_t ys, size_t xp, size_t yp, std::vector<float> planes[3]) {
  static const float kM[9] = {0.30f, 1.0f - 0.078f - 0.30f, 0.078f, 0.23f, 1.0f - 0.078f - 0.23f, 0.078f,
                              0.24342268924547819f, 0.20476744424496821f, 1.0f - 0.24342268924547819f - 0.20476744424496821f};
  const float bias = 0.0037930732552754493f, cb = std::cbrt(bias);
  float lut[256];
  for (int i = 0; i < 256; i++) lut[i] = SrgbToLinear(i / 255.0f);
  for (int c = 0; c < 3; c++) planes[c].assign(xp * yp, 0.0f);
  for (size_t y = 0; y < yp; y++) {
    size_t sy = std::min(y, ys - 1);
    for (size_t x = 0; x < xp; x++) {
      size_t sx = std::min(x, xs - 1);
      const uint8_t* p = rgb + (sy * xs + sx) * 3;
      float r = lut[p[0]], g = lut[p[1]], b = lut[p[2]];
      float mr = kM[0] * r + kM[1] * g + kM[2] * b + bias;
      float mg = kM[3] * r + kM[4] * g + kM[5] * b + bias;
      float mb = kM[6] * r + kM[7] * g + kM[8] * b + bias;
      float gr = std::cbrt(mr) - cb, gg = std::cbrt(mg) - cb, gb = std::cbrt(mb) - cb;
      planes[0][y * xp + x] = 0.5f * (gr - gg);
      planes[1][y * xp + x] = 0.5f * (gr + gg);
      planes[2][y * xp + x] = gb;
    }
  }
}

// The frame's three channels of an image that is NOT xyb_encoded: the sRGB samples in [0, 1] as they are (ColorTransform
// kNone: channel c = component c) or the inverse of the decoder's YCbCr stage (stage_ycbcr.cc:41-60: R = Y' + 1.402 Cr,
// B = Y' + 1.772 Cb, G from the luma equation, Y' = Y + 128 / 255; channels Cb, Y, Cr).
static void RgbToPlain(const uint8_t* rgb, size_t xs, size_t ys, size_t xp, size_t yp, bool ycbcr, std::vector<float> planes[3]) {
  for (int c = 0; c < 3; c++) planes[c].assign(xp * yp, 0.0f);
  for (size_t y = 0; y < yp; y++) {
    size_t sy = std::min(y, ys - 1);
    for (size_t x = 0; x < xp; x++) {
      size_t sx = std::min(x, xs - 1);
      const uint8_t* p = rgb + (sy * xs + sx) * 3;
      const float r = p[0] / 255.0f, g = p[1] / 255.0f, b = p[2] / 255.0f;
      if (!ycbcr) {
        planes[0][y * xp + x] = r;
        planes[1][y * xp + x] = g;
        planes[2][y * xp + x] = b;
        continue;
      }
      const float yy = 0.299f * r + 0.587f * g + 0.114f * b;
      planes[0][y * xp + x] = (b - yy) / 1.772f;
      planes[1][y * xp + x] = yy - 128.0f / 255.0f;
      planes[2][y * xp + x] = (r - yy) / 1.402f;
    }
  }
}

// The RAW table of Params::raw_quant: a JPEG-like ramp per channel (positive integers), and its denominator (an exact
// binary16): weight = 1 / (den * q), so the decoder multiplies by den * q.
static const float kRawQuantDen = 1.0f / 2048.0f;
static std::vector<int32_t> RawQuantTable() {
  std::vector<int32_t> q(3 * 64);
  for (int c = 0; c < 3; c++)
    for (int y = 0; y < 8; y++)
      for (int x = 0; x < 8; x++) q[c * 64 + y * 8 + x] = 1 + (x + y) + (x * y) / 4 + (c == 1 ? 0 : 2 + c);
  return q;
}

struct Basis {
  std::vector<float> m[9];
  Basis() {
    for (int l = 0; l <= 8; l++) {
      int N = 1 << l;
      m[l].resize(size_t(N) * N);
      for (int n = 0; n < N; n++)
        for (int k = 0; k < N; k++) m[l][size_t(n) * N + k] = float((k ? std::sqrt(2.0) : 1.0) * std::cos((n + 0.5) * k * M_PI / N));
    }
  }
};
static const Basis& GetBasis() {
  static const Basis b;
  return b;
}
// Forward scaled DCT of an R x C block into the codestream coefficient layout (rows = short side).
static void ForwardDct(const float* in, size_t stride, int R, int C, float* coef, std::vector<float>& tmp) {
  const float* br = GetBasis().m[FloorLog2(R)].data();
  const float* bc = GetBasis().m[FloorLog2(C)].data();
  tmp.resize(size_t(R) * C);
  for (int y = 0; y < R; y++)
    for (int kx = 0; kx < C; kx++) {
      float s = 0;
      for (int x = 0; x < C; x++) s += in[y * stride + x] * bc[size_t(x) * C + kx];
      tmp[size_t(y) * C + kx] = s;
    }
  const float norm = 1.0f / (float(R) * float(C));
  for (int ky = 0; ky < R; ky++)
    for (int kx = 0; kx < C; kx++) {
      float s = 0;
      for (int y = 0; y < R; y++) s += tmp[size_t(y) * C + kx] * br[size_t(y) * R + ky];
      s *= norm;
      if (R < C) coef[ky * C + kx] = s;
      else coef[kx * R + ky] = s;
    }
}
static inline float ResampleScale(int n, int i) {
  double N = 8.0 * n;
  return float(1.0 / (std::cos(i / (2 * N) * M_PI) * std::cos(i / N * M_PI) * std::cos(i / (N / 2) * M_PI)));
}

// ---------------------------------------------------------------- frame model shared by both modes
struct FrameModel {
  size_t xs, ys, xb, yb;  // pixels, blocks
  std::vector<uint8_t> acs;        // (strategy<<1)|first, 0xFF = unset
  std::vector<int32_t> qf;         // 1..256 at first blocks
  std::vector<uint8_t> sharp;      // per block
  std::vector<int8_t> ytox, ytob;  // per 64x64 tile
  std::vector<int32_t> dc[3];      // quantised DC ints per block, stored X, Y, B
  // quantised AC, per group: [c][65536] block-contiguous
  std::vector<std::vector<int32_t>> coeffs;
  // ... or, when the forward path delivered them in one block: [group][3][65536] (then `coeffs` stays empty)
  std::unique_ptr<int32_t[]> flat_coeffs;
  const int32_t* GroupCoeffs(size_t g) const { return flat_coeffs ? flat_coeffs.get() + g * 3 * 65536 : coeffs[g].data(); }
  // ... or they never came to the host: the forward path tokenised them on the device (jxlhip_enc_tokens): every group's
  // (context, value) pairs in bitstream order, group g at dev_tokens[dev_token_base[g] .. + dev_token_count[g])
  std::vector<uint64_t> dev_tokens;  // {context, value} pairs of uint32
  std::vector<uint32_t> dev_token_base, dev_token_count;
  uint32_t global_scale, quant_dc;
  int epf_iters, gab;
  uint64_t flags;
  size_t img_xs = 0, img_ys = 0;  // image size when the frame is coded downsampled (upsampling > 1)
  std::vector<uint8_t> alpha;     // optional 8-bit alpha plane (xs * ys): one Modular-coded extra channel (dec_frame.cc:511-542)
  // ... or, with an extra-channel upsampling factor `alpha_ups` (1, 2, 4, 8; at least the frame's own: frame_header.cc:272-283),
  // alpha_w x alpha_h = ceil(image / alpha_ups) samples: the channel's shift against the frame is log2(alpha_ups / upsampling)
  // (dec_modular.cc:262-271)
  size_t alpha_w = 0, alpha_h = 0;
  uint32_t alpha_ups = 1, alpha_shift = 0;
  // YCbCr chroma subsampling (frame_header.h:81-166): channel_mode per channel (0 = 1x1, 1 = 2x2, 2 = 2x1, 3 = 1x2 samples per
  // MCU), and the shifts that follow from it: channel c has (xb >> hs[c]) x (yb >> vs[c]) blocks; its block at (sx, sy) is
  // coded with the frame's block (sx << hs, sy << vs); its DC sits in the top-left part of dc[c] (row stride xb)
  uint32_t cmode[3] = {0, 0, 0};
  int hs[3] = {0, 0, 0}, vs[3] = {0, 0, 0}, maxhs = 0, maxvs = 0;
  bool Subsampled() const { return maxhs || maxvs; }
  bool Present(int c, size_t bx, size_t by) const { return ((bx & ((size_t(1) << hs[c]) - 1)) | (by & ((size_t(1) << vs[c]) - 1))) == 0; }
  void SetSubsampling(uint32_t packed) {
    static const int kH[4] = {0, 1, 1, 0}, kV[4] = {0, 1, 0, 1};
    maxhs = maxvs = 0;
    for (int c = 0; c < 3; c++) {
      cmode[c] = (packed >> (2 * c)) & 3;
      maxhs = std::max(maxhs, kH[cmode[c]]);
      maxvs = std::max(maxvs, kV[cmode[c]]);
    }
    for (int c = 0; c < 3; c++) {
      hs[c] = maxhs - kH[cmode[c]];
      vs[c] = maxvs - kV[cmode[c]];
    }
  }
  // blocks per side: whole MCUs (frame_dimensions.h:43-44)
  void SetSize(size_t x, size_t y) {
    xs = x; ys = y;
    xb = DivCeil(x, size_t(8) << maxhs) << maxhs;
    yb = DivCeil(y, size_t(8) << maxvs) << maxvs;
  }
};

// Test aid: a coded ICC profile (the byte stream lib/jxl/icc_codec.cc reads: U64 size, histograms, ANS data) that the
// next streams carry after their headers, with ImageMetadata.color_encoding.want_icc set (jxlenc_set_embedded_icc).
// jxlenc_set_orientation: the ImageMetadata orientation the next streams carry (1..8; 1 writes no extra_fields).
static uint32_t g_orientation = 1;
// jxlenc_set_animation: the next streams are frames of an animation (ImageMetadata.have_animation with this
// AnimationHeader; each frame header carries `duration` ticks and `is_last`). A whole animation is the first frame's
// stream followed by the frame parts (from jxlenc_last_header_bytes() on) of the others.
struct AnimationState {
  bool enabled = false;  // multi-frame mode: is_last is coded from here
  bool timed = true;     // ... with an AnimationHeader and per-frame durations (false: a layered still)
  uint32_t tps_num = 10, tps_den = 1, loops = 0, duration = 1;
  bool is_last = true;
};
static AnimationState g_anim;
// jxlenc_set_layer: how the next frame sits on the canvas (frame_header.cc:303-370): crop origin, the canvas size (to
// know whether the frame covers it), BlendingInfo of the colour channels and of the alpha channel, the slot it is kept in.
struct LayerState {
  bool enabled = false;
  int32_t x0 = 0, y0 = 0;
  uint32_t canvas_w = 0, canvas_h = 0;
  uint32_t mode = 0, alpha_mode = 0, source = 0, alpha_source = 0, clamp = 0, save_as = 0;
};
static LayerState g_layer;
// jxlenc_set_reference_frame: the next frame is a kReferenceOnly frame (frame_header.cc:215-240, 372-411): never shown,
// kept before its colour transform in slot `save_as` for the patches of later frames; coded with its own size.
// jxlenc_set_image_size: the image size the next stream's headers declare (a reference frame that comes first is smaller
// or larger than the image). jxlenc_set_patches: the patch dictionary of the next frames (flat: number of references, per
// reference: slot, x0, y0, xsize, ysize, count, then per position: x, y, colour blend mode 0..7, clamp, the first extra
// channel's blend mode 0..7 and clamp), written the way PatchDictionary::Decode reads it (dec_patch_dictionary.cc:32-175;
// further extra channels are left alone: mode kNone).
static int g_reference_slot = -1;
// jxlenc_set_dc_frame: the next frame is a kDCFrame of this level (frame_header.cc:310-320, 372-411: a Passes bundle and the
// level, then no size, no blending, no timing, no save_as_reference), coded at the size the image header implies
// (image / 8^level); jxlenc_set_use_dc_frame: the next VarDCT frame sets kUseDcFrame (no upsampling fields in its header,
// no DC stream in its DC groups: frame_header.cc:263, dec_frame.cc:322-326).
static int g_dc_frame_level = 0;
static bool g_use_dc_frame = false;
static uint32_t g_image_w = 0, g_image_h = 0;
static std::vector<int32_t> g_patches;
// jxlenc_set_frame_name: the name the next frames carry (frame_header.cc:431 VisitNameString: U32 length, bytes)
static std::string g_frame_name;
static void WriteFrameName(BitWriter& bw) {
  static const uint32_t b[4] = {0, 4, 5, 10}, o[4] = {0, 0, 16, 48};
  WriteU32Sel(bw, uint32_t(g_frame_name.size()), b, o);
  for (unsigned char ch : g_frame_name) bw.Write(8, ch);
}
// jxlenc_set_color_encoding: the enum ColorEncoding the next LOSSLESS streams declare (color_encoding_internal.cc:144-200):
// white point / primaries / transfer function values of color_encoding.h (2 = custom xy in millionths; have_gamma with
// gamma in 1e-7), rendering intent. The samples are written as given: a non-XYB image's colour encoding is metadata.
struct ColorState {
  bool enabled = false;
  uint32_t white_point = 1, primaries = 1, have_gamma = 0, gamma = 0, transfer_function = 13, intent = 1;
  int32_t xy[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // white, red, green, blue
};
static ColorState g_color;
static void WriteEnum(BitWriter& bw, uint32_t v) {  // U32(Val(0), Val(1), BitsOffset(4, 2), BitsOffset(6, 18))
  static const uint32_t b[4] = {0, 0, 4, 6}, o[4] = {0, 1, 2, 18};
  WriteU32Sel(bw, v, b, o);
}
static void WriteCustomXy(BitWriter& bw, const int32_t* xy) {
  static const uint32_t b[4] = {19, 19, 20, 21}, o[4] = {0, 524288, 1048576, 2097152};
  for (int i = 0; i < 2; i++) WriteU32Sel(bw, xy[i] >= 0 ? uint32_t(xy[i]) * 2 : uint32_t(-(xy[i] + 1)) * 2 + 1, b, o);
}
static void WriteColorEncodingFields(BitWriter& bw, bool gray) {
  const ColorState& C = g_color;
  bw.Write(1, 0);  // not all_default
  bw.Write(1, 0);  // no ICC
  WriteEnum(bw, gray ? 1 : 0);
  WriteEnum(bw, C.white_point);
  if (C.white_point == 2) WriteCustomXy(bw, C.xy);
  if (!gray) {
    WriteEnum(bw, C.primaries);
    if (C.primaries == 2)
      for (int c = 0; c < 3; c++) WriteCustomXy(bw, C.xy + 2 + 2 * c);
  }
  bw.Write(1, C.have_gamma ? 1 : 0);
  if (C.have_gamma) bw.Write(24, C.gamma);
  else WriteEnum(bw, C.transfer_function);
  WriteEnum(bw, C.intent);
}
// jxlenc_set_alpha_premultiplied: the alpha channel is declared associated (ExtraChannelInfo.alpha_associated,
// image_metadata.cc:158-200); the samples are written as given.
static bool g_alpha_premultiplied = false;
static void WriteAlphaChannelInfo(BitWriter& bw) {
  if (!g_alpha_premultiplied) {
    bw.Write(1, 1);  // ExtraChannelInfo all_default: 8-bit alpha
    return;
  }
  bw.Write(1, 0);  // not all_default
  bw.Write(2, 0);  // type: alpha
  bw.Write(1, 0);  // integer samples
  bw.Write(2, 0);  //   8 bits
  bw.Write(2, 0);  // dim_shift 0
  bw.Write(2, 0);  // no name
  bw.Write(1, 1);  // alpha_associated
}
// Crop + blending fields of a frame header for a frame of fw x fh pixels; returns whether a save_before_color_transform
// bit follows the timing fields (frame_header.cc:396-408).
static bool WriteCropAndBlending(BitWriter& bw, uint32_t fw, uint32_t fh, bool have_alpha) {
  const LayerState& L = g_layer;
  static const uint32_t db[4] = {8, 11, 14, 30}, dof[4] = {0, 256, 2304, 18688};
  if (g_dc_frame_level > 0) return false;  // kDCFrame: nothing (its size follows from the image's)
  if (g_reference_slot >= 0) {  // kReferenceOnly: its own size, no origin, no blending info
    bw.Write(1, 1);
    WriteU32Sel(bw, fw, db, dof);
    WriteU32Sel(bw, fh, db, dof);
    return false;
  }
  bool partial = false;
  if (!L.enabled) {
    bw.Write(1, 0);  // no custom size/origin
  } else {
    bw.Write(1, 1);
    auto pack = [](int32_t v) { return v >= 0 ? uint32_t(v) * 2 : uint32_t(-(v + 1)) * 2 + 1; };
    WriteU32Sel(bw, pack(L.x0), db, dof);
    WriteU32Sel(bw, pack(L.y0), db, dof);
    WriteU32Sel(bw, fw, db, dof);
    WriteU32Sel(bw, fh, db, dof);
    partial = L.x0 > 0 || L.y0 > 0 || int64_t(fw) + L.x0 < int64_t(L.canvas_w) || int64_t(fh) + L.y0 < int64_t(L.canvas_h);
  }
  auto info = [&](uint32_t mode, uint32_t source) {
    static const uint32_t mb[4] = {0, 0, 0, 2}, mo[4] = {0, 1, 2, 3};
    WriteU32Sel(bw, mode, mb, mo);
    const bool alpha_modes = have_alpha && (mode == 2 || mode == 3);
    if (alpha_modes) bw.Write(2, 0);  // alpha_channel 0
    if (alpha_modes || mode == 4) bw.Write(1, L.clamp ? 1 : 0);
    if (mode != 0 || partial) bw.Write(2, source);
  };
  info(L.enabled ? L.mode : 0, L.source);
  if (have_alpha) info(L.enabled ? L.alpha_mode : 0, L.alpha_source);
  return (!L.enabled || L.mode == 0) && !partial;
}
static size_t g_last_header_bytes = 0;  // where the frame header of the last written stream starts
static struct {
  bool enabled = false;
  uint32_t xs = 0, ys = 0;
} g_preview;  // jxlenc_set_preview: the image header announces a preview frame of this size (headers.cc:155-183 PreviewHeader)
static bool ExtraFields() { return g_orientation != 1 || (g_anim.enabled && g_anim.timed) || g_preview.enabled; }
// image_metadata.cc:283-300: extra_fields = orientation, no intrinsic size, no preview, animation (:235-250).
static void WriteExtraFields(BitWriter& bw) {
  if (!ExtraFields()) {
    bw.Write(1, 0);
    return;
  }
  bw.Write(1, 1);
  bw.Write(3, g_orientation - 1);
  bw.Write(1, 0);  // no intrinsic size
  bw.Write(1, g_preview.enabled ? 1 : 0);
  if (g_preview.enabled) {  // headers.cc:155-183, ratio 0 (both sides coded)
    const bool div8 = g_preview.xs % 8 == 0 && g_preview.ys % 8 == 0;
    static const uint32_t b8[4] = {0, 0, 5, 9}, o8[4] = {16, 32, 1, 33};
    static const uint32_t b1[4] = {6, 8, 10, 12}, o1[4] = {1, 65, 321, 1345};
    bw.Write(1, div8 ? 1 : 0);
    WriteU32Sel(bw, div8 ? g_preview.ys / 8 : g_preview.ys, div8 ? b8 : b1, div8 ? o8 : o1);
    bw.Write(3, 0);
    WriteU32Sel(bw, div8 ? g_preview.xs / 8 : g_preview.xs, div8 ? b8 : b1, div8 ? o8 : o1);
  }
  bw.Write(1, g_anim.enabled && g_anim.timed ? 1 : 0);
  if (g_anim.enabled && g_anim.timed) {
    static const uint32_t nb[4] = {0, 0, 10, 30}, no[4] = {100, 1000, 1, 1};
    static const uint32_t db[4] = {0, 0, 8, 10}, dof[4] = {1, 1001, 1, 1};
    static const uint32_t lb[4] = {0, 3, 16, 32}, lo[4] = {0, 0, 0, 0};
    WriteU32Sel(bw, g_anim.tps_num, nb, no);
    WriteU32Sel(bw, g_anim.tps_den, db, dof);
    WriteU32Sel(bw, g_anim.loops, lb, lo);
    bw.Write(1, 0);  // no timecodes
  }
}
// image_metadata.cc:340-344: with extra_fields a ToneMapping bundle follows the colour encoding (all_default here).
static void WriteToneMapping(BitWriter& bw) {
  if (ExtraFields()) bw.Write(1, 1);
}
// frame_header.cc:130-150, 372-399: the animation fields of a frame header, is_last, and (not last) save_as_reference 0.
// A frame with a duration and no reference slot cannot be referenced: no save_before_color_transform bit follows.
static void WriteFrameTiming(BitWriter& bw, bool replace_whole_canvas = true) {
  if (g_dc_frame_level > 0) return;  // kDCFrame: no timing, never last, no reference slot
  if (g_reference_slot >= 0) {  // no duration, no is_last (= false); save_as_reference, then save_before_color_transform
    bw.Write(2, uint32_t(g_reference_slot));
    bw.Write(1, 1);
    return;
  }
  const bool timed = g_anim.enabled && g_anim.timed;
  if (timed) {
    static const uint32_t b[4] = {0, 0, 8, 32}, o[4] = {0, 1, 0, 0};
    WriteU32Sel(bw, g_anim.duration, b, o);
  }
  const bool last = !g_anim.enabled || g_anim.is_last;
  const uint32_t duration = timed ? g_anim.duration : 0, save_as = g_layer.enabled ? g_layer.save_as : 0;
  bw.Write(1, last ? 1 : 0);
  if (!last) bw.Write(2, save_as);  // save_as_reference
  // CanBeReferenced (frame_header.h:373-379) and a full-canvas replace: save_before_color_transform = false
  if (!last && (duration == 0 || save_as != 0) && replace_whole_canvas) bw.Write(1, 0);
}
// jxlenc_set_splines: the quantised spline dictionary the next streams carry (frame flag kSplines), flat:
// [quantisation adjustment, number of splines, then per spline: start x, start y, number of further control points,
// their double deltas (x, y each), 3 x 32 colour DCT coefficients (X, Y, B), 32 sigma DCT coefficients]. Written the way
// Splines::Decode reads it (splines.cc:596-648): 6 contexts, starting points as deltas of each other, ANS coded.
static std::vector<int32_t> g_splines;
static void WriteCodeHeader(BitWriter& bw, const EncCode& code);
static void WritePatches(BitWriter& bw, size_t num_extra) {
  const int32_t* d = g_patches.data();
  const size_t nref = size_t(*d++);
  std::vector<Token> tk;
  tk.push_back({0, uint32_t(nref)});
  auto pack = [](int64_t v) { return v >= 0 ? uint32_t(v) * 2 : uint32_t(-(v + 1)) * 2 + 1; };
  for (size_t r = 0; r < nref; r++) {
    tk.push_back({1, uint32_t(d[0])});
    tk.push_back({3, uint32_t(d[1])});
    tk.push_back({3, uint32_t(d[2])});
    tk.push_back({2, uint32_t(d[3] - 1)});
    tk.push_back({2, uint32_t(d[4] - 1)});
    const size_t count = size_t(d[5]);
    tk.push_back({7, uint32_t(count - 1)});
    d += 6;
    int64_t px = 0, py = 0;
    for (size_t i = 0; i < count; i++, d += 6) {
      if (i == 0) {
        tk.push_back({4, uint32_t(d[0])});
        tk.push_back({4, uint32_t(d[1])});
      } else {
        tk.push_back({6, pack(int64_t(d[0]) - px)});
        tk.push_back({6, pack(int64_t(d[1]) - py)});
      }
      px = d[0];
      py = d[1];
      // dec_patch_dictionary.cc:135-163: the colour channels' mode, then every extra channel's; a mode that blends through
      // alpha (4..7) names the alpha channel when the image has more than one extra channel (never here), and it and kMul
      // carry a clamp flag (contexts patch_dictionary_internal.h:12-24)
      tk.push_back({5, uint32_t(d[2])});
      if (d[2] >= 3) tk.push_back({9, uint32_t(d[3] ? 1 : 0)});
      for (size_t e = 0; e < num_extra; e++) {
        const uint32_t em = e == 0 ? uint32_t(d[4]) : 0u;
        tk.push_back({5, em});
        if (em >= 3) tk.push_back({9, uint32_t(d[5] ? 1 : 0)});
      }
    }
  }
  jxh::HybridCfg cfg;
  cfg.split_exp = 4; cfg.split_token = 16; cfg.msb = 2; cfg.lsb = 0;
  EncCode code;
  BuildCode({&tk}, 10, 10, cfg, &code);
  WriteCodeHeader(bw, code);
  WriteTokens(bw, tk.data(), tk.size(), code);
}
static void WriteSplines(BitWriter& bw) {
  const int32_t* d = g_splines.data();
  const int32_t adjust = *d++;
  const size_t n = size_t(*d++);
  std::vector<Token> tk;
  tk.push_back({2, uint32_t(n - 1)});
  struct One {
    int64_t x, y;
    const int32_t *deltas, *dct;
    size_t nd;
  };
  std::vector<One> sp(n);
  for (auto& o : sp) {
    o.x = *d++;
    o.y = *d++;
    o.nd = size_t(*d++);
    o.deltas = d;
    d += 2 * o.nd;
    o.dct = d;
    d += 128;
  }
  int64_t lx = 0, ly = 0;
  auto pack = [](int64_t v) { return v >= 0 ? uint32_t(v) * 2 : uint32_t(-(v + 1)) * 2 + 1; };
  for (size_t i = 0; i < n; i++) {
    tk.push_back({1, i ? pack(sp[i].x - lx) : uint32_t(sp[i].x)});
    tk.push_back({1, i ? pack(sp[i].y - ly) : uint32_t(sp[i].y)});
    lx = sp[i].x;
    ly = sp[i].y;
  }
  tk.push_back({0, pack(adjust)});
  for (const auto& o : sp) {
    tk.push_back({3, uint32_t(o.nd)});
    for (size_t i = 0; i < 2 * o.nd; i++) tk.push_back({4, pack(o.deltas[i])});
    for (size_t i = 0; i < 128; i++) tk.push_back({5, pack(o.dct[i])});
  }
  jxh::HybridCfg cfg;
  cfg.split_exp = 4; cfg.split_token = 16; cfg.msb = 2; cfg.lsb = 0;
  EncCode code;
  BuildCode({&tk}, 6, 6, cfg, &code);
  WriteCodeHeader(bw, code);
  WriteTokens(bw, tk.data(), tk.size(), code);
}
static std::vector<uint8_t> g_embedded_icc;
// CustomTransformData with coded upsampling weights (image_metadata.cc:87-214): bit k of the mask = the 2^(k+1)-fold
// matrix is coded, as the default weights scaled by seeded factors in [0.75, 1.25] and rounded to half precision.
static uint32_t g_custom_ups_mask = 0, g_custom_ups_seed = 1;
#include "../host/upsampling_weights.inc"
static float RoundToF16(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  u = (u + 0x1000u) & ~0x1FFFu;  // nearest value with a 10-bit mantissa
  float r;
  memcpy(&r, &u, 4);
  return std::fabs(r) < 6.2e-5f ? 0.0f : r;  // (below the normal half-precision range: zero)
}
static size_t g_embedded_icc_bits = 0;  // its exact length (the decoder aligns to a byte right after the last bit)
static void AppendEmbeddedIcc(BitWriter& bw) {
  for (size_t i = 0; i < g_embedded_icc_bits; i += 8) {
    const unsigned n = unsigned(std::min<size_t>(8, g_embedded_icc_bits - i));
    bw.Write(n, g_embedded_icc[i / 8] & ((1u << n) - 1));
  }
}

struct Params {
  float distance;
  int32_t epf_iters;       // -1 = choose from distance like the reference (enc_frame.cc:317-341)
  int32_t gab;             // -1 = on
  int32_t strategy_mode;   // 0 = DCT8 only, 1 = heuristic mix (8..64), 2 = uniform random over `strategy_mask`
  uint32_t strategy_mask;  // bit per strategy for mode 2 (0 = all 27)
  uint32_t seed;
  int32_t max_clusters;    // 0 = default (64)
  int32_t skip_dc_smoothing;
  int32_t random_cmap;     // random chroma-from-luma factors (always on in random mode)
  int32_t zero_ac;         // random mode: leave every AC coefficient zero (DC-only stream)
  int32_t num_histograms;  // AC histogram sets (group g uses set g % num_histograms); 0 or 1 = one
  int32_t big_coeffs;      // random mode: sprinkle magnitudes beyond 16 bits (forces int32 coefficient storage in decoders)
  int32_t num_passes;      // 1..3; progressive: pass p of n carries the coefficient's bits above shift n - 1 - p that the passes
                           // before it have not (shifts n - 1 .. 0); three passes also name a downsampling level (4x after pass 0)
  int32_t upsampling;      // 0/1, 2, 4 or 8: the frame is coded at ceil(size / upsampling) and flagged for upsampling
  int32_t custom_orders;   // 1 = code a (seeded random) custom coefficient order for every used order bucket, channel and pass
  int32_t custom_bctx;     // 1 = code a block context map with quant-field and DC thresholds (entropy_coder.cc:25-60)
  int32_t custom_cmap;     // 1 = non-default colour-correlation header (factor 100, bases 0.25 / 0.75, DC factors 3 / -5) and
                           //     x/b quant-matrix scales 2 / 4: valid streams, but image mode does not compensate for them
  int32_t custom_lf;       // 1 = non-default loop filter header: Gaborish weights, EPF sharpness LUT, channel scales and
                           //     sigma parameters, and non-default DC dequantisation steps (valid streams, not tuned ones)
  int32_t ac_code_mode;    // AC coefficient streams: bit 0 = prefix codes instead of ANS (libjxl's fastest efforts), bit 1 = LZ77
  int32_t noise;           // > 0: frame flag kNoise with the strength LUT point i = min(1023, noise + 40 * i) / 1024
  int32_t cfl_fit;         // 1 = image mode fits the chroma-from-luma factor of every 64x64 tile the way the reference's
                           //     fast path does (enc_chroma_from_luma.cc:128-151, 204-352): least squares over the tile's AC
  int32_t color_transform; // ColorTransform of the frame (frame_header.h): 0 = XYB, 1 = none (the sRGB samples themselves in
                           //     the three channels), 2 = YCbCr (full-range BT.601, stage_ycbcr.cc:41-60; channels Cb, Y, Cr);
                           //     1 and 2 make an image that is not xyb_encoded
  int32_t raw_quant;       // 1 = the 8x8 DCT's dequantisation table is coded RAW (quant_weights.cc:268-276: a denominator and
                           //     the table as a small Modular image, what JPEG recompression writes): a JPEG-like ramp
  int32_t chroma_subsampling;  // YCbCr frames (color_transform 2) only: channel_mode of Cb, Y, Cr in bits 0-1, 2-3, 4-5
                           //     (frame_header.h:81-166; 0 = one sample per MCU side, 1 = 2x2, 2 = 2x1, 3 = 1x2): 4:2:0 = 0b000100 = 4,
                           //     4:2:2 = 8, 4:4:0 = 12. Only transforms that cover one block; no adaptive DC smoothing.
  int32_t ec_upsampling;   // jxlenc_encode_rgba8: the alpha channel is coded at ceil(size / this) (0/1, 2, 4, 8; at least `upsampling`,
                           //     at most four times it) and flagged for upsampling (frame_header.cc:265-283)
};

static bool Fits(const FrameModel& f, size_t bx, size_t by, int st) {
  size_t cx = jxh::kCoveredX[st], cy = jxh::kCoveredY[st];
  if (bx + cx > f.xb || by + cy > f.yb) return false;
  if ((bx % 32) + cx > 32 || (by % 32) + cy > 32) return false;
  for (size_t y = 0; y < cy; y++)
    for (size_t x = 0; x < cx; x++)
      if (f.acs[(by + y) * f.xb + bx + x] != 0xFF) return false;
  return true;
}
static void Place(FrameModel& f, size_t bx, size_t by, int st) {
  size_t cx = jxh::kCoveredX[st], cy = jxh::kCoveredY[st];
  for (size_t y = 0; y < cy; y++)
    for (size_t x = 0; x < cx; x++) f.acs[(by + y) * f.xb + bx + x] = uint8_t((st << 1) | ((x | y) == 0));
}

// ---------------------------------------------------------------- bitstream assembly
static void WriteSizeDim(BitWriter& bw, uint32_t v) {
  static const uint32_t bits[4] = {9, 13, 18, 30}, offs[4] = {1, 1, 1, 1};
  WriteU32Sel(bw, v, bits, offs);
}

static void Assemble(const FrameModel& f, const Params& p, std::vector<uint8_t>* out) {
  const size_t xg = DivCeil(f.xs, 256), yg = DivCeil(f.ys, 256), num_groups = xg * yg;
  const size_t xdg = DivCeil(f.xb, 256), ydg = DivCeil(f.yb, 256), ndc = xdg * ydg;
  const TreeSpec tree = MakeTree(ndc);
  // ---- tokenise modular streams
  std::vector<Token> tree_tokens;
  TreeTokens(tree, &tree_tokens);
  std::vector<std::vector<Token>> dc_tokens(ndc), meta_tokens(ndc);
  std::vector<size_t> meta_count(ndc);
  for (size_t g = 0; g < ndc; g++) {
    size_t bx0 = (g % xdg) * 256, by0 = (g / xdg) * 256;
    size_t bw_ = std::min<size_t>(256, f.xb - bx0), bh = std::min<size_t>(256, f.yb - by0);
    std::vector<int32_t> tmp(bw_ * bh);
    static const int kDcChan[3] = {1, 0, 2};  // stream channel order Y, X, B
    for (int ch = 0; ch < 3; ch++) {
      // (a subsampled channel's part of the DC group: dec_modular.cc:443-452)
      const int c = kDcChan[ch];
      const size_t sx0 = bx0 >> f.hs[c], sy0 = by0 >> f.vs[c], sw = bw_ >> f.hs[c], sh = bh >> f.vs[c];
      for (size_t y = 0; y < sh; y++)
        for (size_t x = 0; x < sw; x++) tmp[y * sw + x] = f.dc[c][(sy0 + y) * f.xb + sx0 + x];
      ModularTokens(tree, tmp.data(), sw, sh, ch, int(1 + g), &dc_tokens[g]);
    }
    // AC metadata
    size_t cw = (bw_ + 7) / 8, chh = (bh + 7) / 8, tiles_x = DivCeil(f.xb, 8);
    std::vector<int32_t> cm(cw * chh);
    for (int which = 0; which < 2; which++) {
      for (size_t y = 0; y < chh; y++)
        for (size_t x = 0; x < cw; x++) cm[y * cw + x] = (which ? f.ytob : f.ytox)[(by0 / 8 + y) * tiles_x + bx0 / 8 + x];
      ModularTokens(tree, cm.data(), cw, chh, which, int(1 + 2 * ndc + g), &meta_tokens[g]);
    }
    std::vector<int32_t> l0, l1;
    for (size_t y = 0; y < bh; y++)
      for (size_t x = 0; x < bw_; x++) {
        uint8_t a = f.acs[(by0 + y) * f.xb + bx0 + x];
        if (a & 1) {
          l0.push_back(a >> 1);
          l1.push_back(f.qf[(by0 + y) * f.xb + bx0 + x] - 1);
        }
      }
    meta_count[g] = l0.size();
    std::vector<int32_t> lst(l0);
    lst.insert(lst.end(), l1.begin(), l1.end());
    ModularTokens(tree, lst.data(), l0.size(), 2, 2, int(1 + 2 * ndc + g), &meta_tokens[g]);
    std::vector<int32_t> sh(bw_ * bh);
    for (size_t y = 0; y < bh; y++)
      for (size_t x = 0; x < bw_; x++) sh[y * bw_ + x] = f.sharp[(by0 + y) * f.xb + bx0 + x];
    ModularTokens(tree, sh.data(), bw_, bh, 3, int(1 + 2 * ndc + g), &meta_tokens[g]);
  }
  jxh::HybridCfg cfg420;
  cfg420.split_exp = 4; cfg420.split_token = 16; cfg420.msb = 2; cfg420.lsb = 0;
  EncCode tree_code, mod_code;
  std::vector<Token> raw_quant_tokens;
  // alpha: channel 0 of the frame's global Modular image. Up to a group in size it is coded whole in stream 0 (DC global),
  // else one rectangle per AC group section, behind the coefficients (stream ids: dec_modular.h:44-67).
  const bool have_alpha = !f.alpha.empty();
  const size_t aw = have_alpha ? (f.alpha_w ? f.alpha_w : f.xs) : 0, ah = have_alpha ? (f.alpha_h ? f.alpha_h : f.ys) : 0;
  const uint32_t ash = f.alpha_shift;
  if (have_alpha && (f.alpha.size() != aw * ah || ash > 2)) throw std::runtime_error("alpha plane: size / shift");
  const bool alpha_global = have_alpha && aw <= 256 && ah <= 256;
  std::vector<Token> alpha_global_tokens;
  std::vector<std::vector<Token>> alpha_group_tokens(have_alpha && !alpha_global ? num_groups : 0);
  std::vector<uint8_t> alpha_group_present(alpha_group_tokens.size(), 0);  // (a group the shifted channel does not reach codes nothing)
  if (alpha_global) {
    std::vector<int32_t> px(f.alpha.begin(), f.alpha.end());
    ModularTokens(tree, px.data(), aw, ah, 0, 0, &alpha_global_tokens);
  }
  for (size_t g = 0; g < alpha_group_tokens.size(); g++) {
    // the group's rectangle of the channel (dec_modular.cc:343-368: the frame rectangle shifted by the channel's shift)
    const size_t x0 = ((g % xg) * 256) >> ash, y0 = ((g / xg) * 256) >> ash;
    if (x0 >= aw || y0 >= ah) continue;
    const size_t w = std::min<size_t>(256 >> ash, aw - x0), h = std::min<size_t>(256 >> ash, ah - y0);
    alpha_group_present[g] = 1;
    std::vector<int32_t> px(w * h);
    for (size_t y = 0; y < h; y++)
      for (size_t x = 0; x < w; x++) px[y * w + x] = f.alpha[(y0 + y) * aw + x0 + x];
    const size_t last_pass = size_t(p.num_passes >= 2 && p.num_passes <= 3 ? p.num_passes : 1) - 1;
    ModularTokens(tree, px.data(), w, h, 0, int(1 + 3 * ndc + 17 + num_groups * last_pass + g), &alpha_group_tokens[g]);
  }
  BuildCode({&tree_tokens}, 6, 6, cfg420, &tree_code);
  {
    std::vector<const std::vector<Token>*> all;
    for (auto& t : dc_tokens) all.push_back(&t);
    for (auto& t : meta_tokens) all.push_back(&t);
    size_t nleaf = 0;
    for (int c : tree.leaf_ctx) nleaf += c >= 0;
    if (alpha_global) all.push_back(&alpha_global_tokens);
    for (const auto& t : alpha_group_tokens) all.push_back(&t);
    if (p.raw_quant) {  // the RAW table of the 8x8 DCT: Modular stream 1 + 3 * num_dc_groups + 0 (dec_modular.h:59-61)
      const std::vector<int32_t> q = RawQuantTable();
      for (int c = 0; c < 3; c++) ModularTokens(tree, q.data() + c * 64, 8, 8, c, int(1 + 3 * ndc), &raw_quant_tokens);
      all.push_back(&raw_quant_tokens);
    }
    BuildCode(all, nleaf, 8, cfg420, &mod_code);
  }
  // ---- tokenise AC groups
  jxh::BlockCtxMap bctx;
  if (p.custom_bctx) {
    // thresholds at quantiles of what the frame actually holds, so that every bucket is exercised
    auto quantile = [](std::vector<int32_t> v, double q) {
      std::sort(v.begin(), v.end());
      return v.empty() ? 0 : v[std::min(v.size() - 1, size_t(q * double(v.size())))];
    };
    std::vector<int32_t> qfs;
    for (size_t i = 0; i < f.acs.size(); i++)
      if (f.acs[i] != 0xFF && (f.acs[i] & 1)) qfs.push_back(f.qf[i]);
    const int32_t q1 = quantile(qfs, 0.33), q2 = quantile(qfs, 0.66);
    bctx.qf_thresholds.clear();
    if (q1 >= 1) bctx.qf_thresholds.push_back(uint32_t(q1));
    if (q2 > q1) bctx.qf_thresholds.push_back(uint32_t(q2));
    bctx.dc_thresholds[0].clear();
    bctx.dc_thresholds[1] = {quantile(f.dc[1], 0.5)};
    bctx.dc_thresholds[2] = {quantile(f.dc[2], 0.3), quantile(f.dc[2], 0.7)};
    if (bctx.dc_thresholds[2][1] <= bctx.dc_thresholds[2][0]) bctx.dc_thresholds[2].pop_back();
    bctx.num_dc_ctxs = (bctx.dc_thresholds[1].size() + 1) * (bctx.dc_thresholds[2].size() + 1);
    const size_t nq = bctx.qf_thresholds.size() + 1;
    bctx.ctx_map.assign(3 * jxh::kNumOrders * nq * bctx.num_dc_ctxs, 0);
    size_t mx = 0;
    for (size_t i = 0; i < bctx.ctx_map.size(); i++) {
      const size_t dc = i % bctx.num_dc_ctxs, qf = (i / bctx.num_dc_ctxs) % nq, ord = (i / (bctx.num_dc_ctxs * nq)) % jxh::kNumOrders,
                   c = i / (bctx.num_dc_ctxs * nq * jxh::kNumOrders);
      const size_t v = c * 5 + (std::min<size_t>(ord, 2) + qf + dc) % 5;  // 15 contexts, every input matters
      bctx.ctx_map[i] = uint8_t(v);
      mx = std::max(mx, v);
    }
    bctx.num_ctxs = mx + 1;
  }
  // DC bucket of a block: from the quantised DC at its top-left corner (dec_modular.cc / compressed_dc.cc)
  auto dc_bucket = [&](size_t i) -> int {
    if (bctx.num_dc_ctxs <= 1) return 0;
    int kx = 0, ky = 0, kb = 0;
    // (compressed_dc.cc:256-290: a subsampled channel's sample that covers the block)
    const size_t bx = i % f.xb, by = i / f.xb;
    auto at = [&](int c) { return f.dc[c][(by >> f.vs[c]) * f.xb + (bx >> f.hs[c])]; };
    for (int t : bctx.dc_thresholds[0]) kx += at(0) > t;
    for (int t : bctx.dc_thresholds[1]) ky += at(1) > t;
    for (int t : bctx.dc_thresholds[2]) kb += at(2) > t;
    int b = kx;
    b = b * int(bctx.dc_thresholds[2].size() + 1) + kb;
    b = b * int(bctx.dc_thresholds[1].size() + 1) + ky;
    return b;
  };
  const size_t nctx = bctx.NumACContexts();
  const size_t num_hist = (p.num_histograms > 1 && num_groups > 1) ? std::min<size_t>(size_t(p.num_histograms), num_groups) : 1;
  const size_t num_passes = p.num_passes >= 2 && p.num_passes <= 3 ? size_t(p.num_passes) : 1;
  std::vector<std::vector<Token>> ac_tokens(num_groups * num_passes);  // [pass * num_groups + group]
  std::vector<std::vector<uint32_t>> natural(13);
  for (int s = 0; s < 27; s++)
    if (natural[jxh::kStrategyOrder[s]].empty()) jxh::NaturalOrder(s, &natural[jxh::kStrategyOrder[s]]);
  // coefficient orders actually used: the natural ones, or (custom_orders) natural[perm[k]] with a coded permutation
  // (coeff_order.cc:37-62,102-156: Lehmer code of the permutation, entries below the LLF count fixed)
  bool used_bucket[13] = {};
  for (uint8_t a : f.acs)
    if (a != 0xFF && (a & 1)) used_bucket[jxh::kStrategyOrder[a >> 1]] = true;
  uint32_t used_orders = 0;
  std::vector<std::vector<uint32_t>> scan(num_passes * 13 * 3);  // [(pass * 13 + ord) * 3 + c]
  std::vector<std::vector<Token>> perm_tokens(num_passes);
  auto order_ctx = [](uint32_t v) { return v == 0 ? 0u : std::min<uint32_t>(1u + uint32_t(jxh::FloorLog2(v)), 7u); };
  for (size_t pass = 0; pass < num_passes; pass++) {
    Rng prng(p.seed * 7919u + uint32_t(pass) + 17u);
    bool done[13] = {};
    for (int o = 0; o < 27; o++) {  // the decoder's bucket order: first strategy of each bucket
      const int ord = jxh::kStrategyOrder[o];
      if (done[ord]) continue;
      done[ord] = true;
      if (!used_bucket[ord]) continue;
      const size_t llf = size_t(jxh::kCoveredX[o]) * jxh::kCoveredY[o], size = 64 * llf;
      for (int c = 0; c < 3; c++) {
        std::vector<uint32_t>& sc = scan[(pass * 13 + ord) * 3 + c];
        sc = natural[ord];
        if (!p.custom_orders) continue;
        used_orders |= 1u << ord;
        const size_t win = std::min<size_t>(size - llf, 48);
        std::vector<uint32_t> perm(size);
        for (size_t k = 0; k < size; k++) perm[k] = uint32_t(k);
        for (size_t k = win; k > 1; k--) std::swap(perm[llf + k - 1], perm[llf + prng.Below(uint32_t(k))]);
        for (size_t k = 0; k < size; k++) sc[k] = natural[ord][perm[k]];
        // Lehmer code: how many later entries are smaller (all entries beyond the shuffled window are larger)
        std::vector<uint32_t> lehmer(llf + win, 0);
        size_t end = llf;
        for (size_t i = llf; i < llf + win; i++) {
          for (size_t j = i + 1; j < llf + win; j++) lehmer[i] += perm[j] < perm[i];
          if (lehmer[i]) end = i + 1;
        }
        perm_tokens[pass].push_back({order_ctx(uint32_t(size)), uint32_t(end - llf)});
        uint32_t last = 0;
        for (size_t i = llf; i < end; i++) {
          perm_tokens[pass].push_back({order_ctx(last), lehmer[i]});
          last = lehmer[i];
        }
      }
    }
  }
  std::vector<EncCode> perm_codes(num_passes);
  if (used_orders)
    for (size_t pass = 0; pass < num_passes; pass++) BuildCode({&perm_tokens[pass]}, 8, 8, cfg420, &perm_codes[pass]);
#pragma omp parallel for schedule(dynamic)
  for (size_t pg = 0; pg < num_groups * num_passes; pg++) {
    const size_t g = pg % num_groups, pass = pg / num_groups;
    if (!f.dev_tokens.empty()) {  // tokenised on the device: the same pairs, already in order
      const Token* t = reinterpret_cast<const Token*>(f.dev_tokens.data()) + f.dev_token_base[g];
      ac_tokens[pg].assign(t, t + f.dev_token_count[g]);
      continue;
    }
    // the part of a coefficient this pass carries (the decoder adds value << shift over the passes, dec_group.cc:335-338)
    auto part = [&](int32_t v) -> int32_t {
      const int sh = int(num_passes - 1 - pass);  // this pass's shift; the passes before carried v >> (sh + 1)
      return pass == 0 ? (v >> sh) : (v >> sh) - ((v >> (sh + 1)) << 1);
    };
    const size_t bx0 = (g % xg) * 32, by0 = (g / xg) * 32;
    const size_t gw = std::min<size_t>(32, f.xb - bx0), gh = std::min<size_t>(32, f.yb - by0);
    std::vector<int32_t> nzmap(3 * 1024, 0);
    std::vector<Token>& out_t = ac_tokens[pg];
    size_t offset = 0;
    for (size_t by_ = 0; by_ < gh; by_++)
      for (size_t bx_ = 0; bx_ < gw; bx_++) {
        const size_t bx = bx_, by = by_;
        uint8_t a = f.acs[(by0 + by) * f.xb + bx0 + bx];
        if (!(a & 1)) continue;
        const int st = a >> 1;
        const size_t cx = jxh::kCoveredX[st], cy = jxh::kCoveredY[st], log2c = jxh::kLog2Covered[st];
        const size_t covered = size_t(1) << log2c, size = covered * 64;
        const int ord = jxh::kStrategyOrder[st];
        const uint32_t qf = uint32_t(f.qf[(by0 + by) * f.xb + bx0 + bx]);
        static const int kOrder[3] = {1, 0, 2};
        for (int ci = 0; ci < 3; ci++) {
          const int c = kOrder[ci];
          if (!f.Present(c, bx0 + bx, by0 + by)) continue;  // (dec_group.cc:572-578: only the blocks a subsampled channel has)
          const int32_t* q = f.GroupCoeffs(g) + size_t(c) * 65536 + offset;
          int32_t* nzc = nzmap.data() + c * 1024;
          const size_t bx = bx_ >> f.hs[c], by = by_ >> f.vs[c];
          const int32_t* top = by ? nzc + (by - 1) * 32 : nullptr;
          int32_t* cur = nzc + by * 32;
          int32_t pred = bx == 0 ? (top ? top[0] : 32) : (!top ? cur[bx - 1] : (top[bx] + cur[bx - 1] + 1) / 2);
          const uint32_t* order = scan[(pass * 13 + ord) * 3 + c].data();
          size_t nz = 0;
          for (size_t k = covered; k < size; k++) nz += part(q[order[k]]) != 0;
          size_t bc = bctx.Context(dc_bucket((by0 + by_) * f.xb + bx0 + bx_), qf, ord, c);
          const size_t hist_off = (g % num_hist) * nctx;  // this group's histogram set
          out_t.push_back({uint32_t(hist_off + bctx.NonZeroContext(uint32_t(pred), bc)), uint32_t(nz)});
          for (size_t y = 0; y < cy; y++)
            for (size_t x = 0; x < cx; x++) cur[bx + x + y * 32] = int32_t((nz + covered - 1) >> log2c);
          const size_t hoff = bctx.ZeroDensityOffset(bc);
          size_t prev = nz > size / 16 ? 0 : 1, left = nz;
          for (size_t k = covered; k < size && left != 0; k++) {
            size_t ctx = hist_off + hoff + jxh::ZeroDensityContext(left, k, covered, log2c, prev);
            int32_t v = part(q[order[k]]);
            out_t.push_back({uint32_t(ctx), PackSigned(v)});
            prev = v != 0;
            left -= prev;
          }
        }
        offset += size;
      }
  }
  std::vector<EncCode> ac_codes(num_passes);
  for (size_t pass = 0; pass < num_passes; pass++) {
    std::vector<const std::vector<Token>*> all;
    for (size_t g = 0; g < num_groups; g++) all.push_back(&ac_tokens[pass * num_groups + g]);
    const int mode = p.ac_code_mode & 3;  // bit 0: prefix codes, bit 1: LZ77
    if (mode & 2)
      for (size_t g = 0; g < num_groups; g++) Lz77Pass(&ac_tokens[pass * num_groups + g], uint32_t(nctx * num_hist));
    BuildCode(all, nctx * num_hist + ((mode & 2) ? 1 : 0), p.max_clusters > 0 ? size_t(p.max_clusters) : 64, cfg420, &ac_codes[pass], mode);
  }
  // ---- sections
  auto write_dc_global = [&](BitWriter& bw) {
    if (!g_patches.empty()) WritePatches(bw, have_alpha ? 1 : 0);  // (dec_frame.cc:271-296: patches, splines, noise)
    if (!g_splines.empty()) WriteSplines(bw);
    if (p.noise > 0)  // NoiseParams: eight 10-bit LUT points (dec_noise.cc:154-164)
      for (int i = 0; i < 8; i++) bw.Write(10, uint32_t(std::min(1023, p.noise + 40 * i)));
    if (!p.custom_lf) {
      bw.Write(1, 1);  // DC dequant all_default
    } else {
      bw.Write(1, 0);
      WriteF16(bw, 0.03125f);   // X: step = value / 128 (default 1/4096)
      WriteF16(bw, 0.3125f);    // Y (default 1/512)
      WriteF16(bw, 0.4375f);    // B (default 1/256)
    }
    {
      static const uint32_t bits[4] = {11, 11, 12, 16}, offs[4] = {1, 2049, 4097, 8193};
      WriteU32Sel(bw, f.global_scale, bits, offs);
    }
    {
      static const uint32_t bits[4] = {0, 5, 8, 16}, offs[4] = {16, 1, 1, 1};
      WriteU32Sel(bw, f.quant_dc, bits, offs);
    }
    if (!p.custom_bctx) {
      bw.Write(1, 1);  // default block context map
    } else {
      bw.Write(1, 0);
      static const uint32_t tb[4] = {4, 8, 16, 32}, to[4] = {0, 16, 272, 65808};
      for (int j = 0; j < 3; j++) {
        bw.Write(4, bctx.dc_thresholds[j].size());
        for (int32_t t : bctx.dc_thresholds[j]) WriteU32Sel(bw, PackSigned(t), tb, to);
      }
      bw.Write(4, bctx.qf_thresholds.size());
      static const uint32_t qb[4] = {2, 3, 5, 8}, qo[4] = {0, 4, 12, 44};
      for (uint32_t t : bctx.qf_thresholds) WriteU32Sel(bw, t - 1, qb, qo);
      EncCode cm;
      cm.ctx_map = bctx.ctx_map;
      cm.num_clusters = bctx.num_ctxs;
      WriteContextMap(bw, cm);
    }
    if (!p.custom_cmap) {
      bw.Write(1, 1);  // default colour correlation
    } else {           // chroma_from_luma.h:112-140
      bw.Write(1, 0);
      bw.Write(2, 2);      // color_factor = 2 + 8 bits
      bw.Write(8, 98);     //   = 100
      bw.Write(16, 0x3400);  // base_correlation_x = 0.25 (F16)
      bw.Write(16, 0x3A00);  // base_correlation_b = 0.75
      bw.Write(8, 128 + 3);  // ytox_dc
      bw.Write(8, 128 - 5);  // ytob_dc
    }
    bw.Write(1, 1);  // has global tree
    WriteCodeHeader(bw, tree_code);
    WriteTokens(bw, tree_tokens.data(), tree_tokens.size(), tree_code);
    WriteCodeHeader(bw, mod_code);
    if (have_alpha) {  // stream 0 of the global Modular image: its group header, and the channel if it fits a group
      bw.Write(1, 1);  // use global tree
      bw.Write(1, 1);  // default weighted-predictor header
      bw.Write(2, 0);  // no transforms
      if (alpha_global) WriteTokens(bw, alpha_global_tokens.data(), alpha_global_tokens.size(), mod_code);
    }
  };
  auto write_group_header = [&](BitWriter& bw) {
    bw.Write(1, 1);  // use global tree
    bw.Write(1, 1);  // default weighted-predictor header
    bw.Write(2, 0);  // no transforms
  };
  auto write_dc_group = [&](BitWriter& bw, size_t g) {
    if (!g_use_dc_frame) {  // (dec_frame.cc:322-326: the DC stream is absent when the DC image comes from a DC frame)
      bw.Write(2, 0);  // extra precision
      write_group_header(bw);
      WriteTokens(bw, dc_tokens[g].data(), dc_tokens[g].size(), mod_code);
    }
    size_t bx0 = (g % xdg) * 256, by0 = (g / xdg) * 256;
    size_t bw_ = std::min<size_t>(256, f.xb - bx0), bh = std::min<size_t>(256, f.yb - by0);
    bw.Write(CeilLog2(bw_ * bh), meta_count[g] - 1);
    write_group_header(bw);
    WriteTokens(bw, meta_tokens[g].data(), meta_tokens[g].size(), mod_code);
  };
  auto write_ac_global = [&](BitWriter& bw) {
    if (!p.raw_quant) {
      bw.Write(1, 1);                          // default dequant tables
    } else {                                   // quant_weights.cc:497-511: all 17 encodings; the first one RAW, the others library
      bw.Write(1, 0);
      bw.Write(3, 7);
      WriteF16(bw, kRawQuantDen);
      bw.Write(1, 1);  // (GroupHeader) use global tree
      bw.Write(1, 1);  //   default weighted-predictor header
      bw.Write(2, 0);  //   no transforms
      WriteTokens(bw, raw_quant_tokens.data(), raw_quant_tokens.size(), mod_code);
      for (int k = 1; k < 17; k++) bw.Write(3, 0);
    }
    bw.Write(CeilLog2(num_groups), uint32_t(num_hist - 1));  // number of histogram sets - 1
    for (size_t pass = 0; pass < num_passes; pass++) {
      if (!used_orders) {
        bw.Write(2, 2);                        // used_orders = 0
      } else {
        bw.Write(2, 3);                        // used_orders: 13-bit mask of the order buckets with a coded permutation
        bw.Write(13, used_orders);
        WriteCodeHeader(bw, perm_codes[pass]);
        WriteTokens(bw, perm_tokens[pass].data(), perm_tokens[pass].size(), perm_codes[pass]);
      }
      WriteCodeHeader(bw, ac_codes[pass]);
    }
  };
  auto write_ac_group = [&](BitWriter& bw, size_t pg) {
    const size_t g = pg % num_groups;
    bw.Write(CeilLog2(num_hist), uint32_t(g % num_hist));  // histogram selector (dec_group.cc:594-610)
    WriteTokens(bw, ac_tokens[pg].data(), ac_tokens[pg].size(), ac_codes[pg / num_groups]);
    if (!alpha_group_tokens.empty() && alpha_group_present[g] && pg / num_groups == num_passes - 1) {  // Modular data of the group, behind the coefficients
      write_group_header(bw);
      WriteTokens(bw, alpha_group_tokens[g].data(), alpha_group_tokens[g].size(), mod_code);
    }
  };

  std::vector<std::vector<uint8_t>> sections;
  if (num_groups == 1 && num_passes == 1) {
    BitWriter bw;
    write_dc_global(bw);
    write_dc_group(bw, 0);
    write_ac_global(bw);
    write_ac_group(bw, 0);
    bw.ZeroPad();
    sections.push_back(bw.bytes());
  } else {
    sections.resize(2 + ndc + num_groups * num_passes);
    {
      BitWriter bw;
      write_dc_global(bw);
      bw.ZeroPad();
      sections[0] = bw.bytes();
    }
    for (size_t g = 0; g < ndc; g++) {
      BitWriter bw;
      write_dc_group(bw, g);
      bw.ZeroPad();
      sections[1 + g] = bw.bytes();
    }
    {
      BitWriter bw;
      write_ac_global(bw);
      bw.ZeroPad();
      sections[1 + ndc] = bw.bytes();
    }
#pragma omp parallel for schedule(dynamic)
    for (size_t pg = 0; pg < num_groups * num_passes; pg++) {  // TOC order: pass-major (toc.h: AcGroupIndex)
      BitWriter bw;
      write_ac_group(bw, pg);
      bw.ZeroPad();
      sections[2 + ndc + pg] = bw.bytes();
    }
  }
  // ---- headers
  BitWriter bw;
  bw.Write(16, 0x0AFF);
  bw.Write(1, 0);  // not "small"
  const uint32_t ups = (p.upsampling == 2 || p.upsampling == 4 || p.upsampling == 8) ? uint32_t(p.upsampling) : 1;
  // the image is the frame times the upsampling factor (the frame is ceil(image / factor): any image size in
  // (factor * (frame - 1), factor * frame] is valid; img_xs / img_ys pick one)
  WriteSizeDim(bw, g_image_h ? g_image_h : uint32_t(ups == 1 ? f.ys : f.img_ys));
  bw.Write(3, 0);  // no aspect-ratio shortcut
  WriteSizeDim(bw, g_image_w ? g_image_w : uint32_t(ups == 1 ? f.xs : f.img_xs));
  const bool with_icc = !g_embedded_icc.empty();
  if (!have_alpha && !with_icc && !ExtraFields() && !p.color_transform) {
    bw.Write(1, 1);  // ImageMetadata all_default (8-bit sRGB, XYB encoded)
  } else {           // image_metadata.cc:283-356
    bw.Write(1, 0);  // not all_default
    WriteExtraFields(bw);
    bw.Write(1, 0);  // integer samples
    bw.Write(2, 0);  //   8 bits
    bw.Write(1, 1);  // modular_16_bit_buffer_sufficient
    bw.Write(2, have_alpha ? 1 : 0);  // extra channels
    if (have_alpha) WriteAlphaChannelInfo(bw);
    bw.Write(1, p.color_transform ? 0 : 1);  // xyb_encoded
    if (!with_icc) {
      bw.Write(1, 1);  // ColorEncoding all_default (sRGB)
    } else {
      bw.Write(1, 0);  // ColorEncoding not all_default (color_encoding_internal.cc:144-158)
      bw.Write(1, 1);  //   want_icc
      bw.Write(2, 0);  //   colour space RGB
    }
    WriteToneMapping(bw);
    bw.Write(2, 0);  // no extensions
  }
  if (!g_custom_ups_mask) {
    bw.Write(1, 1);  // CustomTransformData all_default
  } else {
    bw.Write(1, 0);
    if (!p.color_transform) bw.Write(1, 1);  // OpsinInverseMatrix all_default (only coded when the image is XYB encoded)
    bw.Write(3, g_custom_ups_mask & 7);
    uint32_t st = g_custom_ups_seed * 2654435761u + 12345u;
    const float* defaults[3] = {kUpsamplingWeights2, kUpsamplingWeights4, kUpsamplingWeights8};
    const int counts[3] = {15, 55, 210};
    for (int k = 0; k < 3; k++)
      if (g_custom_ups_mask & (1u << k))
        for (int i = 0; i < counts[k]; i++) {
          st = st * 1664525u + 1013904223u;
          WriteF16(bw, RoundToF16(defaults[k][i] * (0.75f + 0.5f * float(st >> 8) * (1.0f / 16777216.0f))));
        }
  }
  if (with_icc) AppendEmbeddedIcc(bw);  // (decode.cc: after the transform data, before the byte boundary)
  bw.ZeroPad();
  g_last_header_bytes = bw.bytes().size();
  // FrameHeader
  bw.Write(1, 0);  // not all_default
  bw.Write(2, g_dc_frame_level > 0 ? 1 : (g_reference_slot >= 0 ? 2 : 0));  // regular frame, kDCFrame or kReferenceOnly
  bw.Write(1, 0);  // VarDCT
  // (dec_frame.cc:206-212: a subsampled frame must not ask for adaptive DC smoothing)
  const uint64_t hflags = f.flags | (g_use_dc_frame ? 32 : 0) | (f.Subsampled() ? 128 : 0);
  if (hflags == 0) {
    bw.Write(2, 0);
  } else if (hflags <= 16) {  // U64 selector 1: 1 + 4 bits
    bw.Write(2, 1);
    bw.Write(4, hflags - 1);
  } else {  // U64 selector 2: 17 + 8 bits
    bw.Write(2, 2);
    bw.Write(8, hflags - 17);
  }
  if (p.color_transform) {  // (frame_header.cc:247-258: only images that are not xyb_encoded say whether the frame is YCbCr)
    bw.Write(1, p.color_transform == 2 ? 1 : 0);
    if (p.color_transform == 2 && !g_use_dc_frame)
      for (int c = 0; c < 3; c++) bw.Write(2, f.cmode[c]);  // YCbCrChromaSubsampling (all 0: 4:4:4)
  }
  if (!g_use_dc_frame) {  // (frame_header.cc:263: no upsampling fields with kUseDcFrame)
    bw.Write(2, ups == 1 ? 0 : (ups == 2 ? 1 : (ups == 4 ? 2 : 3)));  // upsampling factor
    if (have_alpha) bw.Write(2, f.alpha_ups == 8 ? 3 : (f.alpha_ups == 4 ? 2 : (f.alpha_ups == 2 ? 1 : 0)));  // extra channel upsampling
  }
  if (!p.color_transform) {  // (frame_header.cc:287-291: an image that is not xyb_encoded has both at 2)
    bw.Write(3, p.custom_cmap ? 2 : 3);  // x_qm_scale
    bw.Write(3, p.custom_cmap ? 4 : 2);  // b_qm_scale
  }
  if (g_reference_slot >= 0) {
    // (frame_header.cc:303: a kReferenceOnly frame has no Passes bundle)
  } else if (num_passes == 1) {
    bw.Write(2, 0);  // one pass
  } else if (num_passes == 2) {
    bw.Write(2, 1);  // two passes
    bw.Write(2, 0);  // no downsampling brackets
    bw.Write(2, 1);  // shift of pass 0 = 1 (the last pass always has shift 0)
  } else {
    bw.Write(2, 2);  // three passes
    bw.Write(2, 1);  // one downsampling bracket (frame_header.h:299-309)
    bw.Write(2, 2);  // shift of pass 0 = 2
    bw.Write(2, 1);  // shift of pass 1 = 1
    bw.Write(2, 2);  // downsample[0] = 4 ...
    bw.Write(2, 0);  // ... is reached with last_pass[0] = 0
  }
  if (g_dc_frame_level > 0) bw.Write(2, uint32_t(g_dc_frame_level - 1));  // dc_level: U32(Val(1), Val(2), Val(3), Val(4))
  const bool whole = WriteCropAndBlending(bw, uint32_t(f.xs), uint32_t(f.ys), have_alpha);
  WriteFrameTiming(bw, whole);
  WriteFrameName(bw);
  bw.Write(1, 0);  // loop filter not all_default
  bw.Write(1, f.gab ? 1 : 0);
  if (f.gab) {
    if (!p.custom_lf) {
      bw.Write(1, 0);  // default gaborish weights
    } else {
      bw.Write(1, 1);
      static const float w[6] = {0.125f, 0.046875f, 0.09375f, 0.0625f, 0.109375f, 0.03125f};
      for (float v : w) WriteF16(bw, v);
    }
  }
  bw.Write(2, f.epf_iters);
  if (f.epf_iters > 0) {
    if (!p.custom_lf) {
      bw.Write(1, 0);  // default sharpness LUT
      bw.Write(1, 0);  // default channel weights
      bw.Write(1, 0);  // default sigma parameters
    } else {
      bw.Write(1, 1);
      static const float lut[8] = {0.0f, 0.25f, 0.375f, 0.5f, 0.625f, 0.75f, 0.875f, 1.0f};
      for (float v : lut) WriteF16(bw, v);
      bw.Write(1, 1);
      WriteF16(bw, 32.0f);   // channel scales
      WriteF16(bw, 6.0f);
      WriteF16(bw, 2.5f);
      WriteF16(bw, 0.5f);    // pass1 / pass2 zero-flush (parsed, unused by the decoder)
      WriteF16(bw, 0.25f);
      bw.Write(1, 1);
      WriteF16(bw, 0.5f);    // quant_mul
      WriteF16(bw, 0.75f);   // pass0 sigma scale
      WriteF16(bw, 5.0f);    // pass2 sigma scale
      WriteF16(bw, 0.5f);    // border SAD multiplier
    }
  }
  bw.Write(2, 0);  // no loop-filter extensions
  bw.Write(2, 0);  // no frame-header extensions
  // TOC
  bw.Write(1, 0);  // not permuted
  bw.ZeroPad();
  for (const auto& s : sections) {
    static const uint32_t bits[4] = {10, 14, 22, 30}, offs[4] = {0, 1024, 17408, 4211712};
    WriteU32Sel(bw, uint32_t(s.size()), bits, offs);
  }
  bw.ZeroPad();
  *out = bw.bytes();
  for (const auto& s : sections) out->insert(out->end(), s.begin(), s.end());
}

// ---------------------------------------------------------------- image mode
static void QuantParams(float distance, FrameModel* f, float* quant_ac) {
  const float kAcQuant = 0.765f, kDcQuant = 1.095924047623553f, kDcMul = 0.3f, kDcQuantPow = 0.83f;
  float target_dc = std::max(0.5f * distance, std::min(distance, kDcMul * std::pow((1.0f / kDcMul) * distance, kDcQuantPow)));
  float qdc = std::min(kDcQuant / target_dc, 50.0f);
  float qac = kAcQuant / distance;
  float scale = 65536.0f * qac / 5.0f;
  scale = std::max(1.0f, std::min(32768.0f, scale));
  int gs = int(scale);
  int scaled_qdc = int(qdc * 4096 * 1.6);
  if (gs > scaled_qdc) gs = std::max(1, scaled_qdc);
  f->global_scale = uint32_t(gs);
  float inv_gs = 65536.0f / float(gs);
  f->quant_dc = uint32_t(std::min<float>(65536.0f, qdc * inv_gs + 0.5f));
  if (f->quant_dc < 1) f->quant_dc = 1;
  *quant_ac = qac;
}

// The pixel-domain half of the encode done elsewhere (the HIP forward path, jxlhip_enc_forward of include/jxl_amd_hip.h,
// whose signature this is): the caller hands the function and its context over, this library does not link against it.
typedef int (*ForwardFn)(void* ctx, const uint8_t* rgb, size_t stride, const JxlHipEncDesc* desc, uint8_t* acs, int32_t* qf, int32_t* dc,
                         int32_t* coeffs);
// ... and the tokenisation of its coefficients (jxlhip_enc_token_counts / jxlhip_enc_tokens): when both are given the
// coefficients stay on the device and the host entropy coder starts from the tokens.
typedef int (*TokenCountsFn)(void* ctx, const JxlHipEncTokDesc* desc, uint32_t* totals);
typedef int (*TokensFn)(void* ctx, const uint32_t* bases, uint32_t* tokens, size_t capacity);
struct ForwardHook {
  ForwardFn fn;
  void* ctx;
  double seconds[2];  // out: forward call, assembly (entropy coding + headers)
  TokenCountsFn tok_counts = nullptr;
  TokensFn tok_emit = nullptr;
  uint64_t device_tokens = 0;  // out: how many tokens came from the device (0: the host tokenised)
  // test access: when set, the raw outputs of the forward call are copied here and nothing is assembled
  uint8_t* cap_acs = nullptr;
  int32_t *cap_qf = nullptr, *cap_dc = nullptr, *cap_coeffs = nullptr;
};

// (jxlenc_encode_rgba8 -> EncodeImage: width, height, upsampling factor and shift of a subsampled alpha plane; 0 = frame sized)
static thread_local uint32_t g_alpha_dims[4] = {0, 0, 0, 0};
// model_only: stop before the bitstream assembly and hand the frame model out (the CPU form of the forward path).
static void EncodeImage(const uint8_t* rgb, size_t xs, size_t ys, const Params& p, std::vector<uint8_t>* out, size_t img_xs = 0,
                        size_t img_ys = 0, const std::vector<uint8_t>* alpha = nullptr, ForwardHook* hook = nullptr,
                        FrameModel* model_only = nullptr) {
  FrameModel f;
  if (alpha) {
    f.alpha = *alpha;
    if (g_alpha_dims[0]) {
      f.alpha_w = g_alpha_dims[0];
      f.alpha_h = g_alpha_dims[1];
      f.alpha_ups = g_alpha_dims[2];
      f.alpha_shift = g_alpha_dims[3];
    }
  }
  f.SetSubsampling(p.color_transform == 2 ? uint32_t(p.chroma_subsampling) : 0u);
  f.SetSize(xs, ys);
  if (hook && f.Subsampled()) throw std::runtime_error("forward hook: unsupported parameters");
  f.img_xs = img_xs ? img_xs : xs;
  f.img_ys = img_ys ? img_ys : ys;
  const size_t xp = f.xb * 8, yp = f.yb * 8;
  float quant_ac;
  QuantParams(p.distance, &f, &quant_ac);
  f.gab = p.gab < 0 ? 1 : p.gab;
  if (hook) {
    // image mode with the default colour correlation only: what the device path implements
    if (p.strategy_mode > 1 || p.random_cmap || p.custom_cmap) throw std::runtime_error("forward hook: unsupported parameters");
    const double t0 = NowSeconds();
    f.epf_iters = p.epf_iters >= 0 ? p.epf_iters : (p.distance >= 4.0f ? 3 : p.distance >= 1.5f ? 2 : p.distance >= 0.7f ? 1 : 0);
    f.flags = (p.skip_dc_smoothing ? 128 : 0) | (p.noise > 0 ? 1 : 0) | (g_splines.empty() ? 0 : 16) | (g_patches.empty() ? 0 : 2);
    f.sharp.assign(f.xb * f.yb, 4);
    f.ytox.assign(DivCeil(f.xb, 8) * DivCeil(f.yb, 8), 0);
    f.ytob.assign(f.ytox.size(), 0);
    // the default dequantisation tables, flat, built once (they do not depend on the image)
    static std::vector<float> flat;
    static uint32_t flat_offset[17], flat_size[17];
    static std::once_flag flat_once;
    std::call_once(flat_once, [] {
      jxh::DequantTables dq;
      for (int k = 0; k < 17; k++) {
        flat_offset[k] = flat_size[k] = 0;
        if (k >= 13) continue;  // 128 / 256 point tables: never selected here, not built
        dq.Compute(k);
        flat_offset[k] = uint32_t(flat.size());
        flat_size[k] = uint32_t(dq.table[k].size() / 3);
        flat.insert(flat.end(), dq.table[k].begin(), dq.table[k].end());
      }
    });
    JxlHipEncDesc d;
    memset(&d, 0, sizeof(d));
    memcpy(d.dequant_offset, flat_offset, sizeof(flat_offset));
    memcpy(d.dequant_size, flat_size, sizeof(flat_size));
    d.dequant = flat.data();
    d.dequant_floats = uint32_t(flat.size());
    d.xsize = uint32_t(xs);
    d.ysize = uint32_t(ys);
    d.distance = p.distance;
    d.gaborish = f.gab ? 1 : 0;
    d.strategy_mode = uint32_t(p.strategy_mode);
    d.global_scale = f.global_scale;
    d.quant_dc = f.quant_dc;
    d.quant_ac = quant_ac;
    d.cfl_fit = p.cfl_fit ? 1 : 0;
    d.ytox = f.ytox.data();
    d.ytob = f.ytob.data();
    const size_t nb = f.xb * f.yb, ng = DivCeil(xs, 256) * DivCeil(ys, 256);
    f.acs.assign(nb, 0);
    f.qf.assign(nb, 0);
    // (no zero fill and no second copy of the coefficients: 100 MB at 4K)
    std::vector<int32_t> dc(3 * nb);
    // device tokenisation: one pass, natural orders, the default block context map (what it is written for)
    const bool dev_tok = hook->tok_counts && hook->tok_emit && !hook->cap_acs && !(p.num_passes >= 2 && p.num_passes <= 3) && !p.custom_orders && !p.custom_bctx;
    if (!dev_tok) f.flat_coeffs.reset(new int32_t[ng * 3 * 65536]);
    int32_t* const co = f.flat_coeffs.get();
    const int r = hook->fn(hook->ctx, rgb, xs * 3, &d, f.acs.data(), f.qf.data(), dc.data(), co);
    if (r) throw std::runtime_error("forward hook failed (" + std::to_string(r) + ")");
    if (dev_tok) {
      JxlHipEncTokDesc td;
      memset(&td, 0, sizeof(td));
      std::vector<uint16_t> orders;
      bool have[13] = {};
      for (int s2 = 0; s2 < 27; s2++) {
        const int ord = jxh::kStrategyOrder[s2];
        if (have[ord]) continue;
        have[ord] = true;
        td.order_offset[ord] = uint32_t(orders.size());
        if (size_t(jxh::kCoveredX[s2]) * jxh::kCoveredY[s2] > 64) continue;  // (the forward path selects up to 64x64)
        std::vector<uint32_t> nat;
        jxh::NaturalOrder(s2, &nat);
        for (uint32_t v : nat) orders.push_back(uint16_t(v));
      }
      td.orders = orders.data();
      td.orders_size = uint32_t(orders.size());
      const jxh::BlockCtxMap bctx;
      memcpy(td.ctx_map, bctx.ctx_map.data(), sizeof(td.ctx_map));
      td.num_ctxs = uint32_t(bctx.num_ctxs);
      td.num_hist = uint32_t((p.num_histograms > 1 && ng > 1) ? std::min<size_t>(size_t(p.num_histograms), ng) : 1);
      f.dev_token_count.assign(ng, 0);
      int tr = hook->tok_counts(hook->ctx, &td, f.dev_token_count.data());
      if (tr) throw std::runtime_error("token hook failed (" + std::to_string(tr) + ")");
      f.dev_token_base.assign(ng, 0);
      uint64_t total = 0;
      for (size_t g = 0; g < ng; g++) {
        f.dev_token_base[g] = uint32_t(total);
        total += f.dev_token_count[g];
      }
      if (total >= (uint64_t(1) << 32)) throw std::runtime_error("token hook: too many tokens");
      f.dev_tokens.assign(size_t(total) + 1, 0);
      tr = hook->tok_emit(hook->ctx, f.dev_token_base.data(), reinterpret_cast<uint32_t*>(f.dev_tokens.data()), size_t(total));
      if (tr) throw std::runtime_error("token hook failed (" + std::to_string(tr) + ")");
      hook->device_tokens = total;
    }
    if (hook->cap_acs) {
      memcpy(hook->cap_acs, f.acs.data(), nb);
      memcpy(hook->cap_qf, f.qf.data(), nb * 4);
      memcpy(hook->cap_dc, dc.data(), dc.size() * 4);
      memcpy(hook->cap_coeffs, co, ng * 3 * 65536 * 4);
      return;
    }
    for (int c = 0; c < 3; c++) f.dc[c].assign(dc.begin() + c * nb, dc.begin() + (c + 1) * nb);
    const double t1 = NowSeconds();
    Assemble(f, p, out);
    hook->seconds[0] = t1 - t0;
    hook->seconds[1] = NowSeconds() - t1;
    return;
  }
  std::vector<float> xyb[3];
  if (p.color_transform) RgbToPlain(rgb, xs, ys, xp, yp, p.color_transform == 2, xyb);
  else RgbToXyb(rgb, xs, ys, xp, yp, xyb);
  if (f.gab) {
    // Approximate inverse of the decoder's Gaborish blur K (3x3, default weights): y <- y + (x - K*y), 4 rounds.
    const float w1 = 1.1f * 0.104699568f, w2 = 1.1f * 0.055680538f, nrm = 1.0f / (1.0f + 4 * (w1 + w2));
    for (int c = 0; c < 3; c++) {
      std::vector<float> y = xyb[c], t(xp * yp);
      for (int it = 0; it < 4; it++) {
#pragma omp parallel for
        for (size_t yy = 0; yy < yp; yy++) {
          size_t y0 = yy ? yy - 1 : 0, y1 = yy + 1 < yp ? yy + 1 : yp - 1;
          for (size_t xx = 0; xx < xp; xx++) {
            size_t x0 = xx ? xx - 1 : 0, x1 = xx + 1 < xp ? xx + 1 : xp - 1;
            float side = y[yy * xp + x0] + y[yy * xp + x1] + y[y0 * xp + xx] + y[y1 * xp + xx];
            float corner = y[y0 * xp + x0] + y[y0 * xp + x1] + y[y1 * xp + x0] + y[y1 * xp + x1];
            float blur = (y[yy * xp + xx] + w1 * side + w2 * corner) * nrm;
            t[yy * xp + xx] = y[yy * xp + xx] + (xyb[c][yy * xp + xx] - blur);
          }
        }
        y.swap(t);
      }
      xyb[c].swap(y);
    }
  }
  if (f.Subsampled()) {
    // a subsampled channel: the mean of each 2x1 / 1x2 / 2x2 cell, kept in the top-left part of its plane (same row stride)
    for (int c = 0; c < 3; c++) {
      if (!f.hs[c] && !f.vs[c]) continue;
      const size_t sw = xp >> f.hs[c], sh = yp >> f.vs[c], nx = size_t(1) << f.hs[c], ny = size_t(1) << f.vs[c];
      std::vector<float> sub(xp * yp, 0.0f);
      for (size_t y = 0; y < sh; y++)
        for (size_t x = 0; x < sw; x++) {
          float sum = 0;
          for (size_t dy = 0; dy < ny; dy++)
            for (size_t dx = 0; dx < nx; dx++) sum += xyb[c][(y * ny + dy) * xp + x * nx + dx];
          sub[y * xp + x] = sum / float(nx * ny);
        }
      xyb[c].swap(sub);
    }
  }
  f.epf_iters = p.epf_iters >= 0 ? p.epf_iters : (p.distance >= 4.0f ? 3 : p.distance >= 1.5f ? 2 : p.distance >= 0.7f ? 1 : 0);
  f.flags = (p.skip_dc_smoothing ? 128 : 0) | (p.noise > 0 ? 1 : 0) | (g_splines.empty() ? 0 : 16) | (g_patches.empty() ? 0 : 2);
  f.acs.assign(f.xb * f.yb, 0xFF);
  f.qf.assign(f.xb * f.yb, 0);
  f.sharp.assign(f.xb * f.yb, 4);
  f.ytox.assign(DivCeil(f.xb, 8) * DivCeil(f.yb, 8), 0);
  f.ytob.assign(f.ytox.size(), 0);
  Rng rng(p.seed + 12345);
  if (p.random_cmap)
    for (size_t i = 0; i < f.ytox.size(); i++) {
      f.ytox[i] = int8_t(int(rng.Below(17)) - 8);
      f.ytob[i] = int8_t(int(rng.Below(17)) - 8);
    }
  for (auto& d : f.dc) d.assign(f.xb * f.yb, 0);
  // per-block activity of Y (mean abs deviation from the block mean)
  std::vector<float> act(f.xb * f.yb);
#pragma omp parallel for
  for (size_t by = 0; by < f.yb; by++)
    for (size_t bx = 0; bx < f.xb; bx++) {
      const float* s = xyb[1].data() + by * 8 * xp + bx * 8;
      float mean = 0;
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) mean += s[y * xp + x];
      mean /= 64;
      float a = 0;
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) a += std::fabs(s[y * xp + x] - mean);
      act[by * f.xb + bx] = a / 64;
    }
  auto region_max = [&](size_t bx, size_t by, size_t w, size_t h) {
    float m = 0;
    for (size_t y = 0; y < h; y++)
      for (size_t x = 0; x < w; x++) m = std::max(m, act[(by + y) * f.xb + bx + x]);
    return m;
  };
  // strategy selection
  const float T64 = 0.004f * p.distance, T32 = 0.008f * p.distance, T16 = 0.016f * p.distance, TR = 0.011f * p.distance;
  uint32_t mask = p.strategy_mask ? p.strategy_mask : 0x7FFFFFFu;
  for (size_t by = 0; by < f.yb; by++)
    for (size_t bx = 0; bx < f.xb; bx++) {
      if (f.acs[by * f.xb + bx] != 0xFF) continue;
      int st = 0;
      if (p.strategy_mode == 2) {
        // uniformly random strategy among those allowed that fit; restricted to the DCT family in image mode
        static const int kFam[18] = {0, 4, 5, 6, 7, 8, 9, 10, 11, 18, 19, 20, 21, 22, 23, 24, 25, 26};
        for (int tries = 0; tries < 8; tries++) {
          int cand = kFam[rng.Below(18)];
          if (!(mask & (1u << cand))) continue;
          size_t cx = jxh::kCoveredX[cand], cy = jxh::kCoveredY[cand];
          if (bx % cx || by % cy) continue;
          if (f.Subsampled() && cx * cy != 1) continue;  // (dec_modular.cc:534-538)
          if (Fits(f, bx, by, cand)) {
            st = cand;
            break;
          }
        }
      } else if (p.strategy_mode == 1 && !f.Subsampled()) {
        auto ok = [&](int cand, float thr) {
          size_t cx = jxh::kCoveredX[cand], cy = jxh::kCoveredY[cand];
          return bx % cx == 0 && by % cy == 0 && Fits(f, bx, by, cand) && region_max(bx, by, cx, cy) < thr;
        };
        if (ok(18, T64)) st = 18;
        else if (ok(20, T64 * 1.3f)) st = 20;  // 32x64
        else if (ok(19, T64 * 1.3f)) st = 19;  // 64x32
        else if (ok(5, T32)) st = 5;
        else if (ok(11, TR)) st = 11;  // 16x32
        else if (ok(10, TR)) st = 10;  // 32x16
        else if (ok(4, T16)) st = 4;
        else if (ok(9, T16 * 0.8f)) st = 9;   // 8x32
        else if (ok(8, T16 * 0.8f)) st = 8;   // 32x8
        else if (ok(7, T16 * 1.5f)) st = 7;   // 8x16
        else if (ok(6, T16 * 1.5f)) st = 6;   // 16x8
      }
      Place(f, bx, by, st);
    }
  // adaptive quant field
  for (size_t by = 0; by < f.yb; by++)
    for (size_t bx = 0; bx < f.xb; bx++) {
      uint8_t a = f.acs[by * f.xb + bx];
      if (!(a & 1)) continue;
      int st = a >> 1;
      float m = region_max(bx, by, jxh::kCoveredX[st], jxh::kCoveredY[st]);
      float mul = 1.35f - 0.12f * std::log2(1.0f + m * 400.0f);
      mul = std::max(0.8f, std::min(1.4f, mul));
      float inv_gs = 65536.0f / float(f.global_scale);
      int q = int(quant_ac * mul * inv_gs + 0.5f);
      f.qf[by * f.xb + bx] = std::max(1, std::min(256, q));
    }
  if (p.cfl_fit && !p.random_cmap && !f.Subsampled()) {
    // enc_chroma_from_luma.cc:204-352 ComputeTile + :128-151 FindBestMultiplier (fast): over the AC coefficients of the
    // tile's transforms (lowest frequencies excluded), weighted by the inverse quantisation matrix of the chroma channel
    // and q = Scale() * 128 * quant field, the least-squares factor of chroma against luma, pulled 2.6 towards zero.
    jxh::DequantTables dqc;
    for (int s2 = 0; s2 < 27; s2++) dqc.Matrix(s2, 0);
    const float scale = float(f.global_scale) / 65536.0f;
    const size_t tx_n = DivCeil(f.xb, 8), ty_n = DivCeil(f.yb, 8);
#pragma omp parallel for schedule(dynamic)
    for (size_t t = 0; t < tx_n * ty_n; t++) {
      const size_t bx0 = (t % tx_n) * 8, by0 = (t / tx_n) * 8;
      double sa2[2] = {0, 0}, sab[2] = {0, 0};
      size_t num = 0;
      std::vector<float> cy, cxv, cb, tmp;
      for (size_t by = by0; by < std::min(by0 + 8, f.yb); by++)
        for (size_t bx = bx0; bx < std::min(bx0 + 8, f.xb); bx++) {
          const uint8_t a = f.acs[by * f.xb + bx];
          if (!(a & 1)) continue;
          const int st = a >> 1, cx = jxh::kCoveredX[st], cyb = jxh::kCoveredY[st], R = cyb * 8, C = cx * 8;
          const size_t size = size_t(R) * C, cstride = size_t(std::max(cx, cyb)) * 8;
          const size_t lrows = size_t(std::min(cx, cyb)), lcols = size_t(std::max(cx, cyb));
          cy.resize(size); cxv.resize(size); cb.resize(size);
          ForwardDct(xyb[1].data() + by * 8 * xp + bx * 8, xp, R, C, cy.data(), tmp);
          ForwardDct(xyb[0].data() + by * 8 * xp + bx * 8, xp, R, C, cxv.data(), tmp);
          ForwardDct(xyb[2].data() + by * 8 * xp + bx * 8, xp, R, C, cb.data(), tmp);
          const float q = scale * 128.0f * float(f.qf[by * f.xb + bx]);
          const float *mx = dqc.Matrix(st, 0), *mb = dqc.Matrix(st, 2);
          for (size_t k = 0; k < size; k++) {
            if (k / cstride < lrows && k % cstride < lcols) continue;
            const float wx = q / mx[k], wb = q / mb[k];
            const float ax = (1.0f / 84) * (cy[k] * wx), bxv = 0.0f * (cy[k] * wx) - cxv[k] * wx;
            const float ab = (1.0f / 84) * (cy[k] * wb), bbv = 1.0f * (cy[k] * wb) - cb[k] * wb;
            sa2[0] += double(ax) * ax; sab[0] += double(ax) * bxv;
            sa2[1] += double(ab) * ab; sab[1] += double(ab) * bbv;
          }
          num += size;
        }
      for (int c = 0; c < 2; c++) {
        float x = num ? float(-sab[c] / (sa2[c] + double(num) * 1e-9 * 0.5)) : 0.0f;
        x = x >= 2.6f ? x - 2.6f : (x <= -2.6f ? x + 2.6f : 0.0f);
        const int8_t v = int8_t(std::max(-128.0f, std::min(127.0f, std::round(x))));
        (c == 0 ? f.ytox : f.ytob)[t] = v;
      }
    }
  }
  // transform + quantise per group
  const size_t xg = DivCeil(xs, 256), yg = DivCeil(ys, 256);
  f.coeffs.assign(xg * yg, {});
  jxh::DequantTables dq;
  if (p.raw_quant) {
    dq.enc[0] = jxh::QuantEncoding();
    dq.enc[0].mode = 7;
    dq.enc[0].qraw_den = kRawQuantDen;
    dq.enc[0].qraw = RawQuantTable();
  }
  for (int s = 0; s < 27; s++) dq.Matrix(s, 0);  // precompute (not thread-safe lazily)
  const float inv_gs = 65536.0f / float(f.global_scale);
  // (x_qm_scale 3, b_qm_scale 2 in the frame header; an image that is not xyb_encoded codes neither: both are 2)
  const float x_dm = p.color_transform ? 1.0f : std::pow(1.25f, 2.0f - 3.0f), b_dm = std::pow(1.25f, 2.0f - 2.0f);
  const float inv_quant_dc = inv_gs / float(f.quant_dc);
  const float dc_step[3] = {inv_quant_dc / 4096.0f, inv_quant_dc / 512.0f, inv_quant_dc / 256.0f};
  const size_t tiles_x = DivCeil(f.xb, 8);
  const float biases[4] = {1.0f - 0.05465007330715401f, 1.0f - 0.07005449891748593f, 1.0f - 0.049935103337343655f, 0.145f};
#pragma omp parallel for schedule(dynamic)
  for (size_t g = 0; g < xg * yg; g++) {
    std::vector<int32_t>& co = f.coeffs[g];
    co.assign(3 * 65536, 0);
    const size_t bx0 = (g % xg) * 32, by0 = (g / xg) * 32;
    const size_t gw = std::min<size_t>(32, f.xb - bx0), gh = std::min<size_t>(32, f.yb - by0);
    std::vector<float> coef[3], tmp, llf, dcb;
    size_t offset = 0;
    for (size_t by = 0; by < gh; by++)
      for (size_t bx = 0; bx < gw; bx++) {
        const size_t abx = bx0 + bx, aby = by0 + by;
        uint8_t a = f.acs[aby * f.xb + abx];
        if (!(a & 1)) continue;
        const int st = a >> 1;
        const int cx = jxh::kCoveredX[st], cy = jxh::kCoveredY[st];
        const int R = cy * 8, C = cx * 8;
        const size_t size = size_t(R) * C;
        const size_t cstride = size_t(std::max(cx, cy)) * 8;  // coefficient columns
        const bool present[3] = {f.Present(0, abx, aby), f.Present(1, abx, aby), f.Present(2, abx, aby)};
        for (int c = 0; c < 3; c++) {
          coef[c].assign(size, 0.0f);
          if (present[c]) ForwardDct(xyb[c].data() + (aby >> f.vs[c]) * 8 * xp + (abx >> f.hs[c]) * 8, xp, R, C, coef[c].data(), tmp);
        }
        // DC samples of the covered blocks from the LLF corner (inverse of LowestFrequenciesFromDC)
        float dcv[3][32 * 32];
        for (int c = 0; c < 3; c++) {
          llf.assign(size_t(cx) * cy, 0.0f);
          for (int ky = 0; ky < cy; ky++)
            for (int kx = 0; kx < cx; kx++) {
              float cf = (R < C) ? coef[c][ky * cstride + kx] : coef[c][kx * cstride + ky];
              llf[ky * cx + kx] = cf / (ResampleScale(cy, ky) * ResampleScale(cx, kx));
            }
          // scaled IDCT of size cy x cx (natural ky,kx layout here)
          const float* brr = GetBasis().m[FloorLog2(cy)].data();
          const float* bcc = GetBasis().m[FloorLog2(cx)].data();
          for (int y = 0; y < cy; y++)
            for (int x = 0; x < cx; x++) {
              float s = 0;
              for (int ky = 0; ky < cy; ky++)
                for (int kx = 0; kx < cx; kx++) s += llf[ky * cx + kx] * brr[y * cy + ky] * bcc[x * cx + kx];
              dcv[c][y * cx + x] = s;
            }
        }
        if (f.Subsampled()) {  // (compressed_dc.cc:230-250: no chroma from luma on the DC of such a frame)
          for (int c = 0; c < 3; c++)
            if (present[c]) f.dc[c][(aby >> f.vs[c]) * f.xb + (abx >> f.hs[c])] = int32_t(std::lround(dcv[c][0] / dc_step[c]));
        } else
        for (int y = 0; y < cy; y++)
          for (int x = 0; x < cx; x++) {
            size_t bi = (aby + y) * f.xb + abx + x;
            int32_t qy = int32_t(std::lround(dcv[1][y * cx + x] / dc_step[1]));
            float yd = qy * dc_step[1];
            f.dc[1][bi] = qy;
            f.dc[0][bi] = int32_t(std::lround((dcv[0][y * cx + x] - 0.0f * yd) / dc_step[0]));
            f.dc[2][bi] = int32_t(std::lround((dcv[2][y * cx + x] - 1.0f * yd) / dc_step[2]));
          }
        // AC quantisation (Y first: X and B are coded as residuals of the chroma-from-luma prediction)
        const float scaled = inv_gs / float(f.qf[aby * f.xb + abx]);
        const float mulc[3] = {scaled * x_dm, scaled, scaled * b_dm};
        const float x_cc = 0.0f + float(f.ytox[(aby / 8) * tiles_x + abx / 8]) / 84.0f;
        const float b_cc = 1.0f + float(f.ytob[(aby / 8) * tiles_x + abx / 8]) / 84.0f;
        const float *my = dq.Matrix(st, 1), *mx = dq.Matrix(st, 0), *mb = dq.Matrix(st, 2);
        int32_t* qx = co.data() + 0 * 65536 + offset;
        int32_t* qyv = co.data() + 1 * 65536 + offset;
        int32_t* qb = co.data() + 2 * 65536 + offset;
        const size_t lrows = size_t(std::min(cx, cy)), lcols = size_t(std::max(cx, cy));
        for (size_t k = 0; k < size; k++) {
          size_t row = k / cstride, col = k % cstride;
          if (row < lrows && col < lcols) continue;  // LLF corner comes from DC
          auto quant = [](float v) {
            float r = std::nearbyint(v);
            return std::fabs(v) < 0.58f ? 0 : int32_t(r);
          };
          // (a subsampled frame: a channel's block is coded with the frame's block that shares its top-left corner in block
          // units, and the chroma-from-luma term is whatever luma that block carries: dec_group.cc:432-452)
          int32_t iy = present[1] ? quant(coef[1][k] / (my[k] * mulc[1])) : 0;
          qyv[k] = iy;
          float ybias = iy == 0 ? 0.0f : (iy == 1 ? biases[1] : iy == -1 ? -biases[1] : float(iy) - biases[3] / float(iy));
          float ydeq = ybias * (my[k] * mulc[1]);
          if (present[0]) qx[k] = quant((coef[0][k] - x_cc * ydeq) / (mx[k] * mulc[0]));
          if (present[2]) qb[k] = quant((coef[2][k] - b_cc * ydeq) / (mb[k] * mulc[2]));
        }
        offset += size;
      }
  }
  if (model_only) {
    *model_only = std::move(f);
    return;
  }
  Assemble(f, p, out);
}

// ---------------------------------------------------------------- random mode
static void EncodeRandom(size_t img_xs, size_t img_ys, const Params& p, std::vector<uint8_t>* out) {
  FrameModel f;
  const size_t ups = (p.upsampling == 2 || p.upsampling == 4 || p.upsampling == 8) ? size_t(p.upsampling) : 1;
  const size_t xs = DivCeil(img_xs, ups), ys = DivCeil(img_ys, ups);  // the coded frame
  f.img_xs = img_xs;
  f.img_ys = img_ys;
  f.SetSubsampling(p.color_transform == 2 ? uint32_t(p.chroma_subsampling) : 0u);
  f.SetSize(xs, ys);
  Rng rng(p.seed);
  f.global_scale = 3000 + rng.Below(9000);
  f.quant_dc = 8 + rng.Below(16);
  f.gab = p.gab < 0 ? 1 : p.gab;
  f.epf_iters = p.epf_iters < 0 ? 1 : p.epf_iters;
  f.flags = (p.skip_dc_smoothing ? 128 : 0) | (p.noise > 0 ? 1 : 0) | (g_splines.empty() ? 0 : 16) | (g_patches.empty() ? 0 : 2);
  f.acs.assign(f.xb * f.yb, 0xFF);
  f.qf.assign(f.xb * f.yb, 0);
  f.sharp.assign(f.xb * f.yb, 0);
  f.ytox.assign(DivCeil(f.xb, 8) * DivCeil(f.yb, 8), 0);
  f.ytob.assign(f.ytox.size(), 0);
  for (size_t i = 0; i < f.ytox.size(); i++) {
    f.ytox[i] = int8_t(int(rng.Below(17)) - 8);
    f.ytob[i] = int8_t(int(rng.Below(17)) - 8);
  }
  for (auto& s : f.sharp) s = uint8_t(rng.Below(8));
  const bool basis_mode = p.strategy_mode == 3;  // known-answer streams: see below
  if (basis_mode) {
    std::fill(f.ytox.begin(), f.ytox.end(), 0);
    std::fill(f.ytob.begin(), f.ytob.end(), 0);
  }
  uint32_t mask = p.strategy_mask ? p.strategy_mask : 0x7FFFFFFu;
  if (f.Subsampled()) {  // only the transforms that cover one block (dec_modular.cc:534-538)
    mask &= 0x3F00Fu;
    if (!mask) mask = 1;
  }
  std::vector<int> allowed;
  for (int s = 0; s < 27; s++)
    if (mask & (1u << s)) allowed.push_back(s);
  for (size_t by = 0; by < f.yb; by++)
    for (size_t bx = 0; bx < f.xb; bx++) {
      if (f.acs[by * f.xb + bx] != 0xFF) continue;
      int st = -1;
      if (allowed.size() <= 8 && bx % 32 == 0 && by % 32 == 0) {
        // small masks are used to target specific strategies: guarantee the largest one that fits at group origins
        size_t best_area = 0;
        for (int cand : allowed) {
          size_t area = size_t(jxh::kCoveredX[cand]) * jxh::kCoveredY[cand];
          if (area > best_area && Fits(f, bx, by, cand)) {
            best_area = area;
            st = cand;
          }
        }
      }
      for (int tries = 0; tries < 6 && st < 0; tries++) {
        int cand = allowed[rng.Below(uint32_t(allowed.size()))];
        // big transforms are rarer so that small ones get space too
        size_t area = size_t(jxh::kCoveredX[cand]) * jxh::kCoveredY[cand];
        if (area >= 64 && allowed.size() > 8 && rng.Below(uint32_t(area / 16)) != 0) continue;
        if (Fits(f, bx, by, cand)) st = cand;
      }
      if (st < 0) st = (mask & 1) ? 0 : (Fits(f, bx, by, allowed[0]) ? allowed[0] : 0);
      if (!Fits(f, bx, by, st)) st = 0;
      Place(f, bx, by, st);
      f.qf[by * f.xb + bx] = 1 + int32_t(rng.Below(16));
    }
  const float inv_gs = 65536.0f / float(f.global_scale);
  const float inv_quant_dc = inv_gs / float(f.quant_dc);
  const float dc_step[3] = {inv_quant_dc / 4096.0f, inv_quant_dc / 512.0f, inv_quant_dc / 256.0f};
  // smooth-ish random DC in a plausible XYB range
  for (auto& d : f.dc) d.assign(f.xb * f.yb, 0);
  if (!basis_mode) {
    float vy = 0.4f, vx = 0.0f, vb = 0.0f;
    for (size_t by = 0; by < f.yb; by++)
      for (size_t bx = 0; bx < f.xb; bx++) {
        vy = std::max(0.05f, std::min(0.8f, vy + (rng.Uniform() - 0.5f) * 0.08f));
        vx = std::max(-0.02f, std::min(0.02f, vx + (rng.Uniform() - 0.5f) * 0.004f));
        vb = std::max(-0.2f, std::min(0.2f, vb + (rng.Uniform() - 0.5f) * 0.03f));
        size_t i = by * f.xb + bx;
        f.dc[1][i] = int32_t(std::lround(vy / dc_step[1]));
        f.dc[0][i] = int32_t(std::lround(vx / dc_step[0]));
        f.dc[2][i] = int32_t(std::lround(vb / dc_step[2]));
      }
  }
  const size_t xg = DivCeil(xs, 256), yg = DivCeil(ys, 256);
  f.coeffs.assign(xg * yg, {});
  size_t basis_count = 0;
  for (size_t g = 0; g < xg * yg; g++) {
    std::vector<int32_t>& co = f.coeffs[g];
    co.assign(3 * 65536, 0);
    const size_t bx0 = (g % xg) * 32, by0 = (g / xg) * 32;
    const size_t gw = std::min<size_t>(32, f.xb - bx0), gh = std::min<size_t>(32, f.yb - by0);
    size_t offset = 0;
    for (size_t by = 0; by < gh; by++)
      for (size_t bx = 0; bx < gw; bx++) {
        uint8_t a = f.acs[(by0 + by) * f.xb + bx0 + bx];
        if (!(a & 1)) continue;
        const int st = a >> 1;
        const size_t cx = jxh::kCoveredX[st], cy = jxh::kCoveredY[st];
        const size_t size = cx * cy * 64, cstride = std::max(cx, cy) * 8, lrows = std::min(cx, cy), lcols = std::max(cx, cy);
        const float density = p.zero_ac ? 0.0f : 0.02f + 0.25f * rng.Uniform();  // per-block sparsity
        if (basis_mode) {
          // Known-answer stream (strategy_mode 3): DC zero, no chroma from luma, and every varblock carries ONE non-zero
          // coefficient, in Y: the j-th varblock of the frame (raster order of groups, then decode order) gets natural
          // position number (j * stride) mod (positions outside the lowest-frequency corner), stride = `seed` | 1. The
          // decoded block is then a single basis function of the inverse transform, times the dequantisation weight.
          size_t n_pos = size - lrows * lcols, want = (basis_count++ * size_t(p.seed | 1)) % n_pos, seen = 0;
          for (size_t k = 0; k < size; k++) {
            const size_t row = k / cstride, col = k % cstride;
            if (row < lrows && col < lcols) continue;
            if (seen++ == want) co[65536 + offset + k] = 3;
          }
          offset += size;
          continue;
        }
        for (int c = 0; c < 3; c++) {
          int32_t* q = co.data() + size_t(c) * 65536 + offset;
          for (size_t k = 0; k < size; k++) {
            size_t row = k / cstride, col = k % cstride;
            if (row < lrows && col < lcols) continue;
            float freq = float(row * (lcols / lrows) + col) / float(cstride);  // 0..~2
            if (rng.Uniform() < density * std::exp(-2.5f * freq)) {
              int mag = 1 + int(rng.Below(3) == 0 ? rng.Below(6) : 0);
              if (rng.Below(64) == 0) mag += int(rng.Below(40));
              if (p.big_coeffs && rng.Below(256) == 0) mag = 40000 + int(rng.Below(200000));
              q[k] = rng.Below(2) ? mag : -mag;
            }
          }
        }
        offset += size;
      }
  }
  Assemble(f, p, out);
}


// ---------------------------------------------------------------- Modular (lossless) frames
// A small lossless encoder for tests and the lossless benchmark stream (BASELINE.json configs[3]): 8-bit grey / RGB (+
// alpha), non-XYB, groups of 256, one global MA tree. `flags`: bit 0 = prefix codes instead of rANS, bit 1 = LZ77,
// bit 2 = weighted-predictor leaves and a split on its error property, bit 3 = Squeeze (default steps), bit 4 = RCT
// (YCoCg) over the colour channels, bit 5 = every leaf uses a different predictor (all 14 occur), bit 6 = a split on a
// previous-channel property (16: |value| of the channel before). Lossless: whatever decodes it must return the input.
struct LosslessOptions {
  uint32_t flags;
  uint32_t seed;
};
struct LChannel {
  size_t w = 0, h = 0;
  int hshift = 0, vshift = 0;
  std::vector<int32_t> d;
  int32_t* Row(size_t y) { return d.data() + y * w; }
  const int32_t* Row(size_t y) const { return d.data() + y * w; }
};
static void FwdSqueezeLine(const int32_t* in, ptrdiff_t si, size_t n, int32_t* avg, ptrdiff_t sa, int32_t* res, ptrdiff_t sr) {
  // enc side of squeeze.cc: pair averages (rounded towards the first sample) and residuals minus the smooth tendency
  const size_t na = (n + 1) / 2, nr = n / 2;
  auto pair_avg = [&](size_t i) -> int64_t {
    if (2 * i + 1 < n) {
      const int64_t A = in[ptrdiff_t(2 * i) * si], B = in[ptrdiff_t(2 * i + 1) * si];
      return (A + B + (A > B)) >> 1;
    }
    return in[ptrdiff_t(2 * i) * si];
  };
  for (size_t i = 0; i < nr; i++) {
    const int64_t A = in[ptrdiff_t(2 * i) * si], B = in[ptrdiff_t(2 * i + 1) * si];
    const int64_t a = pair_avg(i), next = i + 1 < na ? pair_avg(i + 1) : a, before = i ? in[ptrdiff_t(2 * i - 1) * si] : a;
    avg[ptrdiff_t(i) * sa] = int32_t(a);
    res[ptrdiff_t(i) * sr] = int32_t((A - B) - jxh::SqueezeTendency(before, a, next));
  }
  if (na > nr) avg[ptrdiff_t(na - 1) * sa] = in[ptrdiff_t(n - 1) * si];
}

// `px`: interleaved samples as the integers the stream codes: `bits`-bit unsigned values, or (exp_bits != 0) the bit patterns
// of floats with `bits` bits of which `exp_bits` are the exponent (dec_modular.cc:128-185 reads them back the same way).
static void EncodeLossless(const int32_t* px, size_t xs, size_t ys, size_t nc, const LosslessOptions& o, std::vector<uint8_t>* out,
                           uint32_t bits = 8, uint32_t exp_bits = 0) {
  const bool gray = nc <= 2, alpha = nc == 2 || nc == 4;
  const uint32_t flags = o.flags;
  const size_t gdim = 256, xg = DivCeil(xs, gdim), yg = DivCeil(ys, gdim), num_groups = xg * yg;
  const size_t xdg = DivCeil(xs, gdim * 8), ydg = DivCeil(ys, gdim * 8), ndc = xdg * ydg;
  // ---- channels + forward transforms
  std::vector<LChannel> ch(nc);
  for (size_t c = 0; c < nc; c++) {
    ch[c].w = xs;
    ch[c].h = ys;
    ch[c].d.resize(xs * ys);
    for (size_t i = 0; i < xs * ys; i++) ch[c].d[i] = px[i * nc + c];
  }
  // flag 128: an XYB Modular frame ("lossy Modular": dec_modular.cc:583-631): the colour is converted to XYB and coded as
  // the integers Y, X, B - Y in units of the default DC quantisation steps (1/512, 1/4096, 1/256); the decoder's colour
  // stage brings it back. Colour images only; the alpha channel stays as it is.
  const bool xyb = (flags & 128) && !gray;
  if (xyb) {
    std::vector<float> planes[3];
    std::vector<uint8_t> rgb(xs * ys * 3);
    for (size_t i = 0; i < xs * ys; i++)
      for (int c = 0; c < 3; c++) rgb[i * 3 + c] = uint8_t(px[i * nc + c]);
    RgbToXyb(rgb.data(), xs, ys, xs, ys, planes);
    for (size_t i = 0; i < xs * ys; i++) {
      const int32_t Y = int32_t(std::lround(planes[1][i] * 512.0f));
      ch[0].d[i] = Y;
      ch[1].d[i] = int32_t(std::lround(planes[0][i] * 4096.0f));
      ch[2].d[i] = int32_t(std::lround(planes[2][i] * 256.0f)) - Y;
    }
  }
  const bool rct = (flags & 16) && !gray && !xyb;
  if (rct)  // forward YCoCg (rct.cc, type 6): the decoder computes tmp = Y - (Cg >> 1), G = Cg + tmp, B = tmp - (Co >> 1), R = B + Co
    for (size_t i = 0; i < xs * ys; i++) {
      const int32_t R = ch[0].d[i], G = ch[1].d[i], B = ch[2].d[i];
      const int32_t Co = R - B, tmp = B + (Co >> 1), Cg = G - tmp, Y = tmp + (Cg >> 1);
      ch[0].d[i] = Y;
      ch[1].d[i] = Co;
      ch[2].d[i] = Cg;
    }
  const bool squeeze = (flags & 8) != 0;
  std::vector<jxh::SqueezeStep> steps;
  if (squeeze) {
    std::vector<jxh::MChannel> shapes(nc);
    for (size_t c = 0; c < nc; c++) {
      shapes[c].w = xs;
      shapes[c].h = ys;
    }
    jxh::DefaultSqueeze(shapes, 0, &steps);
    for (const jxh::SqueezeStep& q : steps) {
      const size_t b = q.begin_c, e = b + q.num_c;
      size_t at = q.in_place ? e : ch.size();
      for (size_t c = b; c < e; c++, at++) {
        const LChannel src = ch[c];
        LChannel a, r;
        a.hshift = r.hshift = src.hshift + (q.horizontal ? 1 : 0);
        a.vshift = r.vshift = src.vshift + (q.horizontal ? 0 : 1);
        if (q.horizontal) {
          a.w = (src.w + 1) / 2;
          a.h = src.h;
          r.w = src.w - a.w;
          r.h = src.h;
        } else {
          a.w = src.w;
          a.h = (src.h + 1) / 2;
          r.w = src.w;
          r.h = src.h - a.h;
        }
        a.d.assign(a.w * a.h, 0);
        r.d.assign(r.w * r.h, 0);
        if (q.horizontal) {
          std::vector<int32_t> dummy(1);
          for (size_t y = 0; y < src.h; y++) FwdSqueezeLine(src.Row(y), 1, src.w, a.Row(y), 1, r.w ? r.Row(y) : dummy.data(), 1);
        } else {
          std::vector<int32_t> dummy(1);
          for (size_t x = 0; x < src.w; x++)
            FwdSqueezeLine(src.d.data() + x, ptrdiff_t(src.w), src.h, a.d.data() + x, ptrdiff_t(a.w), r.h ? r.d.data() + x : dummy.data(), ptrdiff_t(r.w));
        }
        ch[c] = a;
        ch.insert(ch.begin() + at, r);
      }
    }
  }
  // ---- the global tree: split on the channel index, optionally on the WP error / gradient property / previous channel
  struct Leaf {
    uint32_t predictor;
    int ctx;
  };
  struct Node {
    int prop;
    int32_t split;
    int l, r;
    Leaf leaf;
  };
  std::vector<Node> nodes;
  int num_leaves = 0;
  auto pick_predictor = [&](int k) -> uint32_t {
    if (flags & 32) return uint32_t((k + o.seed) % 14);
    return (flags & 4) ? 6u : 5u;
  };
  // BFS construction: chain over channel index (up to 6 buckets), each with optional sub-splits
  {
    struct Todo { int id; int lo, hi; int depth; };  // channel range [lo, hi]
    const int maxc = int(std::min<size_t>(ch.size(), 6)) - 1;
    nodes.push_back({-2, 0, 0, 0, {0, 0}});
    std::vector<Todo> queue{{0, 0, maxc, 0}};
    for (size_t qi = 0; qi < queue.size(); qi++) {
      const Todo t = queue[qi];
      if (t.lo < t.hi) {
        const int mid = (t.lo + t.hi) / 2;  // property 0 > mid ? left : right
        nodes[t.id].prop = 0;
        nodes[t.id].split = mid;
        nodes[t.id].l = int(nodes.size());
        nodes.push_back({-2, 0, 0, 0, {0, 0}});
        nodes[t.id].r = int(nodes.size());
        nodes.push_back({-2, 0, 0, 0, {0, 0}});
        queue.push_back({nodes[t.id].l, mid + 1, t.hi, 0});
        queue.push_back({nodes[t.id].r, t.lo, mid, 0});
      } else if (t.depth == 0 && (flags & 4)) {  // WP error property
        nodes[t.id].prop = 15;
        nodes[t.id].split = 0;
        nodes[t.id].l = int(nodes.size());
        nodes.push_back({-2, 0, 0, 0, {0, 0}});
        nodes[t.id].r = int(nodes.size());
        nodes.push_back({-2, 0, 0, 0, {0, 0}});
        queue.push_back({nodes[t.id].l, t.lo, t.hi, 1});
        queue.push_back({nodes[t.id].r, t.lo, t.hi, 1});
      } else if (t.depth <= 1 && (flags & 64) && t.lo > 0) {  // previous channel's |value|
        nodes[t.id].prop = 16;
        nodes[t.id].split = 20;
        nodes[t.id].l = int(nodes.size());
        nodes.push_back({-2, 0, 0, 0, {0, 0}});
        nodes[t.id].r = int(nodes.size());
        nodes.push_back({-2, 0, 0, 0, {0, 0}});
        queue.push_back({nodes[t.id].l, t.lo, t.hi, 2});
        queue.push_back({nodes[t.id].r, t.lo, t.hi, 2});
      } else if (t.depth <= 2) {  // gradient property 9 against the local magnitude
        nodes[t.id].prop = 5;  // |left|
        nodes[t.id].split = 40;
        nodes[t.id].l = int(nodes.size());
        nodes.push_back({-2, 0, 0, 0, {0, 0}});
        nodes[t.id].r = int(nodes.size());
        nodes.push_back({-2, 0, 0, 0, {0, 0}});
        queue.push_back({nodes[t.id].l, t.lo, t.hi, 3});
        queue.push_back({nodes[t.id].r, t.lo, t.hi, 3});
      } else {
        nodes[t.id].prop = -1;
        nodes[t.id].leaf = {pick_predictor(num_leaves), num_leaves};
        num_leaves++;
      }
    }
  }
  std::vector<Token> tree_tokens;
  for (const Node& n : nodes) {
    if (n.prop < 0) {
      tree_tokens.push_back({1, 0});
      tree_tokens.push_back({2, n.leaf.predictor});
      tree_tokens.push_back({3, 0});
      tree_tokens.push_back({4, 0});
      tree_tokens.push_back({5, 0});
    } else {
      tree_tokens.push_back({1, uint32_t(n.prop + 1)});
      tree_tokens.push_back({0, PackSigned(n.split)});
    }
  }
  const bool uses_wp = (flags & 4) || ((flags & 32) != 0);
  // ---- tokenise a stream: the same walk as encoding.cc:148-506 with the sample known
  jxh::WpHeader wph;
  auto tokenise = [&](const std::vector<const LChannel*>& part, const std::vector<std::pair<size_t, size_t>>& origin,
                      const std::vector<std::pair<size_t, size_t>>& size, int stream_id, int first_index, std::vector<Token>* toks) {
    for (size_t ci = 0; ci < part.size(); ci++) {
      const LChannel& full = *part[ci];
      const size_t w = size[ci].first, h = size[ci].second, ox = origin[ci].first, oy = origin[ci].second;
      if (!w || !h) continue;
      auto at = [&](const LChannel& c, size_t cox, size_t coy, ptrdiff_t x, ptrdiff_t y) -> int64_t { return c.Row(coy + size_t(y))[cox + size_t(x)]; };
      // previous channel of the same shape (for property 16..19)
      int ref = -1;
      for (int j = int(ci) - 1; j >= 0; j--)
        if (size[j] == size[ci] && part[j]->hshift == full.hshift && part[j]->vshift == full.vshift) {
          ref = j;
          break;
        }
      jxh::WpState wp(wph, w);
      int32_t prev_p9 = 0;
      for (size_t y = 0; y < h; y++) {
        prev_p9 = 0;
        for (size_t x = 0; x < w; x++) {
          const int64_t left = x ? at(full, ox, oy, x - 1, y) : (y ? at(full, ox, oy, x, y - 1) : 0);
          const int64_t top = y ? at(full, ox, oy, x, y - 1) : left;
          const int64_t topleft = (x && y) ? at(full, ox, oy, x - 1, y - 1) : left;
          const int64_t topright = (x + 1 < w && y) ? at(full, ox, oy, x + 1, y - 1) : top;
          const int64_t leftleft = x > 1 ? at(full, ox, oy, x - 2, y) : left;
          const int64_t toptop = y > 1 ? at(full, ox, oy, x, y - 2) : top;
          const int64_t toprightright = (x + 2 < w && y) ? at(full, ox, oy, x + 2, y - 1) : topright;
          int32_t props[20] = {0};
          props[0] = int32_t(first_index + int(ci));
          props[1] = stream_id;
          props[2] = int32_t(y);
          props[3] = int32_t(x);
          props[4] = int32_t(top > 0 ? top : -top);
          props[5] = int32_t(left > 0 ? left : -left);
          props[6] = int32_t(top);
          props[7] = int32_t(left);
          props[8] = int32_t(left - prev_p9);
          props[9] = int32_t(left + top - topleft);
          prev_p9 = props[9];
          int64_t wp_pred = 0;
          if (uses_wp) wp_pred = wp.Predict(x, y, w, top, left, topright, topleft, toptop, &props[15]);
          if (ref >= 0) {
            const LChannel& rc = *part[ref];
            const size_t rx = origin[ref].first, ry = origin[ref].second;
            const int64_t v = at(rc, rx, ry, x, y), vl = x ? at(rc, rx, ry, x - 1, y) : 0;
            const int64_t vt = y ? at(rc, rx, ry, x, y - 1) : vl, vtl = (x && y) ? at(rc, rx, ry, x - 1, y - 1) : vl;
            const int64_t vp = jxh::ClampedGradient(int32_t(vl), int32_t(vt), int32_t(vtl));
            props[16] = int32_t(v < 0 ? -v : v);
            props[17] = int32_t(v);
            props[18] = int32_t(v - vp < 0 ? vp - v : v - vp);
            props[19] = int32_t(v - vp);
          }
          int pos = 0;
          while (nodes[pos].prop >= 0) pos = props[nodes[pos].prop] > nodes[pos].split ? nodes[pos].l : nodes[pos].r;
          const Leaf& lf = nodes[pos].leaf;
          const int64_t guess = jxh::PredictOne(lf.predictor, left, top, toptop, topleft, topright, leftleft, toprightright, wp_pred);
          const int64_t cur = at(full, ox, oy, x, y);
          toks->push_back({uint32_t(lf.ctx), PackSigned(int32_t(cur - guess))});
          if (uses_wp) wp.Update(cur, x, y, w);
        }
      }
    }
  };
  // stream 0: channels no larger than a group; groups: rectangles by shift bracket (dec_modular.cc:320-425)
  size_t first_big = 0;
  while (first_big < ch.size() && ch[first_big].w <= gdim && ch[first_big].h <= gdim) first_big++;
  std::vector<Token> global_tokens;
  {
    std::vector<const LChannel*> part;
    std::vector<std::pair<size_t, size_t>> org, sz;
    for (size_t c = 0; c < first_big; c++) {
      part.push_back(&ch[c]);
      org.push_back({0, 0});
      sz.push_back({ch[c].w, ch[c].h});
    }
    tokenise(part, org, sz, 0, 0, &global_tokens);
  }
  auto group_tokens = [&](size_t x0, size_t y0, size_t span, int min_shift, int max_shift, int stream_id, std::vector<Token>* toks) {
    std::vector<const LChannel*> part;
    std::vector<std::pair<size_t, size_t>> org, sz;
    for (size_t c = first_big; c < ch.size(); c++) {
      const LChannel& fc = ch[c];
      const int shift = std::min(fc.hshift, fc.vshift);
      if (shift < min_shift || shift > max_shift) continue;
      const size_t rx = x0 >> fc.hshift, ry = y0 >> fc.vshift;
      if (rx >= fc.w || ry >= fc.h) continue;
      const size_t rw = std::min(span >> fc.hshift, fc.w - rx), rh = std::min(span >> fc.vshift, fc.h - ry);
      if (!rw || !rh) continue;
      part.push_back(&fc);
      org.push_back({rx, ry});
      sz.push_back({rw, rh});
    }
    if (part.empty()) return false;
    tokenise(part, org, sz, stream_id, 0, toks);
    return true;
  };
  const bool multi = num_groups > 1;
  std::vector<std::vector<Token>> dc_tokens(multi ? ndc : 0), ac_tokens(multi ? num_groups : 0);
  std::vector<uint8_t> dc_present(dc_tokens.size(), 0), ac_present(ac_tokens.size(), 0);
  if (multi) {
#pragma omp parallel for schedule(dynamic)
    for (size_t g = 0; g < ndc; g++)
      dc_present[g] = group_tokens((g % xdg) * gdim * 8, (g / xdg) * gdim * 8, gdim * 8, 3, 1000, int(1 + ndc + g), &dc_tokens[g]);
#pragma omp parallel for schedule(dynamic)
    for (size_t g = 0; g < num_groups; g++)
      ac_present[g] = group_tokens((g % xg) * gdim, (g / xg) * gdim, gdim, 0, 2, int(1 + 3 * ndc + 17 + g), &ac_tokens[g]);
  }
  // ---- codes
  jxh::HybridCfg cfg420;
  cfg420.split_exp = 4;
  cfg420.split_token = 16;
  cfg420.msb = 2;
  cfg420.lsb = 0;
  EncCode tree_code, code;
  BuildCode({&tree_tokens}, 6, 6, cfg420, &tree_code);
  const int mode = int(flags & 3);
  std::vector<std::vector<Token>*> streams{&global_tokens};
  for (auto& t : dc_tokens) streams.push_back(&t);
  for (auto& t : ac_tokens) streams.push_back(&t);
  if (mode & 2)
    for (auto* t : streams) Lz77Pass(t, uint32_t(num_leaves), 120);
  std::vector<const std::vector<Token>*> all(streams.begin(), streams.end());
  BuildCode(all, size_t(num_leaves) + ((mode & 2) ? 1 : 0), 32, cfg420, &code, mode);
  // ---- sections
  auto write_stream_header = [&](BitWriter& bw, bool global) {
    bw.Write(1, 1);  // use the global tree
    bw.Write(1, 1);  // default weighted-predictor header
    if (!global) {
      bw.Write(2, 0);  // no transforms
      return;
    }
    const uint32_t nt = (rct ? 1 : 0) + (squeeze ? 1 : 0);
    bw.Write(2, nt);  // U32(Val(0), Val(1), BitsOffset(4, 2), ...): 0, 1 or selector 2 + 4 bits
    if (nt == 2) bw.Write(4, 0);
    if (rct) {
      bw.Write(2, 0);  // RCT
      bw.Write(2, 0);  // begin_c = 0 (3 bits follow)
      bw.Write(3, 0);
      bw.Write(2, 0);  // type 6 (YCoCg)
    }
    if (squeeze) {
      bw.Write(2, 2);  // Squeeze
      bw.Write(2, 0);  // default parameters
    }
  };
  std::vector<std::vector<uint8_t>> sections;
  auto dc_global = [&](BitWriter& bw) {
    if (!g_patches.empty()) WritePatches(bw, alpha ? 1 : 0);
    if (!g_splines.empty()) WriteSplines(bw);
    bw.Write(1, 1);  // default DC dequantisation
    bw.Write(1, 1);  // global tree present
    WriteCodeHeader(bw, tree_code);
    WriteTokens(bw, tree_tokens.data(), tree_tokens.size(), tree_code);
    WriteCodeHeader(bw, code);
    write_stream_header(bw, true);
    if (!global_tokens.empty()) WriteTokens(bw, global_tokens.data(), global_tokens.size(), code);
  };
  if (!multi) {
    BitWriter bw;
    dc_global(bw);
    bw.ZeroPad();
    sections.push_back(bw.bytes());
  } else {
    sections.resize(2 + ndc + num_groups);
    {
      BitWriter bw;
      dc_global(bw);
      bw.ZeroPad();
      sections[0] = bw.bytes();
    }
    for (size_t g = 0; g < ndc; g++) {
      BitWriter bw;
      if (dc_present[g]) {
        write_stream_header(bw, false);
        WriteTokens(bw, dc_tokens[g].data(), dc_tokens[g].size(), code);
      }
      bw.ZeroPad();
      sections[1 + g] = bw.bytes();
    }
#pragma omp parallel for schedule(dynamic)
    for (size_t g = 0; g < num_groups; g++) {
      BitWriter bw;
      if (ac_present[g]) {
        write_stream_header(bw, false);
        WriteTokens(bw, ac_tokens[g].data(), ac_tokens[g].size(), code);
      }
      bw.ZeroPad();
      sections[2 + ndc + g] = bw.bytes();
    }
  }
  // ---- headers
  BitWriter bw;
  bw.Write(16, 0x0AFF);
  bw.Write(1, 0);  // not "small"
  WriteSizeDim(bw, g_image_h ? g_image_h : uint32_t(ys));
  bw.Write(3, 0);
  WriteSizeDim(bw, g_image_w ? g_image_w : uint32_t(xs));
  bw.Write(1, 0);  // ImageMetadata not all_default
  WriteExtraFields(bw);
  if (exp_bits == 0) {  // image_metadata.cc BitDepth: U32(Val(8), Val(10), Val(12), BitsOffset(6, 1))
    bw.Write(1, 0);
    if (bits == 8 || bits == 10 || bits == 12) {
      bw.Write(2, (bits - 8) / 2);
    } else {
      bw.Write(2, 3);
      bw.Write(6, bits - 1);
    }
  } else {  // U32(Val(32), Val(16), Val(24), BitsOffset(6, 1)), then Bits(4) + 1 exponent bits
    bw.Write(1, 1);
    if (bits == 32 || bits == 16 || bits == 24) {
      bw.Write(2, bits == 32 ? 0 : (bits == 16 ? 1 : 2));
    } else {
      bw.Write(2, 3);
      bw.Write(6, bits - 1);
    }
    bw.Write(4, exp_bits - 1);
  }
  bw.Write(1, exp_bits == 0 && bits <= 12 ? 1 : 0);  // modular_16_bit_buffer_sufficient
  bw.Write(2, alpha ? 1 : 0);  // extra channels
  if (alpha) WriteAlphaChannelInfo(bw);
  const bool xyb_frame = (o.flags & 128) && !gray;
  bw.Write(1, xyb_frame ? 1 : 0);  // xyb_encoded
  const bool with_icc = !g_embedded_icc.empty() && !gray;
  if (with_icc) {
    bw.Write(1, 0);  // not all_default
    bw.Write(1, 1);  //   want_icc
    bw.Write(2, 0);  //   colour space RGB
  } else if (g_color.enabled) {
    WriteColorEncodingFields(bw, gray);
  } else if (!gray) {
    bw.Write(1, 1);  // ColorEncoding all_default (sRGB)
  } else {
    bw.Write(1, 0);  // not all_default
    bw.Write(1, 0);  // no ICC
    bw.Write(2, 1);  // colour space: grey
    bw.Write(2, 1);  // white point D65
    bw.Write(1, 0);  // no gamma
    bw.Write(2, 2);  // transfer function: enum selector 2 + 4 bits, value 13 (sRGB)
    bw.Write(4, 13 - 2);
    bw.Write(2, 1);  // rendering intent: relative
  }
  WriteToneMapping(bw);
  bw.Write(2, 0);  // no extensions
  bw.Write(1, 1);  // CustomTransformData all_default
  if (with_icc) AppendEmbeddedIcc(bw);
  bw.ZeroPad();
  g_last_header_bytes = bw.bytes().size();
  // FrameHeader
  bw.Write(1, 0);  // not all_default
  bw.Write(2, g_dc_frame_level > 0 ? 1 : (g_reference_slot >= 0 ? 2 : 0));  // regular frame, kDCFrame or kReferenceOnly
  bw.Write(1, 1);  // Modular
  {
    const uint32_t mflags = (g_splines.empty() ? 0 : 16) | (g_patches.empty() ? 0 : 2);
    if (mflags == 0) {
      bw.Write(2, 0);
    } else {  // U64 selector 1: 1 + 4 bits (<= 16), selector 2: 17 + 8 bits
      bw.Write(2, mflags <= 16 ? 1 : 2);
      if (mflags <= 16) bw.Write(4, mflags - 1);
      else bw.Write(8, mflags - 17);
    }
  }
  if (!xyb_frame) bw.Write(1, 0);  // (not XYB:) no YCbCr
  bw.Write(2, 0);  // upsampling 1
  if (alpha) bw.Write(2, 0);
  bw.Write(2, 1);  // group_size_shift 1 (256)
  if (g_reference_slot < 0 || g_dc_frame_level > 0) bw.Write(2, 0);  // one pass (a kReferenceOnly frame has no Passes bundle: frame_header.cc:303)
  if (g_dc_frame_level > 0) bw.Write(2, uint32_t(g_dc_frame_level - 1));  // dc_level
  const bool whole = WriteCropAndBlending(bw, uint32_t(xs), uint32_t(ys), alpha);
  WriteFrameTiming(bw, whole);
  WriteFrameName(bw);
  bw.Write(1, 0);  // loop filter not all_default
  bw.Write(1, 0);  // no gaborish
  bw.Write(2, 0);  // no EPF
  bw.Write(2, 0);  // no loop-filter extensions
  bw.Write(2, 0);  // no frame-header extensions
  bw.Write(1, 0);  // TOC not permuted
  bw.ZeroPad();
  for (const auto& sct : sections) {
    static const uint32_t bits[4] = {10, 14, 22, 30}, offs[4] = {0, 1024, 17408, 4211712};
    WriteU32Sel(bw, uint32_t(sct.size()), bits, offs);
  }
  bw.ZeroPad();
  *out = bw.bytes();
  for (const auto& sct : sections) out->insert(out->end(), sct.begin(), sct.end());
}

}  // namespace jxe

extern "C" {

struct JxlEncParams {
  float distance;
  int32_t epf_iters, gab, strategy_mode;
  uint32_t strategy_mask, seed;
  int32_t max_clusters, skip_dc_smoothing, random_cmap, zero_ac;
  int32_t num_histograms;  // AC histogram sets (group g uses set g % num_histograms); 0 or 1 = one
  int32_t big_coeffs;      // random mode: sprinkle magnitudes beyond 16 bits (forces int32 coefficient storage in decoders)
  int32_t num_passes;      // 1..3; progressive: pass p of n carries the coefficient's bits above shift n - 1 - p that the passes
                           // before it have not (shifts n - 1 .. 0); three passes also name a downsampling level (4x after pass 0)
  int32_t upsampling;      // 0/1, 2, 4 or 8: the frame is coded at ceil(size / upsampling) and flagged for upsampling
  int32_t custom_orders;   // 1 = code a (seeded random) custom coefficient order for every used order bucket, channel and pass
  int32_t custom_bctx;     // 1 = code a block context map with quant-field and DC thresholds (entropy_coder.cc:25-60)
  int32_t custom_cmap;     // 1 = non-default colour-correlation header (factor 100, bases 0.25 / 0.75, DC factors 3 / -5) and
                           //     x/b quant-matrix scales 2 / 4: valid streams, but image mode does not compensate for them
  int32_t custom_lf;       // 1 = non-default loop filter header: Gaborish weights, EPF sharpness LUT, channel scales and
                           //     sigma parameters, and non-default DC dequantisation steps (valid streams, not tuned ones)
  int32_t ac_code_mode;    // AC coefficient streams: bit 0 = prefix codes instead of ANS, bit 1 = LZ77
  int32_t noise;           // > 0: noise synthesis, see jxe::Params
  int32_t cfl_fit;         // 1 = per-tile chroma-from-luma fit (the reference's fast FindBestMultiplier), see jxe::Params
  int32_t color_transform; // 0 = XYB, 1 = none, 2 = YCbCr, see jxe::Params
  int32_t raw_quant;       // 1 = RAW dequantisation table for the 8x8 DCT, see jxe::Params
  int32_t chroma_subsampling;  // YCbCr frames: channel modes of Cb, Y, Cr (4 = 4:2:0, 8 = 4:2:2, 12 = 4:4:0), see jxe::Params
  int32_t ec_upsampling;   // jxlenc_encode_rgba8: upsampling factor of the alpha channel, see jxe::Params
};

// The next VarDCT streams code their own upsampling weights (mask bit k: the 2^(k+1)-fold matrix; 0: default weights again).
// Test aid, not thread-safe.
void jxlenc_set_custom_upsampling(uint32_t mask, uint32_t seed) {
  jxe::g_custom_ups_mask = mask & 7;
  jxe::g_custom_ups_seed = seed;
}

// The next encoded streams embed this coded ICC profile of exactly `bits` bits (n = 0: none again). Test aid, not thread-safe.
void jxlenc_set_embedded_icc(const uint8_t* coded, size_t n, size_t bits) {
  jxe::g_embedded_icc.assign(coded, coded + n);
  jxe::g_embedded_icc_bits = bits <= n * 8 ? bits : n * 8;
}

// The next encoded streams declare this orientation (1..8, codestream_header.h:45-54; 1 = identity). Test aid, not thread-safe.
// The next streams' image header announces a preview frame of xs x ys (enabled = 0: none again). Test aid, not thread-safe.
void jxlenc_set_preview(int enabled, uint32_t xs, uint32_t ys) {
  jxe::g_preview.enabled = enabled != 0;
  jxe::g_preview.xs = xs;
  jxe::g_preview.ys = ys;
}
void jxlenc_set_orientation(uint32_t orientation) { jxe::g_orientation = orientation >= 1 && orientation <= 8 ? orientation : 1; }

// Spline dictionary of the next streams (n = 0: none again); layout at jxe::g_splines. Test aid, not thread-safe.
void jxlenc_set_splines(const int32_t* data, size_t n) { jxe::g_splines.assign(data, data + n); }

// Animation mode for the next streams (enabled = 0: stills again). Test aid, not thread-safe.
void jxlenc_set_animation(int enabled, uint32_t tps_numerator, uint32_t tps_denominator, uint32_t num_loops, uint32_t duration, int is_last) {
  jxe::g_anim.enabled = enabled != 0;
  jxe::g_anim.tps_num = tps_numerator;
  jxe::g_anim.tps_den = tps_denominator;
  jxe::g_anim.loops = num_loops;
  jxe::g_anim.duration = duration;
  jxe::g_anim.is_last = is_last != 0;
}
// How the next frame sits on the canvas (enabled = 0: a plain full frame again): crop origin, canvas size, BlendMode of
// colour and of alpha (0 replace, 1 add, 2 blend, 3 alpha-weighted add, 4 multiply), their reference slots, the clamp
// flag, the slot the blended canvas is saved in. `timed` = 0 turns the multi-frame mode of jxlenc_set_animation into a
// layered still (no animation header, no durations). Test aid, not thread-safe.
void jxlenc_set_layer(int enabled, int32_t x0, int32_t y0, uint32_t canvas_w, uint32_t canvas_h, uint32_t mode, uint32_t alpha_mode,
                      uint32_t source, uint32_t alpha_source, uint32_t clamp, uint32_t save_as, int timed) {
  jxe::g_layer.enabled = enabled != 0;
  jxe::g_layer.x0 = x0;
  jxe::g_layer.y0 = y0;
  jxe::g_layer.canvas_w = canvas_w;
  jxe::g_layer.canvas_h = canvas_h;
  jxe::g_layer.mode = mode;
  jxe::g_layer.alpha_mode = alpha_mode;
  jxe::g_layer.source = source;
  jxe::g_layer.alpha_source = alpha_source;
  jxe::g_layer.clamp = clamp;
  jxe::g_layer.save_as = save_as;
  jxe::g_anim.timed = timed != 0;
}
void jxlenc_set_color_encoding(int enabled, uint32_t white_point, uint32_t primaries, uint32_t have_gamma, uint32_t gamma,
                               uint32_t transfer_function, uint32_t intent, const int32_t* xy8) {
  jxe::g_color.enabled = enabled != 0;
  jxe::g_color.white_point = white_point;
  jxe::g_color.primaries = primaries;
  jxe::g_color.have_gamma = have_gamma;
  jxe::g_color.gamma = gamma;
  jxe::g_color.transfer_function = transfer_function;
  jxe::g_color.intent = intent;
  for (int i = 0; i < 8; i++) jxe::g_color.xy[i] = xy8 ? xy8[i] : 0;
}
void jxlenc_set_reference_frame(int slot) { jxe::g_reference_slot = slot; }
// The next frame is a kDCFrame of `level` (1..4; 0 = a regular frame again) / the next VarDCT frame takes its DC image from
// the DC frame before it (kUseDcFrame). Test aids, not thread-safe.
void jxlenc_set_dc_frame(int level) { jxe::g_dc_frame_level = level >= 1 && level <= 4 ? level : 0; }
void jxlenc_set_use_dc_frame(int on) { jxe::g_use_dc_frame = on != 0; }
void jxlenc_set_image_size(uint32_t w, uint32_t h) {
  jxe::g_image_w = w;
  jxe::g_image_h = h;
}
void jxlenc_set_patches(const int32_t* data, size_t n) { jxe::g_patches.assign(data, data + n); }
void jxlenc_set_frame_name(const char* name) { jxe::g_frame_name = name ? name : ""; }
void jxlenc_set_alpha_premultiplied(int premultiplied) { jxe::g_alpha_premultiplied = premultiplied != 0; }
// Byte offset of the frame header in the stream written last (signature + image header come before it).
size_t jxlenc_last_header_bytes(void) { return jxe::g_last_header_bytes; }

static int Finish(std::vector<uint8_t>& v, uint8_t** out, size_t* n) {
  *out = static_cast<uint8_t*>(malloc(v.size()));
  if (!*out) return -1;
  memcpy(*out, v.data(), v.size());
  *n = v.size();
  return 0;
}

int jxlenc_encode_rgb8(const uint8_t* rgb, uint32_t xs, uint32_t ys, const JxlEncParams* p, uint8_t** out, size_t* n) {
  jxe::UseThreads();
  if (!rgb || !xs || !ys || !p || p->distance <= 0) return -1;
  jxe::Params q;
  static_assert(sizeof(jxe::Params) == sizeof(JxlEncParams), "param layout");
  memcpy(&q, p, sizeof(q));
  std::vector<uint8_t> v;
  try {
    if (q.upsampling == 2 || q.upsampling == 4 || q.upsampling == 8) {
      // code a box-downsampled frame; the decoder upsamples it back to xs x ys
      const uint32_t N = uint32_t(q.upsampling), sx = (xs + N - 1) / N, sy = (ys + N - 1) / N;
      std::vector<uint8_t> small(size_t(sx) * sy * 3);
      for (uint32_t y = 0; y < sy; y++)
        for (uint32_t x = 0; x < sx; x++)
          for (int c = 0; c < 3; c++) {
            uint32_t sum = 0, cnt = 0;
            for (uint32_t dy = 0; dy < N && y * N + dy < ys; dy++)
              for (uint32_t dx = 0; dx < N && x * N + dx < xs; dx++) {
                sum += rgb[(size_t(y * N + dy) * xs + x * N + dx) * 3 + c];
                cnt++;
              }
            small[(size_t(y) * sx + x) * 3 + c] = uint8_t((sum + cnt / 2) / cnt);
          }
      jxe::EncodeImage(small.data(), sx, sy, q, &v, xs, ys);
    } else
    jxe::EncodeImage(rgb, xs, ys, q, &v);
  } catch (...) {
    return -2;
  }
  return Finish(v, out, n);
}

// jxlenc_encode_rgb8 with the pixel-domain half (colour, sharpening, transform selection, forward DCT, quantisation) done
// by `forward` (jxlhip_enc_forward and its context); entropy coding and headers here. seconds (may be NULL): forward
// call, assembly.
int jxlenc_encode_rgb8_forward(const uint8_t* rgb, uint32_t xs, uint32_t ys, const JxlEncParams* p, jxe::ForwardFn forward, void* ctx,
                               uint8_t** out, size_t* n, double* seconds) {
  jxe::UseThreads();
  if (!rgb || !xs || !ys || !p || p->distance <= 0 || !forward) return -1;
  jxe::Params q;
  memcpy(&q, p, sizeof(q));
  if (q.upsampling > 1) return -1;
  std::vector<uint8_t> v;
  jxe::ForwardHook hook = {forward, ctx, {0, 0}};
  try {
    jxe::EncodeImage(rgb, xs, ys, q, &v, 0, 0, nullptr, &hook);
  } catch (...) {
    return -2;
  }
  if (seconds) {
    seconds[0] = hook.seconds[0];
    seconds[1] = hook.seconds[1];
  }
  return Finish(v, out, n);
}

// The same with the coefficients tokenised on the device too (jxlhip_enc_token_counts / jxlhip_enc_tokens on the same context):
// the host starts from the tokens. Falls back to host tokenisation for progressive passes, coded orders or a coded block
// context map; seconds[2] receives the number of tokens the device produced (0 after that fallback).
int jxlenc_encode_rgb8_forward_tokens(const uint8_t* rgb, uint32_t xs, uint32_t ys, const JxlEncParams* p, jxe::ForwardFn forward,
                                      jxe::TokenCountsFn tok_counts, jxe::TokensFn tok_emit, void* ctx, uint8_t** out, size_t* n, double* seconds) {
  jxe::UseThreads();
  if (!rgb || !xs || !ys || !p || p->distance <= 0 || !forward || !tok_counts || !tok_emit) return -1;
  jxe::Params q;
  memcpy(&q, p, sizeof(q));
  if (q.upsampling > 1) return -1;
  std::vector<uint8_t> v;
  jxe::ForwardHook hook = {forward, ctx, {0, 0}};
  hook.tok_counts = tok_counts;
  hook.tok_emit = tok_emit;
  try {
    jxe::EncodeImage(rgb, xs, ys, q, &v, 0, 0, nullptr, &hook);
  } catch (...) {
    return -2;
  }
  if (seconds) {
    seconds[0] = hook.seconds[0];
    seconds[1] = hook.seconds[1];
    seconds[2] = double(hook.device_tokens);
  }
  return Finish(v, out, n);
}

// The CPU form of the forward path, with jxlhip_enc_forward's signature (ctx unused): what the GPU tests compare the
// device path with, array by array, and what lets the hook plumbing be tested without a GPU.
int jxlenc_forward_cpu(void*, const uint8_t* rgb, size_t stride, const JxlHipEncDesc* d, uint8_t* acs, int32_t* qf, int32_t* dc,
                       int32_t* coeffs) {
  jxe::UseThreads();
  if (!rgb || !d || !acs || !qf || !dc || !coeffs || stride < size_t(d->xsize) * 3) return -1;
  jxe::Params q;
  memset(&q, 0, sizeof(q));
  q.distance = d->distance;
  q.epf_iters = -1;
  q.gab = int32_t(d->gaborish);
  q.strategy_mode = int32_t(d->strategy_mode);
  q.cfl_fit = int32_t(d->cfl_fit);
  q.seed = 1;
  std::vector<uint8_t> tight;
  if (stride != size_t(d->xsize) * 3) {
    tight.resize(size_t(d->xsize) * d->ysize * 3);
    for (uint32_t y = 0; y < d->ysize; y++) memcpy(tight.data() + size_t(y) * d->xsize * 3, rgb + y * stride, size_t(d->xsize) * 3);
    rgb = tight.data();
  }
  jxe::FrameModel f;
  try {
    jxe::EncodeImage(rgb, d->xsize, d->ysize, q, nullptr, 0, 0, nullptr, nullptr, &f);
  } catch (...) {
    return -2;
  }
  if (f.global_scale != d->global_scale || f.quant_dc != d->quant_dc) return -3;  // the descriptor is not this distance's
  const size_t nb = f.xb * f.yb;
  memcpy(acs, f.acs.data(), nb);
  if (d->ytox) memcpy(d->ytox, f.ytox.data(), f.ytox.size());
  if (d->ytob) memcpy(d->ytob, f.ytob.data(), f.ytob.size());
  memcpy(qf, f.qf.data(), nb * 4);
  for (int c = 0; c < 3; c++) memcpy(dc + c * nb, f.dc[c].data(), nb * 4);
  for (size_t g = 0; g < f.coeffs.size(); g++) memcpy(coeffs + g * 3 * 65536, f.coeffs[g].data(), size_t(3) * 65536 * 4);
  return 0;
}

// Test access: the raw outputs of one forward call made with the descriptor jxlenc_encode_rgb8_forward builds
// (acs / qf: yb * xb, dc: 3 * yb * xb, coeffs: groups * 3 * 65536).
int jxlenc_forward_model(const uint8_t* rgb, uint32_t xs, uint32_t ys, const JxlEncParams* p, jxe::ForwardFn forward, void* ctx,
                         uint8_t* acs, int32_t* qf, int32_t* dc, int32_t* coeffs) {
  jxe::UseThreads();
  if (!rgb || !xs || !ys || !p || p->distance <= 0 || !forward || !acs || !qf || !dc || !coeffs) return -1;
  jxe::Params q;
  memcpy(&q, p, sizeof(q));
  jxe::ForwardHook hook = {forward, ctx, {0, 0}};
  hook.cap_acs = acs;
  hook.cap_qf = qf;
  hook.cap_dc = dc;
  hook.cap_coeffs = coeffs;
  std::vector<uint8_t> v;
  try {
    jxe::EncodeImage(rgb, xs, ys, q, &v, 0, 0, nullptr, &hook);
  } catch (...) {
    return -2;
  }
  return 0;
}

// RGBA8 image: the colour as jxlenc_encode_rgb8, alpha (channel 3) losslessly as a Modular-coded extra channel.
int jxlenc_encode_rgba8(const uint8_t* rgba, uint32_t xs, uint32_t ys, const JxlEncParams* p, uint8_t** out, size_t* n) {
  jxe::UseThreads();
  if (!rgba || !xs || !ys || !p || p->distance <= 0) return -1;
  jxe::Params q;
  memcpy(&q, p, sizeof(q));
  const uint32_t ups = (q.upsampling == 2 || q.upsampling == 4 || q.upsampling == 8) ? uint32_t(q.upsampling) : 1;
  uint32_t ecu = (q.ec_upsampling == 2 || q.ec_upsampling == 4 || q.ec_upsampling == 8) ? uint32_t(q.ec_upsampling) : 1;
  if (ecu < ups) ecu = ups;
  if (ecu > 4 * ups) return -1;
  // colour and alpha box-downsampled by their factors; the decoder upsamples each back to xs x ys
  auto shrink = [&](uint32_t N, int first, int count, std::vector<uint8_t>* dst, uint32_t* w, uint32_t* h) {
    *w = (xs + N - 1) / N;
    *h = (ys + N - 1) / N;
    dst->resize(size_t(*w) * *h * count);
    for (uint32_t y = 0; y < *h; y++)
      for (uint32_t x = 0; x < *w; x++)
        for (int c = 0; c < count; c++) {
          uint32_t sum = 0, cnt = 0;
          for (uint32_t dy = 0; dy < N && y * N + dy < ys; dy++)
            for (uint32_t dx = 0; dx < N && x * N + dx < xs; dx++) {
              sum += rgba[(size_t(y * N + dy) * xs + x * N + dx) * 4 + first + c];
              cnt++;
            }
          (*dst)[(size_t(y) * *w + x) * count + c] = uint8_t((sum + cnt / 2) / cnt);
        }
  };
  std::vector<uint8_t> rgb, alpha;
  uint32_t sx, sy, aw, ah;
  shrink(ups, 0, 3, &rgb, &sx, &sy);
  shrink(ecu, 3, 1, &alpha, &aw, &ah);
  uint32_t shift = 0;
  while ((ups << shift) < ecu) shift++;
  std::vector<uint8_t> v;
  try {
    jxe::g_alpha_dims[0] = ecu == 1 ? 0 : aw;
    jxe::g_alpha_dims[1] = ah;
    jxe::g_alpha_dims[2] = ecu;
    jxe::g_alpha_dims[3] = shift;
    jxe::EncodeImage(rgb.data(), sx, sy, q, &v, ups == 1 ? 0 : xs, ups == 1 ? 0 : ys, &alpha);
    jxe::g_alpha_dims[0] = 0;
  } catch (...) {
    jxe::g_alpha_dims[0] = 0;
    return -2;
  }
  return Finish(v, out, n);
}

// Lossless (Modular frame) stream of an interleaved 8-bit image with 1-4 channels; see jxe::EncodeLossless for `flags`.
int jxlenc_encode_lossless(const uint8_t* px, uint32_t xs, uint32_t ys, uint32_t channels, uint32_t flags, uint32_t seed, uint8_t** out,
                           size_t* n) {
  jxe::UseThreads();
  if (!px || !xs || !ys || channels < 1 || channels > 4) return -1;
  std::vector<uint8_t> v;
  try {
    jxe::LosslessOptions o{flags, seed};
    std::vector<int32_t> wide(size_t(xs) * ys * channels);
    for (size_t i = 0; i < wide.size(); i++) wide[i] = px[i];
    jxe::EncodeLossless(wide.data(), xs, ys, channels, o, &v);
  } catch (...) {
    return -2;
  }
  return Finish(v, out, n);
}

// The same for samples of any depth the format has: `px` holds the colour samples as the integers the stream codes --
// `bits`-bit unsigned values (exp_bits 0, bits 1..31) or the bit patterns of floats of `bits` bits with `exp_bits` exponent
// bits (bits 32 / exp_bits 8 = binary32 as it is, 16 / 5 = binary16); an alpha channel stays 8-bit. No XYB flag here.
int jxlenc_encode_lossless_samples(const int32_t* px, uint32_t xs, uint32_t ys, uint32_t channels, uint32_t flags, uint32_t seed,
                                   uint32_t bits, uint32_t exp_bits, uint8_t** out, size_t* n) {
  jxe::UseThreads();
  if (!px || !xs || !ys || channels < 1 || channels > 4 || (flags & 128)) return -1;
  if (exp_bits == 0 ? (bits < 1 || bits > 31) : (exp_bits < 2 || exp_bits > 8 || bits < exp_bits + 3 || bits > exp_bits + 24 || bits > 32)) return -1;
  std::vector<uint8_t> v;
  try {
    jxe::LosslessOptions o{flags, seed};
    jxe::EncodeLossless(px, xs, ys, channels, o, &v, bits, exp_bits);
  } catch (...) {
    return -2;
  }
  return Finish(v, out, n);
}

int jxlenc_encode_random(uint32_t xs, uint32_t ys, const JxlEncParams* p, uint8_t** out, size_t* n) {
  jxe::UseThreads();
  if (!xs || !ys || !p) return -1;
  jxe::Params q;
  memcpy(&q, p, sizeof(q));
  std::vector<uint8_t> v;
  try {
    jxe::EncodeRandom(xs, ys, q, &v);
  } catch (...) {
    return -2;
  }
  return Finish(v, out, n);
}

// 0 = back to the default
int jxlenc_set_threads(int n) {
  jxe::g_enc_threads = n > 0 ? n : 0;
  jxe::UseThreads();
  return omp_get_max_threads();
}
void jxlenc_free(uint8_t* p) { free(p); }

// Deterministic synthetic test image: dark gradient background, seeded rectangles, discs, texture and noise
// (structure after the reference's GetSomeTestImage, lib/jxl/test_image.cc:147-200; not the same pixels).
void jxlenc_synth_image(uint32_t xs, uint32_t ys, uint32_t seed, uint8_t* rgb) {
  jxe::Rng rng(seed);
  struct Shape { int x0, y0, x1, y1, kind; uint8_t col[3]; float freq; };
  std::vector<Shape> shapes(24 + (uint64_t(xs) * ys) / (512 * 512) * 6);
  for (auto& s : shapes) {
    int w = 16 + int(rng.Below(std::max(17u, xs / 3))), h = 16 + int(rng.Below(std::max(17u, ys / 3)));
    s.x0 = int(rng.Below(xs)) - w / 2;
    s.y0 = int(rng.Below(ys)) - h / 2;
    s.x1 = s.x0 + w;
    s.y1 = s.y0 + h;
    s.kind = int(rng.Below(4));
    for (auto& c : s.col) c = uint8_t(rng.Below(256));
    s.freq = 0.05f + rng.Uniform() * 0.9f;
  }
  // bucket shapes by 64-row bands to keep this O(pixels)
  std::vector<std::vector<int>> band((ys + 63) / 64);
  for (size_t i = 0; i < shapes.size(); i++)
    for (int b = std::max(0, shapes[i].y0 / 64); b <= std::min<int>(int(band.size()) - 1, shapes[i].y1 / 64); b++)
      band[b].push_back(int(i));
#pragma omp parallel for
  for (uint32_t y = 0; y < ys; y++) {
    jxe::Rng r2(seed * 7919u + y);
    for (uint32_t x = 0; x < xs; x++) {
      float v[3] = {20.0f + 60.0f * x / xs, 30.0f + 50.0f * y / ys, 25.0f + 40.0f * (x + y) / float(xs + ys)};
      for (int i : band[y / 64]) {
        const Shape& s = shapes[i];
        if (int(x) < s.x0 || int(x) >= s.x1 || int(y) < s.y0 || int(y) >= s.y1) continue;
        float cxm = 0.5f * (s.x0 + s.x1), cym = 0.5f * (s.y0 + s.y1);
        float dx = (x - cxm) / (0.5f * (s.x1 - s.x0)), dy = (y - cym) / (0.5f * (s.y1 - s.y0));
        if (s.kind == 1 && dx * dx + dy * dy > 1.0f) continue;
        for (int c = 0; c < 3; c++) {
          float t = s.col[c];
          if (s.kind == 2) t *= 0.5f + 0.5f * std::sin(s.freq * (x + 2 * y));  // texture
          if (s.kind == 3) t *= 1.0f - 0.5f * (dx * dx + dy * dy);             // soft blob
          v[c] = t;
        }
      }
      for (int c = 0; c < 3; c++) {
        float n = (r2.Uniform() - 0.5f) * 6.0f;
        rgb[(size_t(y) * xs + x) * 3 + c] = uint8_t(std::max(0.0f, std::min(255.0f, v[c] + n)));
      }
    }
  }
}

}  // extern "C"
