// ORACLE (test infrastructure): VarDCT — strategy tables, coefficient orders, dequantisation tables,
// block context map, AC coefficient decode, dequantisation + chroma-from-luma, low frequencies from DC,
// inverse transforms for all 27 strategies, DC dequantisation / adaptive smoothing, EPF sigma.
// Follows: reference lib/jxl/ac_strategy.{h,cc}, lib/jxl/coeff_order.{h,cc}, lib/jxl/quant_weights.{h,cc},
// lib/jxl/base/fast_math-inl.h:35-82, lib/jxl/entropy_coder.{h,cc}, lib/jxl/ac_context.h,
// lib/jxl/dec_group.cc:115-181,469-542, lib/jxl/quantizer.h:82-135, lib/jxl/quantizer-inl.h:34-67,
// lib/jxl/dec_transforms-inl.h, lib/jxl/dct-inl.h (definition of the scaled (I)DCT),
// lib/jxl/dct_scales.h:34-40, lib/jxl/compressed_dc.cc:50-296, lib/jxl/epf.cc:39-133.
//
// Transforms are evaluated from their closed-form definition with double accumulation (the reference's
// own tests pin its fast float implementation to this definition within 1e-7*N, dct_test.cc:165-216).
#ifndef JXLO_VARDCT_H_
#define JXLO_VARDCT_H_

#include <cmath>
#include <cstring>
#include <functional>
#include <vector>

#include "jxlo_entropy.h"
#include "jxlo_headers.h"

namespace jxlo {

static const int kNumStrategies = 27;
static const int kNumOrders = 13;
static const int kNumQuantTables = 17;
static const uint8_t kCoveredX[27] = {1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32};
static const uint8_t kCoveredY[27] = {1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16};
static const uint8_t kLog2Covered[27] = {0, 0, 0, 0, 2, 4, 1, 1, 2, 2, 3, 3, 0, 0, 0, 0, 0, 0, 6, 5, 5, 8, 7, 7, 10, 9, 9};
static const uint8_t kStrategyOrder[27] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
static const uint8_t kStrategyQuantTable[27] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};
static const uint16_t kCoeffOrderOffset[3 * 13 + 1] = {0,   1,   2,   3,   4,   5,    6,    10,   14,   18,   34,   50,   66,  68,
                                                        70,  72,  76,  80,  84,  92,   100,  108,  172,  236,  300,  332, 364, 396,
                                                        652, 908, 1164, 1292, 1420, 1548, 2572, 3596, 4620, 5132, 5644, 6156};
static const int kQTReqX[17] = {1, 1, 1, 1, 2, 4, 1, 1, 2, 1, 1, 8, 4, 16, 8, 32, 16};
static const int kQTReqY[17] = {1, 1, 1, 1, 2, 4, 2, 4, 4, 1, 1, 8, 8, 16, 16, 32, 32};

static inline size_t CoeffOrderOffset(int ord, int c) { return size_t(kCoeffOrderOffset[3 * ord + c]) * 64; }

// Natural (zig-zag) order of a strategy: order[k] = raster index of the k-th coefficient; the first
// cx*cy entries are the LLF corner (ac_strategy.cc:28-79).
static inline void NaturalOrder(int strategy, std::vector<uint32_t>* out) {
  size_t cx = kCoveredX[strategy], cy = kCoveredY[strategy];
  if (cy > cx) std::swap(cx, cy);  // coefficient layout: rows = short side
  out->assign(cx * cy * 64, 0);
  size_t xs = cx / cy, xsm = xs - 1, xss = CeilLog2(xs);
  size_t cur = cx * cy, n = cx * 8;
  for (size_t i = 0; i < n; i++) {
    for (size_t j = 0; j <= i; j++) {
      size_t x = j, y = i - j;
      if (i % 2) std::swap(x, y);
      if ((y & xsm) != 0) continue;
      y >>= xss;
      size_t val = (x < cx && y < cy) ? y * cx + x : cur++;
      (*out)[val] = uint32_t(y * n + x);
    }
  }
  for (size_t ip = n - 1; ip > 0; ip--) {
    size_t i = ip - 1;
    for (size_t j = 0; j <= i; j++) {
      size_t x = n - 1 - (i - j), y = n - 1 - j;
      if (i % 2) std::swap(x, y);
      if ((y & xsm) != 0) continue;
      y >>= xss;
      (*out)[cur++] = uint32_t(y * n + x);
    }
  }
}

// Representative strategy of each order bucket (first strategy mapping to it).
static inline int OrderBucketStrategy(int ord) {
  for (int s = 0; s < 27; s++)
    if (kStrategyOrder[s] == ord) return s;
  return 0;
}

// Decodes the coefficient orders of one pass (coeff_order.cc:102-156). `orders` has 6156*64 entries.
static inline void DecodeCoeffOrders(BitReader& br, uint32_t used_orders, uint32_t used_acs, std::vector<uint32_t>* orders) {
  orders->assign(size_t(6156) * 64, 0);
  EntropyCode code;
  std::unique_ptr<SymbolReader> rd;
  if (used_orders) {
    DecodeHistograms(br, 8, &code);
    rd.reset(new SymbolReader(&code, &br));
  }
  uint32_t acs_mask = 0;
  for (int s = 0; s < 27; s++)
    if (used_acs & (1u << s)) acs_mask |= 1u << kStrategyOrder[s];
  uint32_t computed = 0;
  for (int s = 0; s < 27; s++) {
    int ord = kStrategyOrder[s];
    if (computed & (1u << ord)) continue;
    computed |= 1u << ord;
    bool used = (acs_mask & (1u << ord)) != 0;
    size_t llf = size_t(kCoveredX[s]) * kCoveredY[s], size = 64 * llf;
    std::vector<uint32_t> natural;
    if (used || (used_orders & (1u << ord))) NaturalOrder(s, &natural);
    if (!(used_orders & (1u << ord))) {
      if (used)
        for (int c = 0; c < 3; c++) memcpy(&(*orders)[CoeffOrderOffset(ord, c)], natural.data(), size * 4);
    } else {
      for (int c = 0; c < 3; c++) {
        std::vector<uint32_t> perm;
        ReadPermutation(br, *rd, llf, size, &perm);
        if (used)
          for (size_t k = 0; k < size; k++) (*orders)[CoeffOrderOffset(ord, c) + k] = natural[perm[k]];
      }
    }
  }
  if (used_orders) JXLO_CHECK(rd->FinalStateOk(), "coefficient orders: bad ANS final state");
}

// ---- fast math used by the dequant-table generator (fast_math-inl.h:35-82)
static inline float FastLog2f(float x) {
  int32_t xb;
  memcpy(&xb, &x, 4);
  int32_t eb = xb - 0x3f2aaaab;
  int32_t es = eb >> 23;
  int32_t mb = xb - int32_t(uint32_t(es) << 23);  // (es may be negative)
  float m;
  memcpy(&m, &mb, 4);
  float ev = float(es);
  float t = m - 1.0f;
  float yp = 7.4245873327820566E-01f * t + 1.4287160470083755E+00f;
  yp = yp * t + -1.8503833400518310E-06f;
  float yq = 1.7409343003366853E-01f * t + 1.0096718572241148E+00f;
  yq = yq * t + 9.9032814277590719E-01f;
  return yp / yq + ev;
}
static inline float FastPow2f(float x) {
  float fl = std::floor(x);
  int32_t e = (int32_t(fl) + 127) << 23;
  float ex;
  memcpy(&ex, &e, 4);
  float frac = x - fl;
  float num = frac + 1.01749063e+01f;
  num = num * frac + 4.88687798e+01f;
  num = num * frac + 9.85506591e+01f;
  num = num * ex;
  float den = frac * 2.10242958e-01f + -2.22328856e-02f;
  den = den * frac + -1.94414990e+01f;
  den = den * frac + 9.85506633e+01f;
  return num / den;
}
static inline float FastPowf(float b, float e) { return FastPow2f(FastLog2f(b) * e); }

// ---- dequantisation tables
struct QuantEncoding {
  int mode = 0;  // 0 library, 1 ID, 2 DCT2, 3 DCT4, 4 DCT4X8, 5 AFV, 6 DCT, 7 RAW
  float idw[3][3] = {};
  float dct2w[3][6] = {};
  float dct4mul[3][2] = {};
  float dct4x8mul[3] = {};
  float afvw[3][9] = {};
  int nb = 0;
  float bands[3][17] = {};
  int nb4 = 0;
  float bands4[3][17] = {};  // AFV: 4x4 bands
  float qraw_den = 0.0f;      // RAW (quant_weights.cc:268-276, dec_modular.cc:795-841): weights = 1 / (qraw_den * qraw[i])
  std::vector<int32_t> qraw;  //   3 channels of 8 * required_size_x by 8 * required_size_y integers, as the Modular stream lays them out
};
// Reads the three channels (w x h each) of RAW table `kind` from the Modular sub-stream at the reader's position (stream id
// 1 + 3 * num_dc_groups + kind, with the frame's global tree: dec_modular.cc:805-819) into out[c * w * h + y * w + x].
typedef std::function<void(BitReader&, size_t w, size_t h, int kind, std::vector<int32_t>* out)> RawTableReader;

#include "quant_library.inc"

static inline QuantEncoding LibraryEncoding(int kind) {
  QuantEncoding e;
  e = QuantEncoding();
  static const int modes[17] = {6, 1, 2, 3, 6, 6, 6, 6, 6, 4, 5, 6, 6, 6, 6, 6, 6};
  e.mode = modes[kind];
  e.nb = kQLNumBands[kind];
  for (int c = 0; c < 3; c++) {
    for (int i = 0; i < e.nb; i++) e.bands[c][i] = float(kQLBands[kind][c][i]);
    for (int i = 0; i < 3; i++) e.idw[c][i] = float(kQLIdentity[c][i]);
    for (int i = 0; i < 6; i++) e.dct2w[c][i] = float(kQLDct2[c][i]);
    for (int i = 0; i < 2; i++) e.dct4mul[c][i] = float(kQLDct4Mul[c][i]);
    e.dct4x8mul[c] = float(kQLDct4x8Mul[c]);
    for (int i = 0; i < 9; i++) e.afvw[c][i] = float(kQLAfv[c][i]);
  }
  if (kind == 10) {  // AFV: 4x4 part uses the DCT4X4 bands
    e.nb4 = kQLNumBands[3];
    for (int c = 0; c < 3; c++)
      for (int i = 0; i < e.nb4; i++) e.bands4[c][i] = float(kQLBands[3][c][i]);
  }
  return e;
}

static inline void ReadDctParams(BitReader& br, int* nb, float bands[3][17]) {
  *nb = int(br.Read(4)) + 1;
  for (int c = 0; c < 3; c++) {
    for (int i = 0; i < *nb; i++) bands[c][i] = ReadF16(br);
    JXLO_CHECK(bands[c][0] >= 1e-8f, "distance band seed too small");
    bands[c][0] *= 64.0f;
  }
}

static inline void ReadQuantEncoding(BitReader& br, int kind, QuantEncoding* e, const RawTableReader* raw = nullptr) {
  int req = kQTReqX[kind] * kQTReqY[kind];
  int mode = int(br.Read(3));
  *e = QuantEncoding();
  e->mode = mode;
  switch (mode) {
    case 0:  // library: ceil(log2(kNumPredefinedTables = 1)) = 0 selector bits
      *e = LibraryEncoding(kind);
      break;
    case 1:
      JXLO_CHECK(req == 1, "invalid quant mode for table");
      for (int c = 0; c < 3; c++)
        for (int i = 0; i < 3; i++) {
          e->idw[c][i] = ReadF16(br);
          JXLO_CHECK(std::fabs(e->idw[c][i]) >= 1e-8f, "quantizer too small");
          e->idw[c][i] *= 64;
        }
      break;
    case 2:
      JXLO_CHECK(req == 1, "invalid quant mode for table");
      for (int c = 0; c < 3; c++)
        for (int i = 0; i < 6; i++) {
          e->dct2w[c][i] = ReadF16(br);
          JXLO_CHECK(std::fabs(e->dct2w[c][i]) >= 1e-8f, "quantizer too small");
          e->dct2w[c][i] *= 64;
        }
      break;
    case 4:
      JXLO_CHECK(req == 1, "invalid quant mode for table");
      for (int c = 0; c < 3; c++) {
        e->dct4x8mul[c] = ReadF16(br);
        JXLO_CHECK(std::fabs(e->dct4x8mul[c]) >= 1e-8f, "multiplier too small");
      }
      ReadDctParams(br, &e->nb, e->bands);
      break;
    case 3:
      JXLO_CHECK(req == 1, "invalid quant mode for table");
      for (int c = 0; c < 3; c++)
        for (int i = 0; i < 2; i++) {
          e->dct4mul[c][i] = ReadF16(br);
          JXLO_CHECK(std::fabs(e->dct4mul[c][i]) >= 1e-8f, "multiplier too small");
        }
      ReadDctParams(br, &e->nb, e->bands);
      break;
    case 5:
      JXLO_CHECK(req == 1, "invalid quant mode for table");
      for (int c = 0; c < 3; c++) {
        for (int i = 0; i < 9; i++) e->afvw[c][i] = ReadF16(br);
        for (int i = 0; i < 6; i++) e->afvw[c][i] *= 64;
      }
      ReadDctParams(br, &e->nb, e->bands);
      ReadDctParams(br, &e->nb4, e->bands4);
      break;
    case 6:
      ReadDctParams(br, &e->nb, e->bands);
      break;
    default: {  // RAW: a denominator and the table itself as a small Modular image (what JPEG recompression writes)
      if (!raw) throw Error("unsupported: RAW quant tables");
      const float den = ReadF16(br);
      JXLO_CHECK(den >= 1e-8f, "invalid qtable_den: value too small");
      std::vector<int32_t> q;
      (*raw)(br, size_t(8) * kQTReqX[kind], size_t(8) * kQTReqY[kind], kind, &q);
      JXLO_CHECK(q.size() == size_t(3) * 64 * req, "RAW quant table size");
      for (int32_t v : q) JXLO_CHECK(v > 0, "invalid raw quantization table");
      e->mode = 7;
      e->qraw_den = den;
      e->qraw = std::move(q);
      break;
    }
  }
}

static inline float BandMult(float v) { return v > 0.0f ? 1.0f + v : 1.0f / (1.0f - v); }

static inline void QuantWeights(size_t rows, size_t cols, const float dbands[3][17], int nb, float* out) {
  for (int c = 0; c < 3; c++) {
    float bands[17] = {dbands[c][0]};
    JXLO_CHECK(bands[0] >= 1e-8f, "invalid distance bands");
    for (int i = 1; i < nb; i++) {
      bands[i] = bands[i - 1] * BandMult(dbands[c][i]);
      JXLO_CHECK(bands[i] >= 1e-8f, "invalid distance bands");
    }
    float scale = (nb - 1) / (1.41421356237f + 1e-6f);
    float rcpcol = scale / (cols - 1), rcprow = scale / (rows - 1);
    for (size_t y = 0; y < rows; y++) {
      float dy = y * rcprow, dy2 = dy * dy;
      for (size_t x = 0; x < cols; x++) {
        float dx = float(x) * rcpcol;
        float dist = std::sqrt(dx * dx + dy2);
        float w;
        if (nb == 1) {
          w = bands[0];
        } else {
          int idx = int(dist);
          float frac = dist - float(idx);
          float a = bands[idx], b = bands[idx + 1];
          w = a * FastPowf(b / a, frac);
        }
        out[c * cols * rows + y * cols + x] = w;
      }
    }
  }
}

struct DequantTables {
  QuantEncoding enc[17];
  std::vector<float> table[17];  // [kind]: 3 * num floats (X, Y, B), computed lazily
  float dc_quant[3] = {1.0f / 4096, 1.0f / 512, 1.0f / 256};
  DequantTables() {
    for (int k = 0; k < 17; k++) enc[k] = LibraryEncoding(k);
  }
  void Compute(int kind) {
    if (!table[kind].empty()) return;
    const QuantEncoding& e = enc[kind];
    size_t wrows = 8 * kQTReqX[kind], wcols = 8 * kQTReqY[kind], num = wrows * wcols;
    std::vector<float> w(3 * num, 0.0f);
    switch (e.mode) {
      case 1:
        for (int c = 0; c < 3; c++) {
          for (int i = 0; i < 64; i++) w[64 * c + i] = e.idw[c][0];
          w[64 * c + 1] = e.idw[c][1];
          w[64 * c + 8] = e.idw[c][1];
          w[64 * c + 9] = e.idw[c][2];
        }
        break;
      case 2:
        for (int c = 0; c < 3; c++) {
          size_t s = c * 64;
          w[s] = 0xBAD;
          w[s + 1] = w[s + 8] = e.dct2w[c][0];
          w[s + 9] = e.dct2w[c][1];
          for (int y = 0; y < 2; y++)
            for (int x = 0; x < 2; x++) {
              w[s + y * 8 + x + 2] = e.dct2w[c][2];
              w[s + (y + 2) * 8 + x] = e.dct2w[c][2];
              w[s + (y + 2) * 8 + x + 2] = e.dct2w[c][3];
            }
          for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
              w[s + y * 8 + x + 4] = e.dct2w[c][4];
              w[s + (y + 4) * 8 + x] = e.dct2w[c][4];
              w[s + (y + 4) * 8 + x + 4] = e.dct2w[c][5];
            }
        }
        break;
      case 3: {
        float w44[3 * 16];
        QuantWeights(4, 4, e.bands, e.nb, w44);
        for (int c = 0; c < 3; c++) {
          for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) w[c * 64 + y * 8 + x] = w44[c * 16 + (y / 2) * 4 + (x / 2)];
          w[c * 64 + 1] /= e.dct4mul[c][0];
          w[c * 64 + 8] /= e.dct4mul[c][0];
          w[c * 64 + 9] /= e.dct4mul[c][1];
        }
        break;
      }
      case 4: {
        float w48[3 * 32];
        QuantWeights(4, 8, e.bands, e.nb, w48);
        for (int c = 0; c < 3; c++) {
          for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) w[c * 64 + y * 8 + x] = w48[c * 32 + (y / 2) * 8 + x];
          w[c * 64 + 8] /= e.dct4x8mul[c];
        }
        break;
      }
      case 6:
        QuantWeights(wrows, wcols, e.bands, e.nb, w.data());
        break;
      case 5: {
        static const float kFreqs[16] = {0xBAD, 0xBAD, 0.8517778890324296f, 5.37778436506804f,
                                         0xBAD, 0xBAD, 4.734747904497923f, 5.449245381693219f,
                                         1.6598270267479331f, 4, 7.275749096817861f, 10.423227632456525f,
                                         2.662932286148962f, 7.630657783650829f, 8.962388608184032f, 12.97166202570235f};
        float w48[3 * 32], w44[3 * 16];
        QuantWeights(4, 8, e.bands, e.nb, w48);
        QuantWeights(4, 4, e.bands4, e.nb4, w44);
        const float lo = 0.8517778890324296f;
        const float hi = 12.97166202570235f - lo + 1e-6f;
        for (int c = 0; c < 3; c++) {
          float bands[4];
          bands[0] = e.afvw[c][5];
          JXLO_CHECK(bands[0] >= 1e-8f, "invalid AFV bands");
          for (int i = 1; i < 4; i++) {
            bands[i] = bands[i - 1] * BandMult(e.afvw[c][i + 5]);
            JXLO_CHECK(bands[i] >= 1e-8f, "invalid AFV bands");
          }
          size_t s = c * 64;
          w[s] = 1;
          w[s + 1 * 8 + 0] = e.afvw[c][0];  // (x=0,y=1)
          w[s + 0 * 8 + 1] = e.afvw[c][1];  // (x=1,y=0)
          w[s + 2 * 8 + 0] = e.afvw[c][2];  // (0,2)
          w[s + 0 * 8 + 2] = e.afvw[c][3];  // (2,0)
          w[s + 2 * 8 + 2] = e.afvw[c][4];  // (2,2)
          for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
              if (x < 2 && y < 2) continue;
              float pos = kFreqs[y * 4 + x] - lo;
              float sp = pos * 3 / hi;
              size_t idx = size_t(sp);
              JXLO_CHECK(idx + 1 < 4, "AFV interpolation range");
              float a = bands[idx], b = bands[idx + 1];
              w[s + (2 * y) * 8 + 2 * x] = a * FastPowf(b / a, sp - idx);
            }
          for (int y = 0; y < 4; y++)
            for (int x = 0; x < 8; x++) {
              if (x == 0 && y == 0) continue;
              w[s + (2 * y + 1) * 8 + x] = w48[c * 32 + y * 8 + x];
            }
          for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
              if (x == 0 && y == 0) continue;
              w[s + (2 * y) * 8 + 2 * x + 1] = w44[c * 16 + y * 4 + x];
            }
        }
        break;
      }
      case 7:
        JXLO_CHECK(e.qraw.size() == 3 * num, "RAW quant table size");
        for (size_t i = 0; i < 3 * num; i++) w[i] = 1.0f / (e.qraw_den * float(e.qraw[i]));
        break;
      default:
        throw Error("unsupported quant table mode");
    }
    for (size_t i = 0; i < 3 * num; i++) {
      JXLO_CHECK(w[i] < 1.0f / 1e-8f && w[i] >= 1e-8f, "invalid quantization table");
      w[i] = 1.0f / w[i];
    }
    table[kind] = std::move(w);
  }
  // Dequant matrix (multipliers) for a strategy and channel.
  const float* Matrix(int strategy, int c) {
    int k = kStrategyQuantTable[strategy];
    Compute(k);
    return table[k].data() + c * (table[k].size() / 3);
  }
};

// ---- block context map (entropy_coder.cc:25-60, ac_context.h:86-150)
struct BlockCtxMap {
  std::vector<int32_t> dc_thresholds[3];
  std::vector<uint32_t> qf_thresholds;
  std::vector<uint8_t> ctx_map;
  size_t num_ctxs = 15, num_dc_ctxs = 1;
  BlockCtxMap() {
    static const uint8_t kDefault[39] = {0, 1, 2, 2, 3,  3,  4,  5,  6,  6,  6,  6,  6,  7, 8, 9, 9, 10, 11, 12,
                                         13, 14, 14, 14, 14, 14, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14};
    ctx_map.assign(kDefault, kDefault + 39);
  }
  size_t Context(int dc_idx, uint32_t qf, size_t ord, size_t c) const {
    size_t qf_idx = 0;
    for (uint32_t t : qf_thresholds)
      if (qf > t) qf_idx++;
    size_t idx = c < 2 ? c ^ 1 : 2;
    idx = idx * kNumOrders + ord;
    idx = idx * (qf_thresholds.size() + 1) + qf_idx;
    idx = idx * num_dc_ctxs + dc_idx;
    return ctx_map[idx];
  }
  size_t NumACContexts() const { return num_ctxs * (37 + 458); }
  size_t ZeroDensityOffset(size_t block_ctx) const { return num_ctxs * 37 + 458 * block_ctx; }
  size_t NonZeroContext(uint32_t nz, size_t block_ctx) const {
    if (nz >= 64) nz = 64;
    uint32_t ctx = nz < 8 ? nz : 4 + nz / 2;
    return ctx * num_ctxs + block_ctx;
  }
};

static inline void ReadBlockCtxMap(BitReader& br, BlockCtxMap* m) {
  if (br.ReadBool()) {
    *m = BlockCtxMap();
    return;
  }
  m->num_dc_ctxs = 1;
  for (int j = 0; j < 3; j++) {
    m->dc_thresholds[j].resize(br.Read(4));
    m->num_dc_ctxs *= m->dc_thresholds[j].size() + 1;
    for (auto& t : m->dc_thresholds[j])
      t = UnpackSigned(ReadU32(br, Bits(4), BitsOffset(8, 16), BitsOffset(16, 272), BitsOffset(32, 65808)));
  }
  m->qf_thresholds.resize(br.Read(4));
  for (auto& t : m->qf_thresholds) t = ReadU32(br, Bits(2), BitsOffset(3, 4), BitsOffset(5, 12), BitsOffset(8, 44)) + 1;
  JXLO_CHECK(m->num_dc_ctxs * (m->qf_thresholds.size() + 1) <= 64, "block context map too big");
  m->ctx_map.assign(3 * kNumOrders * m->num_dc_ctxs * (m->qf_thresholds.size() + 1), 0);
  DecodeContextMap(br, &m->ctx_map, &m->num_ctxs);
  JXLO_CHECK(m->num_ctxs <= 16, "too many block contexts");
}

static const uint16_t kCoeffFreqContext[64] = {
    0xBAD, 0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 15, 16, 16, 17, 17,
    18,    18, 19, 19, 20, 20, 21, 21, 22, 22, 23, 23, 23, 23, 24, 24, 24, 24, 25, 25, 25, 25,
    26,    26, 26, 26, 27, 27, 27, 27, 28, 28, 28, 28, 29, 29, 29, 29, 30, 30, 30, 30};
static const uint16_t kCoeffNumNonzeroContext[64] = {
    0xBAD, 0,   31,  62,  62,  93,  93,  93,  93,  123, 123, 123, 123, 152, 152, 152, 152, 152, 152, 152, 152, 180,
    180,   180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206,
    206,   206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206};

static inline size_t ZeroDensityContext(size_t nz_left, size_t k, size_t covered, size_t log2_covered, size_t prev) {
  nz_left = (nz_left + covered - 1) >> log2_covered;
  k >>= log2_covered;
  return (kCoeffNumNonzeroContext[nz_left & 63] + kCoeffFreqContext[k & 63]) * 2 + prev;
}

// ---- transforms (closed form)
// Basis of the scaled DCT: C_N(n,k) = (k ? sqrt(2) : 1) * cos((n+1/2) k pi / N).
struct DctBasis {
  std::vector<double> m[9];  // index log2(N), N = 1..256
  DctBasis() {
    for (int l = 0; l <= 8; l++) {
      int N = 1 << l;
      m[l].resize(size_t(N) * N);
      for (int n = 0; n < N; n++)
        for (int k = 0; k < N; k++)
          m[l][size_t(n) * N + k] = (k ? std::sqrt(2.0) : 1.0) * std::cos((n + 0.5) * k * M_PI / N);
    }
  }
  const double* Get(int N) const { return m[FloorLog2(N)].data(); }
};
static inline const DctBasis& Basis() {
  static const DctBasis b;
  return b;
}

// Inverse scaled DCT of an R x C pixel block. Coefficient layout as in the codestream: rows = the short
// side; if R < C element (ky,kx) is at [ky*C+kx], otherwise at [kx*R+ky] (dct-inl.h:376-397).
static inline void Idct2D(const float* coef, int R, int C, float* out, size_t stride) {
  const double* br = Basis().Get(R);
  const double* bc = Basis().Get(C);
  std::vector<double> tmp(size_t(R) * C);  // tmp[ky][x]
  for (int ky = 0; ky < R; ky++)
    for (int x = 0; x < C; x++) {
      double s = 0;
      for (int kx = 0; kx < C; kx++) {
        float v = (R < C) ? coef[ky * C + kx] : coef[kx * R + ky];
        if (v != 0.0f) s += double(v) * bc[size_t(x) * C + kx];
      }
      tmp[size_t(ky) * C + x] = s;
    }
  for (int y = 0; y < R; y++)
    for (int x = 0; x < C; x++) {
      double s = 0;
      for (int ky = 0; ky < R; ky++) s += tmp[size_t(ky) * C + x] * br[size_t(y) * R + ky];
      out[y * stride + x] = float(s);
    }
}

// Forward scaled DCT (coefficients = averages; DC = mean), same layout rule. Used for LLF-from-DC.
static inline void Dct2D(const float* in, size_t in_stride, int R, int C, float* coef) {
  const double* br = Basis().Get(R);
  const double* bc = Basis().Get(C);
  for (int ky = 0; ky < R; ky++)
    for (int kx = 0; kx < C; kx++) {
      double s = 0;
      for (int y = 0; y < R; y++)
        for (int x = 0; x < C; x++) s += double(in[y * in_stride + x]) * br[size_t(y) * R + ky] * bc[size_t(x) * C + kx];
      s /= double(R) * C;
      if (R < C) coef[ky * C + kx] = float(s);
      else coef[kx * R + ky] = float(s);
    }
}

// Scale that maps coefficient i of an n-point DCT onto coefficient i of an 8n-point DCT (dct_scales.h:34-40).
static inline float ResampleScale(int n, int i) {
  double N = 8.0 * n;
  double v = std::cos(i / (2 * N) * M_PI) * std::cos(i / N * M_PI) * std::cos(i / (N / 2) * M_PI);
  return float(1.0 / v);
}

#include "afv_basis.inc"

// Fills the lowest-frequency corner of a coefficient block from the DC image (dec_transforms-inl.h:691-818).
static inline void LowestFrequenciesFromDC(int strategy, const float* dc, size_t dc_stride, float* block) {
  int cx = kCoveredX[strategy], cy = kCoveredY[strategy];
  if (cx == 1 && cy == 1) {
    block[0] = dc[0];
    return;
  }
  std::vector<float> f(size_t(cx) * cy);
  Dct2D(dc, dc_stride, cy, cx, f.data());
  size_t stride = size_t(std::max(cx, cy)) * 8;
  if (cy < cx) {
    for (int y = 0; y < cy; y++)
      for (int x = 0; x < cx; x++) block[y * stride + x] = f[y * cx + x] * ResampleScale(cy, y) * ResampleScale(cx, x);
  } else {
    for (int y = 0; y < cx; y++)
      for (int x = 0; x < cy; x++) block[y * stride + x] = f[y * cy + x] * ResampleScale(cx, y) * ResampleScale(cy, x);
  }
}

static inline void AfvToPixels(int kind, const float* coefficients, float* pixels, size_t stride) {
  int afv_x = kind & 1, afv_y = kind / 2;
  float b00 = coefficients[0], b01 = coefficients[1], b10 = coefficients[8];
  float dcs[3] = {(b00 + b10 + b01) * 4.0f, (b00 + b10 - b01), b00 - b10};
  float coeff[16];
  coeff[0] = dcs[0];
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++)
      if (ix || iy) coeff[iy * 4 + ix] = coefficients[iy * 2 * 8 + ix * 2];
  float block[32];
  for (int i = 0; i < 16; i++) {
    double p = 0;
    for (int j = 0; j < 16; j++) p += double(coeff[j]) * double(kAfvBasis[j][i]);
    block[i] = float(p);
  }
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++)
      pixels[(iy + afv_y * 4) * stride + afv_x * 4 + ix] = block[(afv_y == 1 ? 3 - iy : iy) * 4 + (afv_x == 1 ? 3 - ix : ix)];
  block[0] = dcs[1];
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++)
      if (ix || iy) block[iy * 4 + ix] = coefficients[iy * 2 * 8 + ix * 2 + 1];
  Idct2D(block, 4, 4, pixels + afv_y * 4 * stride + (afv_x == 1 ? 0 : 4), stride);
  block[0] = dcs[2];
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 8; ix++)
      if (ix || iy) block[iy * 8 + ix] = coefficients[(1 + iy * 2) * 8 + ix];
  Idct2D(block, 4, 8, pixels + (afv_y == 1 ? 0 : 4) * stride, stride);
}

// Coefficients -> pixels for one varblock (dec_transforms-inl.h:456-689).
static inline void TransformToPixels(int strategy, const float* coefficients, float* pixels, size_t stride) {
  switch (strategy) {
    case 1: {  // IDENTITY
      float b00 = coefficients[0], b01 = coefficients[1], b10 = coefficients[8], b11 = coefficients[9];
      float dcs[4] = {b00 + b01 + b10 + b11, b00 + b01 - b10 - b11, b00 - b01 + b10 - b11, b00 - b01 - b10 + b11};
      for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
          float block_dc = dcs[y * 2 + x];
          float residual_sum = 0;
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++)
              if (ix || iy) residual_sum += coefficients[(y + iy * 2) * 8 + x + ix * 2];
          float base = block_dc - residual_sum * (1.0f / 16);
          pixels[(4 * y + 1) * stride + 4 * x + 1] = base;
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++) {
              if (ix == 1 && iy == 1) continue;
              pixels[(y * 4 + iy) * stride + x * 4 + ix] = coefficients[(y + iy * 2) * 8 + x + ix * 2] + base;
            }
          pixels[y * 4 * stride + x * 4] = coefficients[(y + 2) * 8 + x + 2] + base;
        }
      break;
    }
    case 13: {  // DCT8X4: two 8-row x 4-col halves side by side
      float b0 = coefficients[0], b1 = coefficients[8];
      float dcs[2] = {b0 + b1, b0 - b1};
      for (int x = 0; x < 2; x++) {
        float block[32];
        block[0] = dcs[x];
        for (int iy = 0; iy < 4; iy++)
          for (int ix = 0; ix < 8; ix++)
            if (ix || iy) block[iy * 8 + ix] = coefficients[(x + iy * 2) * 8 + ix];
        Idct2D(block, 8, 4, pixels + x * 4, stride);
      }
      break;
    }
    case 12: {  // DCT4X8: two 4-row x 8-col halves stacked
      float b0 = coefficients[0], b1 = coefficients[8];
      float dcs[2] = {b0 + b1, b0 - b1};
      for (int y = 0; y < 2; y++) {
        float block[32];
        block[0] = dcs[y];
        for (int iy = 0; iy < 4; iy++)
          for (int ix = 0; ix < 8; ix++)
            if (ix || iy) block[iy * 8 + ix] = coefficients[(y + iy * 2) * 8 + ix];
        Idct2D(block, 4, 8, pixels + y * 4 * stride, stride);
      }
      break;
    }
    case 3: {  // DCT4X4
      float b00 = coefficients[0], b01 = coefficients[1], b10 = coefficients[8], b11 = coefficients[9];
      float dcs[4] = {b00 + b01 + b10 + b11, b00 + b01 - b10 - b11, b00 - b01 + b10 - b11, b00 - b01 - b10 + b11};
      for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
          float block[16];
          block[0] = dcs[y * 2 + x];
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++)
              if (ix || iy) block[iy * 4 + ix] = coefficients[(y + iy * 2) * 8 + x + ix * 2];
          Idct2D(block, 4, 4, pixels + y * 4 * stride + x * 4, stride);
        }
      break;
    }
    case 2: {  // DCT2X2: three nested 2x2 Hadamard stages
      float c[64], t[64];
      memcpy(c, coefficients, sizeof(c));
      for (int S = 2; S <= 8; S *= 2) {
        int h = S / 2;
        memcpy(t, c, sizeof(t));
        for (int y = 0; y < h; y++)
          for (int x = 0; x < h; x++) {
            float c00 = c[y * 8 + x], c01 = c[y * 8 + h + x], c10 = c[(y + h) * 8 + x], c11 = c[(y + h) * 8 + h + x];
            t[y * 2 * 8 + x * 2] = c00 + c01 + c10 + c11;
            t[y * 2 * 8 + x * 2 + 1] = c00 + c01 - c10 - c11;
            t[(y * 2 + 1) * 8 + x * 2] = c00 - c01 + c10 - c11;
            t[(y * 2 + 1) * 8 + x * 2 + 1] = c00 - c01 - c10 + c11;
          }
        for (int y = 0; y < S; y++)
          for (int x = 0; x < S; x++) c[y * 8 + x] = t[y * 8 + x];
      }
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) pixels[y * stride + x] = c[y * 8 + x];
      break;
    }
    case 14: case 15: case 16: case 17:
      AfvToPixels(strategy - 14, coefficients, pixels, stride);
      break;
    default:
      Idct2D(coefficients, kCoveredY[strategy] * 8, kCoveredX[strategy] * 8, pixels, stride);
  }
}

// Dequantisation bias (quantizer-inl.h:34-67; scalar-target arithmetic).
static inline float AdjustQuantBias(int c, int32_t q, const float* biases) {
  if (q == 0) return 0.0f;
  if (q == 1) return biases[c];
  if (q == -1) return -biases[c];
  float qf = float(q);
  return qf - biases[3] * (1.0f / qf);
}

}  // namespace jxlo
#endif  // JXLO_VARDCT_H_
