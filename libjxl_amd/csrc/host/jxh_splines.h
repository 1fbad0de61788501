// Splines of a frame (host half): the dictionary is entropy-decoded and dequantised here, every spline is sampled at unit
// arc length into Gaussian-like segments, and the per-row segment lists the device kernel draws from (k_splines_add,
// csrc/hip/jxl_hip_filter_fused.h) are laid out flat. Follows the reference:
//   dictionary     lib/jxl/splines.cc:541-648 (Splines::Decode, QuantizedSpline::Decode, DecodeAllStartingPoints :272-300)
//   dequantisation lib/jxl/splines.cc:439-536 (QuantizedSpline::Dequantize; channel weights :270)
//   geometry       lib/jxl/splines.cc:326-405 (centripetal Catmull-Rom, equally spaced points), :661-768 (draw cache)
//   segments       lib/jxl/splines.cc:55-82 (ContinuousIDCT), :148-176 (ComputeSegments); FastCosf fast_math-inl.h:95-124
#ifndef JXH_SPLINES_H_
#define JXH_SPLINES_H_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "jxh_bits.h"
#include "jxh_entropy.h"

namespace jxh {

struct SplinePoint {
  float x, y;
};
struct QuantSpline {
  std::vector<std::pair<int64_t, int64_t>> deltas;  // double deltas of the control points behind the starting point
  int32_t color_dct[3][32];
  int32_t sigma_dct[32];
};
struct Splines {
  int32_t quant_adjust = 0;
  std::vector<SplinePoint> start;
  std::vector<QuantSpline> splines;
  // draw cache, in the layout of JxlHipFrameDesc: 8 floats per segment {center_x, center_y, maximum_distance, inv_sigma,
  // sigma_over_4_times_intensity, colour X, Y, B}; row y draws segments row_segments[row_start[y] .. row_start[y + 1])
  std::vector<float> segments;
  std::vector<uint32_t> row_segments;
  std::vector<uint32_t> row_start;  // [ysize + 1]
};

static inline float SplineFastCos(float x) {
  const float kPi = 3.14159265358979323846f;
  const float pi2 = kPi * 2.0f;
  const float npi2 = std::floor(x * (0.5f / kPi)) * pi2;
  const float xmodpi2 = x - npi2;
  const float x_pi = std::min(xmodpi2, pi2 - xmodpi2);
  const bool above = x_pi >= kPi / 2.0f;
  const float x_pihalf = above ? kPi - x_pi : x_pi;
  const float xs = x_pihalf * 0.25f;
  const float x2 = xs * xs, x4 = x2 * x2;
  const float pre = std::fma(x4, 0.06960438f, std::fma(x2, -0.84087373f, 1.68179268f));
  const float s1 = std::fma(pre, pre, -1.414213562f);
  const float s2 = std::fma(s1, s1, -1.0f);
  return above ? -s2 : s2;
}
// splines.cc:55-82: cosine interpolation of 32 DCT coefficients at position t in [0, 31].
static inline float ContinuousIDCT(const float* dct, float t) {
  const float kPi = 3.14159265358979323846f;
  float result = 0.0f;
  for (int i = 0; i < 32; i++) result = std::fma(1.41421356237f, dct[i] * SplineFastCos((kPi / 32 * i) * (t + 0.5f)), result);
  return result;
}

static inline int64_t SplineUnpackSigned(uint64_t v) { return (v & 1) ? -int64_t((v + 1) >> 1) : int64_t(v >> 1); }

static inline void SplinePosCheck(double x, double y) {
  const double lim = double(1u << 23);
  JXH_CHECK(x < lim && x > -lim && y < lim && y > -lim, "spline coordinates out of bounds");
}

static inline void DecodeSplines(BitReader& br, size_t num_pixels, Splines* out) {
  EntropyCode code;
  DecodeHistograms(br, 6, &code);
  SymbolReader rd(&code, &br);
  size_t num_splines = rd.Read(2);
  const size_t max_cp = std::min<size_t>(size_t(1) << 20, num_pixels / 2);
  JXH_CHECK(num_splines <= max_cp && num_splines + 1 <= max_cp, "too many splines");
  num_splines++;
  int64_t lx = 0, ly = 0;
  for (size_t i = 0; i < num_splines; i++) {
    const uint32_t dx = rd.Read(1), dy = rd.Read(1);
    int64_t x, y;
    if (i) {
      x = SplineUnpackSigned(dx) + lx;
      y = SplineUnpackSigned(dy) + ly;
    } else {
      x = dx;
      y = dy;
    }
    SplinePosCheck(double(x), double(y));
    out->start.push_back({float(x), float(y)});
    lx = x;
    ly = y;
  }
  out->quant_adjust = int32_t(SplineUnpackSigned(rd.Read(0)));
  size_t total_cp = num_splines;
  for (size_t i = 0; i < num_splines; i++) {
    QuantSpline q;
    const size_t n = rd.Read(3);
    JXH_CHECK(n <= max_cp, "too many control points");
    total_cp += n;
    JXH_CHECK(total_cp <= max_cp, "too many control points");
    q.deltas.resize(n);
    for (auto& d : q.deltas) {
      d.first = SplineUnpackSigned(rd.Read(4));
      d.second = SplineUnpackSigned(rd.Read(4));
      const int64_t lim = int64_t(1) << 30;
      JXH_CHECK(d.first < lim && d.first > -lim && d.second < lim && d.second > -lim, "spline delta out of bounds");
    }
    for (int c = 0; c < 4; c++)
      for (int k = 0; k < 32; k++) {
        const int64_t v = SplineUnpackSigned(rd.Read(5));
        JXH_CHECK(v != INT32_MIN, "weird number in spline DCT");
        (c < 3 ? q.color_dct[c][k] : q.sigma_dct[k]) = int32_t(v);
      }
    out->splines.push_back(std::move(q));
  }
  JXH_CHECK(rd.FinalStateOk(), "splines: bad ANS final state");
}

// splines.cc:661-768: dequantise, sample every spline at unit arc length, one Gaussian-like segment per sample.
static inline void InitSplineDrawCache(Splines* s, size_t xsize, size_t ysize, float y_to_x, float y_to_b) {
  static const float kChannelWeight[4] = {0.0042f, 0.075f, 0.07f, 0.3333f};
  const float inv_quant = s->quant_adjust >= 0 ? 1.0f / (1.0f + 0.125f * s->quant_adjust) : 1.0f - 0.125f * s->quant_adjust;
  std::vector<std::pair<size_t, size_t>> spans;
  const uint64_t image_size = uint64_t(xsize) * ysize;
  const uint64_t area_limit = std::min<uint64_t>(1024 * image_size + (uint64_t(1) << 32), uint64_t(1) << 42);
  uint64_t total_area = 0;
  for (size_t si = 0; si < s->splines.size(); si++) {
    const QuantSpline& q = s->splines[si];
    // control points
    std::vector<SplinePoint> cp;
    int cx = int(std::round(s->start[si].x)), cy = int(std::round(s->start[si].y));
    cp.push_back({float(cx), float(cy)});
    int ddx = 0, ddy = 0;
    uint64_t manhattan = 0;
    for (const auto& d : q.deltas) {
      ddx += int(d.first);
      ddy += int(d.second);
      manhattan += uint64_t(std::abs(ddx)) + uint64_t(std::abs(ddy));
      JXH_CHECK(manhattan <= area_limit, "spline too long");
      SplinePosCheck(ddx, ddy);
      cx += ddx;
      cy += ddy;
      SplinePosCheck(cx, cy);
      cp.push_back({float(cx), float(cy)});
    }
    float color_dct[3][32], sigma_dct[32];
    for (int c = 0; c < 3; c++)
      for (int i = 0; i < 32; i++) color_dct[c][i] = float(q.color_dct[c][i]) * (i == 0 ? 0.70710678118f : 1.0f) * kChannelWeight[c] * inv_quant;
    for (int i = 0; i < 32; i++) {
      color_dct[0][i] += y_to_x * color_dct[1][i];
      color_dct[2][i] += y_to_b * color_dct[1][i];
    }
    // the reference's area estimate (a decode-time limit, not rendering)
    uint64_t color[3] = {0, 0, 0};
    for (int c = 0; c < 3; c++)
      for (int i = 0; i < 32; i++) color[c] += uint64_t(std::ceil(inv_quant * std::abs(float(q.color_dct[c][i]))));
    color[0] += uint64_t(std::ceil(std::abs(y_to_x))) * color[1];
    color[2] += uint64_t(std::ceil(std::abs(y_to_b))) * color[1];
    const uint64_t max_color = std::max({color[1], color[0], color[2]});
    uint64_t logcolor = 1;
    while ((uint64_t(1) << logcolor) < 1 + max_color) logcolor++;
    if (1 + max_color <= 1) logcolor = 1;
    const float weight_limit = std::ceil(std::sqrt((float(area_limit) / float(logcolor)) / float(std::max<uint64_t>(1, manhattan))));
    uint64_t width_estimate = 0;
    for (int i = 0; i < 32; i++) {
      sigma_dct[i] = float(q.sigma_dct[i]) * (i == 0 ? 0.70710678118f : 1.0f) * kChannelWeight[3] * inv_quant;
      const float wf = std::ceil(inv_quant * std::abs(float(q.sigma_dct[i])));
      const uint64_t w = uint64_t(std::min(weight_limit, std::max(1.0f, wf)));
      width_estimate += w * w * logcolor;
    }
    total_area += width_estimate * manhattan;
    JXH_CHECK(total_area <= area_limit, "splines cover too large an area");
    // The reference only warns beyond this (splines.cc:688-698) and fails in its fuzzing build: a few bytes of stream can
    // otherwise ask for gigabytes of segments. Here it is a refusal.
    JXH_CHECK(total_area <= std::min<uint64_t>(8 * image_size + (uint64_t(1) << 25), uint64_t(1) << 30), "splines cover too large an area");
    for (size_t i = 0; i + 1 < cp.size(); i++) JXH_CHECK(cp[i].x != cp[i + 1].x || cp[i].y != cp[i + 1].y, "identical successive control points");
    // centripetal Catmull-Rom through the control points, 16 samples per span (splines.cc:326-367)
    std::vector<SplinePoint> inter;
    if (cp.size() == 1) {
      inter.push_back(cp[0]);
    } else {
      std::vector<SplinePoint> p = cp;
      p.insert(p.begin(), {cp[0].x + (cp[0].x - cp[1].x), cp[0].y + (cp[0].y - cp[1].y)});
      const size_t n = p.size();
      p.push_back({p[n - 1].x + (p[n - 1].x - p[n - 2].x), p[n - 1].y + (p[n - 1].y - p[n - 2].y)});
      for (size_t st = 0; st + 3 < p.size(); st++) {
        const SplinePoint* q4 = &p[st];
        inter.push_back(q4[1]);
        float d[3], t[4];
        t[0] = 0;
        for (int k = 0; k < 3; k++) {
          d[k] = std::sqrt(hypotf(q4[k + 1].x - q4[k].x, q4[k + 1].y - q4[k].y));
          t[k + 1] = t[k] + d[k];
        }
        for (int i = 1; i < 16; i++) {
          const float tt = d[0] + (float(i) / 16) * d[1];
          SplinePoint a[3], b[2];
          for (int k = 0; k < 3; k++) {
            const float f = (tt - t[k]) / d[k];
            a[k] = {q4[k].x + f * (q4[k + 1].x - q4[k].x), q4[k].y + f * (q4[k + 1].y - q4[k].y)};
          }
          for (int k = 0; k < 2; k++) {
            const float f = (tt - t[k]) / (d[k] + d[k + 1]);
            b[k] = {a[k].x + f * (a[k + 1].x - a[k].x), a[k].y + f * (a[k + 1].y - a[k].y)};
          }
          const float f = (tt - t[1]) / d[1];
          inter.push_back({b[0].x + f * (b[1].x - b[0].x), b[0].y + f * (b[1].y - b[0].y)});
        }
      }
      inter.push_back(p[p.size() - 2]);
    }
    // equally spaced points along the polyline (splines.cc:374-405)
    std::vector<std::pair<SplinePoint, float>> draw;
    {
      SplinePoint current = inter.front();
      draw.push_back({current, 1.0f});
      size_t next = 0;
      bool done = false;
      while (next < inter.size() && !done) {
        const SplinePoint* previous = &current;
        float from_previous = 0.0f;
        for (;;) {
          if (next == inter.size()) {
            draw.push_back({*previous, from_previous});
            done = true;
            break;
          }
          const float dxn = inter[next].x - previous->x, dyn = inter[next].y - previous->y;
          const float to_next = std::sqrt(dxn * dxn + dyn * dyn);
          if (from_previous + to_next >= 1.0f) {
            const float f = (1.0f - from_previous) / to_next;
            current = {previous->x + f * dxn, previous->y + f * dyn};
            JXH_CHECK(draw.size() < (size_t(1) << 22), "spline too long");
            draw.push_back({current, 1.0f});
            break;
          }
          from_previous += to_next;
          previous = &inter[next];
          next++;
        }
      }
    }
    const float arc_length = float(draw.size() - 2) * 1.0f + draw.back().second;
    if (arc_length <= 0.0f) continue;
    const float inv_arc = 1.0f / arc_length;
    int k = 0;
    for (const auto& pd : draw) {
      const float progress = std::min(1.0f, float(k) * inv_arc);
      k++;
      float colr[3];
      for (int c = 0; c < 3; c++) colr[c] = ContinuousIDCT(color_dct[c], 31.0f * progress);
      const float sigma = ContinuousIDCT(sigma_dct, 31.0f * progress);
      const float intensity = pd.second;
      if (!(std::isfinite(sigma) && sigma != 0.0f && std::isfinite(1.0f / sigma) && std::isfinite(intensity))) continue;
      float max_color = 0.01f;
      for (int c = 0; c < 3; c++) max_color = std::max(max_color, std::abs(colr[c] * intensity));
      const float kDistanceExp = 5;  // JXL_HIGH_PRECISION
      const float maxd = std::sqrt(-2.0f * sigma * sigma * (std::log(0.1f) * kDistanceExp - std::log(max_color)));
      long long y0 = std::llround(pd.first.y - maxd);
      y0 = std::max<long long>(y0, 0);
      long long y1 = std::llround(pd.first.y + maxd) + 1;
      y1 = std::min<long long>(y1, (long long)ysize);
      if (y1 <= y0) continue;
      JXH_CHECK(s->segments.size() < size_t(8) << 22, "too many spline segments");
      const float seg[8] = {pd.first.x, pd.first.y, maxd, 1.0f / sigma, 0.25f * sigma * intensity, colr[0], colr[1], colr[2]};
      s->segments.insert(s->segments.end(), seg, seg + 8);
      spans.push_back({size_t(y0), size_t(y1)});
    }
  }
  // per-row lists in segment order (splines.cc:730-766)
  s->row_start.assign(ysize + 1, 0);
  uint64_t total_rows = 0;
  for (const auto& sp : spans) total_rows += sp.second - sp.first;
  JXH_CHECK(total_rows < (uint64_t(1) << 26), "splines cover too large an area");
  for (const auto& sp : spans)
    for (size_t y = sp.first; y < sp.second; y++) s->row_start[y + 1]++;
  for (size_t y = 0; y < ysize; y++) s->row_start[y + 1] += s->row_start[y];
  JXH_CHECK(s->row_start[ysize] < (1u << 26), "splines cover too large an area");
  s->row_segments.assign(s->row_start[ysize], 0);
  std::vector<uint32_t> fill(s->row_start.begin(), s->row_start.end() - 1);
  for (size_t i = 0; i < spans.size(); i++)
    for (size_t y = spans[i].first; y < spans[i].second; y++) s->row_segments[fill[y]++] = uint32_t(i);
}

}  // namespace jxh
#endif  // JXH_SPLINES_H_
