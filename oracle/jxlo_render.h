// ORACLE (test infrastructure): post-IDCT stages — Gaborish, edge-preserving filter (EPF0/1/2),
// XYB -> linear RGB, sRGB transfer function, float -> uint8 with blue-noise dither.
// Follows: reference lib/jxl/render_pipeline/stage_gaborish.cc:33-100, stage_epf.cc:47-494, lib/jxl/epf.h:19-22,
// stage_xyb.cc:80-92 + lib/jxl/dec_xyb-inl.h:38-86 + lib/jxl/dec_xyb.cc:144-250 + lib/jxl/opsin_params.cc:35-45,
// stage_from_linear.cc:42-54 + lib/jxl/cms/transfer_functions-inl.h:245-268, stage_write.cc:266-286,548-590.
// Image edges: every stage reads its input mirrored about the frame size (lib/jxl/image_ops.h:184-196,
// render_pipeline/low_memory_render_pipeline.cc:475-517); see DESIGN.md for why this equals the reference's
// row-streaming behaviour for these (reflection-symmetric) kernels.
#ifndef JXLO_RENDER_H_
#define JXLO_RENDER_H_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <vector>

#include "jxlo_headers.h"

namespace jxlo {

#include "dither.inc"

static inline int64_t Mirror(int64_t x, int64_t n) {
  while (x < 0 || x >= n) {
    if (x < 0) x = -x - 1;
    else x = 2 * n - 1 - x;
  }
  return x;
}

// A planar 3-channel float image with row stride `stride`; only [0,xs) x [0,ys) is meaningful.
struct Planes3 {
  size_t xs = 0, ys = 0, stride = 0;
  std::vector<float> p[3];
  void Alloc(size_t x, size_t y, size_t st) {
    xs = x; ys = y; stride = st;
    for (auto& v : p) v.assign(st * y, 0.0f);
  }
  float At(int c, int64_t x, int64_t y) const { return p[c][size_t(Mirror(y, ys)) * stride + size_t(Mirror(x, xs))]; }
};

// Chroma upsampling of one channel of a subsampled YCbCr frame, in place (render_pipeline/stage_chroma_upsampling.cc:29-111;
// dec_cache.cc:138-150: horizontally first, then vertically). The subsampled samples sit in the top-left part of the plane:
// ceil(xsize / 2) columns / ceil(ysize / 2) rows of them belong to the image, and the stage mirrors about that size
// (low_memory_render_pipeline.cc:348-355, 668-683); each sample becomes two: 3/4 of itself + 1/4 of the neighbour on that
// side, as one multiply and one fused multiply-add like the reference's Mul / MulAdd.
static inline void ChromaUpsample(std::vector<float>* plane, size_t stride, size_t xsize, size_t ysize, size_t rows, bool horizontal,
                                  bool vertical) {
  std::vector<float>& p = *plane;
  const size_t in_rows = vertical ? (ysize + 1) / 2 : ysize;
  if (horizontal) {
    const int64_t ws = int64_t((xsize + 1) / 2);
    std::vector<float> row(static_cast<size_t>(ws), 0.0f);
    for (size_t y = 0; y < in_rows; y++) {
      float* r = p.data() + y * stride;
      memcpy(row.data(), r, size_t(ws) * sizeof(float));
      for (int64_t x = 0; x < ws; x++) {
        const float cur = row[size_t(x)] * 0.75f;
        r[2 * x] = std::fma(0.25f, row[size_t(Mirror(x - 1, ws))], cur);
        r[2 * x + 1] = std::fma(0.25f, row[size_t(Mirror(x + 1, ws))], cur);
      }
    }
  }
  if (vertical) {
    const int64_t hh = int64_t(in_rows);
    const size_t w = std::min(stride, 2 * ((xsize + 1) / 2));
    std::vector<float> src(size_t(hh) * w);
    for (int64_t y = 0; y < hh; y++) memcpy(src.data() + size_t(y) * w, p.data() + size_t(y) * stride, w * sizeof(float));
    for (int64_t y = 0; y < hh && size_t(2 * y + 1) < rows; y++) {
      const float *top = src.data() + size_t(Mirror(y - 1, hh)) * w, *mid = src.data() + size_t(y) * w,
                  *bot = src.data() + size_t(Mirror(y + 1, hh)) * w;
      float *o0 = p.data() + size_t(2 * y) * stride, *o1 = o0 + stride;
      for (size_t x = 0; x < w; x++) {
        const float m = mid[x] * 0.75f;
        o0[x] = std::fma(top[x], 0.25f, m);
        o1[x] = std::fma(bot[x], 0.25f, m);
      }
    }
  }
}

static inline void Gaborish(const Planes3& in, const LoopFilter& lf, Planes3* out) {
  out->Alloc(in.xs, in.ys, in.stride);
  for (int c = 0; c < 3; c++) {
    float w0 = 1.0f, w1 = lf.gab_w[c][0], w2 = lf.gab_w[c][1];
    const float mul = 1.0f / (w0 + 4 * (w1 + w2));
    w0 *= mul; w1 *= mul; w2 *= mul;
#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < int64_t(in.ys); y++)
      for (int64_t x = 0; x < int64_t(in.xs); x++) {
        float m = in.At(c, x, y);
        float s1 = (in.At(c, x - 1, y) + in.At(c, x + 1, y)) + (in.At(c, x, y - 1) + in.At(c, x, y + 1));
        float s2 = (in.At(c, x - 1, y - 1) + in.At(c, x + 1, y - 1)) + (in.At(c, x - 1, y + 1) + in.At(c, x + 1, y + 1));
        out->p[c][size_t(y) * in.stride + size_t(x)] = s2 * w2 + (s1 * w1 + m * w0);
      }
  }
}

static const float kMinSigma = -3.90524291751269967465540850526868f;
static const float kInvSigmaNum = -1.1715728752538099024f;

// One EPF pass. stage: 0, 1 or 2. inv_sigma: per 8x8 block (xblocks wide), holds 1/sigma (negative).
static inline void EpfPass(int stage, const Planes3& in, const LoopFilter& lf, const std::vector<float>& inv_sigma,
                           size_t xblocks, Planes3* out) {
  out->Alloc(in.xs, in.ys, in.stride);
  const float pass_scale = stage == 0 ? lf.epf_pass0_sigma_scale : stage == 2 ? lf.epf_pass2_sigma_scale : 1.0f;
  const float sm = stage == 1 ? 1.65f : float(pass_scale * 1.65);
  const float bsm = sm * lf.epf_border_sad_mul;
  static const int kOff0[12][2] = {{-2, 0}, {-1, -1}, {-1, 0}, {-1, 1}, {0, -2}, {0, -1},
                                   {0, 1},  {0, 2},   {1, -1}, {1, 0},  {1, 1},  {2, 0}};  // {dy, dx}
  static const int kOff1[4][2] = {{-1, 0}, {0, -1}, {0, 1}, {1, 0}};
  static const int kPlus[5][2] = {{0, 0}, {-1, 0}, {0, -1}, {1, 0}, {0, 1}};
  const int noff = stage == 0 ? 12 : 4;
  const int(*offs)[2] = stage == 0 ? kOff0 : kOff1;
#pragma omp parallel for schedule(static)
  for (int64_t y = 0; y < int64_t(in.ys); y++)
    for (int64_t x = 0; x < int64_t(in.xs); x++) {
      const size_t o = size_t(y) * in.stride + size_t(x);
      const float is = inv_sigma[size_t(y / 8) * xblocks + size_t(x / 8)];
      if (is < kMinSigma) {
        for (int c = 0; c < 3; c++) out->p[c][o] = in.p[c][o];
        continue;
      }
      const bool border = (x % 8 == 0) || (x % 8 == 7) || (y % 8 == 0) || (y % 8 == 7);
      const float inv_sig = is * (border ? bsm : sm);
      float w = 1.0f;
      float acc[3] = {in.p[0][o], in.p[1][o], in.p[2][o]};
      for (int i = 0; i < noff; i++) {
        const int dy = offs[i][0], dx = offs[i][1];
        float sad = 0.0f;
        if (stage == 2) {
          sad = std::fabs(in.At(0, x + dx, y + dy) - in.p[0][o]) * lf.epf_channel_scale[0];
          sad = std::fabs(in.At(1, x + dx, y + dy) - in.p[1][o]) * lf.epf_channel_scale[1] + sad;
          sad = std::fabs(in.At(2, x + dx, y + dy) - in.p[2][o]) * lf.epf_channel_scale[2] + sad;
        } else {
          for (int c = 0; c < 3; c++) {
            float s = 0.0f;
            for (int k = 0; k < 5; k++) {
              const int py = kPlus[k][0], px = kPlus[k][1];
              s += std::fabs(in.At(c, x + px, y + py) - in.At(c, x + dx + px, y + dy + py));
            }
            sad = s * lf.epf_channel_scale[c] + sad;
          }
        }
        float weight = sad * inv_sig + 1.0f;
        if (weight < 0.0f) weight = 0.0f;
        w += weight;
        for (int c = 0; c < 3; c++) acc[c] = weight * in.At(c, x + dx, y + dy) + acc[c];
      }
      const float inv_w = 1.0f / w;
      for (int c = 0; c < 3; c++) out->p[c][o] = acc[c] * inv_w;
    }
}

struct OpsinParams {
  float inv[9];
  float bias[3], bias_cbrt[3];
};
static inline OpsinParams MakeOpsinParams(const ImageHeader& h) {
  OpsinParams o;
  for (int i = 0; i < 9; i++) o.inv[i] = h.inv_opsin[i] * (255.0f / h.intensity_target);
  for (int i = 0; i < 3; i++) {
    o.bias[i] = h.opsin_bias[i];
    o.bias_cbrt[i] = cbrtf(h.opsin_bias[i]);
  }
  return o;
}
// ---- upsampling by N = 2, 4, 8 (lib/jxl/render_pipeline/stage_upsampling.cc:49-282): every output pixel
// (N x + ox, N y + oy) is a 5x5 weighted sum of the input around (x, y) with kernel k = N oy + ox, clamped to the
// minimum / maximum of that 5x5 window. The N*N kernels come from the upper triangle of a symmetric weight matrix
// (the weights the image header codes, or the default ones: image_metadata.cc:87-214) by the four mirror symmetries.
#include "upsampling_weights.inc"
static inline void UpsamplingKernels(uint32_t N, const float* coded, float* kernel /* N*N*25 */) {
  const float* weights = coded ? coded : (N == 2 ? kUpsamplingWeights2 : (N == 4 ? kUpsamplingWeights4 : kUpsamplingWeights8));
  const size_t H = N / 2;
  for (size_t ky = 0; ky < H; ++ky)
    for (size_t kx = 0; kx < H; ++kx) {
      const size_t o0 = (ky * N + kx) * 25, o1 = (ky * N + (N - 1 - kx)) * 25, o2 = ((N - 1 - ky) * N + kx) * 25,
                   o3 = ((N - 1 - ky) * N + (N - 1 - kx)) * 25;
      for (size_t py = 0; py < 5; ++py)
        for (size_t px = 0; px < 5; ++px) {
          const size_t j = 5 * ky + py, i = 5 * kx + px, my = std::min(i, j), mx = std::max(i, j);
          const float w = weights[5 * H * my - my * (my - 1) / 2 + mx - my];
          kernel[o0 + py * 5 + px] = w;
          kernel[o1 + py * 5 + (4 - px)] = w;
          kernel[o2 + (4 - py) * 5 + px] = w;
          kernel[o3 + (4 - py) * 5 + (4 - px)] = w;
        }
    }
}
static inline void Upsample(const Planes3& in, uint32_t N, const float* coded_weights, size_t out_xs, size_t out_ys, Planes3* out,
                            int nplanes = 3) {
  std::vector<float> kernel(size_t(N) * N * 25);
  UpsamplingKernels(N, coded_weights, kernel.data());
  out->Alloc(out_xs, out_ys, out_xs);
  for (int c = 0; c < nplanes; c++)
    for (size_t y = 0; y < in.ys; y++)
      for (size_t x = 0; x < in.xs; x++) {
        float v[25], mn = 0, mx = 0;
        for (int iy = -2; iy <= 2; iy++)
          for (int ix = -2; ix <= 2; ix++) {
            const float t = in.At(c, int64_t(x) + ix, int64_t(y) + iy);
            v[5 * (iy + 2) + ix + 2] = t;
            if (iy == -2 && ix == -2) mn = mx = t;
            mn = std::min(mn, t);
            mx = std::max(mx, t);
          }
        for (uint32_t oy = 0; oy < N; oy++)
          for (uint32_t ox = 0; ox < N; ox++) {
            const size_t X = x * N + ox, Y = y * N + oy;
            if (X >= out_xs || Y >= out_ys) continue;
            const float* k = kernel.data() + size_t(N * oy + ox) * 25;
            // three accumulation chains, as the reference: acc0 takes taps 0,3,..,24, acc1 1,4,..,22, acc2 2,5,..,23
            float a0 = v[0] * k[0], a1 = v[1] * k[1], a2 = v[2] * k[2];
            for (int i = 3; i < 24; i += 3) {
              a0 = std::fma(v[i], k[i], a0);
              a1 = std::fma(v[i + 1], k[i + 1], a1);
              a2 = std::fma(v[i + 2], k[i + 2], a2);
            }
            a0 = std::fma(v[24], k[24], a0);
            float r = (a1 + a2) + a0;
            r = r < mn ? mn : (r > mx ? mx : r);
            out->p[c][Y * out_xs + X] = r;
          }
      }
}

// ---- noise synthesis (frame flag kNoise): lib/jxl/xorshift128plus-inl.h:30-96 (generator), dec_noise.cc:43-110 (one
// generator per 256x256 group of the image, seeded by the frame indices and the group origin, fills three planes one after
// the other, 16 floats in [1, 2) per step), render_pipeline/stage_noise.cc:262-310 (5x5 high-pass) and :64-260 (strength
// from the pixel's intensity through the 8-point LUT, added to X, Y, B), dec_cache.cc:216-219 (after upsampling).
struct Xorshift128Plus {
  uint64_t s0[8], s1[8];
  static uint64_t SplitMix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  explicit Xorshift128Plus(uint64_t seed) {
    s0[0] = SplitMix64(seed + 0x9E3779B97F4A7C15ull);
    s1[0] = SplitMix64(s0[0]);
    for (int i = 1; i < 8; i++) {
      s0[i] = SplitMix64(s1[i - 1]);
      s1[i] = SplitMix64(s0[i]);
    }
  }
  Xorshift128Plus(uint32_t seed1, uint32_t seed2, uint32_t seed3, uint32_t seed4) {
    s0[0] = SplitMix64(((uint64_t(seed1) << 32) + seed2) + 0x9E3779B97F4A7C15ull);
    s1[0] = SplitMix64(((uint64_t(seed3) << 32) + seed4) + 0x9E3779B97F4A7C15ull);
    for (int i = 1; i < 8; i++) {
      s0[i] = SplitMix64(s0[i - 1]);
      s1[i] = SplitMix64(s1[i - 1]);
    }
  }
  void Fill(uint64_t* bits) {
    for (int i = 0; i < 8; i++) {
      uint64_t a = s0[i];
      const uint64_t b = s1[i];
      bits[i] = a + b;
      s0[i] = b;
      a ^= a << 23;
      a ^= b ^ (a >> 18) ^ (b >> 5);
      s1[i] = a;
    }
  }
};

// Fills the three image-sized planes (stride xs) with the raw random values.
static inline void NoiseRandom(size_t xs, size_t ys, uint32_t visible_frame_index, uint32_t nonvisible_frame_index,
                               std::vector<float> out[3]) {
  for (int c = 0; c < 3; c++) out[c].assign(xs * ys, 0.0f);
  const size_t kGroup = 256;
  for (size_t y0 = 0; y0 < ys; y0 += kGroup)
    for (size_t x0 = 0; x0 < xs; x0 += kGroup) {
      Xorshift128Plus rng(visible_frame_index, nonvisible_frame_index, uint32_t(x0), uint32_t(y0));
      const size_t w = std::min(kGroup, xs - x0), h = std::min(kGroup, ys - y0);
      for (int c = 0; c < 3; c++)
        for (size_t y = 0; y < h; y++) {
          float* row = out[c].data() + (y0 + y) * xs + x0;
          uint64_t batch[8];
          auto put = [&](size_t x) {  // the 16 floats of `batch` at row[x ..], as far as the row goes
            for (size_t i = 0; i < 16 && x + i < w; i++) {
              const uint32_t bits = uint32_t(batch[i >> 1] >> (32 * (i & 1)));
              const uint32_t f = (bits >> 9) | 0x3F800000u;
              memcpy(&row[x + i], &f, 4);
            }
          };
          size_t x = 0;
          for (; x + 16 < w; x += 16) {
            rng.Fill(batch);
            put(x);
          }
          rng.Fill(batch);
          put(x);
        }
    }
}

static inline float NoiseStrength(const float* lut, float x) {
  float scaled = std::max(0.0f, x * 6.0f);
  float fl = std::floor(scaled), frac = scaled - fl;
  if (scaled >= 7.0f) {
    fl = 6.0f;
    frac = 1.0f;
  }
  const int i = int(fl);
  const float v = (lut[i + 1] - lut[i]) * frac + lut[i];
  return std::min(std::max(v, 0.0f), 1.0f);
}

// Adds the noise to the X, Y, B planes `img` (image size xs x ys inside).
static inline void AddNoise(Planes3* img, size_t xs, size_t ys, const float* lut, float ytox, float ytob, uint32_t visible_frame_index,
                            uint32_t nonvisible_frame_index) {
  std::vector<float> raw[3];
  NoiseRandom(xs, ys, visible_frame_index, nonvisible_frame_index, raw);
  auto at = [&](int c, int64_t x, int64_t y) { return raw[c][size_t(Mirror(y, int64_t(ys))) * xs + size_t(Mirror(x, int64_t(xs)))]; };
#pragma omp parallel for schedule(static)
  for (size_t y = 0; y < ys; y++)
    for (size_t x = 0; x < xs; x++) {
      float rnd[3];
      for (int c = 0; c < 3; c++) {
        float others = 0.0f;
        for (int dy = -2; dy <= 2; dy++)
          for (int dx = -2; dx <= 2; dx++)
            if (dx || dy) others += at(c, int64_t(x) + dx, int64_t(y) + dy);
        rnd[c] = (others * 0.16f + at(c, int64_t(x), int64_t(y)) * -3.84f) * 0.22f;
      }
      const size_t i = y * img->stride + x;
      const float vx = img->p[0][i], vy = img->p[1][i];
      const float str_g = NoiseStrength(lut, (vy - vx) * 0.5f), str_r = NoiseStrength(lut, (vy + vx) * 0.5f);
      const float red = str_r * (0.0078125f * rnd[0] + 0.9921875f * rnd[2]);
      const float green = str_g * (0.0078125f * rnd[1] + 0.9921875f * rnd[2]);
      const float rg = red + green;
      img->p[0][i] = (ytox * rg + (red - green)) + vx;
      img->p[1][i] = vy + rg;
      img->p[2][i] = ytob * rg + img->p[2][i];
    }
}

static inline void XybToRgb(const OpsinParams& op, float X, float Y, float B, float* r, float* g, float* b) {
  float gr = (Y + X) - op.bias_cbrt[0];
  float gg = (Y - X) - op.bias_cbrt[1];
  float gb = B - op.bias_cbrt[2];
  float mr = (gr * gr) * gr + op.bias[0];
  float mg = (gg * gg) * gg + op.bias[1];
  float mb = (gb * gb) * gb + op.bias[2];
  *r = op.inv[2] * mb + (op.inv[1] * mg + op.inv[0] * mr);
  *g = op.inv[5] * mb + (op.inv[4] * mg + op.inv[3] * mr);
  *b = op.inv[8] * mb + (op.inv[7] * mg + op.inv[6] * mr);
}

static inline float LinearToSrgb(float v) {
  float a = std::fabs(v);
  float r;
  if (a > 0.0031308f) {
    float s = std::sqrt(a);
    float yp = 7.352629620e-01f * s + 1.474205315e+00f;
    yp = yp * s + 3.903842876e-01f;
    yp = yp * s + 5.287254571e-03f;
    yp = yp * s + -5.135152395e-04f;
    float yq = 2.424867759e-02f * s + 9.258482155e-01f;
    yq = yq * s + 1.340816930e+00f;
    yq = yq * s + 3.036675394e-01f;
    yq = yq * s + 1.004519624e-02f;
    r = yp / yq;
  } else {
    r = a * 12.92f;
  }
  return std::copysign(r, v);
}

// Float sample (nominal range [0,1]) -> uint8 with dither; c = interleaved channel index.
static inline uint8_t ToU8(float v, size_t x, size_t y, size_t c) {
  v = v * 255.0f;
  v += kDither32[(y + c * 13) % 32][(x + c * 23) % 32];
  if (!(v >= 0.0f)) v = 0.0f;  // also maps NaN to 0 like Clamp(Zero, v, mul)
  if (v > 255.0f) v = 255.0f;
  return uint8_t(std::nearbyint(v));
}

}  // namespace jxlo
#endif  // JXLO_RENDER_H_
