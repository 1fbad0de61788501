// libjxl_amd — fused loop filters + colour for gfx950 (MI355X): Gaborish -> EPF0/1/2 -> XYB->RGB -> sRGB -> RGB8 in one
// pass over an LDS-resident tile. Replaces the reference's render-pipeline stage list for VarDCT frames
// (lib/jxl/render_pipeline/stage_gaborish.cc:56-100, stage_epf.cc:82-494, stage_xyb.cc:80-92 + dec_xyb-inl.h:38-86,
// stage_from_linear.cc:114-144 + cms/transfer_functions-inl.h:245-268, stage_write.cc:266-286,548-590) and the edge
// mirroring of low_memory_render_pipeline.cc:475-517.
//
// One workgroup owns a 64x16 output tile. The inverse-transform output (3 f32 planes) is read ONCE from HBM for the
// tile plus a halo of H pixels (H = sum of the enabled stages' radii, coordinates mirrored about the frame size), every
// stage then runs LDS -> LDS on a region that shrinks by its radius, and the last region (the tile itself) is colour
// converted and written as packed RGB8. HBM traffic per pixel: 12 B * halo overhead in, 3 B out, versus one full
// read + write of the 3 planes per stage in the unfused kernels.
//
// Computing a stage at a halo position that lies outside the frame from mirrored inputs gives exactly the mirrored
// output of that stage (Gaborish and EPF are reflection-symmetric once sigma and the block-border flag are also taken
// at the mirrored position), which is what the reference's row pipeline feeds the next stage.
#ifndef JXL_HIP_FILTER_FUSED_H_
#define JXL_HIP_FILTER_FUSED_H_

#include <type_traits>

#include "jxl_hip_kernels.h"

namespace jxlhip {

struct FusedFilterParams {
  FilterParams f;          // geometry, sigma, Gaborish weights, channel scales, colour constants, rgb; f.in = IDCT output
  float sm[3], bsm[3];     // per EPF stage (0, 1, 2): sigma multipliers for block-interior / block-border pixels
  float* filtered;         // optional (may be NULL): filtered XYB planes for tests (same geometry as f.in)
  uint32_t debug;          // measurement aid (results invalid when set): bit 0 skip Gaborish, 1 skip EPF, 2 skip colour maths
};

constexpr int kFusedTW = 64, kFusedTH = 16;  // 37 KB of LDS with a 3-pixel halo: two tiles fit beside an entropy workgroup
constexpr int kFusedThreads = 512;  // 8 waves per tile: two resident tiles give every SIMD four waves to hide LDS latency
constexpr int kFusedSigW = kFusedTW / 8 + 3, kFusedSigH = kFusedTH / 8 + 3;  // sigma blocks around a tile
__host__ __device__ constexpr int FusedHalo(bool gab, int epf) {
  return (gab ? 1 : 0) + (epf >= 3 ? 3 : 0) + (epf >= 1 ? 2 : 0) + (epf >= 2 ? 1 : 0);
}
__host__ __device__ constexpr size_t FusedLdsBytes(bool gab, int epf) {
  return (size_t(2) * 3 * (kFusedTW + 2 * FusedHalo(gab, epf)) * (kFusedTH + 2 * FusedHalo(gab, epf)) + kFusedSigW * kFusedSigH) *
         sizeof(float);
}

// One EPF stage (STAGE 0: 12 neighbours within distance 2, plus-shaped 5-point SADs; STAGE 1: 4 neighbours, plus-shaped
// SADs; STAGE 2: 4 neighbours, single-point SADs) for the pixel at LDS offset `o`; S = row stride, PL = plane stride.
template <int STAGE, int S, int PL>
__device__ __forceinline__ void EpfPixel(const FusedFilterParams& P, const float* src, int o, float inv_sigma_raw, bool border,
                                         float* out0, float* out1, float* out2) {
  const float c0 = src[o], c1 = src[PL + o], c2 = src[2 * PL + o];
  if (inv_sigma_raw < -3.90524291751269967465540850526868f) {  // sigma too small: pixel unchanged (stage_epf.cc kMinSigma)
    *out0 = c0;
    *out1 = c1;
    *out2 = c2;
    return;
  }
  const float inv_sig = inv_sigma_raw * (border ? P.bsm[STAGE] : P.sm[STAGE]);
  constexpr int NOFF = STAGE == 0 ? 12 : 4;
  constexpr int off0[12][2] = {{-2, 0}, {-1, -1}, {-1, 0}, {-1, 1}, {0, -2}, {0, -1}, {0, 1}, {0, 2}, {1, -1}, {1, 0}, {1, 1}, {2, 0}};
  constexpr int off1[4][2] = {{-1, 0}, {0, -1}, {0, 1}, {1, 0}};
  constexpr int plus[5][2] = {{0, 0}, {-1, 0}, {0, -1}, {1, 0}, {0, 1}};
  float w = 1.0f, a0 = c0, a1 = c1, a2 = c2;
#pragma unroll
  for (int i = 0; i < NOFF; i++) {
    const int dy = STAGE == 0 ? off0[i][0] : off1[i][0], dx = STAGE == 0 ? off0[i][1] : off1[i][1];
    const int n = o + dy * S + dx;
    float sad = 0.0f;
    if (STAGE == 2) {
      sad = fabsf(src[n] - c0) * P.f.ch_scale[0];
      sad = fabsf(src[PL + n] - c1) * P.f.ch_scale[1] + sad;
      sad = fabsf(src[2 * PL + n] - c2) * P.f.ch_scale[2] + sad;
    } else {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float* p = src + c * PL;
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 5; k++) {
          const int d = plus[k][0] * S + plus[k][1];
          s += fabsf(p[o + d] - p[n + d]);
        }
        sad = s * P.f.ch_scale[c] + sad;
      }
    }
    float weight = sad * inv_sig + 1.0f;
    weight = weight < 0.0f ? 0.0f : weight;
    w += weight;
    a0 = weight * src[n] + a0;
    a1 = weight * src[PL + n] + a1;
    a2 = weight * src[2 * PL + n] + a2;
  }
  const float inv_w = 1.0f / w;
  *out0 = a0 * inv_w;
  *out1 = a1 * inv_w;
  *out2 = a2 * inv_w;
}

// EPF stage 1 over the region that extends HO pixels around the tile, with shared absolute differences: the plus-shaped
// SAD between the neighbourhoods of a pixel and of its right / lower neighbour is a 5-point sum of the per-position maps
//   D_h(q) = sum_c scale_c * |p_c(q) - p_c(q + (0,1))|,   D_v(q) = sum_c scale_c * |p_c(q) - p_c(q + (1,0))|
// (left / upper neighbour: the same maps one column / row earlier), so every absolute difference is computed once per
// workgroup instead of up to ten times. The maps live in the two lower planes of `dst`; the stage's outputs are kept in
// registers until every thread is done reading the maps, then stored over them.
template <int S, int PL, int HO, int H>
__device__ __forceinline__ void Epf1Stage(const FusedFilterParams& P, const float* src, float* dst, const float* l_sig, int x0,
                                          int y0, int xs, int ys, int tid) {
  const int sbx = (x0 >> 3) - 1, sby = (y0 >> 3) - 1;
  constexpr int TW = kFusedTW, TH = kFusedTH;
  {
    constexpr int w = TW + 2 * HO + 3, n = w * (TH + 2 * HO + 3), NITD = (n + kFusedThreads - 1) / kFusedThreads;
#pragma unroll 4
    for (int it = 0; it < NITD; it++) {
      const int i = tid + it * kFusedThreads;
      if (i >= n) break;
      const int ry = i / w, rx = i - ry * w;
      const int q = (ry + H - HO - 2) * S + rx + H - HO - 2;
      float dh = 0.0f, dv = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float v = src[c * PL + q];
        dh = fabsf(v - src[c * PL + q + 1]) * P.f.ch_scale[c] + dh;
        dv = fabsf(v - src[c * PL + q + S]) * P.f.ch_scale[c] + dv;
      }
      dst[q] = dh;
      dst[PL + q] = dv;
    }
  }
  __syncthreads();
  constexpr int w = TW + 2 * HO, n = w * (TH + 2 * HO), NIT = (n + kFusedThreads - 1) / kFusedThreads;
  float out[NIT][3];
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int i = tid + it * kFusedThreads;
    if (i < n) {
      const int ry = i / w, rx = i - ry * w;
      const int o = (ry + H - HO) * S + rx + H - HO;
      const int mx = MirrorI(x0 + rx - HO, xs), my = MirrorI(y0 + ry - HO, ys);
      const float is = l_sig[((my >> 3) - sby) * kFusedSigW + (mx >> 3) - sbx];
      const float c0 = src[o], c1 = src[PL + o], c2 = src[2 * PL + o];
      if (is < -3.90524291751269967465540850526868f) {
        out[it][0] = c0;
        out[it][1] = c1;
        out[it][2] = c2;
      } else {
        const bool border = ((mx & 7) == 0) || ((mx & 7) == 7) || ((my & 7) == 0) || ((my & 7) == 7);
        const float inv_sig = is * (border ? P.bsm[1] : P.sm[1]);
        const float* dh = dst;
        const float* dv = dst + PL;
        // neighbours in the reference's order: up, left, right, down
        const int nb[4] = {o - S, o - 1, o + 1, o + S};
        const float sad[4] = {
            (dv[o - S] + dv[o - 2 * S]) + (dv[o - S - 1] + dv[o]) + dv[o - S + 1],
            (dh[o - 1] + dh[o - S - 1]) + (dh[o - 2] + dh[o + S - 1]) + dh[o],
            (dh[o] + dh[o - S]) + (dh[o - 1] + dh[o + S]) + dh[o + 1],
            (dv[o] + dv[o - S]) + (dv[o - 1] + dv[o + S]) + dv[o + 1]};
        float wsum = 1.0f, a0 = c0, a1 = c1, a2 = c2;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float weight = sad[j] * inv_sig + 1.0f;
          weight = weight < 0.0f ? 0.0f : weight;
          wsum += weight;
          a0 = weight * src[nb[j]] + a0;
          a1 = weight * src[PL + nb[j]] + a1;
          a2 = weight * src[2 * PL + nb[j]] + a2;
        }
        const float inv_w = __builtin_amdgcn_rcpf(wsum);  // 1 ulp; wsum >= 1
        out[it][0] = a0 * inv_w;
        out[it][1] = a1 * inv_w;
        out[it][2] = a2 * inv_w;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int i = tid + it * kFusedThreads;
    if (i < n) {
      const int ry = i / w, rx = i - ry * w;
      const int o = (ry + H - HO) * S + rx + H - HO;
      dst[o] = out[it][0];
      dst[PL + o] = out[it][1];
      dst[2 * PL + o] = out[it][2];
    }
  }
  __syncthreads();
}

// Upsampling by N = 2, 4, 8 + colour (stage_upsampling.cc:49-282, then the same XYB -> RGB8 conversion as the fused kernel):
// one thread per pixel of the coded frame produces its N x N image pixels. Every image pixel is a 5x5 weighted sum of the
// filtered XYB frame around the source pixel (kernel N * oy + ox, mirrored edges), clamped to the window's min / max.
struct UpsampleParams {
  FilterParams f;       // f.in = filtered XYB planes of the frame, f.rgb = image-sized output
  const float* kernel;  // [N * N][25]
  uint32_t n, oxs, oys;
  PixelOut po;          // po.dst != NULL: the image in this format instead of RGB8 in f.rgb
  float* xyb_out;       // != NULL: the upsampled X, Y, B as planes [3][oys][oxp] instead of pixels (the frame's noise is added
  uint32_t oxp;         // to them at the image's resolution before the colour stage: dec_cache.cc:206-216)
};

// XYB -> linear RGB -> (sRGB) for one pixel (stage_xyb.cc:80-92 + dec_xyb-inl.h:38-86, stage_from_linear.cc:114-144).
__device__ __forceinline__ void XybToRgb(const FilterParams& f, float X, float Y, float Bc, float* r, float* g, float* b) {
  if (f.linear_output >= 2) {  // a frame of an image that is not XYB encoded (wave-uniform)
    if (f.linear_output == 2) {  // kYCbCr (stage_ycbcr.cc:41-60): full-range BT.601 of JFIF; channels Cb, Y, Cr
      const float y = Y + 128.0f / 255;
      *r = fmaf(1.402f, Bc, y);
      *g = fmaf(-0.299f * 1.402f / 0.587f, Bc, fmaf(-0.114f * 1.772f / 0.587f, X, y));
      *b = fmaf(1.772f, X, y);
    } else {  // kNone: the channels are the samples
      *r = X;
      *g = Y;
      *b = Bc;
    }
    return;
  }
  const float gr = (Y + X) - f.opsin_bias_cbrt[0], gg = (Y - X) - f.opsin_bias_cbrt[1], gb = Bc - f.opsin_bias_cbrt[2];
  const float mr = (gr * gr) * gr + f.opsin_bias[0], mg = (gg * gg) * gg + f.opsin_bias[1], mb = (gb * gb) * gb + f.opsin_bias[2];
  *r = f.opsin_inv[2] * mb + (f.opsin_inv[1] * mg + f.opsin_inv[0] * mr);
  *g = f.opsin_inv[5] * mb + (f.opsin_inv[4] * mg + f.opsin_inv[3] * mr);
  *b = f.opsin_inv[8] * mb + (f.opsin_inv[7] * mg + f.opsin_inv[6] * mr);
  if (!f.linear_output) {
    *r = LinearToSrgb(*r);
    *g = LinearToSrgb(*g);
    *b = LinearToSrgb(*b);
  }
}

// Colour conversion + write for the output formats the filter kernels do not produce themselves (anything but RGB8,
// and RGB f32 on the row-streaming kernel): the filter kernel leaves the filtered XYB planes, this kernel makes the pixels.
struct ColorOutParams {
  FilterParams f;  // f.in = filtered XYB planes; y_begin / y_end = rows to produce
  PixelOut po;
};
__global__ __launch_bounds__(256) void k_color_out(ColorOutParams P) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = int(P.f.y_begin) + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= int(P.f.xs) || y >= int(P.f.y_end)) return;
  const size_t plane = size_t(P.f.xp) * P.f.yp, gi = size_t(y) * P.f.xp + x;
  float r, g, b;
  XybToRgb(P.f, P.f.in[gi], P.f.in[plane + gi], P.f.in[2 * plane + gi], &r, &g, &b);
  StorePixel(P.po, x, y, r, g, b);
}

__global__ __launch_bounds__(256) void k_upsample_color(UpsampleParams P) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int xs = int(P.f.xs), ys = int(P.f.ys);
  if (x >= xs || y >= ys) return;
  const size_t plane = size_t(P.f.xp) * P.f.yp;
  float v[3][25], mn[3], mx[3];
#pragma unroll
  for (int iy = 0; iy < 5; iy++) {
    const size_t row = size_t(MirrorI(y + iy - 2, ys)) * P.f.xp;
#pragma unroll
    for (int ix = 0; ix < 5; ix++) {
      const size_t g = row + MirrorI(x + ix - 2, xs);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float t = P.f.in[c * plane + g];
        v[c][iy * 5 + ix] = t;
        mn[c] = (iy | ix) == 0 ? t : fminf(mn[c], t);
        mx[c] = (iy | ix) == 0 ? t : fmaxf(mx[c], t);
      }
    }
  }
  const int N = int(P.n);
  for (int oy = 0; oy < N; oy++) {
    const int Y = y * N + oy;
    if (Y >= int(P.oys)) break;
    for (int ox = 0; ox < N; ox++) {
      const int X = x * N + ox;
      if (X >= int(P.oxs)) break;
      const float* k = P.kernel + (N * oy + ox) * 25;
      float ch[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        // three accumulation chains in the reference's order: taps 0,3,..,24 / 1,4,..,22 / 2,5,..,23
        float a0 = v[c][0] * k[0], a1 = v[c][1] * k[1], a2 = v[c][2] * k[2];
#pragma unroll
        for (int i = 3; i < 24; i += 3) {
          a0 = fmaf(v[c][i], k[i], a0);
          a1 = fmaf(v[c][i + 1], k[i + 1], a1);
          a2 = fmaf(v[c][i + 2], k[i + 2], a2);
        }
        a0 = fmaf(v[c][24], k[24], a0);
        const float r = (a1 + a2) + a0;
        ch[c] = r < mn[c] ? mn[c] : (r > mx[c] ? mx[c] : r);
      }
      const float Xc = ch[0], Yc = ch[1], Bc = ch[2];
      if (P.xyb_out) {
        const size_t oplane = size_t(P.oxp) * P.oys, gi = size_t(Y) * P.oxp + X;
        P.xyb_out[gi] = Xc;
        P.xyb_out[oplane + gi] = Yc;
        P.xyb_out[2 * oplane + gi] = Bc;
        continue;
      }
      float r, g, b;
      XybToRgb(P.f, Xc, Yc, Bc, &r, &g, &b);
      if (P.po.dst) {
        StorePixel(P.po, X, Y, r, g, b);
        continue;
      }
      uint8_t* dst = P.f.rgb + (size_t(Y) * P.oxs + X) * 3;
      dst[0] = ToU8(r, X, Y, 0);
      dst[1] = ToU8(g, X, Y, 1);
      dst[2] = ToU8(b, X, Y, 2);
    }
  }
}

// The upsampling stage on ONE plane (an extra channel with an upsampling factor: dec_cache.cc:172-190, 203-212): the same
// 5x5 sums, accumulation order and clamp as k_upsample_color, one thread per sample of the coded plane.
struct UpsamplePlaneParams {
  const float* in;      // ys rows of xs samples, dense
  float* out;           // oys rows of oxs samples, dense
  const float* kernel;  // [N * N][25]
  uint32_t xs, ys, n, oxs, oys;
};
__global__ __launch_bounds__(256) void k_upsample_plane(UpsamplePlaneParams P) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int xs = int(P.xs), ys = int(P.ys);
  if (x >= xs || y >= ys) return;
  float v[25], mn = 0.0f, mx = 0.0f;
#pragma unroll
  for (int iy = 0; iy < 5; iy++) {
    const size_t row = size_t(MirrorI(y + iy - 2, ys)) * P.xs;
#pragma unroll
    for (int ix = 0; ix < 5; ix++) {
      const float t = P.in[row + MirrorI(x + ix - 2, xs)];
      v[iy * 5 + ix] = t;
      mn = (iy | ix) == 0 ? t : fminf(mn, t);
      mx = (iy | ix) == 0 ? t : fmaxf(mx, t);
    }
  }
  const int N = int(P.n);
  for (int oy = 0; oy < N; oy++) {
    const int Y = y * N + oy;
    if (Y >= int(P.oys)) break;
    for (int ox = 0; ox < N; ox++) {
      const int X = x * N + ox;
      if (X >= int(P.oxs)) break;
      const float* k = P.kernel + (N * oy + ox) * 25;
      float a0 = v[0] * k[0], a1 = v[1] * k[1], a2 = v[2] * k[2];
#pragma unroll
      for (int i = 3; i < 24; i += 3) {
        a0 = fmaf(v[i], k[i], a0);
        a1 = fmaf(v[i + 1], k[i + 1], a1);
        a2 = fmaf(v[i + 2], k[i + 2], a2);
      }
      a0 = fmaf(v[24], k[24], a0);
      const float r = (a1 + a2) + a0;
      P.out[size_t(Y) * P.oxs + X] = r < mn ? mn : (r > mx ? mx : r);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-streaming form of Gaborish -> EPF1 -> colour (the d1.0 configuration): no LDS at all, so it runs beside the
// LDS-hungry entropy workgroups instead of queueing for their LDS.
//
// A wave owns 64 adjacent columns (lane l <-> column x0 - 3 + l, the middle 58 produce output) and walks down a strip of
// rows. Everything vertical lives in the lane's registers as short sliding windows (3 input rows, 5 Gaborish rows, 3 rows
// of each absolute-difference map); everything horizontal comes from the neighbouring lanes through DPP wave shifts.
// The arithmetic is that of k_filter_fused<true, 1> expression for expression (same association), including the shared
// plus-shaped sums: with
//   D_h(q) = sum_c scale_c * |p_c(q) - p_c(q + (0,1))|,   D_v(q) = sum_c scale_c * |p_c(q) - p_c(q + (1,0))|,
//   P_h(q) = (D_h(q) + D_h(q - (1,0))) + (D_h(q - (0,1)) + D_h(q + (1,0))) + D_h(q + (0,1))   (P_v alike from D_v)
// the four SADs of stage_epf.cc:225-367 for pixel o are P_v(o - (1,0)), P_h(o - (0,1)), P_h(o), P_v(o): every P is
// used by two pixels and computed once.
constexpr int kRowsHalo = 3;                          // Gaborish 1 + EPF1 2
constexpr int kRowsLanes = 64 - 2 * kRowsHalo;        // output columns per wave
constexpr int kRowsStrip = 64;                        // output rows per wave
constexpr int kRowsWaves = 4;                         // waves (adjacent column groups) per workgroup

__device__ __forceinline__ float FromLeft(float v) {  // the value held by the lane of column x - 1 (wave_shr:1)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float FromRight(float v) {  // the value held by the lane of column x + 1 (wave_shl:1)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}
// acc + FromLeft(v) * w / acc + FromRight(v) * w as ONE instruction. The compiler folds a neighbour-lane read into an add or
// a multiply but not into a multiply-add (it picks the three-address form first, which has no DPP encoding, and only later
// turns it into v_fmac): a v_mov_dpp + v_fma pair otherwise. `v` must not have been written by the two instructions before
// (a DPP source needs two wait states behind a vector write, and inline assembly is not checked for it): here it is always a
// value of an earlier step's ring.
__device__ __forceinline__ float FmacFromLeft(float acc, float v, float w) {
  asm("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(v), "v"(w));
  return acc;
}
__device__ __forceinline__ float FmacFromRight(float acc, float v, float w) {
  asm("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(v), "v"(w));
  return acc;
}

__global__ __launch_bounds__(64 * kRowsWaves) void k_filter_rows(const FusedFilterParams* params) {
  FusedFilterParams P;
  LoadParams(P, params + blockIdx.z);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xs = int(P.f.xs), ys = int(P.f.ys);
  const int xw0 = (int(blockIdx.x) * kRowsWaves + wave) * kRowsLanes;  // first output column of the wave
  const int y0 = int(P.f.y_begin) + int(blockIdx.y) * kRowsStrip;
  const int y1 = y0 + kRowsStrip < int(P.f.y_end) ? y0 + kRowsStrip : int(P.f.y_end);
  if (xw0 >= xs || y0 >= y1) return;  // the grid covers the largest frame (band) of the launch
  const int x = xw0 - kRowsHalo + lane;
  const int mx = MirrorI(x, xs);
  const bool emit = lane >= kRowsHalo && lane < 64 - kRowsHalo && x < xs;
  const bool xborder = ((mx & 7) == 0) || ((mx & 7) == 7);
  const size_t gplane = size_t(P.f.xp) * P.f.yp;
  typedef const float __attribute__((address_space(1)))* GF32;  // global (not generic) accesses: no LDS-counter coupling
  typedef float __attribute__((address_space(1)))* GF32W;
  typedef uint8_t __attribute__((address_space(1)))* GU8W;
  const GF32 in = (GF32)(uintptr_t)P.f.in + mx;
  const GF32 sig = (GF32)(uintptr_t)P.f.inv_sigma + (mx >> 3);
  const GF32 dither = (GF32)(uintptr_t)c_dither;
  const GF32W filtered = (GF32W)(uintptr_t)P.filtered;
  const GU8W rgb = (GU8W)(uintptr_t)P.f.rgb;
  const int dcol[3] = {x & 31, (x + 23) & 31, (x + 46) & 31};
  // sliding windows; index 0 = oldest row
  float p[3][3], h1[3][3];  // input rows and their horizontal pair sums p(x - 1) + p(x + 1)
  float g[3][5];            // Gaborish rows yg - 4 .. yg
  float dh[3], dv[3];       // D_h rows yg - 3 .. yg - 1 (the newest, yg, enters at the next step), D_v rows yg - 3 .. yg - 1
  float pv_prev = 0.0f;     // P_v of the previous output row
#pragma unroll
  for (int c = 0; c < 3; c++) {
#pragma unroll
    for (int k = 0; k < 3; k++) p[c][k] = h1[c][k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 5; k++) g[c][k] = 0.0f;
  }
  dh[0] = dh[1] = dh[2] = dv[0] = dv[1] = dv[2] = 0.0f;
  float dh_new = 0.0f;
  const int steps = y1 - y0 + 2 * kRowsHalo;
  // Everything a step reads from memory (its input row, the sigma and the dither values of its output row) is loaded
  // one step ahead, so the only wait for memory is at the top of a step, for loads a whole step old.
  float nx[3], nis = 0.0f, ndi[3] = {0.0f, 0.0f, 0.0f};
  {
    const size_t row = size_t(MirrorI(y0 - kRowsHalo, ys)) * P.f.xp;
    nx[0] = in[row];
    nx[1] = in[gplane + row];
    nx[2] = in[2 * gplane + row];
  }
  for (int j = 0; j < steps; j++) {
    const int yi = y0 - kRowsHalo + j;  // input row of this step
    const float cur[3] = {nx[0], nx[1], nx[2]};
    const float is = nis;
    const float di[3] = {ndi[0], ndi[1], ndi[2]};
    const int r = yi - kRowsHalo;  // output row of this step (= yg - 2 = y0 - 6 + j), valid from step 6
    const int my = MirrorI(r, ys);
    if (j + 1 < steps) {
      const size_t row = size_t(MirrorI(yi + 1, ys)) * P.f.xp;
      nx[0] = in[row];
      nx[1] = in[gplane + row];
      nx[2] = in[2 * gplane + row];
      if (j + 1 >= 2 * kRowsHalo) {  // r + 1 is an output row (inside the frame)
        nis = sig[size_t((r + 1) >> 3) * P.f.xb];
        ndi[0] = dither[((r + 1) & 31) * 32 + dcol[0]];
        ndi[1] = dither[((r + 1 + 13) & 31) * 32 + dcol[1]];
        ndi[2] = dither[((r + 1 + 26) & 31) * 32 + dcol[2]];
      }
    }
    // ---- input window
#pragma unroll
    for (int c = 0; c < 3; c++) {
      p[c][0] = p[c][1];
      p[c][1] = p[c][2];
      p[c][2] = cur[c];
      h1[c][0] = h1[c][1];
      h1[c][1] = h1[c][2];
      h1[c][2] = FromLeft(cur[c]) + FromRight(cur[c]);
    }
    // ---- Gaborish row yg = yi - 1 (stage_gaborish.cc:56-100)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float m = p[c][1];
      const float s1 = h1[c][1] + (p[c][0] + p[c][2]);
      const float s2 = h1[c][0] + h1[c][2];
      const float v = s2 * P.f.gab_w[c * 3 + 2] + (s1 * P.f.gab_w[c * 3 + 1] + m * P.f.gab_w[c * 3]);
#pragma unroll
      for (int k = 0; k < 4; k++) g[c][k] = g[c][k + 1];
      g[c][4] = v;
    }
    // ---- difference maps: D_h(yg - 1) enters the window (computed last step as dh_new), D_v(yg - 1) = |G(yg - 1) - G(yg)|
    dh[0] = dh[1];
    dh[1] = dh[2];
    dh[2] = dh_new;
    dv[0] = dv[1];
    dv[1] = dv[2];
    {
      float a = 0.0f, b = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        a = fabsf(g[c][4] - FromRight(g[c][4])) * P.f.ch_scale[c] + a;
        b = fabsf(g[c][3] - g[c][4]) * P.f.ch_scale[c] + b;
      }
      dh_new = a;  // D_h(yg)
      dv[2] = b;   // D_v(yg - 1)
    }
    // ---- plus-shaped sums of output row r = yg - 2: window index 1 is row r for dh / dv, index 2 for g
    const float pv = (dv[1] + dv[0]) + (FromLeft(dv[1]) + dv[2]) + FromRight(dv[1]);
    const float ph = (dh[1] + dh[0]) + (FromLeft(dh[1]) + dh[2]) + FromRight(dh[1]);
    const float ph_left = FromLeft(ph);
    if (j >= 2 * kRowsHalo) {
      float o0 = g[0][2], o1 = g[1][2], o2 = g[2][2];
      float l0 = FromLeft(o0), l1 = FromLeft(o1), l2 = FromLeft(o2);
      float r0 = FromRight(o0), r1 = FromRight(o1), r2 = FromRight(o2);
      if (!(is < -3.90524291751269967465540850526868f)) {  // else sigma too small: pixel unchanged (stage_epf.cc kMinSigma)
        const bool border = xborder || ((my & 7) == 0) || ((my & 7) == 7);
        const float inv_sig = is * (border ? P.bsm[1] : P.sm[1]);
        // neighbours in the reference's order: up, left, right, down
        const float sad[4] = {pv_prev, ph_left, ph, pv};
        const float n0[4] = {g[0][1], l0, r0, g[0][3]};
        const float n1[4] = {g[1][1], l1, r1, g[1][3]};
        const float n2[4] = {g[2][1], l2, r2, g[2][3]};
        float wsum = 1.0f, a0 = o0, a1 = o1, a2 = o2;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          float weight = sad[k] * inv_sig + 1.0f;
          weight = weight < 0.0f ? 0.0f : weight;
          wsum += weight;
          a0 = weight * n0[k] + a0;
          a1 = weight * n1[k] + a1;
          a2 = weight * n2[k] + a2;
        }
        const float inv_w = __builtin_amdgcn_rcpf(wsum);  // 1 ulp; wsum >= 1
        o0 = a0 * inv_w;
        o1 = a1 * inv_w;
        o2 = a2 * inv_w;
      }
      if (emit) {
        if (filtered) {
          const size_t gi = size_t(r) * P.f.xp + x;
          filtered[gi] = o0;
          filtered[gplane + gi] = o1;
          filtered[2 * gplane + gi] = o2;
        }
        if (rgb) {
          const float X = o0, Y = o1, Bc = o2;
          const float gr = (Y + X) - P.f.opsin_bias_cbrt[0], gg = (Y - X) - P.f.opsin_bias_cbrt[1], gb = Bc - P.f.opsin_bias_cbrt[2];
          const float mr = (gr * gr) * gr + P.f.opsin_bias[0], mg = (gg * gg) * gg + P.f.opsin_bias[1], mb = (gb * gb) * gb + P.f.opsin_bias[2];
          float cr = P.f.opsin_inv[2] * mb + (P.f.opsin_inv[1] * mg + P.f.opsin_inv[0] * mr);
          float cg = P.f.opsin_inv[5] * mb + (P.f.opsin_inv[4] * mg + P.f.opsin_inv[3] * mr);
          float cb = P.f.opsin_inv[8] * mb + (P.f.opsin_inv[7] * mg + P.f.opsin_inv[6] * mr);
          if (!P.f.linear_output) {
            cr = LinearToSrgb(cr);
            cg = LinearToSrgb(cg);
            cb = LinearToSrgb(cb);
          }
          const GU8W dst = rgb + (size_t(r) * xs + x) * 3;
          dst[0] = ToU8D(cr, di[0]);
          dst[1] = ToU8D(cg, di[1]);
          dst[2] = ToU8D(cb, di[2]);
        }
      }
    }
    pv_prev = pv;
  }
}

// The same kernel with TWO adjacent columns per lane, held as 2-vectors so that most of the arithmetic compiles to packed
// f32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two pixels per issue slot; 25 % fewer vector instructions
// per pixel). A wave owns 128 columns (lane l <-> columns x0 - 4 + 2l, + 1; the middle 120 produce output); the inner
// horizontal neighbour of a pixel is the other element of the pair, the outer one comes from the adjacent lane.
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kRows2Cols = 120;  // output columns per wave: 128 - 2 * 4 (left halo rounded up to a whole pair)
__device__ __forceinline__ f2 LeftOf(f2 v) { return f2{FromLeft(v.y), v.x}; }     // the columns x - 1 of the pair's columns
__device__ __forceinline__ f2 RightOf(f2 v) { return f2{v.y, FromRight(v.x)}; }   // the columns x + 1
__device__ __forceinline__ f2 Abs2(f2 v) { return __builtin_elementwise_abs(v); }
// LinearToSrgb for two values: the same operations per element, the polynomials as packed FMAs
__device__ __forceinline__ f2 LinearToSrgb2(f2 v) {
  const f2 a = Abs2(v);
  const f2 s = f2{__builtin_amdgcn_sqrtf(a.x), __builtin_amdgcn_sqrtf(a.y)};
  f2 yp = 7.352629620e-01f * s + 1.474205315e+00f;
  yp = yp * s + 3.903842876e-01f;
  yp = yp * s + 5.287254571e-03f;
  yp = yp * s + -5.135152395e-04f;
  f2 yq = 2.424867759e-02f * s + 9.258482155e-01f;
  yq = yq * s + 1.340816930e+00f;
  yq = yq * s + 3.036675394e-01f;
  yq = yq * s + 1.004519624e-02f;
  const f2 hi = yp * f2{__builtin_amdgcn_rcpf(yq.x), __builtin_amdgcn_rcpf(yq.y)};
  const f2 lo = a * 12.92f;
  return f2{copysignf(a.x > 0.0031308f ? hi.x : lo.x, v.x), copysignf(a.y > 0.0031308f ? hi.y : lo.y, v.y)};
}

// The sRGB transfer function for samples that are about to become 8-bit integers: the curve itself,
// 1.055 * v^(1/2.4) - 0.055 through the hardware's base-2 logarithm and exponential (two transcendental and four plain
// instructions per value instead of two and about thirteen). It differs from the reference's rational polynomial
// (stage_from_linear.cc:114-144 via transfer_functions-inl.h:245-268, itself an approximation of this curve to ~5e-7) by
// about 1e-6, far below half an 8-bit level; float output keeps the polynomial (LinearToSrgb2).
__device__ __forceinline__ f2 LinearToSrgb2ForU8(f2 v) {
  // no absolute value / sign restoration: a negative sample takes the linear branch, stays negative, and the 8-bit
  // conversion saturates it to 0 exactly as it does the reference's -f(|v|)
  const f2 p = f2{__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(v.x) * (1.0f / 2.4f)), __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(v.y) * (1.0f / 2.4f))};
  const f2 hi = p * 1.055f - 0.055f;
  const f2 lo = v * 12.92f;
  return f2{v.x > 0.0031308f ? hi.x : lo.x, v.y > 0.0031308f ? hi.y : lo.y};
}

// U8SRGB: the launch's frames all write 8-bit sRGB and nothing else (no float output, no filtered planes, no linear
// output): the common case, compiled without the other writers so that its scalars fit the SGPR file (the general form
// spills 32 of them to VGPR lanes and pays a v_readlane per use).
//
// The kernel is bound by instruction issue (a packed f32 instruction occupies the SIMD for 8 cycles, any other vector
// instruction for 4; one step of one wave measured 1430 cycles = 84 packed + 190 other instructions), so the step is
// written to the instruction: whatever involves a neighbouring lane is spelled per element, so that the DPP shift folds
// into the consuming add / multiply-add (a pair built from a shifted element costs two moves before a packed
// instruction can use it), absolute values ride on source modifiers of unpacked instructions (packed ones have none),
// every address is a scalar base plus a loop-invariant 32-bit lane offset (the saddr form of the global instructions:
// no vector address arithmetic in the loop), the mirrored input row is a two-scalar state machine instead of a
// reflection loop, and every load of the next step is unconditional (clamped rows), which leaves no phi copies.
// Two adjacent pixels as two plain floats (see k_filter_rows2: packed f32 instructions buy no issue time on this chip
// and want their scalar operands duplicated into aligned SGPR pairs).
struct P2 {
  float x, y;
};
__device__ __forceinline__ P2 operator+(P2 a, P2 b) { return P2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ P2 operator-(P2 a, P2 b) { return P2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ P2 operator*(P2 a, P2 b) { return P2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ P2 operator+(P2 a, float b) { return P2{a.x + b, a.y + b}; }
__device__ __forceinline__ P2 operator-(P2 a, float b) { return P2{a.x - b, a.y - b}; }
__device__ __forceinline__ P2 operator*(P2 a, float b) { return P2{a.x * b, a.y * b}; }
__device__ __forceinline__ P2 operator*(float b, P2 a) { return P2{b * a.x, b * a.y}; }
__device__ __forceinline__ P2 Max0(P2 a) { return P2{__builtin_fmaxf(a.x, 0.0f), __builtin_fmaxf(a.y, 0.0f)}; }
__device__ __forceinline__ P2 Srgb2(P2 v) {
  const f2 r = LinearToSrgb2(f2{v.x, v.y});
  return P2{r.x, r.y};
}
__device__ __forceinline__ P2 Srgb2ForU8(P2 v) {
  const f2 r = LinearToSrgb2ForU8(f2{v.x, v.y});
  return P2{r.x, r.y};
}
// gfx9 raw buffer resource over "everything from p on": base + scalar offset + 32-bit lane offset addressing with no
// vector address arithmetic (the offsets the kernel forms are inside the allocation by construction)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t RawBuffer(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xFFFFFFFFu, 0x00020000);
}
template <int AUX = 0>
__device__ __forceinline__ float BufF32(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ P2 BufP2(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
  const f2 f = __builtin_bit_cast(f2, v);  // (the whole vector: __builtin_bit_cast of one element reads element 0 with this clang)
  return P2{f.x, f.y};
}

// EPF = 2 (two iterations: what encoders choose between d1.5 and d4): the EPF1 outputs are not emitted but kept in one
// more ring of rows, and the last stage (stage_epf.cc:369-494: four plus-shaped neighbours, single-point SADs) runs on
// them one row behind, with one more halo row and the fourth halo column the pair layout already had. Rows just outside
// the frame are EPF1 outputs of mirrored inputs with sigma and block-border flag taken at the mirrored row, which is the
// mirrored EPF1 output (the filters are reflection-symmetric).
// GAB = false (what the reference's encoder writes at its fast efforts, Gaborish off: enc_frame.cc:316-322): the
// "Gaborish output" ring is fed with the input row itself, one row behind like the filtered one would be; nothing else
// changes (the halo stays the same, one row and column more than needed).
template <bool U8SRGB, int EPF = 1, bool GAB = true>
__global__ __launch_bounds__(64 * kRowsWaves) void k_filter_rows2(const FusedFilterParams* params, int strip_rows) {
  static_assert(EPF == 1 || EPF == 2, "(Gaborish +) one or two EPF iterations");
  constexpr int HALO = kRowsHalo + (EPF == 2 ? 1 : 0);
  FusedFilterParams P;
  LoadParams(P, params + blockIdx.z);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xs = int(P.f.xs), ys = int(P.f.ys);
  const int xw0 = (int(blockIdx.x) * kRowsWaves + wave) * kRows2Cols;  // first output column of the wave (even)
  // output rows per wave: the host picks the tallest strip that still fills the chip (6 halo steps per strip are overhead)
  const int y0 = int(P.f.y_begin) + int(blockIdx.y) * strip_rows;
  const int y1 = y0 + strip_rows < int(P.f.y_end) ? y0 + strip_rows : int(P.f.y_end);
  if (xw0 >= xs || y0 >= y1) return;  // the grid covers the largest frame (band) of the launch
  const int x = xw0 - 4 + 2 * lane;   // the pair's first column
  const int mx0 = MirrorI(x, xs), mx1 = MirrorI(x + 1, xs);
  const bool emit_lane = lane >= 2 && lane < 62;
  const bool emit0 = emit_lane && x < xs, emit1 = emit_lane && x + 1 < xs;
  const bool xb0 = ((mx0 & 7) == 0) || ((mx0 & 7) == 7), xb1 = ((mx1 & 7) == 0) || ((mx1 & 7) == 7);
  const size_t gplane = size_t(P.f.xp) * P.f.yp;
  typedef float __attribute__((address_space(1)))* GF32W;
  typedef f2 __attribute__((address_space(1)))* GF32x2W;
  // whole wave inside the frame: one 8-byte load per lane and plane (x is even, rows are 8-byte aligned); waves at the
  // left / right frame edge load the two mirrored columns separately
  const bool paired = __all(x >= 0 && x + 1 < xs) && (P.f.xp & 1) == 0;
  // loop-invariant lane offsets in bytes; the per-step part of every address is a scalar
  const uint32_t vo0 = uint32_t(mx0) * 4u, vo1 = uint32_t(mx1) * 4u;  // input columns
  const uint32_t vsig = uint32_t(mx0 >> 3) * 4u;                      // x is even: the pair shares its 8x8 block
  const uint32_t vout = uint32_t(x) * 3u;                             // interleaved RGB column (lanes with x < 0 never store)
  const uint32_t vd0 = uint32_t(x & 31) * 4u;                         // dither columns: channel 0 (even: pair contiguous)
  const uint32_t vd1a = uint32_t((x + 23) & 31) * 4u, vd1b = uint32_t((x + 24) & 31) * 4u;  // channel 1 (odd: may wrap)
  const uint32_t vd2 = uint32_t((x + 46) & 31) * 4u;                  // channel 2 (even)
  const __amdgpu_buffer_rsrc_t in_buf = RawBuffer(P.f.in), sig_buf = RawBuffer(P.f.inv_sigma);
  const __amdgpu_buffer_rsrc_t dither_buf = RawBuffer(c_dither), rgb_buf = RawBuffer(P.f.rgb);
  const uint32_t plane_bytes = uint32_t(gplane * 4);  // (three planes of at most 1 GiB: offsets fit 32 bits)
  const GF32W filtered = U8SRGB ? nullptr : (GF32W)(uintptr_t)P.filtered;
  const bool has_rgb = U8SRGB || P.f.rgb != nullptr;
  const GF32W rgbf = U8SRGB ? nullptr : (GF32W)(uintptr_t)P.f.rgbf;
  const bool linear_output = U8SRGB ? false : P.f.linear_output != 0;
  const P2 zero2 = P2{0.0f, 0.0f};
  // sliding windows as rings of 4 rows indexed with the step's phase (the row loop is unrolled by 4): a window shift is a
  // renaming, not register moves
  P2 p[3][4], h1[3][4], g[3][4], dh[4], dv[4];
  P2 e[3][4];            // EPF == 2: the EPF1 outputs of the last rows
  float is_prev = 0.0f;  // EPF == 2: 1 / sigma of the row before the current EPF1 row (the row the last stage emits)
  P2 pv_prev = zero2, dh_new = zero2;
#pragma unroll
  for (int c = 0; c < 3; c++) {
#pragma unroll
    for (int k = 0; k < 4; k++) p[c][k] = h1[c][k] = g[c][k] = e[c][k] = zero2;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) dh[k] = dv[k] = zero2;
  const int steps = (y1 - y0 + 2 * HALO + 3) / 4 * 4;  // whole groups of 4 steps (the extra steps emit nothing)
  // the mirrored input row as a state machine: image_ops.h:184-196 maps ..., -2, -1, 0, 1, ... to ..., 1, 0, 0, 1, ...
  // (and likewise at the far end), a triangle wave whose turning samples repeat; (ym, ydir) walks it one row per step
  int ym = MirrorI(y0 - HALO, ys), ydir;
  {
    const int ym_next = MirrorI(y0 - HALO + 1, ys);
    ydir = ym_next > ym ? 1 : (ym_next < ym ? -1 : (ym == 0 ? -1 : 1));
  }
  constexpr int kPlaneAux = 0;  // (default policy: neighbouring waves and strips re-read the halo)
  auto load_row = [&](P2 (&dst)[3]) {  // row ym, then advance the state machine
    const uint32_t row = uint32_t(ym) * uint32_t(P.f.xp) * 4u;
    if (paired) {
#pragma unroll
      for (int c = 0; c < 3; c++) dst[c] = BufP2<kPlaneAux>(in_buf, vo0, row + c * plane_bytes);
    } else {
#pragma unroll
      for (int c = 0; c < 3; c++) dst[c] = P2{BufF32<kPlaneAux>(in_buf, vo0, row + c * plane_bytes), BufF32<kPlaneAux>(in_buf, vo1, row + c * plane_bytes)};
    }
    int nxt = ym + ydir;
    if (nxt < 0) {
      nxt = 0;
      ydir = 1;
    } else if (nxt >= ys) {
      nxt = ys - 1;
      ydir = -1;
    }
    ym = nxt;
  };
  // everything a step reads from memory is loaded one step ahead (see k_filter_rows), unconditionally: rows outside the
  // strip's output range are clamped into the frame and their values never used
  // (rs: the row whose sigma the next step's EPF1 row needs; r: the row the next step emits, for the dither cells)
  auto load_aux = [&](int rs, int r, float& is, P2 (&di)[3]) {
    const int rc = rs < 0 ? 0 : (rs >= ys ? ys - 1 : rs);
    is = BufF32(sig_buf, vsig, uint32_t(rc >> 3) * uint32_t(P.f.xb) * 4u);
    if (!has_rgb) return;
    di[0] = BufP2(dither_buf, vd0, uint32_t(r & 31) * 128u);
    di[1] = P2{BufF32(dither_buf, vd1a, uint32_t((r + 13) & 31) * 128u), BufF32(dither_buf, vd1b, uint32_t((r + 13) & 31) * 128u)};
    di[2] = BufP2(dither_buf, vd2, uint32_t((r + 26) & 31) * 128u);
  };
  P2 nx[3], ndi[3] = {zero2, zero2, zero2};
  float nis = 0.0f;
  load_row(nx);
  load_aux(y0 - 2 * HALO + (EPF == 2 ? 1 : 0), y0 - 2 * HALO, nis, ndi);
  auto step = [&](auto phase, int j) {
    constexpr int PH = decltype(phase)::value;
    // ring slots: N = newest (written in this step), M1 / M2 / M3 = one / two / three rows older
    constexpr int N = PH % 4, M1 = (PH + 3) % 4, M2 = (PH + 2) % 4, M3 = (PH + 1) % 4;
    const P2 cur[3] = {nx[0], nx[1], nx[2]};
    const float is = nis;
    const P2 di[3] = {ndi[0], ndi[1], ndi[2]};
    // r: the row this step emits (inside the frame when it does: r == its mirror); r1: the row whose EPF1 output this
    // step forms (EPF == 1: the same row; EPF == 2: one further, from y0 - 1 to y1, i.e. possibly one row outside the frame)
    const int r = y0 - 2 * HALO + j;
    const int r1 = EPF == 2 ? r + 1 : r;
    load_row(nx);
    load_aux(r1 + 1, r + 1, nis, ndi);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      p[c][N] = cur[c];
      if constexpr (GAB) h1[c][N] = P2{FromLeft(cur[c].y) + cur[c].y, cur[c].x + FromRight(cur[c].x)};  // columns x - 1 and x + 1 of each element
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const P2 m = p[c][M1];
      if constexpr (GAB) {
        const P2 s1 = h1[c][M1] + (p[c][M2] + p[c][N]);
        const P2 s2 = h1[c][M2] + h1[c][N];
        g[c][N] = s2 * P.f.gab_w[c * 3 + 2] + (s1 * P.f.gab_w[c * 3 + 1] + m * P.f.gab_w[c * 3]);
      } else {
        g[c][N] = m;
      }
    }
    dh[N] = dh_new;
    {
      P2 a = zero2, b = zero2;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const P2 gn = g[c][N], gm = g[c][M1];
        a.x = __builtin_fabsf(gn.x - gn.y) * P.f.ch_scale[c] + a.x;  // |G(x) - G(x + 1)|: the pair's other element ...
        a.y = __builtin_fabsf(FromRight(gn.x) - gn.y) * P.f.ch_scale[c] + a.y;  // ... and the next lane's first (the shifted operand first: it folds into the subtraction)
        b.x = __builtin_fabsf(gm.x - gn.x) * P.f.ch_scale[c] + b.x;
        b.y = __builtin_fabsf(gm.y - gn.y) * P.f.ch_scale[c] + b.y;
      }
      dh_new = a;
      dv[N] = b;
    }
    // (D(M1) + D(M2)) + (left of D(M1) + D(N)) + right of D(M1), the neighbour terms per element
    auto plus_sum = [&](const P2 (&d)[4]) {
      const P2 t = d[M1] + d[M2];
      const P2 u = P2{FromLeft(d[M1].y) + d[N].x, d[M1].x + d[N].y};
      const P2 w = t + u;
      return P2{w.x + d[M1].y, w.y + FromRight(d[M1].x)};
    };
    const P2 pv = plus_sum(dv);
    const P2 ph = plus_sum(dh);
    if (EPF == 2 ? (j >= 2 * HALO - 2 && r1 <= y1) : (j >= 2 * HALO && r < y1)) {
      P2 o[3] = {g[0][M2], g[1][M2], g[2][M2]};
      {
        // no branch on the lane's sigma here: the DPP reads below must see their neighbours whatever the neighbours' own
        // sigma is (a lane switched off by a divergent branch reads as 0); the unfiltered pixels are selected at the end
        // sigma too small: pixels unchanged (the pair shares a block). Not a select per output: 1 / sigma becomes -infinity,
        // every weight then max(0, sad * -inf + 1) = 0 (a zero sad gives NaN, and v_max returns its other operand), the sum
        // of weights 1, its reciprocal exactly 1 and the output the centre pixel itself, bit for bit.
        const float is_k = is < -3.90524291751269967465540850526868f ? -__builtin_inff() : is;
        const int rm1 = EPF == 2 ? (r1 < 0 ? -1 - r1 : (r1 >= ys ? 2 * ys - 1 - r1 : r1)) : r;  // (the EPF1 row, mirrored into the frame)
        const bool yb = ((rm1 & 7) == 0) || ((rm1 & 7) == 7);
        const P2 inv_sig = P2{is_k * ((xb0 || yb) ? P.bsm[1] : P.sm[1]), is_k * ((xb1 || yb) ? P.bsm[1] : P.sm[1])};
        // neighbours in the reference's order: up, left, right, down (SADs pv_prev, ph of column x - 1, ph, pv)
        P2 wk[4];
        wk[0] = pv_prev * inv_sig + 1.0f;
        wk[1] = P2{FromLeft(ph.y) * inv_sig.x + 1.0f, ph.x * inv_sig.y + 1.0f};
        wk[2] = ph * inv_sig + 1.0f;
        wk[3] = pv * inv_sig + 1.0f;
        P2 wsum = P2{1.0f, 1.0f};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          wk[k] = Max0(wk[k]);
          wsum = wsum + wk[k];
        }
        const P2 inv_w = P2{__builtin_amdgcn_rcpf(wsum.x), __builtin_amdgcn_rcpf(wsum.y)};  // 1 ulp; wsum >= 1
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const P2 oc = o[c];
          P2 a = wk[0] * g[c][M3] + oc;
          a.x = FmacFromLeft(a.x, oc.y, wk[1].x);
          a.y = wk[1].y * oc.x + a.y;
          a.x = wk[2].x * oc.y + a.x;
          a.y = FmacFromRight(a.y, oc.x, wk[2].y);
          a = wk[3] * g[c][M1] + a;
          o[c] = a * inv_w;
        }
      }
      if constexpr (EPF == 2) {
#pragma unroll
        for (int c = 0; c < 3; c++) e[c][N] = o[c];
      }
     if (EPF == 1 || (j >= 2 * HALO && r < y1)) {
      if constexpr (EPF == 2) {
        // the last stage on rows r - 1 (M2), r (M1), r + 1 (N) of the EPF1 outputs; neighbours in the reference's order: up,
        // left, right, down; a neighbour's weight from the three channels' differences at that one position
        const float is2 = is_prev;
        const bool keep2 = is2 < -3.90524291751269967465540850526868f;
        const bool yb2 = ((r & 7) == 0) || ((r & 7) == 7);
        const P2 inv_sig2 = P2{is2 * ((xb0 || yb2) ? P.bsm[2] : P.sm[2]), is2 * ((xb1 || yb2) ? P.bsm[2] : P.sm[2])};
        P2 nb[4][3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const P2 cc = e[c][M1];
          nb[0][c] = e[c][M2];
          nb[1][c] = P2{FromLeft(cc.y), cc.x};
          nb[2][c] = P2{cc.y, FromRight(cc.x)};
          nb[3][c] = e[c][N];
        }
        P2 wsum = P2{1.0f, 1.0f};
        P2 acc[3] = {e[0][M1], e[1][M1], e[2][M1]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          P2 sad = zero2;
#pragma unroll
          for (int c = 0; c < 3; c++) {
            sad.x = __builtin_fabsf(nb[k][c].x - e[c][M1].x) * P.f.ch_scale[c] + sad.x;
            sad.y = __builtin_fabsf(nb[k][c].y - e[c][M1].y) * P.f.ch_scale[c] + sad.y;
          }
          const P2 wk2 = Max0(sad * inv_sig2 + 1.0f);
          wsum = wsum + wk2;
#pragma unroll
          for (int c = 0; c < 3; c++) acc[c] = wk2 * nb[k][c] + acc[c];
        }
        const P2 inv_w2 = P2{__builtin_amdgcn_rcpf(wsum.x), __builtin_amdgcn_rcpf(wsum.y)};
#pragma unroll
        for (int c = 0; c < 3; c++) o[c] = P2{keep2 ? e[c][M1].x : acc[c].x * inv_w2.x, keep2 ? e[c][M1].y : acc[c].y * inv_w2.y};
      }
      // U8SRGB: only the stores sit under the lanes' emit mask. With the whole block under it the compiler sinks the EPF
      // sums and the colour arithmetic into the masked region too (nothing else reads them) and must then take every
      // neighbour-lane read out of it as a v_mov_dpp of its own (a DPP operand of a masked instruction reads a disabled
      // neighbour as 0): 12 extra vector instructions per step.
      if (U8SRGB || emit0) {
        if (!U8SRGB && filtered) {
          const size_t gi = size_t(r) * P.f.xp + x;
#pragma unroll
          for (int c = 0; c < 3; c++) {
            filtered[c * gplane + gi] = o[c].x;
            if (emit1) filtered[c * gplane + gi + 1] = o[c].y;
          }
        }
        if (has_rgb || rgbf) {
          const P2 X = o[0], Y = o[1], Bc = o[2];
          const P2 gr = (Y + X) - P.f.opsin_bias_cbrt[0], gg = (Y - X) - P.f.opsin_bias_cbrt[1], gb = Bc - P.f.opsin_bias_cbrt[2];
          const P2 mr = (gr * gr) * gr + P.f.opsin_bias[0], mg = (gg * gg) * gg + P.f.opsin_bias[1], mb = (gb * gb) * gb + P.f.opsin_bias[2];
          P2 cr = P.f.opsin_inv[2] * mb + (P.f.opsin_inv[1] * mg + P.f.opsin_inv[0] * mr);
          P2 cg = P.f.opsin_inv[5] * mb + (P.f.opsin_inv[4] * mg + P.f.opsin_inv[3] * mr);
          P2 cb = P.f.opsin_inv[8] * mb + (P.f.opsin_inv[7] * mg + P.f.opsin_inv[6] * mr);
          const bool even = ((r & xs) & 1) == 0;  // (r * xs + x) * 3 with x even: the row's parity decides the alignment
          if (!U8SRGB && rgbf) {  // float output (stage_write.cc:334-370): the samples as they are
            if (!linear_output) {
              cr = Srgb2(cr);
              cg = Srgb2(cg);
              cb = Srgb2(cb);
            }
            const GF32W d = rgbf + (size_t(r) * xs + x) * 3;
            if (emit1 && even) {  // six floats from an 8-byte aligned offset
              *(GF32x2W)(d) = f2{cr.x, cg.x};
              *(GF32x2W)(d + 2) = f2{cb.x, cr.y};
              *(GF32x2W)(d + 4) = f2{cg.y, cb.y};
            } else {
              d[0] = cr.x;
              d[1] = cg.x;
              d[2] = cb.x;
              if (emit1) {
                d[3] = cr.y;
                d[4] = cg.y;
                d[5] = cb.y;
              }
            }
          } else {
            if (!linear_output) {
              cr = Srgb2ForU8(cr);
              cg = Srgb2ForU8(cg);
              cb = Srgb2ForU8(cb);
            }
            // v_cvt_pk_u8_f32 rounds to nearest even, saturates to 0..255 and drops the byte into place
            // (scripts/cvt_probe.hip): one instruction per sample for clamp + round + pack
            uint32_t w01 = __builtin_amdgcn_cvt_pk_u8_f32(cr.x * 255.0f + di[0].x, 0, 0u);
            w01 = __builtin_amdgcn_cvt_pk_u8_f32(cg.x * 255.0f + di[1].x, 1, w01);
            w01 = __builtin_amdgcn_cvt_pk_u8_f32(cb.x * 255.0f + di[2].x, 2, w01);
            w01 = __builtin_amdgcn_cvt_pk_u8_f32(cr.y * 255.0f + di[0].y, 3, w01);
            uint32_t w2 = __builtin_amdgcn_cvt_pk_u8_f32(cg.y * 255.0f + di[1].y, 0, 0u);
            w2 = __builtin_amdgcn_cvt_pk_u8_f32(cb.y * 255.0f + di[2].y, 1, w2);
            const uint32_t orow = uint32_t(r) * uint32_t(xs) * 3u;  // (at most 16K x 16K x 3 bytes)
            // (cache policy, measured alone / pipelined step: the non-temporal hint on these 2-byte stores 22.6 -> 24.0 ms /
            // 58.2 -> 60.0 ms: they need the L2 to merge them into lines; on the plane loads 22.6 -> 24.8 ms)
            constexpr int kRgbAux = 0;
            // (the packed bytes exist for every lane before the masked stores: without this the optimiser sinks the
            // arithmetic that only feeds them under the mask again)
            if constexpr (U8SRGB) asm volatile("" : "+v"(w01), "+v"(w2));
            // (the RGB stores are 11 % of the kernel: 22.6 ms alone, 20.0 without them. Measured and not kept: two lanes' twelve
            // bytes as an 8-byte store of the even lane and a 4-byte one of the odd lane, 22.6 -> 22.8 ms, pipelined step + 3 %.)
            if (U8SRGB && !emit0) {
              // (a lane of the halo columns, or beyond the frame: nothing to store)
            } else if (even) {  // six bytes from an even offset: 16-bit stores
              __builtin_amdgcn_raw_buffer_store_b16(uint16_t(w01), rgb_buf, vout, orow, kRgbAux);
              if (emit1) {
                __builtin_amdgcn_raw_buffer_store_b16(uint16_t(w01 >> 16), rgb_buf, vout + 2, orow, kRgbAux);
                __builtin_amdgcn_raw_buffer_store_b16(uint16_t(w2), rgb_buf, vout + 4, orow, kRgbAux);
              } else {
                __builtin_amdgcn_raw_buffer_store_b8(uint8_t(w01 >> 16), rgb_buf, vout + 2, orow, 0);
              }
            } else {
              __builtin_amdgcn_raw_buffer_store_b8(uint8_t(w01), rgb_buf, vout, orow, 0);
              __builtin_amdgcn_raw_buffer_store_b8(uint8_t(w01 >> 8), rgb_buf, vout + 1, orow, 0);
              __builtin_amdgcn_raw_buffer_store_b8(uint8_t(w01 >> 16), rgb_buf, vout + 2, orow, 0);
              if (emit1) {
                __builtin_amdgcn_raw_buffer_store_b8(uint8_t(w01 >> 24), rgb_buf, vout + 3, orow, 0);
                __builtin_amdgcn_raw_buffer_store_b8(uint8_t(w2), rgb_buf, vout + 4, orow, 0);
                __builtin_amdgcn_raw_buffer_store_b8(uint8_t(w2 >> 8), rgb_buf, vout + 5, orow, 0);
              }
            }
          }
        }
      }
     }
    }
    if constexpr (EPF == 2) is_prev = is;
    pv_prev = pv;
  };
  for (int j = 0; j < steps; j += 4) {
    step(std::integral_constant<int, 0>{}, j);
    step(std::integral_constant<int, 1>{}, j + 1);
    step(std::integral_constant<int, 2>{}, j + 2);
    step(std::integral_constant<int, 3>{}, j + 3);
  }
}

template <bool GAB, int EPF>
__global__ __launch_bounds__(kFusedThreads) void k_filter_fused(const FusedFilterParams* params) {
  // one frame per grid z slice; its parameter block is read through the constant address space (scalar loads that the
  // compiler may hoist out of the pixel loops: a plain global reference is re-read after every store)
  FusedFilterParams P;
  LoadParams(P, params + blockIdx.z);
  constexpr int H = FusedHalo(GAB, EPF);
  constexpr int TW = kFusedTW, TH = kFusedTH, S = TW + 2 * H, SH = TH + 2 * H, PL = S * SH;
  extern __shared__ __align__(16) float lds_ff[];
  float* src = lds_ff;
  float* dst = lds_ff + 3 * PL;
  float* l_sig = lds_ff + 6 * PL;  // 1 / sigma of the 8x8 blocks around the tile (block columns bx0 - 1 .., rows by0 - 1 ..)
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TW, y0 = int(P.f.y_begin) + blockIdx.y * TH;
  const int xs = int(P.f.xs), ys = int(P.f.ys);
  if (x0 >= xs || y0 >= int(P.f.y_end)) return;  // the grid covers the largest frame (band) of the launch
  const size_t gplane = size_t(P.f.xp) * P.f.yp;
  const int sbx = (x0 >> 3) - 1, sby = (y0 >> 3) - 1;
  if (EPF > 0 && tid < kFusedSigW * kFusedSigH) {
    int bx = sbx + tid % kFusedSigW, by = sby + tid / kFusedSigW;
    bx = bx < 0 ? 0 : (bx >= int(P.f.xb) ? int(P.f.xb) - 1 : bx);
    by = by < 0 ? 0 : (by >= int((P.f.ys + 7) / 8) ? int((P.f.ys + 7) / 8) - 1 : by);
    l_sig[tid] = P.f.inv_sigma[size_t(by) * P.f.xb + bx];
  }
  // ---- tile + halo from HBM, mirrored about the frame size
  {
    // all loads of the thread are issued before the first LDS store (the HBM latency is paid once, not per element)
    constexpr int NLD = (PL + kFusedThreads - 1) / kFusedThreads;
    float v[NLD][3];
    const bool interior = x0 >= H && y0 >= H && x0 + TW + H <= xs && y0 + TH + H <= ys;  // no mirroring needed (uniform)
#pragma unroll
    for (int it = 0; it < NLD; it++) {
      const int i = tid + it * kFusedThreads;
      if (i < PL) {
        const int ly = i / S, lx = i - ly * S;
        const size_t g = interior ? size_t(y0 + ly - H) * P.f.xp + (x0 + lx - H)
                                  : size_t(MirrorI(y0 + ly - H, ys)) * P.f.xp + MirrorI(x0 + lx - H, xs);
        v[it][0] = P.f.in[g];
        v[it][1] = P.f.in[gplane + g];
        v[it][2] = P.f.in[2 * gplane + g];
      }
    }
#pragma unroll
    for (int it = 0; it < NLD; it++) {
      const int i = tid + it * kFusedThreads;
      if (i < PL) {
        src[i] = v[it][0];
        src[PL + i] = v[it][1];
        src[2 * PL + i] = v[it][2];
      }
    }
  }
  __syncthreads();
  int h = H;  // halo still valid around the tile in `src`
  if (GAB && !(P.debug & 1)) {
    h -= 1;
    constexpr int HG = H - 1, w = TW + 2 * HG, n = w * (TH + 2 * HG), NIT = (n + kFusedThreads - 1) / kFusedThreads;
#pragma unroll 4
    for (int it = 0; it < NIT; it++) {
      const int i = tid + it * kFusedThreads;
      if (i >= n) break;
      const int ry = i / w, rx = i - ry * w;
      const int o = (ry + H - HG) * S + rx + H - HG;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float* p = src + c * PL;
        const float m = p[o];
        const float s1 = (p[o - 1] + p[o + 1]) + (p[o - S] + p[o + S]);
        const float s2 = (p[o - S - 1] + p[o - S + 1]) + (p[o + S - 1] + p[o + S + 1]);
        dst[c * PL + o] = s2 * P.f.gab_w[c * 3 + 2] + (s1 * P.f.gab_w[c * 3 + 1] + m * P.f.gab_w[c * 3]);
      }
    }
    __syncthreads();
    float* t = src;
    src = dst;
    dst = t;
  }
#define JXL_EPF_STAGE(STAGE_, RADIUS_)                                                                       \
  {                                                                                                          \
    h -= RADIUS_;                                                                                            \
    const int w = TW + 2 * h, n = w * (TH + 2 * h);                                                          \
    for (int i = tid; i < n; i += kFusedThreads) {                                                                     \
      const int ry = i / w, rx = i - ry * w;                                                                 \
      const int o = (ry + H - h) * S + rx + H - h;                                                           \
      const int mx = MirrorI(x0 + rx - h, xs), my = MirrorI(y0 + ry - h, ys);                                \
      const float is = l_sig[((my >> 3) - sby) * kFusedSigW + (mx >> 3) - sbx];                              \
      const bool border = ((mx & 7) == 0) || ((mx & 7) == 7) || ((my & 7) == 0) || ((my & 7) == 7);          \
      EpfPixel<STAGE_, S, PL>(P, src, o, is, border, &dst[o], &dst[PL + o], &dst[2 * PL + o]);               \
    }                                                                                                        \
    __syncthreads();                                                                                         \
    float* t = src;                                                                                          \
    src = dst;                                                                                               \
    dst = t;                                                                                                 \
  }
  if (EPF >= 3) JXL_EPF_STAGE(0, 3)
  if (EPF >= 1 && !(P.debug & 2)) {
    constexpr int HO = H - (GAB ? 1 : 0) - (EPF >= 3 ? 3 : 0) - 2;
    Epf1Stage<S, PL, HO, H>(P, src, dst, l_sig, x0, y0, xs, ys, tid);
    h = HO;
    float* t = src;
    src = dst;
    dst = t;
  }
  if (EPF >= 2) JXL_EPF_STAGE(2, 1)
#undef JXL_EPF_STAGE
  // ---- colour: XYB -> linear RGB -> sRGB -> dithered 8 bit, staged as bytes in LDS for wide stores
  uint8_t* bytes = reinterpret_cast<uint8_t*>(dst);
#pragma unroll 2
  for (int i = tid; i < TW * TH; i += kFusedThreads) {
    const int ry = i / TW, rx = i - ry * TW;
    const int o = (ry + H) * S + rx + H;
    const int x = x0 + rx, y = y0 + ry;
    const float X = src[o], Y = src[PL + o], Bc = src[2 * PL + o];
    if (P.filtered && x < xs && y < ys) {
      const size_t g = size_t(y) * P.f.xp + x;
      P.filtered[g] = X;
      P.filtered[gplane + g] = Y;
      P.filtered[2 * gplane + g] = Bc;
    }
    const float gr = (Y + X) - P.f.opsin_bias_cbrt[0], gg = (Y - X) - P.f.opsin_bias_cbrt[1], gb = Bc - P.f.opsin_bias_cbrt[2];
    const float mr = (gr * gr) * gr + P.f.opsin_bias[0], mg = (gg * gg) * gg + P.f.opsin_bias[1], mb = (gb * gb) * gb + P.f.opsin_bias[2];
    float r = P.f.opsin_inv[2] * mb + (P.f.opsin_inv[1] * mg + P.f.opsin_inv[0] * mr);
    float g = P.f.opsin_inv[5] * mb + (P.f.opsin_inv[4] * mg + P.f.opsin_inv[3] * mr);
    float b = P.f.opsin_inv[8] * mb + (P.f.opsin_inv[7] * mg + P.f.opsin_inv[6] * mr);
    if (!P.f.linear_output && !(P.debug & 4)) {
      r = LinearToSrgb(r);
      g = LinearToSrgb(g);
      b = LinearToSrgb(b);
    }
    bytes[i * 3] = ToU8(r, x, y, 0);
    bytes[i * 3 + 1] = ToU8(g, x, y, 1);
    bytes[i * 3 + 2] = ToU8(b, x, y, 2);
  }
  __syncthreads();
  if (!P.f.rgb) return;  // upsampled frames: only the filtered planes are produced here (k_upsample_color follows)
  const int cols = xs - x0 < TW ? xs - x0 : TW;  // valid pixels per tile row
  const int rows = int(P.f.y_end) - y0 < TH ? int(P.f.y_end) - y0 : TH;
  const int row_bytes = cols * 3;
  if (((size_t(xs) * 3) & 3) == 0) {  // every tile row starts 4-byte aligned (x0 * 3 is a multiple of 192)
    const int dw = row_bytes >> 2;
    const uint32_t* b32 = reinterpret_cast<const uint32_t*>(bytes);
    for (int i = tid; i < rows * (TW * 3 / 4); i += kFusedThreads) {
      const int ry = i / (TW * 3 / 4), j = i - ry * (TW * 3 / 4);
      if (j < dw) reinterpret_cast<uint32_t*>(P.f.rgb + (size_t(y0 + ry) * xs + x0) * 3)[j] = b32[i];
    }
    for (int i = tid; i < rows * 4; i += kFusedThreads) {  // up to 3 tail bytes per row
      const int ry = i >> 2, j = (dw << 2) + (i & 3);
      if (j < row_bytes) P.f.rgb[(size_t(y0 + ry) * xs + x0) * 3 + j] = bytes[ry * TW * 3 + j];
    }
  } else {
    for (int i = tid; i < rows * TW * 3; i += kFusedThreads) {
      const int ry = i / (TW * 3), j = i - ry * (TW * 3);
      if (j < row_bytes) P.f.rgb[(size_t(y0 + ry) * xs + x0) * 3 + j] = bytes[i];
    }
  }
}

// ---- noise synthesis (frame flag kNoise)
// Replaces lib/jxl/dec_noise.cc:43-110 (Random3Planes: one Xorshift128+ generator of 8 lanes per 256x256 group of the
// image, seeded by the frame indices and the group origin, fills the group's three planes one after the other, 16 floats
// in [1, 2) per step, a last partial step per row), xorshift128plus-inl.h:30-96, and render_pipeline/stage_noise.cc:
// 262-310 (5x5 high-pass of the random planes, mirrored at the image edges) + :64-260 (strength from the pixel's
// intensity through the 8-point LUT; the noise goes to X, Y, B with the base colour correlation).
struct NoiseParams {
  float* raw;       // [3][ysize][xsize] random planes (k_noise_random -> k_noise_add)
  float* planes;    // filtered X, Y, B: [3][yp][xp] (modified in place)
  uint32_t xsize, ysize, xp, yp, xgroups, ngroups;
  uint32_t seed[2];
  float lut[8];
  float ytox, ytob;
  uint32_t y_begin, y_end;  // pixel rows to produce (band decode)
};
__device__ __forceinline__ uint64_t NoiseSplitMix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// One thread per (group, generator lane): the generator is serial over the group's rows and planes; lane i of a step
// yields floats 2i and 2i + 1 of the step's 16.
__global__ __launch_bounds__(64) void k_noise_random(NoiseParams P) {
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  const uint32_t g = t >> 3, lane = t & 7;
  if (g >= P.ngroups) return;
  const uint32_t x0 = (g % P.xgroups) * 256, y0 = (g / P.xgroups) * 256;
  uint64_t s0 = NoiseSplitMix64(((uint64_t(P.seed[0]) << 32) + P.seed[1]) + 0x9E3779B97F4A7C15ull);
  uint64_t s1 = NoiseSplitMix64(((uint64_t(x0) << 32) + y0) + 0x9E3779B97F4A7C15ull);
  for (uint32_t i = 0; i < lane; i++) {
    s0 = NoiseSplitMix64(s0);
    s1 = NoiseSplitMix64(s1);
  }
  const uint32_t w = min(256u, P.xsize - x0), h = min(256u, P.ysize - y0);
  const uint32_t steps = (w > 16 ? (w - 16 + 15) / 16 : 0) + 1;  // whole steps while x + 16 < w, then one for the rest
  for (uint32_t c = 0; c < 3; c++)
    for (uint32_t y = 0; y < h; y++) {
      float* row = P.raw + (size_t(c) * P.ysize + y0 + y) * P.xsize + x0;
      for (uint32_t b = 0; b < steps; b++) {
        uint64_t a = s0;
        const uint64_t bb = s1;
        const uint64_t bits = a + bb;
        s0 = bb;
        a ^= a << 23;
        a ^= bb ^ (a >> 18) ^ (bb >> 5);
        s1 = a;
        const uint32_t x = b * 16 + 2 * lane;
        if (x < w) row[x] = __uint_as_float((uint32_t(bits) >> 9) | 0x3F800000u);
        if (x + 1 < w) row[x + 1] = __uint_as_float((uint32_t(bits >> 32) >> 9) | 0x3F800000u);
      }
    }
}
__device__ __forceinline__ uint32_t NoiseMirror(int v, int n) {
  while (v < 0 || v >= n) v = v < 0 ? -v - 1 : 2 * n - 1 - v;
  return uint32_t(v);
}
__device__ __forceinline__ float NoiseStrength(const float* lut, float x) {
  const float scaled = fmaxf(0.0f, x * 6.0f);
  float fl = floorf(scaled), frac = scaled - fl;
  if (scaled >= 7.0f) {
    fl = 6.0f;
    frac = 1.0f;
  }
  const int i = int(fl);
  const float lo = i == 0 ? lut[0] : i == 1 ? lut[1] : i == 2 ? lut[2] : i == 3 ? lut[3] : i == 4 ? lut[4] : i == 5 ? lut[5] : lut[6];
  const float hi = i == 0 ? lut[1] : i == 1 ? lut[2] : i == 2 ? lut[3] : i == 3 ? lut[4] : i == 4 ? lut[5] : i == 5 ? lut[6] : lut[7];
  return __builtin_amdgcn_fmed3f((hi - lo) * frac + lo, 0.0f, 1.0f);
}
__global__ __launch_bounds__(256) void k_noise_add(NoiseParams P) {
  const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = P.y_begin + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= P.xsize || y >= P.y_end) return;
  float rnd[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float* pl = P.raw + size_t(c) * P.ysize * P.xsize;
    float others = 0.0f, centre = 0.0f;
#pragma unroll
    for (int dy = -2; dy <= 2; dy++) {
      const float* row = pl + size_t(NoiseMirror(int(y) + dy, int(P.ysize))) * P.xsize;
#pragma unroll
      for (int dx = -2; dx <= 2; dx++) {
        const float v = row[NoiseMirror(int(x) + dx, int(P.xsize))];
        if (dx == 0 && dy == 0) centre = v;
        else others += v;
      }
    }
    rnd[c] = (others * 0.16f + centre * -3.84f) * 0.22f;
  }
  const size_t plane = size_t(P.xp) * P.yp, i = size_t(y) * P.xp + x;
  const float vx = P.planes[i], vy = P.planes[plane + i], vb = P.planes[2 * plane + i];
  const float str_g = NoiseStrength(P.lut, (vy - vx) * 0.5f), str_r = NoiseStrength(P.lut, (vy + vx) * 0.5f);
  const float red = str_r * (0.0078125f * rnd[0] + 0.9921875f * rnd[2]);
  const float green = str_g * (0.0078125f * rnd[1] + 0.9921875f * rnd[2]);
  const float rg = red + green;
  P.planes[i] = (P.ytox * rg + (red - green)) + vx;
  P.planes[plane + i] = vy + rg;
  P.planes[2 * plane + i] = P.ytob * rg + vb;
}

// ---- splines (render_pipeline/stage_splines.cc, lib/jxl/splines.cc:84-133,178-187): every row adds the segments of its
// list, in list order, to the three colour planes: a Gaussian-like splat around the segment centre, FastErff
// (fast_math-inl.h:126-148) of the distance. One workgroup per row; a thread owns the samples x = tid (mod 256), so the
// additions to one sample happen in segment order without synchronisation.
struct SplineParams {
  float* planes;              // [3] planes, `plane_stride` floats apart, rows of `stride` floats
  const float* segments;      // 8 floats each: centre x, y, maximum distance, 1 / sigma, sigma / 4 * intensity, colour[3]
  const uint32_t* row_start;  // [ysize + 1]
  const uint32_t* row_segments;
  size_t stride, plane_stride;
  uint32_t xsize, y_begin, y_end;
};
__device__ __forceinline__ float SplineErf(float x) {
  const float a = fabsf(x);
  const float d1 = __builtin_fmaf(a, 7.77394369e-02f, 2.05260015e-04f);
  const float d2 = __builtin_fmaf(d1, a, 2.32120216e-01f);
  const float d3 = __builtin_fmaf(d2, a, 2.77820801e-01f);
  const float d4 = __builtin_fmaf(d3, a, 1.0f);
  const float d5 = d4 * d4;
  const float inv = 1.0f / d5;
  const float r = __builtin_fmaf(-inv, inv, 1.0f);
  return x <= 0.0f ? -r : r;
}
__device__ __forceinline__ void SplinesAddRow(const SplineParams& P, uint32_t y) {
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = P.row_start[y]; i < P.row_start[y + 1]; i++) {
    const float* g = P.segments + size_t(P.row_segments[i]) * 8;
    const float cx = g[0], cy = g[1], maxd = g[2], inv_sigma = g[3], s4i = g[4];
    const long long start = llroundf(cx - maxd), end = llroundf(cx + maxd);
    if (end < 0 || start >= (long long)P.xsize) continue;
    const uint32_t x0 = uint32_t(max(0ll, start)), x1 = uint32_t(min((long long)P.xsize, end + 1));
    const float dy = float(y) - cy;
    for (uint32_t x = (x0 & ~255u) + tid; x < x1; x += 256) {
      if (x < x0) continue;
      const float dx = float(int(x)) - cx;
      const float distance = sqrtf(__builtin_fmaf(dx, dx, dy * dy));
      const float f = SplineErf(__builtin_fmaf(distance, 0.5f, 0.353553391f) * inv_sigma) -
                      SplineErf(__builtin_fmaf(distance, 0.5f, -0.353553391f) * inv_sigma);
      const float local = s4i * (f * f);
      float* p = P.planes + size_t(y) * P.stride + x;
      for (int c = 0; c < 3; c++) p[c * P.plane_stride] = __builtin_fmaf(g[5 + c], local, p[c * P.plane_stride]);
    }
  }
}
__global__ __launch_bounds__(256) void k_splines_add(SplineParams P) {
  const uint32_t y = P.y_begin + blockIdx.x;
  if (y < P.y_end) SplinesAddRow(P, y);
}
__global__ __launch_bounds__(256) void k_splines_add_batch(const SplineParams* ops) {
  const SplineParams& P = ops[blockIdx.y];
  const uint32_t y = P.y_begin + blockIdx.x;
  if (y < P.y_end) SplinesAddRow(P, y);
}

// ---- patches (lib/jxl/dec_patch_dictionary.cc:317-356 AddOneRow + blending.cc:40-190 PerformBlending + alpha.cc:17-101):
// like the splines, one workgroup per row walks the row's positions in dictionary order and a thread owns the samples
// x = tid (mod 256), so that overlapping patches combine in the order the dictionary gives. Every PatchBlendMode on the
// colour channels; on images whose one extra channel is alpha, every mode on that channel too (`alpha` != NULL: the frame's
// alpha plane, read before and written after each position like the reference's rows; slot_alpha: the reference frames').
// The alpha channel's own mode is evaluated first, from the values before blending; the colour channels' blend above /
// below then writes the alpha channel as well (blending.cc:127-136), whatever its own mode gave. Not contracted: the
// oracle's (the reference's scalar) operation order.
struct PatchParams {
  float* planes;  // [3] planes `plane_stride` floats apart, rows of `stride` floats
  const uint32_t* records;
  const uint32_t* row_start;
  const uint32_t* row_list;
  const float* slot_planes[4];
  uint32_t slot_w[4], slot_h[4];
  size_t stride, plane_stride;
  uint32_t xsize, y_begin, y_end;
  float* alpha;                // NULL: the image has no alpha channel (modes 4 / 5 then replace, 6 / 7 add: blending.cc:154-168)
  const float* slot_alpha[4];  // [slot_h][slot_w]
  size_t alpha_stride;
  uint32_t premultiplied;      // ExtraChannelInfo::alpha_associated
};
__device__ __forceinline__ float PatchClamp01(float v, bool clamp) { return clamp ? __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f) : v; }
__device__ __forceinline__ void PatchesAddRow(const PatchParams& P, uint32_t y, uint32_t tid) {
#pragma clang fp contract(off)
  for (uint32_t i = P.row_start[y]; i < P.row_start[y + 1]; i++) {
    const uint32_t* r = P.records + size_t(P.row_list[i]) * 8;
    const uint32_t px = r[0], py = r[1], w = r[2], rx0 = r[4], ry0 = r[5], slot = r[6];
    uint32_t mode = r[7] & 255;
    const bool clamp = ((r[7] >> 8) & 1) != 0, ec_clamp = ((r[7] >> 24) & 1) != 0;
    const uint32_t ec_mode = P.alpha ? (r[7] >> 16) & 255 : 0;
    if (!P.alpha && mode >= 4) mode = mode >= 6 ? 2 : 1;
    if (mode == 0 && ec_mode == 0) continue;
    const float* src = P.slot_planes[slot];
    const float* src_a = P.alpha ? P.slot_alpha[slot] : nullptr;
    const size_t sw = P.slot_w[slot], splane = sw * P.slot_h[slot];
    const uint32_t x1 = min(px + w, P.xsize);
    for (uint32_t x = (px & ~255u) + tid; x < x1; x += 256) {
      if (x < px) continue;
      const size_t si = size_t(ry0 + (y - py)) * sw + rx0 + (x - px);
      float* p = P.planes + size_t(y) * P.stride + x;
      float* ap = P.alpha ? P.alpha + size_t(y) * P.alpha_stride + x : nullptr;
      const float fga = src_a ? src_a[si] : 1.0f, bga = ap ? *ap : 1.0f;
      float a_out = bga;
      switch (ec_mode) {
        case 1: a_out = fga; break;
        case 2: a_out = bga + fga; break;
        case 3: a_out = bga * PatchClamp01(fga, ec_clamp); break;
        case 4: a_out = 1.0f - (1.0f - PatchClamp01(fga, ec_clamp)) * (1.0f - bga); break;
        case 5: a_out = 1.0f - (1.0f - PatchClamp01(bga, ec_clamp)) * (1.0f - fga); break;
        case 7: a_out = fga; break;
        default: break;  // kNone, and kAlphaWeightedAddAbove of a channel that is its own alpha: unchanged (alpha.cc:74-77)
      }
      // blend above: the patch over the frame; blend below: the frame over the patch (top / bottom layer of alpha.cc:17-43)
      const float top_a = PatchClamp01(mode == 5 ? bga : fga, clamp), bot_a = mode == 5 ? fga : bga;
      const float new_a = 1.0f - (1.0f - top_a) * (1.0f - bot_a);
      const float rnew_a = new_a > 0.0f ? 1.0f / new_a : 0.0f;
      for (int c = 0; c < 3; c++) {
        const float fg = src[c * splane + si], bg = p[c * P.plane_stride];
        float o = bg;
        switch (mode) {
          case 1: o = fg; break;
          case 2: o = bg + fg; break;
          case 3: o = bg * PatchClamp01(fg, clamp); break;
          case 4: o = P.premultiplied ? fg + bg * (1.0f - top_a) : (fg * top_a + bg * bga * (1.0f - top_a)) * rnew_a; break;
          case 5: o = P.premultiplied ? bg + fg * (1.0f - top_a) : (bg * top_a + fg * fga * (1.0f - top_a)) * rnew_a; break;
          case 6: o = bg + fg * PatchClamp01(fga, clamp); break;
          case 7: o = fg + bg * PatchClamp01(bga, clamp); break;
          default: break;
        }
        p[c * P.plane_stride] = o;
      }
      if (mode == 4 || mode == 5) a_out = new_a;
      if (ap) *ap = a_out;
    }
  }
}

__global__ __launch_bounds__(256) void k_patches_add(PatchParams P) {
  const uint32_t y = P.y_begin + blockIdx.x;
  if (y < P.y_end) PatchesAddRow(P, y, threadIdx.x);
}
// The same for a set of frames (the Modular path's parameter-block launches): one grid row of workgroups per frame.
__global__ __launch_bounds__(256) void k_patches_add_batch(const PatchParams* ops) {
  const PatchParams& P = ops[blockIdx.y];
  const uint32_t y = P.y_begin + blockIdx.x;
  if (y < P.y_end) PatchesAddRow(P, y, threadIdx.x);
}

}  // namespace jxlhip
#endif  // JXL_HIP_FILTER_FUSED_H_
