// libjxl_amd: the block-resolution stencils of the DC path as kernels.
//   k_dc_dequant = DequantDC, reference lib/jxl/compressed_dc.cc:201-296 (the three coded integer planes times the channels'
//                  quantisation steps, each DC group's extra-precision shift, chroma from luma with the DC factors).
//   k_dc_smooth  = AdaptiveDCSmoothing, reference lib/jxl/compressed_dc.cc:50-52 (weights), :64-128 (ComputePixelChannel /
//                  ComputePixel: the multiply-adds below are the reference's MulAdd chain), :130-198 (borders kept).
//   k_epf_sigma  = ComputeSigma, reference lib/jxl/epf.cc:39-81 (1 / sigma per 8x8 block from the varblock's quant-field
//                  value and the block's sharpness; the reference's mirrored padding is not stored: the filter kernels
//                  mirror their reads about the frame).
// They run inside jxlhip_frame_upload, on the copy stream behind the frame's table copy, for frames whose descriptor asks
// for it (JxlHipFrameDesc::dc_quantised / ::dc_smoothing / ::sharpness): the host front-end then does no per-block float
// work at all.
#ifndef JXL_HIP_DC_H_
#define JXL_HIP_DC_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/jxl_amd_hip.h"

namespace jxlhip {

struct DcDequantParams {
  const int32_t* q;   // [3][ys][xs]: the coded integers of X, Y, B
  float* out;         // [3][ys][xs]
  uint32_t xs, ys, xgroups;   // xgroups: DC groups (256 x 256 blocks) per row
  const uint8_t* extra_precision;  // per DC group: the values are in units of step / 2^this (NULL: 0)
  float step[3];      // DC quantisation step of X, Y, B
  float cfl_x, cfl_b; // chroma from luma at DC: base correlation + DC factor * colour scale
  uint32_t cs;        // chroma-subsampled frames (JxlHipFrameDesc::chroma_hshift / _vshift): hshift of channel c in bit 2 * c,
                      // vshift in bit 2 * c + 1; 0 = 4:4:4. Non-zero: every channel on its own grid in the top-left part of its
                      // plane, no chroma from luma (compressed_dc.cc:232-250)
};

// One thread per block. The products are not contracted into multiply-adds: the oracle's (and the reference's scalar)
// order of operations, so that the planes agree bit for bit.
__global__ __launch_bounds__(256) void k_dc_dequant(DcDequantParams P) {
#pragma clang fp contract(off)
  const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= P.xs || y >= P.ys) return;
  const size_t plane = size_t(P.xs) * P.ys, at = size_t(y) * P.xs + x;
  if (P.cs) {
    for (uint32_t c = 0; c < 3; c++) {
      const uint32_t hs = (P.cs >> (2 * c)) & 1u, vs = (P.cs >> (2 * c + 1)) & 1u;
      if (x >= (P.xs >> hs) || y >= (P.ys >> vs)) continue;
      // (the sample belongs to the DC group of the frame's block it is coded with)
      const uint32_t epc = P.extra_precision ? P.extra_precision[((y << vs) >> 8) * P.xgroups + ((x << hs) >> 8)] : 0u;
      P.out[c * plane + at] = float(P.q[c * plane + at]) * (P.step[c] * (1.0f / float(1u << (epc & 3u))));
    }
    return;
  }
  const uint32_t ep = P.extra_precision ? P.extra_precision[(y >> 8) * P.xgroups + (x >> 8)] : 0u;
  const float mul = 1.0f / float(1u << (ep & 3u));
  const float in_x = float(P.q[at]) * (P.step[0] * mul);
  const float in_y = float(P.q[plane + at]) * (P.step[1] * mul);
  const float in_b = float(P.q[2 * plane + at]) * (P.step[2] * mul);
  P.out[plane + at] = in_y;
  P.out[at] = in_y * P.cfl_x + in_x;
  P.out[2 * plane + at] = in_y * P.cfl_b + in_b;
}

struct DcSmoothParams {
  const float* in;   // [3][ys][xs] dequantised DC
  float* out;        // [3][ys][xs]
  uint32_t xs, ys;
  float step[3];     // the DC quantisation step of X, Y, B: the gap is measured in steps (compressed_dc.cc:92-93)
};

// One thread per block. 4K: 480 x 270 blocks, 36 B read (through L2: the nine taps of neighbouring threads overlap) and
// 12 B written per block; the launch is ~3 us of work behind a 6 MB copy.
__global__ __launch_bounds__(256) void k_dc_smooth(DcSmoothParams P) {
  const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= P.xs || y >= P.ys) return;
  const size_t plane = size_t(P.xs) * P.ys, at = size_t(y) * P.xs + x;
  if (x == 0 || y == 0 || x + 1 == P.xs || y + 1 == P.ys) {  // (compressed_dc.cc:141-147, 171-175: borders are copied)
    for (int c = 0; c < 3; c++) P.out[plane * c + at] = P.in[plane * c + at];
    return;
  }
  const float w1 = 0.20345139757231578f, w2 = 0.0334829185968739f, w0 = 1.0f - 4.0f * (w1 + w2);
  float mc[3], sm[3], gap = 0.5f;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float* m = P.in + plane * c + at;
    const float* t = m - P.xs;
    const float* b = m + P.xs;
    const float corner = (t[-1] + t[1]) + (b[-1] + b[1]);
    const float side = (m[-1] + m[1]) + (t[0] + b[0]);
    mc[c] = m[0];
    sm[c] = __builtin_fmaf(corner, w2, __builtin_fmaf(side, w1, mc[c] * w0));
    gap = fmaxf(gap, fabsf((mc[c] - sm[c]) / P.step[c]));
  }
  float factor = __builtin_fmaf(-4.0f, gap, 3.0f);
  factor = factor < 0.0f ? 0.0f : factor;
#pragma unroll
  for (int c = 0; c < 3; c++) P.out[plane * c + at] = __builtin_fmaf(sm[c] - mc[c], factor, mc[c]);
}

struct SigmaParams {
  const JxlHipVarBlock* blocks;
  uint32_t num_blocks, xb;
  const uint8_t* sharpness;  // [yb][xb], 0..7
  float* inv_sigma;          // [yb][xb]
  float quant_scale;         // global_scale / 65536 (quantizer.Scale())
  float epf_quant_mul;
  float sharp_lut[8];
};

// One thread per varblock (its covered blocks in a loop: up to 32 x 32 for the largest transform, rare).
__global__ __launch_bounds__(256) void k_epf_sigma(SigmaParams P) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P.num_blocks) return;
  const JxlHipVarBlock v = P.blocks[i];
  // blocks covered per strategy (ac_strategy.h:130-167; the same table jxlhip_frame_upload validates the varblocks with)
  const uint8_t kCx[27] = {1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32};
  const uint8_t kCy[27] = {1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16};
  const uint32_t st = v.strategy < 27 ? v.strategy : 0, cx = kCx[st], cy = kCy[st];
  const float kInvSigmaNum = -1.1715728752538099024f;
  const float sigma_quant = P.epf_quant_mul / (P.quant_scale * float(v.qf) * kInvSigmaNum);
  for (uint32_t iy = 0; iy < cy; iy++)
    for (uint32_t ix = 0; ix < cx; ix++) {
      const size_t at = size_t(v.by + iy) * P.xb + v.bx + ix;
      float sigma = sigma_quant * P.sharp_lut[P.sharpness[at] & 7];
      sigma = fminf(-1e-4f, sigma);
      P.inv_sigma[at] = 1.0f / sigma;
    }
}

}  // namespace jxlhip
#endif  // JXL_HIP_DC_H_
