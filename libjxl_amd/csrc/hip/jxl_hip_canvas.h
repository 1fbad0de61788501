// The canvas of a multi-frame codestream on the device: frames are blended, in the output colour space, with a
// reference slot (lib/jxl/blending.cc:40-190 PerformBlending, lib/jxl/alpha.cc:20-88, the placement rules of
// render_pipeline/stage_blending.cc:140-235: outside the frame's rectangle the background shows, an unwritten slot is
// zeros), and what the caller gets is the canvas (dec_cache.cc:268-290). Planes are f32 [4][ysize][xsize]: R, G, B, alpha.
#ifndef JXL_HIP_CANVAS_H_
#define JXL_HIP_CANVAS_H_

namespace jxlhip {

struct BlendParams {
  float* out;             // [4][h][w]
  const float* bg_color;  // slot of the colour channels, or NULL (zeros)
  const float* bg_alpha;  // slot of the alpha channel, or NULL
  const float* fg;        // the frame: interleaved f32 x 4 (R, G, B, alpha), fw x fh
  uint32_t w, h, fw, fh;
  int32_t x0, y0;
  uint32_t mode, alpha_mode, clamp, alpha_clamp, has_alpha, premultiplied;
};

__device__ __forceinline__ float Clamp01(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }

__global__ __launch_bounds__(256) void k_canvas_blend(BlendParams P) {
#pragma clang fp contract(off)
  const uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= P.w) return;
  const size_t plane = size_t(P.w) * P.h, i = size_t(y) * P.w + x;
  float bg[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (P.bg_color)
    for (int c = 0; c < 3; c++) bg[c] = P.bg_color[c * plane + i];
  if (P.bg_alpha) bg[3] = P.bg_alpha[3 * plane + i];
  const int fx = int(x) - P.x0, fy = int(y) - P.y0;
  float o[4];
  if (fx < 0 || fy < 0 || fx >= int(P.fw) || fy >= int(P.fh)) {
    for (int c = 0; c < 4; c++) o[c] = bg[c];
  } else {
    const float4 f4 = reinterpret_cast<const float4*>(P.fg)[size_t(fy) * P.fw + fx];
    const float fg[4] = {f4.x, f4.y, f4.z, P.has_alpha ? f4.w : 1.0f};
    if (P.has_alpha) {  // the alpha channel first, from the alpha values before blending
      const float fa = P.alpha_clamp ? Clamp01(fg[3]) : fg[3];
      switch (P.alpha_mode) {
        case 1: o[3] = bg[3] + fg[3]; break;
        case 2: o[3] = 1.0f - (1.0f - fa) * (1.0f - bg[3]); break;
        case 3: o[3] = bg[3]; break;
        case 4: o[3] = bg[3] * fa; break;
        default: o[3] = fg[3]; break;
      }
    } else {
      o[3] = 1.0f;
    }
    const float fa = P.clamp ? Clamp01(fg[3]) : fg[3];
    switch (P.mode) {
      case 1:
        for (int c = 0; c < 3; c++) o[c] = bg[c] + fg[c];
        break;
      case 3:
        for (int c = 0; c < 3; c++) o[c] = P.has_alpha ? bg[c] + fg[c] * fa : bg[c] + fg[c];
        break;
      case 2:
        if (!P.has_alpha) {
          for (int c = 0; c < 3; c++) o[c] = fg[c];
        } else if (P.premultiplied) {
          for (int c = 0; c < 3; c++) o[c] = fg[c] + bg[c] * (1.0f - fa);
          o[3] = 1.0f - (1.0f - fa) * (1.0f - bg[3]);
        } else {
          const float new_a = 1.0f - (1.0f - fa) * (1.0f - bg[3]);
          const float rnew_a = new_a > 0 ? 1.0f / new_a : 0.0f;
          for (int c = 0; c < 3; c++) o[c] = (fg[c] * fa + bg[c] * bg[3] * (1.0f - fa)) * rnew_a;
          o[3] = new_a;
        }
        break;
      case 4:
        for (int c = 0; c < 3; c++) o[c] = bg[c] * (P.clamp ? Clamp01(fg[c]) : fg[c]);
        break;
      default:
        for (int c = 0; c < 3; c++) o[c] = fg[c];
        break;
    }
  }
  for (int c = 0; c < 4; c++) P.out[c * plane + i] = o[c];
}

// Three planes of w x h samples out of planes with a row stride (a frame's XYB planes into a reference slot).
// An integer plane (a Modular frame's alpha channel) as floats in [0, 1]: sample * scale (dec_modular.cc:640-700).
__global__ __launch_bounds__(256) void k_int_plane_to_float(const int32_t* __restrict__ src, float* __restrict__ dst, size_t n, float scale) {
  const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) dst[i] = float(src[i]) * scale;
}

__global__ __launch_bounds__(256) void k_copy_planes(const float* __restrict__ src, size_t src_stride, size_t src_plane, float* __restrict__ dst,
                                                     uint32_t w, uint32_t h) {
  const uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, c = blockIdx.z;
  if (x < w) dst[(size_t(c) * h + y) * w + x] = src[c * src_plane + size_t(y) * src_stride + x];
}

// The canvas in the caller's sample format and orientation (StorePixel: stage_write.cc's conversions and dither).
__global__ __launch_bounds__(256) void k_canvas_out(const float* __restrict__ canvas, PixelOut po) {
  const uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= po.xsize) return;
  const size_t plane = size_t(po.xsize) * po.ysize, i = size_t(y) * po.xsize + x;
  StorePixel(po, int(x), int(y), canvas[i], canvas[plane + i], canvas[2 * plane + i]);
}

}  // namespace jxlhip
#endif  // JXL_HIP_CANVAS_H_
