// Issue cost of DPP forms on gfx950, by waves per SIMD (measurement aid): plain v_add_f32 against v_add_f32 with a row shift,
// with a wave shift (wave_shr:1 / wave_shl:1: what the row-streaming filter uses for its neighbour columns), v_mov_b32_dpp
// and v_fmac_f32_dpp with wave shifts, and the transcendental v_exp_f32 / v_log_f32 / v_rcp_f32 the filter's colour stage uses.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_dpp scripts/ubench_dpp.hip ; run: /tmp/ubench_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float* out, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; i++) {
    a[i] = float(threadIdx.x + i);
    b[i] = float(i) * 0.001f + 1.0f;
  }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
      else if (MODE == 1) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b[i]));
      else if (MODE == 2) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b[i]));
      else if (MODE == 3) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b[i]));
      else if (MODE == 4) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b[i]));
      else if (MODE == 5) asm volatile("v_fmac_f32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b[i]));
      else if (MODE == 6) asm volatile("v_exp_f32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
      else if (MODE == 7) asm volatile("v_log_f32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
      else if (MODE == 8) asm volatile("v_rcp_f32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
      else if (MODE == 9) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b[i]));
      else asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(a[i]) : "v"(b[i]));
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static float run(float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float* out;
  hipMalloc(&out, 4 * 1024 * 1024 * 16);
  const int iters = 20000;
  const char* names[11] = {"v_add_f32", "v_add_f32_dpp row_shr:1", "v_add_f32_dpp wave_shr:1", "v_add_f32_dpp wave_shl:1", "v_mov_b32_dpp wave_shr:1",
                           "v_fmac_f32_dpp wave_shr:1", "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_fma_f32", "v_cvt_pk_u8_f32"};
  for (int wps = 1; wps <= 8; wps *= 2) {
    const int blocks = 256 * wps;
    float ms[11] = {run<0>(out, blocks, iters), run<1>(out, blocks, iters), run<2>(out, blocks, iters), run<3>(out, blocks, iters),
                    run<4>(out, blocks, iters), run<5>(out, blocks, iters), run<6>(out, blocks, iters), run<7>(out, blocks, iters),
                    run<8>(out, blocks, iters), run<9>(out, blocks, iters), run<10>(out, blocks, iters)};
    for (int m = 0; m < 11; m++)
      printf("waves/SIMD %d  %-28s %8.3f ms  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", wps, names[m], ms[m],
             ms[m] * 1e-3 * 2.4e9 / (double(iters) * 8 * wps));
  }
  return 0;
}
