// Issue rate of packed f32 against plain f32 vector instructions on gfx950, by waves per SIMD (measurement aid).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_pk scripts/ubench_pk.hip ; run: /tmp/ubench_pk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters) {
  f2 a[8];
  for (int i = 0; i < 8; i++) a[i] = f2{float(threadIdx.x + i), float(i)};
  const f2 m = f2{1.0001f, 0.9999f}, c = f2{0.5f, 0.25f};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) {  // packed fma: 2 flops x 2 per lane
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      } else if (MODE == 1) {  // two plain fmas
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(m.y), "v"(c.y));
      } else if (MODE == 2) {  // packed add
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      } else {  // packed mul
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 4 * 1024 * 1024 * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  const char* names[4] = {"v_pk_fma_f32 x8", "v_fma_f32 x16", "v_pk_add_f32 x8", "v_pk_mul_f32 x8"};
  for (int wps = 1; wps <= 8; wps *= 2) {      // waves per SIMD: workgroups of 256 threads (one wave per SIMD of a CU) x wps per CU
    for (int mode = 0; mode < 4; mode++) {
      const int blocks = 256 * wps;
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters);
      };
      launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_wave = double(iters) * (mode == 1 ? 16 : 8);
      // cycles per instruction per SIMD at 2.4 GHz, if the waves of a SIMD share it evenly
      printf("waves/SIMD %d  %-18s %8.3f ms  %.2f cycles per wave-instruction per SIMD (2.4 GHz), %.1f TFLOP/s\n", wps, names[mode], ms,
             ms * 1e-3 * 2.4e9 / (instr_per_wave * wps), (mode <= 1 ? 4.0 : 2.0) * 8 * iters * 64.0 * 4 * blocks / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
