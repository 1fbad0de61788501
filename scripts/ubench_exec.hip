// Does a vector instruction cost less VALU time when whole 16-lane quarters of the wave are inactive (gfx950)? Throughput of
// dependent-free v_add_u32 / v_fma_f32 chains at 8 waves per SIMD under several EXEC masks (measurement aid, GPU box).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_exec scripts/ubench_exec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
__global__ __launch_bounds__(256) void k(uint32_t* out, uint64_t mask, int iters, int kind) {
  uint32_t a = threadIdx.x, b = 1, c = 2, d = 3;
  float fa = threadIdx.x, fb = 1.5f, fc = 0.5f, fd = 2.0f;
  const uint64_t full = __builtin_amdgcn_read_exec();
  asm volatile("s_mov_b64 exec, %0" : : "s"(mask));
  for (int it = 0; it < iters; it++) {
    if (kind == 0) asm volatile(R16("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(7u));
    else asm volatile(R16("v_fma_f32 %0, %0, %4, %4\n\tv_fma_f32 %1, %1, %4, %4\n\tv_fma_f32 %2, %2, %4, %4\n\tv_fma_f32 %3, %3, %4, %4\n\t") : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(1.0001f));
  }
  asm volatile("s_mov_b64 exec, %0" : : "s"(full));
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + uint32_t(fa + fb + fc + fd);
}
int main() {
  const int blocks = 256 * 8, iters = 2000;  // 8 waves per SIMD
  uint32_t* out;
  hipMalloc(&out, size_t(blocks) * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const uint64_t masks[6] = {~0ull, 0xFFFFFFFFull, 0xFFFFull, 0x1111111111111111ull, 0xFFFF0000FFFF0000ull, 0x1ull};
  const char* names[6] = {"all 64 lanes", "lanes 0-31", "lanes 0-15", "every 4th lane (16 lanes)", "quarters 1 and 3", "one lane"};
  for (int kind = 0; kind < 2; kind++)
    for (int m = 0; m < 6; m++) {
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, masks[m], iters, kind);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, masks[m], iters, kind);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%s %-28s %.3f ms: %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", kind ? "v_fma_f32" : "v_add_u32", names[m], ms,
             ms * 1e-3 * 2.4e9 / (double(iters) * 64 * 8));
    }
  return 0;
}
