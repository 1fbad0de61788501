// Issue / dependency latencies of a LONE wave on gfx950 (one wave per SIMD): cycles per instruction for dependent chains and
// for interleaved independent chains of the instruction kinds the lane loop is made of (measurement aid, GPU box).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_lat scripts/ubench_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define BODY(name, text, per)                                                                  \
  if (which == name) {                                                                         \
    const uint64_t t0 = __builtin_readcyclecounter();                                         \
    for (int it = 0; it < iters; it++) asm volatile(R64(text) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(sa), "+s"(sb) : "v"(e), "s"(sm) : "vcc", "scc", "s90", "s91"); \
    const uint64_t t1 = __builtin_readcyclecounter();                                         \
    cyc = double(t1 - t0) / (double(iters) * 64 * per);                                        \
  }
__global__ __launch_bounds__(64) void k(double* out, uint32_t* sink, int which, int iters) {
  extern __shared__ uint32_t lds[];
  uint32_t a = threadIdx.x, b = threadIdx.x * 3, c = 7, d = 9, e = threadIdx.x & 31;
  uint32_t sa = 1, sb = 2;
  const uint32_t sm = 0x05040100u;
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  double cyc = 0;
  BODY(0, "v_add_u32 %0, %0, %6\n\t", 1)                                             // dependent adds
  BODY(1, "v_add_u32 %0, %0, %6\n\tv_add_u32 %1, %1, %6\n\t", 2)                     // two chains
  BODY(2, "v_add_u32 %0, %0, %6\n\tv_add_u32 %1, %1, %6\n\tv_add_u32 %2, %2, %6\n\tv_add_u32 %3, %3, %6\n\t", 4)  // four chains
  BODY(3, "v_mad_u32_u24 %0, %0, %6, %0\n\t", 1)                                      // dependent VOP3
  BODY(4, "v_cmp_lt_u32 vcc, %0, %6\n\tv_cndmask_b32 %0, %0, %1, vcc\n\t", 2)         // compare -> select, dependent
  BODY(5, "v_cmp_lt_u32 vcc, %0, %6\n\tv_add_u32 %2, %2, %6\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_add_u32 %3, %3, %6\n\t", 4)  // same with fillers
  BODY(6, "v_perm_b32 %0, %0, %1, %7\n\t", 1)
  BODY(7, "v_alignbit_b32 %0, %0, %1, %6\n\t", 1)
  BODY(8, "v_bfe_u32 %0, %0, 3, 12\n\t", 1)
  BODY(9, "s_add_u32 %4, %4, %5\n\t", 1)                                              // dependent scalar adds
  BODY(10, "v_add_u32 %0, %0, %6\n\ts_add_u32 %4, %4, %5\n\t", 2)                     // vector + scalar interleaved
  BODY(11, "v_cndmask_b32_sdwa %0, %1, %0, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t", 1)  // dependent SDWA
  BODY(12, "v_cmp_ne_u32_e64 s[90:91], 0, %0\n\tv_cndmask_b32_e64 %0, %0, %1, s[90:91]\n\t", 2)  // compare to SGPR pair -> select
  BODY(13, "ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t", 1)                       // LDS round trip (address = previous result & small)
  BODY(14, "v_and_b32 %0, 0xfff, %0\n\t", 1)                                          // dependent, 32-bit literal
  BODY(15, "v_add3_u32 %0, %0, %1, %2\n\tv_lshl_add_u32 %1, %1, 1, %0\n\t", 2)
  BODY(16, "v_add_u32 %0, %0, %6\n\ts_nop 0\n\t", 1)                                  // dependent adds with a nop between
  BODY(17, "v_add_u32 %0, %0, %6\n\tv_add_u32 %1, %1, %6\n\tv_add_u32 %2, %2, %6\n\t", 3)  // three chains
  BODY(18, "v_cmp_ne_u32 vcc, 0, %0\n\ts_and_b64 s[90:91], vcc, exec\n\tv_add_u32 %0, %0, %6\n\t", 3)  // VALU -> VCC -> SALU
  BODY(19, "v_subbrev_co_u32 %0, vcc, 0, %0, vcc\n\t", 1)
  if (threadIdx.x == 0) out[blockIdx.x] = cyc;
  sink[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + sa + sb;
}
int main() {
  const int blocks = 1024, iters = 50;
  double* out;
  uint32_t* sink;
  hipMalloc(&out, blocks * 8);
  hipMalloc(&sink, blocks * 64 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
  const char* names[20] = {"v_add_u32 dependent", "v_add_u32 two chains", "v_add_u32 four chains", "v_mad_u32_u24 dependent", "v_cmp -> v_cndmask dependent",
                           "v_cmp, filler, v_cndmask, filler", "v_perm_b32 dependent", "v_alignbit_b32 dependent", "v_bfe_u32 dependent", "s_add_u32 dependent",
                           "v_add_u32 + s_add_u32 interleaved", "v_cndmask_b32_sdwa dependent", "v_cmp_e64 sgpr -> v_cndmask_e64", "ds_read_b32 + wait round trip",
                           "v_and_b32 literal dependent", "v_add3 / v_lshl_add cross-dependent", "v_add_u32 dependent + s_nop 0", "v_add_u32 three chains",
                           "v_cmp -> s_and(vcc) -> v_add", "v_subbrev_co_u32 dependent (vcc in/out)"};
  for (int w = 0; w < 20; w++) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 42 * 1024, 0, out, sink, w, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("failed at %d\n", w); return 1; }
    double h[1024], s = 0;
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < blocks; i++) s += h[i];
    printf("%-44s %6.2f cycles per instruction\n", names[w], s / blocks);
  }
  return 0;
}
