// Cycles per trip of the hand-written lane loop (csrc/hip/jxl_hip_lanes_trip.inc) in isolation: one wave per SIMD, random
// alias tables / context map / stream words in LDS, no transition passes, no refills (measurement aid, GPU box).
// build: hipcc --offload-arch=gfx950 -O3 -I libjxl_amd/csrc/hip -o /tmp/ubench_trip3 scripts/ubench_trip3.hip
// run:   /tmp/ubench_trip3 [groups] [variant]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#include "jxl_hip_lanes_trip.inc"
__global__ __launch_bounds__(64) void k(uint64_t* out, uint8_t* coef, uint32_t groups, uint32_t split_exp, uint32_t live_lanes) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint32_t* l32 = (uint32_t*)lds;
  const uint32_t lane = threadIdx.x;
  // 128 B f2 table | 32 KB alias (64 clusters x 64 entries x 8 B) | 8 KB context map | nnz2 | rings
  uint32_t seed = blockIdx.x * 977 + lane * 13 + 1;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  for (uint32_t i = lane; i < 128; i += 64) lds[i] = uint8_t((i % 30) * 2);
  for (uint32_t i = lane; i < 4096; i += 64) {  // alias entries: x = freq-1 | cfg << 12 | cutoff << 24; y = freq-1 | offs << 12 | right << 24
    const uint32_t cutoff = rnd() % 65, right = (rnd() % 16) ? (rnd() % 3) : (rnd() % 40);
    l32[32 + 2 * i] = (rnd() & 0x3FF) | ((split_exp | 2 << 4 | 0 << 8) << 12) | cutoff << 24;
    l32[32 + 2 * i + 1] = (rnd() & 0x3FF) | ((rnd() & 0xFFF) << 12) | right << 24;
  }
  const uint32_t ctx_off = 128 + 32768, nnz2_off = ctx_off + 8192, ring_off = nnz2_off + 128;
  for (uint32_t i = lane; i < 8192; i += 64) lds[ctx_off + i] = uint8_t(rnd() & 63);
  for (uint32_t i = lane; i < 64; i += 64) ((uint16_t*)(lds + nnz2_off))[i] = uint16_t((i * 7) & 0x1FE);
  for (uint32_t r = 0; r < 18; r++) l32[ring_off / 4 + r * 64 + lane] = rnd() * 2654435761u;
  __syncthreads();
  uint32_t st = 0x130000 + rnd() % 50000, bp = 0, kk = 1, nzl = 1u << 30, ctxe = rnd() & 63, aa = ctx_off + 37 * 15, nb = 62, dst = (blockIdx.x * 64 + lane) * (groups * 8 + 64);
  uint32_t alo = 0, ahi = 0;
  const uint32_t log2c = 0, cb = ctx_off + 37 * 15, covm2 = uint32_t(-2), ring = ring_off + lane * 4, rend = 1u << 30, size = 1 + groups * 4;
  const uint64_t act = live_lanes >= 64 ? ~0ull : ((1ull << live_lanes) - 1);
  const uint32_t cont = 1, le = 6, em = 63, clm = 512, nnz2 = nnz2_off, shift = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  uint32_t grp = 0;
  JXL_LANES_TRIP_LOOP(grp, st, bp, kk, nzl, ctxe, aa, nb, dst, alo, ahi, log2c, cb, covm2, ring, rend, size, act, cont, le, em, clm, nnz2, shift, coef);
  const uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x] = t1 - t0;
  if (st == 0xdeadbeef) out[0] = alo + ahi + kk + bp;
}
int main(int argc, char** argv) {
  const uint32_t groups = argc > 1 ? atoi(argv[1]) : 2000;
  const int per_cu = argc > 2 ? atoi(argv[2]) : 3;  // workgroups (one wave each) per CU: the dynamic LDS request sets it
  const int blocks = 256 * per_cu;
  uint64_t* out;
  uint8_t* coef;
  hipMalloc(&out, blocks * 8);
  const size_t coef_bytes = size_t(blocks) * 64 * (groups * 8 + 64);
  hipMalloc(&coef, coef_bytes);
  const size_t lds_bytes = per_cu <= 1 ? 100 * 1024 : (per_cu == 2 ? 70 * 1024 : (per_cu == 3 ? 45 * 1024 : 42 * 1024 * 3 / per_cu));
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (uint32_t split = 4; split <= 8; split += 4)        // split 4: some tokens carry extra bits; 8: (almost) none
    for (uint32_t lanes = 64; lanes >= 16; lanes /= 2) {
      hipLaunchKernelGGL(k, dim3(blocks), dim3(64), lds_bytes, 0, out, coef, groups, split, lanes);
      if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return 1; }
      uint64_t h[2048];
      hipMemcpy(h, out, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0;
      for (int i = 0; i < blocks; i++) s += double(h[i]);
      printf("%d per CU, split_exp %u lanes %2u: %.1f shader cycles (s_memtime) per trip\n", per_cu, split, lanes, s / blocks / (groups * 4.0));
    }
  return 0;
}
