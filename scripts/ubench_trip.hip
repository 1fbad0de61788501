// Measurement aid (not product code): cost of the entropy lane kernel's hot trip on gfx950 as a function of
//   CH     independent decode chains interleaved per lane (1 = the shipped kernel's shape)
//   waves  per SIMD (set through the LDS a workgroup claims)
// The loop body is the shipped hot trip (LaneSymbol<true> + next-context speculation + coefficient store) run on random
// tables and a random, never refilled stream ring: decoded values are garbage, the instruction and LDS-access mix is real.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude scripts/ubench_trip.hip -o libjxl_amd/_build/ubench_trip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../libjxl_amd/csrc/hip/jxl_hip_kernels.h"
#include "../libjxl_amd/csrc/hip/jxl_hip_entropy_lanes.h"

using namespace jxlhip;

constexpr uint32_t kAliasBytes = 64 * 32 * 8;  // 64 clusters, log_alpha 5
constexpr uint32_t kCtxBytes = 7424 + 16;
constexpr uint32_t kTables = kAliasBytes + kCtxBytes + 128 + 512;


// Copy of LaneSymbol<true> (jxl_hip_entropy_lanes.h) with measurement switches:
//   MODE bit 0: no coefficient store; bit 1: alias / cfg reads at lane-linear (conflict-free) addresses;
//   bit 2: ring reads replaced by register values; bit 3: context-entry reads at lane-linear addresses
template <int MODE>
__device__ __forceinline__ uint32_t SymbolM(uint32_t cluster, uint32_t& state, uint32_t& bitpos, const uint32_t* ring, uint32_t LS,
                                            uint32_t log_ls, const uint8_t* lds, const uint16_t* l_cfg, uint32_t log_entry, uint32_t lane) {
  const uint32_t ctxe = (MODE & 2) ? l_cfg[lane] : l_cfg[cluster];
  const uint32_t s0 = ((bitpos >> 5) & (kLanesRingWords - 1)) << log_ls;
  typedef const volatile __attribute__((address_space(3))) uint32_t* LdsVolatile;
  uint32_t w0, w1, w2;
  if (MODE & 4) {
    w0 = state * 2654435761u;
    w1 = w0 ^ bitpos;
    w2 = w1 + 12345;
  } else {
    w0 = *(LdsVolatile)(ring + s0);
    w1 = *(LdsVolatile)(ring + s0 + LS);
    w2 = *(LdsVolatile)(ring + s0 + 2 * LS);
  }
  const uint32_t boff = bitpos & 31;
  const uint32_t res = state & 0xFFFu, slot = res >> log_entry, pos = res & ((1u << log_entry) - 1);
  const uint2 e = (MODE & 2) ? *reinterpret_cast<const uint2*>(lds + lane * 8 + ((cluster + slot) & 1) * 512)
                             : *reinterpret_cast<const uint2*>(lds + (cluster << (15 - log_entry)) + slot * 8);
  const bool gt = pos >= (e.x >> 24);
  const uint32_t x = gt ? e.y : e.x;
  uint32_t tok = gt ? (x >> 24) : slot;
  const uint32_t hi = state >> 12;
  state = (x & 0xFFFu) * hi + hi + ((x >> 12) & 0xFFFu) + pos;
  const uint32_t win = __builtin_amdgcn_alignbit(w1, w0, boff);
  const bool need = state < (1u << 16);
  const uint32_t sh = need ? 16u : 0u;
  state = (state << sh) | (need ? (win & 0xFFFFu) : 0u);
  const uint32_t boff2 = boff + sh;
  bitpos += sh;
  const uint32_t se = ctxe & 15;
  const bool take = tok >= (1u << se);
  const uint32_t msb = (ctxe >> 4) & 15, lsb = (ctxe >> 8) & 15;
  const uint32_t nb = (se - (msb + lsb) + ((tok - (1u << se)) >> (msb + lsb))) & 31u;
  const uint32_t low = tok & ((1u << lsb) - 1), top = tok >> lsb;
  const bool up = boff2 >= 32;
  const uint32_t xb = __builtin_amdgcn_alignbit(up ? w2 : w1, up ? w1 : w0, boff2 & 31) & ((1u << nb) - 1);
  const uint32_t big = (((((1u << msb) | (top & ((1u << msb) - 1))) << nb) | xb) << lsb) | low;
  bitpos += take ? nb : 0u;
  return take ? big : tok;
}

template <int CH, int WPG, int MODE, int UNROLL = 1>
__global__ __launch_bounds__(64 * WPG) void k_trip(const uint32_t* tables, uint16_t* out, uint32_t iters, unsigned long long* cycles) {
  extern __shared__ __align__(16) uint8_t lds_raw[];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < kTables / 4; i += 64 * WPG) reinterpret_cast<uint32_t*>(lds_raw)[i] = tables[i];
  uint32_t* ring_base = reinterpret_cast<uint32_t*>(lds_raw + kTables) + wave * CH * 18 * 64;
  for (uint32_t i = lane; i < CH * 18 * 64; i += 64) ring_base[i] = tables[(i * 7 + blockIdx.x) % (kTables / 4)] * 2654435761u;
  __syncthreads();
  const uint32_t L_ctx = kAliasBytes, L_ctx2 = kAliasBytes + kCtxBytes, L_cfg = L_ctx2 + 128;
  const uint16_t* l_cfg = reinterpret_cast<const uint16_t*>(lds_raw + L_cfg);
  const uint16_t* l_nnz2 = reinterpret_cast<const uint16_t*>(lds_raw + L_ctx2);
  const uint32_t LS = 64, log_ls = 6, log_entry = 7;
  uint32_t state[CH], bitpos[CH], ctxe[CH], addr_a[CH], nnz_b[CH], nzeros[CH], k[CH], dptr[CH];
  const uint32_t* ring[CH];
  const uint32_t log2c = 0, covm1 = 0, cbase = L_ctx + 37 * 15;
#pragma unroll
  for (int c = 0; c < CH; c++) {
    state[c] = 0x130000u + lane * 977 + c * 31;
    bitpos[c] = lane & 31;
    ctxe[c] = (lane + c) & 63;
    addr_a[c] = cbase + 2 * 31;
    nnz_b[c] = 62;
    nzeros[c] = 20 + (lane & 7);
    k[c] = 1;
    dptr[c] = ((blockIdx.x * WPG + wave) * 64 + lane) * 4096 * CH + c * 4096;
    ring[c] = ring_base + c * 18 * 64 + lane;
  }
  if (MODE & 16) {  // desynchronise the waves of a CU: a different delay per workgroup before the loop
    const uint32_t spin = (blockIdx.x * 2654435761u >> 20) & 4095;
    for (uint32_t i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(1);
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll UNROLL
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const uint32_t kn = k[c] + 1;
      const uint32_t b = (kn >> log2c) & 63;
      const uint32_t f2 = min(min(b - 1, 7 + (b >> 1)), 15 + (b >> 2)) << 1;
      const uint32_t addr_b = cbase + 1 + nnz_b[c];
      const uint32_t e_zero = (MODE & 8) ? lds_raw[L_ctx + lane * 4 + (f2 & 2)] : lds_raw[addr_a[c] + f2];
      const uint32_t e_nonzero = (MODE & 8) ? lds_raw[L_ctx + 256 + lane * 4 + (f2 & 2)] : lds_raw[addr_b + f2];
      const uint32_t nnz_c = (MODE & 8) ? l_nnz2[lane] : l_nnz2[((nzeros[c] - 2 + covm1) >> log2c) & 63];
      const uint32_t tok = SymbolM<MODE>(ctxe[c] & 63, state[c], bitpos[c], ring[c], LS, log_ls, lds_raw, l_cfg, log_entry, lane);
      const uint32_t sgn = uint32_t(-int32_t(tok & 1));
      const int32_t coeff = int32_t((tok >> 1) ^ sgn);
      if (!(MODE & 1)) out[dptr[c] & ~0u] = uint16_t(coeff);
      else if (coeff == 0x7FFFFFF) out[0] = 1;
      dptr[c] = (dptr[c] & ~4095u) | ((dptr[c] + 1) & 4095u);
      k[c] = kn & 63;
      const bool nz = tok != 0;
      nzeros[c] -= nz ? 1u : 0u;
      ctxe[c] = nz ? e_nonzero : e_zero;
      addr_a[c] = nz ? addr_b - 1 : addr_a[c];
      nnz_b[c] = nz ? nnz_c : nnz_b[c];
      nzeros[c] = nzeros[c] < 3 ? 40u : nzeros[c];
      state[c] |= 0x10000u;  // keep the garbage state in range
      state[c] &= 0x7FFFFFFFu;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cycles[blockIdx.x * WPG + wave] = t1 - t0;
  if (state[0] == 0x12345u) out[0] = uint16_t(bitpos[CH - 1]);
}

template <int CH, int WPG, int MODE = 0, int UNROLL = 1>
static void Run(int wgs_per_cu, uint32_t iters, const uint32_t* d_tables, uint16_t* d_out, unsigned long long* d_cyc) {
  const size_t lds_min = kTables + size_t(WPG) * CH * 18 * 64 * 4;
  size_t lds = (160 * 1024 / wgs_per_cu) & ~size_t(255);  // a CU takes floor(160 KiB / lds) workgroups
  if (lds < lds_min) {
    printf("CH=%d WPG=%d wgs/CU=%d: needs %zu B of LDS per workgroup, only %zu available: skipped\n", CH, WPG, wgs_per_cu, lds_min, lds);
    return;
  }
  auto kern = k_trip<CH, WPG, MODE, UNROLL>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
  const int grid = 256 * wgs_per_cu, waves = grid * WPG;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPG), lds, 0, d_tables, d_out, 64u, d_cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPG), lds, 0, d_tables, d_out, iters, d_cyc);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> cyc(waves);
  hipMemcpy(cyc.data(), d_cyc, waves * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : cyc) mean += double(v);
  mean /= waves;
  const double tokens = double(waves) * 64 * CH * iters;
  printf("UNROLL=%d MODE=%d CH=%d WPG=%d wgs/CU=%d (%.2f waves/SIMD) lds/wg=%zu: %.3f ms, %.0f cycles/iteration/wave (%.0f per token-trip), %.1f Gtokens/s, err=%s\n", UNROLL, MODE, CH, WPG,
         wgs_per_cu, wgs_per_cu * WPG / 4.0, lds, ms, mean / iters, mean / iters / CH, tokens / ms * 1e-6, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
  const uint32_t iters = argc > 1 ? uint32_t(atoi(argv[1])) : 20000;
  std::vector<uint32_t> t(kTables / 4);
  uint32_t s = 12345;
  for (auto& v : t) {
    s = s * 1664525u + 1013904223u;
    v = s;
  }
  // context map bytes: cluster ids < 64; uint configs: split 4, msb 2, lsb 0 (4 | 2 << 4)
  uint8_t* bytes = reinterpret_cast<uint8_t*>(t.data());
  for (uint32_t i = 0; i < kCtxBytes; i++) bytes[kAliasBytes + i] &= 63;
  for (uint32_t i = 0; i < 64; i++) reinterpret_cast<uint16_t*>(bytes + kAliasBytes + kCtxBytes)[i] = uint16_t((i * 31 / 10) & ~1u);
  for (uint32_t i = 0; i < 256; i++) reinterpret_cast<uint16_t*>(bytes + kAliasBytes + kCtxBytes + 128)[i] = 4 | (2 << 4);
  uint32_t* d_tables;
  uint16_t* d_out;
  unsigned long long* d_cyc;
  hipMalloc(reinterpret_cast<void**>(&d_tables), kTables);
  hipMemcpy(d_tables, t.data(), kTables, hipMemcpyHostToDevice);
  hipMalloc(reinterpret_cast<void**>(&d_out), size_t(4096) * 64 * 4096 * 4 * 2 + 4096);
  hipMalloc(reinterpret_cast<void**>(&d_cyc), 65536 * 8);
  const int which = argc > 2 ? atoi(argv[2]) : 0;
  if (which == 2) {  // code footprint: the same trip unrolled 1 / 4 / 16 / 64 times (no stores)
    Run<1, 1, 1, 1>(4, iters, d_tables, d_out, d_cyc);
    Run<1, 1, 1, 4>(4, iters, d_tables, d_out, d_cyc);
    Run<1, 1, 1, 16>(4, iters, d_tables, d_out, d_cyc);
    Run<1, 1, 1, 64>(4, iters, d_tables, d_out, d_cyc);
    Run<1, 1, 1, 64>(2, iters, d_tables, d_out, d_cyc);
    Run<1, 1, 1, 64>(1, iters, d_tables, d_out, d_cyc);
    // the same with the waves of a CU at different places of the loop (MODE bit 4)
    Run<1, 1, 17, 1>(4, iters, d_tables, d_out, d_cyc);
    Run<1, 1, 17, 16>(4, iters, d_tables, d_out, d_cyc);
    Run<1, 1, 17, 64>(4, iters, d_tables, d_out, d_cyc);
    Run<1, 1, 17, 64>(3, iters, d_tables, d_out, d_cyc);
    Run<1, 1, 17, 64>(1, iters, d_tables, d_out, d_cyc);
  } else if (which == 0) {
    // one wave per workgroup (a frame's tables per wave, as shipped): 4 per CU = one wave per SIMD
    Run<1, 1>(4, iters, d_tables, d_out, d_cyc);
    Run<2, 1>(4, iters, d_tables, d_out, d_cyc);
    Run<3, 1>(4, iters, d_tables, d_out, d_cyc);
    Run<1, 1>(2, iters, d_tables, d_out, d_cyc);  // half the SIMDs occupied
    Run<2, 1>(2, iters, d_tables, d_out, d_cyc);
    // tables shared by the waves of a workgroup (a frame with many sections): 1, 2, 3, 4 waves per SIMD
    Run<1, 4>(1, iters, d_tables, d_out, d_cyc);
    Run<1, 8>(1, iters, d_tables, d_out, d_cyc);
    Run<1, 12>(1, iters, d_tables, d_out, d_cyc);
    Run<1, 16>(1, iters, d_tables, d_out, d_cyc);
    Run<2, 8>(1, iters, d_tables, d_out, d_cyc);
    Run<2, 4>(2, iters, d_tables, d_out, d_cyc);
  } else {
#define VARIANT(M)                                  \
  Run<1, 1, M>(1, iters, d_tables, d_out, d_cyc);   \
  Run<1, 1, M>(2, iters, d_tables, d_out, d_cyc);   \
  Run<1, 1, M>(4, iters, d_tables, d_out, d_cyc);   \
  Run<2, 1, M>(1, iters, d_tables, d_out, d_cyc);   \
  Run<2, 1, M>(4, iters, d_tables, d_out, d_cyc);
    VARIANT(0) VARIANT(1) VARIANT(2) VARIANT(3) VARIANT(4) VARIANT(8) VARIANT(10) VARIANT(14) VARIANT(15)
  }
  return 0;
}
