// Do 8-byte / 4-byte global stores and 8-byte loads at 2-byte alignment work on this device and runtime (measurement aid)?
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_unaligned scripts/ubench_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned char* buf, unsigned long long* out) {
  const unsigned off = 2 + threadIdx.x * 34;  // 2-byte aligned, every residue mod 64 over the lanes
  const u2 v = {0x11110000u + threadIdx.x, 0x22220000u + threadIdx.x};
  asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(off), "v"(v), "s"(buf) : "memory");
  asm volatile("global_store_dword %0, %1, %2" : : "v"(off + 8), "v"(0x33330000u + threadIdx.x), "s"(buf) : "memory");
  asm volatile("global_store_short %0, %1, %2" : : "v"(off + 12), "v"(0x4400u + threadIdx.x), "s"(buf) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __threadfence();
  u2 r;
  asm volatile("global_load_dwordx2 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(off), "s"(buf) : "memory");
  out[threadIdx.x] = (unsigned long long)r.x | ((unsigned long long)r.y << 32);
}
int main() {
  unsigned char* buf;
  unsigned long long* out;
  hipMalloc(&buf, 4096);
  hipMalloc(&out, 64 * 8);
  hipMemset(buf, 0xEE, 4096);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, buf, out);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  std::vector<unsigned char> h(4096);
  std::vector<unsigned long long> o(64);
  hipMemcpy(h.data(), buf, 4096, hipMemcpyDeviceToHost);
  hipMemcpy(o.data(), out, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (unsigned t = 0; t < 64; t++) {
    const unsigned off = 2 + t * 34;
    unsigned a, b, c;
    unsigned short d;
    memcpy(&a, &h[off], 4); memcpy(&b, &h[off + 4], 4); memcpy(&c, &h[off + 8], 4); memcpy(&d, &h[off + 12], 2);
    const bool ok = a == 0x11110000u + t && b == 0x22220000u + t && c == 0x33330000u + t && d == 0x4400u + t && h[off + 14] == 0xEE && h[off - 1] == 0xEE &&
                    o[t] == ((unsigned long long)(0x22220000u + t) << 32 | (0x11110000u + t));
    if (!ok) { bad++; printf("lane %u off %u: %08x %08x %08x %04x load %016llx\n", t, off, a, b, c, d, o[t]); }
  }
  printf("unaligned (2-byte) dwordx2 / dword / short stores and dwordx2 loads: %s (%d bad lanes)\n", bad ? "BROKEN" : "ok", bad);
  return bad != 0;
}
