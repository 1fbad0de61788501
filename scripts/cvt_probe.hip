#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n) {
  int i = threadIdx.x;
  if (i < n) {
    unsigned r;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(r) : "v"(in[i]));
    out[i] = r;
  }
}
int main() {
  float h[16] = {0.0f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 3.5f, 254.4f, 254.5f, 254.6f, 255.4f, 255.6f, 300.0f, -0.4f, -3.0f, 127.5f};
  float* d; unsigned* o; unsigned ho[16];
  hipMalloc(&d, 64); hipMalloc(&o, 64);
  hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 16);
  hipMemcpy(ho, o, 64, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; i++) printf("%g -> %u\n", h[i], ho[i]);
  return 0;
}
