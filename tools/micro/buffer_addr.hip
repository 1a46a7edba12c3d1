// Micro check: raw buffer store/load addressing (voffset per lane + scalar soffset) on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* p, int soff) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7FFFFFFF, 0x00020000);
  i32x4 v;
  for (int q = 0; q < 4; ++q) v[q] = threadIdx.x * 4 + q;
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(threadIdx.x * 16), soff, 0);
}
int main() {
  float* d; hipMalloc(&d, 4096 * 4); hipMemset(d, 0xff, 4096 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1024);
  int h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) if (h[256 + i] != i) ++bad;
  printf("expected words 256..511 = 0..255: %d mismatches; h[256..263] = %d %d %d %d %d %d %d %d; h[0]=%d\n", bad, h[256], h[257], h[258], h[259], h[260], h[261], h[262], h[263], h[0]);
  return 0;
}
