// Microbenchmark: cycles per fp32 MFMA for ONE wave per SIMD as a function of the number of independent accumulator
// chains (a dependent v_mfma_f32_32x32x2_f32 / 16x16x4 cannot issue until the previous result on that accumulator is back).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH, int MF>
__global__ __launch_bounds__(256) void k(int iters, float* out, unsigned long long* cyc) {
  f32x16 a[8];
  f32x4 b[8];
  for (int c = 0; c < 8; ++c) { a[c] = {0}; b[c] = {0}; }
  float x = threadIdx.x * 0.01f, y = 1.0f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 8 / CH; ++rep)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (MF == 32) a[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[c], 0, 0, 0);
        else b[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b[c], 0, 0, 0);
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float s = 0;
  for (int c = 0; c < CH; ++c) s += a[c][0] + b[c][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH, int MF>
void run(int iters, float* out, unsigned long long* cyc) {
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<CH, MF>), dim3(256), dim3(256), 0, 0, iters, out, cyc);
    hipDeviceSynchronize();
  }
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 256; ++i) s += (double)h[i];
  printf("MFMA %dx%d, %d chain(s): %.1f cycles per MFMA\n", MF, MF, CH, s / 256 / (iters * 8.0));
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  run<1, 32>(2000, out, cyc); run<2, 32>(2000, out, cyc); run<4, 32>(2000, out, cyc); run<8, 32>(2000, out, cyc);
  run<1, 16>(2000, out, cyc); run<2, 16>(2000, out, cyc); run<4, 16>(2000, out, cyc); run<8, 16>(2000, out, cyc);
  return 0;
}
