// Microbenchmark: what can ride in the gaps of ONE wave's own fp32 MFMA stream for free?
// One wave per SIMD; per loop iteration 4 MFMAs (32x32x2, 64 cycles each) plus `n` extra instructions of a kind:
// kind 0 VALU fma, 1 ds_read_b128, 2 global store dwordx4, 3 ds_write_b32, 4 global load dwordx4 (+late use).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int N>
__global__ __launch_bounds__(256) void k(int iters, float* out, unsigned long long* cyc, float* sink) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 0.001f;
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 0.01f, y = 1.0f, v = 1.0f;
  f32x4 acc4 = {0, 0, 0, 0};
  float* gp = sink + (size_t)blockIdx.x * 262144 + threadIdx.x * 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (KIND == 0) v = fmaf(v, 1.0001f, 0.5f);
      if (KIND == 1) { f32x4 t = *reinterpret_cast<f32x4*>(lds + ((threadIdx.x * 4 + j * 1024 + i * 4) & 8188)); acc4 += t; }
      if (KIND == 2) *reinterpret_cast<f32x4*>(gp + (((i * N + j) * 1024) & 262143)) = acc4;
      if (KIND == 3) lds[(threadIdx.x + j * 256 + i) & 8191] = v;
      if (KIND == 4) { f32x4 t = *reinterpret_cast<const f32x4*>(gp + (((i * N + j) * 1024) & 262143)); acc4 += t; }
    }
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + v + acc4[0] + acc4[3];
}
template <int KIND, int N>
void run(int iters, float* out, unsigned long long* cyc, float* sink, const char* name) {
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<KIND, N>), dim3(256), dim3(256), 0, 0, iters, out, cyc, sink);
    hipDeviceSynchronize();
  }
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 256; ++i) s += (double)h[i];
  printf("%-22s %2d per 4 MFMAs: %.1f cycles per MFMA\n", name, N, s / 256 / (iters * 4.0));
}
int main() {
  float *out, *sink; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, (size_t)256 * 262144 * 4);
  hipMemset(sink, 0, (size_t)256 * 262144 * 4);
  const int iters = 2000;
  run<0, 0>(iters, out, cyc, sink, "baseline");
  run<0, 8>(iters, out, cyc, sink, "VALU fma");
  run<0, 32>(iters, out, cyc, sink, "VALU fma");
  run<0, 56>(iters, out, cyc, sink, "VALU fma");
  run<1, 2>(iters, out, cyc, sink, "ds_read_b128");
  run<1, 8>(iters, out, cyc, sink, "ds_read_b128");
  run<3, 8>(iters, out, cyc, sink, "ds_write_b32");
  run<2, 1>(iters, out, cyc, sink, "global store x4");
  run<2, 2>(iters, out, cyc, sink, "global store x4");
  run<4, 1>(iters, out, cyc, sink, "global load x4");
  run<4, 2>(iters, out, cyc, sink, "global load x4");
  return 0;
}
