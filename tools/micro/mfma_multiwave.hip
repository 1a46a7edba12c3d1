// Microbenchmark: the fp32 matrix pipe shared by SEVERAL waves of one SIMD that all stream v_mfma_f32_32x32x2_f32.
// Blocks of 256 * W threads (W = 1, 2, 3 waves per SIMD), one block per CU.  Every wave issues `iters` groups of 8 MFMAs
// on two alternating accumulators (the conv kernel's pattern), optionally with 3 ds_read_b128 per group (mode 1) and
// with a per-wave static priority (mode 2: wave w of a SIMD gets s_setprio (W-1-w)).  Prints SIMD cycles per MFMA
// (wall cycles of the block / MFMAs issued per SIMD): 64.0 means the pipe is never idle and switching waves is free.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(768) void k(int iters, float* out, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 0.001f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (MODE == 2) {
    const int w = wave >> 2;
    if (w == 0) __builtin_amdgcn_s_setprio(2);
    else if (w == 1) __builtin_amdgcn_s_setprio(1);
  }
  f32x16 a0 = {0}, a1 = {0};
  f32x4 x = {1.f, 2.f, 3.f, 4.f}, y = {0.5f, 0.25f, 0.125f, 1.f}, b = {1.f, 1.f, 1.f, 1.f};
  const float* p = lds + lane * 20 + wave * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 1) {
      x = *reinterpret_cast<const f32x4*>(p + ((i * 4) & 1023));
      y = *reinterpret_cast<const f32x4*>(p + ((i * 4 + 1280) & 2047));
      b = *reinterpret_cast<const f32x4*>(p + ((i * 4 + 2560) & 4095));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], x[j], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], y[j], a1, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = t2 - t0; }
  out[blockIdx.x * 768 + threadIdx.x] = a0[0] + a1[1];
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  for (int W = 1; W <= 3; ++W) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * W), 0, 0, iters, out, cyc);
      hipDeviceSynchronize();
    }
    unsigned long long h[512];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s0 = 0, s1 = 0;
    for (int i = 0; i < 256; ++i) { s0 += (double)h[2 * i]; s1 += (double)h[2 * i + 1]; }
    printf("%-28s %d wave(s)/SIMD: first wave %.1f cycles per own MFMA; block %.2f SIMD cycles per MFMA\n", name, W,
           s0 / 256 / (iters * 8.0), s1 / 256 / (iters * 8.0 * W));
  }
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 768 * 4); hipMalloc(&cyc, 512 * 8);
  run<0>("bare MFMA streams", out, cyc);
  run<1>("+3 ds_read_b128 per 8 MFMAs", out, cyc);
  run<2>("bare, static priority", out, cyc);
  return 0;
}
