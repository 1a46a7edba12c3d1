// Microbenchmark: does VALU work of ANOTHER wave on the same SIMD slow down a wave streaming fp32 MFMAs?
// 512-thread blocks, one per CU: waves 0-3 stream MFMAs, waves 4-7 run a co-runner (mode): 0 none, 1 VALU fma chain,
// 2 LDS reads, 3 global stores, 4 VALU integer/address-like ops.  Prints cycles per MFMA for wave 0 of block 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out, unsigned long long* cyc, float* sink, int mfma_on,
                                         unsigned long long* cyc2) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i * 0.001f;
  __syncthreads();
  if (wave < 4) {
    if (!mfma_on) return;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 0.01f, y = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    float v = threadIdx.x, w = 1.0001f;
    int n = iters * 16;
    // modes 5 / 6: the VALU chain of mode 1 at raised priority (5), or with the MFMA waves being the YOUNGER ones is not
    // expressible here -- instead mode 6 runs the chain in short bursts separated by s_sleep, to see whether a prioritised
    // wave's VALU instructions cut into another wave's fp32 MFMA stream at all
    if (mode == 5 || mode == 6) __builtin_amdgcn_s_setprio(3);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (mode == 1 || mode == 5) {
      for (int i = 0; i < n; ++i) { v = fmaf(v, w, 1.0f); v = fmaf(v, w, 2.0f); v = fmaf(v, w, 3.0f); v = fmaf(v, w, 4.0f); }
    } else if (mode == 2) {
      int idx = threadIdx.x & 1023;
      for (int i = 0; i < n; ++i) { v += lds[idx]; idx = (idx + 64) & 4095; }
    } else if (mode == 3) {
      for (int i = 0; i < n / 8; ++i) sink[(size_t)blockIdx.x * 65536 + ((i * 512 + threadIdx.x) & 65535)] = v;
    } else if (mode == 6) {
      for (int i = 0; i < n / 64; ++i) {
        for (int j = 0; j < 16; ++j) { v = fmaf(v, w, 1.0f); v = fmaf(v, w, 2.0f); v = fmaf(v, w, 3.0f); v = fmaf(v, w, 4.0f); }
        __builtin_amdgcn_s_sleep(8);
      }
    } else if (mode == 4) {
      unsigned u = threadIdx.x;
      for (int i = 0; i < n; ++i) { u = u * 1664525u + 1013904223u; u ^= u >> 7; u += i; u = (u << 3) | (u >> 29); }
      v = (float)u;
    }
    out[blockIdx.x * 512 + threadIdx.x] = v;
    if (threadIdx.x == 256) cyc2[blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
  }
}
int main() {
  float *out, *sink; unsigned long long *cyc, *cyc2;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&cyc2, 256 * 8); hipMalloc(&sink, (size_t)256 * 65536 * 4);
  const int iters = 4000;
  for (int mode = 0; mode <= 6; ++mode) {
    double res[2] = {0, 0}, co[2] = {0, 0};
    for (int on = 1; on >= 0; --on) {
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc, sink, on, cyc2);
        hipDeviceSynchronize();
      }
      unsigned long long h[256], h2[256];
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      hipMemcpy(h2, cyc2, sizeof(h2), hipMemcpyDeviceToHost);
      double s = 0, s2 = 0; for (int i = 0; i < 256; ++i) { s += (double)h[i]; s2 += (double)h2[i]; }
      res[on] = s / 256 / (iters * 4.0); co[on] = s2 / 256;
    }
    printf("mode %d: %.1f cycles per MFMA; co-runner %.0f cycles with MFMA stream, %.0f alone (x%.2f)\n", mode, res[1], co[1], co[0], co[0] > 0 ? co[1] / co[0] : 0.0);
  }
  return 0;
}
