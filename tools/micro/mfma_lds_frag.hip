// Microbenchmark: the weight-gradient k-step (9 fp32 MFMAs + fragment reads from LDS, one wave per SIMD) in isolation.
// MODE 0: no reads, 1: 10 ds_read_b32 per k-step (counted lgkmcnt), 2: 10 ds_read2st64_b32 per PAIR of k-steps,
// 3: as 1 but all reads of a k-step issued before the MFMAs of the previous one (no interleave difference; control).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(int iters, float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i * 0.001f;
  __syncthreads();
  f32x16 acc[9];
  for (int t = 0; t < 9; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(lds) + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 8192;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {
    float a = 1.f, b = 2.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int kk = 0; kk < 32; ++kk)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
  } else if (MODE == 1) {
    float afr[2][9], bfr[2];
    for (int i = 0; i < iters; ++i) {
      auto load = [&](int kk, float* av, float& bv) {
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bv) : "v"(base), "n"(256 * (kk % 16)));
#pragma unroll
        for (int t = 0; t < 9; ++t) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(av[t]) : "v"(base), "n"(256 * (kk % 16) + 128 * t));
      };
      load(0, afr[0], bfr[0]);
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) {
        if (kk + 1 < 32) load(kk + 1, afr[(kk + 1) & 1], bfr[(kk + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 1 < 32) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[kk & 1][t], bfr[kk & 1], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    f32x2 afr[2][9], bfr[2];
    for (int i = 0; i < iters; ++i) {
      auto load = [&](int kp, f32x2* av, f32x2& bv) {
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(bv) : "v"(base), "n"(2 * (kp % 8)), "n"(2 * (kp % 8) + 1));
#pragma unroll
        for (int t = 0; t < 9; ++t) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(av[t]) : "v"(base), "n"(2 * (kp % 8) + t), "n"(2 * (kp % 8) + t + 1));
      };
      load(0, afr[0], bfr[0]);
#pragma unroll
      for (int kp = 0; kp < 16; ++kp) {
        if (kp + 1 < 16) load(kp + 1, afr[(kp + 1) & 1], bfr[(kp + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (kp + 1 < 16) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[kp & 1][t].x, bfr[kp & 1].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[kp & 1][t].y, bfr[kp & 1].y, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float s = 0.f;
  for (int t = 0; t < 9; ++t) s += acc[t][t];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(int iters, float* out, unsigned long long* cyc, const char* name) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 65536, 0, iters, out, cyc);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 256; ++i) s += (double)h[i];
  const double mf = iters * 288.0;
  printf("%-34s s_memtime ticks per MFMA %.2f;  wall %.3f ms -> %.1f TF/s\n", name, s / 256 / mf, ms, 256 * 4 * mf * 4096.0 / ms / 1e9);
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 400;
  run<0>(iters, out, cyc, "9 MFMA, no reads");
  run<1>(iters, out, cyc, "9 MFMA + 10 ds_read_b32");
  run<2>(iters, out, cyc, "18 MFMA + 10 ds_read2st64_b32");
  return 0;
}
