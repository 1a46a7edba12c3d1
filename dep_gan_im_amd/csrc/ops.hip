// HBM-bound operators of the DEP-GAN step (see ops.h).  All 16-byte vectorised
// over the contiguous channel axis, reductions as wave-shuffle -> LDS -> a
// second deterministic pass (no float atomics).
#include "ops.h"
#include <hip/hip_bf16.h>
#include "epilogue.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
// sum over a 256-thread block; result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* sh4) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
}

static inline int nblk(size_t n, int cap = 2048) {
  size_t b = (n + 255) / 256;
  return (int)(b > (size_t)cap ? cap : (b < 1 ? 1 : b));
}

// ---------------------------------------------------------------------------
// pooling
// ---------------------------------------------------------------------------
__device__ __forceinline__ int first_argmax4(float a0, float a1, float a2, float a3) {
  int k = 0;
  float m = a0;
  if (a1 > m) { m = a1; k = 1; }
  if (a2 > m) { m = a2; k = 2; }
  if (a3 > m) { m = a3; k = 3; }
  return k;
}

__global__ void maxpool_kernel(TView in, TView out, int B, int Ho, int Wo, int C4) {
  const size_t total = (size_t)B * Ho * Wo * C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t q = i;
    const int c = (int)(q % C4) * 4;
    q /= C4;
    const int x = (int)(q % Wo);
    q /= Wo;
    const int y = (int)(q % Ho);
    const int b = (int)(q / Ho);
    const float* p = in.p + view_off(in, b, 2 * y, 2 * x) + c;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(p);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(p + in.sX);
    const f32x4 a2 = *reinterpret_cast<const f32x4*>(p + in.sY);
    const f32x4 a3 = *reinterpret_cast<const f32x4*>(p + in.sY + in.sX);
    f32x4 m;
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = fmaxf(fmaxf(a0[k], a1[k]), fmaxf(a2[k], a3[k]));
    *reinterpret_cast<f32x4*>(out.p + view_off(out, b, y, x) + c) = m;
  }
}

int dg_maxpool(TView in, TView out, int B, int Ho, int Wo, int C, hipStream_t st) {
  if (C % 4) { dg_set_error("dg_maxpool: C %% 4 != 0"); return DG_ERR_ARG; }
  const size_t total = (size_t)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool_kernel, dim3(nblk(total, 8192)), dim3(256), 0, st, in, out, B, Ho, Wo, C / 4);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

template <int MODE>  // 0: unpool+mask(+skip), 1: gather
__global__ void pool_bwd_kernel(TView d, TView a, TView skip, TView out, int B, int Ho, int Wo, int C4) {
  const size_t total = (size_t)B * Ho * Wo * C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t q = i;
    const int c = (int)(q % C4) * 4;
    q /= C4;
    const int x = (int)(q % Wo);
    q /= Wo;
    const int y = (int)(q % Ho);
    const int b = (int)(q / Ho);
    const float* pa = a.p + view_off(a, b, 2 * y, 2 * x) + c;
    f32x4 av[4];
    av[0] = *reinterpret_cast<const f32x4*>(pa);
    av[1] = *reinterpret_cast<const f32x4*>(pa + a.sX);
    av[2] = *reinterpret_cast<const f32x4*>(pa + a.sY);
    av[3] = *reinterpret_cast<const f32x4*>(pa + a.sY + a.sX);
    if (MODE == 0) {
      const f32x4 dv = *reinterpret_cast<const f32x4*>(d.p + view_off(d, b, y, x) + c);
      f32x4 o[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) o[w] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (skip.p) {
        const float* ps = skip.p + view_off(skip, b, 2 * y, 2 * x) + c;
        o[0] = *reinterpret_cast<const f32x4*>(ps);
        o[1] = *reinterpret_cast<const f32x4*>(ps + skip.sX);
        o[2] = *reinterpret_cast<const f32x4*>(ps + skip.sY);
        o[3] = *reinterpret_cast<const f32x4*>(ps + skip.sY + skip.sX);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int am = first_argmax4(av[0][k], av[1][k], av[2][k], av[3][k]);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float v = o[w][k] + ((w == am) ? dv[k] : 0.f);
          o[w][k] = (av[w][k] > 0.f) ? v : 0.f;
        }
      }
      float* po = out.p + view_off(out, b, 2 * y, 2 * x) + c;
      *reinterpret_cast<f32x4*>(po) = o[0];
      *reinterpret_cast<f32x4*>(po + out.sX) = o[1];
      *reinterpret_cast<f32x4*>(po + out.sY) = o[2];
      *reinterpret_cast<f32x4*>(po + out.sY + out.sX) = o[3];
    } else {
      const float* pu = d.p + view_off(d, b, 2 * y, 2 * x) + c;
      f32x4 uv[4];
      uv[0] = *reinterpret_cast<const f32x4*>(pu);
      uv[1] = *reinterpret_cast<const f32x4*>(pu + d.sX);
      uv[2] = *reinterpret_cast<const f32x4*>(pu + d.sY);
      uv[3] = *reinterpret_cast<const f32x4*>(pu + d.sY + d.sX);
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int am = first_argmax4(av[0][k], av[1][k], av[2][k], av[3][k]);
        o[k] = (am == 0) ? uv[0][k] : (am == 1) ? uv[1][k] : (am == 2) ? uv[2][k] : uv[3][k];
      }
      *reinterpret_cast<f32x4*>(out.p + view_off(out, b, y, x) + c) = o;
    }
  }
}

int dg_unpool_mask(TView dpool, TView a, TView skip, TView out, int B, int Ho, int Wo, int C, hipStream_t st) {
  if (C % 4) { dg_set_error("dg_unpool_mask: C %% 4 != 0"); return DG_ERR_ARG; }
  const size_t total = (size_t)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(pool_bwd_kernel<0>, dim3(nblk(total, 8192)), dim3(256), 0, st, dpool, a, skip, out, B, Ho, Wo,
                     C / 4);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}
int dg_gather_pool(TView u, TView a, TView out, int B, int Ho, int Wo, int C, hipStream_t st) {
  if (C % 4) { dg_set_error("dg_gather_pool: C %% 4 != 0"); return DG_ERR_ARG; }
  const size_t total = (size_t)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(pool_bwd_kernel<1>, dim3(nblk(total, 8192)), dim3(256), 0, st, u, a, null_view(), out, B, Ho,
                     Wo, C / 4);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// BN affine
// ---------------------------------------------------------------------------
__global__ void bn_prepare_kernel(const float* gamma, const float* beta, const float* mean, const float* var,
                                  float eps, float* s, float* t, float* rstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float r = 1.0f / sqrtf(var[c] + eps);
  const float sc = gamma[c] * r;
  s[c] = sc;
  t[c] = beta[c] - mean[c] * sc;
  rstd[c] = r;
}
int dg_bn_prepare(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* s,
                  float* t, float* rstd, int C, hipStream_t st) {
  hipLaunchKernelGGL(bn_prepare_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, gamma, beta, mean, var, eps, s, t,
                     rstd, C);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// all BatchNorms of a network in one launch: block = one BatchNorm
__global__ __launch_bounds__(256) void bn_prepare_batch_kernel(const BnJob* __restrict__ jobs, float eps) {
  const BnJob J = jobs[blockIdx.x];
  for (int c = threadIdx.x; c < J.C; c += 256) {
    const float r = 1.0f / sqrtf(J.var[c] + eps);
    const float sc = J.gamma[c] * r;
    J.s[c] = sc;
    J.t[c] = J.beta[c] - J.mean[c] * sc;
    J.rstd[c] = r;
    if (J.mean_copy) J.mean_copy[c] = J.mean[c];
  }
}
int dg_bn_prepare_batch(const BnJob* jobs_dev, int njobs, float eps, hipStream_t st) {
  if (njobs <= 0) return DG_OK;
  hipLaunchKernelGGL(bn_prepare_batch_kernel, dim3(njobs), dim3(256), 0, st, jobs_dev, eps);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// generator head
// ---------------------------------------------------------------------------
__global__ void head_fwd_kernel(const float* __restrict__ a, const float* __restrict__ w, const float* __restrict__ b,
                                float* __restrict__ out, long P, int C, int LP, int tanh_act) {
  const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long p = t / LP;
  const int part = (int)(t % LP);
  float v = 0.f;
  if (p < P) {
    const f32x4 av = *reinterpret_cast<const f32x4*>(a + p * C + part * 4);
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + part * 4);
    v = av[0] * wv[0] + av[1] * wv[1] + av[2] * wv[2] + av[3] * wv[3];
  }
  for (int o = LP >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (p < P && part == 0) {
    v += b[0];
    out[p] = tanh_act ? tanhf(v) : v;
  }
}
int dg_head_fwd(const float* a, const float* w, const float* b, float* out, long P, int C, int tanh_act,
                hipStream_t st) {
  const int LP = C / 4;
  if ((C % 4) || LP > 64 || (LP & (LP - 1))) { dg_set_error("dg_head_fwd: C/4 must be a power of two <= 64"); return DG_ERR_ARG; }
  const long threads = P * LP;
  hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a, w, b, out, P, C,
                     LP, tanh_act);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

__global__ void head_bwd_kernel(const float* __restrict__ dpre, const float* __restrict__ w,
                                const float* __restrict__ a, float* __restrict__ dz, long P, int C4) {
  const size_t total = (size_t)P * C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const long p = (long)(i / C4);
    const int c = (int)(i % C4) * 4;
    const float d = dpre[p];
    const f32x4 av = *reinterpret_cast<const f32x4*>(a + p * (C4 * 4) + c);
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (av[k] > 0.f) ? d * wv[k] : 0.f;
    *reinterpret_cast<f32x4*>(dz + p * (C4 * 4) + c) = o;
  }
}
int dg_head_bwd(const float* dpre, const float* w, const float* a, float* dz, long P, int C, hipStream_t st) {
  hipLaunchKernelGGL(head_bwd_kernel, dim3(nblk((size_t)P * (C / 4), 8192)), dim3(256), 0, st, dpre, w, a, dz, P,
                     C / 4);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// critic tail
// ---------------------------------------------------------------------------
// One block per sample, 16 waves, one pixel per wave and iteration: with 4 waves the 256 pixels of a sample were 64
// dependent load -> wave-reduce rounds per block and the launch was latency-bound (40 us for 14 MB).
__global__ __launch_bounds__(1024) void critic_tail_fwd_kernel(const float* __restrict__ a,
                                                               const float* __restrict__ w9,
                                                               const float* __restrict__ b9,
                                                               const float* __restrict__ wd,
                                                               const float* __restrict__ bd, float* __restrict__ t9,
                                                               float* __restrict__ out, int HW, int C) {
  __shared__ float sh16[16];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* an = a + (size_t)n * HW * C;
  f32x4 wv9 = {0.f, 0.f, 0.f, 0.f};
  if (lane * 4 < C) wv9 = *reinterpret_cast<const f32x4*>(w9 + lane * 4);
  float part = 0.f;
  for (int p = wv; p < HW; p += 16) {
    float v = 0.f;
    if (lane * 4 < C) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(an + (size_t)p * C + lane * 4);
      v = av[0] * wv9[0] + av[1] * wv9[1] + av[2] * wv9[2] + av[3] * wv9[3];
    }
    v = wave_sum(v);
    if (lane == 0) {
      v += b9[0];
      t9[(size_t)n * HW + p] = v;
      part += v * wd[p];
    }
  }
  if (lane == 0) sh16[wv] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += sh16[w];
    out[n] = tot + bd[0];
  }
}
int dg_critic_tail_fwd(const float* a, const float* w9, const float* b9, const float* wd, const float* bd, float* t9,
                       float* out, int N, int HW, int C, hipStream_t st) {
  if ((C % 4) || C > 256) { dg_set_error("dg_critic_tail_fwd: C must be a multiple of 4 and <= 256"); return DG_ERR_ARG; }
  hipLaunchKernelGGL(critic_tail_fwd_kernel, dim3(N), dim3(1024), 0, st, a, w9, b9, wd, bd, t9, out, HW, C);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

__global__ void critic_tail_bwd_kernel(const float* __restrict__ a, const float* __restrict__ w9,
                                       const float* __restrict__ wd, const float* __restrict__ coefs, int per,
                                       float* __restrict__ dz, int N, int HW, int C4) {
  const size_t total = (size_t)N * HW * C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const size_t np = i / C4;
    const int p = (int)(np % HW);
    const int n = (int)(np / HW);
    const float k = coefs[n / per] * wd[p];
    const f32x4 av = *reinterpret_cast<const f32x4*>(a + np * (C4 * 4) + c);
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w9 + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (av[j] > 0.f) ? k * wv[j] : 0.f;
    *reinterpret_cast<f32x4*>(dz + np * (C4 * 4) + c) = o;
  }
}
int dg_critic_tail_bwd(const float* a, const float* w9, const float* wd, const float* coefs, int per, float* dz, int N,
                       int HW, int C, hipStream_t st) {
  hipLaunchKernelGGL(critic_tail_bwd_kernel, dim3(nblk((size_t)N * HW * (C / 4), 4096)), dim3(256), 0, st, a, w9, wd,
                     coefs, per, dz, N, HW, C / 4);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// per-sample partial sums (16 waves per block, as the forward)
__global__ __launch_bounds__(1024) void critic_tail_wgrad_partial(const float* __restrict__ src,
                                                                  const float* __restrict__ w9,
                                                                  const float* __restrict__ wd,
                                                                  float* __restrict__ pw9, float* __restrict__ pwd,
                                                                  int HW, int C) {
  __shared__ float red[16 * 256];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* sn = src + (size_t)n * HW * C;
  f32x4 wv9 = {0.f, 0.f, 0.f, 0.f};
  const bool act = lane * 4 < C;
  if (act) wv9 = *reinterpret_cast<const f32x4*>(w9 + lane * 4);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int p = wv; p < HW; p += 16) {
    float v = 0.f;
    if (act) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(sn + (size_t)p * C + lane * 4);
      v = av[0] * wv9[0] + av[1] * wv9[1] + av[2] * wv9[2] + av[3] * wv9[3];
      const float wp = wd[p];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = fmaf(av[k], wp, acc[k]);
    }
    v = wave_sum(v);
    if (lane == 0) pwd[(size_t)n * HW + p] = v;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[wv * 256 + lane * 4 + k] = acc[k];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 1024) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w * 256 + c];
    pw9[(size_t)n * C + c] = s;
  }
}
// One wave per output element (C kernel taps of dw9, then HW entries of dwd): lanes split the samples, wave reduction
// in a fixed order.  The last block also forms the two bias gradients.
__global__ __launch_bounds__(256) void critic_tail_wgrad_final(const float* __restrict__ pw9,
                                                               const float* __restrict__ pwd,
                                                               const float* __restrict__ coefs, int per,
                                                               int add_bias_terms, int accumulate,
                                                               const float* __restrict__ b9,
                                                               const float* __restrict__ wd, float* __restrict__ dw9,
                                                               float* __restrict__ db9, float* __restrict__ dwd,
                                                               float* __restrict__ dbd, int N, int HW, int C) {
  __shared__ float sh4[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float csum = 0.f;                                  // sum_n coefs[n / per], the same value in every thread
  for (int g = 0; g * per < N; ++g) csum += coefs[g] * (float)min(per, N - g * per);
  const int o = blockIdx.x * 4 + wv;
  if (o < C + HW) {
    const bool isw = o < C;
    const float* src = isw ? pw9 + o : pwd + (o - C);
    const int ld = isw ? C : HW;
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s = fmaf(coefs[n / per], src[(size_t)n * ld], s);
    s = wave_sum(s);
    if (lane == 0) {
      if (isw) {
        dw9[o] = accumulate ? dw9[o] + s : s;
      } else {
        if (add_bias_terms) s += b9[0] * csum;
        dwd[o - C] = accumulate ? dwd[o - C] + s : s;
      }
    }
  }
  if (add_bias_terms && blockIdx.x == gridDim.x - 1) {
    float wsum = 0.f;
    for (int p = threadIdx.x; p < HW; p += 256) wsum += wd[p];
    wsum = block_sum(wsum, sh4);
    if (threadIdx.x == 0) {
      db9[0] = accumulate ? db9[0] + csum * wsum : csum * wsum;
      dbd[0] = accumulate ? dbd[0] + csum : csum;
    }
  }
}
int dg_critic_tail_wgrad(const float* src, const float* w9, const float* b9, const float* wd, const float* coefs,
                         int per, int add_bias_terms, int accumulate, float* dw9, float* db9, float* dwd, float* dbd, float* scratch,
                         int N, int HW, int C, hipStream_t st) {
  if ((C % 4) || C > 256) { dg_set_error("dg_critic_tail_wgrad: C must be a multiple of 4 and <= 256"); return DG_ERR_ARG; }
  float* pw9 = scratch;
  float* pwd = scratch + (size_t)N * C;
  hipLaunchKernelGGL(critic_tail_wgrad_partial, dim3(N), dim3(1024), 0, st, src, w9, wd, pw9, pwd, HW, C);
  HIPCHECK(hipGetLastError());
  hipLaunchKernelGGL(critic_tail_wgrad_final, dim3(cdiv(C + HW, 4)), dim3(256), 0, st, pw9, pwd, coefs, per, add_bias_terms,
                     accumulate, b9, wd, dw9, db9, dwd, dbd, N, HW, C);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// column sums of an NHWC view
// ---------------------------------------------------------------------------
// FLAT: the view is pixel-contiguous (sY == W*sX, sB == H*sY: plain tensors and channel slices of the concat
// buffers), so pixel q sits at q*sX and the loop carries no divisions; 4 independent loads in flight per thread.
template <bool FLAT>
__global__ void colsum_partial(TView v, long npix, int H, int W, int C4, float* __restrict__ part, int pixPerBlock,
                               const float* __restrict__ rowmul) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [256][4]
  const int LP = C4;                 // lanes per pixel
  const int PP = 256 / LP;           // pixels per iteration
  const int lp = threadIdx.x % LP, pp = threadIdx.x / LP;
  const long q0 = (long)blockIdx.x * pixPerBlock;
  const long q1 = min(q0 + pixPerBlock, npix);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (pp < PP) {
    if (FLAT) {
      f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f}, a3 = {0.f, 0.f, 0.f, 0.f};
      const float* base = v.p + lp * 4;
      long q = q0 + pp;
      for (; q + 3 * PP < q1; q += 4 * PP) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(base + q * v.sX);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(base + (q + PP) * v.sX);
        const f32x4 x2 = *reinterpret_cast<const f32x4*>(base + (q + 2 * PP) * v.sX);
        const f32x4 x3 = *reinterpret_cast<const f32x4*>(base + (q + 3 * PP) * v.sX);
        const float m0 = rowmul ? rowmul[q] : 1.0f, m1 = rowmul ? rowmul[q + PP] : 1.0f;
        const float m2 = rowmul ? rowmul[q + 2 * PP] : 1.0f, m3 = rowmul ? rowmul[q + 3 * PP] : 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[k] = fmaf(x0[k], m0, acc[k]);
          a1[k] = fmaf(x1[k], m1, a1[k]);
          a2[k] = fmaf(x2[k], m2, a2[k]);
          a3[k] = fmaf(x3[k], m3, a3[k]);
        }
      }
      for (; q < q1; q += PP) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(base + q * v.sX);
        const float m0 = rowmul ? rowmul[q] : 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fmaf(x0[k], m0, acc[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = (acc[k] + a1[k]) + (a2[k] + a3[k]);
    } else {
      for (long q = q0 + pp; q < q1; q += PP) {
        const int x = (int)(q % W);
        const long r = q / W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        const f32x4 a = *reinterpret_cast<const f32x4*>(v.p + view_off(v, b, y, x) + lp * 4);
        const float m = rowmul ? rowmul[q] : 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fmaf(a[k], m, acc[k]);
      }
    }
  }
  *reinterpret_cast<f32x4*>(sh + threadIdx.x * 4) = acc;
  __syncthreads();
  if (threadIdx.x < LP * 4) {
    const int l = threadIdx.x / 4, k = threadIdx.x % 4;
    float s = 0.f;
    for (int j = 0; j < PP; ++j) s += sh[(j * LP + l) * 4 + k];
    part[(size_t)blockIdx.x * (C4 * 4) + l * 4 + k] = s;
  }
}
__global__ void colsum_final(const float* __restrict__ part, int nb, int C, const float* __restrict__ scale,
                             float* __restrict__ out, float* __restrict__ raw, int accumulate) {
  // one 256-thread block per channel
  __shared__ float sh4[4];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) s += part[(size_t)b * C + c];
  s = block_sum(s, sh4);
  if (threadIdx.x == 0) {
    if (raw) raw[c] = s;
    if (out) {
      float v = scale ? s * scale[c] : s;
      if (accumulate) v += out[c];
      out[c] = v;
    }
  }
}
static int colsum_impl(TView v, int B, int H, int W, int C, const float* scale, float* out, float* raw, int accumulate,
                       const float* rowmul, float* scratch, hipStream_t st) {
  if ((C % 4) || C > 256) { dg_set_error("dg_colsum: C must be a multiple of 4 and <= 256"); return DG_ERR_ARG; }
  const long npix = (long)B * H * W;
  int nb = (int)((npix + 255) / 256);
  if (nb > 2048) nb = 2048;
  const int ppb = (int)((npix + nb - 1) / nb);
  nb = (int)((npix + ppb - 1) / ppb);
  const bool flat = v.sY == (long)W * v.sX && v.sB == (long)H * v.sY;
  if (flat)
    hipLaunchKernelGGL(colsum_partial<true>, dim3(nb), dim3(256), 256 * 4 * sizeof(float), st, v, npix, H, W, C / 4,
                       scratch, ppb, rowmul);
  else
    hipLaunchKernelGGL(colsum_partial<false>, dim3(nb), dim3(256), 256 * 4 * sizeof(float), st, v, npix, H, W, C / 4,
                       scratch, ppb, rowmul);
  HIPCHECK(hipGetLastError());
  hipLaunchKernelGGL(colsum_final, dim3(C), dim3(256), 0, st, scratch, nb, C, scale, out, raw, accumulate);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

int dg_colsum(TView v, int B, int H, int W, int C, const float* scale, float* out, float* raw, int accumulate,
              float* scratch, hipStream_t st) {
  return colsum_impl(v, B, H, W, C, scale, out, raw, accumulate, nullptr, scratch, st);
}
int dg_colsum_rowmul(TView v, int B, int H, int W, int C, const float* rowmul, float* out, float* scratch,
                     hipStream_t st) {
  return colsum_impl(v, B, H, W, C, nullptr, out, nullptr, 0, rowmul, scratch, st);
}

__global__ void sum_partial(const float* __restrict__ in, size_t n, float* __restrict__ part) {
  __shared__ float sh4[4];
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
  acc = block_sum(acc, sh4);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ void sum_final(const float* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ float sh4[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += part[i];
  acc = block_sum(acc, sh4);
  if (threadIdx.x == 0) out[0] = acc;
}
int dg_sum(const float* in, size_t n, float* out, float* scratch, hipStream_t st) {
  const int nb = nblk(n, 1024);
  hipLaunchKernelGGL(sum_partial, dim3(nb), dim3(256), 0, st, in, n, scratch);
  HIPCHECK(hipGetLastError());
  hipLaunchKernelGGL(sum_final, dim3(1), dim3(256), 0, st, scratch, nb, out);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// critic inputs / elementwise
// ---------------------------------------------------------------------------
__global__ void critic_inputs_kernel(const float* __restrict__ y2, const float* __restrict__ x, int nicg,
                                     const float* __restrict__ attr, const float* __restrict__ ep,
                                     float* __restrict__ out, int B, long HW, int which) {
  const size_t total = (size_t)B * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const float y1 = x[i * nicg];
    const float at = attr[i];
    const float real = which ? (y2[i] - y1) : y2[i];
    const float fake = which ? at : (y1 + at);
    const float e = ep[b];
    out[i] = real;
    out[total + i] = fake;
    out[2 * total + i] = e * real + (1.0f - e) * fake;
  }
}
int dg_critic_inputs(const float* y2, const float* x, int nicg, const float* attr, const float* ep, float* out, int B,
                     long HW, int which, hipStream_t st) {
  hipLaunchKernelGGL(critic_inputs_kernel, dim3(nblk((size_t)B * HW, 4096)), dim3(256), 0, st, y2, x, nicg, attr, ep,
                     out, B, HW, which);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

__global__ void add_ch0_kernel(const float* __restrict__ x, int nicg, const float* __restrict__ attr,
                               float* __restrict__ out, long P) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)P; i += (size_t)gridDim.x * blockDim.x)
    out[i] = x[i * nicg] + attr[i];
}
int dg_add_ch0(const float* x, int nicg, const float* attr, float* out, long P, hipStream_t st) {
  hipLaunchKernelGGL(add_ch0_kernel, dim3(nblk((size_t)P, 4096)), dim3(256), 0, st, x, nicg, attr, out, P);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// gradient penalty
// ---------------------------------------------------------------------------
#define GP_SPLIT 64
__global__ void gp_sq_partial(const float* __restrict__ g0, long HW, float* __restrict__ part) {
  __shared__ float sh4[4];
  const int b = blockIdx.x, s = blockIdx.y;
  const long per = (HW + GP_SPLIT - 1) / GP_SPLIT;
  const long i0 = s * per, i1 = min(i0 + per, HW);
  float acc = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const float v = g0[(size_t)b * HW + i];
    acc = fmaf(v, v, acc);
  }
  acc = block_sum(acc, sh4);
  if (threadIdx.x == 0) part[b * GP_SPLIT + s] = acc;
}
__global__ void gp_scale_kernel(const float* __restrict__ g0, float* __restrict__ u0, const float* __restrict__ part,
                                float* __restrict__ norms, float delta, int B, long HW) {
  const int b = blockIdx.x, s = blockIdx.y;
  float ss = 0.f;
  for (int k = 0; k < GP_SPLIT; ++k) ss += part[b * GP_SPLIT + k];
  const float nrm = sqrtf(ss);
  // d/dg [ delta * mean_b (||g||-1)^2 ] ; no epsilon, as in the reference (GT:544)
  const float coef = delta * (2.0f / (float)B) * (nrm - 1.0f) / nrm;
  if (s == 0 && threadIdx.x == 0) norms[b] = nrm;
  const long per = (HW + GP_SPLIT - 1) / GP_SPLIT;
  const long i0 = s * per, i1 = min(i0 + per, HW);
  for (long i = i0 + threadIdx.x; i < i1; i += blockDim.x) u0[(size_t)b * HW + i] = coef * g0[(size_t)b * HW + i];
}
__global__ void gp_value_kernel(const float* __restrict__ norms, float* __restrict__ gp_out, int B) {
  __shared__ float sh4[4];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float d = norms[b] - 1.0f;
    acc += d * d;
  }
  acc = block_sum(acc, sh4);
  if (threadIdx.x == 0) gp_out[0] = acc / (float)B;
}
int dg_gp_u0(const float* g0, float* u0, float* norms, float* gp_out, float delta, int B, long HW, float* scratch,
             hipStream_t st) {
  hipLaunchKernelGGL(gp_sq_partial, dim3(B, GP_SPLIT), dim3(256), 0, st, g0, HW, scratch);
  HIPCHECK(hipGetLastError());
  hipLaunchKernelGGL(gp_scale_kernel, dim3(B, GP_SPLIT), dim3(256), 0, st, g0, u0, scratch, norms, delta, B, HW);
  HIPCHECK(hipGetLastError());
  if (gp_out) {   // the step drivers take sum (norm-1)^2 from dg_critic_stats instead
    hipLaunchKernelGGL(gp_value_kernel, dim3(1), dim3(256), 0, st, norms, gp_out, B);
    HIPCHECK(hipGetLastError());
  }
  return DG_OK;
}

// ---------------------------------------------------------------------------
// generator loss pieces
// ---------------------------------------------------------------------------
__global__ void gloss_partial(const float* __restrict__ x, int nicg, const float* __restrict__ y2,
                              const float* __restrict__ attr, float thr, long P, float* __restrict__ part) {
  __shared__ float sh4[4];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)P; i += (size_t)gridDim.x * blockDim.x) {
    const float y1 = x[i * nicg], yy = y2[i], at = attr[i];
    s0 += fabsf(at - (yy - y1));
    const float wr = (yy >= thr) ? 1.f : 0.f;
    const float wf = ((y1 + at) >= thr) ? 1.f : 0.f;
    s1 += wr;
    s2 += wf;
    s3 += wr * wf;
  }
  s0 = block_sum(s0, sh4);
  s1 = block_sum(s1, sh4);
  s2 = block_sum(s2, sh4);
  s3 = block_sum(s3, sh4);
  if (threadIdx.x == 0) {
    part[blockIdx.x * 4 + 0] = s0;
    part[blockIdx.x * 4 + 1] = s1;
    part[blockIdx.x * 4 + 2] = s2;
    part[blockIdx.x * 4 + 3] = s3;
  }
}
__global__ void gloss_final(const float* __restrict__ part, int nb, float* __restrict__ sums) {
  __shared__ float sh4[4];
  for (int k = 0; k < 4; ++k) {
    float s = 0.f;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) s += part[b * 4 + k];
    s = block_sum(s, sh4);
    if (threadIdx.x == 0) sums[k] = s;
  }
}
int dg_gloss_sums(const float* x, int nicg, const float* y2, const float* attr, float thr, float* sums, long P,
                  float* scratch, hipStream_t st) {
  const int nb = nblk((size_t)P, 1024);
  hipLaunchKernelGGL(gloss_partial, dim3(nb), dim3(256), 0, st, x, nicg, y2, attr, thr, P, scratch);
  HIPCHECK(hipGetLastError());
  hipLaunchKernelGGL(gloss_final, dim3(1), dim3(256), 0, st, scratch, nb, sums);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

__global__ void g_dpre_kernel(const float* __restrict__ x, int nicg, const float* __restrict__ y2,
                              const float* __restrict__ attr, const float* __restrict__ g1,
                              const float* __restrict__ g2, float* __restrict__ dpre, float invB, float m1c, long P) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)P; i += (size_t)gridDim.x * blockDim.x) {
    const float at = attr[i];
    const float diff = at - (y2[i] - x[i * nicg]);
    const float sg = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
    const float d = -(g1[i] + g2[i]) * invB + m1c * sg;
    dpre[i] = d * (1.0f - at * at);
  }
}
int dg_g_dpre(const float* x, int nicg, const float* y2, const float* attr, const float* g1, const float* g2,
              float* dpre, int B, long P, hipStream_t st) {
  hipLaunchKernelGGL(g_dpre_kernel, dim3(nblk((size_t)P, 4096)), dim3(256), 0, st, x, nicg, y2, attr, g1, g2, dpre,
                     1.0f / (float)B, 100.0f / (float)P, P);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// FiLM backward
// ---------------------------------------------------------------------------
#define FILM_SPLIT 64
__global__ void film_bwd_partial(const float* __restrict__ dr, const float* __restrict__ u,
                                 const float* __restrict__ fmul, const float* __restrict__ fadd, int film_ld,
                                 float* __restrict__ du, float* __restrict__ part, long HW, int C4) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [256][8]
  const int b = blockIdx.x, s = blockIdx.y;
  const int LP = C4, PP = 256 / LP;
  const int lp = threadIdx.x % LP, pp = threadIdx.x / LP;
  const long per = (HW + FILM_SPLIT - 1) / FILM_SPLIT;
  const long q0 = s * per, q1 = min(q0 + per, HW);
  const int C = C4 * 4;
  f32x4 am = {0.f, 0.f, 0.f, 0.f}, aa = {0.f, 0.f, 0.f, 0.f};
  if (pp < PP) {
    const f32x4 fm = *reinterpret_cast<const f32x4*>(fmul + (size_t)b * film_ld + lp * 4);
    const f32x4 fa = *reinterpret_cast<const f32x4*>(fadd + (size_t)b * film_ld + lp * 4);
    for (long q = q0 + pp; q < q1; q += PP) {
      const size_t off = ((size_t)b * HW + q) * C + lp * 4;
      const f32x4 d = *reinterpret_cast<const f32x4*>(dr + off);
      const f32x4 uu = *reinterpret_cast<const f32x4*>(u + off);
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = film_preact(uu[k], fm[k], fa[k]);
        const float dv = (v > 0.f) ? d[k] : 0.f;
        aa[k] += dv;
        am[k] = fmaf(dv, uu[k], am[k]);
        o[k] = dv * fm[k];
      }
      *reinterpret_cast<f32x4*>(du + off) = o;
    }
  }
  *reinterpret_cast<f32x4*>(sh + threadIdx.x * 8) = am;
  *reinterpret_cast<f32x4*>(sh + threadIdx.x * 8 + 4) = aa;
  __syncthreads();
  if (threadIdx.x < LP * 8) {
    const int l = threadIdx.x / 8, k = threadIdx.x % 8;
    float acc = 0.f;
    for (int j = 0; j < PP; ++j) acc += sh[(j * LP + l) * 8 + k];
    // part layout: [b][split][2][C]
    part[(((size_t)b * FILM_SPLIT + s) * 2 + (k >> 2)) * C + l * 4 + (k & 3)] = acc;
  }
}
__global__ void film_bwd_final(const float* __restrict__ part, float* __restrict__ dmul, float* __restrict__ dadd,
                               int film_ld, int C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float sm = 0.f, sa = 0.f;
    for (int s = 0; s < FILM_SPLIT; ++s) {
      sm += part[(((size_t)b * FILM_SPLIT + s) * 2 + 0) * C + c];
      sa += part[(((size_t)b * FILM_SPLIT + s) * 2 + 1) * C + c];
    }
    dmul[(size_t)b * film_ld + c] = sm;
    dadd[(size_t)b * film_ld + c] = sa;
  }
}
int dg_film_bwd(const float* dr, const float* u, const float* fmul, const float* fadd, int film_ld, float* du,
                float* dmul, float* dadd, int B, long HW, int C, float* scratch, hipStream_t st) {
  if ((C % 4) || C > 128) { dg_set_error("dg_film_bwd: C must be a multiple of 4 and <= 128"); return DG_ERR_ARG; }
  hipLaunchKernelGGL(film_bwd_partial, dim3(B, FILM_SPLIT), dim3(256), 256 * 8 * sizeof(float), st, dr, u, fmul, fadd,
                     film_ld, du, scratch, HW, C / 4);
  HIPCHECK(hipGetLastError());
  hipLaunchKernelGGL(film_bwd_final, dim3(B), dim3(256), 0, st, scratch, dmul, dadd, film_ld, C);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// BN gamma gradient
// ---------------------------------------------------------------------------
__global__ void bn_gamma_grad_kernel(const float* __restrict__ W, const float* __restrict__ dWraw, int K, int Cout,
                                     int oi, int Cin, const float* __restrict__ bias, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, const float* __restrict__ S,
                                     float* __restrict__ dgamma) {
  // one 256-thread block per output channel
  __shared__ float sh4[4];
  const int co = blockIdx.x;
  float acc = 0.f;
  if (!oi) {
    for (int k = threadIdx.x; k < K; k += blockDim.x)
      acc = fmaf(W[(size_t)k * Cout + co], dWraw[(size_t)k * Cout + co], acc);
  } else {
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
      const int t = k / Cin, ci = k - t * Cin;
      const size_t o = ((size_t)t * Cout + co) * Cin + ci;
      acc = fmaf(W[o], dWraw[o], acc);
    }
  }
  acc = block_sum(acc, sh4);
  if (threadIdx.x == 0) dgamma[co] = rstd[co] * (acc + (bias[co] - mean[co]) * S[co]);
}
__global__ void bn_gamma_grad_batch_kernel(const GammaJob* __restrict__ jobs, int njobs) {
  __shared__ float sh4[4];
  // the job of this block: jobs are few (tens), blk0 ascending
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].blk0) ++j;
  const GammaJob J = jobs[j];
  const int co = (int)blockIdx.x - J.blk0;
  float acc = 0.f;
  if (!J.oi) {
    for (int k = threadIdx.x; k < J.K; k += blockDim.x)
      acc = fmaf(J.W[(size_t)k * J.Cout + co], J.dWraw[(size_t)k * J.Cout + co], acc);
  } else {
    for (int k = threadIdx.x; k < J.K; k += blockDim.x) {
      const int t = k / J.Cin, ci = k - t * J.Cin;
      const size_t o = ((size_t)t * J.Cout + co) * J.Cin + ci;
      acc = fmaf(J.W[o], J.dWraw[o], acc);
    }
  }
  acc = block_sum(acc, sh4);
  if (threadIdx.x == 0) J.dgamma[co] = J.rstd[co] * (acc + (J.bias[co] - J.mean[co]) * J.S[co]);
}
int dg_bn_gamma_grad_batch(const GammaJob* jobs_dev, int njobs, int nblocks, hipStream_t st) {
  if (njobs <= 0) return DG_OK;
  hipLaunchKernelGGL(bn_gamma_grad_batch_kernel, dim3(nblocks), dim3(256), 0, st, jobs_dev, njobs);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

int dg_bn_gamma_grad(const float* W, const float* dWraw, int K, int Cout, int oi, int Cin, const float* bias,
                     const float* mean, const float* rstd, const float* S, float* dgamma, hipStream_t st) {
  hipLaunchKernelGGL(bn_gamma_grad_kernel, dim3(Cout), dim3(256), 0, st, W, dWraw, K, Cout, oi, Cin, bias,
                     mean, rstd, S, dgamma);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// Adam
// ---------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float lr_t, float b1, float b2, float eps, float gscale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}
int dg_adam(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps,
            float gscale, hipStream_t st) {
  hipLaunchKernelGGL(adam_kernel, dim3(nblk(n, 2048)), dim3(256), 0, st, p, g, m, v, n, lr_t, b1, b2, eps, gscale);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

__global__ void scale_copy_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n, float s) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = in[i] * s;
}
int dg_scale_copy(const float* in, float* out, size_t n, float s, hipStream_t st) {
  hipLaunchKernelGGL(scale_copy_kernel, dim3(nblk(n, 2048)), dim3(256), 0, st, in, out, n, s);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

__global__ void mean_groups_kernel(const float* __restrict__ in, float* __restrict__ out, int per) {
  __shared__ float sh4[4];
  const int g = blockIdx.x;
  float acc = 0.f;
  for (int i = threadIdx.x; i < per; i += blockDim.x) acc += in[(size_t)g * per + i];
  acc = block_sum(acc, sh4);
  if (threadIdx.x == 0) out[g] = acc / (float)per;
}
int dg_mean_groups(const float* in, float* out, int groups, int per, hipStream_t st) {
  hipLaunchKernelGGL(mean_groups_kernel, dim3(groups), dim3(256), 0, st, in, out, per);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// un-normalised pieces of one critic evaluation (one block): [sum D(real), sum D(fake), sum (norm-1)^2, B]
__global__ void critic_stats_kernel(const float* __restrict__ d_out, const float* __restrict__ norms,
                                    float* __restrict__ out, int B) {
  __shared__ float sh4[4];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    a0 += d_out[i];
    a1 += d_out[B + i];
    const float d = norms[i] - 1.0f;
    a2 += d * d;
  }
  a0 = block_sum(a0, sh4);
  a1 = block_sum(a1, sh4);
  a2 = block_sum(a2, sh4);
  if (threadIdx.x == 0) {
    out[0] = a0;
    out[1] = a1;
    out[2] = a2;
    out[3] = (float)B;
  }
}
int dg_critic_stats(const float* d_out, const float* norms, float* out, int B, hipStream_t st) {
  hipLaunchKernelGGL(critic_stats_kernel, dim3(1), dim3(256), 0, st, d_out, norms, out, B);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

__global__ void sum_groups_consts_kernel(const float* __restrict__ in, float* __restrict__ out, int per, int coff,
                                         float c0, float c1) {
  __shared__ float sh4[4];
  const int g = blockIdx.x;
  float acc = 0.f;
  for (int i = threadIdx.x; i < per; i += blockDim.x) acc += in[(size_t)g * per + i];
  acc = block_sum(acc, sh4);
  if (threadIdx.x == 0) {
    out[g] = acc;
    if (g == 0) {
      out[coff] = c0;
      out[coff + 1] = c1;
    }
  }
}
int dg_sum_groups_consts(const float* in, float* out, int groups, int per, int coff, float c0, float c1,
                         hipStream_t st) {
  hipLaunchKernelGGL(sum_groups_consts_kernel, dim3(groups), dim3(256), 0, st, in, out, per, coff, c0, c1);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// The generator's total loss from its 8 un-normalised pieces: the SAME statement sequence as g_loss_from_sums() in
// model.hip (host), double arithmetic without contraction, rounded to float -- so the device's arg-min is the host's.
__device__ float g_total_loss_dev(const float* s) {
#pragma clang fp contract(off)
  const double n = s[6], npix = s[7];
  const double lf = (double)s[0] / n, lfd = (double)s[1] / n;
  const double m1 = 100.0 * (double)s[2] / npix;
  const double dv = (double)s[3] / 1000.0 - (double)s[4] / 1000.0;
  const double m3 = 100.0 * dv * dv;
  const double dice = (2.0 * (double)s[5] + 1e-7) / ((double)s[3] + (double)s[4] + 1e-7);
  const double m4 = 1.0 - dice;
  return (float)(-lf - lfd + m1 + m3 + m4);
}
__global__ void best_noise_kernel(const float* __restrict__ stats, int k, const float* __restrict__ z_all, long zfloats,
                                  int* __restrict__ best, float* __restrict__ z_out) {
  __shared__ int sbest;
  if (threadIdx.x == 0) {
    int bi = 0;
    float bv = g_total_loss_dev(stats);
    for (int i = 1; i < k; ++i) {
      const float v = g_total_loss_dev(stats + 8 * i);
      if (v < bv) { bv = v; bi = i; }     // first minimum, like np.argmin; NaN never wins (as in NumPy only if first)
    }
    sbest = bi;
    *best = bi;
  }
  __syncthreads();
  const float* src = z_all + (size_t)sbest * zfloats;
  for (long i = threadIdx.x; i < zfloats; i += blockDim.x) z_out[i] = src[i];
}
int dg_best_noise(const float* stats, int k, const float* z_all, long zfloats, int* best, float* z_out,
                  hipStream_t st) {
  hipLaunchKernelGGL(best_noise_kernel, dim3(1), dim3(256), 0, st, stats, k, z_all, zfloats, best, z_out);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// bf16-weights mode: round the kernels of the fp32 master into the compute copy
// ---------------------------------------------------------------------------
__global__ void round_bf16_masked_kernel(const float* __restrict__ src, const unsigned char* __restrict__ mask,
                                         float* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = src[i];
    // plain cast: v_cvt_pk_bf16_f32, round-to-nearest-even, NaN stays NaN
    dst[i] = mask[i] ? __bfloat162float(__float2bfloat16(v)) : v;
  }
}
int dg_round_bf16_masked(const float* src, const unsigned char* mask, float* dst, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(round_bf16_masked_kernel, dim3(nblk(n, 2048)), dim3(256), 0, st, src, mask, dst, n);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// evaluation step after the path (DEP-GAN_testing_4fold.py "GE":616-807)
// ---------------------------------------------------------------------------
// acc += weight * pred * mask   (GE:623-625: the running sum of the n_repeat masked predictions)
// GE:617-624: the running sum is float64 (np.zeros), each term the float32 product prediction * mask
__global__ void eval_accumulate_kernel(const float* __restrict__ pred, const float* __restrict__ mask,
                                       double* __restrict__ acc, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float m = mask ? mask[i] : 1.0f;
    acc[i] = __dadd_rn(acc[i], (double)__fmul_rn(pred[i], m));
  }
}
int dg_eval_accumulate(const float* pred, const float* mask, double* acc, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(eval_accumulate_kernel, dim3(nblk(n, 2048)), dim3(256), 0, st, pred, mask, acc, n);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}
// GE:628: output_img_pred_mean / float(n_repeat), a float64 division
__global__ void eval_divide_kernel(double* __restrict__ acc, size_t n, double d) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc[i] = __ddiv_rn(acc[i], d);
}
int dg_eval_divide(double* acc, size_t n, double d, hipStream_t st) {
  hipLaunchKernelGGL(eval_divide_kernel, dim3(nblk(n, 2048)), dim3(256), 0, st, acc, n, d);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// Integer census behind the volume and Dice figures of GE:637-790 (all counts: exact, order-independent).
//  [0] nnz(mask1*wmh1)  [1] nnz(mask2*wmh2)  [2] #(x >= T, all input channels)  [3] #(prob2 >= T)
//  [4] nnz(mask2 * [fake > T])            fake = clip(x0 + pred, -1, 1), x0 = channel 0 of x
//  change code of the prediction: 1 (shrink) fake < T & x0 >= T; 2 (grow) fake >= T & x0 < T; 3 (stay) both >= T
//  [5+3(k-1) ..] for k = 1,2,3: #(fake_code == k & real_code == k), #(real_code == k), #(fake_code == k)
//  [14..16] the same for code > 0 (whole WMH), [17..19] for code in {1,2} (changing WMH)
#define DG_EVAL_NCOUNT 20
// dtypes follow the reference's NumPy statements: the mean prediction is float64, so fake = x0 + pred, its clip and
// its comparisons run in float64 against the float64 threshold; x and prob2 are float32 arrays, which NumPy compares
// with a Python-float threshold in float32.
__global__ void eval_counts_kernel(const float* __restrict__ x, int nicg, const double* __restrict__ pred,
                                   const float* __restrict__ code_real, const float* __restrict__ mask1,
                                   const float* __restrict__ wmh1, const float* __restrict__ mask2,
                                   const float* __restrict__ wmh2, const float* __restrict__ prob2, size_t npix,
                                   double thr_d, unsigned long long* __restrict__ out) {
  const float thr = (float)thr_d;
  __shared__ unsigned int sh[DG_EVAL_NCOUNT];
  if (threadIdx.x < DG_EVAL_NCOUNT) sh[threadIdx.x] = 0;
  __syncthreads();
  unsigned int c[DG_EVAL_NCOUNT];
#pragma unroll
  for (int k = 0; k < DG_EVAL_NCOUNT; ++k) c[k] = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    const float x0 = x[i * nicg];
    double fake = __dadd_rn((double)x0, pred[i]);
    fake = fake < -1.0 ? -1.0 : (fake > 1.0 ? 1.0 : fake);
    if (mask1 && wmh1) c[0] += (__fmul_rn(mask1[i], wmh1[i]) != 0.0f);
    if (mask2 && wmh2) c[1] += (__fmul_rn(mask2[i], wmh2[i]) != 0.0f);
    for (int ch = 0; ch < nicg; ++ch) c[2] += (x[i * nicg + ch] >= thr);
    if (prob2) c[3] += (prob2[i] >= thr);
    const float m2 = mask2 ? mask2[i] : 1.0f;
    c[4] += (fake > thr_d) && (m2 != 0.0f);
    const bool fhi = fake >= thr_d, xhi = x0 >= thr;
    const int fc = (!fhi && xhi) ? 1 : ((fhi && !xhi) ? 2 : ((fhi && xhi) ? 3 : 0));
    const float rcf = code_real ? code_real[i] : 0.0f;
#pragma unroll
    for (int k = 1; k <= 3; ++k) {
      const bool r = (rcf == (float)k), f = (fc == k);
      c[5 + 3 * (k - 1)] += (r && f);
      c[6 + 3 * (k - 1)] += r;
      c[7 + 3 * (k - 1)] += f;
    }
    {
      const bool r = rcf > 0.0f, f = fc > 0;
      c[14] += (r && f); c[15] += r; c[16] += f;
    }
    {
      const bool r = (rcf == 1.0f) || (rcf == 2.0f), f = (fc == 1) || (fc == 2);
      c[17] += (r && f); c[18] += r; c[19] += f;
    }
  }
#pragma unroll
  for (int k = 0; k < DG_EVAL_NCOUNT; ++k)
    if (c[k]) atomicAdd(&sh[k], c[k]);       // integer adds: any order gives the same total
  __syncthreads();
  if (threadIdx.x < DG_EVAL_NCOUNT && sh[threadIdx.x]) atomicAdd(&out[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}
int dg_eval_counts(const float* x, int nicg, const double* pred, const float* code_real, const float* mask1,
                   const float* wmh1, const float* mask2, const float* wmh2, const float* prob2, size_t npix, double thr,
                   unsigned long long* out_dev, hipStream_t st) {
  HIPCHECK(hipMemsetAsync(out_dev, 0, DG_EVAL_NCOUNT * sizeof(unsigned long long), st));
  hipLaunchKernelGGL(eval_counts_kernel, dim3(nblk(npix, 1024)), dim3(256), 0, st, x, nicg, pred, code_real, mask1, wmh1,
                     mask2, wmh2, prob2, npix, thr, out_dev);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}
