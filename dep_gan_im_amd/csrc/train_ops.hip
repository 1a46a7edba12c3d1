// Learning-phase-1 operators (see train_ops.h).  All HBM-bound: 16-byte accesses over the channel
// axis, block reductions through LDS, deterministic second passes.
#include "train_ops.h"
#include "epilogue.h"

__device__ __forceinline__ float t_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ float t_block_sum(float v, float* sh4) {
  v = t_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
}
static inline int t_nblk(size_t n, int cap) {
  size_t b = (n + 255) / 256;
  return (int)(b > (size_t)cap ? cap : (b < 1 ? 1 : b));
}

// MODE 0: (sum x, -)   MODE 1: (sum (x-m)^2, -)   MODE 2: (sum d, sum d*(x-m))   [v = d, w = x]
// FLAT: both views are pixel-contiguous (sY == W*sX, sB == H*sY), so pixel q sits at q*sX: no divisions in the loop.
template <int MODE, bool FLAT>
__global__ void colsum2_partial(TView v, TView w, const float* __restrict__ m, long npix, int H, int W, int C4,
                                float* __restrict__ part, int pixPerBlock) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [256][8]
  const int LP = C4, PP = 256 / LP;
  const int lp = threadIdx.x % LP, pp = threadIdx.x / LP;
  const long q0 = (long)blockIdx.x * pixPerBlock, q1 = min(q0 + pixPerBlock, npix);
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  if (pp < PP) {
    f32x4 mv = {0.f, 0.f, 0.f, 0.f};
    if (MODE >= 1) mv = *reinterpret_cast<const f32x4*>(m + lp * 4);
    for (long q = q0 + pp; q < q1; q += PP) {
      long ov, ow;
      if (FLAT) {
        ov = q * v.sX;
        ow = q * w.sX;
      } else {
        const int x = (int)(q % W);
        const long r = q / W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        ov = view_off(v, b, y, x);
        ow = (MODE == 2) ? view_off(w, b, y, x) : 0;
      }
      const f32x4 xv = *reinterpret_cast<const f32x4*>(v.p + ov + lp * 4);
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a0[k] += xv[k];
      } else if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d = xv[k] - mv[k];
          a0[k] = fmaf(d, d, a0[k]);
        }
      } else {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w.p + ow + lp * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          a0[k] += xv[k];
          a1[k] = fmaf(xv[k], wv[k] - mv[k], a1[k]);
        }
      }
    }
  }
  *reinterpret_cast<f32x4*>(sh + threadIdx.x * 8) = a0;
  *reinterpret_cast<f32x4*>(sh + threadIdx.x * 8 + 4) = a1;
  __syncthreads();
  for (int idx = threadIdx.x; idx < LP * 8; idx += 256) {
    const int l = idx / 8, k = idx % 8;
    float s = 0.f;
    for (int j = 0; j < PP; ++j) s += sh[(j * LP + l) * 8 + k];
    const int C = C4 * 4;
    part[((size_t)blockIdx.x * 2 + (k >> 2)) * C + l * 4 + (k & 3)] = s;
  }
}
// out[j][c] = scale * sum_blk part[blk][j][c], one block per (c, j)
__global__ void colsum2_final(const float* __restrict__ part, int nb, int C, float scale, float* __restrict__ out0,
                              float* __restrict__ out1) {
  __shared__ float sh4[4];
  const int c = blockIdx.x, j = blockIdx.y;
  float s = 0.f;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) s += part[((size_t)b * 2 + j) * C + c];
  s = t_block_sum(s, sh4);
  if (threadIdx.x == 0) (j == 0 ? out0 : out1)[c] = s * scale;
}

static int colsum2_launch(int mode, TView v, TView w, const float* m, int B, int H, int W, int C, float scale,
                          float* out0, float* out1, float* scratch, hipStream_t st) {
  if ((C % 4) || C > 1024) {
    dg_set_error("train colsum: C must be a multiple of 4 and <= 1024 (got %d)", C);
    return DG_ERR_ARG;
  }
  const long npix = (long)B * H * W;
  int nb = (int)((npix + 255) / 256);
  if (nb > 1024) nb = 1024;
  const int ppb = (int)((npix + nb - 1) / nb);
  nb = (int)((npix + ppb - 1) / ppb);
  const size_t lds = 256 * 8 * sizeof(float);
  auto is_flat = [&](const TView& t) { return !t.p || (t.sY == (long)W * t.sX && t.sB == (long)H * t.sY); };
  const bool flat = is_flat(v) && is_flat(w);
#define DG_CS2(MODE)                                                                                                  \
  do {                                                                                                                 \
    if (flat)                                                                                                          \
      hipLaunchKernelGGL((colsum2_partial<MODE, true>), dim3(nb), dim3(256), lds, st, v, w, m, npix, H, W, C / 4,      \
                         scratch, ppb);                                                                                \
    else                                                                                                               \
      hipLaunchKernelGGL((colsum2_partial<MODE, false>), dim3(nb), dim3(256), lds, st, v, w, m, npix, H, W, C / 4,     \
                         scratch, ppb);                                                                                \
  } while (0)
  if (mode == 0) DG_CS2(0);
  else if (mode == 1) DG_CS2(1);
  else DG_CS2(2);
#undef DG_CS2
  HIPCHECK(hipGetLastError());
  hipLaunchKernelGGL(colsum2_final, dim3(C, out1 ? 2 : 1), dim3(256), 0, st, scratch, nb, C, scale, out0, out1);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// Batch mean and (biased) variance per channel in ONE pass over the tensor (round 3; was two: mean, then centred squares).
// Plain E[x^2] - E[x]^2 loses the variance of channels with |mean| >> sigma, so every block sums (x - s) and (x - s)^2
// around a shift s of its own -- the value of its first pixel, a sample of the very distribution, so |mean_b - s| is a
// few sigma and the block's subtraction M2_b = S2 - S1^2 / n_b is benign -- and the per-channel finish combines the
// blocks' (n_b, mean_b, M2_b) with the parallel-variance formula (Chan et al.) in double precision:
//   mean = sum n_b mean_b / N,   M2 = sum M2_b + sum n_b (mean_b - mean)^2,   var = M2 / N.
template <bool FLAT>
__global__ void moments_partial(TView v, long npix, int H, int W, int C4, float* __restrict__ part, int pixPerBlock) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [256][8]
  const int LP = C4, PP = 256 / LP;
  const int lp = threadIdx.x % LP, pp = threadIdx.x / LP;
  const long q0 = (long)blockIdx.x * pixPerBlock, q1 = min(q0 + pixPerBlock, npix);
  auto off = [&](long q) -> long {
    if (FLAT) return q * v.sX;
    const int x = (int)(q % W);
    const long r = q / W;
    return view_off(v, (int)(r / H), (int)(r % H), x);
  };
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f}, sv = {0.f, 0.f, 0.f, 0.f};
  if (pp < PP) {
    sv = *reinterpret_cast<const f32x4*>(v.p + off(q0) + lp * 4);          // the block's shift: its first pixel
    for (long q = q0 + pp; q < q1; q += PP) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(v.p + off(q) + lp * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = xv[k] - sv[k];
        a0[k] += d;
        a1[k] = fmaf(d, d, a1[k]);
      }
    }
  }
  *reinterpret_cast<f32x4*>(sh + threadIdx.x * 8) = a0;
  *reinterpret_cast<f32x4*>(sh + threadIdx.x * 8 + 4) = a1;
  __syncthreads();
  const int C = C4 * 4;
  for (int idx = threadIdx.x; idx < LP * 8; idx += 256) {
    const int l = idx / 8, k = idx % 8;
    float s = 0.f;
    for (int j = 0; j < PP; ++j) s += sh[(j * LP + l) * 8 + k];
    part[((size_t)blockIdx.x * 3 + (k >> 2)) * C + l * 4 + (k & 3)] = s;
  }
  if (pp == 0) *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 3 + 2) * C + lp * 4) = sv;
}
// one block per channel: Chan's combination of the blocks' (n_b, mean_b, M2_b), double precision, fixed order
__global__ void moments_final(const float* __restrict__ part, int nb, int C, long npix, int pixPerBlock,
                              float* __restrict__ mean, float* __restrict__ var) {
  __shared__ double shd[256];
  const int c = blockIdx.x;
  auto nof = [&](int b) -> double {
    const long q0 = (long)b * pixPerBlock;
    const long q1 = q0 + pixPerBlock < npix ? q0 + pixPerBlock : npix;
    return (double)(q1 - q0);
  };
  auto block_sum = [&](double x) -> double {
    shd[threadIdx.x] = x;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o];
      __syncthreads();
    }
    const double r = shd[0];
    __syncthreads();
    return r;
  };
  double sm = 0.0;
  for (int b = threadIdx.x; b < nb; b += 256) {
    const double s1 = part[((size_t)b * 3 + 0) * C + c], sh_ = part[((size_t)b * 3 + 2) * C + c];
    sm += nof(b) * sh_ + s1;                       // n_b mean_b = n_b s_b + S1_b
  }
  const double mu = block_sum(sm) / (double)npix;
  double m2 = 0.0;
  for (int b = threadIdx.x; b < nb; b += 256) {
    const double n = nof(b);
    const double s1 = part[((size_t)b * 3 + 0) * C + c], s2 = part[((size_t)b * 3 + 1) * C + c];
    const double mb = (double)part[((size_t)b * 3 + 2) * C + c] + s1 / n;
    m2 += (s2 - s1 * s1 / n) + n * (mb - mu) * (mb - mu);
  }
  const double M2 = block_sum(m2);
  if (threadIdx.x == 0) {
    mean[c] = (float)mu;
    var[c] = (float)(M2 / (double)npix);
  }
}

int dg_col_moments(TView v, int B, int H, int W, int C, float* mean, float* var, float* scratch, hipStream_t st) {
  if ((C % 4) || C > 1024) {
    dg_set_error("train moments: C must be a multiple of 4 and <= 1024 (got %d)", C);
    return DG_ERR_ARG;
  }
  const long npix = (long)B * H * W;
  int nb = (int)((npix + 255) / 256);
  if (nb > 1024) nb = 1024;
  const int ppb = (int)((npix + nb - 1) / nb);
  nb = (int)((npix + ppb - 1) / ppb);
  const size_t lds = 256 * 8 * sizeof(float);
  const bool flat = (v.sY == (long)W * v.sX && v.sB == (long)H * v.sY);
  if (flat)
    hipLaunchKernelGGL((moments_partial<true>), dim3(nb), dim3(256), lds, st, v, npix, H, W, C / 4, scratch, ppb);
  else
    hipLaunchKernelGGL((moments_partial<false>), dim3(nb), dim3(256), lds, st, v, npix, H, W, C / 4, scratch, ppb);
  HIPCHECK(hipGetLastError());
  hipLaunchKernelGGL(moments_final, dim3(C), dim3(256), 0, st, scratch, nb, C, npix, ppb, mean, var);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}
int dg_colsum_pair(TView d, TView x, const float* mean, int B, int H, int W, int C, float* sums, float* scratch,
                   hipStream_t st) {
  return colsum2_launch(2, d, x, mean, B, H, W, C, 1.0f, sums, sums + C, scratch, st);
}

__global__ void bn_train_prepare_kernel(const float* gamma, const float* beta, const float* mean, const float* var,
                                        float eps, float momentum, float corr, float* mm, float* mv, float* s,
                                        float* t, float* rstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float r = 1.0f / sqrtf(var[c] + eps);
  const float sc = gamma[c] * r;
  s[c] = sc;
  t[c] = beta[c] - mean[c] * sc;
  rstd[c] = r;
  if (mm) {
    mm[c] = mm[c] * momentum + mean[c] * (1.0f - momentum);
    mv[c] = mv[c] * momentum + var[c] * corr * (1.0f - momentum);
  }
}
int dg_bn_train_prepare(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                        float momentum, float corr, float* moving_mean, float* moving_var, float* s, float* t,
                        float* rstd, int C, hipStream_t st) {
  hipLaunchKernelGGL(bn_train_prepare_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, gamma, beta, mean, var, eps,
                     momentum, corr, moving_mean, moving_var, s, t, rstd, C);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

__device__ __forceinline__ unsigned hash_u32(unsigned i, unsigned seed) {
  unsigned x = (i * 0x9E3779B1u) ^ seed;
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}

// FLAT: every view is pixel-contiguous, so pixel q of view t sits at q * t.sX -- 32-bit index arithmetic only.
template <bool FLAT>
__global__ void affine_act_kernel(const AffineActArgs a, unsigned drop_thr, float drop_scale) {
  const int C4 = a.C / 4;
  const size_t total = (size_t)a.B * a.H * a.W * C4;
  const unsigned HW = (unsigned)a.H * (unsigned)a.W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c, b;
    long o_in, o_out, o_pre, o_res;
    if (FLAT) {
      const unsigned pix = (unsigned)(i / (unsigned)C4);
      c = (int)((unsigned)i - pix * (unsigned)C4) * 4;
      b = (int)(pix / HW);
      o_in = (long)pix * a.in.sX;
      o_out = (long)pix * a.out.sX;
      o_pre = (long)pix * a.out_pre.sX;
      o_res = (long)pix * a.res.sX;
    } else {
      size_t q = i;
      c = (int)(q % C4) * 4;
      q /= C4;
      const int x = (int)(q % a.W);
      q /= a.W;
      const int y = (int)(q % a.H);
      b = (int)(q / a.H);
      o_in = view_off(a.in, b, y, x);
      o_out = view_off(a.out, b, y, x);
      o_pre = a.out_pre.p ? view_off(a.out_pre, b, y, x) : 0;
      o_res = a.res.p ? view_off(a.res, b, y, x) : 0;
    }
    f32x4 v = *reinterpret_cast<const f32x4*>(a.in.p + o_in + c);
    const f32x4 sv = *reinterpret_cast<const f32x4*>(a.s + c);
    const f32x4 tv = *reinterpret_cast<const f32x4*>(a.t + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __fadd_rn(__fmul_rn(v[k], sv[k]), tv[k]);
    if (a.out_pre.p) *reinterpret_cast<f32x4*>(a.out_pre.p + o_pre + c) = v;
    if (a.film_mul) {
      const f32x4 fm = *reinterpret_cast<const f32x4*>(a.film_mul + (size_t)b * a.film_ld + c);
      const f32x4 fa = *reinterpret_cast<const f32x4*>(a.film_add + (size_t)b * a.film_ld + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = film_preact(v[k], fm[k], fa[k]);
    }
    if (a.relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    if (a.drop_seed) {
      const unsigned base = (unsigned)(i * 4);  // NHWC linear index of element 0 (dense tensor)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (hash_u32(base + k, a.drop_seed) >= drop_thr) ? v[k] * drop_scale : 0.f;
    }
    if (a.res.p) {
      const f32x4 r = *reinterpret_cast<const f32x4*>(a.res.p + o_res + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += r[k];
    }
    *reinterpret_cast<f32x4*>(a.out.p + o_out + c) = v;
  }
}
int dg_affine_act(const AffineActArgs& a, hipStream_t st) {
  if (a.C % 4) { dg_set_error("dg_affine_act: C %% 4 != 0"); return DG_ERR_ARG; }
  const size_t total = (size_t)a.B * a.H * a.W * (a.C / 4);
  const unsigned thr = (unsigned)(a.drop_rate * 4294967296.0);
  auto is_flat = [&](const TView& t) { return !t.p || (t.sY == (long)a.W * t.sX && t.sB == (long)a.H * t.sY); };
  const bool flat = is_flat(a.in) && is_flat(a.out) && is_flat(a.out_pre) && is_flat(a.res) &&
                    (size_t)a.B * a.H * a.W < (1ull << 31);
  if (flat)
    hipLaunchKernelGGL(affine_act_kernel<true>, dim3(t_nblk(total, 8192)), dim3(256), 0, st, a, thr,
                       1.0f / (1.0f - a.drop_rate));
  else
    hipLaunchKernelGGL(affine_act_kernel<false>, dim3(t_nblk(total, 8192)), dim3(256), 0, st, a, thr,
                       1.0f / (1.0f - a.drop_rate));
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

__global__ void bn_bwd_coeffs_kernel(const float* sums, const float* mean, const float* rstd, const float* s,
                                     float invN, float dyscale, float* dgamma, float* dbeta, float* A, float* Bc,
                                     float* Cc, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sd = sums[c] * dyscale, sdx = sums[C + c] * dyscale;
  const float dg = rstd[c] * sdx;   // sdx is already centred: sum dy*(raw - mean)
  dgamma[c] = dg;
  dbeta[c] = sd;
  // draw = s*(dy - dbeta/N - xhat*dgamma/N),  xhat = (raw - mean)*rstd
  A[c] = s[c] * dyscale;
  const float k = s[c] * rstd[c] * dg * invN;
  Bc[c] = -k;
  Cc[c] = -s[c] * sd * invN + k * mean[c];
}
int dg_bn_bwd_coeffs(const float* sums, const float* mean, const float* rstd, const float* s, float invN,
                     float dyscale, float* dgamma, float* dbeta, float* coefA, float* coefB, float* coefC, int C, hipStream_t st) {
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, sums, mean, rstd, s, invN, dyscale,
                     dgamma, dbeta, coefA, coefB, coefC, C);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

template <bool FLAT>
__global__ void axpby_ch_kernel(TView d, TView xv, TView out, int B, int H, int W, int C4, const float* A,
                                const float* Bc, const float* Cc) {
  const size_t total = (size_t)B * H * W * C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c;
    long o_d, o_x, o_o;
    if (FLAT) {
      const unsigned pix = (unsigned)(i / (unsigned)C4);
      c = (int)((unsigned)i - pix * (unsigned)C4) * 4;
      o_d = (long)pix * d.sX;
      o_x = (long)pix * xv.sX;
      o_o = (long)pix * out.sX;
    } else {
      size_t q = i;
      c = (int)(q % C4) * 4;
      q /= C4;
      const int x = (int)(q % W);
      q /= W;
      const int y = (int)(q % H);
      const int b = (int)(q / H);
      o_d = view_off(d, b, y, x);
      o_x = view_off(xv, b, y, x);
      o_o = view_off(out, b, y, x);
    }
    const f32x4 dv = *reinterpret_cast<const f32x4*>(d.p + o_d + c);
    const f32x4 rv = *reinterpret_cast<const f32x4*>(xv.p + o_x + c);
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(A + c);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(Bc + c);
    const f32x4 c4 = *reinterpret_cast<const f32x4*>(Cc + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = fmaf(a4[k], dv[k], fmaf(b4[k], rv[k], c4[k]));
    *reinterpret_cast<f32x4*>(out.p + o_o + c) = o;
  }
}
int dg_axpby_ch(TView d, TView x, TView out, int B, int H, int W, int C, const float* A, const float* Bc,
                const float* Cc, hipStream_t st) {
  const size_t total = (size_t)B * H * W * (C / 4);
  auto is_flat = [&](const TView& t) { return t.sY == (long)W * t.sX && t.sB == (long)H * t.sY; };
  if (is_flat(d) && is_flat(x) && is_flat(out) && (size_t)B * H * W < (1ull << 31))
    hipLaunchKernelGGL(axpby_ch_kernel<true>, dim3(t_nblk(total, 8192)), dim3(256), 0, st, d, x, out, B, H, W, C / 4, A,
                       Bc, Cc);
  else
    hipLaunchKernelGGL(axpby_ch_kernel<false>, dim3(t_nblk(total, 8192)), dim3(256), 0, st, d, x, out, B, H, W, C / 4,
                       A, Bc, Cc);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// softmax + categorical cross-entropy (keras, probabilities path)
// ---------------------------------------------------------------------------
__global__ void softmax_ce4_kernel(const float* __restrict__ logits, const float* __restrict__ onehot,
                                   float* __restrict__ probs, float* __restrict__ dz, float* __restrict__ part,
                                   long P, float invN) {
  __shared__ float sh4[4];
  float lsum = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)P; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 z = *reinterpret_cast<const f32x4*>(logits + i * 4);
    const float m = fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3]));
    f32x4 p;
    float S0 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      p[k] = expf(z[k] - m);
      S0 += p[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] /= S0;
    *reinterpret_cast<f32x4*>(probs + i * 4) = p;
    if (onehot) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(onehot + i * 4);
      const float S = (p[0] + p[1]) + (p[2] + p[3]);
      f32x4 gq;
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float q = p[k] / S;
        const float r = fminf(fmaxf(q, 1e-7f), 1.0f - 1e-7f);
        lsum -= t[k] * logf(r);
        const bool in = (q > 1e-7f) && (q < 1.0f - 1e-7f);
        gq[k] = in ? (-t[k] * invN / q) : 0.f;    // dL/dq
        dot += gq[k] * p[k];
      }
      // q = p/S: dL/dp_j = gq_j/S - dot/S^2 ; softmax: dL/dz_k = p_k (dL/dp_k - sum_j p_j dL/dp_j)
      f32x4 gp;
      float pg = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        gp[k] = gq[k] / S - dot / (S * S);
        pg += p[k] * gp[k];
      }
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = p[k] * (gp[k] - pg);
      *reinterpret_cast<f32x4*>(dz + i * 4) = o;
    }
  }
  if (onehot) {
    lsum = t_block_sum(lsum, sh4);
    if (threadIdx.x == 0) part[blockIdx.x] = lsum;
  }
}
__global__ void sum_small_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ float sh4[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += part[i];
  acc = t_block_sum(acc, sh4);
  if (threadIdx.x == 0) out[0] = acc;
}
int dg_softmax_ce4(const float* logits, const float* onehot, float* probs, float* dz, float* loss_sum, long P,
                   float* scratch, hipStream_t st) {
  const int nb = t_nblk((size_t)P, 1024);
  hipLaunchKernelGGL(softmax_ce4_kernel, dim3(nb), dim3(256), 0, st, logits, onehot, probs, dz, scratch, P,
                     1.0f / (float)P);
  HIPCHECK(hipGetLastError());
  hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(256), 0, st, scratch, nb, loss_sum);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}
int dg_softmax4(const float* logits, float* probs, long P, hipStream_t st) {
  hipLaunchKernelGGL(softmax_ce4_kernel, dim3(t_nblk((size_t)P, 1024)), dim3(256), 0, st, logits, nullptr, probs,
                     nullptr, nullptr, P, 0.f);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// BN over the rows of small matrices (noise MLP), thread = column
// ---------------------------------------------------------------------------
// One 256-thread block per column: rows are strided over the threads, sums go through a fixed-order block reduction.
__global__ void bn_rows_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int C, int ld,
                                   const float* gamma, const float* beta, float eps, float momentum, float corr,
                                   float* mm, float* mv, float* mean, float* rstd, int relu) {
  __shared__ float sh4[4];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int r = threadIdx.x; r < R; r += blockDim.x) s += x[(size_t)r * ld + c];
  const float mu = t_block_sum(s, sh4) / (float)R;
  float q = 0.f;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    const float d = x[(size_t)r * ld + c] - mu;
    q = fmaf(d, d, q);
  }
  const float var = t_block_sum(q, sh4) / (float)R;
  const float rs = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean[c] = mu;
    rstd[c] = rs;
    if (mm) {
      mm[c] = mm[c] * momentum + mu * (1.0f - momentum);
      mv[c] = mv[c] * momentum + var * corr * (1.0f - momentum);
    }
  }
  const float g = gamma[c], bt = beta[c];
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    float v = (x[(size_t)r * ld + c] - mu) * rs * g + bt;
    if (relu) v = fmaxf(v, 0.f);
    y[(size_t)r * ld + c] = v;
  }
}
int dg_bn_rows_fwd(const float* x, float* y, int R, int C, int ld, const float* gamma, const float* beta, float eps,
                   float momentum, float corr, float* moving_mean, float* moving_var, float* mean, float* rstd,
                   int relu, hipStream_t st) {
  hipLaunchKernelGGL(bn_rows_fwd_kernel, dim3(C), dim3(256), 0, st, x, y, R, C, ld, gamma, beta, eps, momentum, corr,
                     moving_mean, moving_var, mean, rstd, relu);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}
__global__ void bn_rows_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                   const float* __restrict__ relu_out, float* __restrict__ dx, int R, int C, int ld,
                                   const float* gamma, const float* mean, const float* rstd, float* dgamma,
                                   float* dbeta) {
  __shared__ float sh4[4];
  const int c = blockIdx.x;
  const float mu = mean[c], rs = rstd[c];
  float sd = 0.f, sdx = 0.f;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    const size_t o = (size_t)r * ld + c;
    const float d = (relu_out && !(relu_out[o] > 0.f)) ? 0.f : dy[o];
    sd += d;
    sdx = fmaf(d, (x[o] - mu) * rs, sdx);
  }
  sd = t_block_sum(sd, sh4);
  sdx = t_block_sum(sdx, sh4);
  if (threadIdx.x == 0) {
    dgamma[c] = sdx;
    dbeta[c] = sd;
  }
  const float s = gamma[c] * rs, invR = 1.0f / (float)R;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    const size_t o = (size_t)r * ld + c;
    const float d = (relu_out && !(relu_out[o] > 0.f)) ? 0.f : dy[o];
    dx[o] = s * (d - sd * invR - (x[o] - mu) * rs * sdx * invR);
  }
}
int dg_bn_rows_bwd(const float* dy, const float* x, const float* relu_out, float* dx, int R, int C, int ld,
                   const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                   hipStream_t st) {
  hipLaunchKernelGGL(bn_rows_bwd_kernel, dim3(C), dim3(256), 0, st, dy, x, relu_out, dx, R, C, ld, gamma, mean, rstd,
                     dgamma, dbeta);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

__global__ void small_gemm_kernel(const float* A, const float* Bm, const float* bias, float* Cm, int M, int K, int N) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(A[(size_t)m * K + k], Bm[(size_t)k * N + n], acc);
  Cm[i] = acc + (bias ? bias[n] : 0.f);
}
int dg_small_gemm(const float* A, const float* Bm, const float* bias, float* Cm, int M, int K, int N, hipStream_t st) {
  hipLaunchKernelGGL(small_gemm_kernel, dim3(cdiv((long)M * N, 256)), dim3(256), 0, st, A, Bm, bias, Cm, M, K, N);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}
// C[k][n] = sum_m A[m][k] D[m][n]: one 256-thread block per output element, rows strided over the threads
__global__ void small_gemm_at_kernel(const float* A, const float* D, float* Cm, int M, int K, int N) {
  __shared__ float sh4[4];
  const int k = blockIdx.x / N, n = blockIdx.x % N;
  float acc = 0.f;
  for (int m = threadIdx.x; m < M; m += blockDim.x) acc = fmaf(A[(size_t)m * K + k], D[(size_t)m * N + n], acc);
  acc = t_block_sum(acc, sh4);
  if (threadIdx.x == 0) Cm[blockIdx.x] = acc;
}
int dg_small_gemm_at(const float* A, const float* D, float* Cm, int M, int K, int N, hipStream_t st) {
  hipLaunchKernelGGL(small_gemm_at_kernel, dim3(K * N), dim3(256), 0, st, A, D, Cm, M, K, N);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}
__global__ void small_gemm_bt_kernel(const float* D, const float* Bm, float* Cm, int M, int K, int N) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)M * K) return;
  const int m = (int)(i / K), k = (int)(i % K);
  float acc = 0.f;
  for (int n = 0; n < N; ++n) acc = fmaf(D[(size_t)m * N + n], Bm[(size_t)k * N + n], acc);
  Cm[i] = acc;
}
int dg_small_gemm_bt(const float* D, const float* Bm, float* Cm, int M, int K, int N, hipStream_t st) {
  hipLaunchKernelGGL(small_gemm_bt_kernel, dim3(cdiv((long)M * K, 256)), dim3(256), 0, st, D, Bm, Cm, M, K, N);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}
__global__ void colsum_small_kernel(const float* x, float* out, int R, int C, int ld) {
  __shared__ float sh4[4];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int r = threadIdx.x; r < R; r += blockDim.x) s += x[(size_t)r * ld + c];
  s = t_block_sum(s, sh4);
  if (threadIdx.x == 0) out[c] = s;
}
int dg_colsum_small(const float* x, float* out, int R, int C, int ld, hipStream_t st) {
  hipLaunchKernelGGL(colsum_small_kernel, dim3(C), dim3(256), 0, st, x, out, R, C, ld);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}
