// Small HBM-bound operators around the convolutions (pooling, heads, losses,
// reductions, Adam).  All launch on the caller's stream and never allocate.
#pragma once
#include "common.h"

// MaxPooling2D((2,2)) GT:321.. ; C % 4 == 0
int dg_maxpool(TView in, TView out, int B, int Ho, int Wo, int C, hipStream_t st);
// backward of pool + ReLU mask of the layer feeding the pool:
//   out[full] = ((pixel is first arg-max of its 2x2 window of a) ? dpool : 0) + skip[full]) * (a > 0)
int dg_unpool_mask(TView dpool, TView a, TView skip /*optional*/, TView out, int B, int Ho, int Wo, int C,
                   hipStream_t st);
// adjoint of unpool: out[pooled] = u[arg-max of a]
int dg_gather_pool(TView u, TView a, TView out, int B, int Ho, int Wo, int C, hipStream_t st);

// BN inference affine: s = gamma*rsqrt(var+eps), t = beta - mean*s, rstd = rsqrt(var+eps)
int dg_bn_prepare(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* s,
                  float* t, float* rstd, int C, hipStream_t st);

// generator head: out[p] = act(sum_c a[p][c] w[c] + b)   (gen_segmentation GT:494-495), C % 4 == 0, C <= 256
int dg_head_fwd(const float* a, const float* w, const float* b, float* out, long P, int C, int tanh_act,
                hipStream_t st);
// backward of the head fused with the ReLU mask of the feeding layer:
//   dz[p][c] = dpre[p] * w[c] * (a[p][c] > 0)
int dg_head_bwd(const float* dpre, const float* w, const float* a, float* dz, long P, int C, hipStream_t st);

// critic tail dis_9 + Flatten + Dense(1) (GT:339-342):
//   t9[n][p] = sum_c a[n][p][c] w9[c] + b9 ;  out[n] = sum_p wd[p] t9[n][p] + bd
int dg_critic_tail_fwd(const float* a, const float* w9, const float* b9, const float* wd, const float* bd, float* t9,
                       float* out, int N, int HW, int C, hipStream_t st);
//   dz[n][p][c] = coef(n) * wd[p] * w9[c] * (a[n][p][c] > 0),  coef(n) = coefs[n / per]
int dg_critic_tail_bwd(const float* a, const float* w9, const float* wd, const float* coefs, int per, float* dz, int N,
                       int HW, int C, hipStream_t st);
// tail weight gradients from T[n][p][c] = coef(n) * src[n][p][c]:
//   dw9[c] (+)= sum_{n,p} T wd[p] ; dwd[p] (+)= sum_{n,c} T w9[c] (+ b9 * sum coef if add_bias_terms)
//   db9 (+)= sum_n coef(n) * sum_p wd[p] ; dbd (+)= sum_n coef(n)      (only if add_bias_terms)
// accumulate = 0 overwrites the outputs, 1 adds to them
// scratch: N*(C+HW) floats
int dg_critic_tail_wgrad(const float* src, const float* w9, const float* b9, const float* wd, const float* coefs,
                         int per, int add_bias_terms, int accumulate, float* dw9, float* db9, float* dwd, float* dbd,
                         float* scratch, int N, int HW, int C, hipStream_t st);

// column sums of an NHWC view: out[c] (+)= scale[c] * sum_{b,y,x} v[b,y,x,c]; raw (optional) gets the bare sum.
// scratch: 1024*C floats
// one BatchNorm of a batched dg_bn_prepare_batch launch; mean_copy (optional) receives a copy of the moving mean
struct BnJob {
  const float *gamma, *beta, *mean, *var;
  float *s, *t, *rstd, *mean_copy;
  int C;
};
int dg_bn_prepare_batch(const BnJob* jobs_dev, int njobs, float eps, hipStream_t st);
// every BN-gamma gradient of a network in one launch (one 256-thread block per output channel, jobs walked by blk0)
struct GammaJob {
  const float *W, *dWraw, *bias, *mean, *rstd, *S;
  float* dgamma;
  int K, Cout, oi, Cin;
  int blk0;      // first block of this job in the batched grid
};
int dg_bn_gamma_grad_batch(const GammaJob* jobs_dev, int njobs, int nblocks, hipStream_t st);
int dg_colsum(TView v, int B, int H, int W, int C, const float* scale, float* out, float* raw, int accumulate,
              float* scratch, hipStream_t st);

// out[c] = sum_{pixels q} rowmul[q] * v[q][c]   (q = dense (b,y,x) index)
int dg_colsum_rowmul(TView v, int B, int H, int W, int C, const float* rowmul, float* out, float* scratch,
                     hipStream_t st);
// out[0] = sum in[0..n)   scratch: 1024 floats
int dg_sum(const float* in, size_t n, float* out, float* scratch, hipStream_t st);

// build the 3B critic input batch [real | fake | mixed] (GT:528-538, 555-557). which: 0 = Y2 critic, 1 = DEM critic
int dg_critic_inputs(const float* y2, const float* x, int nicg, const float* attr, const float* ep, float* out, int B,
                     long HW, int which, hipStream_t st);
// fake_y2 = x[...,0] + attr
int dg_add_ch0(const float* x, int nicg, const float* attr, float* out, long P, hipStream_t st);

// gradient penalty (GT:544-545): per-sample norms of g0, GP value, and u0 = delta*(2/B)*(norm-1)/norm * g0
// scratch: B*64 floats ; norms: B floats ; gp_out: 1 float
int dg_gp_u0(const float* g0, float* u0, float* norms, float* gp_out, float delta, int B, long HW, float* scratch,
             hipStream_t st);

// generator loss pieces (GT:576-589): sums[0]=sum|attr-(y2-y1)|, [1]=sum wr, [2]=sum wf, [3]=sum wr*wf
// scratch: 1024*4 floats
int dg_gloss_sums(const float* x, int nicg, const float* y2, const float* attr, float thr, float* sums, long P,
                  float* scratch, hipStream_t st);
// dpre = ( -(g1+g2)/B + (100/P) sign(attr - (y2-y1)) ) * (1 - attr^2)      (GT:576, 592; tanh GT:495)
int dg_g_dpre(const float* x, int nicg, const float* y2, const float* attr, const float* g1, const float* g2,
              float* dpre, int B, long P, hipStream_t st);

// FiLM backward (GT:403-405): v = fmul*u + fadd; dv = dr*(v>0); du = dv*fmul;
// dadd[b,c] = sum_hw dv ; dmul[b,c] = sum_hw dv*u.   scratch: B*64*2*C floats
int dg_film_bwd(const float* dr, const float* u, const float* fmul, const float* fadd, int film_ld, float* du,
                float* dmul, float* dadd, int B, long HW, int C, float* scratch, hipStream_t st);

// BN gamma gradient from the raw weight gradient (see oracle/manual.py):
//   dgamma[co] = rstd[co] * ( sum_k W[k,co]*dWraw[k,co] + (bias[co]-mean[co]) * S[co] )
// W/dWraw are [K][Cout] (oi=0) or [taps][Cout][Cin] (oi=1, K = taps*Cin)
int dg_bn_gamma_grad(const float* W, const float* dWraw, int K, int Cout, int oi, int Cin, const float* bias,
                     const float* mean, const float* rstd, const float* S, float* dgamma, hipStream_t st);

// Keras Adam over a flat arena (App. B.6).  lr_t computed on the host.
// gscale multiplies the gradient first (1/world after a summing all-reduce; 1 otherwise)
int dg_adam(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps,
            float gscale, hipStream_t st);

int dg_scale_copy(const float* in, float* out, size_t n, float s, hipStream_t st);
int dg_mean_groups(const float* in, float* out, int groups, int per, hipStream_t st);
// un-normalised loss pieces of one critic evaluation: out = [sum in[0..B), sum in[B..2B), sum_b (norms[b]-1)^2, B]
int dg_critic_stats(const float* d_out, const float* norms, float* out, int B, hipStream_t st);
// out[g] = sum of group g (g < groups); out[coff] = c0, out[coff+1] = c1
int dg_sum_groups_consts(const float* in, float* out, int groups, int per, int coff, float c0, float c1, hipStream_t st);
// best-of-k noise search on the device (GT:868-877): stats = k x 8 un-normalised pieces
// [sum D_y2(fake), sum D_dem(attr), sum|attr-real_dem|, sum wr, sum wf, sum wr*wf, n, n*H*W]; forms the k total losses
// with the host's algebra (double, rounded to float), takes the FIRST minimum, writes its index to *best and copies
// z_all[best] (zfloats values) to z_out
int dg_best_noise(const float* stats, int k, const float* z_all, long zfloats, int* best, float* z_out, hipStream_t st);

// dst[i] = mask[i] ? float(bf16_rne(src[i])) : src[i]      (bf16-weights mode: master -> compute copy)
int dg_round_bf16_masked(const float* src, const unsigned char* mask, float* dst, size_t n, hipStream_t st);

// evaluation step after the path (GE:616-807)
int dg_eval_accumulate(const float* pred, const float* mask, double* acc, size_t n, hipStream_t st);
int dg_eval_divide(double* acc, size_t n, double d, hipStream_t st);
int dg_eval_counts(const float* x, int nicg, const double* pred, const float* code_real, const float* mask1,
                   const float* wmh1, const float* mask2, const float* wmh2, const float* prob2, size_t npix, double thr,
                   unsigned long long* out_dev, hipStream_t st);
