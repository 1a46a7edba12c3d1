// Operators only the learning-phase-1 path needs (DEP-UResNet `fit`):
// batch-statistics BatchNorm forward / backward, Dropout, softmax + categorical cross-entropy.
#pragma once
#include "common.h"

// per-channel batch mean and biased variance of an NHWC view (two passes: mean, then centred squares)
// scratch: 2 * 1024 * C floats
int dg_col_moments(TView v, int B, int H, int W, int C, float* mean, float* var, float* scratch, hipStream_t st);
// sums[0..C) = sum_p d[p][c] ; sums[C..2C) = sum_p d[p][c] * (x[p][c] - mean[c])      scratch: 2 * 1024 * C floats
int dg_colsum_pair(TView d, TView x, const float* mean, int B, int H, int W, int C, float* sums, float* scratch,
                   hipStream_t st);

// training-mode BN bookkeeping for one layer: s = gamma*rsqrt(var+eps), t = beta - mean*s, rstd;
// moving_mean/var <- momentum*moving + (1-momentum)*(mean, var*corr)       
int dg_bn_train_prepare(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                        float momentum, float corr, float* moving_mean, float* moving_var, float* s, float* t,
                        float* rstd, int C, hipStream_t st);

// y = act( film( x*s[c] + t[c] ) ) + res, optional pre-FiLM copy, optional dropout (hash RNG shared with the oracle)
struct AffineActArgs {
  TView in, out, out_pre, res;
  const float *s, *t, *film_mul, *film_add;
  int film_ld, relu;
  int B, H, W, C;
  unsigned drop_seed;   // 0 = no dropout
  float drop_rate;
};
int dg_affine_act(const AffineActArgs& a, hipStream_t st);

// BN backward coefficients from sums = [sum dy, sum dy*(raw-mean)]:
//   dbeta = sum dy ; dgamma = rstd * sum dy*(raw-mean)
//   draw = A*dy + Bc*raw + Cc with A = s, Bc = -s*rstd^2*dgamma'/N ... (see train_ops.hip)
// dyscale: constant still to be applied to dy (dropout's 1/(1-rate) on the kept elements)
int dg_bn_bwd_coeffs(const float* sums, const float* mean, const float* rstd, const float* s, float invN,
                     float dyscale, float* dgamma, float* dbeta, float* coefA, float* coefB, float* coefC, int C, hipStream_t st);
// out[p][c] = A[c]*d[p][c] + Bc[c]*x[p][c] + Cc[c]
int dg_axpby_ch(TView d, TView x, TView out, int B, int H, int W, int C, const float* A, const float* Bc,
                const float* Cc, hipStream_t st);

// softmax + keras categorical cross-entropy on 4-class logits: probs out, dz = dLoss/dlogits (loss = mean over
// pixels), loss_sum[0] = sum over pixels of the per-pixel loss.    scratch: 1024 floats
int dg_softmax_ce4(const float* logits, const float* onehot, float* probs, float* dz, float* loss_sum, long P,
                   float* scratch, hipStream_t st);
int dg_softmax4(const float* logits, float* probs, long P, hipStream_t st);

// ---- noise MLP in training mode: BN over the rows of small [R][C] matrices ----
// y = relu?( gamma*(x-mean)*rstd + beta ), stats over the R rows; also updates the moving stats
int dg_bn_rows_fwd(const float* x, float* y, int R, int C, int ld, const float* gamma, const float* beta, float eps,
                   float momentum, float corr, float* moving_mean, float* moving_var, float* mean, float* rstd,
                   int relu, hipStream_t st);
// dx, dgamma, dbeta from dy (after the ReLU mask of y when relu_out != null)
int dg_bn_rows_bwd(const float* dy, const float* x, const float* relu_out, float* dx, int R, int C, int ld,
                   const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                   hipStream_t st);
// small dense helpers: C[M][N] = A[M][K] @ B[K][N] (+bias) ; At: C[K][N] = A[M][K]^T @ D[M][N] ; Bt: C[M][K] = D[M][N] @ B[K][N]^T
int dg_small_gemm(const float* A, const float* Bm, const float* bias, float* Cm, int M, int K, int N, hipStream_t st);
int dg_small_gemm_at(const float* A, const float* D, float* Cm, int M, int K, int N, hipStream_t st);
int dg_small_gemm_bt(const float* D, const float* Bm, float* Cm, int M, int K, int N, hipStream_t st);
int dg_colsum_small(const float* x, float* out, int R, int C, int ld, hipStream_t st);
