#pragma once
#include "common.h"

#define NOISE_NHEADS 14

// device pointers into the generator parameter / BN-affine arenas
struct NoiseParams {
  const float *W0, *b0, *s0, *t0, *mean0, *rstd0;  // dense_noise_1_add_f0 (+BN)
  const float *W1, *b1, *s1, *t1, *mean1, *rstd1;  // dense_noise_1_add_f1 (+BN)
  const float* Wh[NOISE_NHEADS];                   // [1024][n] each
  const float* bh[NOISE_NHEADS];
  int col0[NOISE_NHEADS], ncol[NOISE_NHEADS];      // column range of each head in the 1024-wide concat
  const float *sh, *th, *meanh, *rstdh;            // concatenated BN affine of the heads [1024]
};

struct NoiseActs {
  float *h0, *a0, *h1, *a1;  // [B][32][32]
  float *lin, *heads;        // [B][1024]
};

struct NoiseGrads {
  float *dW0, *db0, *dgamma0, *dbeta0;
  float *dW1, *db1, *dgamma1, *dbeta1;
  float* dWh[NOISE_NHEADS];
  float* dbh[NOISE_NHEADS];
  float* dgamma_h[NOISE_NHEADS];
  float* dbeta_h[NOISE_NHEADS];
};

int dg_noise_fwd(const NoiseParams& P, const float* z, NoiseActs A, int B, hipStream_t st);
// scratch: 4*B*1024 floats
int dg_noise_bwd(const NoiseParams& P, const NoiseGrads& G, const float* z, NoiseActs A, const float* dheads,
                 float* scratch, int B, hipStream_t st);

int dg_noise_heads_lin(const NoiseParams& P, const float* flat, float* lin, float* heads, int B, hipStream_t st);
int dg_noise_heads_bwd_lin(const NoiseParams& P, const NoiseGrads& G, const float* flat, const float* dl, float* dflat,
                           int B, hipStream_t st);
