// DEP-UResNet supervised path: the same U-ResNet as the DEP-GAN generator with a
// 4-class softmax head, trained by my_network.fit / train_on_batch in Keras learning phase 1
// (DEP-UResNet-wNoises-training-4fold.py "UT":355-427 model, 583-606 compile + fit).
//
// What phase 1 changes relative to the GAN closures (which never feed the learning phase):
//   * every BatchNormalization uses the statistics of the current batch and the gradient flows
//     through them; moving_mean / moving_variance are updated with momentum 0.99 (App. B.3);
//   * Dropout(0.25) after conv_10 is active (UT:388);
//   * loss = keras categorical_crossentropy on the softmax probabilities (App. B.9);
//   * Adam(1e-4, beta_1 0.9, beta_2 0.999).
//
// Per conv layer, forward:  igemm (bias only) -> RAW ; per-channel batch moments of RAW ;
//                           y = act(film(RAW*s + t)) (+res, dropout) in one elementwise pass.
//            backward: sums (dy, dy*RAW) -> dgamma, dbeta and the three coefficients of
//                           dRAW = A*dy + B*RAW + C ; then the same wgrad / bwd-data kernels as the GAN path
//                           on dRAW with unscaled weights.
#include <string.h>

#include <string>

#include "model.h"
#include "train_ops.h"

static const float kBnEps = 1e-3f, kBnMomentum = 0.99f, kDropRate = 0.25f;
static const char* kDropLayer = "gen_10";

static const char* kUHeadSfx[NOISE_NHEADS] = {"add_m3", "mul_m3", "add_m2", "mul_m2", "add_m1", "mul_m1", "add",
                                              "mul",    "add_p3", "mul_p3", "add_p2", "mul_p2", "add_p1", "mul_p1"};

struct UNoiseBn {
  float *gamma, *beta, *mm, *mv, *dgamma, *dbeta;
};
static UNoiseBn noise_bn(Net& g, const std::string& nm) {
  UNoiseBn b;
  b.gamma = g.p(nm + "/gamma");
  b.beta = g.p(nm + "/beta");
  b.mm = g.p(nm + "/moving_mean");
  b.mv = g.p(nm + "/moving_variance");
  b.dgamma = g.g(nm + "/gamma");
  b.dbeta = g.g(nm + "/beta");
  return b;
}

int uresnet_build(depgan_ctx* c) {
  const int B = c->cfg.batch;
  size_t maxOut = 0, nsmall = 0;
  for (GLayer& L : c->gl)
    if (L.kind == G_CONV || L.kind == G_FILM || L.kind == G_DECONV) nsmall += 11 * (size_t)((L.Cout + 3) & ~3);
  float* sm = nullptr;
  DGCHECK(dmalloc(c, &sm, nsmall + 64));
  auto take = [&](int n) {
    float* r = sm;
    sm += (n + 3) & ~3;
    return r;
  };
  for (GLayer& L : c->gl) {
    if (L.kind != G_CONV && L.kind != G_FILM && L.kind != G_DECONV) continue;
    const int up = (L.kind == G_DECONV) ? 2 : 1;
    DGCHECK(talloc(c, &L.raw, B, L.H * up, L.W * up, L.Cout));
    const size_t per = (size_t)L.H * up * L.W * up * L.Cout;
    if (per > maxOut) maxOut = per;
    L.bmean = take(L.Cout); L.bvar = take(L.Cout); L.bs = take(L.Cout); L.bt = take(L.Cout); L.brstd = take(L.Cout);
    L.cA = take(L.Cout); L.cB = take(L.Cout); L.cC = take(L.Cout); L.sums = take(2 * L.Cout);
  }
  DGCHECK(dmalloc(c, &c->draw_tmp.p, (size_t)B * maxOut));
  const size_t P = (size_t)B * c->cfg.height * c->cfg.width;
  DGCHECK(dmalloc(c, &c->logits, P * 4));
  DGCHECK(dmalloc(c, &c->dz, P * 4));
  DGCHECK(dmalloc(c, &c->loss_dev, 4));
  DGCHECK(dmalloc(c, &c->ones1k, 1024));
  DGCHECK(dmalloc(c, &c->zeros1k, 1024));
  {
    std::vector<float> one(1024, 1.0f);
    HIPCHECK(hipMemcpy(c->ones1k, one.data(), 1024 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHECK(hipMemset(c->zeros1k, 0, 1024 * sizeof(float)));
  }
  DGCHECK(dmalloc(c, &c->n_mean0, 32)); DGCHECK(dmalloc(c, &c->n_rstd0, 32));
  DGCHECK(dmalloc(c, &c->n_mean1, 32)); DGCHECK(dmalloc(c, &c->n_rstd1, 32));
  DGCHECK(dmalloc(c, &c->n_meanh, 1024)); DGCHECK(dmalloc(c, &c->n_rstdh, 1024));
  DGCHECK(dmalloc(c, &c->n_dl, (size_t)B * 1024)); DGCHECK(dmalloc(c, &c->n_dflat, (size_t)B * 1024));
  DGCHECK(dmalloc(c, &c->n_dl1, (size_t)B * 1024)); DGCHECK(dmalloc(c, &c->n_da0, (size_t)B * 1024));
  DGCHECK(dmalloc(c, &c->n_dl0, (size_t)B * 1024));
  return DG_OK;
}

// ---------------------------------------------------------------------------
// noise MLP, learning phase 1
// ---------------------------------------------------------------------------
static NoiseParams lin_params(depgan_ctx* c) {
  NoiseParams P = c->np;
  P.sh = c->ones1k;
  P.th = c->zeros1k;
  return P;
}

static int u_noise_fwd(depgan_ctx* c, const float* z, int n) {
  Net& g = c->g;
  const int R = n * 32;
  const float corrR = (float)((double)R / ((double)R - (1.0 + (double)kBnEps)));
  const float corrN = (float)((double)n / ((double)n - (1.0 + (double)kBnEps)));
  UNoiseBn b0 = noise_bn(g, "dense_bn_noise_1_add_f0"), b1 = noise_bn(g, "dense_bn_noise_1_add_f1");
  DGCHECK(dg_small_gemm(z, c->np.W0, c->np.b0, c->na.h0, R, 1, 32, c->st));
  DGCHECK(dg_bn_rows_fwd(c->na.h0, c->na.a0, R, 32, 32, b0.gamma, b0.beta, kBnEps, kBnMomentum, corrR, b0.mm, b0.mv,
                         c->n_mean0, c->n_rstd0, 1, c->st));
  DGCHECK(dg_small_gemm(c->na.a0, c->np.W1, c->np.b1, c->na.h1, R, 32, 32, c->st));
  DGCHECK(dg_bn_rows_fwd(c->na.h1, c->na.a1, R, 32, 32, b1.gamma, b1.beta, kBnEps, kBnMomentum, corrR, b1.mm, b1.mv,
                         c->n_mean1, c->n_rstd1, 1, c->st));
  DGCHECK(dg_noise_heads_lin(lin_params(c), c->na.a1, c->na.lin, c->na.heads, n, c->st));
  for (int h = 0; h < NOISE_NHEADS; ++h) {
    UNoiseBn bh = noise_bn(g, std::string("dense_bn_noise_2_") + kUHeadSfx[h]);
    const int c0 = c->np.col0[h], nc = c->np.ncol[h];
    DGCHECK(dg_bn_rows_fwd(c->na.lin + c0, c->na.heads + c0, n, nc, 1024, bh.gamma, bh.beta, kBnEps, kBnMomentum,
                           corrN, bh.mm, bh.mv, c->n_meanh + c0, c->n_rstdh + c0, 0, c->st));
  }
  return DG_OK;
}

static int u_noise_bwd(depgan_ctx* c, const float* z, int n) {
  Net& g = c->g;
  const int R = n * 32;
  NoiseGrads& G = c->ng;
  for (int h = 0; h < NOISE_NHEADS; ++h) {
    UNoiseBn bh = noise_bn(g, std::string("dense_bn_noise_2_") + kUHeadSfx[h]);
    const int c0 = c->np.col0[h], nc = c->np.ncol[h];
    DGCHECK(dg_bn_rows_bwd(c->dheads + c0, c->na.lin + c0, nullptr, c->n_dl + c0, n, nc, 1024, bh.gamma,
                           c->n_meanh + c0, c->n_rstdh + c0, bh.dgamma, bh.dbeta, c->st));
    DGCHECK(dg_colsum_small(c->n_dl + c0, G.dbh[h], n, nc, 1024, c->st));
  }
  DGCHECK(dg_noise_heads_bwd_lin(lin_params(c), G, c->na.a1, c->n_dl, c->n_dflat, n, c->st));
  UNoiseBn b0 = noise_bn(g, "dense_bn_noise_1_add_f0"), b1 = noise_bn(g, "dense_bn_noise_1_add_f1");
  // layer f1: rows = (sample, position), 32 columns
  DGCHECK(dg_bn_rows_bwd(c->n_dflat, c->na.h1, c->na.a1, c->n_dl1, R, 32, 32, b1.gamma, c->n_mean1, c->n_rstd1,
                         b1.dgamma, b1.dbeta, c->st));
  DGCHECK(dg_colsum_small(c->n_dl1, G.db1, R, 32, 32, c->st));
  DGCHECK(dg_small_gemm_at(c->na.a0, c->n_dl1, G.dW1, R, 32, 32, c->st));
  DGCHECK(dg_small_gemm_bt(c->n_dl1, c->np.W1, c->n_da0, R, 32, 32, c->st));
  // layer f0
  DGCHECK(dg_bn_rows_bwd(c->n_da0, c->na.h0, c->na.a0, c->n_dl0, R, 32, 32, b0.gamma, c->n_mean0, c->n_rstd0,
                         b0.dgamma, b0.dbeta, c->st));
  DGCHECK(dg_colsum_small(c->n_dl0, G.db0, R, 32, 32, c->st));
  return dg_small_gemm_at(z, c->n_dl0, G.dW0, R, 1, 32, c->st);
}

// ---------------------------------------------------------------------------
// trunk, learning phase 1
// ---------------------------------------------------------------------------
static int u_bn_act(depgan_ctx* c, GLayer& L, int Ho, int Wo, int n, unsigned drop_seed) {
  ProfScope ps(c, 2, 0.0, "bn fwd: moments + affine / act");
  const double N = (double)n * Ho * Wo;
  DGCHECK(dg_col_moments(L.raw.view(), n, Ho, Wo, L.Cout, L.bmean, L.bvar, c->scratch, c->st));
  DGCHECK(dg_bn_train_prepare(L.gamma, L.beta, L.bmean, L.bvar, kBnEps, kBnMomentum, (float)(N / (N - 1.0)), L.mean,
                              L.var, L.bs, L.bt, L.brstd, L.Cout, c->st));
  AffineActArgs a;
  memset(&a, 0, sizeof(a));
  a.in = L.raw.view();
  a.out = L.out;
  a.out_pre = a.res = null_view();
  a.s = L.bs;
  a.t = L.bt;
  a.relu = 1;
  a.B = n; a.H = Ho; a.W = Wo; a.C = L.Cout;
  a.drop_rate = kDropRate;
  if (L.kind == G_FILM) {
    a.film_mul = c->na.heads + L.col_mul;
    a.film_add = c->na.heads + L.col_add;
    a.film_ld = 1024;
    a.res = L.in;
    a.out_pre = L.u.view();
  }
  if (L.kind == G_CONV && L.name == kDropLayer) a.drop_seed = drop_seed;
  return dg_affine_act(a, c->st);
}

static int u_forward_train(depgan_ctx* c, const float* x, const float* z, int n, unsigned drop_seed) {
  {
    ProfScope ps(c, 2, 0.0, "noise mlp fwd");
    DGCHECK(u_noise_fwd(c, z, n));
  }
  for (size_t i = 0; i < c->gl.size(); ++i) {
    GLayer& L = c->gl[i];
    if (L.kind == G_CONV || L.kind == G_FILM) {
      ConvArgs a;
      memset(&a, 0, sizeof(a));
      zero_ep(&a.ep);
      a.in = (i == 0) ? make_view(const_cast<float*>(x), L.H, L.W, L.Cin) : L.in;
      a.out = L.raw.view();
      a.B = n; a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Cout = L.Cout;
      a.ep.bias = L.b;
      if (L.pf.variant >= 0) {
        a.w = L.wpf[0];
      } else {
        a.w = L.Wt;
        a.wsT = (long)L.Cin * L.Cout; a.wsI = L.Cout; a.wsO = 1; a.flip = 0;
      }
      DGCHECK(conv_launch(c, L.pf, a, 3));
      DGCHECK(u_bn_act(c, L, L.H, L.W, n, drop_seed));
    } else if (L.kind == G_POOL) {
      ProfScope ps(c, 2, 0.0, "maxpool");
      DGCHECK(dg_maxpool(c->gl[L.skip_of].out, L.out, n, L.H / 2, L.W / 2, L.Cout, c->st));
    } else if (L.kind == G_DECONV) {
      for (int t = 0; t < 4 && !deconv_fused(c, L, n); ++t) {
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        zero_ep(&a.ep);
        a.in = L.in;
        a.out = strided2(L.raw.view(), t / 2, t % 2);
        a.B = n; a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Cout = L.Cout;
        a.ep.bias = L.b;
        a.w = L.wpf[t];
        DGCHECK(conv_launch(c, L.pf, a, 1));
      }
      if (deconv_fused(c, L, n)) DGCHECK(deconv_fwd_launch(c, L, L.raw.view(), L.b, nullptr, nullptr, 0, n));
      DGCHECK(u_bn_act(c, L, 2 * L.H, 2 * L.W, n, 0));
    }
  }
  return DG_OK;
}

// 1x1 head to 4 logits (direct kernel: N = 4 is far below an MFMA tile)
static int u_head_logits(depgan_ctx* c, int n) {
  GLayer& L = c->gl.back();
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  zero_ep(&a.ep);
  a.in = L.in;
  a.out = make_view(c->logits, L.H, L.W, 4);
  a.B = n; a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Cout = 4;
  a.ep.bias = L.b;
  a.w = L.Wt;
  a.wsT = (long)L.Cin * 4; a.wsI = 4; a.wsO = 1;
  ConvPlan none = {};
  memset(&none, 0, sizeof(none));
  none.variant = -1;
  return conv_launch(c, none, a, 1);
}

// BN backward of one layer: dy (grad at the BN output, ReLU / FiLM already applied) -> dRAW in draw_tmp,
// dgamma / dbeta written.  dyscale: constant factor still to be applied to dy (dropout's 1/(1-rate)).
static int u_bn_bwd(depgan_ctx* c, GLayer& L, TView dy, int Ho, int Wo, int n, float dyscale, TView* draw) {
  ProfScope ps(c, 2, 0.0, "bn bwd: sums + dRAW");
  const double N = (double)n * Ho * Wo;
  *draw = make_view(c->draw_tmp.p, Ho, Wo, L.Cout);
  DGCHECK(dg_colsum_pair(dy, L.raw.view(), L.bmean, n, Ho, Wo, L.Cout, L.sums, c->scratch, c->st));
  DGCHECK(dg_bn_bwd_coeffs(L.sums, L.bmean, L.brstd, L.bs, (float)(1.0 / N), dyscale, L.dgamma, L.dbeta, L.cA, L.cB,
                           L.cC, L.Cout, c->st));
  return dg_axpby_ch(dy, L.raw.view(), *draw, n, Ho, Wo, L.Cout, L.cA, L.cB, L.cC, c->st);
}

static int u_conv_bwd(depgan_ctx* c, GLayer& L, size_t li, const float* x_user, TView dy, TView res, int n,
                      float dyscale) {
  TView xin = (li == 0) ? make_view(const_cast<float*>(x_user), L.H, L.W, L.Cin) : L.in;
  TView draw;
  DGCHECK(u_bn_bwd(c, L, dy, L.H, L.W, n, dyscale, &draw));
  const ColSum cs = {n, nullptr, L.db, nullptr};
  DGCHECK(wgrad_full(c, 3, xin, draw, n, L.H, L.W, L.Cin, L.Cout, nullptr, L.dW, nullptr, 0, 0, &cs));
  if (li == 0) return DG_OK;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  zero_ep(&a.ep);
  a.in = draw;
  a.out = L.din;
  a.w = L.wpb[0];
  a.B = n; a.H = L.H; a.W = L.W; a.Cin = L.Cout; a.Cout = L.Cin;
  a.ep.res = res;
  a.ep.mask = L.in_mask;
  return conv_launch(c, L.pb, a, 3);
}

static int u_backward(depgan_ctx* c, const float* x, const float* z, int n) {
  for (int i = (int)c->gl.size() - 1; i >= 0; --i) {
    GLayer& L = c->gl[i];
    if (L.kind == G_HEAD) {
      TView dzv = make_view(c->dz, L.H, L.W, 4);
      const ColSum cs = {n, nullptr, L.db, nullptr};
      DGCHECK(wgrad_full(c, 1, L.in, dzv, n, L.H, L.W, L.Cin, 4, nullptr, L.dW, nullptr, 0, 0, &cs));
      ConvArgs a;
      memset(&a, 0, sizeof(a));
      zero_ep(&a.ep);
      a.in = dzv;
      a.out = L.din;
      a.B = n; a.H = L.H; a.W = L.W; a.Cin = 4; a.Cout = L.Cin;
      a.w = L.Wt;                               // W[ci][co] read as (k = co, n = ci)
      a.wsT = (long)L.Cin * 4; a.wsI = 1; a.wsO = 4;
      a.ep.mask = L.in_mask;
      ConvPlan none = {};
      memset(&none, 0, sizeof(none));
      none.variant = -1;
      DGCHECK(conv_launch(c, none, a, 1));
    } else if (L.kind == G_CONV) {
      const float k = (L.name == kDropLayer && c->last_drop_seed) ? 1.0f / (1.0f - kDropRate) : 1.0f;
      DGCHECK(u_conv_bwd(c, L, (size_t)i, x, L.dout, null_view(), n, k));
    } else if (L.kind == G_FILM) {
      TView du = make_view(c->du_tmp.p, L.H, L.W, L.Cout);
      {
        ProfScope ps(c, 2, 0.0, "film bwd");
        DGCHECK(dg_film_bwd(L.dout.p, L.u.p, c->na.heads + L.col_mul, c->na.heads + L.col_add, 1024, du.p,
                            c->dheads + L.col_mul, c->dheads + L.col_add, n, (long)L.H * L.W, L.Cout, c->scratch,
                            c->st));
      }
      DGCHECK(u_conv_bwd(c, L, (size_t)i, x, du, L.dout, n, 1.0f));
    } else if (L.kind == G_POOL) {
      ProfScope ps(c, 2, 0.0, "unpool+mask");
      DGCHECK(dg_unpool_mask(L.pool_dsrc, c->gl[L.skip_of].out, L.pool_skipgrad, L.pool_dst, n, L.H / 2, L.W / 2,
                             L.Cout, c->st));
    } else if (L.kind == G_DECONV) {
      const int Ho = 2 * L.H, Wo = 2 * L.W;
      TView draw;
      DGCHECK(u_bn_bwd(c, L, L.dout, Ho, Wo, n, 1.0f, &draw));
      DGCHECK(deconv_wgrad_all(c, L, draw, n, nullptr, nullptr, nullptr, L.db, nullptr));
      DGCHECK(deconv_bwd_data(c, L, draw, n));
    }
  }
  ProfScope ps(c, 2, 0.0, "noise mlp bwd");
  return u_noise_bwd(c, z, n);
}

// ---------------------------------------------------------------------------
// entry points
// ---------------------------------------------------------------------------
static int u_check(depgan_ctx* c, const char* who) {
  if (!c->train_bn) {
    dg_set_error("%s: the context was not created with nc_out = 4", who);
    return DG_ERR_ARG;
  }
  return DG_OK;
}

// phase 0 (predict / validation): moving statistics, no dropout
static int u_forward_infer(depgan_ctx* c, const float* x, const float* z, int n) {
  DGCHECK(g_forward(c, x, z, n, false));
  return u_head_logits(c, n);
}

int uresnet_predict(depgan_ctx* c, const float* x, const float* z, float* out, int n) {
  DGCHECK(u_forward_infer(c, x, z, n));
  const long P = (long)n * c->cfg.height * c->cfg.width;
  ProfScope ps(c, 2, 0.0, "softmax");
  return dg_softmax4(c->logits, out, P, c->st);
}

static int u_loss_to_host(depgan_ctx* c, long P, float* loss_host) {
  float s = 0.f;
  HIPCHECK(hipMemcpyAsync(&s, c->loss_dev, sizeof(float), hipMemcpyDeviceToHost, c->st));
  HIPCHECK(hipStreamSynchronize(c->st));
  c->last_sums[0] = s;
  c->last_sums[1] = (float)P;
  if (loss_host) *loss_host = s / (float)P;
  return DG_OK;
}

static int u_grads(depgan_ctx* c, const float* x, const float* z, const float* labels, int n, unsigned drop_seed,
                   float* loss_host, bool refresh_bn) {
  DGCHECK(u_check(c, "uresnet_grads"));
  if (n < 1 || n > c->cfg.batch) {
    dg_set_error("uresnet: n must be in [1, batch]");
    return DG_ERR_ARG;
  }
  const long P = (long)n * c->cfg.height * c->cfg.width;
  c->last_drop_seed = drop_seed;
  DGCHECK(u_forward_train(c, x, z, n, drop_seed));
  DGCHECK(u_head_logits(c, n));
  {
    ProfScope ps(c, 2, 0.0, "softmax + cross-entropy");
    DGCHECK(dg_softmax_ce4(c->logits, labels, c->attr.p, c->dz, c->loss_dev, P, c->scratch, c->st));
  }
  DGCHECK(u_backward(c, x, z, n));
  // the forward pass moved the BN moving statistics: the phase-0 affines are stale (the step variant
  // refreshes everything after Adam anyway)
  if (refresh_bn) DGCHECK(refresh_generator_bn(c));
  return u_loss_to_host(c, P, loss_host);
}

extern "C" {

int depgan_uresnet_grads(depgan_ctx* c, const float* x, const float* z, const float* labels, int n,
                         unsigned drop_seed, float* loss_host) {
  return u_grads(c, x, z, labels, n, drop_seed, loss_host, true);
}

int depgan_uresnet_step(depgan_ctx* c, const float* x, const float* z, const float* labels, int n,
                        unsigned drop_seed, float* loss_host) {
  DGCHECK(u_grads(c, x, z, labels, n, drop_seed, loss_host, false));
  return depgan_apply_adam(c, DEPGAN_NET_G);
}

int depgan_uresnet_eval(depgan_ctx* c, const float* x, const float* z, const float* labels, int n,
                        float* loss_host) {
  DGCHECK(u_check(c, "uresnet_eval"));
  if (n < 1 || n > c->cfg.batch) { dg_set_error("uresnet_eval: n must be in [1, batch]"); return DG_ERR_ARG; }
  const long P = (long)n * c->cfg.height * c->cfg.width;
  DGCHECK(u_forward_infer(c, x, z, n));
  {
    ProfScope ps(c, 2, 0.0, "softmax + cross-entropy");
    DGCHECK(dg_softmax_ce4(c->logits, labels, c->attr.p, c->dz, c->loss_dev, P, c->scratch, c->st));
  }
  return u_loss_to_host(c, P, loss_host);
}

}  // extern "C"
