// Step drivers of the DEP-GAN hot path: Gen_UNet2D / Dis_C2D_FCN1 forward
// (GT:316-498), the WGAN-GP critic update with its hand-derived double
// backward (GT:533-571), the generator evaluation and
// update (GT:574-598; A7-A9) and Keras Adam (A10).  The algebra follows
// oracle/manual.py step for step.
#include "model.h"

#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[1024] = "";
void dg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* dg_get_error() { return g_err; }

// ---------------------------------------------------------------------------
// Net
// ---------------------------------------------------------------------------
void Net::add(const std::string& name, std::vector<int> shape, bool trainable) {
  PInfo pi;
  pi.name = name;
  pi.ndim = (int)shape.size();
  pi.size = 1;
  for (int i = 0; i < 4; ++i) {
    pi.shape[i] = i < pi.ndim ? shape[i] : 1;
    pi.size *= pi.shape[i];
  }
  pi.trainable = trainable;
  pi.off = trainable ? nTrain : nNon;
  // keep every tensor 16-byte aligned inside its arena
  const size_t padded = (pi.size + 3) & ~(size_t)3;
  if (trainable) nTrain += padded; else nNon += padded;
  index[name] = (int)params.size();
  params.push_back(pi);
}
float* Net::p(const std::string& name) const {
  const PInfo& pi = params[index.at(name)];
  return (pi.trainable ? (Pq ? Pq : P) : NT) + pi.off;
}
float* Net::g(const std::string& name) const {
  const PInfo& pi = params[index.at(name)];
  return G + pi.off;
}

int dmalloc(depgan_ctx* c, float** p, size_t floats) {
  void* q = nullptr;
  if (floats == 0) floats = 4;
  HIPCHECK(hipMalloc(&q, floats * sizeof(float)));
  HIPCHECK(hipMemset(q, 0, floats * sizeof(float)));
  c->allocs.push_back(q);
  *p = (float*)q;
  return DG_OK;
}
int talloc(depgan_ctx* c, Tn* t, int N, int H, int W, int C) {
  t->H = H;
  t->W = W;
  t->C = C;
  return dmalloc(c, &t->p, (size_t)N * H * W * C);
}
// floats behind the gradients in every GRADS arena: the update's un-normalised loss pieces travel in the same
// all-reduce message as the gradient (nTrain is a multiple of 4, so the tail is 16-byte aligned)
#define STATS_TAIL 8
static int net_alloc(depgan_ctx* c, Net* n) {
  DGCHECK(dmalloc(c, &n->P, n->nTrain));
  DGCHECK(dmalloc(c, &n->G, n->nTrain + STATS_TAIL));   // tail: un-normalised loss pieces, reduced with the gradient
  DGCHECK(dmalloc(c, &n->M, n->nTrain));
  DGCHECK(dmalloc(c, &n->V, n->nTrain));
  DGCHECK(dmalloc(c, &n->NT, n->nNon));
  if (c->cfg.bf16_weights) {
    DGCHECK(dmalloc(c, &n->Pq, n->nTrain));
    std::vector<unsigned char> m(n->nTrain ? n->nTrain : 4, 0);
    for (const PInfo& pi : n->params) {
      const std::string& nm = pi.name;
      if (pi.trainable && nm.size() > 7 && nm.compare(nm.size() - 7, 7, "/kernel") == 0)
        for (size_t i = 0; i < pi.size; ++i) m[pi.off + i] = 1;
    }
    void* q = nullptr;
    HIPCHECK(hipMalloc(&q, m.size()));
    c->allocs.push_back(q);
    HIPCHECK(hipMemcpy(q, m.data(), m.size(), hipMemcpyHostToDevice));
    n->qmask = (unsigned char*)q;
  }
  return DG_OK;
}

// master -> compute copy (bf16-weights mode only)
static int net_requantize(depgan_ctx* c, Net& n) {
  if (!n.Pq || n.nTrain == 0) return DG_OK;
  ProfScope ps(c, 2, 0.0, "bf16 round");
  return dg_round_bf16_masked(n.P, n.qmask, n.Pq, n.nTrain, c->st);
}

// the plan of one convolution of this context: bf16 matrix pipe where configured and covered, else fp32
// N, H, W: the batch and spatial size the layer is planned for (they fix the number of work items of its launches).  N
// is BASELINE's batch (32; 96 for the critics' [real | fake | mixed] passes), NOT the context's: the chunk size decides
// the summation order, and a sample's result must not depend on how many samples share its batch
// (test_bench_size_properties_batch32_256 checks batch 32 against four batches of 8 bit for bit).
static ConvPlan plan_conv(const depgan_ctx* c, int KS, int Cin, int Cout, int N = 0, int H = 0, int W = 0) {
  if (c->cfg.f32_split) return dg_plan_conv_split(KS, Cin, Cout, c->cfg.f32_split == 6 ? 3 : 2);
  if (c->cfg.bf16_mfma) return dg_plan_conv_bf16(KS, Cin, Cout);
  const long items = (long)N * cdiv(H, 16) * cdiv(W, 16) * cdiv(Cout, 32);
  // 3x3 layers the Winograd kernel covers (igemm_wino.hip: 4/9 of the MFMAs of the direct form)
  if (c->winograd && KS == 3 && H > 0 && W > 0 && !((H | W) & 1)) {
    const ConvPlan w = dg_plan_conv_wino(Cin, Cout);
    if (w.variant == 9) return w;
  }
  return dg_plan_conv_items(KS, Cin, Cout, items);
}

// ---------------------------------------------------------------------------
// profiling helpers
// ---------------------------------------------------------------------------

int conv_launch(depgan_ctx* c, const ConvPlan& pl, const ConvArgs& a, int KS) {
  const int ng = a.groups > 1 ? a.groups : 1;
  const double fl = 2.0 * a.B * a.H * a.W * (double)a.Cin * a.Cout * KS * KS * ng;
  char lb[56];
  snprintf(lb, sizeof(lb), "conv%s k%d b%d %dx%d %d->%d%s",
           pl.variant >= 200 ? (pl.bf16 == 3 ? "(bf16x6)" : "(bf16x3)") : (pl.bf16 ? "(bf16)" : ""), KS, a.B, a.H, a.W, a.Cin,
           a.Cout, ng > 1 ? " x4" : "");
  // algorithmic bytes: every operand the epilogue names read once, every result written once, weights once
  const double px = 4.0 * a.B * a.H * a.W;
  const double by = px * a.Cin + ng * (px * a.Cout * (1 + (a.ep.res.p ? 1 : 0) + (a.ep.mask.p ? 1 : 0) +
                                                      (a.ep.out_pre.p ? 1 : 0) + (a.ep.accumulate ? 1 : 0) +
                                                      (a.ep.pool.p ? 0.25 : 0) - (a.ep.head_skip_out ? 1 : 0)) +
                                          4.0 * KS * KS * a.Cin * a.Cout) +
                    (a.ep.head_out ? px + 4.0 * a.Cout : 0.0);
  if (pl.variant >= 0) {
    char kn[48] = "";
    if (c->prof_on) dg_conv_igemm_name(pl, a, kn, sizeof(kn));
    ProfScope ps(c, 0, fl, lb, by, kn);
    return dg_conv_igemm(pl, a, c->st);
  }
  ProfScope ps(c, 2, fl, lb);
  return dg_conv_direct(KS, a, c->st);
}

// Forward of a transposed convolution on the fused four-tap kernel (deconv_fwd.hip): fp32 contexts only -- the bf16 /
// split modes keep their own matrix-pipe kernels through the grouped launch.
bool deconv_fused(const depgan_ctx* c, const GLayer& L, int n) {
  if (c->cfg.bf16_mfma || c->cfg.f32_split) return false;
  return dg_deconv_fwd_supported(n, L.H, L.W, L.Cin, L.Cout, L.in, L.out);
}
int deconv_fwd_launch(depgan_ctx* c, const GLayer& L, TView out, const float* bias, const float* scale,
                      const float* shift, int relu, int n) {
  DeconvArgs d;
  memset(&d, 0, sizeof(d));
  d.in = L.in.p;
  d.w = L.Wt;
  d.out = out;
  d.bias = bias; d.scale = scale; d.shift = shift; d.relu = relu;
  d.H = L.H; d.W = L.W; d.Cin = L.Cin; d.Cout = L.Cout;
  char lb[56];
  snprintf(lb, sizeof(lb), "conv k1 b%d %dx%d %d->%d x4", n, L.H, L.W, L.Cin, L.Cout);
  const double px = 4.0 * n * L.H * L.W;
  ProfScope ps(c, 0, 2.0 * n * L.H * L.W * (double)L.Cin * L.Cout * 4, lb,
               px * L.Cin + 4 * (px * L.Cout + 4.0 * L.Cin * L.Cout), "deconv_fwd_kernel");
  return dg_deconv_fwd(d, n, c->st);
}

// Weight gradient of a transposed convolution (all four taps) and the column sums of its upstream gradient:
// dW = scale . sum, raw (optional) = the unscaled sum, db = colscale . colsum, draw_col (optional) = colsum.
// One fused launch + one reduction where deconv_wgrad.hip covers the layer, else the column-sum pass and four
// per-tap launches of the general kernel.
int deconv_wgrad_all(depgan_ctx* c, const GLayer& L, TView dsrc, int n, const float* scale, float* raw,
                     const float* colscale, float* colout, float* colraw) {
  if (!c->cfg.bf16_mfma && !c->cfg.f32_split &&
      dg_deconv_wgrad_supported(n, L.H, L.W, L.Cin, L.Cout, L.in, dsrc) &&
      dg_deconv_wgrad_part_floats(n, L.H, L.W, L.Cin, L.Cout) <= c->partFloats) {
    DeconvWgradArgs d;
    memset(&d, 0, sizeof(d));
    d.in = L.in.p;
    d.dout = dsrc;
    d.part = c->part;
    d.colpart = c->scratch;
    d.H = L.H; d.W = L.W; d.Cin = L.Cin; d.Cout = L.Cout;
    int nch = 0;
    {
      char lb[56];
      snprintf(lb, sizeof(lb), "wgrad k1 b%d %dx%d %d->%d x4", n, L.H, L.W, L.Cin, L.Cout);
      ProfScope ps(c, 1, 2.0 * n * L.H * L.W * (double)L.Cin * L.Cout * 4, lb);
      DGCHECK(dg_deconv_wgrad(d, n, &nch, c->st));
    }
    ProfScope ps(c, 2, 0.0, "slab reduce");
    return dg_wgrad_finish_rows(c->part, nch, 4, L.Cin, L.Cout, scale, L.dW, raw, 0, 1, c->scratch, 4 * nch, L.Cout,
                                colscale, colout, colraw, c->st);
  }
  {
    ProfScope ps(c, 2, 0.0, "colsum");
    DGCHECK(dg_colsum(dsrc, n, 2 * L.H, 2 * L.W, L.Cout, colscale, colout, colraw, 0, c->scratch, c->st));
  }
  for (int t = 0; t < 4; ++t) {
    const size_t o = (size_t)t * L.Cout * L.Cin;
    DGCHECK(wgrad_full(c, 1, L.in, strided2(dsrc, t / 2, t % 2), n, L.H, L.W, L.Cin, L.Cout, scale, L.dW + o,
                       raw ? raw + o : nullptr, 0, 1));
  }
  return DG_OK;
}

int deconv_bwd_data(depgan_ctx* c, GLayer& L, TView dsrc, int n) {
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  zero_ep(&a.ep);
  a.out = L.din;
  a.B = n; a.H = L.H; a.W = L.W; a.Cout = L.Cin;
  a.ep.mask = L.in_mask;
  if (L.wpb_all) {
    // dIn[p] = sum_t W_t^T dOut[2p + t]: one 1x1 convolution whose K axis gathers the four strided pixel grids
    a.in = strided2(dsrc, 0, 0);
    a.Cin = 4 * L.Cout;
    a.cpt = L.Cout / L.pb.CK;
    for (int t = 0; t < 4; ++t) a.in_run_off[t] = (long)(t / 2) * dsrc.sY + (long)(t % 2) * dsrc.sX;
    a.w = L.wpb_all;
    return conv_launch(c, L.pbf, a, 1);
  }
  a.Cin = L.Cout;
  for (int t = 0; t < 4; ++t) {
    a.in = strided2(dsrc, t / 2, t % 2);
    a.w = L.wpb[t];
    a.ep.accumulate = (t > 0);
    DGCHECK(conv_launch(c, L.pb, a, 1));
  }
  return DG_OK;
}

void zero_ep(Epilogue* e) {
  memset(e, 0, sizeof(*e));
  e->out_pre = e->res = e->mask = e->pool = null_view();
}

TView view_offset(TView v, long samples) {
  v.p += samples * v.sB;
  return v;
}
TView strided2(TView v, int di, int dj) {  // pixel grid (2i+di, 2j+dj)
  TView r = v;
  r.p = v.p + di * v.sY + dj * v.sX;
  r.sY = 2 * v.sY;
  r.sX = 2 * v.sX;
  return r;
}

// weight gradient: slabs + reduction (+ optional BN scale / raw copy / OI layout)
int wgrad_full(depgan_ctx* c, int KS, TView x, TView dy, int N, int H, int W, int Cin, int Cout,
               const float* scale, float* out, float* raw, int accumulate, int oi, const ColSum* cs) {
  WgradArgs a;
  a.x = x;
  a.dy = dy;
  a.part = c->part;
  a.B = N;
  a.H = H;
  a.W = W;
  a.Cin = Cin;
  a.Cout = Cout;
  a.nTiles = a.tilesPerChunk = 0;
  a.colpart = nullptr;
  a.colB = 0;
  int nch = 0;
  const double fl = 2.0 * N * H * W * (double)Cin * Cout * KS * KS;
  char lb[56];
  snprintf(lb, sizeof(lb), "wgrad k%d b%d %dx%d %d->%d", KS, N, H, W, Cin, Cout);
  const bool mfma = Cin % 4 == 0 && Cout % 4 == 0 && Cin >= 8;
  // BASELINE configs[3] on the bf16 matrix pipe: the contraction on v_mfma_f32_32x32x16_bf16 (operands rounded while
  // staged, fp32 accumulation); the column sums keep their own streaming pass
  if (c->cfg.bf16_mfma && mfma && c->wgrad_bf16 && dg_wgrad_bf16_supported(KS, Cin, Cout)) {
    if (dg_wgrad_bf16_part_floats(KS, N, H, W, Cin, Cout) > c->partFloats) {
      dg_set_error("wgrad slab workspace too small");
      return DG_ERR_ARG;
    }
    if (cs) {
      a.colpart = c->scratch;
      a.colB = cs->B;
    }
    snprintf(lb, sizeof(lb), "wgrad(bf16) k%d b%d %dx%d %d->%d", KS, N, H, W, Cin, Cout);
    {
      ProfScope ps(c, 1, fl, lb);
      DGCHECK(dg_wgrad_bf16(KS, a, &nch, c->st));
    }
    ProfScope ps(c, 2, 0.0, "slab reduce");
    return dg_wgrad_finish(c->part, nch, KS * KS, Cin, Cout, scale, out, raw, accumulate, oi, cs ? c->scratch : nullptr,
                           Cout, cs ? cs->scale : nullptr, cs ? cs->out : nullptr, cs ? cs->raw : nullptr, c->st);
  }
  // Column sums of dy (bias / BN-beta gradients) ride in the MFMA weight-gradient kernel, whose B fragments are the
  // dy values anyway (2 FMAs per 18 MFMAs in one workgroup column; with the register-staged kernel of earlier in the
  // round the same idea cost 10 % -- it sat at the VGPR limit of two workgroups per CU).  The edge-layer kernels keep
  // the separate streaming pass.
  if (cs && !mfma) {
    ProfScope ps(c, 2, 0.0, "colsum");
    DGCHECK(dg_colsum(dy, cs->B, H, W, Cout, cs->scale, cs->out, cs->raw, 0, c->scratch, c->st));
  }
  if (mfma) {
    if (dg_wgrad_part_floats(KS, N, H, W, Cin, Cout) > c->partFloats) {
      dg_set_error("wgrad slab workspace too small");
      return DG_ERR_ARG;
    }
    if (cs) {
      a.colpart = c->scratch;
      a.colB = cs->B;
    }
    {
      ProfScope ps(c, 1, fl, lb);
      DGCHECK(dg_wgrad(KS, a, &nch, c->st));
    }
    // slab reduction and the column-sum finish in one launch
    ProfScope ps(c, 2, 0.0, "slab reduce");
    return dg_wgrad_finish(c->part, nch, KS * KS, Cin, Cout, scale, out, raw, accumulate, oi, cs ? c->scratch : nullptr,
                           Cout, cs ? cs->scale : nullptr, cs ? cs->out : nullptr, cs ? cs->raw : nullptr, c->st);
  } else {
    if (dg_wgrad_small_part_floats(KS, N, H, W, Cin, Cout) > c->partFloats) {
      dg_set_error("wgrad slab workspace too small");
      return DG_ERR_ARG;
    }
    ProfScope ps(c, 2, fl, lb);
    DGCHECK(dg_wgrad_small(KS, a, &nch, c->st));
  }
  ProfScope ps(c, 2, 0.0, "slab reduce");
  return dg_wgrad_reduce(c->part, nch, KS * KS, Cin, Cout, scale, out, raw, accumulate, oi, c->st);
}

// ---------------------------------------------------------------------------
// generator construction (GT:349-498)
// ---------------------------------------------------------------------------
static const char* kHeadSfx[NOISE_NHEADS] = {"add_m3", "mul_m3", "add_m2", "mul_m2", "add_m1", "mul_m1", "add",
                                             "mul",    "add_p3", "mul_p3", "add_p2", "mul_p2", "add_p1", "mul_p1"};
static const int kHeadMult[NOISE_NHEADS] = {3, 3, 2, 2, 1, 1, 4, 4, 3, 3, 2, 2, 1, 1};

struct TrunkEnt {
  GKind kind;
  const char* name;
  int ci, co;       // in units of first_fm (ci = -1: nicg; co = -1: 1 output channel)
  const char* aux;  // film key / skip name
};
static const TrunkEnt kTrunk[] = {
    {G_CONV, "gen_0", -1, 1, ""},        {G_FILM, "gen_noise_m1", 1, 1, "m1"}, {G_CONV, "gen_1", 1, 1, ""},
    {G_POOL, "skip1", 0, 0, ""},         {G_CONV, "gen_2", 1, 2, ""},          {G_FILM, "gen_noise_m2", 2, 2, "m2"},
    {G_CONV, "gen_3", 2, 2, ""},         {G_POOL, "skip2", 0, 0, ""},          {G_CONV, "gen_4", 2, 3, ""},
    {G_FILM, "gen_noise_m3", 3, 3, "m3"}, {G_CONV, "gen_5", 3, 3, ""},         {G_POOL, "skip3", 0, 0, ""},
    {G_CONV, "gen_8", 3, 4, ""},         {G_FILM, "gen_noise_p4", 4, 4, ""},   {G_CONV, "gen_9", 4, 4, ""},
    {G_DECONV, "de_gen_9", 4, 4, "skip3"}, {G_CONV, "gen_10", 7, 3, ""},       {G_FILM, "gen_noise_p3", 3, 3, "p3"},
    {G_CONV, "gen_11", 3, 3, ""},        {G_DECONV, "de_gen_11", 3, 3, "skip2"}, {G_CONV, "gen_14", 5, 2, ""},
    {G_FILM, "gen_noise_p2", 2, 2, "p2"}, {G_CONV, "gen_15", 2, 2, ""},        {G_DECONV, "de_gen_15", 2, 2, "skip1"},
    {G_CONV, "gen_16", 3, 1, ""},        {G_FILM, "gen_noise_p1", 1, 1, "p1"}, {G_CONV, "gen_17", 1, 1, ""},
    {G_HEAD, "gen_segmentation", 1, -1, ""},
};
static const int kNTrunk = sizeof(kTrunk) / sizeof(kTrunk[0]);

static void add_bn(Net& n, const std::string& nm, int c) {
  n.add(nm + "/gamma", {c}, true);
  n.add(nm + "/beta", {c}, true);
  n.add(nm + "/moving_mean", {c}, false);
  n.add(nm + "/moving_variance", {c}, false);
}

static int build_generator(depgan_ctx* c) {
  const int fm = c->cfg.first_fm, B = c->cfg.batch, H0 = c->cfg.height, W0 = c->cfg.width;
  Net& g = c->g;
  auto dense = [&](const std::string& nm, int fin, int fout) {
    g.add("dense_" + nm + "/kernel", {fin, fout}, true);
    g.add("dense_" + nm + "/bias", {fout}, true);
    add_bn(g, "dense_bn_" + nm, fout);
  };
  dense("noise_1_add_f0", 1, fm);
  dense("noise_1_add_f1", fm, fm);
  for (int h = 0; h < NOISE_NHEADS; ++h) dense(std::string("noise_2_") + kHeadSfx[h], 32 * fm, fm * kHeadMult[h]);
  size_t bnch = 2 * fm + 32 * fm;  // BN channels so far
  for (int i = 0; i < kNTrunk; ++i) {
    const TrunkEnt& e = kTrunk[i];
    const int ci = e.ci < 0 ? c->cfg.nicg : e.ci * fm, co = e.co < 0 ? c->cfg.nc_out : e.co * fm;
    if (e.kind == G_CONV || e.kind == G_FILM) {
      g.add(std::string("conv2d_") + e.name + "/kernel", {3, 3, ci, co}, true);
      g.add(std::string("conv2d_") + e.name + "/bias", {co}, true);
      add_bn(g, std::string("bn_") + e.name, co);
      bnch += co;
    } else if (e.kind == G_DECONV) {
      g.add(std::string("deconv2d_") + e.name + "/kernel", {2, 2, co, ci}, true);
      g.add(std::string("deconv2d_") + e.name + "/bias", {co}, true);
      add_bn(g, std::string("bn_") + e.name, co);
      bnch += co;
    } else if (e.kind == G_HEAD) {
      g.add(std::string(e.name) + "/kernel", {1, 1, ci, co}, true);
      g.add(std::string(e.name) + "/bias", {co}, true);
    }
  }
  g.lr = c->cfg.lrG;
  DGCHECK(net_alloc(c, &g));
  DGCHECK(dmalloc(c, &c->derived, 3 * bnch + 64));
  DGCHECK(dmalloc(c, &c->heads_mean, 1024));
  float* dv = c->derived;
  auto take = [&](int n) {
    float* r = dv;
    dv += (n + 3) & ~3;
    return r;
  };

  // ---- noise MLP pointers ----
  if (fm != 32) {
    dg_set_error("first_fm must be 32 (noise MLP kernels are sized for 32x32)");
    return DG_ERR_UNSUPPORTED;
  }
  NoiseParams& np = c->np;
  NoiseGrads& ng = c->ng;
  np.W0 = g.p("dense_noise_1_add_f0/kernel");
  np.b0 = g.p("dense_noise_1_add_f0/bias");
  np.mean0 = g.p("dense_bn_noise_1_add_f0/moving_mean");
  np.s0 = take(32); np.t0 = take(32); np.rstd0 = take(32);
  np.W1 = g.p("dense_noise_1_add_f1/kernel");
  np.b1 = g.p("dense_noise_1_add_f1/bias");
  np.mean1 = g.p("dense_bn_noise_1_add_f1/moving_mean");
  np.s1 = take(32); np.t1 = take(32); np.rstd1 = take(32);
  np.sh = take(1024); np.th = take(1024); np.rstdh = take(1024);
  np.meanh = c->heads_mean;
  ng.dW0 = g.g("dense_noise_1_add_f0/kernel"); ng.db0 = g.g("dense_noise_1_add_f0/bias");
  ng.dgamma0 = g.g("dense_bn_noise_1_add_f0/gamma"); ng.dbeta0 = g.g("dense_bn_noise_1_add_f0/beta");
  ng.dW1 = g.g("dense_noise_1_add_f1/kernel"); ng.db1 = g.g("dense_noise_1_add_f1/bias");
  ng.dgamma1 = g.g("dense_bn_noise_1_add_f1/gamma"); ng.dbeta1 = g.g("dense_bn_noise_1_add_f1/beta");
  int col = 0;
  std::map<std::string, int> headcol;
  for (int h = 0; h < NOISE_NHEADS; ++h) {
    const std::string nm = std::string("noise_2_") + kHeadSfx[h];
    np.Wh[h] = g.p("dense_" + nm + "/kernel");
    np.bh[h] = g.p("dense_" + nm + "/bias");
    np.col0[h] = col;
    np.ncol[h] = fm * kHeadMult[h];
    ng.dWh[h] = g.g("dense_" + nm + "/kernel");
    ng.dbh[h] = g.g("dense_" + nm + "/bias");
    ng.dgamma_h[h] = g.g("dense_bn_" + nm + "/gamma");
    ng.dbeta_h[h] = g.g("dense_bn_" + nm + "/beta");
    headcol[nm] = col;
    col += np.ncol[h];
  }
  DGCHECK(dmalloc(c, &c->na.h0, (size_t)B * 1024));
  DGCHECK(dmalloc(c, &c->na.a0, (size_t)B * 1024));
  DGCHECK(dmalloc(c, &c->na.h1, (size_t)B * 1024));
  DGCHECK(dmalloc(c, &c->na.a1, (size_t)B * 1024));
  DGCHECK(dmalloc(c, &c->na.lin, (size_t)B * 1024));
  DGCHECK(dmalloc(c, &c->na.heads, (size_t)B * 1024));
  DGCHECK(dmalloc(c, &c->dheads, (size_t)B * 1024));
  DGCHECK(dmalloc(c, &c->zbuf, (size_t)B * 32));

  // ---- trunk tensors ----
  struct Cat {
    Tn fwd, grad;
    int co_deconv;
  };
  std::map<std::string, Cat> cats;
  for (int i = 0; i < kNTrunk; ++i)
    if (kTrunk[i].kind == G_DECONV) cats[kTrunk[i].aux].co_deconv = kTrunk[i].co * fm;

  c->gl.resize(kNTrunk);
  int H = H0, W = W0;
  TView cur = null_view();
  int producer = -1;  // index of the layer that produced `cur`
  size_t maxFilm = 0;
  for (int i = 0; i < kNTrunk; ++i) {
    const TrunkEnt& e = kTrunk[i];
    GLayer& L = c->gl[i];
    L.kind = e.kind;
    L.name = e.name;
    L.Cin = e.ci < 0 ? c->cfg.nicg : e.ci * fm;
    L.Cout = e.co < 0 ? c->cfg.nc_out : e.co * fm;
    L.H = H;
    L.W = W;
    L.in = cur;
    L.din = L.dout = L.in_mask = null_view();
    // where the gradient wrt this layer's input goes
    if (producer >= 0) {
      const GLayer& Pd = c->gl[producer];
      if (Pd.kind == G_CONV) { L.din = Pd.dout; L.in_mask = Pd.out; }
      else if (Pd.kind == G_FILM) { L.din = Pd.dout; }
      else if (Pd.kind == G_POOL) { L.din = Pd.pool_dsrc; }
      else if (Pd.kind == G_DECONV) {
        Cat& ct = cats[kTrunk[producer].aux];
        L.din = ct.grad.view();
        L.in_mask = ct.fwd.view();
      }
    }
    if (e.kind == G_CONV || e.kind == G_FILM || e.kind == G_DECONV) {
      const std::string pre = (e.kind == G_DECONV) ? "deconv2d_" : "conv2d_";
      L.Wt = g.p(pre + e.name + "/kernel"); L.b = g.p(pre + e.name + "/bias");
      L.dW = g.g(pre + e.name + "/kernel"); L.db = g.g(pre + e.name + "/bias");
      const std::string bn = std::string("bn_") + e.name;
      L.gamma = g.p(bn + "/gamma"); L.beta = g.p(bn + "/beta");
      L.mean = g.p(bn + "/moving_mean"); L.var = g.p(bn + "/moving_variance");
      L.dgamma = g.g(bn + "/gamma"); L.dbeta = g.g(bn + "/beta");
      L.s = take(L.Cout); L.t = take(L.Cout); L.rstd = take(L.Cout);
    }
    if (e.kind == G_CONV) {
      L.pf = plan_conv(c, 3, L.Cin, L.Cout, 32, H, W);
      L.pb = plan_conv(c, 3, L.Cout, L.Cin, 32, H, W);
      if (L.pf.variant >= 0) DGCHECK(dmalloc(c, &L.wpf[0], L.pf.packedFloats));
      if (i > 0 && L.pb.variant >= 0) DGCHECK(dmalloc(c, &L.wpb[0], L.pb.packedFloats));
      const bool skip = (i + 1 < kNTrunk && kTrunk[i + 1].kind == G_POOL);
      if (skip) {
        Cat& ct = cats[kTrunk[i + 1].name];
        DGCHECK(talloc(c, &ct.fwd, B, H, W, ct.co_deconv + L.Cout));
        DGCHECK(talloc(c, &ct.grad, B, H, W, ct.co_deconv + L.Cout));
        L.out = ct.fwd.slice(ct.co_deconv);
        Tn dsk;
        DGCHECK(talloc(c, &dsk, B, H, W, L.Cout));
        L.dout = dsk.view();
      } else {
        Tn a, d;
        DGCHECK(talloc(c, &a, B, H, W, L.Cout));
        DGCHECK(talloc(c, &d, B, H, W, L.Cout));
        L.out = a.view();
        L.dout = d.view();
      }
      cur = L.out;
    } else if (e.kind == G_FILM) {
      L.pf = plan_conv(c, 3, L.Cin, L.Cout, 32, H, W);
      L.pb = plan_conv(c, 3, L.Cout, L.Cin, 32, H, W);
      DGCHECK(dmalloc(c, &L.wpf[0], L.pf.packedFloats));
      DGCHECK(dmalloc(c, &L.wpb[0], L.pb.packedFloats));
      const std::string key = e.aux;
      const std::string sfx = key.empty() ? "" : ("_" + key);
      L.col_mul = headcol["noise_2_mul" + sfx];
      L.col_add = headcol["noise_2_add" + sfx];
      Tn r, d;
      DGCHECK(talloc(c, &r, B, H, W, L.Cout));
      DGCHECK(talloc(c, &d, B, H, W, L.Cout));
      DGCHECK(talloc(c, &L.u, B, H, W, L.Cout));
      L.out = r.view();
      L.dout = d.view();
      if (r.per_sample() > maxFilm) maxFilm = r.per_sample();
      cur = L.out;
    } else if (e.kind == G_POOL) {
      const GLayer& Pc = c->gl[i - 1];
      Cat& ct = cats[e.name];
      Tn p, dp;
      DGCHECK(talloc(c, &p, B, H / 2, W / 2, Pc.Cout));
      DGCHECK(talloc(c, &dp, B, H / 2, W / 2, Pc.Cout));
      L.Cin = L.Cout = Pc.Cout;
      L.skip_of = i - 1;
      L.out = p.view();
      L.pool_dsrc = dp.view();
      L.pool_skipgrad = ct.grad.slice(ct.co_deconv);
      L.pool_dst = Pc.dout;
      H /= 2;
      W /= 2;
      cur = L.out;
    } else if (e.kind == G_DECONV) {
      L.pf = plan_conv(c, 1, L.Cin, L.Cout);
      L.pb = plan_conv(c, 1, L.Cout, L.Cin);
      for (int t = 0; t < 4; ++t) {
        DGCHECK(dmalloc(c, &L.wpf[t], L.pf.packedFloats));
        DGCHECK(dmalloc(c, &L.wpb[t], L.pb.packedFloats));
      }
      L.pbf = plan_conv(c, 1, 4 * L.Cout, L.Cin);
      if (L.pb.variant >= 0 && L.pbf.variant == L.pb.variant && L.pbf.bf16 == L.pb.bf16 && (L.Cout % L.pb.CK) == 0 &&
          L.pbf.packedFloats == 4 * L.pb.packedFloats)
        DGCHECK(dmalloc(c, &L.wpb_all, L.pbf.packedFloats));
      Cat& ct = cats[e.aux];
      L.out = ct.fwd.slice(0);    // (2H, 2W) grid, first Cout channels
      L.dout = ct.grad.slice(0);
      H *= 2;
      W *= 2;
      cur = ct.fwd.view();
    } else if (e.kind == G_HEAD) {
      L.Wt = g.p(std::string(e.name) + "/kernel"); L.b = g.p(std::string(e.name) + "/bias");
      L.dW = g.g(std::string(e.name) + "/kernel"); L.db = g.g(std::string(e.name) + "/bias");
      DGCHECK(talloc(c, &c->attr, B, H, W, c->cfg.nc_out));
      L.out = c->attr.view();
    }
    producer = i;
  }
  DGCHECK(dmalloc(c, &c->du_tmp.p, (size_t)B * maxFilm));
  DGCHECK(dmalloc(c, &c->dpre, (size_t)B * H0 * W0));
  DGCHECK(dmalloc(c, &c->fake_y2, (size_t)B * H0 * W0));
  return DG_OK;
}

// ---------------------------------------------------------------------------
// critic construction (GT:316-345)
// ---------------------------------------------------------------------------
struct DEnt {
  const char* name;
  int k, ci, co;
  bool pool;
};
static const DEnt kDis[11] = {
    {"dis_0a", 5, 1, 16, false}, {"dis_0b", 5, 16, 16, true},  {"dis_1a", 5, 16, 32, false}, {"dis_1b", 5, 32, 32, true},
    {"dis_2", 3, 32, 64, false}, {"dis_3", 3, 64, 64, true},   {"dis_4", 3, 64, 128, false}, {"dis_5", 3, 128, 128, true},
    {"dis_6", 3, 128, 256, false}, {"dis_7", 3, 256, 256, false}, {"dis_8", 3, 256, 256, false},
};

static int build_critics(depgan_ctx* c) {
  const int B = c->cfg.batch, H0 = c->cfg.height, W0 = c->cfg.width;
  c->NB3 = 3 * B;
  c->dl.resize(11);
  int H = H0, W = W0;
  for (int l = 0; l < 11; ++l) {
    DLayer& L = c->dl[l];
    L.name = kDis[l].name;
    L.KS = kDis[l].k;
    L.Cin = kDis[l].ci;
    L.Cout = kDis[l].co;
    L.pool = kDis[l].pool;
    L.H = H;
    L.W = W;
    L.pf = plan_conv(c, L.KS, L.Cin, L.Cout, 96, H, W);     // the critics mostly run on [real | fake | mixed]
    L.pb = plan_conv(c, L.KS, L.Cout, L.Cin, 96, H, W);
    DGCHECK(talloc(c, &c->d_act[l], c->NB3, H, W, L.Cout));
    DGCHECK(talloc(c, &c->d_dz[l], c->NB3, H, W, L.Cout));
    if (L.pool) {
      DGCHECK(talloc(c, &c->d_pool[l], c->NB3, H / 2, W / 2, L.Cout));
      DGCHECK(talloc(c, &c->d_dpool[l], c->NB3, H / 2, W / 2, L.Cout));
      H /= 2;
      W /= 2;
    }
  }
  const int HW = H * W;
  // largest pooled-layer output (dis_0b at full resolution) bounds the u-forward scratch
  DGCHECK(talloc(c, &c->d_ufull, B, H0, W0, 16));
  DGCHECK(dmalloc(c, &c->d_in, (size_t)c->NB3 * H0 * W0));
  DGCHECK(dmalloc(c, &c->d_t9, (size_t)c->NB3 * HW));
  DGCHECK(dmalloc(c, &c->d_out, (size_t)c->NB3));
  DGCHECK(dmalloc(c, &c->g0, (size_t)2 * B * H0 * W0));
  DGCHECK(dmalloc(c, &c->coefs, 8));
  DGCHECK(dmalloc(c, &c->norms, (size_t)B));
  DGCHECK(dmalloc(c, &c->gp, 4));
  const float hc[4] = {-1.0f / B, 1.0f / B, 1.0f, 1.0f};
  HIPCHECK(hipMemcpy(c->coefs, hc, sizeof(hc), hipMemcpyHostToDevice));
  for (int k = 0; k < 2; ++k) {
    DNet& D = c->d[k];
    Net& n = D.net;
    for (int l = 0; l < 11; ++l) {
      n.add(std::string("conv2d_") + kDis[l].name + "/kernel", {kDis[l].k, kDis[l].k, kDis[l].ci, kDis[l].co}, true);
      n.add(std::string("conv2d_") + kDis[l].name + "/bias", {kDis[l].co}, true);
    }
    n.add("dis_9/kernel", {1, 1, 256, 1}, true);
    n.add("dis_9/bias", {1}, true);
    n.add("dense_1/kernel", {HW, 1}, true);
    n.add("dense_1/bias", {1}, true);
    n.lr = c->cfg.lrD;
    DGCHECK(net_alloc(c, &n));
    for (int l = 0; l < 11; ++l) {
      const std::string nm = std::string("conv2d_") + kDis[l].name;
      D.W[l] = n.p(nm + "/kernel"); D.b[l] = n.p(nm + "/bias");
      D.dW[l] = n.g(nm + "/kernel"); D.db[l] = n.g(nm + "/bias");
      D.wpf[l] = D.wpb[l] = nullptr;
      if (c->dl[l].pf.variant >= 0) DGCHECK(dmalloc(c, &D.wpf[l], c->dl[l].pf.packedFloats));
      if (c->dl[l].pb.variant >= 0) DGCHECK(dmalloc(c, &D.wpb[l], c->dl[l].pb.packedFloats));
    }
    D.w9 = n.p("dis_9/kernel"); D.b9 = n.p("dis_9/bias"); D.wd = n.p("dense_1/kernel"); D.bd = n.p("dense_1/bias");
    D.dw9 = n.g("dis_9/kernel"); D.db9 = n.g("dis_9/bias"); D.dwd = n.g("dense_1/kernel"); D.dbd = n.g("dense_1/bias");
  }
  return DG_OK;
}

// ---------------------------------------------------------------------------
// derived state (BN affines, packed weights)
// ---------------------------------------------------------------------------
template <typename T>
static int upload_table(depgan_ctx* c, const std::vector<T>& host, T** dev) {
  float* p = nullptr;
  DGCHECK(dmalloc(c, &p, (host.size() * sizeof(T) + 3) / 4 + 4));
  HIPCHECK(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  *dev = reinterpret_cast<T*>(p);
  return DG_OK;
}

int refresh_generator_bn(depgan_ctx* c) {
  Net& g = c->g;
  const float eps = 1e-3f;  // keras BatchNormalization default
  if (!c->g_bn_jobs) {
    // every BatchNorm of the generator (2 trunk + 14 head BNs of the noise MLP, 24 conv BNs) as one launch
    NoiseParams& np = c->np;
    std::vector<BnJob> jobs;
    jobs.push_back({g.p("dense_bn_noise_1_add_f0/gamma"), g.p("dense_bn_noise_1_add_f0/beta"), np.mean0,
                    g.p("dense_bn_noise_1_add_f0/moving_variance"), (float*)np.s0, (float*)np.t0, (float*)np.rstd0,
                    nullptr, 32});
    jobs.push_back({g.p("dense_bn_noise_1_add_f1/gamma"), g.p("dense_bn_noise_1_add_f1/beta"), np.mean1,
                    g.p("dense_bn_noise_1_add_f1/moving_variance"), (float*)np.s1, (float*)np.t1, (float*)np.rstd1,
                    nullptr, 32});
    for (int h = 0; h < NOISE_NHEADS; ++h) {
      const std::string bn = std::string("dense_bn_noise_2_") + kHeadSfx[h];
      const int c0 = np.col0[h], n = np.ncol[h];
      jobs.push_back({g.p(bn + "/gamma"), g.p(bn + "/beta"), g.p(bn + "/moving_mean"), g.p(bn + "/moving_variance"),
                      (float*)np.sh + c0, (float*)np.th + c0, (float*)np.rstdh + c0, c->heads_mean + c0, n});
    }
    for (size_t i = 0; i < c->gl.size(); ++i) {
      GLayer& L = c->gl[i];
      if (L.kind == G_CONV || L.kind == G_FILM || L.kind == G_DECONV)
        jobs.push_back({L.gamma, L.beta, L.mean, L.var, L.s, L.t, L.rstd, nullptr, L.Cout});
    }
    c->g_n_bn = (int)jobs.size();
    DGCHECK(upload_table(c, jobs, &c->g_bn_jobs));
  }
  ProfScope ps(c, 2, 0.0, "bn affine refresh");
  return dg_bn_prepare_batch(c->g_bn_jobs, c->g_n_bn, eps, c->st);
}

int refresh_generator(depgan_ctx* c) {
  DGCHECK(net_requantize(c, c->g));
  DGCHECK(refresh_generator_bn(c));
  if (!c->g_pack_jobs) {
    std::vector<PackJob> jobs;
    PackJob j;
    for (size_t i = 0; i < c->gl.size(); ++i) {
      GLayer& L = c->gl[i];
      // backward packs carry the phase-0 BN scale; the learning-phase-1 path differentiates through the batch
      // statistics instead and needs them unscaled
      const float* ks = c->train_bn ? nullptr : L.s;
      if (L.kind == G_CONV || L.kind == G_FILM) {
        if (L.wpf[0]) {
          DGCHECK(dg_pack_job(L.pf, L.Wt, L.Cin, L.Cout, 0, 0, 0, nullptr, L.wpf[0], 0, &j));
          jobs.push_back(j);
        }
        if (L.wpb[0]) {
          DGCHECK(dg_pack_job(L.pb, L.Wt, L.Cin, L.Cout, 0, 1, 1, ks, L.wpb[0], 0, &j));
          jobs.push_back(j);
        }
      } else if (L.kind == G_DECONV) {
        for (int t = 0; t < 4; ++t) {
          const float* src = L.Wt + (size_t)t * L.Cout * L.Cin;  // (kh,kw,Cout,Cin)
          DGCHECK(dg_pack_job(L.pf, src, L.Cin, L.Cout, 1, 0, 0, nullptr, L.wpf[t], 0, &j));
          jobs.push_back(j);
          if (L.wpb_all) {
            // panel of channel tile nt, tap t -> [nt][t][chunk][n][k]: the K axis of the fused backward-data launch
            // is (tap, channel); the per-tap panels are written straight into that interleaved layout
            const size_t blk = (size_t)L.pb.nCC * L.pb.NT * L.pb.CK;      // elements
            float* dst = reinterpret_cast<float*>(reinterpret_cast<char*>(L.wpb_all) + t * blk * dg_plan_elem_bytes(L.pb));
            DGCHECK(dg_pack_job(L.pb, src, L.Cin, L.Cout, 1, 1, 0, ks, dst, 4 * blk, &j));
          } else {
            DGCHECK(dg_pack_job(L.pb, src, L.Cin, L.Cout, 1, 1, 0, ks, L.wpb[t], 0, &j));
          }
          jobs.push_back(j);
        }
      }
    }
    c->g_n_pack = (int)jobs.size();
    c->g_pack_blocks = dg_pack_layout(jobs.data(), c->g_n_pack);
    DGCHECK(upload_table(c, jobs, &c->g_pack_jobs));
  }
  ProfScope ps(c, 2, 0.0, "pack weights");
  return dg_pack_weights_batch(c->g_pack_jobs, c->g_n_pack, c->g_pack_blocks, c->st);
}

static int refresh_critic(depgan_ctx* c, DNet& D) {
  if (c->dl.empty()) return DG_OK;  // supervised context: no critics
  DGCHECK(net_requantize(c, D.net));
  if (!D.pack_jobs) {
    std::vector<PackJob> jobs;
    PackJob j;
    for (int l = 0; l < 11; ++l) {
      const DLayer& L = c->dl[l];
      if (D.wpf[l]) {
        DGCHECK(dg_pack_job(L.pf, D.W[l], L.Cin, L.Cout, 0, 0, 0, nullptr, D.wpf[l], 0, &j));
        jobs.push_back(j);
      }
      if (D.wpb[l]) {
        DGCHECK(dg_pack_job(L.pb, D.W[l], L.Cin, L.Cout, 0, 1, 1, nullptr, D.wpb[l], 0, &j));
        jobs.push_back(j);
      }
    }
    D.n_pack = (int)jobs.size();
    D.pack_blocks = dg_pack_layout(jobs.data(), D.n_pack);
    DGCHECK(upload_table(c, jobs, &D.pack_jobs));
  }
  ProfScope ps(c, 2, 0.0, "pack weights");
  return dg_pack_weights_batch(D.pack_jobs, D.n_pack, D.pack_blocks, c->st);
}

// ---------------------------------------------------------------------------
// generator forward / backward
// ---------------------------------------------------------------------------
int g_forward(depgan_ctx* c, const float* x, const float* z, int n, bool store_u) {
  {
    ProfScope ps(c, 2, 0.0, "noise mlp fwd");
    DGCHECK(dg_noise_fwd(c->np, z, c->na, n, c->st));
  }
  bool pooled_by_conv = false, head_by_conv = false;
  for (size_t i = 0; i < c->gl.size(); ++i) {
    GLayer& L = c->gl[i];
    if (L.kind == G_CONV || L.kind == G_FILM) {
      ConvArgs a;
      memset(&a, 0, sizeof(a));
      zero_ep(&a.ep);
      a.in = (i == 0) ? make_view(const_cast<float*>(x), L.H, L.W, L.Cin) : L.in;
      a.out = L.out;
      a.B = n; a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Cout = L.Cout;
      a.ep.bias = L.b; a.ep.scale = L.s; a.ep.shift = L.t; a.ep.relu = 1;
      if (L.kind == G_FILM) {
        a.ep.film_mul = c->na.heads + L.col_mul;
        a.ep.film_add = c->na.heads + L.col_add;
        a.ep.film_ld = 1024;
        a.ep.res = L.in;
        if (store_u) a.ep.out_pre = L.u.view();
      }
      if (L.pf.variant >= 0) {
        a.w = L.wpf[0];
        // the 2x2 max-pool that follows (gen_1 / gen_3 / gen_5, GT:409/422/435) rides in this launch's epilogue
        if (i + 1 < c->gl.size() && c->gl[i + 1].kind == G_POOL && c->gl[i + 1].skip_of == (int)i &&
            !((L.H | L.W) & 1))
          a.ep.pool = c->gl[i + 1].out;
      } else {
        a.w = L.Wt;
        a.wsT = (long)L.Cin * L.Cout; a.wsI = L.Cout; a.wsO = 1; a.flip = 0;
      }
      pooled_by_conv = a.ep.pool.p != nullptr;
      // gen_segmentation (1x1 to one channel, tanh: GT:494-495) rides in gen_17's epilogue where the layer runs on the
      // 8-channel-chunk kernel; forward-only passes then do not store gen_17's own output at all
      head_by_conv = false;
      if (L.kind == G_CONV && i + 1 < c->gl.size() && c->gl[i + 1].kind == G_HEAD && c->cfg.nc_out == 1 &&
          c->head_fused && dg_conv_igemm_head_supported(L.pf, a)) {
        const GLayer& Hd = c->gl[i + 1];
        a.ep.head_w = Hd.Wt; a.ep.head_b = Hd.b; a.ep.head_out = c->attr.p;
        a.ep.head_tanh = 1;
        a.ep.head_skip_out = (!store_u && !c->dbg_capture) ? 1 : 0;
        head_by_conv = true;
      }
      DGCHECK(conv_launch(c, L.pf, a, 3));
    } else if (L.kind == G_POOL) {
      if (pooled_by_conv && L.skip_of == (int)i - 1) continue;
      ProfScope ps(c, 2, 0.0, "maxpool");
      DGCHECK(dg_maxpool(c->gl[L.skip_of].out, L.out, n, L.H / 2, L.W / 2, L.Cout, c->st));
    } else if (L.kind == G_DECONV) {
      // 2x2 / stride-2 transposed convolution = four 1x1 convolutions of the same input, tap (di, dj) writing the
      // pixel grid (2i+di, 2j+dj): one grouped launch, the input tile is fetched once per XCD
      if (deconv_fused(c, L, n)) {
        DGCHECK(deconv_fwd_launch(c, L, L.out, L.b, L.s, L.t, 1, n));
        continue;
      }
      ConvArgs a;
      memset(&a, 0, sizeof(a));
      zero_ep(&a.ep);
      a.in = L.in;
      a.out = strided2(L.out, 0, 0);
      a.B = n; a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Cout = L.Cout;
      a.ep.bias = L.b; a.ep.scale = L.s; a.ep.shift = L.t; a.ep.relu = 1;
      a.groups = 4;
      for (int t = 0; t < 4; ++t) {
        a.w_group[t] = L.wpf[t];
        a.out_group_off[t] = strided2(L.out, t / 2, t % 2).p - a.out.p;
      }
      a.w = L.wpf[0];
      DGCHECK(conv_launch(c, L.pf, a, 1));
    } else if (L.kind == G_HEAD && c->cfg.nc_out == 1) {
      if (head_by_conv) continue;
      ProfScope ps(c, 2, 0.0, "head fwd");
      DGCHECK(dg_head_fwd(L.in.p, L.Wt, L.b, c->attr.p, (long)n * L.H * L.W, L.Cin, 1, c->st));
    }
  }
  return DG_OK;
}

// conv + phase-0 BN backward given dy (grad at the BN output); see _conv_bn_bwd in oracle/manual.py
static int g_conv_bn_bwd(depgan_ctx* c, GLayer& L, size_t li, const float* x_user, TView dy, TView res, int n) {
  TView xin = (li == 0) ? make_view(const_cast<float*>(x_user), L.H, L.W, L.Cin) : L.in;
  // db = s * sum dy, dbeta = sum dy: column sums fused into the weight-gradient launch
  const ColSum cs = {n, L.s, L.db, L.dbeta};
  // the un-scaled gradient stays in the layer's slot of raw_all: g_backward forms every BN gamma gradient at its end
  DGCHECK(wgrad_full(c, 3, xin, dy, n, L.H, L.W, L.Cin, L.Cout, L.s, L.dW, c->raw_all + (L.dW - c->g.G), 0, 0, &cs));
  if (li == 0) return DG_OK;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  zero_ep(&a.ep);
  a.in = dy;
  a.out = L.din;
  a.w = L.wpb[0];
  a.B = n; a.H = L.H; a.W = L.W; a.Cin = L.Cout; a.Cout = L.Cin;
  a.ep.res = res;
  a.ep.mask = L.in_mask;
  return conv_launch(c, L.pb, a, 3);
}

int g_backward(depgan_ctx* c, const float* x, const float* z, int n) {
  for (int i = (int)c->gl.size() - 1; i >= 0; --i) {
    GLayer& L = c->gl[i];
    if (L.kind == G_HEAD) {
      ProfScope ps(c, 2, 0.0, "head bwd");
      const long P = (long)n * L.H * L.W;
      // dW[c] = sum_p dpre[p] a[p][c] ; db = sum dpre
      DGCHECK(dg_colsum_rowmul(L.in, n, L.H, L.W, L.Cin, c->dpre, L.dW, c->scratch, c->st));
      DGCHECK(dg_sum(c->dpre, (size_t)P, L.db, c->scratch, c->st));
      DGCHECK(dg_head_bwd(c->dpre, L.Wt, L.in.p, L.din.p, P, L.Cin, c->st));
    } else if (L.kind == G_CONV) {
      DGCHECK(g_conv_bn_bwd(c, L, (size_t)i, x, L.dout, null_view(), n));
    } else if (L.kind == G_FILM) {
      TView du = make_view(c->du_tmp.p, L.H, L.W, L.Cout);
      {
        ProfScope ps(c, 2, 0.0, "film bwd");
        DGCHECK(dg_film_bwd(L.dout.p, L.u.p, c->na.heads + L.col_mul, c->na.heads + L.col_add, 1024, du.p,
                            c->dheads + L.col_mul, c->dheads + L.col_add, n, (long)L.H * L.W, L.Cout, c->scratch,
                            c->st));
      }
      DGCHECK(g_conv_bn_bwd(c, L, (size_t)i, x, du, L.dout, n));
    } else if (L.kind == G_POOL) {
      ProfScope ps(c, 2, 0.0, "unpool+mask");
      DGCHECK(dg_unpool_mask(L.pool_dsrc, c->gl[L.skip_of].out, L.pool_skipgrad, L.pool_dst, n, L.H / 2, L.W / 2,
                             L.Cout, c->st));
    } else if (L.kind == G_DECONV) {
      DGCHECK(deconv_wgrad_all(c, L, L.dout, n, L.s, c->raw_all + (L.dW - c->g.G), L.s, L.db, L.dbeta));
      DGCHECK(deconv_bwd_data(c, L, L.dout, n));
    }
  }
  {
    // BN-gamma gradients of all 24 layers in one launch: d gamma = rstd (sum_k W dWraw + (b - mu) S), S = d beta
    if (!c->g_gamma_jobs) {
      std::vector<GammaJob> jobs;
      int blk = 0;
      for (size_t i = 0; i < c->gl.size(); ++i) {
        const GLayer& L = c->gl[i];
        if (L.kind != G_CONV && L.kind != G_FILM && L.kind != G_DECONV) continue;
        const bool de = L.kind == G_DECONV;
        jobs.push_back({L.Wt, c->raw_all + (L.dW - c->g.G), L.b, L.mean, L.rstd, L.dbeta, L.dgamma,
                        (de ? 4 : 9) * L.Cin, L.Cout, de ? 1 : 0, L.Cin, blk});
        blk += L.Cout;
      }
      c->g_n_gamma = (int)jobs.size();
      c->g_gamma_blocks = blk;
      DGCHECK(upload_table(c, jobs, &c->g_gamma_jobs));
    }
    ProfScope ps(c, 2, 0.0, "bn gamma grad");
    DGCHECK(dg_bn_gamma_grad_batch(c->g_gamma_jobs, c->g_n_gamma, c->g_gamma_blocks, c->st));
  }
  ProfScope ps(c, 2, 0.0, "noise mlp bwd");
  return dg_noise_bwd(c->np, c->ng, z, c->na, c->dheads, c->scratch, n, c->st);
}

// ---------------------------------------------------------------------------
// critic forward / backward
// ---------------------------------------------------------------------------
static TView d_in_view(depgan_ctx* c, int l, long s0) {  // input tensor of critic layer l (l >= 1)
  const Tn& t = c->dl[l - 1].pool ? c->d_pool[l - 1] : c->d_act[l - 1];
  return view_offset(t.view(), s0);
}

static int d_forward(depgan_ctx* c, DNet& D, const float* img, long s0, int N) {
  const int H0 = c->cfg.height, W0 = c->cfg.width;
  for (int l = 0; l < 11; ++l) {
    const DLayer& L = c->dl[l];
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    zero_ep(&a.ep);
    a.in = (l == 0) ? make_view(const_cast<float*>(img), H0, W0, 1) : d_in_view(c, l, s0);
    a.out = view_offset(c->d_act[l].view(), s0);
    a.B = N; a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Cout = L.Cout;
    a.ep.bias = D.b[l];
    a.ep.relu = 1;
    if (L.pf.variant >= 0) {
      a.w = D.wpf[l];
      if (L.pool && !((L.H | L.W) & 1)) a.ep.pool = view_offset(c->d_pool[l].view(), s0);   // pooled in the epilogue
    } else {
      a.w = D.W[l];
      a.wsT = (long)L.Cin * L.Cout; a.wsI = L.Cout; a.wsO = 1; a.flip = 0;
    }
    DGCHECK(conv_launch(c, L.pf, a, L.KS));
    if (L.pool && !a.ep.pool.p) {
      ProfScope ps(c, 2, 0.0, "maxpool");
      DGCHECK(dg_maxpool(a.out, view_offset(c->d_pool[l].view(), s0), N, L.H / 2, L.W / 2, L.Cout, c->st));
    }
  }
  const DLayer& T = c->dl[10];
  const int HW = T.H * T.W;
  ProfScope ps(c, 2, 0.0, "critic tail fwd");
  return dg_critic_tail_fwd(c->d_act[10].p + s0 * c->d_act[10].per_sample(), D.w9, D.b9, D.wd, D.bd,
                            c->d_t9 + s0 * HW, c->d_out + s0, N, HW, 256, c->st);
}

// backward-data chain.  coefs/per: upstream d out per sample = coefs[(n)/per].
// img_s0/img_n: sample range (relative to s0) for which the gradient wrt the image is produced into g0_out.
static int d_backward_chain(depgan_ctx* c, DNet& D, long s0, int N, const float* coefs, int per, long img_s0,
                            int img_n, float* g0_out) {
  const DLayer& T = c->dl[10];
  const int HW = T.H * T.W;
  {
    ProfScope ps(c, 2, 0.0, "critic tail bwd");
    DGCHECK(dg_critic_tail_bwd(c->d_act[10].p + s0 * c->d_act[10].per_sample(), D.w9, D.wd, coefs, per,
                               c->d_dz[10].p + s0 * c->d_dz[10].per_sample(), N, HW, 256, c->st));
  }
  for (int l = 10; l >= 1; --l) {
    const DLayer& L = c->dl[l];
    const DLayer& Pv = c->dl[l - 1];
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    zero_ep(&a.ep);
    a.in = view_offset(c->d_dz[l].view(), s0);
    a.w = D.wpb[l];
    a.B = N; a.H = L.H; a.W = L.W; a.Cin = L.Cout; a.Cout = L.Cin;
    if (Pv.pool) {
      a.out = view_offset(c->d_dpool[l - 1].view(), s0);
    } else {
      a.out = view_offset(c->d_dz[l - 1].view(), s0);
      a.ep.mask = view_offset(c->d_act[l - 1].view(), s0);
    }
    DGCHECK(conv_launch(c, L.pb, a, L.KS));
    if (Pv.pool) {
      ProfScope ps(c, 2, 0.0, "unpool+mask");
      DGCHECK(dg_unpool_mask(a.out, view_offset(c->d_act[l - 1].view(), s0), null_view(),
                             view_offset(c->d_dz[l - 1].view(), s0), N, Pv.H / 2, Pv.W / 2, Pv.Cout, c->st));
    }
  }
  if (img_n > 0) {
    const DLayer& L = c->dl[0];
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    zero_ep(&a.ep);
    a.in = view_offset(c->d_dz[0].view(), s0 + img_s0);
    a.out = make_view(g0_out, L.H, L.W, 1);
    a.B = img_n; a.H = L.H; a.W = L.W; a.Cin = L.Cout; a.Cout = 1;
    // transposed + flipped access of the HWIO kernel (Cin_f = 1): w(tap', k = co_f, n = 0)
    a.w = D.W[0];
    a.wsT = (long)L.Cin * L.Cout; a.wsI = 1; a.wsO = L.Cout; a.flip = 1;
    ConvPlan none = {};
    none.variant = -1;
    DGCHECK(conv_launch(c, none, a, L.KS));
  }
  return DG_OK;
}

int net_adam(depgan_ctx* c, Net& n, float gscale) {
  n.adam_t += 1;
  const double b1 = c->cfg.beta1, b2 = c->cfg.beta2;
  const double t = (double)n.adam_t;
  const double lr_t = n.lr * sqrt(1.0 - pow(b2, t)) / (1.0 - pow(b1, t));
  ProfScope ps(c, 2, 0.0, "adam");
  return dg_adam(n.P, n.G, n.M, n.V, n.nTrain, (float)lr_t, (float)b1, (float)b2, c->cfg.adam_eps, gscale, c->st);
}

// ---------------------------------------------------------------------------
// closures
// ---------------------------------------------------------------------------
// Enqueues one critic evaluation + its gradients (GT:540-549 / 562-568).  Leaves d loss / d theta_D in the GRADS arena
// and the un-normalised loss pieces [sum D(real), sum D(fake), sum (norm-1)^2, B] in its tail.  No host synchronisation.
static int critic_enqueue(depgan_ctx* c, int which, const float* y2, const float* x, const float* z, const float* ep) {
  if (c->cfg.nc_out != 1) { dg_set_error("the WGAN-GP closures need nc_out == 1"); return DG_ERR_ARG; }
  DNet& D = c->d[which];
  const int B = c->cfg.batch, H0 = c->cfg.height, W0 = c->cfg.width;
  const long HW0 = (long)H0 * W0;
  DGCHECK(g_forward(c, x, z, B, false));
  {
    ProfScope ps(c, 2, 0.0, "critic inputs");
    DGCHECK(dg_critic_inputs(y2, x, c->cfg.nicg, c->attr.p, ep, c->d_in, B, HW0, which, c->st));
  }
  DGCHECK(d_forward(c, D, c->d_in, 0, 3 * B));
  if (c->dbg_capture) {
    // parity-test surface: the mixed pass's activations, before the penalty's u-forward overwrites them in place
    for (int l = 0; l < 11; ++l) {
      const size_t per = c->d_act[l].per_sample();
      if (!c->dbg_mixed[l]) DGCHECK(dmalloc(c, &c->dbg_mixed[l], (size_t)B * per));
      HIPCHECK(hipMemcpyAsync(c->dbg_mixed[l], c->d_act[l].p + (size_t)2 * B * per, (size_t)B * per * sizeof(float),
                              hipMemcpyDeviceToDevice, c->st));
    }
    c->dbg_mixed_valid = true;
  }
  // upstream: real -1/B, fake +1/B, mixed 1 (GT:540-547)
  DGCHECK(d_backward_chain(c, D, 0, 3 * B, c->coefs, B, 2 * B, B, c->g0));
  float* u0 = c->d_in + 2 * B * HW0;
  {
    ProfScope ps(c, 2, 0.0, "gp norms+u0");
    DGCHECK(dg_gp_u0(c->g0, u0, c->norms, nullptr, c->cfg.delta, B, HW0, c->scratch, c->st));
  }
  // u-forward through the masks of the mixed pass, overwriting the mixed slots
  for (int l = 0; l < 11; ++l) {
    const DLayer& L = c->dl[l];
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    zero_ep(&a.ep);
    a.in = (l == 0) ? make_view(u0, H0, W0, 1) : d_in_view(c, l, 2 * B);
    TView act = view_offset(c->d_act[l].view(), 2 * B);
    a.out = L.pool ? c->d_ufull.view() : act;
    if (L.pool) {  // scratch has 16 channels at full size; re-view it for this layer's shape
      a.out = make_view(c->d_ufull.p, L.H, L.W, L.Cout);
    }
    a.ep.mask = act;
    a.B = B; a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Cout = L.Cout;
    if (L.pf.variant >= 0) {
      a.w = D.wpf[l];
    } else {
      a.w = D.W[l];
      a.wsT = (long)L.Cin * L.Cout; a.wsI = L.Cout; a.wsO = 1; a.flip = 0;
    }
    DGCHECK(conv_launch(c, L.pf, a, L.KS));
    if (L.pool) {
      ProfScope ps(c, 2, 0.0, "gather pool");
      DGCHECK(dg_gather_pool(a.out, act, view_offset(c->d_pool[l].view(), 2 * B), B, L.H / 2, L.W / 2, L.Cout,
                             c->st));
    }
  }
  // weight gradients: X = [a_real, a_fake, u], D = [dz_real, dz_fake, g_z] in one launch per layer
  for (int l = 0; l < 11; ++l) {
    const DLayer& L = c->dl[l];
    TView xin = (l == 0) ? make_view(c->d_in, H0, W0, 1) : d_in_view(c, l, 0);
    // bias gradient = column sums over the real and fake samples only (the penalty has no bias gradient, A6)
    const ColSum cs = {2 * B, nullptr, D.db[l], nullptr};
    DGCHECK(wgrad_full(c, L.KS, xin, c->d_dz[l].view(), 3 * B, L.H, L.W, L.Cin, L.Cout, nullptr, D.dW[l], nullptr, 0,
                       0, &cs));
  }
  {
    ProfScope ps(c, 2, 0.0, "critic tail wgrad+stats");
    const DLayer& T = c->dl[10];
    const int HW = T.H * T.W;
    DGCHECK(dg_critic_tail_wgrad(c->d_act[10].p, D.w9, D.b9, D.wd, c->coefs, B, 1, 0, D.dw9, D.db9, D.dwd, D.dbd,
                                 c->scratch, 2 * B, HW, 256, c->st));
    DGCHECK(dg_critic_tail_wgrad(c->d_act[10].p + (size_t)2 * B * c->d_act[10].per_sample(), D.w9, D.b9, D.wd,
                                 c->coefs + 2, B, 0, 1, D.dw9, D.db9, D.dwd, D.dbd, c->scratch, B, HW, 256, c->st));
    DGCHECK(dg_critic_stats(c->d_out, c->norms, D.net.G + D.net.nTrain, B, c->st));
  }
  return DG_OK;
}

// device part of one generator-loss evaluation; leaves the 8 un-normalised pieces [sum D_y2(fake), sum D_dem(attr),
// sum|attr-real_dem|, sum wr, sum wf, sum wr*wf, B, B*H*W] in stats_dev[0..8).  train: also d loss / d theta_G in the
// generator's GRADS arena.  No host synchronisation.
static int g_eval_enqueue(depgan_ctx* c, const float* x, const float* y2, const float* z, bool train,
                          float* stats_dev) {
  const int B = c->cfg.batch, H0 = c->cfg.height, W0 = c->cfg.width;
  const long HW0 = (long)H0 * W0, P = (long)B * HW0;
  DGCHECK(g_forward(c, x, z, B, train));
  {
    ProfScope ps(c, 2, 0.0, "fake_y2");
    DGCHECK(dg_add_ch0(x, c->cfg.nicg, c->attr.p, c->fake_y2, P, c->st));
  }
  DGCHECK(d_forward(c, c->d[0], c->fake_y2, 0, B));
  DGCHECK(d_forward(c, c->d[1], c->attr.p, B, B));
  {
    ProfScope ps(c, 2, 0.0, "g loss sums");
    DGCHECK(dg_sum_groups_consts(c->d_out, stats_dev, 2, B, 6, (float)B, (float)P, c->st));
    DGCHECK(dg_gloss_sums(x, c->cfg.nicg, y2, c->attr.p, c->cfg.im_thresh, stats_dev + 2, P, c->scratch, c->st));
  }
  if (train) {
    // d loss / d attr needs dD/dimage of both critics with upstream 1 per sample (GT:592)
    DGCHECK(d_backward_chain(c, c->d[0], 0, B, c->coefs + 2, B, 0, B, c->g0));
    DGCHECK(d_backward_chain(c, c->d[1], B, B, c->coefs + 2, B, 0, B, c->g0 + P));
    {
      ProfScope ps(c, 2, 0.0, "g dpre");
      DGCHECK(dg_g_dpre(x, c->cfg.nicg, y2, c->attr.p, c->g0, c->g0 + P, c->dpre, B, P, c->st));
    }
    DGCHECK(g_backward(c, x, z, B));
  }
  return DG_OK;
}

// host part: the six reported scalars from the 8 un-normalised pieces (GT:576-592).  Kept statement for statement
// identical to g_total_loss_dev() in ops.hip (the device's arg-min must be the host's).
static void g_loss_from_sums(const float s[8], float out[6]) {
#pragma clang fp contract(off)
  const double n = s[6], npix = s[7];
  const double lf = (double)s[0] / n, lfd = (double)s[1] / n;
  const double m1 = 100.0 * (double)s[2] / npix;                                            // GT:576
  const double dv = (double)s[3] / 1000.0 - (double)s[4] / 1000.0;
  const double m3 = 100.0 * dv * dv;                                                        // GT:587-589
  const double dice = (2.0 * (double)s[5] + 1e-7) / ((double)s[3] + (double)s[4] + 1e-7);  // GT:153-157
  const double m4 = 1.0 - dice;                                                             // GT:583
  out[0] = (float)(-lf - lfd + m1 + m3 + m4);                                               // GT:592
  out[1] = (float)lf;
  out[2] = (float)lfd;
  out[3] = (float)m1;
  out[4] = (float)m3;
  out[5] = (float)m4;
}
static void critic_from_sums(const float s[4], float out[2]) {   // GT:540-541
  out[0] = (float)((double)s[0] / (double)s[3]);
  out[1] = (float)((double)s[1] / (double)s[3]);
}

// ---- update plumbing: [all-reduce] -> async fetch of the loss pieces -> Adam -> derived state ----
#define HOST_STATS_FLOATS (8 * (DEPGAN_MAX_CRITIC_STEPS + DEPGAN_MAX_MULTI + 2))

// ---- direct RCCL binding: the few entry points, resolved at run time (no link dependency) ----
namespace {
struct NcclId { char internal[DEPGAN_RCCL_ID_BYTES]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
struct RcclApi {
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int) = nullptr;      // the id travels BY VALUE
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*CommUserRank)(void*, int*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
const int kNcclFloat = 7, kNcclSum = 0;   // ncclFloat32, ncclSum (rccl.h)

RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api.ok ? &api : nullptr;
  tried = true;
  // the copy that is already in the process first (PyTorch-ROCm loads its own librccl with its own HIP runtime: a
  // second copy would talk to another runtime), then the system's
  void* h = dlsym(RTLD_DEFAULT, "ncclAllReduce") ? RTLD_DEFAULT : nullptr;
  if (!h) {
    const char* e = getenv("DEPGAN_RCCL_LIB");
    const char* names[] = {e, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names)
      if (nm && (h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
  }
  if (!h) { dg_set_error("RCCL: librccl.so is not loaded and cannot be opened (%s)", dlerror()); return nullptr; }
  auto sym = [&](const char* n) { return dlsym(h, n); };
  api.GetUniqueId = (int (*)(NcclId*))sym("ncclGetUniqueId");
  api.CommInitRank = (int (*)(void**, int, NcclId, int))sym("ncclCommInitRank");
  api.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
  api.CommCount = (int (*)(void*, int*))sym("ncclCommCount");
  api.CommUserRank = (int (*)(void*, int*))sym("ncclCommUserRank");
  api.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclAllReduce");
  api.Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclBroadcast");
  api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
  api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.CommCount && api.CommUserRank && api.AllReduce &&
           api.Broadcast;
  if (!api.ok) dg_set_error("RCCL: librccl.so lacks one of the entry points this binding needs");
  return api.ok ? &api : nullptr;
}
int rccl_check(RcclApi* a, int rc, const char* what) {
  if (rc == 0) return DG_OK;
  dg_set_error("RCCL: %s failed: %s", what, a->GetErrorString ? a->GetErrorString(rc) : "?");
  return DG_ERR_HIP;
}
}  // namespace

static int dp_reduce(depgan_ctx* c, float* dev, long n) {
  if (c->rccl_comm) {
    ProfScope ps(c, 2, 0.0, "all-reduce");
    RcclApi* a = rccl_api();
    if (!a) return DG_ERR_HIP;
    c->rccl_issued += 1;
    return rccl_check(a, a->AllReduce(dev, dev, (size_t)n, kNcclFloat, kNcclSum, c->rccl_comm, c->st), "ncclAllReduce");
  }
  if (!c->ar_fn) return DG_OK;
  ProfScope ps(c, 2, 0.0, "all-reduce");
  const int rc = c->ar_fn(c->ar_user, dev, n, (void*)c->st);
  if (rc != 0) {
    dg_set_error("the all-reduce hook failed with status %d", rc);
    return DG_ERR_HIP;
  }
  return DG_OK;
}
static int fetch_async(depgan_ctx* c, const float* dev, int n, int slot_floats) {
  HIPCHECK(hipMemcpyAsync(c->host_stats + slot_floats, dev, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->st));
  return DG_OK;
}
static int refresh_net(depgan_ctx* c, int net) { return depgan_weights_changed(c, net); }
// after grads + loss pieces are in the GRADS arena of `net`: reduce across ranks, fetch the pieces into host slot, update
static int finish_update(depgan_ctx* c, int net, Net& n, int nstats, int slot_floats, bool update) {
  if (update) DGCHECK(dp_reduce(c, n.G, (long)n.nTrain + STATS_TAIL));
  DGCHECK(fetch_async(c, n.G + n.nTrain, nstats, slot_floats));
  if (!update) return DG_OK;
  DGCHECK(net_adam(c, n, (c->ar_fn || c->rccl_comm) ? 1.0f / (float)c->world : 1.0f));
  return refresh_net(c, net);
}

static int critic_impl(depgan_ctx* c, int net, const float* y2, const float* x, const float* z, const float* ep,
                       float out[2], bool update) {
  if (net != DEPGAN_NET_D_Y2 && net != DEPGAN_NET_D_DEM) { dg_set_error("not a critic id"); return DG_ERR_ARG; }
  DGCHECK(critic_enqueue(c, net - 1, y2, x, z, ep));
  DGCHECK(finish_update(c, net, c->d[net - 1].net, 4, 0, update));
  HIPCHECK(hipStreamSynchronize(c->st));
  critic_from_sums(c->host_stats, out);
  memcpy(c->last_sums, c->host_stats, 4 * sizeof(float));
  return DG_OK;
}

static int g_impl(depgan_ctx* c, const float* x, const float* y2, const float* z, float out[6], bool train,
                  bool update) {
  if (c->cfg.nc_out != 1) { dg_set_error("the WGAN-GP closures need nc_out == 1"); return DG_ERR_ARG; }
  float* stats = c->g.G + c->g.nTrain;
  DGCHECK(g_eval_enqueue(c, x, y2, z, train, stats));
  if (!train) DGCHECK(dp_reduce(c, stats, 8));   // netG_no_update reports global scalars
  DGCHECK(finish_update(c, DEPGAN_NET_G, c->g, 8, 0, update));
  HIPCHECK(hipStreamSynchronize(c->st));
  g_loss_from_sums(c->host_stats, out);
  memcpy(c->last_sums, c->host_stats, 8 * sizeof(float));
  return DG_OK;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static Net* pick_net(depgan_ctx* c, int net) {
  if (net == DEPGAN_NET_G) return &c->g;
  if (net == DEPGAN_NET_D_Y2) return &c->d[0].net;
  if (net == DEPGAN_NET_D_DEM) return &c->d[1].net;
  dg_set_error("unknown net id %d", net);
  return nullptr;
}

extern "C" {

const char* depgan_last_error(void) { return dg_get_error(); }

int depgan_abi_version(void) { return DEPGAN_ABI_VERSION; }
size_t depgan_config_size(void) { return sizeof(depgan_config); }

int depgan_create(const depgan_config* cfg, depgan_ctx** out) {
  if (!cfg || !out) { dg_set_error("depgan_create: null argument"); return DG_ERR_ARG; }
  if (cfg->struct_size != (int)sizeof(depgan_config)) {
    dg_set_error("depgan_create: depgan_config.struct_size is %d, this library (ABI %d) expects %d -- the caller was "
                 "built against another include/depgan.h", cfg->struct_size, DEPGAN_ABI_VERSION,
                 (int)sizeof(depgan_config));
    return DG_ERR_ARG;
  }
  if (cfg->f32_split != 0 && ((cfg->f32_split != 3 && cfg->f32_split != 6) || cfg->bf16_weights || cfg->bf16_mfma ||
                               (cfg->nc_out != 0 && cfg->nc_out != 1))) {
    dg_set_error("depgan_create: f32_split must be 0, 3 or 6, without bf16_weights / bf16_mfma, nc_out = 1");
    return DG_ERR_ARG;
  }
  if (cfg->bf16_mfma && (!cfg->bf16_weights || (cfg->nc_out != 0 && cfg->nc_out != 1))) {
    dg_set_error("depgan_create: bf16_mfma needs bf16_weights = 1 and the DEP-GAN generator (nc_out = 1)");
    return DG_ERR_ARG;
  }
  if (cfg->batch < 1 || cfg->height % 16 || cfg->width % 16 || cfg->height < 16 || cfg->width < 16 || cfg->nicg < 1 ||
      cfg->nicg > 2) {
    dg_set_error("depgan_create: need batch >= 1, height/width multiples of 16, nicg in {1,2}");
    return DG_ERR_ARG;
  }
  depgan_ctx* c = new depgan_ctx();
  c->cfg = *cfg;
  if (hipGetDevice(&c->device) != hipSuccess) c->device = 0;
  {
    // DEPGAN_WGRAD_BF16=0 (read when the context is created): a bf16_mfma context keeps the fp32 weight-gradient kernel
    // -- the A/B switch of tests/test_gpu_model.py::test_config4_bf16_matrix_pipe
    const char* e = getenv("DEPGAN_WGRAD_BF16");
    c->wgrad_bf16 = !(e && atoi(e) == 0);
    // same pattern: DEPGAN_HEAD_FUSED=0 keeps gen_segmentation in its own launch (the A/B switch of
    // tests/test_gpu_model.py::test_fused_head_matches_the_separate_launch)
    const char* hf = getenv("DEPGAN_HEAD_FUSED");
    c->head_fused = !(hf && atoi(hf) == 0);
    // DEPGAN_WINOGRAD=0: every 3x3 convolution on the direct implicit-GEMM kernel (A/B switch of the parity tests and
    // of bench.py's `direct_conv` line)
    const char* wn = getenv("DEPGAN_WINOGRAD");
    c->winograd = !(wn && atoi(wn) == 0);
  }
  if (c->cfg.nc_out <= 0) c->cfg.nc_out = 1;
  if (c->cfg.nc_out != 1 && c->cfg.nc_out != 4) {
    dg_set_error("depgan_create: nc_out must be 1 (DEP-GAN) or 4 (DEP-UResNet)");
    delete c;
    return DG_ERR_ARG;
  }
  c->train_bn = c->cfg.nc_out != 1;
  memset(c->last_sums, 0, sizeof(c->last_sums));
  int rc = build_generator(c);
  if (rc == DG_OK && !c->train_bn) rc = build_critics(c);   // the supervised path has no critics
  if (rc == DG_OK && c->train_bn) rc = uresnet_build(c);
  if (rc == DG_OK) {
    // slab workspace: the largest weight-gradient call of either network
    size_t mx = 0;
    const int B = cfg->batch;
    for (size_t i = 0; i < c->gl.size(); ++i) {
      const GLayer& L = c->gl[i];
      size_t f = 0;
      if (L.kind == G_CONV || L.kind == G_FILM) {
        f = (L.Cin >= 8) ? dg_wgrad_part_floats(3, B, L.H, L.W, L.Cin, L.Cout)
                         : dg_wgrad_small_part_floats(3, B, L.H, L.W, L.Cin, L.Cout);
        if (cfg->bf16_mfma && L.Cin >= 8 && dg_wgrad_bf16_supported(3, L.Cin, L.Cout)) {
          const size_t fb = dg_wgrad_bf16_part_floats(3, B, L.H, L.W, L.Cin, L.Cout);
          if (fb > f) f = fb;
        }
      }
      else if (L.kind == G_DECONV || (L.kind == G_HEAD && c->train_bn))
        f = dg_wgrad_part_floats(1, B, L.H, L.W, L.Cin, L.Cout);
      if (L.kind == G_DECONV && dg_deconv_wgrad_supported(B, L.H, L.W, L.Cin, L.Cout, L.in, L.dout)) {
        const size_t f4 = dg_deconv_wgrad_part_floats(B, L.H, L.W, L.Cin, L.Cout);   // four taps in one slab
        if (f4 > f) f = f4;
      }
      if (f > mx) mx = f;
    }
    for (size_t l = 0; l < c->dl.size(); ++l) {
      const DLayer& L = c->dl[l];
      size_t f = (L.Cin >= 8) ? dg_wgrad_part_floats(L.KS, 3 * B, L.H, L.W, L.Cin, L.Cout)
                              : dg_wgrad_small_part_floats(L.KS, 3 * B, L.H, L.W, L.Cin, L.Cout);
      if (cfg->bf16_mfma && L.Cin >= 8 && dg_wgrad_bf16_supported(L.KS, L.Cin, L.Cout)) {
        const size_t fb = dg_wgrad_bf16_part_floats(L.KS, 3 * B, L.H, L.W, L.Cin, L.Cout);
        if (fb > f) f = fb;
      }
      if (f > mx) mx = f;
    }
    c->partFloats = mx;
    rc = dmalloc(c, &c->part, mx);
  }
  if (rc == DG_OK) rc = dmalloc(c, &c->raw, (size_t)9 * 256 * 256);
  if (rc == DG_OK && !c->train_bn) rc = dmalloc(c, &c->raw_all, c->g.nTrain);
  if (rc == DG_OK) rc = dmalloc(c, &c->Sraw, 256);
  if (rc == DG_OK) rc = dmalloc(c, &c->scratch, (size_t)(1 << 20) + (size_t)cfg->batch * 20000);
  if (rc == DG_OK) rc = dmalloc(c, &c->scal, 16);
  if (rc == DG_OK) rc = dmalloc(c, &c->scal_multi, 8 * DEPGAN_MAX_MULTI);
  if (rc == DG_OK) rc = dmalloc(c, &c->z_best, (size_t)cfg->batch * 32);
  if (rc == DG_OK) {
    float* p = nullptr;
    rc = dmalloc(c, &p, 4);
    c->best_dev = reinterpret_cast<int*>(p);
  }
  if (rc == DG_OK && hipHostMalloc((void**)&c->host_stats, (HOST_STATS_FLOATS + 4) * sizeof(float)) != hipSuccess) {
    dg_set_error("depgan_create: hipHostMalloc failed");
    rc = DG_ERR_HIP;
  }
  if (rc == DG_OK) c->best_host = reinterpret_cast<int*>(c->host_stats + HOST_STATS_FLOATS);
  if (rc != DG_OK) {
    depgan_destroy(c);
    return rc;
  }
  *out = c;
  return DG_OK;
}

void depgan_destroy(depgan_ctx* c) {
  if (!c) return;
  hipDeviceSynchronize();
  depgan_rccl_shutdown(c);
  for (void* p : c->allocs) hipFree(p);
  if (c->host_stats) hipHostFree(c->host_stats);
  for (ProfRec& r : c->recs) {
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  delete c;
}

int depgan_set_stream(depgan_ctx* c, void* s) {
  c->st = (hipStream_t)s;
  return DG_OK;
}

int depgan_rccl_unique_id(void* id_out) {
  RcclApi* a = rccl_api();
  if (!a || !id_out) { if (a) dg_set_error("rccl_unique_id: null argument"); return a ? DG_ERR_ARG : DG_ERR_HIP; }
  NcclId id;
  DGCHECK(rccl_check(a, a->GetUniqueId(&id), "ncclGetUniqueId"));
  memcpy(id_out, &id, sizeof(id));
  return DG_OK;
}

int depgan_rccl_shutdown(depgan_ctx* c) {
  if (!c || !c->rccl_comm) return DG_OK;
  RcclApi* a = rccl_api();
  hipStreamSynchronize(c->st);
  if (a) a->CommDestroy(c->rccl_comm);
  c->rccl_comm = nullptr;
  c->world = c->ar_fn ? c->world : 1;
  return DG_OK;
}

int depgan_rccl_init(depgan_ctx* c, const void* id, int rank, int world) {
  if (!c || !id || world < 1 || rank < 0 || rank >= world) { dg_set_error("rccl_init: need 0 <= rank < world and an id"); return DG_ERR_ARG; }
  // The loss pieces travel as float32 in the tail of the gradient all-reduce; the voxel counts among them (sum wr, sum
  // wf, sum wr*wf, GT:581-589) are exact only up to 2^24 per GLOBAL batch
  if ((double)world * c->cfg.batch * c->cfg.height * c->cfg.width > 16777216.0) {
    dg_set_error("rccl_init: world x batch x H x W = %.0f exceeds 2^24: the float32 voxel counts of the global batch "
                 "(M3 / M4, GT:581-589) would round", (double)world * c->cfg.batch * c->cfg.height * c->cfg.width);
    return DG_ERR_ARG;
  }
  RcclApi* a = rccl_api();
  if (!a) return DG_ERR_HIP;
  depgan_rccl_shutdown(c);
  HIPCHECK(hipSetDevice(c->device));
  NcclId nid;
  memcpy(&nid, id, sizeof(nid));
  void* comm = nullptr;
  DGCHECK(rccl_check(a, a->CommInitRank(&comm, world, nid, rank), "ncclCommInitRank"));
  c->rccl_comm = comm;
  c->rccl_issued = 0;
  c->world = world;
  c->ar_fn = nullptr;
  c->ar_user = nullptr;
  return DG_OK;
}

int depgan_rccl_broadcast(depgan_ctx* c, float* dev, long n, int root) {
  if (!c || !c->rccl_comm) { dg_set_error("rccl_broadcast: no communicator (depgan_rccl_init)"); return DG_ERR_ARG; }
  if (n <= 0) return DG_OK;
  RcclApi* a = rccl_api();
  return rccl_check(a, a->Broadcast(dev, dev, (size_t)n, kNcclFloat, root, c->rccl_comm, c->st), "ncclBroadcast");
}

int depgan_rccl_info(depgan_ctx* c, int* nranks, int* rank, long* issued) {
  if (!c || !c->rccl_comm) { dg_set_error("rccl_info: no communicator (depgan_rccl_init)"); return DG_ERR_ARG; }
  RcclApi* a = rccl_api();
  int n = 0, r = -1;
  DGCHECK(rccl_check(a, a->CommCount(c->rccl_comm, &n), "ncclCommCount"));
  DGCHECK(rccl_check(a, a->CommUserRank(c->rccl_comm, &r), "ncclCommUserRank"));
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  if (issued) *issued = c->rccl_issued;
  return DG_OK;
}

int depgan_set_allreduce(depgan_ctx* c, depgan_allreduce_fn fn, void* user, int world) {
  if (fn && (double)world * c->cfg.batch * c->cfg.height * c->cfg.width > 16777216.0) {
    dg_set_error("set_allreduce: world x batch x H x W = %.0f exceeds 2^24: the float32 voxel counts of the global batch "
                 "(M3 / M4, GT:581-589) would round", (double)world * c->cfg.batch * c->cfg.height * c->cfg.width);
    return DG_ERR_ARG;
  }
  if (fn && c->rccl_comm) depgan_rccl_shutdown(c);
  if (fn && world >= 1) {   // world == 1 keeps the hook (a one-rank group: rehearses the collective path exactly)
    c->ar_fn = fn;
    c->ar_user = user;
    c->world = world;
  } else {
    c->ar_fn = nullptr;
    c->ar_user = nullptr;
    c->world = 1;
  }
  return DG_OK;
}

int depgan_param_count(depgan_ctx* c, int net) {
  Net* n = pick_net(c, net);
  return n ? (int)n->params.size() : -1;
}

int depgan_param_info(depgan_ctx* c, int net, int index, char* name, int name_cap, int shape[4], int* ndim,
                      long* offset, int* trainable) {
  Net* n = pick_net(c, net);
  if (!n || index < 0 || index >= (int)n->params.size()) { dg_set_error("param index out of range"); return DG_ERR_ARG; }
  const PInfo& pi = n->params[index];
  if (name && name_cap > 0) {
    strncpy(name, pi.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  for (int i = 0; i < 4; ++i) shape[i] = pi.shape[i];
  *ndim = pi.ndim;
  *offset = (long)pi.off;
  *trainable = pi.trainable ? 1 : 0;
  return DG_OK;
}

long depgan_arena_floats(depgan_ctx* c, int net, int arena) {
  Net* n = pick_net(c, net);
  if (!n) return -1;
  return arena == DEPGAN_ARENA_NONTRAINABLE ? (long)n->nNon : (long)n->nTrain;
}

float* depgan_arena_ptr(depgan_ctx* c, int net, int arena) {
  Net* n = pick_net(c, net);
  if (!n) return nullptr;
  switch (arena) {
    case DEPGAN_ARENA_PARAMS: return n->P;
    case DEPGAN_ARENA_NONTRAINABLE: return n->NT;
    case DEPGAN_ARENA_GRADS: return n->G;
    case DEPGAN_ARENA_ADAM_M: return n->M;
    case DEPGAN_ARENA_ADAM_V: return n->V;
  }
  return nullptr;
}

int depgan_weights_changed(depgan_ctx* c, int net) {
  if (net == DEPGAN_NET_G) return refresh_generator(c);
  if (net == DEPGAN_NET_D_Y2) return refresh_critic(c, c->d[0]);
  if (net == DEPGAN_NET_D_DEM) return refresh_critic(c, c->d[1]);
  dg_set_error("unknown net id %d", net);
  return DG_ERR_ARG;
}

int depgan_g_forward(depgan_ctx* c, const float* x, const float* z, float* out, int n) {
  if (n < 1 || n > c->cfg.batch) { dg_set_error("g_forward: n must be in [1, batch]"); return DG_ERR_ARG; }
  if (c->cfg.nc_out != 1) return uresnet_predict(c, x, z, out, n);
  DGCHECK(g_forward(c, x, z, n, false));
  HIPCHECK(hipMemcpyAsync(out, c->attr.p, (size_t)n * c->cfg.height * c->cfg.width * sizeof(float),
                          hipMemcpyDeviceToDevice, c->st));
  return DG_OK;
}

int depgan_d_forward(depgan_ctx* c, int net, const float* img, float* out, int n) {
  if (net != DEPGAN_NET_D_Y2 && net != DEPGAN_NET_D_DEM) { dg_set_error("d_forward: not a critic id"); return DG_ERR_ARG; }
  if (n < 1 || n > c->NB3) { dg_set_error("d_forward: n must be in [1, 3*batch]"); return DG_ERR_ARG; }
  DGCHECK(d_forward(c, c->d[net - 1], img, 0, n));
  HIPCHECK(hipMemcpyAsync(out, c->d_out, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, c->st));
  return DG_OK;
}

int depgan_critic_grads(depgan_ctx* c, int net, const float* y2, const float* x, const float* z, const float* ep,
                        float out[2]) {
  return critic_impl(c, net, y2, x, z, ep, out, false);
}

int depgan_apply_adam(depgan_ctx* c, int net) {
  Net* n = pick_net(c, net);
  if (!n) return DG_ERR_ARG;
  DGCHECK(net_adam(c, *n));
  return depgan_weights_changed(c, net);
}

long depgan_get_adam_step(depgan_ctx* c, int net) {
  Net* n = pick_net(c, net);
  return n ? n->adam_t : -1;
}
int depgan_set_adam_step(depgan_ctx* c, int net, long t) {
  Net* n = pick_net(c, net);
  if (!n || t < 0) { dg_set_error("set_adam_step: bad argument"); return DG_ERR_ARG; }
  n->adam_t = t;
  return DG_OK;
}

int depgan_critic_step(depgan_ctx* c, int net, const float* y2, const float* x, const float* z, const float* ep,
                       float out[2]) {
  return critic_impl(c, net, y2, x, z, ep, out, true);
}

int depgan_g_eval(depgan_ctx* c, const float* x, const float* y2, const float* z, float out[6]) {
  return g_impl(c, x, y2, z, out, false, false);
}
// k evaluations into scal_multi (k x 8 pieces), reduced across ranks when a hook is set.  No host synchronisation.
static int g_eval_multi_enqueue(depgan_ctx* c, const float* x, const float* y2, const float* z_all, int k) {
  if (c->cfg.nc_out != 1) { dg_set_error("the WGAN-GP closures need nc_out == 1"); return DG_ERR_ARG; }
  if (k < 1 || k > DEPGAN_MAX_MULTI) { dg_set_error("best-of-k: k must be in [1, %d]", DEPGAN_MAX_MULTI); return DG_ERR_ARG; }
  const size_t zstride = (size_t)c->cfg.batch * 32;
  for (int i = 0; i < k; ++i) DGCHECK(g_eval_enqueue(c, x, y2, z_all + i * zstride, false, c->scal_multi + 8 * i));
  return dp_reduce(c, c->scal_multi, 8L * k);
}
int depgan_g_eval_multi(depgan_ctx* c, const float* x, const float* y2, const float* z_all, int k, float* out,
                        float* sums) {
  DGCHECK(g_eval_multi_enqueue(c, x, y2, z_all, k));
  DGCHECK(fetch_async(c, c->scal_multi, 8 * k, 0));
  HIPCHECK(hipStreamSynchronize(c->st));      // the only host synchronisation of the k evaluations
  for (int i = 0; i < k; ++i) {
    g_loss_from_sums(c->host_stats + 8 * i, out + 6 * i);
    if (sums) memcpy(sums + 8 * i, c->host_stats + 8 * i, 8 * sizeof(float));
  }
  return DG_OK;
}
int depgan_g_grads(depgan_ctx* c, const float* x, const float* y2, const float* z, float out[6]) {
  return g_impl(c, x, y2, z, out, true, false);
}
int depgan_g_step(depgan_ctx* c, const float* x, const float* y2, const float* z, float out[6]) {
  return g_impl(c, x, y2, z, out, true, true);
}

int depgan_gen_iteration(depgan_ctx* c, const float* x_y2, const float* y2_y2, const float* z_y2, const float* ep_y2,
                         int n_y2, const float* x_dem, const float* y2_dem, const float* z_dem, const float* ep_dem,
                         int n_dem, long batch_stride, const float* x_gen, const float* y2_gen, const float* z_gen, int k,
                         float* out_host, int* best_host) {
  if (c->cfg.nc_out != 1) { dg_set_error("the WGAN-GP closures need nc_out == 1"); return DG_ERR_ARG; }
  if (n_y2 < 0 || n_dem < 0 || n_y2 + n_dem > DEPGAN_MAX_CRITIC_STEPS || k < 1 || k > DEPGAN_MAX_MULTI || !out_host ||
      !best_host || !x_gen || !y2_gen || !z_gen || batch_stride < 0) {
    dg_set_error("gen_iteration: need 0 <= n_y2 + n_dem <= %d, 1 <= k <= %d and non-null generator inputs",
                 DEPGAN_MAX_CRITIC_STEPS, DEPGAN_MAX_MULTI);
    return DG_ERR_ARG;
  }
  if ((n_y2 > 0 && (!x_y2 || !y2_y2 || !z_y2 || !ep_y2)) || (n_dem > 0 && (!x_dem || !y2_dem || !z_dem || !ep_dem))) {
    dg_set_error("gen_iteration: a critic loop with n > 0 needs all four of its inputs");
    return DG_ERR_ARG;
  }
  const int B = c->cfg.batch;
  const long xs = batch_stride * c->cfg.height * c->cfg.width * c->cfg.nicg;
  const long ys = batch_stride * c->cfg.height * c->cfg.width;
  int slot = 0;
  for (int j = 0; j < n_y2; ++j, slot += 4) {                                           // GT:802-814
    DGCHECK(critic_enqueue(c, 0, y2_y2 + j * ys, x_y2 + j * xs, z_y2 + (size_t)j * B * 32, ep_y2 + (size_t)j * B));
    DGCHECK(finish_update(c, DEPGAN_NET_D_Y2, c->d[0].net, 4, slot, true));
  }
  for (int j = 0; j < n_dem; ++j, slot += 4) {                                          // GT:817-829
    DGCHECK(critic_enqueue(c, 1, y2_dem + j * ys, x_dem + j * xs, z_dem + (size_t)j * B * 32, ep_dem + (size_t)j * B));
    DGCHECK(finish_update(c, DEPGAN_NET_D_DEM, c->d[1].net, 4, slot, true));
  }
  DGCHECK(g_eval_multi_enqueue(c, x_gen, y2_gen, z_gen, k));                            // GT:868-874
  DGCHECK(fetch_async(c, c->scal_multi, 8 * k, slot));
  const int slot_k = slot;
  slot += 8 * k;
  {
    ProfScope ps(c, 2, 0.0, "best noise");
    DGCHECK(dg_best_noise(c->scal_multi, k, z_gen, (long)B * 32, c->best_dev, c->z_best, c->st));   // GT:875-876
  }
  HIPCHECK(hipMemcpyAsync(c->best_host, c->best_dev, sizeof(int), hipMemcpyDeviceToHost, c->st));
  DGCHECK(g_eval_enqueue(c, x_gen, y2_gen, c->z_best, true, c->g.G + c->g.nTrain));     // GT:878
  DGCHECK(finish_update(c, DEPGAN_NET_G, c->g, 8, slot, true));
  HIPCHECK(hipStreamSynchronize(c->st));       // the one host synchronisation of the generator iteration
  float* o = out_host;
  for (int j = 0; j < n_y2 + n_dem; ++j, o += 2) critic_from_sums(c->host_stats + 4 * j, o);
  for (int i = 0; i < k; ++i, o += 6) g_loss_from_sums(c->host_stats + slot_k + 8 * i, o);
  g_loss_from_sums(c->host_stats + slot, o);
  memcpy(c->last_sums, c->host_stats + slot, 8 * sizeof(float));
  *best_host = *c->best_host;
  return DG_OK;
}

int depgan_last_sums(depgan_ctx* c, float out[8]) {
  memcpy(out, c->last_sums, sizeof(c->last_sums));
  return DG_OK;
}

int depgan_profile_enable(depgan_ctx* c, int on) {
  c->prof_on = on != 0;
  return DG_OK;
}
int depgan_profile_reset(depgan_ctx* c) {
  hipStreamSynchronize(c->st);
  for (ProfRec& r : c->recs) {
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  c->recs.clear();
  return DG_OK;
}
int depgan_profile_read(depgan_ctx* c, int klass, double* total_ms, long* launches, double* flops) {
  HIPCHECK(hipStreamSynchronize(c->st));
  double ms = 0, fl = 0;
  long n = 0;
  for (ProfRec& r : c->recs)
    if (r.klass == klass) {
      float t = 0;
      HIPCHECK(hipEventElapsedTime(&t, r.a, r.b));
      ms += t;
      fl += r.flops;
      ++n;
    }
  *total_ms = ms;
  *launches = n;
  *flops = fl;
  return DG_OK;
}

int depgan_profile_read_bytes(depgan_ctx* c, int klass, double* bytes) {
  double by = 0;
  for (ProfRec& r : c->recs)
    if (r.klass == klass) by += r.bytes;
  *bytes = by;
  return DG_OK;
}

int depgan_profile_dump(depgan_ctx* c, const char* path) {
  HIPCHECK(hipStreamSynchronize(c->st));
  FILE* f = fopen(path, "w");
  if (!f) { dg_set_error("cannot open %s", path); return DG_ERR_ARG; }
  fprintf(f, "class,label,ms,gflop,mbytes,kernel\n");
  for (ProfRec& r : c->recs) {
    float t = 0;
    hipEventElapsedTime(&t, r.a, r.b);
    // the kernel name holds commas (template arguments): quoted
    fprintf(f, "%d,%s,%.4f,%.3f,%.3f,\"%s\"\n", r.klass, r.label, t, r.flops * 1e-9, r.bytes * 1e-6, r.kernel);
  }
  fclose(f);
  return DG_OK;
}

// ---- parity-test surface ----
int depgan_debug_capture(depgan_ctx* c, int on) {
  c->dbg_capture = on != 0;
  if (!on) c->dbg_mixed_valid = false;
  return DG_OK;
}

int depgan_debug_tensor(depgan_ctx* c, const char* name, float* host, long cap, int shape[4]) {
  if (!c || !name || !shape) { dg_set_error("debug_tensor: null argument"); return DG_ERR_ARG; }
  const std::string nm(name);
  const int B = c->cfg.batch;
  TView v = null_view();
  int N = 0, H = 0, W = 0, C = 0;
  auto rest = [&](const char* pre) { return nm.substr(strlen(pre)); };
  auto starts = [&](const char* pre) { return nm.compare(0, strlen(pre), pre) == 0; };
  if (nm == "g/heads" || nm == "g/noise_a0" || nm == "g/noise_a1") {
    float* p = nm == "g/heads" ? c->na.heads : (nm == "g/noise_a0" ? c->na.a0 : c->na.a1);
    v = make_view(p, 1, 1, 1024);
    N = B; H = 1; W = 1; C = 1024;
  } else if (starts("g/out/") || starts("g/u/")) {
    const bool want_u = starts("g/u/");
    const std::string ln = want_u ? rest("g/u/") : rest("g/out/");
    for (const GLayer& L : c->gl) {
      if (L.name != ln) continue;
      if (want_u) {
        if (L.kind != G_FILM || !L.u.p) break;
        v = L.u.view();
        H = L.H; W = L.W;
      } else {
        v = L.out;
        H = L.kind == G_POOL ? L.H / 2 : (L.kind == G_DECONV ? 2 * L.H : L.H);
        W = L.kind == G_POOL ? L.W / 2 : (L.kind == G_DECONV ? 2 * L.W : L.W);
      }
      N = B; C = L.Cout;
      break;
    }
  } else if (starts("d/act/") || starts("d/mixed/")) {
    const bool mixed = starts("d/mixed/");
    const std::string ln = mixed ? rest("d/mixed/") : rest("d/act/");
    for (size_t l = 0; l < c->dl.size(); ++l) {
      if (c->dl[l].name != ln) continue;
      if (mixed) {
        if (!c->dbg_mixed_valid || !c->dbg_mixed[l]) { dg_set_error("debug_tensor: %s was not captured (depgan_debug_capture)", name); return DG_ERR_ARG; }
        v = make_view(c->dbg_mixed[l], c->dl[l].H, c->dl[l].W, c->dl[l].Cout);
        N = B;
      } else {
        v = c->d_act[l].view();
        N = c->NB3;
      }
      H = c->dl[l].H; W = c->dl[l].W; C = c->dl[l].Cout;
      break;
    }
  }
  if (!v.p || N == 0) { dg_set_error("debug_tensor: unknown tensor '%s'", name); return DG_ERR_ARG; }
  shape[0] = N; shape[1] = H; shape[2] = W; shape[3] = C;
  if (!host) return DG_OK;
  const long need = (long)N * H * W * C;
  if (cap < need) { dg_set_error("debug_tensor: %s needs %ld floats, the buffer holds %ld", name, need, cap); return DG_ERR_ARG; }
  if (v.sY != (long)W * v.sX || v.sB != (long)H * v.sY) { dg_set_error("debug_tensor: %s is not a channel slice of a dense tensor", name); return DG_ERR_UNSUPPORTED; }
  HIPCHECK(hipStreamSynchronize(c->st));
  if (v.sX == C) {
    HIPCHECK(hipMemcpy(host, v.p, (size_t)need * sizeof(float), hipMemcpyDeviceToHost));
    return DG_OK;
  }
  // rows of C floats at a pitch of sX floats (a channel slice of a concat buffer) -> dense on the device, then down
  float* tmp = nullptr;
  HIPCHECK(hipMalloc((void**)&tmp, (size_t)need * sizeof(float)));
  hipError_t e = hipMemcpy2D(tmp, (size_t)C * sizeof(float), v.p, (size_t)v.sX * sizeof(float), (size_t)C * sizeof(float),
                             (size_t)N * H * W, hipMemcpyDeviceToDevice);
  if (e == hipSuccess) e = hipMemcpy(host, tmp, (size_t)need * sizeof(float), hipMemcpyDeviceToHost);
  hipFree(tmp);
  if (e != hipSuccess) { dg_set_error("debug_tensor: copy of %s failed: %s", name, hipGetErrorString(e)); return DG_ERR_HIP; }
  return DG_OK;
}

// ---- single operators (unit tests) ----
static int op_conv(const float* in, const float* w_hwio, const float* bias, float* out, int B, int H, int W, int Cin,
                   int Cout, int KS, int relu, int path, int bwd, hipStream_t st) {
  // bwd: compute dx = conv_bwd_data(dy=in (Cout ch), W) -> out (Cin ch)
  const int ci = bwd ? Cout : Cin, co = bwd ? Cin : Cout;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  zero_ep(&a.ep);
  a.in = make_view(const_cast<float*>(in), H, W, ci);
  a.out = make_view(out, H, W, co);
  a.B = B; a.H = H; a.W = W; a.Cin = ci; a.Cout = co;
  a.ep.bias = bias;
  a.ep.relu = relu;
  ConvPlan pl = (path == 3) ? dg_plan_conv_bf16(KS, ci, co)
                : (path == 4 || path == 5) ? dg_plan_conv_split(KS, ci, co, path == 5 ? 3 : 2)
                : (path == 6) ? dg_plan_conv_items(KS, ci, co, 1L << 30) : dg_plan_conv(KS, ci, co);
  if (path == 7) pl = dg_plan_conv_items(KS, ci, co, 1L << 30);
  if (path == 8) {
    pl = (KS == 3) ? dg_plan_conv_wino(ci, co) : pl;
    if (pl.variant != 9 || !dg_conv_wino_supported(pl, a)) { dg_set_error("op_conv: the Winograd kernel does not cover this shape"); return DG_ERR_UNSUPPORTED; }
  }
  if ((path == 6 || path == 7) && pl.CK != 8) { dg_set_error("op_conv: the 8-channel-chunk variant does not cover this shape"); return DG_ERR_UNSUPPORTED; }
  if (path >= 3 && path <= 5 && !pl.bf16) { dg_set_error("op_conv: the bf16 MFMA kernel does not cover this shape"); return DG_ERR_UNSUPPORTED; }
  if (path == 1 && pl.variant < 0) { dg_set_error("op_conv: MFMA path not available for this shape"); return DG_ERR_UNSUPPORTED; }
  if (path != 2 && pl.variant >= 0) {
    float* wp = nullptr;
    HIPCHECK(hipMalloc((void**)&wp, pl.packedFloats * sizeof(float)));
    int rc = dg_pack_weights(pl, w_hwio, Cin, Cout, 0, bwd, bwd, nullptr, wp, st);
    if (rc == DG_OK) {
      a.w = wp;
      if (path == 7) {
        if (dg_conv_igemm_wp_supported(pl, a, true)) rc = dg_conv_igemm_wp(pl, a, st);
        else { dg_set_error("op_conv: the wave-private kernel does not cover this shape"); rc = DG_ERR_UNSUPPORTED; }
      } else if (path == 6) {
        // the workgroup-tile kernel itself (the reference the wave-private kernel must match bit for bit)
        rc = dg_conv_igemm_tile(pl, a, st);
      } else {
        rc = dg_conv_igemm(pl, a, st);
      }
    }
    hipStreamSynchronize(st);
    hipFree(wp);
    return rc;
  }
  a.w = w_hwio;
  if (!bwd) {
    a.wsT = (long)Cin * Cout; a.wsI = Cout; a.wsO = 1; a.flip = 0;
  } else {
    a.wsT = (long)Cin * Cout; a.wsI = 1; a.wsO = Cout; a.flip = 1;
  }
  return dg_conv_direct(KS, a, st);
}

// diagnostics: run the MFMA conv with per-workgroup phase stamps (16 x u64 per workgroup) into `stamps`
int depgan_op_conv2d_stamps(const float* in, const float* w_hwio, float* out, int B, int H, int W, int Cin, int Cout,
                            int KS, unsigned long long* stamps, int reps, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  zero_ep(&a.ep);
  a.in = make_view(const_cast<float*>(in), H, W, Cin);
  a.out = make_view(out, H, W, Cout);
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.ep.relu = 1;
  ConvPlan pl = dg_plan_conv(KS, Cin, Cout);
  if (pl.variant < 0) { dg_set_error("no MFMA variant"); return DG_ERR_UNSUPPORTED; }
  float* wp = nullptr;
  HIPCHECK(hipMalloc((void**)&wp, pl.packedFloats * sizeof(float)));
  int rc = dg_pack_weights(pl, w_hwio, Cin, Cout, 0, 0, 0, nullptr, wp, st);
  a.w = wp;
  for (int i = 0; i < reps && rc == DG_OK; ++i) {
    a.dbg = (i == reps - 1) ? stamps : nullptr;
    rc = dg_conv_igemm(pl, a, st);
  }
  hipStreamSynchronize(st);
  hipFree(wp);
  return rc;
}

int depgan_op_conv2d(const float* in, const float* w_hwio, const float* bias, float* out, int B, int H, int W,
                     int Cin, int Cout, int KS, int relu, int path, void* stream) {
  return op_conv(in, w_hwio, bias, out, B, H, W, Cin, Cout, KS, relu, path, 0, (hipStream_t)stream);
}
int depgan_op_conv2d_bwd_data(const float* dy, const float* w_hwio, float* dx, int B, int H, int W, int Cin,
                              int Cout, int KS, int path, void* stream) {
  return op_conv(dy, w_hwio, nullptr, dx, B, H, W, Cin, Cout, KS, 0, path, 1, (hipStream_t)stream);
}
int depgan_op_conv2d_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int KS,
                           void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const bool big = (Cin % 4 == 0 && Cout % 4 == 0 && Cin >= 8);
  const size_t pf = big ? dg_wgrad_part_floats(KS, B, H, W, Cin, Cout) : dg_wgrad_small_part_floats(KS, B, H, W, Cin, Cout);
  float* part = nullptr;
  HIPCHECK(hipMalloc((void**)&part, pf * sizeof(float)));
  WgradArgs a;
  a.x = make_view(const_cast<float*>(x), H, W, Cin);
  a.dy = make_view(const_cast<float*>(dy), H, W, Cout);
  a.part = part;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.nTiles = a.tilesPerChunk = 0;
  a.colpart = nullptr;
  a.colB = 0;
  int nch = 0;
  int rc = big ? dg_wgrad(KS, a, &nch, st) : dg_wgrad_small(KS, a, &nch, st);
  if (rc == DG_OK) rc = dg_wgrad_reduce(part, nch, KS * KS, Cin, Cout, nullptr, dw, nullptr, 0, 0, st);
  hipStreamSynchronize(st);
  hipFree(part);
  return rc;
}
int depgan_op_conv2d_wgrad_bf16(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                                int KS, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!dg_wgrad_bf16_supported(KS, Cin, Cout)) { dg_set_error("op_wgrad_bf16: shape not covered"); return DG_ERR_UNSUPPORTED; }
  const size_t pf = dg_wgrad_bf16_part_floats(KS, B, H, W, Cin, Cout);
  float* part = nullptr;
  HIPCHECK(hipMalloc((void**)&part, pf * sizeof(float)));
  WgradArgs a;
  a.x = make_view(const_cast<float*>(x), H, W, Cin);
  a.dy = make_view(const_cast<float*>(dy), H, W, Cout);
  a.part = part;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.nTiles = a.tilesPerChunk = 0;
  a.colpart = nullptr;
  a.colB = 0;
  int nch = 0;
  int rc = dg_wgrad_bf16(KS, a, &nch, st);
  if (rc == DG_OK) rc = dg_wgrad_reduce(part, nch, KS * KS, Cin, Cout, nullptr, dw, nullptr, 0, 0, st);
  hipStreamSynchronize(st);
  hipFree(part);
  return rc;
}
int depgan_op_deconv2x2(const float* in, const float* w_hwoi, const float* bias, const float* scale,
                        const float* shift, float* out, int B, int H, int W, int Cin, int Cout, int relu,
                        void* stream) {
  if (!in || !w_hwoi || !out || B < 1 || H < 1 || W < 1) { dg_set_error("op_deconv2x2: bad argument"); return DG_ERR_ARG; }
  DeconvArgs d;
  memset(&d, 0, sizeof(d));
  d.in = in;
  d.w = w_hwoi;
  d.out = make_view(out, 2 * H, 2 * W, Cout);
  d.bias = bias; d.scale = scale; d.shift = shift; d.relu = relu;
  d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout;
  return dg_deconv_fwd(d, B, (hipStream_t)stream);
}
int depgan_op_deconv2x2_wgrad(const float* in, const float* dout, float* dw_hwoi, float* colsum, int B, int H, int W,
                              int Cin, int Cout, void* stream) {
  if (!in || !dout || !dw_hwoi || B < 1 || H < 1 || W < 1) { dg_set_error("op_deconv2x2_wgrad: bad argument"); return DG_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  DeconvWgradArgs d;
  memset(&d, 0, sizeof(d));
  d.in = in;
  d.dout = make_view(const_cast<float*>(dout), 2 * H, 2 * W, Cout);
  d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout;
  if (!dg_deconv_wgrad_supported(B, H, W, Cin, Cout, make_view(const_cast<float*>(in), H, W, Cin), d.dout)) {
    dg_set_error("op_deconv2x2_wgrad: shape %dx%dx%d %d->%d not covered by the fused kernel", B, H, W, Cin, Cout);
    return DG_ERR_UNSUPPORTED;
  }
  const size_t pf = dg_deconv_wgrad_part_floats(B, H, W, Cin, Cout);
  float *part = nullptr, *col = nullptr;
  HIPCHECK(hipMalloc((void**)&part, pf * sizeof(float)));
  if (hipMalloc((void**)&col, (pf / ((size_t)Cin * Cout)) * Cout * sizeof(float)) != hipSuccess) {
    hipFree(part);
    dg_set_error("op_deconv2x2_wgrad: out of memory");
    return DG_ERR_HIP;
  }
  d.part = part;
  d.colpart = colsum ? col : nullptr;
  int nch = 0;
  int rc = dg_deconv_wgrad(d, B, &nch, st);
  if (rc == DG_OK)
    rc = dg_wgrad_finish_rows(part, nch, 4, Cin, Cout, nullptr, dw_hwoi, nullptr, 0, 1, colsum ? col : nullptr, 4 * nch,
                              Cout, nullptr, colsum, nullptr, st);
  hipStreamSynchronize(st);
  hipFree(part);
  hipFree(col);
  return rc;
}
int depgan_op_maxpool(const float* in, float* out, int B, int Ho, int Wo, int C, void* stream) {
  return dg_maxpool(make_view(const_cast<float*>(in), 2 * Ho, 2 * Wo, C), make_view(out, Ho, Wo, C), B, Ho, Wo, C,
                    (hipStream_t)stream);
}

// ---- evaluation step after the path (GE:616-807): stateless, caller's stream ----
int depgan_eval_accumulate(const float* pred, const float* mask, double* acc, long n, void* stream) {
  if (!pred || !acc || n < 0) { dg_set_error("eval_accumulate: null argument"); return DG_ERR_ARG; }
  return dg_eval_accumulate(pred, mask, acc, (size_t)n, (hipStream_t)stream);
}
int depgan_eval_divide(double* acc, long n, double divisor, void* stream) {
  if (!acc || n < 0) { dg_set_error("eval_divide: null argument"); return DG_ERR_ARG; }
  return dg_eval_divide(acc, (size_t)n, divisor, (hipStream_t)stream);
}
int depgan_eval_counts(const float* x, int nicg, const double* pred, const float* code_real, const float* mask1,
                       const float* wmh1, const float* mask2, const float* wmh2, const float* prob2, long npix,
                       double thr, long long out_host[DEPGAN_EVAL_NCOUNT], void* stream) {
  if (!x || !pred || !out_host || nicg < 1 || npix < 0) { dg_set_error("eval_counts: bad argument"); return DG_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* dev = nullptr;
  HIPCHECK(hipMalloc((void**)&dev, DEPGAN_EVAL_NCOUNT * sizeof(unsigned long long)));
  int rc = dg_eval_counts(x, nicg, pred, code_real, mask1, wmh1, mask2, wmh2, prob2, (size_t)npix, thr, dev, st);
  if (rc == DG_OK) {
    unsigned long long h[DEPGAN_EVAL_NCOUNT];
    if (hipMemcpyAsync(h, dev, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
      dg_set_error("eval_counts: copy back failed");
      rc = DG_ERR_HIP;
    } else {
      for (int k = 0; k < DEPGAN_EVAL_NCOUNT; ++k) out_host[k] = (long long)h[k];
    }
  }
  hipFree(dev);
  return rc;
}

}  // extern "C"
