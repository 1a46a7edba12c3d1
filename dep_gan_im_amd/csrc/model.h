// Host-side model state for the DEP-GAN hot path: parameter arenas, layer
// tables (GT:316-345, GT:349-498), activation storage and the step drivers.
#pragma once
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/depgan.h"
#include "common.h"
#include "deconv_fwd.h"
#include "noise.h"
#include "ops.h"

struct PInfo {
  std::string name;
  int shape[4];
  int ndim;
  size_t off, size;
  bool trainable;
};

struct Net {
  std::vector<PInfo> params;
  std::map<std::string, int> index;
  size_t nTrain = 0, nNon = 0;
  float *P = nullptr, *NT = nullptr, *G = nullptr, *M = nullptr, *V = nullptr;
  // bf16-weights mode (BASELINE config 4): P stays the fp32 master that Adam updates; every kernel reads the
  // compute copy Pq, in which the "/kernel" tensors are rounded to bf16 (RNE) and everything else is P verbatim
  float* Pq = nullptr;
  unsigned char* qmask = nullptr;   // 1 where Pq is rounded
  long adam_t = 0;
  float lr = 1e-4f;
  void add(const std::string& name, std::vector<int> shape, bool trainable);
  float* p(const std::string& name) const;   // parameter pointer the kernels read (either arena; Pq when quantised)
  float* g(const std::string& name) const;   // gradient pointer (trainable only)
};

struct Tn {  // dense NHWC tensor
  float* p = nullptr;
  int H = 0, W = 0, C = 0;
  TView view() const { return make_view(p, H, W, C); }
  TView slice(int c0) const { return make_view_slice(p, H, W, C, c0); }
  size_t per_sample() const { return (size_t)H * W * C; }
};

enum GKind { G_CONV, G_FILM, G_POOL, G_DECONV, G_HEAD };

struct GLayer {
  GKind kind;
  std::string name;
  int Cin = 0, Cout = 0;
  int H = 0, W = 0;      // spatial size of the layer INPUT
  // parameters
  float *Wt = nullptr, *b = nullptr, *gamma = nullptr, *beta = nullptr, *mean = nullptr, *var = nullptr;
  float *dW = nullptr, *db = nullptr, *dgamma = nullptr, *dbeta = nullptr;
  float *s = nullptr, *t = nullptr, *rstd = nullptr;
  // packed weights (deconv: one per tap)
  ConvPlan pf, pb;
  float* wpf[4] = {nullptr, nullptr, nullptr, nullptr};
  float* wpb[4] = {nullptr, nullptr, nullptr, nullptr};
  // deconv backward-data as ONE 1x1 convolution over the four strided grids of the upstream gradient (K = 4 Cout):
  // the four per-tap panels interleaved per channel tile; null when the channel counts do not allow it
  ConvPlan pbf;
  float* wpb_all = nullptr;
  // FiLM
  int col_mul = -1, col_add = -1;
  // tensors
  TView in, out;           // forward views (in has Cin channels, out has Cout)
  TView din, dout;         // gradient views (dout = grad wrt out, after the producer's mask)
  TView in_mask;           // mask applied when writing din (null: none)
  Tn u;                    // FiLM pre-activation (kept when training G)
  // learning-phase-1 path (DEP-UResNet): raw conv output and batch-statistics BN state
  Tn raw;
  float *bmean = nullptr, *bvar = nullptr, *bs = nullptr, *bt = nullptr, *brstd = nullptr;
  float *cA = nullptr, *cB = nullptr, *cC = nullptr, *sums = nullptr;
  int skip_of = -1;        // pool: index of the conv layer whose output is pooled
  TView pool_dsrc;         // pool: gradient wrt the pooled tensor (raw)
  TView pool_skipgrad;     // pool: gradient arriving through the concat
  TView pool_dst;          // pool: masked gradient of the pooled conv's output
};

struct DLayer {
  std::string name;
  int KS, Cin, Cout, H, W;  // H, W: spatial size at this layer
  bool pool;
  ConvPlan pf, pb;
};

struct DNet {
  Net net;
  PackJob* pack_jobs = nullptr;   // device table of this critic's packing jobs (built at the first refresh)
  int n_pack = 0;
  unsigned pack_blocks = 0;
  float* wpf[11];
  float* wpb[11];
  float *W[11], *b[11], *dW[11], *db[11];
  float *w9, *b9, *wd, *bd, *dw9, *db9, *dwd, *dbd;
};

struct ProfRec {
  hipEvent_t a, b;
  int klass;
  double flops;
  double bytes;   // algorithmic HBM bytes of the launch (operands read once, results written once)
  char label[56];
  char kernel[48];   // kernel instantiation (class 0 / 1) as rocprofv3 names it; empty for the HBM-bound helpers
};

struct depgan_ctx {
  depgan_config cfg;
  hipStream_t st = nullptr;
  std::vector<void*> allocs;

  // ---- generator ----
  Net g;
  std::vector<GLayer> gl;
  NoiseParams np;
  NoiseGrads ng;
  NoiseActs na;
  float* derived = nullptr;       // BN affines
  // device tables for the batched refresh launches (built at the first refresh; every pointer in them is stable)
  PackJob* g_pack_jobs = nullptr;
  int g_n_pack = 0;
  unsigned g_pack_blocks = 0;
  BnJob* g_bn_jobs = nullptr;
  int g_n_bn = 0;
  // BN-gamma gradients of the generator: the un-scaled weight gradients of a backward pass are kept per layer
  // (raw_all mirrors the gradient arena) and all gammas are formed in ONE launch at the end of the pass
  float* raw_all = nullptr;
  GammaJob* g_gamma_jobs = nullptr;
  int g_n_gamma = 0, g_gamma_blocks = 0;
  float* heads_mean = nullptr;    // concatenated moving means of the head BNs
  float* dheads = nullptr;        // [B][1024]
  Tn attr;                        // generator output (B,H,W,1)
  Tn du_tmp;                      // FiLM dU scratch (largest FiLM tensor)
  float* dpre = nullptr;          // [B*H*W]
  float* zbuf = nullptr;          // [B][32]

  // ---- critics ----
  DNet d[2];
  std::vector<DLayer> dl;
  int NB3 = 0;                     // 3*B
  float* d_in = nullptr;           // [3B][H][W][1]
  Tn d_act[11];                    // post-ReLU activations [3B]
  Tn d_pool[11];                   // pooled outputs (pool layers only)
  Tn d_dz[11];                     // gradient at the conv output (after ReLU mask)
  Tn d_dpool[11];                  // gradient wrt pooled tensor (raw)
  Tn d_ufull;                      // u-forward scratch for pooled layers [B]
  float *d_t9 = nullptr, *d_out = nullptr;  // [3B][hw], [3B]
  float* g0 = nullptr;             // [2B][H][W][1] image gradients
  float* coefs = nullptr;          // [4] per-group upstream coefficients
  float *norms = nullptr, *gp = nullptr;

  // ---- data parallelism + deferred scalars ----
  depgan_allreduce_fn ar_fn = nullptr;   // all-reduce (sum) hook, enqueued on the stream (include/depgan.h)
  void* ar_user = nullptr;
  int world = 1;
  void* rccl_comm = nullptr;       // ncclComm_t of the direct binding (depgan_rccl_init); takes precedence over ar_fn
  long rccl_issued = 0;
  int device = 0;                  // HIP device the context was created on
  bool winograd = true;           // 3x3 convolutions on the Winograd F(2x2,3x3) kernel where it covers them (DEPGAN_WINOGRAD=0: direct)
  bool head_fused = true;         // gen_segmentation in gen_17's epilogue where the kernel allows (DEPGAN_HEAD_FUSED=0: own launch)
  bool wgrad_bf16 = true;          // bf16_mfma contexts: weight gradients on the bf16 pipe (DEPGAN_WGRAD_BF16=0: fp32)
  float* host_stats = nullptr;     // pinned: un-normalised loss pieces of the updates of one call, fetched asynchronously
  int* best_dev = nullptr;         // arg-min of the best-of-k search (device) and its pinned host copy
  int* best_host = nullptr;
  float* z_best = nullptr;         // [B][32] the chosen noise, gathered on the device

  // ---- shared scratch ----
  float* part = nullptr;           // wgrad slabs
  size_t partFloats = 0;
  float* raw = nullptr;            // dWraw scratch (largest kernel)
  float* Sraw = nullptr;           // [256] raw column sums
  float* scratch = nullptr;        // reductions
  float* scal = nullptr;           // device scalars
  float* scal_multi = nullptr;     // 8 floats per evaluation of depgan_g_eval_multi
  float* fake_y2 = nullptr;        // [B*H*W]
  float last_sums[8];

  // ---- learning-phase-1 path (nc_out == 4) ----
  bool train_bn = false;
  unsigned last_drop_seed = 0;
  Tn draw_tmp;                     // gradient at the raw conv output (largest layer)
  float *logits = nullptr, *dz = nullptr, *loss_dev = nullptr;
  float *ones1k = nullptr, *zeros1k = nullptr;
  float *n_mean0 = nullptr, *n_rstd0 = nullptr, *n_mean1 = nullptr, *n_rstd1 = nullptr, *n_meanh = nullptr,
        *n_rstdh = nullptr;
  float *n_dl = nullptr, *n_dflat = nullptr, *n_dl1 = nullptr, *n_da0 = nullptr, *n_dl0 = nullptr;

  // ---- parity-test surface (depgan_debug_capture / depgan_debug_tensor) ----
  bool dbg_capture = false;
  float* dbg_mixed[11] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool dbg_mixed_valid = false;

  // ---- profiling ----
  bool prof_on = false;
  std::vector<ProfRec> recs;
};

struct ProfScope {
  depgan_ctx* c;
  bool live;
  ProfScope(depgan_ctx* c_, int klass, double flops, const char* label = "", double bytes = 0.0,
            const char* kernel = "")
      : c(c_), live(c_->prof_on) {
    if (!live) return;
    ProfRec r;
    r.klass = klass;
    r.flops = flops;
    r.bytes = bytes;
    strncpy(r.label, label, sizeof(r.label) - 1);
    r.label[sizeof(r.label) - 1] = 0;
    strncpy(r.kernel, kernel, sizeof(r.kernel) - 1);
    r.kernel[sizeof(r.kernel) - 1] = 0;
    hipEventCreate(&r.a);
    hipEventCreate(&r.b);
    hipEventRecord(r.a, c->st);
    c->recs.push_back(r);
  }
  ~ProfScope() {
    if (live) hipEventRecord(c->recs.back().b, c->st);
  }
};

// helpers shared between model.hip and uresnet.hip
int dmalloc(depgan_ctx* c, float** p, size_t floats);
int talloc(depgan_ctx* c, Tn* t, int N, int H, int W, int C);
int conv_launch(depgan_ctx* c, const ConvPlan& pl, const ConvArgs& a, int KS);
void zero_ep(Epilogue* e);
TView view_offset(TView v, long samples);
TView strided2(TView v, int di, int dj);
int deconv_bwd_data(depgan_ctx* c, GLayer& L, TView dsrc, int n);
// weight gradient (four taps) + column sums of the upstream gradient of a transposed convolution
int deconv_wgrad_all(depgan_ctx* c, const GLayer& L, TView dsrc, int n, const float* scale, float* raw,
                     const float* colscale, float* colout, float* colraw);
// forward of a transposed convolution on the fused four-tap kernel (deconv_fwd.hip) where it covers the layer
bool deconv_fused(const depgan_ctx* c, const GLayer& L, int n);
int deconv_fwd_launch(depgan_ctx* c, const GLayer& L, TView out, const float* bias, const float* scale,
                      const float* shift, int relu, int n);
// column sums of dy over its first B samples, delivered with the weight gradient: out = scale * sum, raw = sum
struct ColSum {
  int B;
  const float* scale;
  float *out, *raw;
};
int wgrad_full(depgan_ctx* c, int KS, TView x, TView dy, int N, int H, int W, int Cin, int Cout, const float* scale,
               float* out, float* raw, int accumulate, int oi, const ColSum* cs = nullptr);
int net_adam(depgan_ctx* c, Net& n, float gscale = 1.0f);
int g_forward(depgan_ctx* c, const float* x, const float* z, int n, bool store_u);
int refresh_generator(depgan_ctx* c);
int refresh_generator_bn(depgan_ctx* c);  // phase-0 BN affines only (after the moving statistics moved)
int uresnet_build(depgan_ctx* c);
int uresnet_predict(depgan_ctx* c, const float* x, const float* z, float* out, int n);
