// Shared declarations for the DEP-GAN hot-path HIP library (gfx950 / MI355X).
//
// Data layout: every activation is NHWC fp32 with the channel dimension
// contiguous; a TView carries (batch, row, pixel) strides in floats so one
// kernel serves plain tensors, channel slices of a concat buffer (reference
// concatenate() at GT:450/465/479 is never materialised twice) and the 2x
// strided pixel grids of the 2x2/stride-2 transposed convolution (GT:308).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TView {
  float* p;
  long sB, sY, sX;  // strides in floats; channel stride is 1
};

static inline TView make_view(float* p, int H, int W, int C) {
  TView v;
  v.p = p;
  v.sX = C;
  v.sY = (long)W * C;
  v.sB = (long)H * W * C;
  return v;
}
// channel slice [c0, ...) of a tensor with Ctot channels
static inline TView make_view_slice(float* p, int H, int W, int Ctot, int c0) {
  TView v = make_view(p, H, W, Ctot);
  v.p = p + c0;
  return v;
}
static inline TView null_view() {
  TView v;
  v.p = nullptr;
  v.sB = v.sY = v.sX = 0;
  return v;
}

// Fused epilogue description, applied in this order to acc (the raw contraction):
//   v = (acc + bias[co]) * scale[co] + shift[co]   (phase-0 BatchNorm affine; the MFMA kernels
//                                          evaluate it as fma(acc, scale, bias * scale + shift): one rounding)
//   out_pre = v                            (pre-FiLM tensor kept for the G backward)
//   v = v * film_mul[b,co] + film_add[b,co]  (GT:403-404)
//   v = max(v, 0)                          (relu)
//   v += res                               (residual add GT:407, or a gradient join)
//   v *= (mask > 0)                        (ReLU mask of the consumer side in backward passes)
//   out = accumulate ? out + v : v
struct Epilogue {
  const float* bias;
  const float* scale;
  const float* shift;
  const float* film_mul;
  const float* film_add;
  int film_ld;  // row stride (floats) of film_mul / film_add
  TView out_pre;
  TView res;
  TView mask;
  int relu;
  int accumulate;
  // igemm only: when pool.p is non-null the 2x2 / stride-2 max-pool of the final output is written here as well
  // (view of (H/2, W/2) pixels; H and W must be even).  Saves the pooling pass its read of the whole output.
  TView pool;
  // igemm, 32-channel layers only (the 8-channel-chunk 3x3 kernel): when head_out is non-null the 1x1 convolution to ONE
  // channel that follows (gen_segmentation after gen_17, GT:494-495) rides in this launch: head_out[pixel] =
  // act(sum_c out[pixel][c] head_w[c] + head_b[0]), act = tanh when head_tanh.  head_skip_out: the 32-channel output
  // itself is not stored (forward-only passes, where the head is its only consumer).
  const float* head_w;
  const float* head_b;
  float* head_out;
  int head_tanh, head_skip_out;
};

struct ConvArgs {
  TView in;
  TView out;
  const float* w;  // packed (igemm) or strided (direct) weights
  int B, H, W, Cin, Cout;
  Epilogue ep;
  // direct kernel only: weight element (tap, ci, co) = w[tapidx*wsT + ci*wsI + co*wsO],
  // tapidx = flip ? ntaps-1-tap : tap
  long wsT, wsI, wsO;
  int flip;
  int vec4;  // direct kernel: all output-side views 16-byte aligned (set by the launcher)
  // diagnostics: when non-null, thread 0 of every workgroup writes 8 x u64 phase stamps here
  unsigned long long* dbg;
  // igemm only: grouped launch.  groups > 1 runs `groups` convolutions that share the input and the epilogue
  // constants in ONE launch (the 4 taps of a 2x2 / stride-2 transposed convolution): group g uses the packed weights
  // w_group[g] and writes to out.p + out_group_off[g]; `w` is ignored.  0 or 1: a single convolution.
  int groups;
  int lgx, lgy;   // igemm: logical grid (pixel tiles, channel tiles x groups), filled by the launcher
  const float* w_group[4];
  long out_group_off[4];
  // igemm only: gathered K.  cpt > 0 splits the Cin axis into consecutive runs of cpt channel chunks; run r reads its
  // channels at in.p + in_run_off[r] (floats) instead of contiguously after run r-1 -- the backward-data of the
  // 2x2 / stride-2 transposed convolution as ONE 1x1 convolution over the four strided pixel grids of its upstream
  // gradient (K = 4 Cout).  The run length cpt * CK must divide into whole chunks (no channel tail inside a run).
  int cpt;
  long in_run_off[4];
};

struct WgradArgs {
  TView x;      // operand shifted by the tap ('same' padding)
  TView dy;     // operand at the output pixel
  float* part;  // [nchunks][ntaps][Cin][Cout] partial sums
  int B, H, W, Cin, Cout;
  int nTiles, tilesPerChunk;
  // MFMA kernel only: when non-null, the column sums of dy over the samples b < colB (bias / BN-beta gradients) ride
  // along: the workgroups of input-channel tile 0 / tap group 0 write one partial row per chunk, [nchunks][Cout]
  float* colpart;
  int colB;
};

// One v_max_f32.  fmaxf() compiles to two: IEEE maxNum has to quiet a signalling NaN, so the compiler first
// canonicalises every operand it cannot prove canonical (v_max_f32 x, x, x).  On a SIMD every vector instruction costs
// its issue time next to the fp32 MFMAs: the 32 extra ones per ReLU'd 64x32 wave tile are not free.
// For non-NaN operands the result is bit-identical.
#ifdef __HIPCC__
static __device__ __forceinline__ float dg_vmax(float a, float b) {
  float o;
  asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
  return o;
}
#endif

#define DG_OK 0
#define DG_ERR_ARG 1
#define DG_ERR_HIP 2
#define DG_ERR_UNSUPPORTED 3

void dg_set_error(const char* fmt, ...);

#define HIPCHECK(expr)                                                              \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      dg_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return DG_ERR_HIP;                                                            \
    }                                                                               \
  } while (0)

#define DGCHECK(expr)          \
  do {                         \
    int _r = (expr);           \
    if (_r != DG_OK) return _r; \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Launcher-side state that belongs to a DEVICE, not to the process: a kernel's dynamic-LDS attribute, its occupancy, the
// CU count.  A process may hold contexts on several GPUs (Engine(device=...)): function-local statics would give the
// second device the first one's numbers and never set its attributes.
#define DG_MAX_DEVICES 64
static inline int dg_device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= DG_MAX_DEVICES) return -1;
  return d;
}
struct DgOncePerDevice {   // `if (once.need()) { set attribute ... }`; a device beyond the table is set every time
  bool done[DG_MAX_DEVICES] = {};
  bool need() {
    const int d = dg_device_slot();
    if (d < 0) return true;
    if (done[d]) return false;
    done[d] = true;
    return true;
  }
};
static inline int dg_cu_count() {   // CUs of the current device, asked once per device (the query is slower than a launch)
  static int cus[DG_MAX_DEVICES] = {};
  const int d = dg_device_slot();
  if (d >= 0 && cus[d]) return cus[d];
  int n = 256;
  hipDeviceProp_t p;
  if (d >= 0 && hipGetDeviceProperties(&p, d) == hipSuccess && p.multiProcessorCount > 0) n = p.multiProcessorCount;
  if (d >= 0) cus[d] = n;
  return n;
}

// ---- conv plans -----------------------------------------------------------
// How one convolution maps on the MFMA implicit-GEMM kernel.
struct ConvPlan {
  int KS, Cin, Cout;
  int MF;    // MFMA tile edge: 32 (v_mfma_f32_32x32x2_f32) or 16 (v_mfma_f32_16x16x4_f32)
  int NT;    // output channels per workgroup
  int CK;    // input channels staged per LDS chunk
  int nNT, nCC;
  size_t packedFloats;  // size of the packed panel in 4-byte units (bf16 plans: two elements per unit)
  int variant;  // index into the instantiation table, -1 = not MFMA-eligible
  int bf16;     // 1: bf16 matrix pipe (igemm_bf16.hip): panels are packed as bf16, the activation operand is rounded to
                // bf16 while it is staged, accumulation stays fp32
};
ConvPlan dg_plan_conv(int KS, int Cin, int Cout);
// ... knowing the number of work items of its launches (chunk size by launch size, see igemm_conv.hip)
ConvPlan dg_plan_conv_items(int KS, int Cin, int Cout, long items);
// the bf16 plan where the bf16 kernel covers the shape (Cout % 32 == 0, Cin >= 8), else the fp32 plan
ConvPlan dg_plan_conv_bf16(int KS, int Cin, int Cout);
// fp32 operands split into `planes` (2 or 3) bf16 terms each, 3 or 6 products on the bf16 pipe (igemm_split_kernel);
// the plan's bf16 field then holds the number of planes and its packed layout is
// [nt][cc][tap group][plane][tap][n][k = 16]
ConvPlan dg_plan_conv_split(int KS, int Cin, int Cout, int planes);
int dg_conv_igemm_bf16(const ConvPlan& pl, const ConvArgs& a, hipStream_t st);
// byte size of one packed element of a plan
// bytes of packed storage per element of the plain [nt][cc][tap][n][k] panel (split plans store `bf16` planes of it)
static inline size_t dg_plan_elem_bytes(const ConvPlan& pl) { return pl.variant >= 200 ? 2 * (size_t)pl.bf16 : (pl.bf16 ? 2 : 4); }

// One weight-packing job of a batched launch (dg_pack_weights_batch): what dg_pack_weights takes, plus an optional
// re-spacing of the channel-tile blocks in the destination (nt_stride elements between consecutive channel tiles;
// 0 = dense) so that several sources can be interleaved per channel tile.
struct PackJob {
  const float* src;
  float* dst;
  const float* kscale;
  int ntaps, srcI, srcO, io, transpose, flip;
  int NT, CK, nCC, Kdim, Ndim;
  int bf16;             // 1: dst is a bf16 panel (elements of 2 bytes, RNE from the fp32 source x kscale); 2 / 3: split
                        // panel of that many planes (plane p = bf16 of what planes < p left over), TAPG taps per group
  int tapg;             // split panels only: taps per staged group
  int wino;             // 1: Winograd panel -- ntaps = 16 "frequencies" f = 4a + b, element = (G g G^T)[a][b] of the 3x3 source
  unsigned total;       // packed elements of this job
  unsigned per_nt;      // packed floats per channel tile (nCC * ntaps * NT * CK)
  unsigned nt_stride;   // destination elements between channel tiles
  unsigned blk0, nblk;  // first block and number of blocks of this job inside the batched grid
};
int dg_pack_job(const ConvPlan& pl, const float* src, int srcI, int srcO, int io, int transpose, int flip,
                const float* kscale, float* dst, size_t nt_stride, PackJob* job);
// jobs_dev: device copy of the array (blk0 / nblk filled by dg_pack_layout on the host copy before the upload)
unsigned dg_pack_layout(PackJob* jobs_host, int njobs);
int dg_pack_weights_batch(const PackJob* jobs_dev, int njobs, unsigned nblocks, hipStream_t st);

// weight packing: src is HWIO (io=0) or HWOI (io=1, Conv2DTranspose layout)
// roles: if transpose==0 the GEMM K index is the source I axis and N the O axis
// (forward conv); if transpose==1 K is the source O axis and N the I axis
// (backward-data).  flip reverses the tap order.  kscale (optional) multiplies
// by a per-K-channel factor (BN scale folded into backward-data weights).
int dg_pack_weights(const ConvPlan& pl, const float* src, int srcI, int srcO, int io, int transpose, int flip,
                    const float* kscale, float* dst, hipStream_t st);

int dg_conv_igemm(const ConvPlan& pl, const ConvArgs& a, hipStream_t st);
// wave-private 3x3 kernel (igemm_wp.hip): no workgroup barrier in steady state; reads the 8-channel-chunk plan's panel
bool dg_conv_igemm_wp_supported(const ConvPlan& pl, const ConvArgs& a, bool force);
int dg_conv_igemm_wp(const ConvPlan& pl, const ConvArgs& a, hipStream_t st);
int dg_conv_igemm_tile(const ConvPlan& pl, const ConvArgs& a, hipStream_t st);
// the launch for (pl, a) can carry the fused one-channel head (Epilogue::head_*): 32 -> 32, 8-channel-chunk 3x3 kernel
bool dg_conv_igemm_head_supported(const ConvPlan& pl, const ConvArgs& a);
// Winograd F(2x2,3x3) kernel (igemm_wino.hip): plan variant 9, panel [nt][chunk][16 frequencies][32][8] of G g G^T
ConvPlan dg_plan_conv_wino(int Cin, int Cout);
bool dg_conv_wino_supported(const ConvPlan& pl, const ConvArgs& a);
const char* dg_conv_wino_name(const ConvArgs& a);
int dg_conv_wino(const ConvPlan& pl, const ConvArgs& a, hipStream_t st);
// name of the kernel instantiation dg_conv_igemm launches for (pl, a), as rocprofv3 prints it
void dg_conv_igemm_name(const ConvPlan& pl, const ConvArgs& a, char* buf, size_t cap);
int dg_conv_direct(int KS, const ConvArgs& a, hipStream_t st);

// wgrad: returns number of chunks used through *nchunks; part must hold
// dg_wgrad_part_floats() floats.
size_t dg_wgrad_part_floats(int KS, int B, int H, int W, int Cin, int Cout);
int dg_wgrad(int KS, const WgradArgs& a, int* nchunks, hipStream_t st);
// out[(tap,ci,co)] (+)= scale[co] * sum_chunks part ; raw (optional) gets the unscaled sum.
// oi=1 writes [tap][co][ci] (Conv2DTranspose kernel layout) instead of [tap][ci][co].
// slab reduction and (colpart non-null) the column-sum finish of the same weight-gradient launch, as ONE launch
int dg_wgrad_finish(const float* part, int nchunks, int ntaps, int Cin, int Cout, const float* scale, float* out,
                    float* raw, int accumulate, int oi, const float* colpart, int colC, const float* colscale,
                    float* colout, float* colraw, hipStream_t st);
// ... with a number of partial column rows that differs from the number of slabs (deconv_wgrad.hip: one row per tap)
int dg_wgrad_finish_rows(const float* part, int nchunks, int ntaps, int Cin, int Cout, const float* scale, float* out,
                         float* raw, int accumulate, int oi, const float* colpart, int colrows, int colC,
                         const float* colscale, float* colout, float* colraw, hipStream_t st);
int dg_wgrad_reduce(const float* part, int nchunks, int ntaps, int Cin, int Cout, const float* scale, float* out,
                    float* raw, int accumulate, int oi, hipStream_t st);

// bf16 matrix pipe (wgrad_bf16.hip, BASELINE configs[3]): operands rounded to bf16 while staged, fp32 accumulation;
// same slab format, column-sum rows ([nchunks][Cout], samples b < colB) and finish launch as dg_wgrad
bool dg_wgrad_bf16_supported(int KS, int Cin, int Cout);
size_t dg_wgrad_bf16_part_floats(int KS, int B, int H, int W, int Cin, int Cout);
int dg_wgrad_bf16(int KS, const WgradArgs& a, int* nchunks, hipStream_t st);

// small-Cin wgrad (VALU), same slab format as dg_wgrad
size_t dg_wgrad_small_part_floats(int KS, int B, int H, int W, int Cin, int Cout);
int dg_wgrad_small(int KS, const WgradArgs& a, int* nchunks, hipStream_t st);
