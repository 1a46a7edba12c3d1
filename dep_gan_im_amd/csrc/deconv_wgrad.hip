// Weight gradient of the 2x2 / stride-2 transposed convolution (Conv2DTranspose of GT:308), all four taps and the bias /
// BN-beta column sums in ONE launch, gfx950.
//
//   dW[tap][co][ci] = sum_p dOut[2p + tap][co] * in[p][ci]           colsum[co] = sum over all output pixels dOut[.][co]
//
// As a GEMM per tap: M = Cout, N = Cin, K = the B H W input pixels -- both operands are NHWC tensors read along the
// pixel axis, i.e. K is the slow axis of both and a pixel's channels are contiguous.  The general weight-gradient
// kernel (wgrad_dma_kernel<.,1,1,.>) ran this as four launches per layer, each re-reading the input, at 0.32 - 0.40 of
// the MFMA peak, plus a separate streaming pass for the column sums (0.36 ms per step): 671 MB of operands became
// 1.6 GB of traffic.  Here every byte is read once from HBM:
//  * no LDS at all.  v_mfma_f32_16x16x4_f32 takes A[row = lane % 16][k = lane / 16] and B[k = lane / 16][col = lane % 16]:
//    with k = 4 consecutive pixels and lane % 16 = a group of channels, ONE 16-byte global load per lane (pixel
//    lane / 16, channels 4 (lane % 16) .. + 3 of a 64-channel piece) is the A (or B) fragment of FOUR 16-row tiles at
//    once -- tile j of the piece holds the channels 4 r + j.  The tiles are interleaved in the channel axis; the slab
//    writer un-interleaves them (four tiles = four consecutive channels = one 16-byte store).
//  * wave w of a workgroup owns tap w (its own dOut pixel grid) and the whole Cout x Cin (or Cout x Cin / 2) block of
//    that tap in accumulators; the four waves read the same input pixels (second to fourth reader hit the cache).
//  * a workgroup walks a contiguous range of pixels with a ring of DEPTH fragment stages in flight (plain global
//    loads under compiler-counted vmcnt: every load is unconditional, the last ones re-read the final step).
//  * the column sums ride along: the A fragments are the dOut values, 4 adds per 16-byte piece and k-step.
//  * partial slabs [chunk][tap][ci][co] + partial column rows [chunk x tap][Cout] go to the same reduction launch as
//    the other weight gradients (slab_reduce_kernel: no float atomics, bit-reproducible).
#include "common.h"
#include "deconv_fwd.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

template <class F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// fragment of an operand with NT 16-channel tiles: NT / 4 pieces of 64 channels (16 bytes per lane) and, when
// NT % 4 == 2, one piece of 32 channels (8 bytes per lane)
template <int NT>
struct Frag {
  static constexpr int NF = NT / 4;
  static constexpr bool HALF = (NT % 4) == 2;
  static_assert(NT % 2 == 0 && NT >= 2, "tiles come in pieces of four or two");
  f32x4 q[NF > 0 ? NF : 1];
  f32x2 d;
  __device__ __forceinline__ float tile(int t) const { return t < 4 * NF ? q[t / 4][t % 4] : d[t - 4 * NF]; }
};
// lane-constant channel offset of a piece's first element, and channel of (tile t, row r)
template <int NT>
__device__ __forceinline__ void frag_load(Frag<NT>& f, const float* px /* pixel base + lane's channel origin */,
                                          int r) {
#pragma unroll
  for (int i = 0; i < Frag<NT>::NF; ++i) f.q[i] = *reinterpret_cast<const f32x4*>(px + 64 * i + 4 * r);
  if (Frag<NT>::HALF) f.d = *reinterpret_cast<const f32x2*>(px + 64 * Frag<NT>::NF + 2 * r);
}

constexpr int DEPTH = 4;   // fragment stages in flight

template <int NA, int NB>
__global__ __launch_bounds__(256, 1) void deconv_wgrad_kernel(DeconvWgradArgs a) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int tap = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kk = lane >> 4;
  const int chunk = blockIdx.x, ci0 = blockIdx.y * 16 * NB;
  const int steps = a.steps_per_wg;
  const unsigned step0 = (unsigned)chunk * (unsigned)steps;

  // lane-constant parts of the operand addresses: the lane's pixel of a k-step (kk) and its channel group
  const float* pa = a.dout.p + (long)(tap >> 1) * a.dout.sY + (long)(tap & 1) * a.dout.sX + (long)kk * 2 * a.dout.sX;
  const float* pb = a.in + (long)kk * a.Cin + ci0;
  // first pixel of k-step s (4 consecutive input pixels of one image row; H, W powers of two): scalar arithmetic
  auto a_off = [&](unsigned s) -> size_t {
    const unsigned p = s * 4u;
    const unsigned j = p & (unsigned)(a.W - 1), t = p >> a.lgW;
    const unsigned i = t & (unsigned)(a.H - 1), b = t >> a.lgH;
    return (size_t)(b * (unsigned)a.dout.sB + i * (2u * (unsigned)a.dout.sY) + j * (2u * (unsigned)a.dout.sX));
  };
  auto b_off = [&](unsigned s) -> size_t { return (size_t)s * 4u * (unsigned)a.Cin; };

  Frag<NA> fa[DEPTH];
  Frag<NB> fb[DEPTH];
  f32x4 acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  Frag<NA> cs;   // running column sums of the lane's dOut channels over its pixels
#pragma unroll
  for (int i = 0; i < (Frag<NA>::NF > 0 ? Frag<NA>::NF : 1); ++i) cs.q[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  cs.d = (f32x2){0.f, 0.f};

  const unsigned last = step0 + (unsigned)steps - 1u;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const unsigned s = min(step0 + (unsigned)d, last);
    frag_load(fa[d], pa + a_off(s), r);
    frag_load(fb[d], pb + b_off(s), r);
  }
  __builtin_amdgcn_sched_barrier(0);
  // steps is a multiple of DEPTH (the launcher sees to it)
  for (int g = 0; g < steps; g += DEPTH) {
    sfor<DEPTH>([&](auto dc) __attribute__((always_inline)) {
      constexpr int d = decltype(dc)::value;
      // the MFMAs of this stage ...
      sfor<NA * NB>([&](auto tc) __attribute__((always_inline)) {
        constexpr int ta = decltype(tc)::value / NB, tb = decltype(tc)::value % NB;
        acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[d].tile(ta), fb[d].tile(tb), acc[ta][tb], 0, 0, 0);
      });
#pragma unroll
      for (int i = 0; i < Frag<NA>::NF; ++i) cs.q[i] += fa[d].q[i];
      if (Frag<NA>::HALF) cs.d += fa[d].d;
      // ... then its registers take the fragments of DEPTH steps ahead (sched_barrier: the scheduler would sink the
      // loads to their use, one ring round later, and wait for each with vmcnt(0))
      __builtin_amdgcn_sched_barrier(0);
      const unsigned s = min(step0 + (unsigned)(g + d + DEPTH), last);
      frag_load(fa[d], pa + a_off(s), r);
      frag_load(fb[d], pb + b_off(s), r);
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  // ---- partial slab [tap][ci][co] of this chunk: D[row = 4 (lane / 16) + e][col = lane % 16] per tile ----
  // tile ta of a 64-channel piece q holds the output channels 64 q + 4 row + (ta % 4): the four tiles of a piece are
  // four consecutive channels -> one 16-byte store per (piece, tb, e); input channel of (tb, col) likewise
  float* slab = a.part + (size_t)chunk * 4 * a.Cin * a.Cout + (size_t)tap * a.Cin * a.Cout;
#pragma unroll
  for (int tb = 0; tb < NB; ++tb) {
    const int ci = ci0 + (tb < 4 * Frag<NB>::NF ? 64 * (tb / 4) + 4 * r + tb % 4
                                                : 64 * Frag<NB>::NF + 2 * r + (tb - 4 * Frag<NB>::NF));
    float* row = slab + (size_t)ci * a.Cout;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int q = 0; q < Frag<NA>::NF; ++q) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[4 * q + j][tb][e];
        *reinterpret_cast<f32x4*>(row + 64 * q + 4 * (4 * kk + e)) = v;
      }
      if (Frag<NA>::HALF) {
        f32x2 v;
        v[0] = acc[4 * Frag<NA>::NF][tb][e];
        v[1] = acc[4 * Frag<NA>::NF + 1][tb][e];
        *reinterpret_cast<f32x2*>(row + 64 * Frag<NA>::NF + 2 * (4 * kk + e)) = v;
      }
    }
  }
  // ---- partial column row of (chunk, tap): sum the four pixel lanes of a channel group, lanes kk == 0 write ----
  if (a.colpart && blockIdx.y == 0) {
    float* crow = a.colpart + ((size_t)chunk * 4 + tap) * a.Cout;
#pragma unroll
    for (int q = 0; q < Frag<NA>::NF; ++q) {
      f32x4 v = cs.q[q];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] += __shfl_xor(v[j], 16);
        v[j] += __shfl_xor(v[j], 32);
      }
      if (kk == 0) *reinterpret_cast<f32x4*>(crow + 64 * q + 4 * r) = v;
    }
    if (Frag<NA>::HALF) {
      f32x2 v = cs.d;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        v[j] += __shfl_xor(v[j], 16);
        v[j] += __shfl_xor(v[j], 32);
      }
      if (kk == 0) *reinterpret_cast<f32x2*>(crow + 64 * Frag<NA>::NF + 2 * r) = v;
    }
  }
}

int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// workgroups (= partial slabs) of a launch: the largest count <= 2 per CU whose share of the k-steps is a whole
// number of ring rounds (small problems get fewer workgroups); 0 when the steps are not a multiple of the ring depth
int pick_chunks(long total_steps, int ny) {
  const int cus = dg_cu_count();
  const int want = 2 * cus / ny;
  for (int n = want; n >= 1; --n)
    if (total_steps % ((long)n * DEPTH) == 0) return n;
  return 0;
}

}  // namespace

bool dg_deconv_wgrad_supported(int B, int H, int W, int Cin, int Cout, TView in, TView dout) {
  if (const char* e = getenv("DEPGAN_DECONV_FUSED"))
    if (atoi(e) == 0) return false;
  if (!((Cin == 64 && Cout == 64) || (Cin == 96 && Cout == 96) || (Cin == 128 && Cout == 128))) return false;
  if (W < 4 || (W & (W - 1)) || (H & (H - 1)) || H < 1) return false;
  const long P = (long)B * H * W;
  if (P % 4 || (long)B * dout.sB > 0x7FFFFFFFL || P * Cin > 0x7FFFFFFFL) return false;
  if (in.sX != Cin || in.sY != (long)W * Cin || in.sB != (long)H * W * Cin) return false;
  if (((uintptr_t)in.p | (uintptr_t)dout.p) & 15) return false;
  if ((dout.sX | dout.sY | dout.sB) & 3) return false;
  return pick_chunks(P / 4, Cin == 128 ? 2 : 1) > 0;
}

size_t dg_deconv_wgrad_part_floats(int B, int H, int W, int Cin, int Cout) {
  const int n = pick_chunks((long)B * H * W / 4, Cin == 128 ? 2 : 1);
  return (size_t)(n > 0 ? n : 0) * 4 * Cin * Cout;
}

int dg_deconv_wgrad(DeconvWgradArgs a, int B, int* nchunks, hipStream_t st) {
  if (!dg_deconv_wgrad_supported(B, a.H, a.W, a.Cin, a.Cout, make_view(const_cast<float*>(a.in), a.H, a.W, a.Cin),
                                 a.dout)) {
    dg_set_error("dg_deconv_wgrad: shape %dx%dx%d %d->%d not covered by the fused kernel", B, a.H, a.W, a.Cin, a.Cout);
    return DG_ERR_UNSUPPORTED;
  }
  const int ny = a.Cin == 128 ? 2 : 1;
  const long total = (long)B * a.H * a.W / 4;
  const int n = pick_chunks(total, ny);
  a.steps_per_wg = (int)(total / n);
  a.lgW = ilog2_exact(a.W);
  a.lgH = ilog2_exact(a.H);
  *nchunks = n;
  if (a.Cin == 64) hipLaunchKernelGGL((deconv_wgrad_kernel<4, 4>), dim3(n, 1), dim3(256), 0, st, a);
  else if (a.Cin == 96) hipLaunchKernelGGL((deconv_wgrad_kernel<6, 6>), dim3(n, 1), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((deconv_wgrad_kernel<8, 4>), dim3(n, 2), dim3(256), 0, st, a);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}
