// NHWC implicit-GEMM convolution on the CDNA4 bf16 matrix cores (v_mfma_f32_32x32x16_bf16), fp32 accumulation.
//
// BASELINE configs[3] ("bf16 weights with fp32 accumulate", SURVEY.md 8d: bf16 weights AND activations into the MFMA,
// fp32 master weights / Adam): the same Conv2D / backward-data / penalty u-forward / transposed-conv call sites as
// igemm_conv.hip (GT:286-308, 319-338), selected per layer by a ConvPlan with bf16 = 1.  Activations stay fp32 in HBM;
// they are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while the halo tile is committed to LDS, the weights are packed as
// bf16 panels once per update (dg_pack_weights*, PackJob::bf16), and everything after the contraction -- bias, BN
// affine, FiLM, ReLU, residual, masks, pooling: the shared epilogue -- is fp32.
//
// Mapping: as the fp32 kernel (16x16-pixel tile x 32 output channels per workgroup, one wave per SIMD owning 4 x 16
// pixels = two 32-pixel MFMA tiles), but one MFMA contracts 16 input channels, so a chunk is CK = 32 channels
// = two MFMAs per tap and pixel tile.  An LDS row is 32 bf16 + 16 bytes of padding = 80 bytes, the same row pitch as
// the fp32 kernel's 16-float rows: lane (r, kb) of a wave reads the 16 bytes [8 kb, 8 kb + 8) channels of pixel /
// weight row r, which is exactly the K-slice the instruction expects from that lane on both operands.  At 16x the
// fp32 matrix rate the contraction is a few per cent of the time: this kernel lives on the HBM / store side of the
// roofline, one item per workgroup, no persistence.
#include <stdlib.h>

#include "common.h"
#include "epilogue.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int KS, int TAPG>
__global__ __launch_bounds__(256, 2) void igemm_bf16_kernel(const ConvArgs a) {
  constexpr int MF = 32, NT = 32, MT = 2, CK = 32;
  constexpr int PAD = KS / 2;
  constexpr int TW = 16 + KS - 1;
  constexpr int PIXT = TW * TW;
  constexpr int NTAPS = KS * KS;
  constexpr int NG = NTAPS / TAPG;
  constexpr int ROWB = 80;   // bytes per LDS row: 32 bf16 + 16 bytes of padding (conflict-free 16-byte reads of 16 rows)
  constexpr int XV = CK / 4;  // float4 pieces of one pixel's chunk in global memory
  constexpr int XTOT = PIXT * XV;
  constexpr int XPIECES = (XTOT + 255) / 256;
  constexpr int WV = CK / 8;  // 16-byte pieces (8 bf16) of one packed weight row
  constexpr int WTOT = TAPG * NT * WV;
  constexpr int WPIECES = (WTOT + 255) / 256;
  static_assert(NTAPS % TAPG == 0, "tap grouping");
  typedef f32x16 acc_t;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* xs = reinterpret_cast<char*>(smem);      // [PIXT][ROWB]
  char* ws = xs + PIXT * ROWB;                   // [TAPG][NT][ROWB]

  const int tid = threadIdx.x;
  const int tilesX = (a.W + 15) >> 4, tilesY = (a.H + 15) >> 4;
  // work item -> (pixel tile, channel tile): the XCD-aware order of igemm_conv.hip (each XCD walks a contiguous eighth
  // of the pixel tiles with the channel tile fastest)
  const unsigned nNTall = (unsigned)a.lgy, nPix = (unsigned)a.lgx;
  const unsigned id = blockIdx.x;
  int t, ntile;
  if ((nPix & 7u) == 0) {
    const unsigned x = id & 7u, sl = id >> 3;
    ntile = (int)(sl % nNTall);
    t = (int)(x * (nPix >> 3) + sl / nNTall);
  } else {
    t = (int)(id % nPix);
    ntile = (int)(id / nPix);
  }
  const int tx0 = (t % tilesX) * 16;
  t /= tilesX;
  const int ty0 = (t % tilesY) * 16;
  const int b = t / tilesY;
  const int ngrp = a.groups > 1 ? a.groups : 1;
  const int nNTg = (int)nNTall / ngrp;
  const int grp = ntile / nNTg;
  ntile -= grp * nNTg;
  const __bf16* wbase = reinterpret_cast<const __bf16*>(a.groups > 1 ? a.w_group[grp] : a.w);
  const long out_goff = a.groups > 1 ? a.out_group_off[grp] : 0;
  const int n0 = ntile * NT;
  const int nCC = (a.Cin + CK - 1) / CK;
  const int NS = nCC * NG;
  const float* inb = a.in.p + (long)b * a.in.sB;

  f32x4 xr[XPIECES];
  u32x4 wr[WPIECES];
  auto coff = [&](int cc) -> long {     // gathered K, see ConvArgs::cpt
    if (a.cpt > 0) {
      const int run = cc / a.cpt;
      return a.in_run_off[run] + (long)(cc - run * a.cpt) * CK;
    }
    return (long)cc * CK;
  };
  auto prefetch = [&](int s) {
    const int cc = s / NG, tg = s - cc * NG;
    if (tg == 0) {
#pragma unroll
      for (int i = 0; i < XPIECES; ++i) {
        const int q = tid + i * 256;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q < XTOT) {
          const int pix = q / XV, part = q - pix * XV;
          const int ly = pix / TW, lx = pix - ly * TW;
          const int iy = ty0 + ly - PAD, ix = tx0 + lx - PAD;
          const int c = cc * CK + part * 4;
          if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && c < a.Cin)
            v = *reinterpret_cast<const f32x4*>(inb + (long)iy * a.in.sY + (long)ix * a.in.sX + coff(cc) + part * 4);
        }
        xr[i] = v;
      }
    }
    const __bf16* wsrc = wbase + ((size_t)((size_t)ntile * nCC + cc) * NTAPS + (size_t)tg * TAPG) * (NT * CK);
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) {
      const int q = tid + i * 256;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (q < WTOT) v = *reinterpret_cast<const u32x4*>(wsrc + (size_t)q * 8);
      wr[i] = v;
    }
  };
  auto commit = [&](int s) {
    const int cc = s / NG, tg = s - cc * NG;
    (void)cc;
    if (tg == 0) {
#pragma unroll
      for (int i = 0; i < XPIECES; ++i) {
        const int q = tid + i * 256;
        if (q < XTOT) {
          const int pix = q / XV, part = q - pix * XV;
          // the activation operand becomes bf16 here: plain casts = v_cvt_pk_bf16_f32, round to nearest even
          const bf16x4 h4 = __builtin_convertvector(xr[i], bf16x4);
          *reinterpret_cast<u32x2*>(xs + pix * ROWB + part * 8) = __builtin_bit_cast(u32x2, h4);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) {
      const int q = tid + i * 256;
      if (q < WTOT) {
        const int row = q / WV, part = q - row * WV;
        *reinterpret_cast<u32x4*>(ws + row * ROWB + part * 16) = wr[i];
      }
    }
  };

  const int lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, h = lane >> 5;   // h: which 8 of the 16 k-values of an MFMA this lane carries
  int apix[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int py = 4 * wv + 2 * mt + (r >> 4), px = r & 15;
    apix[mt] = (py * TW + px) * ROWB + 16 * h;
  }
  const int boff = r * ROWB + 16 * h;

  acc_t acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[mt][j] = 0.f;

  prefetch(0);
  for (int s = 0; s < NS; ++s) {
    __syncthreads();
    commit(s);
    __syncthreads();
    if (s + 1 < NS) prefetch(s + 1);
    const int tg = s % NG;
#pragma unroll
    for (int tl = 0; tl < TAPG; ++tl) {
      const int tap = (TAPG == NTAPS) ? tl : (tg * TAPG + tl);
      const int ty = tap / KS, tx = tap - ty * KS;
      const int tapoff = (ty * TW + tx) * ROWB;
#pragma unroll
      for (int sub = 0; sub < CK / 16; ++sub) {
        const bf16x8 bw = *reinterpret_cast<const bf16x8*>(ws + tl * (NT * ROWB) + boff + 32 * sub);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const bf16x8 ax = *reinterpret_cast<const bf16x8*>(xs + apix[mt] + tapoff + 32 * sub);
          // weight fragment first: D[channel][pixel], the layout the shared epilogue expects
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw, ax, acc[mt], 0, 0, 0);
        }
      }
    }
  }
#include "igemm_epilogue.inc"
}

// ---------------------------------------------------------------------------
// fp32 operands on the bf16 matrix pipe: split products
// ---------------------------------------------------------------------------
// Every fp32 value is the exact sum of three bf16 numbers, x = x0 + x1 + x2 (x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1): 3 x 8 significand bits cover fp32's 24).  A product of two such sums is nine bf16 x bf16
// products, each exact in the MFMA's fp32 accumulator; keeping the six with the largest weights (x0w0, x0w1, x1w0,
// x0w2, x2w0, x1w1) drops terms below 2^-24 of |x||w| -- the size of fp32's own rounding of the product -- and keeping
// three (x0w0, x0w1, x1w0) drops terms below 2^-16.  Six v_mfma_f32_32x32x16_bf16 contract 16 channels in 192 cycles per
// SIMD where eight v_mfma_f32_32x32x2_f32 take 512: the fp32 convolution at 2.7x (six terms) or 5.3x (three terms) the
// matrix rate, with fp32 accumulation throughout.  NPL = planes kept per operand (3 -> six products, 2 -> three).
// Selected per context (depgan_config.f32_split); never the default: whether this counts as "fp32" is the reader's
// call, DESIGN.md section 4 gives the measured errors next to the native pipe's.
// Layout: a chunk is CK = 16 channels = one MFMA per tap, pixel tile and product; an LDS row is 16 bf16 + 16 bytes of
// padding = 48 bytes (16 consecutive rows start in 16 different 16-byte bank groups); planes are stored one after
// the other.  The weights are split when they are packed (PackJob::bf16 = number of planes), the activations while the
// halo tile is committed to LDS.
// LDS row pitch of the split kernel: 48 bytes (16 bf16 + 16 bytes of padding: conflict-free 16-byte reads) where two
// workgroups fit a CU anyway (two planes: 59 KB per 3x3 stage), 32 bytes unpadded (2-way conflicts on the fragment reads,
// measured -1.5 %) where only that lets a second workgroup in (three planes: 88 KB -> 59 KB; measured conv class
// 41.7 -> 33.9 ms per step)
#define SPLIT_ROWB(NPL) ((NPL) == 3 ? 32 : 48)
template <int KS, int TAPG, int NPL>
__global__ __launch_bounds__(256, 2) void igemm_split_kernel(const ConvArgs a) {
  constexpr int MF = 32, NT = 32, MT = 2, CK = 16;
  constexpr int PAD = KS / 2;
  constexpr int TW = 16 + KS - 1;
  constexpr int PIXT = TW * TW;
  constexpr int NTAPS = KS * KS;
  constexpr int NG = NTAPS / TAPG;
  constexpr int ROWB = SPLIT_ROWB(NPL);
  constexpr int XV = CK / 4;
  constexpr int XTOT = PIXT * XV;
  constexpr int XPIECES = (XTOT + 255) / 256;
  constexpr int WV = CK / 8;                       // 16-byte pieces of one packed weight row of one plane
  constexpr int WTOT = NPL * TAPG * NT * WV;       // the NPL planes of a stage's panel are contiguous in global memory
  constexpr int WPIECES = (WTOT + 255) / 256;
  constexpr int XPLANE = PIXT * ROWB, WPLANE = TAPG * NT * ROWB;
  static_assert(NTAPS % TAPG == 0, "tap grouping");
  typedef f32x16 acc_t;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* xs = reinterpret_cast<char*>(smem);      // [NPL][PIXT][ROWB]
  char* ws = xs + NPL * XPLANE;                  // [NPL][TAPG][NT][ROWB]

  const int tid = threadIdx.x;
  const int tilesX = (a.W + 15) >> 4, tilesY = (a.H + 15) >> 4;
  const unsigned nNTall = (unsigned)a.lgy, nPix = (unsigned)a.lgx;
  const int nCC = (a.Cin + CK - 1) / CK;
  const int NS = nCC * NG;
  const int lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  int apix[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int py = 4 * wv + 2 * mt + (r >> 4), px = r & 15;
    apix[mt] = (py * TW + px) * ROWB + 16 * h;
  }
  const int boff = r * ROWB + 16 * h;

  for (unsigned id = blockIdx.x; id < nPix * nNTall; id += gridDim.x) {
  int t, ntile;
  if ((nPix & 7u) == 0) {
    const unsigned x = id & 7u, sl = id >> 3;
    ntile = (int)(sl % nNTall);
    t = (int)(x * (nPix >> 3) + sl / nNTall);
  } else {
    t = (int)(id % nPix);
    ntile = (int)(id / nPix);
  }
  const int tx0 = (t % tilesX) * 16;
  t /= tilesX;
  const int ty0 = (t % tilesY) * 16;
  const int b = t / tilesY;
  const int ngrp = a.groups > 1 ? a.groups : 1;
  const int nNTg = (int)nNTall / ngrp;
  const int grp = ntile / nNTg;
  ntile -= grp * nNTg;
  const __bf16* wbase = reinterpret_cast<const __bf16*>(a.groups > 1 ? a.w_group[grp] : a.w);
  const long out_goff = a.groups > 1 ? a.out_group_off[grp] : 0;
  const int n0 = ntile * NT;
  const float* inb = a.in.p + (long)b * a.in.sB;

  f32x4 xr[XPIECES];
  u32x4 wr[WPIECES];
  auto coff = [&](int cc) -> long {
    if (a.cpt > 0) {
      const int run = cc / a.cpt;
      return a.in_run_off[run] + (long)(cc - run * a.cpt) * CK;
    }
    return (long)cc * CK;
  };
  auto prefetch = [&](int s) {
    const int cc = s / NG, tg = s - cc * NG;
    if (tg == 0) {
#pragma unroll
      for (int i = 0; i < XPIECES; ++i) {
        const int q = tid + i * 256;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q < XTOT) {
          const int pix = q / XV, part = q - pix * XV;
          const int ly = pix / TW, lx = pix - ly * TW;
          const int iy = ty0 + ly - PAD, ix = tx0 + lx - PAD;
          const int c = cc * CK + part * 4;
          if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && c < a.Cin)
            v = *reinterpret_cast<const f32x4*>(inb + (long)iy * a.in.sY + (long)ix * a.in.sX + coff(cc) + part * 4);
        }
        xr[i] = v;
      }
    }
    // packed: [nt][cc][tap group][plane][tap in group][n][k]: one stage's planes are one contiguous block
    const __bf16* wsrc = wbase + ((size_t)((size_t)ntile * nCC + cc) * NG + tg) * (size_t)(NPL * TAPG * NT * CK);
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) {
      const int q = tid + i * 256;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (q < WTOT) v = *reinterpret_cast<const u32x4*>(wsrc + (size_t)q * 8);
      wr[i] = v;
    }
  };
  auto commit = [&](int s) {
    const int tg = s % NG;
    if (tg == 0) {
#pragma unroll
      for (int i = 0; i < XPIECES; ++i) {
        const int q = tid + i * 256;
        if (q < XTOT) {
          const int pix = q / XV, part = q - pix * XV;
          f32x4 rem = xr[i];
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            const bf16x4 h4 = __builtin_convertvector(rem, bf16x4);          // RNE
            *reinterpret_cast<u32x2*>(xs + pl * XPLANE + pix * ROWB + part * 8) = __builtin_bit_cast(u32x2, h4);
            if (pl + 1 < NPL) rem = rem - __builtin_convertvector(h4, f32x4);   // exact: the difference is representable
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) {
      const int q = tid + i * 256;
      if (q < WTOT) {
        const int row = q / WV, part = q - row * WV;      // row runs over [plane][tap][n]
        *reinterpret_cast<u32x4*>(ws + row * ROWB + part * 16) = wr[i];
      }
    }
  };

  acc_t acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[mt][j] = 0.f;

  prefetch(0);
  for (int s = 0; s < NS; ++s) {
    __syncthreads();
    commit(s);
    __syncthreads();
    if (s + 1 < NS) prefetch(s + 1);
    const int tg = s % NG;
#pragma unroll
    for (int tl = 0; tl < TAPG; ++tl) {
      const int tap = (TAPG == NTAPS) ? tl : (tg * TAPG + tl);
      const int ty = tap / KS, tx = tap - ty * KS;
      const int tapoff = (ty * TW + tx) * ROWB;
      bf16x8 bw[NPL], ax[NPL][MT];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        bw[pl] = *reinterpret_cast<const bf16x8*>(ws + pl * WPLANE + tl * (NT * ROWB) + boff);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          ax[pl][mt] = *reinterpret_cast<const bf16x8*>(xs + pl * XPLANE + apix[mt] + tapoff);
      }
      // smallest terms first, so that they meet an accumulator that is still small within this tap
#pragma unroll
      for (int ord = 2 * (NPL - 1) > 2 ? 2 : NPL - 1; ord >= 0; --ord)
#pragma unroll
        for (int pa = 0; pa < NPL; ++pa) {
          const int pw = ord - pa;
          if (pw < 0 || pw >= NPL) continue;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[pw], ax[pa][mt], acc[mt], 0, 0, 0);
        }
    }
  }
#include "igemm_epilogue.inc"
  }
}

// ---------------------------------------------------------------------------
// plans and launcher
// ---------------------------------------------------------------------------
ConvPlan dg_plan_conv_bf16(int KS, int Cin, int Cout) {
  ConvPlan p = dg_plan_conv(KS, Cin, Cout);
  // layers the bf16 kernel does not cover keep the fp32 matrix pipe (their weights are bf16-valued all the same):
  // Cout not a multiple of 32 (the 16-channel critic layers), edge layers (Cin < 8), odd channel counts
  if (p.variant < 0 || (Cout % 32) != 0 || Cin < 8 || (Cin % 4) != 0 || !(KS == 1 || KS == 3 || KS == 5)) return p;
  p.bf16 = 1;
  p.MF = 32;
  p.NT = 32;
  p.CK = 32;
  p.nNT = cdiv(Cout, 32);
  p.nCC = cdiv(Cin, 32);
  p.variant = KS == 3 ? 100 : (KS == 5 ? 101 : 102);
  const size_t elems = (size_t)p.nNT * p.nCC * KS * KS * p.NT * p.CK;
  p.packedFloats = (elems + 1) / 2;     // bf16 elements, counted in 4-byte units for the allocator
  return p;
}

// split plans: bf16 = number of planes (2 or 3); Cin a multiple of 4 and >= 8 like the bf16 plans; Cout a multiple of 16 --
// a 16-channel layer (the critics' first 5x5 convolutions) runs as half of a 32-channel tile with zero weight rows: twice
// the MFMAs it needs, still a third of the cycles the fp32 pipe's 16x16x4 form takes for it
ConvPlan dg_plan_conv_split(int KS, int Cin, int Cout, int planes) {
  ConvPlan p = dg_plan_conv(KS, Cin, Cout);
  if (p.variant < 0 || (Cout % 16) != 0 || Cin < 8 || (Cin % 4) != 0 || !(KS == 1 || KS == 3 || KS == 5) ||
      (planes != 2 && planes != 3))
    return p;
  p.bf16 = planes;
  p.MF = 32;
  p.NT = 32;
  p.CK = 16;
  p.nNT = cdiv(Cout, 32);
  p.nCC = cdiv(Cin, 16);
  p.variant = 200 + (KS == 3 ? 0 : (KS == 5 ? 1 : 2));
  const size_t elems = (size_t)planes * p.nNT * p.nCC * KS * KS * p.NT * p.CK;
  p.packedFloats = (elems + 1) / 2;
  return p;
}

template <int KS, int TAPG, int NPL>
static int launch_split(const ConvArgs& a, hipStream_t st) {
  constexpr int TW = 16 + KS - 1;
  constexpr size_t lds_k = (size_t)NPL * (TW * TW + TAPG * 32) * SPLIT_ROWB(NPL);
  constexpr size_t lds_e = (size_t)4 * 64 * (32 + 4) * sizeof(float);
  constexpr size_t lds = lds_k > lds_e ? lds_k : lds_e;
  static DgOncePerDevice once;
  if (once.need()) {
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_split_kernel<KS, TAPG, NPL>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  ConvArgs b = a;
  b.lgx = cdiv(a.W, 16) * cdiv(a.H, 16) * a.B;
  b.lgy = cdiv(a.Cout, 32) * (a.groups > 1 ? a.groups : 1);
  const long total = (long)b.lgx * b.lgy;
  const long per_cu = (long)((160 * 1024) / lds) > 0 ? (long)((160 * 1024) / lds) : 1;
  const long cap = 256L * (per_cu > 2 ? 2 : per_cu);
  const long G = total < cap ? total : cap;        // persistent: every workgroup resident
  hipLaunchKernelGGL((igemm_split_kernel<KS, TAPG, NPL>), dim3((unsigned)G), dim3(256), lds, st, b);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

template <int KS, int TAPG>
static int launch_bf16(const ConvArgs& a, hipStream_t st) {
  constexpr int TW = 16 + KS - 1;
  constexpr size_t lds_k = (size_t)(TW * TW + TAPG * 32) * 80;
  constexpr size_t lds_e = (size_t)4 * 64 * (32 + 4) * sizeof(float);
  constexpr size_t lds = lds_k > lds_e ? lds_k : lds_e;
  static DgOncePerDevice once;
  if (once.need()) {
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_bf16_kernel<KS, TAPG>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  ConvArgs b = a;
  b.lgx = cdiv(a.W, 16) * cdiv(a.H, 16) * a.B;
  b.lgy = cdiv(a.Cout, 32) * (a.groups > 1 ? a.groups : 1);
  const long total = (long)b.lgx * b.lgy;
  hipLaunchKernelGGL((igemm_bf16_kernel<KS, TAPG>), dim3((unsigned)total), dim3(256), lds, st, b);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

int dg_conv_igemm_bf16(const ConvPlan& pl, const ConvArgs& a, hipStream_t st) {
  if (pl.variant >= 200) {
    const int ks = pl.variant - 200;
    if (pl.bf16 == 3) {
      if (ks == 0) return launch_split<3, 9, 3>(a, st);
      if (ks == 1) return launch_split<5, 5, 3>(a, st);
      return launch_split<1, 1, 3>(a, st);
    }
    if (ks == 0) return launch_split<3, 9, 2>(a, st);
    if (ks == 1) return launch_split<5, 5, 2>(a, st);
    return launch_split<1, 1, 2>(a, st);
  }
  switch (pl.variant) {
    case 100: return launch_bf16<3, 9>(a, st);
    case 101: return launch_bf16<5, 5>(a, st);
    case 102: return launch_bf16<1, 1>(a, st);
  }
  dg_set_error("dg_conv_igemm_bf16: no bf16 variant for KS=%d", pl.KS);
  return DG_ERR_UNSUPPORTED;
}
