// NHWC fp32 implicit-GEMM convolution on the CDNA4 fp32 matrix cores.
//
// Replaces the Keras Conv2D(padding='same', strides=1) calls of the reference
// (GT:286, 294, 301; critic GT:319-338) and, with flipped / transposed packed
// weights, their backward-data and the gradient-penalty "u-forward" passes
// (SURVEY.md 8a rows A1, A2, A6).  No im2col buffer is materialised.
//
// Mapping (one workgroup = 256 threads = 4 waves, one wave per SIMD):
//   M = 16x16 output pixels of one sample (each wave: 4 rows x 16 px = 64 px)
//   N = NT output channels (NT = MF: one MFMA column tile per wave)
//   K = taps x Cin, walked as  Cin-chunk (CK channels) -> tap -> 4 MFMAs / b128
// LDS holds the (16+KS-1)^2 x CK input halo tile and the taps x NT x CK weight
// panel; rows are padded by 4 floats (CKP = CK+4) so the 16-byte fragment
// reads of the 32/16 pixel rows of one MFMA tile spread over the banks.
// A-fragment lane (r, h) reads 4 consecutive channels [4h, 4h+4) of pixel r:
// MFMA j of the group uses channel 4h+j on both operands, i.e. K is permuted
// identically for A and B, which a contraction does not care about.
// Global loads for stage s+1 are issued into registers before the MFMAs of
// stage s and written to LDS after them (one register set, T14-style).
#include <stdlib.h>

#include "common.h"
#include "epilogue.h"

// one packed weight element: fp32, or bf16 (RNE; plain cast = v_cvt_pk_bf16_f32) for the bf16 matrix pipe
__device__ __forceinline__ void store_packed(float* dst, size_t i, float v, int bf16) {
  if (bf16) reinterpret_cast<__bf16*>(dst)[i] = (__bf16)v;
  else dst[i] = v;
}
// split panels (igemm_split_kernel): element (nt, cc, tap, n, k) of the plain layout goes to NPL planes at
// [nt][cc][tap / tapg][plane][tap % tapg][n][k]; plane p holds bf16 of what the planes before it left over
__device__ __forceinline__ void store_split(float* dst, size_t nt_base, int cc, int tap, int n, int k, float v, int planes,
                                            int tapg, int ntaps, int NT, int CK) {
  const int ng = ntaps / tapg, tg = tap / tapg, tl = tap - tg * tapg;
  __bf16* d = reinterpret_cast<__bf16*>(dst) + nt_base + ((size_t)cc * ng + tg) * (size_t)(planes * tapg * NT * CK);
  float rem = v;
  for (int p = 0; p < planes; ++p) {
    const __bf16 hb = (__bf16)rem;
    d[((size_t)p * tapg + tl) * (NT * CK) + (size_t)n * CK + k] = hb;
    rem -= (float)hb;
  }
}

template <int MF>
struct Mfma;
template <>
struct Mfma<32> {
  typedef f32x16 acc_t;
  static constexpr int NREG = 16;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int j, int h) { return (j & 3) + 8 * (j >> 2) + 4 * h; }
};
template <>
struct Mfma<16> {
  typedef f32x4 acc_t;
  static constexpr int NREG = 4;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int j, int h) { return 4 * h + j; }
};

// PERS: persistent workgroups (the grid is smaller than the item count and every workgroup loops).  The two forms are
// separate instantiations on purpose: in the looping form the compiler hoists the per-thread staging geometry out of
// the loop (that is the point), which costs ~36 VGPRs and one resident workgroup per CU -- wrong for launches that
// run one item per workgroup.
// HEAD: the epilogue carries the fused one-channel 1x1 head (Epilogue::head_*); its own __global__ wrapper below so
// that the plain instantiations keep their code and their names in the profiles.
template <int MF, int KS, int CK, int TAPG, bool PERS, bool HEAD>
static __device__ __forceinline__ void igemm_conv_body(const ConvArgs& a) {
  constexpr int NT = MF;
  constexpr int PAD = KS / 2;
  constexpr int TW = 16 + KS - 1;
  constexpr int PIXT = TW * TW;
  constexpr int CKP = CK + 4;
  constexpr int NTAPS = KS * KS;
  constexpr int NG = NTAPS / TAPG;
  constexpr int KB = 256 / MF;  // channels covered by one 16-byte fragment read (4 MFMAs)
  constexpr int NSUB = CK / KB;
  constexpr int MT = 64 / MF;  // MFMA row tiles per wave
  constexpr int XV = CK / 4;
  constexpr int XTOT = PIXT * XV;
  constexpr int XPIECES = (XTOT + 255) / 256;
  constexpr int WTOT = TAPG * NT * XV;
  constexpr int WPIECES = (WTOT + 255) / 256;
  static_assert(NTAPS % TAPG == 0 && (TAPG == NTAPS || TAPG == KS), "tap grouping");
  static_assert(CK % KB == 0, "chunk must hold whole fragment reads");
  typedef typename Mfma<MF>::acc_t acc_t;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                // [PIXT][CKP]
  float* ws = smem + PIXT * CKP;   // [TAPG][NT][CKP]

  const int tid = threadIdx.x;
  unsigned long long* dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 16 : nullptr;
  if (dbg && tid == 0) {
    dbg[0] = __builtin_amdgcn_s_memtime();
    dbg[6] = __builtin_amdgcn_s_memrealtime();
    dbg[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
  }
  const int tilesX = (a.W + 15) >> 4, tilesY = (a.H + 15) >> 4;
  // Work item -> (pixel tile, channel tile).  The grid is one-dimensional; a workgroup walks the items id, id + G,
  // id + 2G ... (G = gridDim.x: every item once when G covers them all, persistent workgroups when the launcher
  // starts only as many as are resident).  Ids are dealt round-robin over the 8 XCDs (each with its own L2) in
  // dispatch order.  The channel tiles of one pixel tile read the same halo tile and neighbouring pixel tiles share
  // two halo columns / rows, so XCD x gets a contiguous eighth of the pixel tiles and walks it with the channel tile
  // fastest: id = 8 s + x  ->  channel tile s % nNT of pixel tile x (nPix / 8) + s / nNT.
  const unsigned nNTall = (unsigned)a.lgy, nPix = (unsigned)a.lgx;
  // Persistent form: everything about a staging piece that depends only on the thread -- its byte offset from the
  // halo origin, its LDS byte offset, its halo coordinates -- is computed here, once per workgroup (left to itself the
  // compiler re-derives most of it per item: 1074 of 1095 VALU instructions per item remained).
  constexpr int XFULL = XTOT / 256, WFULL = WTOT / 256;
  constexpr bool XPART = (XTOT % 256) != 0, WPART = (WTOT % 256) != 0;
  unsigned xgb[XPIECES], xlb[XPIECES], wlb[WPIECES];
  int xyx[XPIECES];
  if (PERS) {
#pragma unroll
    for (int i = 0; i < XPIECES; ++i) {
      const int q = min(tid + i * 256, XTOT - 1);
      const int pix = q / XV, part = q - pix * XV;
      const int ly = pix / TW, lx = pix - ly * TW;
      xgb[i] = 4u * (unsigned)(ly * (int)a.in.sY + lx * (int)a.in.sX + part * 4);
      xlb[i] = 4u * (unsigned)(pix * CKP + part * 4);
      xyx[i] = (ly << 16) | (lx << 8) | (part * 4);
    }
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) {
      const int q = min(tid + i * 256, WTOT - 1);
      const int row = q / XV, part = q - row * XV;
      wlb[i] = 4u * (unsigned)(row * CKP + part * 4);
    }
  }
  const bool in_last = tid < (XTOT % 256), w_last = tid < (WTOT % 256);
  const unsigned wob = 16u * (unsigned)tid;
  const unsigned xsb0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)xs;
  const unsigned wsb0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ws;
  unsigned id = blockIdx.x;
  do {
  int t, ntile;
  {
    if ((nPix & 7u) == 0) {
      const unsigned x = id & 7u, sl = id >> 3;
      ntile = (int)(sl % nNTall);
      t = (int)(x * (nPix >> 3) + sl / nNTall);
    } else {
      t = (int)(id % nPix);
      ntile = (int)(id / nPix);
    }
  }
  const int tx0 = (t % tilesX) * 16;
  t /= tilesX;
  const int ty0 = (t % tilesY) * 16;
  const int b = t / tilesY;
  // grouped launch: the channel-tile index also enumerates the groups (taps of a transposed convolution)
  const int ngrp = a.groups > 1 ? a.groups : 1;
  const int nNTg = (int)nNTall / ngrp;
  const int grp = ntile / nNTg;
  ntile -= grp * nNTg;
  const float* wbase = a.groups > 1 ? a.w_group[grp] : a.w;
  const long out_goff = a.groups > 1 ? a.out_group_off[grp] : 0;
  const int n0 = ntile * NT;
  const int nCC = (a.Cin + CK - 1) / CK;
  const int NS = nCC * NG;
  const float* inb = a.in.p + (long)b * a.in.sB;

  f32x4 xr[XPIECES];
  f32x4 wr[WPIECES];

  const bool interior = ty0 >= PAD && ty0 + 16 + PAD <= a.H && tx0 >= PAD && tx0 + 16 + PAD <= a.W;
  const char* halo0 = reinterpret_cast<const char*>(inb + ((long)(ty0 - PAD) * a.in.sY + (long)(tx0 - PAD) * a.in.sX));
  // first input channel (float offset from the pixel) of channel chunk cc; see ConvArgs::cpt for the gathered form
  auto coff = [&](int cc) -> long {
    if (a.cpt > 0) {
      const int run = cc / a.cpt;
      return a.in_run_off[run] + (long)(cc - run * a.cpt) * CK;
    }
    return (long)cc * CK;
  };
  auto prefetch = [&](int s) {
    const int cc = s / NG, tg = s - cc * NG;
    if (PERS) {
      if (tg == 0) {
        const char* src = halo0 + 4 * coff(cc);   // only dereferenced through in-image offsets
        if (interior && (cc + 1) * CK <= a.Cin) {
#pragma unroll
          for (int i = 0; i < XFULL; ++i) xr[i] = *reinterpret_cast<const f32x4*>(src + xgb[i]);
          if (XPART) {
            if (in_last) xr[XPIECES - 1] = *reinterpret_cast<const f32x4*>(src + xgb[XPIECES - 1]);
          }
        } else {
#pragma unroll
          for (int i = 0; i < XPIECES; ++i) {
            const int iy = ty0 + (xyx[i] >> 16) - PAD, ix = tx0 + ((xyx[i] >> 8) & 255) - PAD;
            const bool ok = (i < XFULL || in_last) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W &&
                            (cc * CK + (xyx[i] & 255)) < a.Cin;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(src + xgb[i]);
            xr[i] = v;
          }
        }
      }
      const char* wsrc = reinterpret_cast<const char*>(
          wbase + ((size_t)((size_t)ntile * nCC + cc) * NTAPS + (size_t)tg * TAPG) * (NT * CK));
#pragma unroll
      for (int i = 0; i < WFULL; ++i) wr[i] = *reinterpret_cast<const f32x4*>(wsrc + wob + 4096u * i);
      if (WPART) {
        if (w_last) wr[WPIECES - 1] = *reinterpret_cast<const f32x4*>(wsrc + wob + 4096u * (WPIECES - 1));
      }
      return;
    }
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
    const float* wsrc = wbase + ((size_t)((size_t)ntile * nCC + cc) * NTAPS + (size_t)tg * TAPG) * (NT * CK);
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) {
      const int q = tid + i * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q < WTOT) v = *reinterpret_cast<const f32x4*>(wsrc + (size_t)q * 4);
      wr[i] = v;
    }
  };
  auto commit = [&](int s) {
    const int cc = s / NG, tg = s - cc * NG;
    if (PERS) {
      (void)cc;
      if (tg == 0) {
#pragma unroll
        for (int i = 0; i < XFULL; ++i)
          *reinterpret_cast<__attribute__((address_space(3))) f32x4*>((size_t)(xsb0 + xlb[i])) = xr[i];
        if (XPART) {
          if (in_last)
            *reinterpret_cast<__attribute__((address_space(3))) f32x4*>((size_t)(xsb0 + xlb[XPIECES - 1])) = xr[XPIECES - 1];
        }
      }
#pragma unroll
      for (int i = 0; i < WFULL; ++i)
        *reinterpret_cast<__attribute__((address_space(3))) f32x4*>((size_t)(wsb0 + wlb[i])) = wr[i];
      if (WPART) {
        if (w_last)
          *reinterpret_cast<__attribute__((address_space(3))) f32x4*>((size_t)(wsb0 + wlb[WPIECES - 1])) = wr[WPIECES - 1];
      }
      return;
    }
    if (tg == 0) {
#pragma unroll
      for (int i = 0; i < XPIECES; ++i) {
        const int q = tid + i * 256;
        if (q < XTOT) {
          const int pix = q / XV, part = q - pix * XV;
          *reinterpret_cast<f32x4*>(xs + pix * CKP + part * 4) = xr[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) {
      const int q = tid + i * 256;
      if (q < WTOT) {
        const int row = q / XV, part = q - row * XV;
        *reinterpret_cast<f32x4*>(ws + row * CKP + part * 4) = wr[i];
      }
    }
  };

  const int lane = tid & 63, wv = tid >> 6;
  const int r = lane % MF, h = lane / MF;
  int apix[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int py, px;
    if (MF == 32) {
      py = 4 * wv + 2 * mt + (r >> 4);
      px = r & 15;
    } else {
      py = 4 * wv + mt;
      px = r;
    }
    apix[mt] = (py * TW + px) * CKP + 4 * h;
  }
  const int boff = r * CKP + 4 * h;

  acc_t acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < Mfma<MF>::NREG; ++j) acc[mt][j] = 0.f;

  prefetch(0);
  if (dbg && tid == 0) dbg[1] = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < NS; ++s) {
    __syncthreads();
    commit(s);
    __syncthreads();
    if (dbg && tid == 0 && s == 0) dbg[2] = __builtin_amdgcn_s_memtime();
    if (dbg && tid == 0 && s == 1) dbg[3] = __builtin_amdgcn_s_memtime();
    if (s + 1 < NS) prefetch(s + 1);
    const int tg = s % NG;
    // fragments of step q+1 are read from LDS before the MFMAs of step q issue
    // (two register sets; all indices static after unrolling)
    constexpr int NSTEP = TAPG * NSUB;
    f32x4 av[2][MT], bv[2];
    auto load_frag = [&](int q, f32x4* a_, f32x4& b_) {
      const int tl = q / NSUB, sub = q - tl * NSUB;
      const int tap = (TAPG == NTAPS) ? tl : (tg * TAPG + tl);
      const int ty = tap / KS, tx = tap - ty * KS;
      const int tapoff = (ty * TW + tx) * CKP;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a_[mt] = *reinterpret_cast<const f32x4*>(xs + apix[mt] + tapoff + sub * KB);
      b_ = *reinterpret_cast<const f32x4*>(ws + tl * (NT * CKP) + boff + sub * KB);
    };
    load_frag(0, av[0], bv[0]);
#pragma unroll
    for (int q = 0; q < NSTEP; ++q) {
      if (q + 1 < NSTEP) load_frag(q + 1, av[(q + 1) & 1], bv[(q + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = Mfma<MF>::run(bv[q & 1][j], av[q & 1][mt][j], acc[mt]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if (dbg && tid == 0) dbg[4] = __builtin_amdgcn_s_memtime();
#define EPI_HEAD HEAD
#include "igemm_epilogue.inc"
#undef EPI_HEAD
  if (dbg && tid == 0) {
    dbg[5] = __builtin_amdgcn_s_memtime();
    dbg[6] = __builtin_amdgcn_s_memrealtime() - dbg[6];
  }
  dbg = nullptr;   // stamps describe the first item only
  } while (PERS && (id += gridDim.x) < nPix * nNTall);
}

template <int MF, int KS, int CK, int TAPG, bool PERS>
__global__ __launch_bounds__(256, (PERS && KS == 3) ? (CK == 8 ? 4 : 3) : 1) void igemm_conv_kernel(const ConvArgs a) {
  igemm_conv_body<MF, KS, CK, TAPG, PERS, false>(a);
}

// gen_17 + gen_segmentation in one launch (32 -> 32 3x3, then 32 -> 1 and tanh): the 8-channel-chunk kernel
template <bool PERS>
__global__ __launch_bounds__(256, PERS ? 4 : 1) void igemm_conv_head_kernel(const ConvArgs a) {
  igemm_conv_body<32, 3, 8, 9, PERS, true>(a);
}

// ---------------------------------------------------------------------------
// variants
// ---------------------------------------------------------------------------
struct VariantInfo {
  int MF, KS, CK, TAPG;
};
static const VariantInfo kVariants[] = {
    {32, 3, 16, 9},   // 0
    {16, 3, 16, 9},   // 1
    {32, 5, 8, 25},   // 2
    {16, 5, 16, 25},  // 3
    {32, 1, 32, 1},   // 4
    {16, 1, 16, 1},   // 5
};

ConvPlan dg_plan_conv(int KS, int Cin, int Cout) {
  ConvPlan p;
  p.KS = KS;
  p.Cin = Cin;
  p.Cout = Cout;
  p.variant = -1;
  p.bf16 = 0;
  p.MF = (Cout % 32 == 0) ? 32 : 16;
  p.NT = p.MF;
  p.CK = 16;
  p.nNT = p.nCC = 0;
  p.packedFloats = 0;
  if (Cin < 8 || (Cin % 4) != 0 || Cout < 8) return p;  // direct kernel territory
  for (int i = 0; i < (int)(sizeof(kVariants) / sizeof(kVariants[0])); ++i)
    if (kVariants[i].MF == p.MF && kVariants[i].KS == KS) {
      p.variant = i;
      p.CK = kVariants[i].CK;
    }
  if (p.variant < 0) return p;
  p.nNT = cdiv(Cout, p.NT);
  p.nCC = cdiv(Cin, p.CK);
  p.packedFloats = (size_t)p.nNT * p.nCC * KS * KS * p.NT * p.CK;
  return p;
}

// The plan of a 3x3 convolution whose launches have `items` work items (pixel tiles x channel tiles at the batch it
// mostly runs at).  Large launches take 8-channel chunks: 29 KB of LDS and 124 VGPRs let FOUR workgroups share a CU
// instead of three -- one more wave per SIMD to run while the older ones stall -- at twice the chunk boundaries per item.
// Measured (profiles/r02_conv_experiments.md): -1.2 ... -1.9 % time from 1536 items up, +1 ... +3 % below
// (DEPGAN_IGEMM_CK8=0 keeps 16-channel chunks everywhere).
ConvPlan dg_plan_conv_items(int KS, int Cin, int Cout, long items) {
  ConvPlan p = dg_plan_conv(KS, Cin, Cout);
  const char* e = getenv("DEPGAN_IGEMM_CK8");
  const bool on = !(e && atoi(e) == 0);
  if (on && p.variant == 0 && KS == 3 && p.MF == 32 && (Cin % 8) == 0 && items >= 1536) {
    p.variant = 8;
    p.CK = 8;
    p.nCC = cdiv(Cin, p.CK);
    p.packedFloats = (size_t)p.nNT * p.nCC * KS * KS * p.NT * p.CK;
  }
  return p;
}

// Winograd F(2x2,3x3) plan (igemm_wino.hip): 8-channel chunks, 16 transform-domain panels per chunk
ConvPlan dg_plan_conv_wino(int Cin, int Cout) {
  ConvPlan p = dg_plan_conv(3, Cin, Cout);
  if (p.variant != 0 || (Cin % 8) != 0 || (Cout % 32) != 0) { p.variant = -1; return p; }
  p.variant = 9;
  p.CK = 8;
  p.nCC = Cin / 8;
  p.packedFloats = (size_t)p.nNT * p.nCC * 16 * p.NT * p.CK;
  return p;
}

// Persistent workgroups: as many as are resident per CU (3 by registers, fewer where the LDS tile is large), each
// walking ~total / G items.  The per-thread staging geometry and address arithmetic (about 550 instructions per
// item) is then computed once per workgroup.  Every workgroup of the grid MUST be resident: one that has to wait
// for a slot runs its whole share after the others have finished (752 workgroups at 2 resident per CU: +20 %).
// Measured per layer on one device (tools/cmp_persist.py): 3x3 layers +1..4 %, 5x5 layers +3..5 %, 1x1 -9 %.
// DEPGAN_IGEMM_PERSIST=<workgroups per CU> forces a setting for all shapes (0 = never) for A/B measurements.
// Returns the grid size: < total means the persistent instantiation.
static long igemm_grid(int KS, int CK, size_t lds, long total) {
  const char* e = getenv("DEPGAN_IGEMM_PERSIST");
  // the 3x3 persistent instantiation is compiled for 3 resident workgroups per CU (4 with 8-channel chunks); the
  // others take what their registers allow (2)
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > (KS == 3 ? (CK == 8 ? 4 : 3) : 2)) per_cu = (KS == 3 ? (CK == 8 ? 4 : 3) : 2);
  if (KS == 1) per_cu = 0;   // K is one chunk: nothing to amortise, measured -0.17 ms per step when persistent
  if (e) per_cu = atoi(e) < per_cu ? atoi(e) : per_cu;
  const long cap = 256L * per_cu;
  // any launch with more items than resident workgroups: also at 1.3 items per workgroup the persistent grid beats
  // dispatching the overflow as a second wave of workgroups (64x64 64->64 at batch 32, 1024 items: 85 -> 105 TFLOP/s)
  return (per_cu > 0 && total > cap) ? cap : total;
}

// name_out != nullptr: only report which instantiation the launch would take (profile records, bench.py's
// dominant-kernel line: the names rocprofv3 prints), launch nothing
template <int MF, int KS, int CK, int TAPG>
static int launch_variant(const ConvArgs& a, hipStream_t st, char* name_out = nullptr, size_t name_cap = 0) {
  constexpr int TW = 16 + KS - 1;
  constexpr size_t lds_k = (size_t)(TW * TW * (CK + 4) + TAPG * MF * (CK + 4)) * sizeof(float);
  constexpr size_t lds_e = (size_t)4 * 64 * (MF + 4) * sizeof(float);  // epilogue transpose
  constexpr size_t lds = lds_k > lds_e ? lds_k : lds_e;
  ConvArgs b = a;
  b.lgx = cdiv(a.W, 16) * cdiv(a.H, 16) * a.B;
  b.lgy = cdiv(a.Cout, MF) * (a.groups > 1 ? a.groups : 1);
  const long total = (long)b.lgx * b.lgy;
  const long G = igemm_grid(KS, CK, lds, total);
  constexpr bool CAN_HEAD = (MF == 32 && KS == 3 && CK == 8 && TAPG == 9);
  const bool head = a.ep.head_out != nullptr;
  if (head && (!CAN_HEAD || a.Cout != 32 || a.groups > 1 || a.ep.pool.p != nullptr)) {
    dg_set_error("dg_conv_igemm: the fused head needs the 8-channel-chunk 3x3 kernel and 32 output channels");
    return DG_ERR_ARG;
  }
  if (name_out) {
    if (head) snprintf(name_out, name_cap, "igemm_conv_head_kernel<%s>", G < total ? "true" : "false");
    else snprintf(name_out, name_cap, "igemm_conv_kernel<%d,%d,%d,%d,%s>", MF, KS, CK, TAPG, G < total ? "true" : "false");
    return DG_OK;
  }
  if constexpr (CAN_HEAD) {
    if (head) {
      static DgOncePerDevice once_h;
      if (once_h.need()) {
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_head_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_head_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      }
      if (G < total) hipLaunchKernelGGL((igemm_conv_head_kernel<true>), dim3((unsigned)G), dim3(256), lds, st, b);
      else hipLaunchKernelGGL((igemm_conv_head_kernel<false>), dim3((unsigned)G), dim3(256), lds, st, b);
      HIPCHECK(hipGetLastError());
      return DG_OK;
    }
  }
  // the dynamic-LDS attribute is per device: a process may hold contexts on several GPUs (Engine(device=...))
  static DgOncePerDevice once;
  if (once.need()) {
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_kernel<MF, KS, CK, TAPG, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_kernel<MF, KS, CK, TAPG, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  if (G < total)
    hipLaunchKernelGGL((igemm_conv_kernel<MF, KS, CK, TAPG, true>), dim3((unsigned)G), dim3(256), lds, st, b);
  else
    hipLaunchKernelGGL((igemm_conv_kernel<MF, KS, CK, TAPG, false>), dim3((unsigned)G), dim3(256), lds, st, b);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

static int dispatch_variant(const ConvPlan& pl, const ConvArgs& a, hipStream_t st, char* name_out, size_t name_cap) {
  switch (pl.variant) {
    case 0: return launch_variant<32, 3, 16, 9>(a, st, name_out, name_cap);
    case 1: return launch_variant<16, 3, 16, 9>(a, st, name_out, name_cap);
    case 2: return launch_variant<32, 5, 8, 25>(a, st, name_out, name_cap);
    case 3: return launch_variant<16, 5, 16, 25>(a, st, name_out, name_cap);
    case 4: return launch_variant<32, 1, 32, 1>(a, st, name_out, name_cap);
    case 5: return launch_variant<16, 1, 16, 1>(a, st, name_out, name_cap);
    case 8: return launch_variant<32, 3, 8, 9>(a, st, name_out, name_cap);
  }
  dg_set_error("dg_conv_igemm: bad variant %d", pl.variant);
  return DG_ERR_ARG;
}

bool dg_conv_igemm_head_supported(const ConvPlan& pl, const ConvArgs& a) {
  return (pl.variant == 8 || pl.variant == 9) && !pl.bf16 && a.Cout == 32 && a.groups <= 1 && a.ep.pool.p == nullptr && a.cpt <= 0 &&
         !a.ep.accumulate;
}

void dg_conv_igemm_name(const ConvPlan& pl, const ConvArgs& a, char* buf, size_t cap) {
  if (cap) buf[0] = 0;
  if (pl.variant < 0) { snprintf(buf, cap, "conv_direct"); return; }
  if (pl.bf16) { snprintf(buf, cap, pl.variant >= 200 ? "igemm_split_kernel" : "igemm_bf16_kernel"); return; }
  if (pl.variant == 9) { snprintf(buf, cap, "%s", dg_conv_wino_name(a)); return; }
  if (dg_conv_igemm_wp_supported(pl, a, false)) { snprintf(buf, cap, "igemm_wp_kernel<0>"); return; }
  dispatch_variant(pl, a, nullptr, buf, cap);
}

static int conv_igemm_impl(const ConvPlan& pl, const ConvArgs& a_in, hipStream_t st, bool allow_wp) {
  ConvArgs a = a_in;
  const bool is_bf16 = pl.bf16 != 0;
  if (pl.variant < 0) {
    dg_set_error("dg_conv_igemm: no MFMA variant for KS=%d Cin=%d Cout=%d", pl.KS, pl.Cin, pl.Cout);
    return DG_ERR_UNSUPPORTED;
  }
  auto misaligned = [](const TView& v) {
    return v.p && ((v.sX % 4) || (v.sY % 4) || (v.sB % 4) || (((uintptr_t)v.p) & 15));
  };
  if (misaligned(a.in) || misaligned(a.out) || misaligned(a.ep.res) || misaligned(a.ep.mask) ||
      misaligned(a.ep.out_pre) || (a.Cout % 4) || (a.ep.film_mul && (a.ep.film_ld % 4))) {
    dg_set_error("dg_conv_igemm: views must be 16-byte aligned (pointers, strides and Cout multiples of 4 floats)");
    return DG_ERR_ARG;
  }
  if (a.ep.pool.p && ((a.H | a.W) & 1 || misaligned(a.ep.pool) || a.groups > 1)) {
    dg_set_error("dg_conv_igemm: fused max-pool needs even H, W and a 16-byte aligned pooled view");
    return DG_ERR_ARG;
  }
  if (a.cpt > 0) {
    const int run = a.cpt * pl.CK;
    bool bad = a.groups > 1 || (a.Cin % run) != 0 || a.Cin / run > 4;
    for (int r = 0; !bad && r < a.Cin / run; ++r) bad = (a.in_run_off[r] % 4) != 0;
    if (bad) {
      dg_set_error("dg_conv_igemm: gathered K needs Cin = runs (<= 4) x cpt x %d channels and 16-byte run offsets", pl.CK);
      return DG_ERR_ARG;
    }
  }
  if (a.ep.head_out && !dg_conv_igemm_head_supported(pl, a)) {
    dg_set_error("dg_conv_igemm: the fused head needs the fp32 8-channel-chunk 3x3 kernel, 32 output channels, no pool");
    return DG_ERR_ARG;
  }
  if (is_bf16) return dg_conv_igemm_bf16(pl, a, st);
  if (pl.variant == 9) return dg_conv_wino(pl, a, st);   // its panel fits no other kernel
  if (allow_wp && dg_conv_igemm_wp_supported(pl, a, false)) return dg_conv_igemm_wp(pl, a, st);
  return dispatch_variant(pl, a, st, nullptr, 0);
}

int dg_conv_igemm(const ConvPlan& pl, const ConvArgs& a, hipStream_t st) { return conv_igemm_impl(pl, a, st, true); }
// the workgroup-tile kernels only (unit tests: the reference the wave-private kernel must match bit for bit)
int dg_conv_igemm_tile(const ConvPlan& pl, const ConvArgs& a, hipStream_t st) { return conv_igemm_impl(pl, a, st, false); }

// ---------------------------------------------------------------------------
// weight packing:  dst[nt][cc][tap][n][k]
// ---------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ src, float* __restrict__ dst, int ntaps, int srcI,
                                    int srcO, int io, int transpose, int flip, const float* __restrict__ kscale,
                                    int NT, int CK, int nCC, int Kdim, int Ndim, size_t total, int bf16, int tapg) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t q = i;
    const int k = (int)(q % CK);
    q /= CK;
    const int n = (int)(q % NT);
    q /= NT;
    const int tap = (int)(q % ntaps);
    q /= ntaps;
    const int cc = (int)(q % nCC);
    const int nt = (int)(q / nCC);
    const int kk = cc * CK + k, nn = nt * NT + n;
    float v = 0.f;
    if (kk < Kdim && nn < Ndim) {
      const int ci = transpose ? nn : kk;  // source I-axis index
      const int co = transpose ? kk : nn;  // source O-axis index
      const int ts = flip ? (ntaps - 1 - tap) : tap;
      const size_t off = (size_t)ts * srcI * srcO + (io ? ((size_t)co * srcI + ci) : ((size_t)ci * srcO + co));
      v = src[off];
      if (kscale) v *= kscale[kk];
    }
    if (bf16 >= 2) store_split(dst, (size_t)nt * nCC * ntaps * NT * CK * bf16, cc, tap, n, k, v, bf16, tapg, ntaps, NT, CK);
    else store_packed(dst, i, v, bf16);
  }
}

// Batched form: every packed panel of a network in ONE launch (a refresh after an Adam step used to be ~130 launches
// of a few microseconds each for the generator).  Block -> job by binary search over the jobs' first blocks.
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const PackJob* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].blk0 <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackJob J = jobs[lo];
  const unsigned b = blockIdx.x - J.blk0;
  if (J.wino) {
    // Winograd panel [nt][cc][16 frequencies][n][k]: one thread per (nt, cc, n, k) reads the nine taps once and writes
    // all sixteen U = G g G^T elements (G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; in double, each rounded once).
    // g is what the direct kernel multiplies with: the fp32 source element (x kscale, rounded to fp32 first).
    const unsigned cols = J.total / 16u, plane = (unsigned)(J.NT * J.CK);
    for (unsigned i = b * 256u + threadIdx.x; i < cols; i += J.nblk * 256u) {
      unsigned q = i;
      const int k = (int)(q % J.CK);
      q /= J.CK;
      const int n = (int)(q % J.NT);
      q /= J.NT;
      const int cc = (int)(q % J.nCC);
      const int nt = (int)(q / J.nCC);
      const int kk = cc * J.CK + k, nn = nt * J.NT + n;
      double g[3][3];
#pragma unroll
      for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = 0.0;
      if (kk < J.Kdim && nn < J.Ndim) {
        const int ci = J.transpose ? nn : kk;
        const int co = J.transpose ? kk : nn;
        const float* e0 = J.src + (J.io ? ((size_t)co * J.srcI + ci) : ((size_t)ci * J.srcO + co));
        const size_t ts_stride = (size_t)J.srcI * J.srcO;
        const float ks = J.kscale ? J.kscale[kk] : 1.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          float v = e0[(size_t)(J.flip ? 8 - t : t) * ts_stride];
          if (J.kscale) v *= ks;
          g[t / 3][t % 3] = (double)v;
        }
      }
      double tg[4][3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        tg[0][c] = g[0][c];
        tg[1][c] = 0.5 * (g[0][c] + g[1][c] + g[2][c]);
        tg[2][c] = 0.5 * (g[0][c] - g[1][c] + g[2][c]);
        tg[3][c] = g[2][c];
      }
      float* d = J.dst + (size_t)nt * J.nt_stride + ((size_t)cc * 16u) * plane + (size_t)n * J.CK + k;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        d[(4 * a + 0) * plane] = (float)tg[a][0];
        d[(4 * a + 1) * plane] = (float)(0.5 * (tg[a][0] + tg[a][1] + tg[a][2]));
        d[(4 * a + 2) * plane] = (float)(0.5 * (tg[a][0] - tg[a][1] + tg[a][2]));
        d[(4 * a + 3) * plane] = (float)tg[a][2];
      }
    }
    return;
  }
  for (unsigned i = b * 256u + threadIdx.x; i < J.total; i += J.nblk * 256u) {
    unsigned q = i;
    const int k = (int)(q % J.CK);
    q /= J.CK;
    const int n = (int)(q % J.NT);
    q /= J.NT;
    const int tap = (int)(q % J.ntaps);
    q /= J.ntaps;
    const int cc = (int)(q % J.nCC);
    const int nt = (int)(q / J.nCC);
    const int kk = cc * J.CK + k, nn = nt * J.NT + n;
    float v = 0.f;
    if (kk < J.Kdim && nn < J.Ndim) {
      const int ci = J.transpose ? nn : kk;
      const int co = J.transpose ? kk : nn;
      const size_t eo = J.io ? ((size_t)co * J.srcI + ci) : ((size_t)ci * J.srcO + co);
      const size_t ts_stride = (size_t)J.srcI * J.srcO;
      const int ts = J.flip ? (J.ntaps - 1 - tap) : tap;
      v = J.src[(size_t)ts * ts_stride + eo];
      if (J.kscale) v *= J.kscale[kk];
    }
    if (J.bf16 >= 2)
      store_split(J.dst, (size_t)nt * J.nt_stride * J.bf16, cc, tap, n, k, v, J.bf16, J.tapg, J.ntaps, J.NT, J.CK);
    else
      store_packed(J.dst, (size_t)nt * J.nt_stride + (i - (unsigned)nt * J.per_nt), v, J.bf16);
  }
}

int dg_pack_job(const ConvPlan& pl, const float* src, int srcI, int srcO, int io, int transpose, int flip,
                const float* kscale, float* dst, size_t nt_stride, PackJob* job) {
  const int Kdim = transpose ? srcO : srcI, Ndim = transpose ? srcI : srcO;
  if (Kdim != pl.Cin || Ndim != pl.Cout || pl.variant < 0 || pl.packedFloats >= (1ull << 30)) {
    dg_set_error("dg_pack_job: plan (%d->%d) does not match source roles (%d->%d)", pl.Cin, pl.Cout, Kdim, Ndim);
    return DG_ERR_ARG;
  }
  job->src = src; job->dst = dst; job->kscale = kscale;
  job->wino = (pl.variant == 9) ? 1 : 0;
  job->ntaps = job->wino ? 16 : pl.KS * pl.KS; job->srcI = srcI; job->srcO = srcO; job->io = io; job->transpose = transpose;
  job->flip = flip; job->NT = pl.NT; job->CK = pl.CK; job->nCC = pl.nCC; job->Kdim = Kdim; job->Ndim = Ndim;
  job->bf16 = pl.bf16;
  job->tapg = (pl.variant >= 200) ? (pl.KS == 5 ? 5 : pl.KS * pl.KS) : 0;
  job->total = (unsigned)((size_t)pl.nNT * pl.nCC * job->ntaps * pl.NT * pl.CK);
  job->per_nt = (unsigned)((size_t)pl.nCC * job->ntaps * pl.NT * pl.CK);
  job->nt_stride = nt_stride ? (unsigned)nt_stride : job->per_nt;
  job->blk0 = job->nblk = 0;
  return DG_OK;
}

unsigned dg_pack_layout(PackJob* jobs, int njobs) {
  unsigned b = 0;
  for (int i = 0; i < njobs; ++i) {
    unsigned n = ((jobs[i].wino ? jobs[i].total / 16u : jobs[i].total) + 255u) / 256u;
    if (n > 32u) n = 32u;            // grid-stride inside a job: ~1000 blocks for the whole generator
    if (n < 1u) n = 1u;
    jobs[i].blk0 = b;
    jobs[i].nblk = n;
    b += n;
  }
  return b;
}

int dg_pack_weights_batch(const PackJob* jobs_dev, int njobs, unsigned nblocks, hipStream_t st) {
  if (njobs <= 0) return DG_OK;
  hipLaunchKernelGGL(pack_weights_batch_kernel, dim3(nblocks), dim3(256), 0, st, jobs_dev, njobs);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

int dg_pack_weights(const ConvPlan& pl, const float* src, int srcI, int srcO, int io, int transpose, int flip,
                    const float* kscale, float* dst, hipStream_t st) {
  const int Kdim = transpose ? srcO : srcI, Ndim = transpose ? srcI : srcO;
  if (Kdim != pl.Cin || Ndim != pl.Cout) {
    dg_set_error("dg_pack_weights: plan (%d->%d) does not match source roles (%d->%d)", pl.Cin, pl.Cout, Kdim, Ndim);
    return DG_ERR_ARG;
  }
  if (pl.variant == 9) {   // Winograd panels exist in the batched kernel only: one job
    PackJob job;
    DGCHECK(dg_pack_job(pl, src, srcI, srcO, io, transpose, flip, kscale, dst, 0, &job));
    const unsigned nb = dg_pack_layout(&job, 1);
    PackJob* jd = nullptr;
    HIPCHECK(hipMalloc((void**)&jd, sizeof(PackJob)));
    hipError_t e = hipMemcpyAsync(jd, &job, sizeof(PackJob), hipMemcpyHostToDevice, st);
    int rc = (e == hipSuccess) ? dg_pack_weights_batch(jd, 1, nb, st) : DG_ERR_HIP;
    hipStreamSynchronize(st);   // `job` lives on this stack frame
    hipFree(jd);
    return rc;
  }
  const size_t total = (size_t)pl.nNT * pl.nCC * pl.KS * pl.KS * pl.NT * pl.CK;
  const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, st, src, dst, pl.KS * pl.KS, srcI, srcO, io,
                     transpose, flip, kscale, pl.NT, pl.CK, pl.nCC, Kdim, Ndim, total, pl.bf16,
                     (pl.variant >= 200) ? (pl.KS == 5 ? 5 : pl.KS * pl.KS) : 0);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}
