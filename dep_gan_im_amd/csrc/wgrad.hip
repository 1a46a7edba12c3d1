// Weight-gradient contraction on the fp32 matrix cores.
//
//   dW[tap][ci][co] = sum_{b,y,x} X[b, y+ty-p, x+tx-p, ci] * D[b, y, x, co]
//
// This is what Keras' Adam.get_updates differentiates for every Conv2D kernel
// (GT:549, 568, 594), and with X = u_{l-1}, D = M_l.g_l it is also the weight
// gradient of the gradient penalty.
//
// GEMM view: M = ci (MF rows), N = co (MF cols), K = pixels.  One workgroup owns
// one (ci-tile, co-tile, tap-group) and walks a contiguous range of TH x 16
// pixel tiles; its 4 waves split each tile by rows (split-K inside the
// workgroup), keep one accumulator tile per tap in registers for the whole
// range, are summed through LDS at the end and written as ONE partial slab.
// Slabs are reduced by a separate deterministic pass (no float atomics, so a
// step is bit-reproducible run to run).
// A = X^T: lane (r, h) reads channel r of pixel k_h, B: channel r of the same
// pixel of D -> both are conflict-free 32-bit LDS reads of a [pixel][MF] image.
#include <stdlib.h>

#include "common.h"

template <int MF>
struct MfmaW;
template <>
struct MfmaW<32> {
  typedef f32x16 acc_t;
  static constexpr int NREG = 16;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int j, int h) { return (j & 3) + 8 * (j >> 2) + 4 * h; }
};
template <>
struct MfmaW<16> {
  typedef f32x4 acc_t;
  static constexpr int NREG = 4;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int j, int h) { return 4 * h + j; }
};

// ---------------------------------------------------------------------------
// Staging: LDS-DMA (buffer_load_dwordx4 ... lds), double-buffered, one workgroup per CU.
//
// Nine (or 25) accumulator tiles leave no registers to hold a tile in flight, and with register staging the
// load and MFMA phases of co-resident workgroups ran in lockstep, so the matrix pipe idled during every staging phase
// (85 TFLOP/s).  Here the next tile is copied global -> LDS by the DMA path (no VGPR destination, no ds_write pass)
// into the second of two LDS buffers while the MFMAs of the current tile run: two raw barriers and one counted vmcnt
// per tile, accumulators in AGPRs, no spills (117-121 TFLOP/s on the 3x3 layers inside the step).
// The LDS image is the [pixel][MF] one of the header, which is lane-linear per wave-instruction (64 lanes x 16 B =
// 64*16/(4 MF) pixels) as the DMA requires; out-of-image pixels and channel tails carry an out-of-range buffer offset
// instead (the per-lane SOURCE address is free and the hardware returns zeros for it), so no lane is ever masked and
// every wave issues the same number of pieces.
// ---------------------------------------------------------------------------
template <int MF, int KS, int TPW, int TH>
struct WDmaCfg {
  static constexpr int TW = 16 + KS - 1, THH = TH + KS - 1, PIXT = THH * TW, V = MF / 4;
  static constexpr int XTOT = PIXT * V, DTOT = TH * 16 * V;          // 16-byte pieces per tile
  static constexpr int NXP = (XTOT + 255) / 256, NDP = DTOT / 256;   // DMA instructions per wave per tile
  static constexpr int XBUF = NXP * 256 * 4, DBUF = DTOT * 4, BUF = XBUF + DBUF;   // floats
  static constexpr size_t LDS_TILES = (size_t)2 * BUF * sizeof(float);
  // 8-row tiles of the 3x3 / 32-channel form: two sets of tile buffers (80 KB) fit a CU twice, so TWO workgroups share
  // it -- a second wave per SIMD multiplies while the first one waits at a barrier or for its DMA.  Its final sum of the
  // 4 waves then goes through LDS three taps at a time (48 KB) instead of all nine (144 KB); every other form keeps one
  // round and its launch bound.
  static constexpr bool TWO_PER_CU = (MF == 32 && KS == 3 && TH == 8) || (MF == 32 && KS == 5 && TH == 4);
  static constexpr int RT = TWO_PER_CU ? (KS == 3 ? 3 : 1) : TPW;
  static constexpr size_t LDS_RED = (size_t)RT * 4 * MF * MF * sizeof(float);
  static constexpr size_t LDS_BYTES = LDS_TILES > LDS_RED ? LDS_TILES : LDS_RED;
  static constexpr int WGS_PER_CU = TWO_PER_CU ? 2 : 1;
  static_assert(DTOT % 256 == 0, "dy tile must be whole wave-instructions");
};

template <int MF, int KS, int TPW, int TH>
__global__ __launch_bounds__(256, (WDmaCfg<MF, KS, TPW, TH>::WGS_PER_CU)) void wgrad_dma_kernel(const WgradArgs a) {
  typedef WDmaCfg<MF, KS, TPW, TH> C;
  constexpr int PAD = KS / 2, TW = C::TW, NTAPS = KS * KS, NGT = NTAPS / TPW;
  constexpr int KM = 64 / MF, PW = TH * 4, KSTEPS = PW / KM, V = C::V;
  constexpr int XTOT = C::XTOT, NXP = C::NXP, NDP = C::NDP, XBUF = C::XBUF, BUF = C::BUF;
  constexpr int NPIECE = NXP + NDP;
  constexpr int NREAD = TPW + 1;                       // LDS reads per k-step
  constexpr bool COUNTED = NREAD <= 15;                // lgkmcnt is a 4-bit counter
  static_assert(NTAPS % TPW == 0 && (TPW == NTAPS || TPW == KS), "tap grouping");
  static_assert(NPIECE <= 63, "vmcnt is a 6-bit counter");
  typedef typename MfmaW<MF>::acc_t acc_t;

  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][BUF]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane % MF, h = lane / MF;
  const int nCoT = (a.Cout + MF - 1) / MF;
  // Workgroup -> (pixel-tile chunk, channel-tile pair).  All channel-tile pairs of one chunk read the same x / dy
  // tiles; workgroup ids are dealt round-robin over the 8 XCDs (own L2 each), so the pairs of a chunk are given to one
  // XCD at consecutive dispatch slots: id = 8 s + x  ->  pair s % gridDim.y of chunk 8 (s / gridDim.y) + x.
  int y, chunk;
  {
    const unsigned nY = gridDim.y, nX = gridDim.x;
    const unsigned id = blockIdx.x + blockIdx.y * nX;
    if ((nX & 7u) == 0 && nY > 1) {
      const unsigned x = id & 7u, sl = id >> 3;
      y = (int)(sl % nY);
      chunk = (int)(8u * (sl / nY) + x);
    } else {
      y = (int)blockIdx.y;
      chunk = (int)blockIdx.x;
    }
  }
  const int tg = y % NGT;
  y /= NGT;
  const int co0 = (y % nCoT) * MF;
  const int ci0 = (y / nCoT) * MF;
  const int tilesX = (a.W + 15) >> 4, tilesY = (a.H + TH - 1) / TH;
  const int t0 = chunk * a.tilesPerChunk;
  const int t1 = min(t0 + a.tilesPerChunk, a.nTiles);

  acc_t acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int j = 0; j < MfmaW<MF>::NREG; ++j) acc[t][j] = 0.f;

  // ---- tile staging: buffer-addressed LDS-DMA ----
  // One descriptor per operand; a piece's address is descriptor base + per-lane byte offset (fixed for the whole
  // kernel) + a scalar tile offset, so an interior tile costs no vector ALU work at all.  The x descriptor's base is
  // moved back by the halo, which keeps the per-lane offsets non-negative; lanes that must read zeros (halo outside
  // the image, channels past Cin / Cout, padding pieces) carry an offset beyond num_records, for which the hardware
  // returns 0.
  const float* const xorg = a.x.p + ci0 - PAD * ((long)a.x.sY + (long)a.x.sX);
  const float* const dorg = a.dy.p + co0;
  constexpr int SENT = (int)0x80000000;
  int xvo[NXP], xyx[NXP];
#pragma unroll
  for (int i = 0; i < NXP; ++i) {
    const int q = tid + i * 256;
    const int pix = q / V, part = q & (V - 1);
    const int ly = pix / TW, lx = pix - ly * TW;
    const bool ok = q < XTOT && (ci0 + part * 4) < a.Cin;      // q >= XTOT: padding piece behind the halo tile
    xvo[i] = ok ? 4 * (ly * (int)a.x.sY + lx * (int)a.x.sX + part * 4) : SENT;
    xyx[i] = (ly << 8) | lx;
  }
  constexpr int DROWS = 256 / V / 16;                           // dy rows covered by one wave-instruction round
  const int dly = (tid / V) >> 4, dlx = (tid / V) & 15;
  const int dvo = ((co0 + (tid & (V - 1)) * 4) < a.Cout)
                      ? 4 * (dly * (int)a.dy.sY + dlx * (int)a.dy.sX + (tid & (V - 1)) * 4)
                      : SENT;
  const int dstep = 4 * DROWS * (int)a.dy.sY;
  const int wbase = __builtin_amdgcn_readfirstlane(wv * 256);   // this wave's float offset inside a 256-piece round

  // Column sums of dy ride along where asked for: the B fragment of a k-step IS dy[pixel][co], every pixel of a tile
  // is some lane's fragment exactly once per tap group, so summing the fragments per lane (2 FMAs per pair of
  // k-steps next to 18 MFMAs) and folding lanes and waves at the end gives the tile's column sums.  Only the
  // workgroups of input-channel tile 0 / tap group 0 carry a non-zero multiplier.
  const bool do_cs = a.colpart != nullptr && ci0 == 0 && tg == 0;
  float cs0 = 0.f, cs1 = 0.f;
  int bq[2] = {0, 0};          // sample index of the tile held by each LDS buffer
  // tile coordinates of the next tile to stage, advanced incrementally (no divisions inside the loop)
  int ntx, nty, nb;
  {
    int t = t0;
    ntx = t % tilesX;
    t /= tilesX;
    nty = t % tilesY;
    nb = t / tilesY;
  }

  const int fragB = ((wv * (TH / 4)) * 16 + h) * MF + r;   // float offsets inside a buffer's D / X images
  const int fragX = ((wv * (TH / 4) + (TPW == NTAPS ? 0 : tg)) * TW + h) * MF + r;

  // The loop starts one tile early: that first pass only stages tile t0 (into buffer 0) and computes nothing, so
  // the staging code exists once.
  for (int tile = t0 - 1; tile < t1; ++tile) {
    const int buf = (tile - t0) & 1;
    // every wave has finished the MFMA loop of tile-1 (its LDS reads were consumed by those MFMAs), so the other
    // buffer may be overwritten
    __builtin_amdgcn_s_barrier();
    if (tile + 1 < t1) {
      const int tx0 = ntx * 16, ty0 = nty * TH;
      const int xso = 4 * (nb * (int)a.x.sB + ty0 * (int)a.x.sY + tx0 * (int)a.x.sX);
      const int dso = 4 * (nb * (int)a.dy.sB + ty0 * (int)a.dy.sY + tx0 * (int)a.dy.sX);
      const bool interior = ty0 >= PAD && ty0 + TH + PAD <= a.H && tx0 >= PAD && tx0 + 16 + PAD <= a.W;
      float* xs = smem + (buf ^ 1) * BUF + wbase;
      float* ds = xs + XBUF;
      bq[buf ^ 1] = nb;
      auto mkrsrc = [](const float* p) {
        const unsigned long long u = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7FFFFFFF,
                                                 0x00020000);
      };
      const __amdgpu_buffer_rsrc_t rx = mkrsrc(xorg), rd = mkrsrc(dorg);
      if (interior) {
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
          // (a plain int local: with a type-dependent argument such as xvo[i] hipcc drops the host-side
          // instantiation of the kernel without a diagnostic)
          const int vo = xvo[i];
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(xs + i * 1024), 16,
                                                   vo, xso, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NDP; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (__attribute__((address_space(3))) void*)(ds + i * 1024), 16,
                                                   dvo, dso + i * dstep, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
          const int iy = ty0 + (xyx[i] >> 8) - PAD, ix = tx0 + (xyx[i] & 255) - PAD;
          const int vo = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? xvo[i] : SENT;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(xs + i * 1024), 16,
                                                   vo, xso, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NDP; ++i) {
          const int vo = ((ty0 + dly + i * DROWS) < a.H && (tx0 + dlx) < a.W) ? dvo : SENT;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (__attribute__((address_space(3))) void*)(ds + i * 1024), 16,
                                                   vo, dso + i * dstep, 0, 0);
        }
      }
      if (++ntx == tilesX) {
        ntx = 0;
        if (++nty == tilesY) {
          nty = 0;
          ++nb;
        }
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");   // this wave's pieces of `tile` have landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                      // ... and so have everybody else's
    if (tile < t0) continue;
    const float csmul = (do_cs && bq[buf] < a.colB) ? 1.0f : 0.0f;
    if (COUNTED) {
      // Fragment reads as inline asm with counted waits: for compiler-visible ds_reads hipcc puts
      // s_waitcnt lgkmcnt(0) in front of every other MFMA group, which also waits for the reads of the NEXT k-step
      // issued just before, and with one wave per SIMD nothing hides that LDS latency.  LDS reads retire in order,
      // so leaving the NREAD newest outstanding means the previous k-step's have arrived; sched_barriers pin
      // read -> wait -> MFMA.
      // Two consecutive k-steps of one lane are 256 bytes apart in both images (2 pixels of 32 channels, or 4 of
      // 16), which is the unit of ds_read2st64_b32: one LDS instruction fetches the fragment values of a PAIR of
      // k-steps.  Its offsets count 256-byte units, so the sub-256 remainder of a tap's offset is carried by one of
      // four pre-offset base registers.
      const unsigned bbase =
          (unsigned)(size_t)(__attribute__((address_space(3))) float*)(smem + buf * BUF + XBUF + fragB);
      const unsigned xbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(smem + buf * BUF + fragX);
      const unsigned xb[4] = {xbase, xbase + 64u, xbase + 128u, xbase + 192u};
      static_assert(KSTEPS % 2 == 0 && (16 / KM) % 2 == 0, "k-steps are paired inside a pixel row");
      auto load_pair = [&](int kp, f32x2* av, f32x2& bv) {
        constexpr int SPR = 16 / KM;
        const int kk = 2 * kp;
        const int pyo = kk / SPR, pxo = (kk % SPR) * KM;
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3"
                     : "=v"(bv)
                     : "v"(bbase), "n"((4 * (pyo * 16 + pxo) * MF) / 256), "n"((4 * (pyo * 16 + pxo) * MF) / 256 + 1));
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
          const int ty = (TPW == NTAPS) ? (tl / KS) : 0;   // row offset already in fragX for row groups
          const int tx = (TPW == NTAPS) ? (tl % KS) : tl;
          const int off = 4 * ((pyo + ty) * TW + pxo + tx) * MF;
          asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3"
                       : "=v"(av[tl])
                       : "v"(xb[(off % 256) / 64]), "n"(off / 256), "n"(off / 256 + 1));
        }
      };
      f32x2 afr[2][TPW], bfr[2];
      load_pair(0, afr[0], bfr[0]);
#pragma unroll
      for (int kp = 0; kp < KSTEPS / 2; ++kp) {
        if (kp + 1 < KSTEPS / 2) load_pair(kp + 1, afr[(kp + 1) & 1], bfr[(kp + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (kp + 1 < KSTEPS / 2) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(COUNTED ? NREAD : 0) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) acc[tl] = MfmaW<MF>::run(afr[kp & 1][tl].x, bfr[kp & 1].x, acc[tl]);
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) acc[tl] = MfmaW<MF>::run(afr[kp & 1][tl].y, bfr[kp & 1].y, acc[tl]);
        cs0 = fmaf(bfr[kp & 1].x, csmul, cs0);
        cs1 = fmaf(bfr[kp & 1].y, csmul, cs1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      float afr[2][TPW], bfr[2];
      const float* bbase = smem + buf * BUF + XBUF + fragB;
      const float* xbase = smem + buf * BUF + fragX;
      auto load_frag = [&](int kk, float* av, float& bv) {
        constexpr int SPR = 16 / KM;
        const int pyo = kk / SPR, pxo = (kk % SPR) * KM;
        bv = bbase[(pyo * 16 + pxo) * MF];
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
          const int ty = (TPW == NTAPS) ? (tl / KS) : 0;
          const int tx = (TPW == NTAPS) ? (tl % KS) : tl;
          av[tl] = xbase[((pyo + ty) * TW + pxo + tx) * MF];
        }
      };
      load_frag(0, afr[0], bfr[0]);
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        if (kk + 1 < KSTEPS) load_frag(kk + 1, afr[(kk + 1) & 1], bfr[(kk + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) acc[tl] = MfmaW<MF>::run(afr[kk & 1][tl], bfr[kk & 1], acc[tl]);
        cs0 = fmaf(bfr[kk & 1], csmul, cs0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- sum the 4 waves through LDS and write one slab per workgroup, RT taps per round (two barriers each) ----
  constexpr int RT = C::RT;
  float* red = smem;  // [RT][4][MF*MF]
  const size_t slab = (size_t)NTAPS * a.Cin * a.Cout;
  float* pout = a.part + (size_t)chunk * slab;
#pragma unroll
  for (int t0 = 0; t0 < TPW; t0 += RT) {
    __syncthreads();    // every wave is done with the tile buffers / with the previous round
#pragma unroll
    for (int tl = 0; tl < RT; ++tl)
#pragma unroll
      for (int j = 0; j < MfmaW<MF>::NREG; ++j)
        red[(tl * 4 + wv) * MF * MF + MfmaW<MF>::row(j, h) * MF + r] = acc[t0 + tl][j];
    __syncthreads();
    for (int q = tid; q < RT * MF * MF; q += 256) {
      const int tl = q / (MF * MF), e = q % (MF * MF);
      const float* rt = red + tl * 4 * MF * MF;
      const float sum = (rt[e] + rt[MF * MF + e]) + (rt[2 * MF * MF + e] + rt[3 * MF * MF + e]);
      const int tap = tg * TPW + t0 + tl;
      const int ci = ci0 + e / MF, co = co0 + e % MF;
      if (ci < a.Cin && co < a.Cout) pout[((size_t)tap * a.Cin + ci) * a.Cout + co] = sum;
    }
  }
  if (do_cs) {
    // fold the 64 / MF pixel-parity lanes of each channel and the 4 waves, in a fixed order
    __syncthreads();
    red[tid] = cs0 + cs1;
    __syncthreads();
    if (tid < MF) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int hh = 0; hh < 64 / MF; ++hh) s += red[w * 64 + hh * MF + tid];
      if (co0 + tid < a.Cout) a.colpart[(size_t)chunk * a.Cout + co0 + tid] = s;
    }
  }
}

struct WVar {
  int MF, KS, TPW, TH;
  size_t lds;
};

// 3x3 / 32-channel form on 8-row tiles, two workgroups per CU (DEPGAN_WGRAD_TH8=0: 16-row tiles, one per CU)
static bool wgrad_th8() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("DEPGAN_WGRAD_TH8");
    on = (e && atoi(e) == 0) ? 0 : 1;
  }
  return on != 0;
}

static void pick_variant(int KS, int Cin, int Cout, WVar* v) {
  v->KS = KS;
  v->MF = (Cin % 32 == 0 && Cout % 32 == 0) ? 32 : 16;
  v->TH = 16;
  v->TPW = KS * KS;
  if (KS == 5 && v->MF == 32) {
    v->TPW = 5;
    v->TH = wgrad_th8() ? 4 : 8;
  }
  if (KS == 3 && v->MF == 32 && wgrad_th8()) v->TH = 8;
  v->lds = 0;
  if (KS == 3 && v->MF == 32) v->lds = v->TH == 8 ? WDmaCfg<32, 3, 9, 8>::LDS_BYTES : WDmaCfg<32, 3, 9, 16>::LDS_BYTES;
  if (KS == 3 && v->MF == 16) v->lds = WDmaCfg<16, 3, 9, 16>::LDS_BYTES;
  if (KS == 5 && v->MF == 32) v->lds = v->TH == 4 ? WDmaCfg<32, 5, 5, 4>::LDS_BYTES : WDmaCfg<32, 5, 5, 8>::LDS_BYTES;
  if (KS == 5 && v->MF == 16) v->lds = WDmaCfg<16, 5, 25, 16>::LDS_BYTES;
  if (KS == 1 && v->MF == 32) v->lds = WDmaCfg<32, 1, 1, 16>::LDS_BYTES;
  if (KS == 1 && v->MF == 16) v->lds = WDmaCfg<16, 1, 1, 16>::LDS_BYTES;
}

static void chunking(const WVar& v, int B, int H, int W, int Cin, int Cout, int* nTiles, int* tilesPerChunk,
                     int* nchunks, int* gridY) {
  const int tilesX = cdiv(W, 16), tilesY = cdiv(H, v.TH);
  *nTiles = B * tilesX * tilesY;
  *gridY = cdiv(Cin, v.MF) * cdiv(Cout, v.MF) * (v.KS * v.KS / v.TPW);
  // ONE round of workgroups: as many as are resident at once (one per CU, two where two sets of tile buffers fit the
  // CU's LDS).  A workgroup's prologue (cold first tile) and epilogue (slab) are not overlapped with anything on its
  // CU, and every further round costs them again (measured: 1024 / 512 / 256 workgroups -> 386 / 372 / 358 us).
  const int per_cu = (v.lds && 2 * v.lds <= (size_t)160 * 1024) ? 2 : 1;
  int want = 256 * per_cu / *gridY;
  if (want < 1) want = 1;
  if (want > *nTiles) want = *nTiles;
  *tilesPerChunk = cdiv(*nTiles, want);
  *nchunks = cdiv(*nTiles, *tilesPerChunk);
}

size_t dg_wgrad_part_floats(int KS, int B, int H, int W, int Cin, int Cout) {
  WVar v;
  pick_variant(KS, Cin, Cout, &v);
  int nTiles, tpc, nch, gy;
  chunking(v, B, H, W, Cin, Cout, &nTiles, &tpc, &nch, &gy);
  const size_t slab = (size_t)KS * KS * Cin * Cout;
  return (size_t)nch * slab;
}

template <int MF, int KS, int TPW, int TH>
static int launch_wgrad_dma(WgradArgs a, int nchunks, int gridY, hipStream_t st) {
  constexpr size_t lds = WDmaCfg<MF, KS, TPW, TH>::LDS_BYTES;
  static_assert(lds <= 160 * 1024, "two tile buffers must fit the CU's LDS");
  static DgOncePerDevice once;
  if (once.need()) {
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_dma_kernel<MF, KS, TPW, TH>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL((wgrad_dma_kernel<MF, KS, TPW, TH>), dim3(nchunks, gridY), dim3(256), lds, st, a);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

int dg_wgrad(int KS, const WgradArgs& a_in, int* nchunks_out, hipStream_t st) {
  WgradArgs a = a_in;
  if ((a.Cin % 4) || (a.Cout % 4) || (a.x.sX % 4) || (a.x.sY % 4) || (a.x.sB % 4) || (a.dy.sX % 4) ||
      (a.dy.sY % 4) || (a.dy.sB % 4) || (((uintptr_t)a.x.p) & 15) || (((uintptr_t)a.dy.p) & 15)) {
    dg_set_error("dg_wgrad: channels and strides must be multiples of 4 floats (Cin=%d Cout=%d)", a.Cin, a.Cout);
    return DG_ERR_ARG;
  }
  // the staging DMA addresses both operands with 32-bit byte offsets from one buffer descriptor each
  const long long lim = 0x7FFFFFFFll - (1ll << 20);
  if ((long long)a.B * a.x.sB * 4 + ((long long)a.x.sY + a.x.sX) * 4 * KS >= lim || (long long)a.B * a.dy.sB * 4 >= lim) {
    dg_set_error("dg_wgrad: operand spans 2 GiB or more (B=%d, batch strides %ld / %ld floats)", a.B, (long)a.x.sB,
                 (long)a.dy.sB);
    return DG_ERR_UNSUPPORTED;
  }
  WVar v;
  pick_variant(KS, a.Cin, a.Cout, &v);
  int nTiles, tpc, nch, gy;
  chunking(v, a.B, a.H, a.W, a.Cin, a.Cout, &nTiles, &tpc, &nch, &gy);
  a.nTiles = nTiles;
  a.tilesPerChunk = tpc;
  *nchunks_out = nch;
  if (KS == 3 && v.MF == 32 && v.TH == 8) return launch_wgrad_dma<32, 3, 9, 8>(a, nch, gy, st);
  if (KS == 3 && v.MF == 32) return launch_wgrad_dma<32, 3, 9, 16>(a, nch, gy, st);
  if (KS == 3 && v.MF == 16) return launch_wgrad_dma<16, 3, 9, 16>(a, nch, gy, st);
  if (KS == 5 && v.MF == 32 && v.TH == 4) return launch_wgrad_dma<32, 5, 5, 4>(a, nch, gy, st);
  if (KS == 5 && v.MF == 32) return launch_wgrad_dma<32, 5, 5, 8>(a, nch, gy, st);
  if (KS == 5 && v.MF == 16) return launch_wgrad_dma<16, 5, 25, 16>(a, nch, gy, st);
  if (KS == 1 && v.MF == 32) return launch_wgrad_dma<32, 1, 1, 16>(a, nch, gy, st);
  if (KS == 1 && v.MF == 16) return launch_wgrad_dma<16, 1, 1, 16>(a, nch, gy, st);
  dg_set_error("dg_wgrad: unsupported kernel size %d", KS);
  return DG_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------
// deterministic slab reduction
// ---------------------------------------------------------------------------
// One launch finishes a weight gradient: blocks [0, nbr) reduce the slabs -- a block is 64 output elements x 4 slices of
// the chunk range, every thread sums its slice front to back (4 independent accumulators), the 4 slice sums are added in
// a fixed order -- and, when the launch carried column sums, blocks [nbr, nbr + Cout) reduce the [nchunks][Cout]
// partial rows of one channel each.  No float atomics anywhere: a step is bit-reproducible run to run.
struct ColFin {
  const float* part;     // [nb][C] partial rows (null: none)
  const float* scale;
  float *out, *raw;
  int nb, C;
};
// SL: slices of the slab axis summed side by side (64 SL threads per workgroup).  A small weight tensor has few
// 64-element groups (144 for 32 -> 32 3x3) against hundreds of slabs: with 4 slices every thread walks 128 slabs one
// load after the other and the launch is latency (11 us for 19 MB); 16 slices put four times the loads in flight.
template <int SL>
__global__ __launch_bounds__(64 * SL) void slab_reduce_kernel(const float* __restrict__ in, int nin, int ntaps, int Cin,
                                                              int Cout, const float* __restrict__ scale,
                                                              float* __restrict__ out, float* __restrict__ raw,
                                                              int accumulate, int oi, unsigned nbr, ColFin cf) {
  __shared__ float sh[64 * SL];
  if (blockIdx.x >= nbr) {
    // column-sum finish: one channel per block
    const int c = (int)(blockIdx.x - nbr);
    float s = 0.f;
    for (int b = threadIdx.x; b < cf.nb; b += 64 * SL) s += cf.part[(size_t)b * cf.C + c];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 32 * SL; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const float t = sh[0];
      if (cf.raw) cf.raw[c] = t;
      if (cf.out) cf.out[c] = cf.scale ? t * cf.scale[c] : t;
    }
    return;
  }
  const size_t n = (size_t)ntaps * Cin * Cout;
  const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * 64 + el;
  const int per = (nin + SL - 1) / SL;
  const int c0 = min(sl * per, nin), c1 = min(c0 + per, nin);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int c = c0;
    for (; c + 3 < c1; c += 4) {
      s0 += in[(size_t)c * n + i];
      s1 += in[(size_t)(c + 1) * n + i];
      s2 += in[(size_t)(c + 2) * n + i];
      s3 += in[(size_t)(c + 3) * n + i];
    }
    for (; c < c1; ++c) s0 += in[(size_t)c * n + i];
  }
  sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && i < n) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < SL; k += 4) s += (sh[64 * k + el] + sh[64 * (k + 1) + el]) + (sh[64 * (k + 2) + el] + sh[64 * (k + 3) + el]);
    const int co = (int)(i % Cout);
    const int ci = (int)((i / Cout) % Cin);
    const int tap = (int)(i / ((size_t)Cout * Cin));
    const size_t o = oi ? (((size_t)tap * Cout + co) * Cin + ci) : i;
    if (raw) raw[o] = s;
    if (out) {
      float v = scale ? s * scale[co] : s;
      if (accumulate) v += out[o];
      out[o] = v;
    }
  }
}

int dg_wgrad_finish(const float* part, int nchunks, int ntaps, int Cin, int Cout, const float* scale, float* out,
                    float* raw, int accumulate, int oi, const float* colpart, int colC, const float* colscale,
                    float* colout, float* colraw, hipStream_t st) {
  return dg_wgrad_finish_rows(part, nchunks, ntaps, Cin, Cout, scale, out, raw, accumulate, oi, colpart, nchunks, colC,
                              colscale, colout, colraw, st);
}

int dg_wgrad_finish_rows(const float* part, int nchunks, int ntaps, int Cin, int Cout, const float* scale, float* out,
                         float* raw, int accumulate, int oi, const float* colpart, int colrows, int colC,
                         const float* colscale, float* colout, float* colraw, hipStream_t st) {
  const size_t n = (size_t)ntaps * Cin * Cout;
  const unsigned nbr = (unsigned)((n + 63) / 64);
  ColFin cf = {colpart, colscale, colout, colraw, colrows, colC};
  const unsigned ncol = colpart ? (unsigned)colC : 0u;
  // the slice count is a function of the layer alone (its size and slab count), never of the data: the summation
  // order of a given layer is fixed
  if (nbr < 2048 && nchunks >= 64)
    hipLaunchKernelGGL(slab_reduce_kernel<16>, dim3(nbr + ncol), dim3(1024), 0, st, part, nchunks, ntaps, Cin, Cout,
                       scale, out, raw, accumulate, oi, nbr, cf);
  else
    hipLaunchKernelGGL(slab_reduce_kernel<4>, dim3(nbr + ncol), dim3(256), 0, st, part, nchunks, ntaps, Cin, Cout, scale,
                       out, raw, accumulate, oi, nbr, cf);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

int dg_wgrad_reduce(const float* part, int nchunks, int ntaps, int Cin, int Cout, const float* scale, float* out,
                    float* raw, int accumulate, int oi, hipStream_t st) {
  return dg_wgrad_finish(part, nchunks, ntaps, Cin, Cout, scale, out, raw, accumulate, oi, nullptr, 0, nullptr, nullptr,
                         nullptr, st);
}
