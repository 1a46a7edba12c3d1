// Direct (VALU) NHWC convolution for the HBM-bound edge layers, where an MFMA
// tile would be mostly padding: Cin = 1/2 (gen_0 GT:398, dis_0a GT:319 and its
// gradient-penalty u-forward) and Cout = 1 (backward-data of dis_0a onto the
// image, the d D(x)/dx of GT:543).  Also the fallback for channel counts the
// MFMA kernel does not take.  Thread = (pixel, group of COG output channels);
// the input halo tile sits in LDS channel-major ([c][pixel]) so lanes that walk
// along x read consecutive banks; stores are COG-wide and coalesced.
#include "common.h"
#include "epilogue.h"

template <int KS, int COG, int G>
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvArgs a) {
  constexpr int PAD = KS / 2;
  constexpr int TW = 16 + KS - 1;
  constexpr int PIXT = TW * TW;
  constexpr int NTAPS = KS * KS;
  constexpr int CT = COG * G;       // output channels per block
  constexpr int CIK = 8;            // input channels per LDS chunk
  constexpr int PPP = 256 / G;      // pixels per pass
  constexpr int NPASS = G;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                   // [CIK][PIXT]
  float* ws = smem + CIK * PIXT;      // [NTAPS][CIK][CT]

  const int tid = threadIdx.x;
  const int tilesX = (a.W + 15) >> 4, tilesY = (a.H + 15) >> 4;
  int t = blockIdx.x;
  const int tx0 = (t % tilesX) * 16;
  t /= tilesX;
  const int ty0 = (t % tilesY) * 16;
  const int b = t / tilesY;
  const int co0 = blockIdx.y * CT;
  const int g = tid % G, p0 = tid / G;
  const float* inb = a.in.p + (long)b * a.in.sB;

  float acc[NPASS][COG];
#pragma unroll
  for (int p = 0; p < NPASS; ++p)
#pragma unroll
    for (int c = 0; c < COG; ++c) acc[p][c] = 0.f;

  for (int c0 = 0; c0 < a.Cin; c0 += CIK) {
    const int cik = min(CIK, a.Cin - c0);
    __syncthreads();
    for (int q = tid; q < PIXT * cik; q += 256) {
      const int pix = q / cik, c = q - pix * cik;
      const int ly = pix / TW, lx = pix - ly * TW;
      const int iy = ty0 + ly - PAD, ix = tx0 + lx - PAD;
      float v = 0.f;
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = inb[(long)iy * a.in.sY + (long)ix * a.in.sX + c0 + c];
      xs[c * PIXT + pix] = v;
    }
    for (int q = tid; q < NTAPS * cik * CT; q += 256) {
      const int n = q % CT;
      const int c = (q / CT) % cik;
      const int tap = q / (CT * cik);
      const int ts = a.flip ? (NTAPS - 1 - tap) : tap;
      float v = 0.f;
      if (co0 + n < a.Cout) v = a.w[(long)ts * a.wsT + (long)(c0 + c) * a.wsI + (long)(co0 + n) * a.wsO];
      ws[(tap * CIK + c) * CT + n] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int tap = 0; tap < NTAPS; ++tap) {
      const int ty = tap / KS, tx = tap - ty * KS;
      for (int c = 0; c < cik; ++c) {
        float wv[COG];
#pragma unroll
        for (int k = 0; k < COG; ++k) wv[k] = ws[(tap * CIK + c) * CT + g * COG + k];
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
          const int pix = p * PPP + p0;
          const int py = pix >> 4, px = pix & 15;
          const float xv = xs[c * PIXT + (py + ty) * TW + px + tx];
#pragma unroll
          for (int k = 0; k < COG; ++k) acc[p][k] = fmaf(xv, wv[k], acc[p][k]);
        }
      }
    }
  }

  if constexpr (COG == 4) {
    if (a.vec4) {
      // one 16-byte store per pixel pass instead of four dword stores (the CU is store-issue bound)
      const int co = co0 + g * 4;
      if (co < a.Cout) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
          const int pix = p * PPP + p0;
          const int oy = ty0 + (pix >> 4), ox = tx0 + (pix & 15);
          if (oy < a.H && ox < a.W) {
            const f32x4 v = {acc[p][0], acc[p][1], acc[p][2], acc[p][3]};
            epi_store4(a, b, oy, ox, co, v);
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int k = 0; k < COG; ++k) {
    const int co = co0 + g * COG + k;
    if (co < a.Cout) {
      const EpiChan ch = epi_load_chan(a.ep, b, co, a.Cout);
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int pix = p * PPP + p0;
        const int oy = ty0 + (pix >> 4), ox = tx0 + (pix & 15);
        if (oy < a.H && ox < a.W) epi_store(a, ch, b, oy, ox, co, acc[p][k]);
      }
    }
  }
}

template <int KS, int COG, int G>
static int launch_direct(const ConvArgs& a, hipStream_t st) {
  constexpr int TW = 16 + KS - 1;
  constexpr size_t lds = (size_t)(8 * TW * TW + KS * KS * 8 * COG * G) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_direct_kernel<KS, COG, G>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid((unsigned)(cdiv(a.W, 16) * cdiv(a.H, 16) * a.B), (unsigned)cdiv(a.Cout, COG * G));
  hipLaunchKernelGGL((conv_direct_kernel<KS, COG, G>), grid, dim3(256), lds, st, a);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

int dg_conv_direct(int KS, const ConvArgs& a_in, hipStream_t st) {
  ConvArgs a = a_in;
  auto aligned = [](const TView& v) {
    return !v.p || (!(v.sX % 4) && !(v.sY % 4) && !(v.sB % 4) && !(((uintptr_t)v.p) & 15));
  };
  a.vec4 = (a.Cout % 4 == 0) && aligned(a.out) && aligned(a.ep.res) && aligned(a.ep.mask) && aligned(a.ep.out_pre) &&
           (!a.ep.film_mul || a.ep.film_ld % 4 == 0);
  const int sel = (a.Cout == 1) ? 0 : ((a.Cout <= 16) ? 1 : 2);
  if (KS == 3) {
    if (sel == 0) return launch_direct<3, 1, 1>(a, st);
    if (sel == 1) return launch_direct<3, 4, 4>(a, st);
    return launch_direct<3, 4, 8>(a, st);
  }
  if (KS == 5) {
    if (sel == 0) return launch_direct<5, 1, 1>(a, st);
    if (sel == 1) return launch_direct<5, 4, 4>(a, st);
    return launch_direct<5, 4, 8>(a, st);
  }
  if (KS == 1) {
    if (sel == 0) return launch_direct<1, 1, 1>(a, st);
    if (sel == 1) return launch_direct<1, 4, 4>(a, st);
    return launch_direct<1, 4, 8>(a, st);
  }
  dg_set_error("dg_conv_direct: unsupported kernel size %d", KS);
  return DG_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------
// weight gradient for the small-Cin edge layers (gen_0, dis_0a): same slab
// format as the MFMA wgrad so dg_wgrad_reduce finishes it.
// ---------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void wgrad_small_kernel(const WgradArgs a) {
  constexpr int PAD = KS / 2;
  constexpr int TW = 16 + KS - 1;
  constexpr int PIXT = TW * TW;
  constexpr int NTAPS = KS * KS;
  constexpr int NOUT = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                    // [Cin][PIXT]
  float* ds = smem + a.Cin * PIXT;     // [256][Cout]
  const int tid = threadIdx.x;
  const int nout = NTAPS * a.Cin * a.Cout;
  const int tilesX = (a.W + 15) >> 4, tilesY = (a.H + 15) >> 4;
  const int t0 = blockIdx.x * a.tilesPerChunk, t1 = min(t0 + a.tilesPerChunk, a.nTiles);
  float acc[NOUT];
  int xoff[NOUT], co_[NOUT];
#pragma unroll
  for (int k = 0; k < NOUT; ++k) {
    acc[k] = 0.f;
    const int o = tid + k * 256;
    const int oo = (o < nout) ? o : 0;
    const int co = oo % a.Cout, ci = (oo / a.Cout) % a.Cin, tap = oo / (a.Cout * a.Cin);
    xoff[k] = ci * PIXT + (tap / KS) * TW + (tap % KS);
    co_[k] = co;
  }
  for (int tile = t0; tile < t1; ++tile) {
    int t = tile;
    const int tx0 = (t % tilesX) * 16;
    t /= tilesX;
    const int ty0 = (t % tilesY) * 16;
    const int b = t / tilesY;
    __syncthreads();
    for (int q = tid; q < PIXT * a.Cin; q += 256) {
      const int pix = q / a.Cin, c = q - pix * a.Cin;
      const int ly = pix / TW, lx = pix - ly * TW;
      const int iy = ty0 + ly - PAD, ix = tx0 + lx - PAD;
      float v = 0.f;
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = a.x.p[view_off(a.x, b, iy, ix) + c];
      xs[c * PIXT + pix] = v;
    }
    for (int q = tid; q < 256 * a.Cout; q += 256) {
      const int pix = q / a.Cout, c = q - pix * a.Cout;
      const int iy = ty0 + (pix >> 4), ix = tx0 + (pix & 15);
      float v = 0.f;
      if (iy < a.H && ix < a.W) v = a.dy.p[view_off(a.dy, b, iy, ix) + c];
      ds[pix * a.Cout + c] = v;
    }
    __syncthreads();
    for (int py = 0; py < 16; ++py)
#pragma unroll 4
      for (int px = 0; px < 16; ++px) {
        const int xo = py * TW + px;
        const int dof = (py * 16 + px) * a.Cout;
#pragma unroll
        for (int k = 0; k < NOUT; ++k) acc[k] = fmaf(xs[xoff[k] + xo], ds[dof + co_[k]], acc[k]);
      }
  }
  float* pout = a.part + (size_t)blockIdx.x * nout;
#pragma unroll
  for (int k = 0; k < NOUT; ++k) {
    const int o = tid + k * 256;
    if (o < nout) pout[o] = acc[k];
  }
}

// MFMA form of the edge-layer weight gradient: M = taps*Cin (<= 32), N = Cout (16 or 32),
// K = pixels, v_mfma_f32_16x16x4_f32.  A-lane i holds the (tap, ci) row: its LDS address is the
// pixel offset plus a per-lane tap offset into the channel-major halo tile.
template <int KS>
__global__ __launch_bounds__(256) void wgrad_edge_kernel(const WgradArgs a) {
  constexpr int PAD = KS / 2;
  constexpr int TW = 16 + KS - 1;
  constexpr int PIXT = TW * TW;
  constexpr int NTAPS = KS * KS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int xsz = (a.Cin * PIXT + 3) & ~3;
  float* xs = smem;          // [Cin][PIXT]
  float* ds = smem + xsz;    // [256][Cout]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int M = NTAPS * a.Cin;
  const int MT = (M + 15) >> 4, NTl = a.Cout >> 4;
  int aoff[2];
  bool aval[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = mt * 16 + i;
    aval[mt] = m < M;
    const int mm = aval[mt] ? m : 0;
    const int tap = mm / a.Cin, ci = mm - tap * a.Cin;
    aoff[mt] = ci * PIXT + (tap / KS) * TW + (tap % KS);
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int tilesX = (a.W + 15) >> 4, tilesY = (a.H + 15) >> 4;
  const int t0 = blockIdx.x * a.tilesPerChunk, t1 = min(t0 + a.tilesPerChunk, a.nTiles);
  const int C4 = a.Cout >> 2;
  for (int tile = t0; tile < t1; ++tile) {
    int t = tile;
    const int tx0 = (t % tilesX) * 16;
    t /= tilesX;
    const int ty0 = (t % tilesY) * 16;
    const int b = t / tilesY;
    __syncthreads();
    for (int q = tid; q < PIXT * a.Cin; q += 256) {
      const int pix = q / a.Cin, c = q - pix * a.Cin;
      const int ly = pix / TW, lx = pix - ly * TW;
      const int iy = ty0 + ly - PAD, ix = tx0 + lx - PAD;
      float v = 0.f;
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = a.x.p[view_off(a.x, b, iy, ix) + c];
      xs[c * PIXT + pix] = v;
    }
    for (int q = tid; q < 256 * C4; q += 256) {
      const int pix = q / C4, c = (q - pix * C4) * 4;
      const int iy = ty0 + (pix >> 4), ix = tx0 + (pix & 15);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (iy < a.H && ix < a.W) v = *reinterpret_cast<const f32x4*>(a.dy.p + view_off(a.dy, b, iy, ix) + c);
      *reinterpret_cast<f32x4*>(ds + pix * a.Cout + c) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
      const int kl = kk * 4 + kq;
      const int py = 4 * wv + (kl >> 4), px = kl & 15;
      const int po = py * TW + px;
      float av[2], bv[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) av[mt] = aval[mt] ? xs[aoff[mt] + po] : 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bv[nt] = (nt < NTl) ? ds[(py * 16 + px) * a.Cout + nt * 16 + i] : 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          if (mt < MT && nt < NTl) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
    }
  }
  __syncthreads();
  float* red = smem;  // [4][32*32]
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wv * 1024 + (mt * 16 + 4 * kq + j) * 32 + nt * 16 + i] = acc[mt][nt][j];
  __syncthreads();
  float* pout = a.part + (size_t)blockIdx.x * M * a.Cout;
  for (int e = tid; e < M * a.Cout; e += 256) {
    const int m = e / a.Cout, n = e - m * a.Cout;
    const int o = m * 32 + n;
    pout[e] = (red[o] + red[1024 + o]) + (red[2048 + o] + red[3072 + o]);
  }
}

static void small_chunking(int B, int H, int W, int* nTiles, int* tpc, int* nch) {
  *nTiles = B * cdiv(W, 16) * cdiv(H, 16);
  int want = 1024;
  if (want > *nTiles) want = *nTiles;
  *tpc = cdiv(*nTiles, want);
  *nch = cdiv(*nTiles, *tpc);
}

size_t dg_wgrad_small_part_floats(int KS, int B, int H, int W, int Cin, int Cout) {
  int nTiles, tpc, nch;
  small_chunking(B, H, W, &nTiles, &tpc, &nch);
  const size_t slab = (size_t)KS * KS * Cin * Cout;
  return (size_t)nch * slab + (size_t)cdiv(nch, 32) * slab;
}

int dg_wgrad_small(int KS, const WgradArgs& a_in, int* nchunks, hipStream_t st) {
  WgradArgs a = a_in;
  if (KS * KS * a.Cin * a.Cout > 1024 || a.Cout > 64) {
    dg_set_error("dg_wgrad_small: taps*Cin*Cout = %d too large", KS * KS * a.Cin * a.Cout);
    return DG_ERR_UNSUPPORTED;
  }
  int nTiles, tpc, nch;
  small_chunking(a.B, a.H, a.W, &nTiles, &tpc, &nch);
  a.nTiles = nTiles;
  a.tilesPerChunk = tpc;
  *nchunks = nch;
  const int TW = 16 + KS - 1;
  const bool edge = (KS * KS * a.Cin <= 32) && (a.Cout == 16 || a.Cout == 32) && (a.dy.sX % 4 == 0) &&
                    (a.dy.sY % 4 == 0) && (a.dy.sB % 4 == 0);
  if (edge) {
    size_t lds = (size_t)(((a.Cin * TW * TW + 3) & ~3) + 256 * a.Cout) * sizeof(float);
    if (lds < 4 * 1024 * sizeof(float)) lds = 4 * 1024 * sizeof(float);
    if (KS == 3)
      hipLaunchKernelGGL(wgrad_edge_kernel<3>, dim3(nch), dim3(256), lds, st, a);
    else if (KS == 5)
      hipLaunchKernelGGL(wgrad_edge_kernel<5>, dim3(nch), dim3(256), lds, st, a);
    else {
      dg_set_error("dg_wgrad_small: unsupported kernel size %d", KS);
      return DG_ERR_UNSUPPORTED;
    }
    HIPCHECK(hipGetLastError());
    return DG_OK;
  }
  const size_t lds = (size_t)(a.Cin * TW * TW + 256 * a.Cout) * sizeof(float);
  if (KS == 3)
    hipLaunchKernelGGL(wgrad_small_kernel<3>, dim3(nch), dim3(256), lds, st, a);
  else if (KS == 5)
    hipLaunchKernelGGL(wgrad_small_kernel<5>, dim3(nch), dim3(256), lds, st, a);
  else {
    dg_set_error("dg_wgrad_small: unsupported kernel size %d", KS);
    return DG_ERR_UNSUPPORTED;
  }
  HIPCHECK(hipGetLastError());
  return DG_OK;
}
