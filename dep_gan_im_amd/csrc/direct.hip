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
  static DgOncePerDevice once;
  if (once.need()) {
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_direct_kernel<KS, COG, G>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  dim3 grid((unsigned)(cdiv(a.W, 16) * cdiv(a.H, 16) * a.B), (unsigned)cdiv(a.Cout, COG * G));
  hipLaunchKernelGGL((conv_direct_kernel<KS, COG, G>), grid, dim3(256), lds, st, a);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// Cout = 1 (backward-data of dis_0a onto the image, the dD/dx of GT:543): one output channel leaves nothing to
// amortise an input value over except neighbouring pixels, and the generic kernel above spends two LDS reads per FMA.
// Here a thread owns 4 consecutive pixels of a 32 x 32 tile: per (channel, tap row) it reads 8 inputs (two 16-byte
// LDS reads) and KS weights for 4 KS FMAs -- 4 LDS instructions per 20 FMAs for 5x5.
// ---------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void conv_cout1_kernel(const ConvArgs a) {
  constexpr int PAD = KS / 2;
  constexpr int TWX = 32 + KS - 1, TWY = 32 + KS - 1;
  constexpr int RS = 40;            // LDS row stride (floats): 16-byte aligned rows, 8 floats of slack past column 35
  constexpr int CIK = 8;            // input channels per LDS chunk
  constexpr int WS = 8;             // padded tap-row length
  static_assert(TWX <= RS && KS <= WS, "tile row must fit the LDS row");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                          // [CIK][TWY][RS]
  float* ws = smem + CIK * TWY * RS;         // [CIK][KS][WS]

  const int tid = threadIdx.x;
  const int tilesX = (a.W + 31) >> 5, tilesY = (a.H + 31) >> 5;
  int t = blockIdx.x;
  const int tx0 = (t % tilesX) * 32;
  t /= tilesX;
  const int ty0 = (t % tilesY) * 32;
  const int b = t / tilesY;
  const int qx = (tid & 7) * 4, py = tid >> 3;      // this thread's 4 pixels: row py, columns qx .. qx+3
  const float* inb = a.in.p + (long)b * a.in.sB;
  const bool in4 = !(a.Cin % 4) && !(a.in.sX % 4) && !(a.in.sY % 4) && !(a.in.sB % 4) && !(((uintptr_t)a.in.p) & 15);

  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < a.Cin; c0 += CIK) {
    const int cik = min(CIK, a.Cin - c0);
    __syncthreads();
    if (in4 && cik == CIK) {
      for (int q = tid; q < TWX * TWY * 2; q += 256) {
        const int pix = q >> 1, part = (q & 1) * 4;
        const int ly = pix / TWX, lx = pix - ly * TWX;
        const int iy = ty0 + ly - PAD, ix = tx0 + lx - PAD;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
          v = *reinterpret_cast<const f32x4*>(inb + (long)iy * a.in.sY + (long)ix * a.in.sX + c0 + part);
#pragma unroll
        for (int k = 0; k < 4; ++k) xs[((part + k) * TWY + ly) * RS + lx] = v[k];
      }
    } else {
      for (int q = tid; q < TWX * TWY * cik; q += 256) {
        const int pix = q / cik, c = q - pix * cik;
        const int ly = pix / TWX, lx = pix - ly * TWX;
        const int iy = ty0 + ly - PAD, ix = tx0 + lx - PAD;
        float v = 0.f;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = inb[(long)iy * a.in.sY + (long)ix * a.in.sX + c0 + c];
        xs[(c * TWY + ly) * RS + lx] = v;
      }
    }
    for (int q = tid; q < cik * KS * WS; q += 256) {
      const int tx = q % WS, ty = (q / WS) % KS, c = q / (WS * KS);
      float v = 0.f;
      if (tx < KS) {
        const int tap = ty * KS + tx;
        const int ts = a.flip ? (KS * KS - 1 - tap) : tap;
        v = a.w[(long)ts * a.wsT + (long)(c0 + c) * a.wsI];
      }
      ws[(c * KS + ty) * WS + tx] = v;
    }
    __syncthreads();
    for (int c = 0; c < cik; ++c) {
#pragma unroll
      for (int ty = 0; ty < KS; ++ty) {
        const float* row = xs + (c * TWY + py + ty) * RS + qx;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(row);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(row + 4);
        const float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        const float* wr = ws + (c * KS + ty) * WS;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr);
        const float w4 = (KS > 4) ? wr[4] : 0.f;
        const float wv[5] = {w0[0], w0[1], w0[2], w0[3], w4};
        // tap-major, then channel, then tap row, then tap column: a fixed order, so results are run-to-run identical
#pragma unroll
        for (int tx = 0; tx < KS; ++tx)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv[j + tx], wv[tx], acc[j]);
      }
    }
  }
  const EpiChan ch = epi_load_chan(a.ep, b, 0, a.Cout);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = ty0 + py, ox = tx0 + qx + j;
    if (oy < a.H && ox < a.W) epi_store(a, ch, b, oy, ox, 0, acc[j]);
  }
}

template <int KS>
static int launch_cout1(const ConvArgs& a, hipStream_t st) {
  constexpr int TW = 32 + KS - 1;
  constexpr size_t lds = (size_t)(8 * TW * 40 + 8 * KS * 8) * sizeof(float);
  static DgOncePerDevice once;
  if (once.need()) {
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_cout1_kernel<KS>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  dim3 grid((unsigned)(cdiv(a.W, 32) * cdiv(a.H, 32) * a.B));
  hipLaunchKernelGGL((conv_cout1_kernel<KS>), grid, dim3(256), lds, st, a);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------
// Cin = 1 or 2 (gen_0 GT:398, dis_0a GT:319 and its gradient-penalty u-forward): the output write is the only HBM
// traffic that matters and the work is KS^2 Cin FMAs per output value.  Thread = 4 consecutive pixels x 4 consecutive
// output channels; per (channel, tap row) it reads 8 inputs (two 16-byte LDS reads) and KS weight quads for 16 KS
// FMAs (7 LDS instructions per 80 FMAs for 5x5, where the generic kernel needs 25).  Block = 16 pixels wide,
// 256 / (Cout / 4) / 4 rows high; Cout / 4 threads cover one pixel group's channels, so a wave's 16-byte stores fill
// whole lines.
// ---------------------------------------------------------------------------
template <int KS, int G>   // G = Cout / 4 channel groups: 4 (Cout 16) or 8 (Cout 32)
__global__ __launch_bounds__(256) void conv_cin12_kernel(const ConvArgs a) {
  constexpr int PAD = KS / 2;
  constexpr int ROWS = 256 / G / 4;          // tile height
  constexpr int TWX = 16 + KS - 1, TWY = ROWS + KS - 1;
  constexpr int RS = 24;                     // LDS row stride (floats)
  constexpr int CT = 4 * G;
  static_assert(TWX + 3 <= RS, "row with read slack must fit");
  __shared__ __attribute__((aligned(16))) float xs[2 * TWY * RS];
  __shared__ __attribute__((aligned(16))) float ws[2 * KS * KS * CT];

  const int tid = threadIdx.x;
  const int tilesX = (a.W + 15) >> 4, tilesY = (a.H + ROWS - 1) / ROWS;
  int t = blockIdx.x;
  const int tx0 = (t % tilesX) * 16;
  t /= tilesX;
  const int ty0 = (t % tilesY) * ROWS;
  const int b = t / tilesY;
  const int g = tid % G, pg = tid / G;       // channel group, pixel group
  const int qx = (pg & 3) * 4, py = pg >> 2;
  const float* inb = a.in.p + (long)b * a.in.sB;

  for (int q = tid; q < a.Cin * TWY * RS; q += 256) {
    const int lx = q % RS, ly = (q / RS) % TWY, c = q / (RS * TWY);
    const int iy = ty0 + ly - PAD, ix = tx0 + lx - PAD;
    float v = 0.f;
    if (lx < TWX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = inb[(long)iy * a.in.sY + (long)ix * a.in.sX + c];
    xs[q] = v;
  }
  for (int q = tid; q < a.Cin * KS * KS * CT; q += 256) {
    const int n = q % CT, tap = (q / CT) % (KS * KS), c = q / (CT * KS * KS);
    const int ts = a.flip ? (KS * KS - 1 - tap) : tap;
    ws[q] = (n < a.Cout) ? a.w[(long)ts * a.wsT + (long)c * a.wsI + (long)n * a.wsO] : 0.f;
  }
  __syncthreads();

  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // channel, tap row, tap column: the generic kernel's order is tap-major; either is a fixed order
  for (int c = 0; c < a.Cin; ++c) {
#pragma unroll
    for (int ty = 0; ty < KS; ++ty) {
      const float* row = xs + (c * TWY + py + ty) * RS + qx;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(row);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(row + 4);
      const float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
      for (int tx = 0; tx < KS; ++tx) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(ws + ((c * KS + ty) * KS + tx) * CT + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[j][k] = fmaf(xv[j + tx], w4[k], acc[j][k]);
      }
    }
  }
  const int co = 4 * g;
  if (co < a.Cout) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int oy = ty0 + py, ox = tx0 + qx + j;
      if (oy < a.H && ox < a.W) epi_store4(a, b, oy, ox, co, acc[j]);
    }
  }
}

template <int KS, int G>
static int launch_cin12(const ConvArgs& a, hipStream_t st) {
  constexpr int ROWS = 256 / G / 4;
  dim3 grid((unsigned)(cdiv(a.W, 16) * cdiv(a.H, ROWS) * a.B));
  hipLaunchKernelGGL((conv_cin12_kernel<KS, G>), grid, dim3(256), 0, st, a);
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
  if (a.Cin <= 2 && a.vec4 && (a.Cout == 16 || a.Cout == 32) && (KS == 3 || KS == 5)) {
    if (KS == 3) return (a.Cout == 16) ? launch_cin12<3, 4>(a, st) : launch_cin12<3, 8>(a, st);
    return (a.Cout == 16) ? launch_cin12<5, 4>(a, st) : launch_cin12<5, 8>(a, st);
  }
  if (KS == 3) {
    if (sel == 0) return (a.Cin >= 4) ? launch_cout1<3>(a, st) : launch_direct<3, 1, 1>(a, st);
    if (sel == 1) return launch_direct<3, 4, 4>(a, st);
    return launch_direct<3, 4, 8>(a, st);
  }
  if (KS == 5) {
    if (sel == 0) return (a.Cin >= 4) ? launch_cout1<5>(a, st) : launch_direct<5, 1, 1>(a, st);
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
  // about six tiles per block: a block stages and contracts one tile at a time, so the loads of a CU are hidden only
  // by its other resident blocks (1024 / 2048 / 4096 blocks on the batch-96 5x5 layer: 192 / 181 / 156 us)
  int want = *nTiles / 6;
  if (want < 1024) want = 1024;
  if (want > 8192) want = 8192;
  if (want > *nTiles) want = *nTiles;
  *tpc = cdiv(*nTiles, want);
  *nch = cdiv(*nTiles, *tpc);
}

size_t dg_wgrad_small_part_floats(int KS, int B, int H, int W, int Cin, int Cout) {
  int nTiles, tpc, nch;
  small_chunking(B, H, W, &nTiles, &tpc, &nch);
  const size_t slab = (size_t)KS * KS * Cin * Cout;
  return (size_t)nch * slab;
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
