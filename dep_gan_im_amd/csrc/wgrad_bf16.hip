// Weight-gradient contraction on the CDNA4 bf16 matrix cores (BASELINE configs[3]).
//
//   dW[tap][ci][co] = sum_{b,y,x} X[b, y+ty-p, x+tx-p, ci] * D[b, y, x, co]          (GT:549, 568, 594; A6)
//
// fp32 operands in HBM, rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while a tile is committed to LDS, fp32 accumulation
// in v_mfma_f32_32x32x16_bf16 -- the weight-gradient counterpart of igemm_bf16.hip.  GEMM view as in wgrad.hip: M = ci,
// N = co, K = pixels; one workgroup owns one (ci tile, co tile, tap group) and a contiguous range of TH x 16 pixel
// tiles, its 4 waves split every tile by rows, keep one accumulator tile per tap for the whole range, are summed
// through LDS at the end and written as ONE partial slab in wgrad.hip's format (the same deterministic finish launch
// reduces them: no float atomics).
//
// K runs along PIXELS, which NHWC memory strides by the channel count, while the MFMA wants eight consecutive k of one
// row per lane: both operands are K-major.  The LDS images stay [pixel][32 channels] (64-byte rows, written with 8-byte
// stores straight from the channel-contiguous loads) and are read with ds_read_b64_tr_b16, which hands a group of 16
// lanes a 4-pixel x 16-channel block transposed: lane i gets channel i of 4 consecutive pixels.  Two reads per operand
// and MFMA; the D fragment of a pixel row serves all taps.
//
// At 16x the fp32 matrix rate the contraction is a few per cent of the launch: the kernel is bound by the HBM reads of
// its operands (4 bytes per element), which register staging keeps in flight under the MFMAs of the previous tile.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int KS, int TPW, int TH>
struct WBCfg {
  static constexpr int NTAPS = KS * KS, PAD = KS / 2, TW = 16 + KS - 1;
  // all taps in one workgroup: the full halo; one tap ROW per workgroup (5x5): TH rows shifted by the group's row
  static constexpr int XR = (TPW == NTAPS) ? TH + KS - 1 : TH;
  static constexpr int XPIX = XR * TW, DPIX = TH * 16;
  static constexpr int XTOT = XPIX * 8, DTOT = DPIX * 8;                 // 16-byte fp32 pieces (4 channels) per tile
  static constexpr int NXP = (XTOT + 255) / 256, NDP = (DTOT + 255) / 256;
  static constexpr size_t LDS_TILE = (size_t)(XPIX + DPIX) * 32 * sizeof(__bf16);
  static constexpr size_t LDS_RED = (size_t)TPW * 4 * 32 * 32 * sizeof(float);
  static constexpr size_t LDS_BYTES = LDS_TILE > LDS_RED ? LDS_TILE : LDS_RED;
  static_assert(TH % 4 == 0, "the four waves split a tile by rows");
};

template <int KS, int TPW, int TH>
// (nine accumulator tiles + a tile in flight: 220 registers -- one workgroup per CU with the 512-register budget; the
// 76 KB a workgroup keeps in flight cover the HBM latency on their own)
__global__ __launch_bounds__(256, (TPW > 5) ? 1 : 2) void wgrad_bf16_kernel(const WgradArgs a) {
  typedef WBCfg<KS, TPW, TH> C;
  constexpr int PAD = C::PAD, TW = C::TW, NTAPS = C::NTAPS, NGT = NTAPS / TPW;
  constexpr int NXP = C::NXP, NDP = C::NDP, XTOT = C::XTOT, DTOT = C::DTOT;
  static_assert(NTAPS % TPW == 0 && (TPW == NTAPS || TPW == KS), "tap grouping");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* xs = reinterpret_cast<__bf16*>(smem_raw);      // [XPIX][32]
  __bf16* ds = xs + C::XPIX * 32;                        // [DPIX][32]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nCoT = (a.Cout + 31) / 32;
  int y, chunk;
  {
    // as wgrad_dma_kernel: the channel-tile pairs / tap groups of one pixel chunk on one XCD (they share its tiles)
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
  const int co0 = (y % nCoT) * 32;
  const int ci0 = (y / nCoT) * 32;
  const int tilesX = (a.W + 15) >> 4, tilesY = (a.H + TH - 1) / TH;
  const int t0 = chunk * a.tilesPerChunk;
  const int t1 = min(t0 + a.tilesPerChunk, a.nTiles);
  const int xrow0 = (TPW == NTAPS) ? -PAD : tg - PAD;    // image row of the X tile's first row, relative to the tile

  // ---- staging geometry: piece q = tid + 256 i -> pixel q / 8 = (tid >> 3) + 32 i, channels 4 (tid & 7) .. +3.  Only
  // the halo coordinates of the X pieces are kept in registers; everything else is a constant pattern of i.
  const int part4 = (tid & 7) * 4, pix0 = tid >> 3;
  const bool xch = (ci0 + part4) < a.Cin, dch = (co0 + part4) < a.Cout;
  int xyx[NXP];
#pragma unroll
  for (int i = 0; i < NXP; ++i) {
    const int pix = pix0 + 32 * i;
    const int ly = pix / TW, lx = pix - ly * TW;
    xyx[i] = (ly << 8) | lx;
  }

  f32x4 xr[NXP], dr[NDP];
  auto load_tile = [&](int tile) {
    int t = tile;
    const int tx0 = (t % tilesX) * 16;
    t /= tilesX;
    const int ty0 = (t % tilesY) * TH;
    const int b = t / tilesY;
    const float* xb = a.x.p + ci0 + part4 + (long)b * a.x.sB + (long)(ty0 + xrow0) * a.x.sY + (long)(tx0 - PAD) * a.x.sX;
    const float* db = a.dy.p + co0 + part4 + (long)b * a.dy.sB + (long)ty0 * a.dy.sY + (long)tx0 * a.dy.sX;
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
      const int ly = xyx[i] >> 8, lx = xyx[i] & 255;
      const int iy = ty0 + xrow0 + ly, ix = tx0 - PAD + lx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (xch && (tid + 256 * i) < XTOT && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
        v = *reinterpret_cast<const f32x4*>(xb + (long)ly * a.x.sY + (long)lx * a.x.sX);
      xr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NDP; ++i) {
      const int ly = (pix0 >> 4) + 2 * i, lx = pix0 & 15;
      const int iy = ty0 + ly, ix = tx0 + lx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (dch && (tid + 256 * i) < DTOT && iy < a.H && ix < a.W)
        v = *reinterpret_cast<const f32x4*>(db + (long)ly * a.dy.sY + (long)lx * a.dy.sX);
      dr[i] = v;
    }
  };
  auto commit_tile = [&]() {
    // fp32 -> bf16 here (plain casts = v_cvt_pk_bf16_f32, round to nearest even), 8-byte LDS stores
#pragma unroll
    for (int i = 0; i < NXP; ++i)
      if (tid + 256 * i < XTOT) {
        bf16x4 q = {(__bf16)xr[i][0], (__bf16)xr[i][1], (__bf16)xr[i][2], (__bf16)xr[i][3]};
        *reinterpret_cast<bf16x4*>(xs + (pix0 + 32 * i) * 32 + part4) = q;
      }
#pragma unroll
    for (int i = 0; i < NDP; ++i)
      if (tid + 256 * i < DTOT) {
        bf16x4 q = {(__bf16)dr[i][0], (__bf16)dr[i][1], (__bf16)dr[i][2], (__bf16)dr[i][3]};
        *reinterpret_cast<bf16x4*>(ds + (pix0 + 32 * i) * 32 + part4) = q;
      }
  };

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  // ---- transposed fragment addresses: lane = 16 g + 4 q + p: group g reads channels 16 (g & 1) .. +15 of pixels
  // 8 (g >> 1) + {0..3} (first read) / {4..7} (second): the lane supplies row q, channels 4p..4p+3 of its group's block
  const int g = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
  const int frag = (8 * (g >> 1) + fq) * 32 + 16 * (g & 1) + 4 * fp;     // element offset inside a 16-pixel run
  typedef __attribute__((address_space(3))) bf16x4* lds4_t;
  auto tr8 = [&](const __bf16* base) {   // eight k of one row: two transposed 4-pixel blocks
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(base));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(base + 4 * 32));
    bf16x8 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      r[k] = lo[k];
      r[4 + k] = hi[k];
    }
    return r;
  };

  if (t0 < t1) load_tile(t0);
  for (int tile = t0; tile < t1; ++tile) {
    __syncthreads();           // every wave has read the previous tile's images
    commit_tile();
    __syncthreads();
    if (tile + 1 < t1) load_tile(tile + 1);     // in flight under this tile's MFMAs
#pragma unroll
    for (int ry = 0; ry < TH / 4; ++ry) {
      const int yy = wv * (TH / 4) + ry;
      const bf16x8 bf = tr8(ds + yy * 16 * 32 + frag);
#pragma unroll
      for (int tl = 0; tl < TPW; ++tl) {
        const int ty = (TPW == NTAPS) ? (tl / KS) : 0;
        const int tx = (TPW == NTAPS) ? (tl % KS) : tl;
        const bf16x8 af = tr8(xs + ((yy + ty) * TW + tx) * 32 + frag);
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[tl], 0, 0, 0);
      }
    }
  }

  // ---- sum the 4 waves through LDS and write one slab per workgroup (wgrad.hip's format) ----
  float* red = reinterpret_cast<float*>(smem_raw);   // [TPW][4][32*32]
  const int r = lane & 31, h = lane >> 5;
  const size_t slab = (size_t)NTAPS * a.Cin * a.Cout;
  float* pout = a.part + (size_t)chunk * slab;
  __syncthreads();
#pragma unroll
  for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
    for (int j = 0; j < 16; ++j) red[(tl * 4 + wv) * 1024 + ((j & 3) + 8 * (j >> 2) + 4 * h) * 32 + r] = acc[tl][j];
  __syncthreads();
  for (int q = tid; q < TPW * 1024; q += 256) {
    const int tl = q >> 10, e = q & 1023;
    const float* rt = red + tl * 4096;
    const float sum = (rt[e] + rt[1024 + e]) + (rt[2048 + e] + rt[3072 + e]);
    const int tap = tg * TPW + tl;
    const int ci = ci0 + (e >> 5), co = co0 + (e & 31);
    if (ci < a.Cin && co < a.Cout) pout[((size_t)tap * a.Cin + ci) * a.Cout + co] = sum;
  }
}

struct WBVar {
  int KS, TPW, TH;
  size_t lds;
};
bool pick(int KS, WBVar* v) {
  v->KS = KS;
  if (KS == 3) { v->TPW = 9; v->TH = 16; v->lds = WBCfg<3, 9, 16>::LDS_BYTES; return true; }
  if (KS == 5) { v->TPW = 5; v->TH = 8; v->lds = WBCfg<5, 5, 8>::LDS_BYTES; return true; }
  if (KS == 1) { v->TPW = 1; v->TH = 16; v->lds = WBCfg<1, 1, 16>::LDS_BYTES; return true; }
  return false;
}
void chunking(const WBVar& v, int B, int H, int W, int Cin, int Cout, int* nTiles, int* tpc, int* nch, int* gy) {
  const int tilesX = cdiv(W, 16), tilesY = cdiv(H, v.TH);
  *nTiles = B * tilesX * tilesY;
  *gy = cdiv(Cin, 32) * cdiv(Cout, 32) * (v.KS * v.KS / v.TPW);
  // one round of resident workgroups (the slab count stays small for the finish launch)
  int want = dg_cu_count() * (v.TPW > 5 ? 1 : 2) / *gy;
  if (want < 1) want = 1;
  if (want > *nTiles) want = *nTiles;
  *tpc = cdiv(*nTiles, want);
  *nch = cdiv(*nTiles, *tpc);
}

template <int KS, int TPW, int TH>
int launch(WgradArgs a, int nch, int gy, hipStream_t st) {
  constexpr size_t lds = WBCfg<KS, TPW, TH>::LDS_BYTES;
  static DgOncePerDevice once;
  if (once.need())
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<KS, TPW, TH>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((wgrad_bf16_kernel<KS, TPW, TH>), dim3(nch, gy), dim3(256), lds, st, a);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

}  // namespace

bool dg_wgrad_bf16_supported(int KS, int Cin, int Cout) {
  return (KS == 1 || KS == 3 || KS == 5) && Cin >= 8 && (Cin % 4) == 0 && (Cout % 4) == 0;
}

size_t dg_wgrad_bf16_part_floats(int KS, int B, int H, int W, int Cin, int Cout) {
  WBVar v;
  if (!pick(KS, &v)) return 0;
  int nTiles, tpc, nch, gy;
  chunking(v, B, H, W, Cin, Cout, &nTiles, &tpc, &nch, &gy);
  return (size_t)nch * KS * KS * Cin * Cout;
}

int dg_wgrad_bf16(int KS, const WgradArgs& a_in, int* nchunks_out, hipStream_t st) {
  WgradArgs a = a_in;
  WBVar v;
  if (!pick(KS, &v) || !dg_wgrad_bf16_supported(KS, a.Cin, a.Cout)) {
    dg_set_error("dg_wgrad_bf16: unsupported shape (KS=%d Cin=%d Cout=%d)", KS, a.Cin, a.Cout);
    return DG_ERR_UNSUPPORTED;
  }
  if ((a.x.sX % 4) || (a.x.sY % 4) || (a.x.sB % 4) || (a.dy.sX % 4) || (a.dy.sY % 4) || (a.dy.sB % 4) ||
      (((uintptr_t)a.x.p) & 15) || (((uintptr_t)a.dy.p) & 15)) {
    dg_set_error("dg_wgrad_bf16: strides must be multiples of 4 floats and the operands 16-byte aligned");
    return DG_ERR_ARG;
  }
  int nTiles, tpc, nch, gy;
  chunking(v, a.B, a.H, a.W, a.Cin, a.Cout, &nTiles, &tpc, &nch, &gy);
  a.nTiles = nTiles;
  a.tilesPerChunk = tpc;
  a.colpart = nullptr;
  *nchunks_out = nch;
  if (KS == 3) return launch<3, 9, 16>(a, nch, gy, st);
  if (KS == 5) return launch<5, 5, 8>(a, nch, gy, st);
  return launch<1, 1, 16>(a, nch, gy, st);
}
