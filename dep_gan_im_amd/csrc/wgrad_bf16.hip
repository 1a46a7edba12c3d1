// Weight-gradient contraction on the CDNA4 bf16 matrix cores (BASELINE configs[3]).
//
//   dW[tap][ci][co] = sum_{b,y,x} X[b, y+ty-p, x+tx-p, ci] * D[b, y, x, co]          (GT:549, 568, 594; A6)
//
// fp32 operands in HBM, rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while a tile is committed to LDS, fp32 accumulation
// in v_mfma_f32_32x32x16_bf16 -- the weight-gradient counterpart of igemm_bf16.hip.  GEMM view as in wgrad.hip: M = ci,
// N = co, K = pixels; one workgroup owns one (ci tile, co tile) and a contiguous range of 16 x 16 pixel tiles and
// writes ONE partial slab in wgrad.hip's format (the same deterministic finish launch reduces them: no float atomics).
// Its 4 waves split the TAPS (wave w owns taps w, w + 4, ...: 3 / 2 / 2 / 2 of a 3x3 kernel, 7 / 6 / 6 / 6 of a 5x5 one)
// and walk all 16 pixel rows of a tile: every tap of a kernel size is served from ONE staging of the halo tile (a split
// by tap groups across workgroups multiplied the L2 traffic of the 5x5 layers by five), a wave's accumulators need no
// cross-wave reduction, and 48 / 112 accumulator registers leave room for two workgroups per CU.  (1x1: the waves split
// the rows and are summed through LDS.)
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

template <int KS>
struct WBCfg {
  static constexpr int NTAPS = KS * KS, PAD = KS / 2, TW = 16 + KS - 1, TH = 16;
  static constexpr int TPWV = (NTAPS + 3) / 4;                          // taps per wave (KS > 1)
  static constexpr int XPIX = TW * TW, DPIX = TH * 16;
  static constexpr int XTOT = XPIX * 8, DTOT = DPIX * 8;                 // 16-byte fp32 pieces (4 channels) per tile
  static constexpr int NXP = (XTOT + 255) / 256, NDP = (DTOT + 255) / 256;
  static constexpr size_t LDS_TILE = (size_t)(XPIX + DPIX) * 32 * sizeof(__bf16);
  static constexpr size_t LDS_RED = (KS == 1) ? (size_t)4 * 32 * 32 * sizeof(float) : 0;
  static constexpr size_t LDS_BYTES = LDS_TILE > LDS_RED ? LDS_TILE : LDS_RED;
};

// (5x5: 7 x 16 accumulators + a 20 x 20 halo tile in flight = 200 registers: one workgroup per CU)
template <int KS>
__global__ __launch_bounds__(256, (KS == 5) ? 1 : 2) void wgrad_bf16_kernel(const WgradArgs a) {
  typedef WBCfg<KS> C;
  constexpr int PAD = C::PAD, TW = C::TW, NTAPS = C::NTAPS, TH = C::TH, TPWV = C::TPWV;
  constexpr int NXP = C::NXP, NDP = C::NDP, XTOT = C::XTOT, DTOT = C::DTOT;
  constexpr int NACC = (KS == 1) ? 1 : TPWV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* xs = reinterpret_cast<__bf16*>(smem_raw);      // [XPIX][32]
  __bf16* ds = xs + C::XPIX * 32;                        // [DPIX][32]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nCoT = (a.Cout + 31) / 32;
  int y, chunk;
  {
    // as wgrad_dma_kernel: the channel-tile pairs of one pixel chunk on one XCD (they share its tiles)
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
  const int co0 = (y % nCoT) * 32;
  const int ci0 = (y / nCoT) * 32;
  const int tilesX = (a.W + 15) >> 4, tilesY = (a.H + TH - 1) / TH;
  const int t0 = chunk * a.tilesPerChunk;
  const int t1 = min(t0 + a.tilesPerChunk, a.nTiles);

  // ---- staging geometry: piece q = tid + 256 i -> pixel q / 8 = (tid >> 3) + 32 i, channels 4 (tid & 7) .. +3: a
  // constant pattern of i (the halo coordinates are a division by a constant away; the kernel waits for HBM, not for VALU)
  const int part4 = (tid & 7) * 4, pix0 = tid >> 3;
  const bool xch = (ci0 + part4) < a.Cin, dch = (co0 + part4) < a.Cout;

  f32x4 xr[NXP], dr[NDP];
  auto load_tile = [&](int tile) {
    int t = tile;
    const int tx0 = (t % tilesX) * 16;
    t /= tilesX;
    const int ty0 = (t % tilesY) * TH;
    const int b = t / tilesY;
    const float* xb = a.x.p + ci0 + part4 + (long)b * a.x.sB + (long)(ty0 - PAD) * a.x.sY + (long)(tx0 - PAD) * a.x.sX;
    const float* db = a.dy.p + co0 + part4 + (long)b * a.dy.sB + (long)ty0 * a.dy.sY + (long)tx0 * a.dy.sX;
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
      const int pix = pix0 + 32 * i;
      const int ly = pix / TW, lx = pix - ly * TW;
      const int iy = ty0 - PAD + ly, ix = tx0 - PAD + lx;
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

  // Column sums of dy (bias / BN-beta gradients) ride along where asked for: every dy piece passes through this
  // thread's registers exactly once per workgroup of input-channel tile 0, unrounded, so four adds per piece and tile
  // (samples b < colB only: the critics' penalty third of the batch has no bias gradient) give per-thread partial sums
  // that are folded through LDS at the end -- no second pass over dy.
  const bool do_cs = a.colpart != nullptr && ci0 == 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  const int tilesPerSample = tilesX * tilesY;

  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
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
  // this wave's taps (KS > 1): tap w + 4 i -> halo offset of its shifted pixel row; a tap index beyond the kernel (the
  // last round of the 3x3 / 5x5 split) is clamped to a valid address and its accumulator is never written
  int tapoff[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    const int tap = min(wv + 4 * i, NTAPS - 1);
    tapoff[i] = ((tap / KS) * TW + (tap % KS)) * 32;
  }

  if (t0 < t1) load_tile(t0);
  for (int tile = t0; tile < t1; ++tile) {
    __syncthreads();           // every wave has read the previous tile's images
    commit_tile();
    if (do_cs && tile / tilesPerSample < a.colB) {
#pragma unroll
      for (int i = 0; i < NDP; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) csum[k] += dr[i][k];
    }
    __syncthreads();
    if (tile + 1 < t1) load_tile(tile + 1);     // in flight under this tile's MFMAs
    if (KS == 1) {
#pragma unroll
      for (int ry = 0; ry < TH / 4; ++ry) {
        const int yy = wv * (TH / 4) + ry;
        const bf16x8 bf = tr8(ds + yy * 16 * 32 + frag);
        const bf16x8 af = tr8(xs + yy * TW * 32 + frag);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[0], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int yy = 0; yy < TH; ++yy) {
        const bf16x8 bf = tr8(ds + yy * 16 * 32 + frag);       // the D fragment of a pixel row serves all taps
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
          const bf16x8 af = tr8(xs + yy * TW * 32 + tapoff[i] + frag);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[i], 0, 0, 0);
        }
      }
    }
  }

  const int r = lane & 31, h = lane >> 5;
  const size_t slab = (size_t)NTAPS * a.Cin * a.Cout;
  float* pout = a.part + (size_t)chunk * slab;
  if (do_cs) {
    // thread t holds channels 4 (t & 7) .. +3 of the pixels it staged: fold the 32 threads of each channel quad in a
    // fixed order (deterministic), one partial row per workgroup
    float* cred = reinterpret_cast<float*>(smem_raw);   // [256][4]
    __syncthreads();
    *reinterpret_cast<f32x4*>(cred + tid * 4) = csum;
    __syncthreads();
    if (tid < 32) {
      float sacc = 0.f;
      for (int j = 0; j < 32; ++j) sacc += cred[(8 * j + (tid >> 2)) * 4 + (tid & 3)];
      if (co0 + tid < a.Cout) a.colpart[(size_t)chunk * a.Cout + co0 + tid] = sacc;
    }
    __syncthreads();
  }
  if (KS == 1) {
    // the four waves split the rows: summed through LDS
    float* red = reinterpret_cast<float*>(smem_raw);   // [4][32*32]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) red[wv * 1024 + ((j & 3) + 8 * (j >> 2) + 4 * h) * 32 + r] = acc[0][j];
    __syncthreads();
#pragma unroll
    for (int q = tid; q < 1024; q += 256) {
      const float sum = (red[q] + red[1024 + q]) + (red[2048 + q] + red[3072 + q]);
      const int ci = ci0 + (q >> 5), co = co0 + (q & 31);
      if (ci < a.Cin && co < a.Cout) pout[((size_t)ci) * a.Cout + co] = sum;
    }
  } else {
    // every wave owns its taps: accumulator register j of lane (r, h) is element (ci = row(j, h), co = r) -- a wave
    // store covers two rows of 32 consecutive output channels
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int tap = wv + 4 * i;
      if (tap < NTAPS) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int ci = ci0 + (j & 3) + 8 * (j >> 2) + 4 * h, co = co0 + r;
          if (ci < a.Cin && co < a.Cout) pout[((size_t)tap * a.Cin + ci) * a.Cout + co] = acc[i][j];
        }
      }
    }
  }
}

struct WBVar {
  int KS;
  size_t lds;
};
bool pick(int KS, WBVar* v) {
  v->KS = KS;
  if (KS == 3) { v->lds = WBCfg<3>::LDS_BYTES; return true; }
  if (KS == 5) { v->lds = WBCfg<5>::LDS_BYTES; return true; }
  if (KS == 1) { v->lds = WBCfg<1>::LDS_BYTES; return true; }
  return false;
}
void chunking(const WBVar& v, int B, int H, int W, int Cin, int Cout, int* nTiles, int* tpc, int* nch, int* gy) {
  const int tilesX = cdiv(W, 16), tilesY = cdiv(H, 16);
  *nTiles = B * tilesX * tilesY;
  *gy = cdiv(Cin, 32) * cdiv(Cout, 32);
  // one round of resident workgroups (two per CU; one for 5x5) -- the slab count stays small for the finish launch
  int want = dg_cu_count() * (v.KS == 5 ? 1 : 2) / *gy;
  if (want < 1) want = 1;
  if (want > *nTiles) want = *nTiles;
  *tpc = cdiv(*nTiles, want);
  *nch = cdiv(*nTiles, *tpc);
}

template <int KS>
int launch(WgradArgs a, int nch, int gy, hipStream_t st) {
  constexpr size_t lds = WBCfg<KS>::LDS_BYTES;
  static DgOncePerDevice once;
  if (once.need())
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<KS>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((wgrad_bf16_kernel<KS>), dim3(nch, gy), dim3(256), lds, st, a);
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
  *nchunks_out = nch;
  if (KS == 3) return launch<3>(a, nch, gy, st);
  if (KS == 5) return launch<5>(a, nch, gy, st);
  return launch<1>(a, nch, gy, st);
}
