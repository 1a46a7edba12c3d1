// Forward of the 2x2 / stride-2 transposed convolution (Conv2DTranspose of GT:308, used at GT:449/464/478) with all
// four taps in ONE workgroup, gfx950.
//
//   out[b][2i+di][2j+dj][co] = act( (sum_ci in[b][i][j][ci] W[di][dj][co][ci] + bias[co]) * scale[co] + shift[co] )
//
// As a GEMM this is [P = B H W pixels] x [K = Cin] x [N = 4 Cout] with K of only 64 .. 128: 26 flop per byte of
// (input once + output once) at 64 -> 64, below the 31 flop/B where fp32 MFMA (157 TFLOP/s) meets HBM (5 TB/s
// achievable) -- the layer belongs to the HBM side of the roofline, and the grouped launch of the general kernel
// (igemm<32,1,32,1>: one 256-pixel x 32-channel item with 64 MFMAs per wave between a staging prologue and an
// epilogue built for 288-MFMA items) ran it at 0.37 of the MFMA peak.
//
// Design:
//  * Persistent workgroups of 4 waves.  A wave owns 32 MT consecutive channels of the N axis (ordered [di][dj][co],
//    so a 32-channel tile never straddles a tap) and keeps their weights for the WHOLE K axis in registers
//    (MT * Cin / 2 VGPRs, loaded once per workgroup): the main loop issues no weight traffic at all.
//  * The MFMA takes the weight fragment as its first operand (D[channel][pixel]), so a lane ends up with quads of
//    four consecutive channels of one pixel.
//  * The pixel tile (32 pixels x Cin) is staged once per workgroup through LDS, double buffered, one barrier per tile;
//    the next tile's global loads are issued before the MFMA loop and land in LDS after the epilogue.
//  * The K axis is walked in the order channel(s, h) = 8 (s / 4) + 4 h + s % 4 (s: k-step, h = lane / 32, the half of
//    the 32x32x2 MFMA's K pair): a lane's four consecutive k-steps are four consecutive channels, i.e. one 16-byte LDS
//    read feeds four MFMA steps, and the staging writes are 16-byte writes of what the global load returned.
//  * Epilogue per 32-channel tile: quads -> wave-private LDS block -> rows of 8 lanes x 16 B = one pixel's 32
//    channels = one full 128-byte line per pixel (direct quad stores cover 32 lines x 32 B per instruction, measured
//    slower in the general kernel), bias / BN affine / ReLU in the same pass.  No workgroup barrier in the epilogue.
#include "common.h"
#include "deconv_fwd.h"
#include <stdlib.h>

namespace {

template <int CIN, int MT>
struct DCfg {
  static constexpr int NQ = CIN / 8;            // 16-byte K groups per lane
  static constexpr int ROW4 = 33;               // float4 slots per (q, h) row: 32 pixels + 1 pad
  static constexpr int BUF4 = 2 * NQ * ROW4;    // float4 slots of one staging buffer
  static constexpr int LD4 = CIN / 32;          // float4 loads per thread per tile (32 * CIN / 4 / 256)
  static constexpr int CP = 36;                 // floats per pixel row of the epilogue block
  static constexpr size_t LDS_BYTES = (size_t)2 * BUF4 * 16 + (size_t)4 * 32 * CP * 4;
};

template <int CIN, int MT>
__global__ __launch_bounds__(256, 2) void deconv_fwd_kernel(DeconvArgs a) {
  using C = DCfg<CIN, MT>;
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  f32x4* stage = reinterpret_cast<f32x4*>(smem_raw);                    // [2][2 NQ][33]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* es = smem_raw + 2 * C::BUF4 * 4 + wv * (32 * C::CP);           // this wave's [32][CP] block
  const int r = lane & 31, h = lane >> 5;
  const int nTiles = a.nTiles;
  // channel range of this wave on the N = 4 Cout axis
  const int n0 = (blockIdx.y * 4 + wv) * (32 * MT);

  // ---- weights of the wave's channels, whole K, into registers: wq[m][q] = W[n0 + 32 m + r][8 q + 4 h + 0..3] ----
  // (N axis = [tap][co], weight tensor (kh, kw, Cout, Cin): row n of the [4 Cout][Cin] matrix is a.w + n * CIN)
  f32x4 wq[MT][C::NQ];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float* wrow = a.w + (size_t)(n0 + 32 * m + r) * CIN + 4 * h;
#pragma unroll
    for (int q = 0; q < C::NQ; ++q) wq[m][q] = *reinterpret_cast<const f32x4*>(wrow + 8 * q);
  }

  // ---- staging: thread t fetches float4 f = t + 256 i of the tile's [32][CIN] block: pixel f / (CIN/4), chunk j ----
  // chunk j = channels 4j .. 4j+3 = K group q = j / 2, half h = j % 2 -> slot (2q + h) * 33 + pixel = j * 33 + pixel
  int st_slot[C::LD4];
#pragma unroll
  for (int i = 0; i < C::LD4; ++i) {
    const int f = tid + 256 * i;
    st_slot[i] = (f % (CIN / 4)) * C::ROW4 + f / (CIN / 4);
  }
  const f32x4* in4 = reinterpret_cast<const f32x4*>(a.in);
  constexpr int TILE4 = 32 * CIN / 4;

  int tile = blockIdx.x;
  if (tile < nTiles) {
#pragma unroll
    for (int i = 0; i < C::LD4; ++i) stage[st_slot[i]] = in4[(size_t)tile * TILE4 + tid + 256 * i];
  }
  __syncthreads();

  // epilogue constants of the lane: channel quad c4 of a 32-channel tile, pixel pl0 of an 8-pixel pass
  const int c4 = (lane & 7) * 4, pl0 = lane >> 3;
  const long lane_off = (long)pl0 * 2 * a.out.sX;
  int cur = 0;
  for (; tile < nTiles; tile += gridDim.x) {
    // next tile's loads first: they fly under the MFMA loop and the epilogue
    const int nxt = tile + gridDim.x;
    f32x4 pre[C::LD4];
    if (nxt < nTiles) {
#pragma unroll
      for (int i = 0; i < C::LD4; ++i) pre[i] = in4[(size_t)nxt * TILE4 + tid + 256 * i];
    }

    // ---- contraction: acc[m] (32 channels x 32 pixels) over the whole K ----
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[m][k] = 0.f;
    const f32x4* sb = stage + cur * C::BUF4 + h * C::ROW4 + r;
    f32x4 bq = sb[0];
#pragma unroll
    for (int q = 0; q < C::NQ; ++q) {
      f32x4 bn = bq;
      if (q + 1 < C::NQ) bn = sb[(2 * (q + 1)) * C::ROW4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[m][q][j], bq[j], acc[m], 0, 0, 0);
      bq = bn;
    }

    // ---- epilogue ----
    // pixel rows of the four 8-pixel passes (W % 8 == 0: a pass never crosses an image row)
    long pass_off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned P = (unsigned)tile * 32u + 8u * k;   // wave-uniform
      unsigned j0, i0, b0;
      if (a.lgW >= 0 && a.lgH >= 0) {
        j0 = P & (a.W - 1);
        const unsigned t = P >> a.lgW;
        i0 = t & (a.H - 1);
        b0 = t >> a.lgH;
      } else {
        j0 = P % (unsigned)a.W;
        const unsigned t = P / (unsigned)a.W;
        i0 = t % (unsigned)a.H;
        b0 = t / (unsigned)a.H;
      }
      pass_off[k] = (long)b0 * a.out.sB + (long)(2 * i0) * a.out.sY + (long)(2 * j0) * a.out.sX;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int n = __builtin_amdgcn_readfirstlane(n0 + 32 * m);
      const int tap = n / a.Cout, co0 = n - tap * a.Cout;
      const long tap_off = (long)(tap >> 1) * a.out.sY + (long)(tap & 1) * a.out.sX + co0 + c4;
      f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + co0 + c4);
      if (a.scale) {
        sc4 = *reinterpret_cast<const f32x4*>(a.scale + co0 + c4);
        sh4 = *reinterpret_cast<const f32x4*>(a.shift + co0 + c4);
      }
      // lane (r, h) holds channels 8 g + 4 h + 0..3 of pixel r in acc[m][4g .. 4g+3]
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 q4;
#pragma unroll
        for (int k = 0; k < 4; ++k) q4[k] = acc[m][4 * g + k];
        *reinterpret_cast<f32x4*>(es + r * C::CP + 8 * g + 4 * h) = q4;
      }
      // same wave writes and reads the block: LDS executes a wave's accesses in order, the fences keep the compiler
      // from moving the reads above the writes (and the next tile's writes above these reads)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      f32x4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(es + (8 * k + pl0) * C::CP + c4);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = v[k][e] + bias4[e];
          if (a.scale) x = __fadd_rn(__fmul_rn(x, sc4[e]), sh4[e]);
          if (a.relu) x = fmaxf(x, 0.f);
          v[k][e] = x;
        }
        *reinterpret_cast<f32x4*>(a.out.p + pass_off[k] + tap_off + lane_off) = v[k];
      }
    }

    // ---- next tile into the other staging buffer (its readers finished before the previous barrier) ----
    if (nxt < nTiles) {
#pragma unroll
      for (int i = 0; i < C::LD4; ++i) stage[(cur ^ 1) * C::BUF4 + st_slot[i]] = pre[i];
    }
    __syncthreads();
    cur ^= 1;
  }
}

template <int CIN, int MT>
int launch(const DeconvArgs& a, int ny, hipStream_t st) {
  using C = DCfg<CIN, MT>;
  static int per_cu = 0;
  if (!per_cu) {
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&deconv_fwd_kernel<CIN, MT>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    int occ = 0;
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, deconv_fwd_kernel<CIN, MT>, 256, C::LDS_BYTES));
    if (occ < 1) occ = 1;
    if (const char* e = getenv("DEPGAN_DECONV_PER_CU")) {
      const int v = atoi(e);
      if (v >= 1 && v < occ) occ = v;
    }
    per_cu = occ;
  }
  int cus = 256;
  {
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      hipDeviceProp_t p;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ncu = p.multiProcessorCount;
      if (ncu < 1) ncu = 256;
    }
    cus = ncu;
  }
  int gx = cus * per_cu / ny;
  if (gx > a.nTiles) gx = a.nTiles;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((deconv_fwd_kernel<CIN, MT>), dim3(gx, ny), dim3(256), C::LDS_BYTES, st, a);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

int mt_for(int Cin) { return Cin == 96 ? 3 : 2; }
int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

}  // namespace

bool dg_deconv_fwd_supported(int B, int H, int W, int Cin, int Cout, TView in, TView out) {
  if (const char* e = getenv("DEPGAN_DECONV_FUSED"))
    if (atoi(e) == 0) return false;
  if (Cin != 64 && Cin != 96 && Cin != 128) return false;
  if (Cout % 32) return false;
  if ((4 * Cout) % (128 * mt_for(Cin))) return false;
  if (W % 8 || ((long)B * H * W) % 32) return false;
  if ((long)B * H * W / 32 > 0x3FFFFFF) return false;
  // dense NHWC input (the staging reads it as one flat [pixels][Cin] matrix), 16-byte aligned views
  if (in.sX != Cin || in.sY != (long)W * Cin || in.sB != (long)H * W * Cin) return false;
  if (((uintptr_t)in.p | (uintptr_t)out.p) & 15) return false;
  if ((out.sX | out.sY | out.sB) & 3) return false;
  return true;
}

int dg_deconv_fwd(DeconvArgs a, int B, hipStream_t st) {
  if (!dg_deconv_fwd_supported(B, a.H, a.W, a.Cin, a.Cout, make_view(const_cast<float*>(a.in), a.H, a.W, a.Cin), a.out)) {
    dg_set_error("dg_deconv_fwd: shape %dx%dx%d %d->%d not covered by the fused transposed-convolution kernel", B, a.H,
                 a.W, a.Cin, a.Cout);
    return DG_ERR_UNSUPPORTED;
  }
  if (((uintptr_t)a.w & 15) || (a.bias && ((uintptr_t)a.bias & 15)) ||
      (a.scale && (((uintptr_t)a.scale | (uintptr_t)a.shift) & 15 || !a.shift))) {
    dg_set_error("dg_deconv_fwd: weights / bias / scale / shift must be 16-byte aligned (scale needs shift)");
    return DG_ERR_ARG;
  }
  a.nTiles = (int)((long)B * a.H * a.W / 32);
  a.lgW = ilog2_exact(a.W);
  a.lgH = ilog2_exact(a.H);
  const int MT = mt_for(a.Cin);
  const int ny = 4 * a.Cout / (128 * MT);
  if (a.Cin == 64) return launch<64, 2>(a, ny, st);
  if (a.Cin == 96) return launch<96, 3>(a, ny, st);
  return launch<128, 2>(a, ny, st);
}
