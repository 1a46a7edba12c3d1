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
//  * The pixel tile (32 pixels x Cin) is staged once per workgroup through LDS, double buffered, one barrier per tile.
//  * The K axis is walked in the order channel(s, h) = 8 (s / 4) + 4 h + s % 4 (s: k-step, h = lane / 32, the half of
//    the 32x32x2 MFMA's K pair): a lane's four consecutive k-steps are four consecutive channels, i.e. one 16-byte LDS
//    read feeds four MFMA steps, and the staging writes are 16-byte writes of what the global load returned.
//  * The affine epilogue is folded into the contraction (scaled weights, constant term as the C operand of a tile's
//    first MFMAs); what remains is the ReLU.
//  * A finished tile goes quads -> wave-private LDS blocks (16-byte writes straight from the accumulator registers)
//    and is read back as pieces of 8 pixels x 32 channels = 8 whole 128-byte lines (direct quad stores cover 32 lines
//    x 32 B per instruction, measured slower in the general kernel).  No workgroup barrier in the epilogue.
//  * Everything except the MFMAs and that write -- the next tile's global loads and staging, the B-operand reads, the
//    previous tile's piece reads, ReLU and stores -- is placed slot by slot INSIDE the MFMA stream (see the kernel).
#include "common.h"
#include "deconv_fwd.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

// DECONV_STAMPS (tools/micro/deconv_stamps.hip only): s_memtime at five points of a workgroup's tile number stamp_it (0: top,
// 4 / 5: before / after the tile barrier, 2: after the last MFMA, 3: after the LDS-block writes) and at the top of
// the next one (1)
#ifdef DECONV_STAMPS
#define DSTAMP(k) do { if (it == a.stamp_it) st_t[k] = __builtin_amdgcn_s_memtime(); \
                       if ((k) == 0 && it == a.stamp_it + 1) st_t[1] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DSTAMP(k) do { } while (0)
#endif

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>).  The pipeline below decides per
// MFMA slot which side operations ride in it; with the slot index a constant expression every such test is an
// `if constexpr` and every register-array index a literal (a `#pragma unroll` loop of this size is over the
// unroller's budget and leaves the arrays in scratch memory).
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <int CIN, int MT>
struct DCfg {
  static constexpr int NQ = CIN / 8;            // 16-byte K groups per lane
  static constexpr int ROW4 = 33;               // float4 slots per (q, h) row: 32 pixels + 1 pad
  static constexpr int BUF4 = 2 * NQ * ROW4;    // float4 slots of one staging buffer
  static constexpr int LD4 = CIN / 32;          // float4 loads per thread per tile (32 * CIN / 4 / 256)
  static constexpr int CP = 36;                 // floats per pixel row of an epilogue block
  static constexpr int ES = 32 * CP;            // floats of one [32 pixels][CP] epilogue block
  static constexpr int NWG = 128 * MT;          // channels of the N axis per workgroup
  // staging double buffer | per wave MT epilogue blocks | folded bias of the workgroup's channels
  static constexpr size_t LDS_BYTES = (size_t)2 * BUF4 * 16 + (size_t)4 * MT * ES * 4 + (size_t)NWG * 4;
};

// WIDE: the image rows are at least 32 pixels long, i.e. a 32-pixel tile lies in one row
template <int CIN, int MT, bool WIDE>
__global__ __launch_bounds__(256, 1) void deconv_fwd_kernel(DeconvArgs a) {
  using C = DCfg<CIN, MT>;
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  f32x4* stage = reinterpret_cast<f32x4*>(smem_raw);                    // [2][2 NQ][33]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* es = smem_raw + 2 * C::BUF4 * 4 + wv * (MT * C::ES);           // this wave's MT blocks of [32][CP]
  float* bt = smem_raw + 2 * C::BUF4 * 4 + 4 * MT * C::ES;              // [NWG] folded bias
  const int r = lane & 31, h = lane >> 5;
  const int nTiles = a.nTiles;
  // channel range of this wave on the N = 4 Cout axis
  const int nwg0 = blockIdx.y * C::NWG;
  const int n0 = nwg0 + wv * (32 * MT);

  // ---- the affine epilogue is folded into the contraction: (acc + b) s + t = sum (s w) x + (b s + t) ----
  // the weights are scaled once per workgroup as they enter the registers, the constant term starts the accumulators
  // (read from LDS straight into them) -- what is left after the MFMAs is the ReLU.
  for (int cidx = tid; cidx < C::NWG; cidx += 256) {
    const int co = (nwg0 + cidx) % a.Cout;
    const float s = a.scale ? a.scale[co] : 1.f;
    bt[cidx] = (a.bias ? a.bias[co] : 0.f) * s + (a.scale ? a.shift[co] : 0.f);
  }
  // wq[m][q] = s[n] W[n = n0 + 32 m + r][8 q + 4 h + 0..3]
  // (N axis = [tap][co], weight tensor (kh, kw, Cout, Cin): row n of the [4 Cout][Cin] matrix is a.w + n * CIN)
  f32x4 wq[MT][C::NQ];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int n = n0 + 32 * m + r;
    const float* wrow = a.w + (size_t)n * CIN + 4 * h;
    const float s = a.scale ? a.scale[n % a.Cout] : 1.f;
#pragma unroll
    for (int q = 0; q < C::NQ; ++q) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(wrow + 8 * q);
#pragma unroll
      for (int k = 0; k < 4; ++k) wq[m][q][k] = w4[k] * s;
    }
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
#pragma unroll
  for (int i = 0; i < C::LD4; ++i) stage[st_slot[i]] = in4[(size_t)tile * TILE4 + tid + 256 * i];
  __syncthreads();

  // epilogue constants of the lane: channel quad c4 of a 32-channel tile, pixel pl0 of an 8-pixel pass; per channel
  // tile the (wave-uniform) offset of its tap and first channel
  const int c4 = (lane & 7) * 4, pl0 = lane >> 3;
  const int lane_off = pl0 * 2 * (int)a.out.sX + c4;
  long tap_off[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int n = __builtin_amdgcn_readfirstlane(n0 + 32 * m);
    const int tap = n / a.Cout, co0 = n - tap * a.Cout;
    tap_off[m] = (long)(tap >> 1) * a.out.sY + (long)(tap & 1) * a.out.sX + co0;
  }
  // folded bias of the wave's channels in the accumulator layout (quad g of channel tile m: channels 8 g + 4 h + 0..3):
  // the C operand of every tile's first MFMA per channel tile -- starting the accumulators costs no instruction
  __syncthreads();
  f32x16 cb[MT];
  {
    const f32x4* bt4 = reinterpret_cast<const f32x4*>(bt + wv * (32 * MT)) + h;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = bt4[8 * m + 2 * g];
#pragma unroll
        for (int k = 0; k < 4; ++k) cb[m][4 * g + k] = b4[k];
      }
  }
  const float lo = a.relu ? 0.f : -__builtin_inff();

  // piece p = 4 m + k of a finished tile: the 8 pixels of pass k x the 32 channels of tile m = 8 whole 128-byte lines.
  // H and W are powers of two (the launcher checks): the pixel decomposition is shifts and masks on the scalar unit,
  // the offsets are 32-bit float offsets.
  auto piece_read = [&](int p) __attribute__((always_inline)) {
    return *reinterpret_cast<const f32x4*>(es + (p >> 2) * C::ES + (8 * (p & 3) + pl0) * C::CP + c4);
  };
  // float offset of output pixel (2 i, 2 j) for the first pixel of pass k of tile t_done: H and W are powers of two
  // (the launcher checks), so the pixel decomposition is shifts and masks on the scalar unit; 32-bit offsets
  auto pass_base = [&](int t_done, int k) __attribute__((always_inline)) {
    const unsigned P = (unsigned)t_done * 32u + 8u * (unsigned)k;
    const unsigned j0 = P & (unsigned)(a.W - 1), t = P >> a.lgW;
    const unsigned i0 = t & (unsigned)(a.H - 1), b0 = t >> a.lgH;
    return b0 * (unsigned)a.out.sB + i0 * (2u * (unsigned)a.out.sY) + j0 * (2u * (unsigned)a.out.sX);
  };
  auto piece_store = [&](int p, const unsigned (&base)[4], f32x4 v) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // ReLU (or nothing: lo = -inf); one v_max_f32 -- fmaxf() costs a second one that quiets a signalling NaN first
      float o;
      asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(v[k]), "v"(lo));
      v[k] = o;
    }
    *reinterpret_cast<f32x4*>(a.out.p + (size_t)base[p & 3] + tap_off[p >> 2] + lane_off) = v;
  };
  // ---- software pipeline, per wave ----
  // On this hardware nothing overlaps with a wave's fp32 MFMA stream except what is issued INSIDE it: another wave's
  // vector, LDS or store instructions do not issue while an older wave of the SIMD streams MFMAs (DESIGN.md section
  // 4), and instructions bunched between two streams cost their full latency (stamps, tools/micro/deconv_stamps.hip:
  // 7 070 cycles per tile with the epilogue behind the stream against 4 096 of MFMAs).  So the MFMA stream of tile t
  // carries, in the gaps between its MFMAs and a handful of issues per gap, everything else:
  //   gap 0 ..            global loads of tile t + 1 (registers; into LDS near the end of the stream)
  //   first gap of a K group   the B operand of K group q + BD (16-byte LDS read)
  //   from gap R0, every SR    LDS-block read of one piece of tile t - 1; ReLU and its store three gaps later
  //   gap S0 ..           tile t + 1 into the other staging buffer; the tile barrier before the last K group; the
  //                       first BD K groups of tile t + 1's B operand in the last gaps
  // and what follows the stream is only the transposing write of the accumulators into the wave's LDS blocks (4 MT
  // 16-byte writes straight from the accumulator registers), read back piece by piece inside the next stream.
  // sched_barrier after every MFMA pins this placement.
  constexpr int G4 = 4 * MT;              // MFMAs per K group
  constexpr int NM = C::NQ * G4;          // MFMAs per tile and wave
  constexpr int NW = 4 * MT;              // accumulator quads = pieces per tile and wave
  constexpr int BD = 4;                   // B-operand reads run this many K groups ahead
  constexpr int S0 = NM - G4 - C::LD4 - 2;
  constexpr int R0 = 4;
  constexpr int SR = (S0 - R0) / NW;
  static_assert(SR >= 4 && C::NQ > BD, "pipeline slots");

  f32x4 bfirst[BD];
  int cur = 0, done = -1;           // done: tile whose result sits in the wave's LDS blocks, not yet stored
#pragma unroll
  for (int q = 0; q < BD; ++q) bfirst[q] = stage[h * C::ROW4 + r + 2 * q * C::ROW4];

#ifdef DECONV_STAMPS
  unsigned long long st_t[6] = {0, 0, 0, 0, 0, 0};
  int it = 0;
#endif
  auto body = [&](auto have_c) __attribute__((always_inline)) {
    constexpr bool HAVE = decltype(have_c)::value;
    DSTAMP(0);
    const int nxt = tile + gridDim.x;
    const int nld = nxt < nTiles ? nxt : tile;   // the last tile re-reads itself: no branch around the loads
    const f32x4* sb = stage + cur * C::BUF4 + h * C::ROW4 + r;
    const f32x4* sn = stage + (cur ^ 1) * C::BUF4 + h * C::ROW4 + r;
    f32x16 acc[MT];
    f32x4 bq[C::NQ];
#pragma unroll
    for (int q = 0; q < BD; ++q) bq[q] = bfirst[q];
    f32x4 pre[C::LD4];
    f32x4 pv[2];
    pv[0] = pv[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned base[4] = {0u, 0u, 0u, 0u};        // pass origins of the finished tile (scalar registers)
    static_for<NM>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr int q = i / G4, wi = i % G4, j = wi / MT, m = wi % MT;
      if constexpr (i < C::LD4) pre[i] = in4[(size_t)nld * TILE4 + tid + 256 * i];
      if constexpr (wi == 0 && q + BD < C::NQ) bq[q + BD] = sb[2 * (q + BD) * C::ROW4];
      if constexpr (HAVE && i < 4) {
        if constexpr (i == 0) base[0] = pass_base(done, 0);
        else if constexpr (WIDE) base[i] = base[0] + (unsigned)(16 * i) * (unsigned)a.out.sX;   // passes 8 pixels apart
        else base[i] = pass_base(done, i);
      }
      if constexpr (HAVE && i >= R0 && i < R0 + NW * SR) {
        constexpr int p = (i - R0) / SR, ph = (i - R0) % SR;
        if constexpr (ph == 0) pv[p & 1] = piece_read(p);
        if constexpr (ph == 3) piece_store(p, base, pv[p & 1]);
      }
      if constexpr (i >= S0 && i < S0 + C::LD4) stage[(cur ^ 1) * C::BUF4 + st_slot[i - S0]] = pre[i - S0];
      if constexpr (i == NM - G4) {
        DSTAMP(4);
        __syncthreads();
        DSTAMP(5);
      }
      if constexpr (i > NM - G4 && i <= NM - G4 + BD) {
        constexpr int k = i - (NM - G4) - 1;
        bfirst[k] = sn[2 * k * C::ROW4];
      }
      if constexpr (q == 0 && j == 0)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[m][q][j], bq[q][j], cb[m], 0, 0, 0);
      else
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[m][q][j], bq[q][j], acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
    DSTAMP(2);
    // the finished tile into the wave's LDS blocks: lane (r, h) holds channels 8 g + 4 h + 0..3 of pixel r in
    // acc[m][4g .. 4g+3].  The same wave writes and reads its blocks; LDS executes a wave's accesses in order.
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 q4;
#pragma unroll
        for (int k = 0; k < 4; ++k) q4[k] = acc[m][4 * g + k];
        *reinterpret_cast<f32x4*>(es + m * C::ES + r * C::CP + 8 * g + 4 * h) = q4;
      }
    __builtin_amdgcn_sched_barrier(0);
    DSTAMP(3);
    done = tile;
    cur ^= 1;
#ifdef DECONV_STAMPS
    ++it;
#endif
  };

  body(std::false_type{});
  for (tile += gridDim.x; tile < nTiles; tile += gridDim.x) body(std::true_type{});
  // the last tile's stores
  {
    unsigned base[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) base[k] = pass_base(done, k);
#pragma unroll
    for (int p = 0; p < NW; ++p) piece_store(p, base, piece_read(p));
  }
#ifdef DECONV_STAMPS
  if (lane == 0 && a.stamps)
    for (int k = 0; k < 6; ++k) a.stamps[(blockIdx.x * 4 + wv) * 8 + k] = st_t[k];
#endif
}

template <int CIN, int MT, bool WIDE>
int launch_w(const DeconvArgs& a, int ny, hipStream_t st) {
  using C = DCfg<CIN, MT>;
  static int per_cu_dev[DG_MAX_DEVICES] = {};   // occupancy and the LDS attribute belong to the device
  const int slot = dg_device_slot();
  int per_cu = slot >= 0 ? per_cu_dev[slot] : 0;
  if (!per_cu) {
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&deconv_fwd_kernel<CIN, MT, WIDE>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    int occ = 0;
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, deconv_fwd_kernel<CIN, MT, WIDE>, 256, C::LDS_BYTES));
    if (occ < 1) occ = 1;
    if (const char* e = getenv("DEPGAN_DECONV_PER_CU")) {
      const int v = atoi(e);
      if (v >= 1 && v < occ) occ = v;
    }
    per_cu = occ;
    if (slot >= 0) per_cu_dev[slot] = occ;
  }
  const int cus = dg_cu_count();
  int gx = cus * per_cu / ny;
  if (gx > a.nTiles) gx = a.nTiles;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((deconv_fwd_kernel<CIN, MT, WIDE>), dim3(gx, ny), dim3(256), C::LDS_BYTES, st, a);
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

template <int CIN, int MT>
int launch(const DeconvArgs& a, int ny, hipStream_t st) {
  return a.W >= 32 ? launch_w<CIN, MT, true>(a, ny, st) : launch_w<CIN, MT, false>(a, ny, st);
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
  // power-of-two image sizes of at least 8 pixels (a pass of 8 pixels never crosses an image row, pixel -> (b, i, j) is
  // shifts and masks), whole 32-pixel tiles, the output addressable with 32-bit float offsets
  if (W < 8 || H < 1 || (W & (W - 1)) || (H & (H - 1)) || ((long)B * H * W) % 32) return false;
  if ((long)B * H * W / 32 > 0x3FFFFFF || (long)B * out.sB > 0x7FFFFFFFL) return false;
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
