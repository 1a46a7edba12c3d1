// 3x3 NHWC fp32 convolution as Winograd F(2x2, 3x3) on the CDNA4 fp32 matrix cores: 16 multiplications per 2x2 output
// block and input channel instead of 36 -- the contraction of igemm_conv_kernel<32,3,8,9> with 4/9 of its MFMAs.
//
// Why: on this machine an fp32 MFMA stream is the vector ALU (nothing a SIMD issues overlaps with
// it), igemm_conv_kernel sits at 0.68 ... 0.84 of the fp32 matrix peak on the 3x3 layers and its ablations
// (profiles/r03_conv_experiments.md) put 0.82 as the ceiling of ANY direct form.  The remaining lever is the number of
// MFMAs.  Y = A^T [ (G g G^T) . (B^T d B) ] A with the usual F(2x2,3x3) matrices: B^T and A^T hold only 0 and +-1 (the
// input and output transforms are additions), G holds 1 and 1/2 (the weight transform is done once per weight update by
// the packing kernel, in double, rounded once).  Same operands, same fp32 accumulation over Cin in the same chunk order;
// what changes is the summation tree (measured against an fp64 convolution: 1.7 x the rounding error of the direct
// kernel, both ~1e-7 of the output's range -- tests/test_gpu_ops.py).
//
// Mapping (one workgroup = 256 threads = 4 waves; item = 8 MT x 16 output pixels of one sample x 32 output channels;
// MT = 1: 8-row tiles, 64 accumulator registers per wave, three workgroups per CU -- the default; MT = 2: 16-row tiles,
// 128 registers, two per CU -- the form that carries the fused one-channel head):
//   * the tile is 4 MT x 8 Winograd tiles (2 x 2 outputs each); the 16 "frequencies" (a, b) of the transform domain are
//     16 independent GEMMs  M_f[tile][n] = sum_c V_f[tile][c] U_f[c][n]  (32 MT x Cin x 32);
//   * wave w owns the frequencies a = w (b = 0..3): 4 frequencies x MT row tiles of v_mfma_f32_32x32x2_f32.  Its
//     weight fragments U_f are nobody else's: from the packed panel [nt][chunk][f][n][8] straight into registers, never
//     through LDS.  And it transforms exactly the V rows it multiplies: task = (tile, 4 channels) for row a = w, two
//     halo rows x four columns in, four 16-byte rows of V out -- so V is wave-private and needs no barrier, only the LDS
//     unit's in-order execution of the wave's own writes and reads;
//   * per chunk of 8 input channels: the raw (8 MT + 2) x 18 x 8 halo is copied global -> LDS by the DMA path
//     (buffer_load_dwordx4 ... lds: no registers, no ds_write pass, asynchronous; double-buffered, issued a whole chunk
//     ahead; its completion is the ONE barrier of the chunk), transform (16 packed additions per task), 16 MT MFMAs per
//     wave with one b128 A-fragment read per four of them.  The raw image and the V planes are XOR-swizzled (rslot,
//     vslot): the LDS serves a b128 access eight lanes at a time out of 128 bytes of banks;
//   * epilogue: each wave reduces its four b's to the two output columns in registers (Z[a][q] = row transform), the
//     waves exchange Z through LDS, and the fused epilogue of igemm_conv (igemm_epilogue.inc, same text) fetches
//     v = Z[0] + Z[1] + Z[2] (even rows) or Z[1] - Z[2] - Z[3] (odd rows) where it used to fetch one transposed value.
// Covers KS = 3, stride 1, 'same' padding, Cin % 8 == 0, Cout % 32 == 0, even H and W; gathered K (ConvArgs::cpt) as in
// igemm_conv; no groups.  Everything else stays on igemm_conv_kernel.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "epilogue.h"

namespace {

constexpr int WN_CK = 8;                         // channels per chunk (one b128 fragment read = 4 MFMAs of 2 channels)
constexpr int WN_TW = 18;                       // halo columns of a 16-pixel-wide tile
constexpr int WN_CP = 36;                       // floats per tile row of the Z exchange (32 channels + 4)
constexpr int WN_XV = WN_CK / 4;

// MT = MFMA row tiles (32 Winograd tiles = 8 x 16 output pixels each) per wave: the workgroup's pixel tile is 8 MT rows
// x 16 columns.  MT = 2: 128 accumulator registers per wave, two workgroups per CU.  MT = 1: 64, three per CU -- more
// waves per SIMD to keep the matrix pipe busy (one wave reaches 0.60 of it, two 0.72, four 0.84:
// profiles/r03_conv_experiments.md) at twice the weight-fragment traffic per MFMA and a larger halo share.
template <int MT>
struct WnCfg {
  static constexpr int TH = 8 * MT;                     // output rows per workgroup
  static constexpr int NTILE = 32 * MT;                 // Winograd tiles per workgroup
  static constexpr int PIXT = (TH + 2) * WN_TW;         // raw halo pixels
  // The raw halo chunk is copied global -> LDS by the DMA path (buffer_load_dwordx4 ... lds: no VGPR destination, no
  // ds_write pass, asynchronous): a wave-instruction writes 64 lanes x 16 B lane-linearly, so the LDS image is the
  // unpadded [pixel][8 channels] one, in whole rounds of 256 pieces.  Which piece a lane FETCHES is free, and that is
  // where the bank swizzle goes (rslot): logical piece (pixel, half) lives in slot (2 pixel + half) ^ (bit 2 of pixel
  // << 1), so the transform's reads -- eight lanes = four tiles x two channel halves, 64 B apart -- hit eight different
  // 16-byte slots of the 128-byte bank window.
  static constexpr int XTOT = PIXT * WN_XV;             // 16-byte pieces of a raw chunk
  static constexpr int NXP = (XTOT + 255) / 256;        // DMA instructions per wave and chunk
  static constexpr int RAW = NXP * 256 * 4;             // floats
  // one frequency: tiles x 8 channels, unpadded; its 16-byte slots are XOR-swizzled (vslot below): the LDS serves a
  // b128 access eight lanes at a time out of 128 bytes of banks, so eight consecutive lanes must hit eight different
  // slots modulo 8 -- lanes r and r + 4 of a fragment read (32-byte rows) and the four a's of a transform write
  // (whole planes apart) would not (counters: half of the LDS cycles of the unswizzled form were bank conflicts)
  static constexpr int VPLANE = NTILE * WN_CK;
  static constexpr int V = 16 * VPLANE;
  static constexpr int ZPLANE = NTILE * WN_CP + 32;     // one (a, q): +128 B so that q = 0 / 1 of a pixel pair differ in bank group
  static constexpr int Z = 8 * ZPLANE;
  // two raw buffers (one barrier per chunk: a wave may stage chunk c + 1 while another still transforms chunk c)
  static constexpr size_t LDS = sizeof(float) * (size_t)((2 * RAW + V) > Z ? (2 * RAW + V) : Z);
  static constexpr int WGS_PER_CU = (MT == 2) ? 2 : 3;
};

// ABL: ablation bits for tools/time_wino.py (0 = the product kernel; the others are compiled with
// -DDEPGAN_WINO_ABLATIONS only): 1 no epilogue, 2 no input transform, 4 no MFMAs, 8 no raw staging
// 16-byte slot of the raw LDS image that holds logical piece (pixel, channel half); an involution on slot indices
static __device__ __forceinline__ int rslot(int pix, int half) { return (2 * pix + half) ^ (((pix >> 2) & 1) << 1); }

// float offset inside a V plane of 16-byte slot (tile T, channel half cg) of the frequencies of transform row a
static __device__ __forceinline__ int vslot(int T, int cg, int a) { return ((2 * T + cg) ^ ((T >> 2) & 1) ^ (a << 1)) << 2; }

template <int MT, bool PERS, bool HEAD, int ABL = 0>
static __device__ __forceinline__ void wino_body(const ConvArgs& a) {
  typedef WnCfg<MT> C;
  constexpr int MF = 32, NT = 32;
  constexpr int WN_RAW = C::RAW, WN_VPLANE = C::VPLANE, WN_ZPLANE = C::ZPLANE, WN_XTOT = C::XTOT, NXP = C::NXP;
  typedef f32x16 acc_t;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // raw halo buffers [(8 MT + 2) x 18][12] x 2 and V [16][WN_VPLANE]; the Z planes of the epilogue alias all three
  float* const raw0 = smem;
  float* const raw1 = smem + WN_RAW;
  float* V = smem + 2 * WN_RAW;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int tilesX = (a.W + 15) >> 4, tilesY = (a.H + C::TH - 1) / C::TH;
  const unsigned nNTall = (unsigned)a.lgy, nPix = (unsigned)a.lgx;
  const int nCC = a.Cin / WN_CK;

  // ---- per-thread geometry, once per workgroup ----
  // raw staging: DMA piece i of this thread fills LDS slot q = tid + 256 i with logical piece rslot^-1(q) = rslot of it
  // (an involution): byte offset from the halo origin, or an out-of-range offset for the padding slots behind the halo
  // tile (the hardware returns zeros for those, as for out-of-image pixels)
  constexpr int SENT = (int)0x80000000;
  int xvo[NXP], xyx[NXP];
#pragma unroll
  for (int i = 0; i < NXP; ++i) {
    const int q = tid + i * 256;
    const int lq = q ^ (((q >> 3) & 1) << 1);
    const int pix = lq >> 1, part = lq & 1;
    const int ly = pix / WN_TW, lx = pix - ly * WN_TW;
    xvo[i] = (q < WN_XTOT) ? 4 * (ly * (int)a.in.sY + lx * (int)a.in.sX + part * 4) : SENT;
    xyx[i] = (ly << 8) | lx;
  }
  // transform task i of a lane: q = lane + 64 i -> channel group q & 1, tile q >> 1, transform row a = wave.
  // Row a of B^T d needs two of the tile's four halo rows:  a = 0: d0 - d2,  1: d1 + d2,  2: d2 - d1,  3: d1 - d3
  // = x + s y with (x, y) = rows (0,2) (1,2) (2,1) (1,3) and s = +1 for a = 1, else -1.
  int tA[MT][4], tB[MT][4], tV[MT];
  float tS[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    // a wave transforms exactly the frequencies it multiplies (row a = wave): V needs no barrier between the transform
    // and the MFMAs, only the LDS unit's in-order execution of the wave's own writes and reads
    const int q = lane + 64 * i;
    const int cg = q & 1, aa = wv, T = q >> 1;
    const int tyi = T >> 3, txi = T & 7;
    const int rA = (aa == 0) ? 0 : (aa == 2 ? 2 : 1), rB = (aa == 3) ? 3 : (aa == 2 ? 1 : 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tA[i][j] = 4 * rslot((2 * tyi + rA) * WN_TW + 2 * txi + j, cg);
      tB[i][j] = 4 * rslot((2 * tyi + rB) * WN_TW + 2 * txi + j, cg);
    }
    tV[i] = (4 * aa) * WN_VPLANE + vslot(T, cg, aa);
    tS[i] = (aa == 1) ? 1.f : -1.f;
  }
  const int wbase = __builtin_amdgcn_readfirstlane(wv * 256);   // this wave's float offset inside a 256-piece round
  // fragments: A = V[f][32 mt + r][4h ..], B = panel[f][r][4h ..]
  int aoff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) aoff[mt] = vslot(32 * mt + r, h, wv);
  const int boff = (4 * wv * 32 + r) * WN_CK + 4 * h;

  unsigned id = blockIdx.x;
  do {
    int t, ntile;
    if ((nPix & 7u) == 0) {   // ids dealt over the 8 XCDs as in igemm_conv_kernel
      const unsigned x = id & 7u, sl = id >> 3;
      ntile = (int)(sl % nNTall);
      t = (int)(x * (nPix >> 3) + sl / nNTall);
    } else {
      t = (int)(id % nPix);
      ntile = (int)(id / nPix);
    }
    const int tx0 = (t % tilesX) * 16;
    t /= tilesX;
    const int ty0 = (t % tilesY) * C::TH;
    const int b = t / tilesY;
    const long out_goff = 0;
    const int n0 = ntile * NT;
    const float* inb = a.in.p + (long)b * a.in.sB;
    const bool interior = ty0 >= 1 && ty0 + C::TH + 1 <= a.H && tx0 >= 1 && tx0 + 17 <= a.W;
    const char* halo0 = reinterpret_cast<const char*>(inb + ((long)(ty0 - 1) * a.in.sY + (long)(tx0 - 1) * a.in.sX));
    auto coff = [&](int cc) -> long {
      if (a.cpt > 0) {
        const int run = cc / a.cpt;
        return a.in_run_off[run] + (long)(cc - run * a.cpt) * WN_CK;
      }
      return (long)cc * WN_CK;
    };
    // descriptor at the pixel one row and one column before the image origin: every in-image piece has a non-negative
    // offset from it
    const float* const xorg = a.in.p - ((long)a.in.sY + (long)a.in.sX);
    const unsigned long long xu = (unsigned long long)xorg;
    // (unsigned locals: readfirstlane returns int, and a low half with bit 31 set would sign-extend into the high one)
    const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xu), xhi = __builtin_amdgcn_readfirstlane((unsigned)(xu >> 32));
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)xhi << 32) | xlo), 0, 0x7FFFFFFF, 0x00020000);
    const int xso0 = 4 * (b * (int)a.in.sB + ty0 * (int)a.in.sY + tx0 * (int)a.in.sX);
    auto stage_dma = [&](int cc, int buf) {   // chunk cc -> raw buffer buf
      const int xso = xso0 + 4 * (int)coff(cc);
      float* xs = (buf ? raw1 : raw0) + wbase;
      if (interior) {
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
          // (a plain int local: with a type-dependent argument such as xvo[i] hipcc drops the host-side instantiation
          // of the kernel without a diagnostic)
          const int vo = xvo[i];
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(xs + i * 1024), 16, vo,
                                                   xso, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
          const int iy = ty0 + (xyx[i] >> 8) - 1, ix = tx0 + (xyx[i] & 255) - 1;
          const int vo = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? xvo[i] : SENT;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(xs + i * 1024), 16, vo,
                                                   xso, 0, 0);
        }
      }
    };

    // no zeroing pass: the first MFMA of every accumulator (chunk 0, j = 0) takes a constant-zero C operand -- 64 MT
    // vector moves per item less on a pipe where every vector instruction costs its issue time next to the MFMAs
    acc_t acc[4][MT];

    const float* wnt = a.w + (size_t)ntile * nCC * (16 * NT * WN_CK) + boff;
    if (!(ABL & 8)) stage_dma(0, 0);
    auto chunk = [&](const int cc, auto first_tag) {
      constexpr bool FIRST = decltype(first_tag)::value;
      // this wave's weight fragments of the chunk: four frequencies x (32 channels x 8) -- in flight during the transform
      f32x4 bq[4];
#pragma unroll
      for (int f = 0; f < 4; ++f)
        bq[f] = *reinterpret_cast<const f32x4*>(wnt + (size_t)cc * (16 * NT * WN_CK) + f * (NT * WN_CK));
      // this wave's DMA pieces of chunk cc (issued a whole chunk ago) and its weight fragments have landed ... (the
      // compiler does not know that the DMA writes LDS: the wait is explicit)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // ... and so have everybody else's -- the only barrier of the chunk
      // raw[(cc + 1) & 1] was last read by the transform of chunk cc - 1: every wave is past this chunk's barrier
      if (!(ABL & 8) && cc + 1 < nCC) stage_dma(cc + 1, (cc + 1) & 1);
      const float* raw = (cc & 1) ? raw1 : raw0;
      // ---- input transform: raw -> V ----
#pragma unroll
      for (int i = 0; i < ((ABL & 2) ? 0 : MT); ++i) {
        f32x4 tc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(raw + tA[i][j]);
          const f32x4 y = *reinterpret_cast<const f32x4*>(raw + tB[i][j]);
#pragma unroll
          for (int k = 0; k < 4; ++k) tc[j][k] = fmaf(y[k], tS[i], x[k]);   // exact: s = +-1
        }
        f32x4 v0, v1, v2, v3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v0[k] = tc[0][k] - tc[2][k];
          v1[k] = tc[1][k] + tc[2][k];
          v2[k] = tc[2][k] - tc[1][k];
          v3[k] = tc[1][k] - tc[3][k];
        }
        *reinterpret_cast<f32x4*>(V + tV[i]) = v0;
        *reinterpret_cast<f32x4*>(V + tV[i] + WN_VPLANE) = v1;
        *reinterpret_cast<f32x4*>(V + tV[i] + 2 * WN_VPLANE) = v2;
        *reinterpret_cast<f32x4*>(V + tV[i] + 3 * WN_VPLANE) = v3;
      }
      // ---- 16 GEMMs, this wave's four: 32 MFMAs (V rows of frequency row a = wave: written by this wave just above) ----
      f32x4 av[2][MT];
      auto load_a = [&](int f, f32x4* d) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          d[mt] = *reinterpret_cast<const f32x4*>(V + (4 * wv + f) * WN_VPLANE + aoff[mt]);
      };
      load_a(0, av[0]);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        if (f + 1 < 4) load_a(f + 1, av[(f + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            if (!(ABL & 4)) {
              if (FIRST && j == 0) {
                const acc_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[f][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[f][j], av[f & 1][mt][j], zero, 0, 0, 0);
              } else {
                acc[f][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[f][j], av[f & 1][mt][j], acc[f][mt], 0, 0, 0);
              }
            } else {
              if (FIRST && j == 0) acc[f][mt] = acc_t{};
              acc[f][mt][j] += bq[f][j] * av[f & 1][mt][j];
            }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    chunk(0, std::true_type{});
    for (int cc = 1; cc < nCC; ++cc) chunk(cc, std::false_type{});

    if (ABL & 1) {   // keep the accumulators alive with one store that never happens
      float sacc = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < 16; ++j) sacc += acc[f][mt][j];
      if (sacc == 12345.678f) a.out.p[tid] = sacc;
      __syncthreads();
      continue;
    }
    // ---- output transform, first half (this wave's row a = wv: the four b's -> the two output columns q) ----
    // run from inside the fused epilogue (EPI_STAGE_LATE): after its per-item constant loads have been issued
    auto zstage = [&]() {
      __syncthreads();   // every wave is done with V: the Z exchange may overwrite raw and V
      float* zw = smem + (2 * wv) * WN_ZPLANE + r * WN_CP + 4 * h;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 z0, z1;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float m0 = acc[0][mt][4 * g + k], m1 = acc[1][mt][4 * g + k], m2 = acc[2][mt][4 * g + k],
                        m3 = acc[3][mt][4 * g + k];
            z0[k] = (m0 + m1) + m2;
            z1[k] = (m1 - m2) - m3;
          }
          *reinterpret_cast<f32x4*>(zw + mt * (32 * WN_CP) + 8 * g) = z0;
          *reinterpret_cast<f32x4*>(zw + WN_ZPLANE + mt * (32 * WN_CP) + 8 * g) = z1;
        }
      __syncthreads();
    };
    // second half inside the fused epilogue: pixel (py, px) of the wave's 4 x 16 block is output (py & 1, px & 1) of
    // tile (2 wv + py / 2, px / 2); even rows Z[0] + Z[1] + Z[2], odd rows Z[1] - Z[2] - Z[3]
#define EPI_PRE_SYNC
#define EPI_STAGE
#define EPI_STAGE_LATE zstage()
#define EPI_NPASS (4 * MT)
#define EPI_OYW (ty0 + 2 * MT * __builtin_amdgcn_readfirstlane(wv))
#define EPI_FULL ((ty0 + C::TH <= a.H) && (tx0 + 16 <= a.W))
#define EPI_FETCH(v, py, px)                                                                                         \
  {                                                                                                                  \
    const float* zb = smem + ((px)&1) * WN_ZPLANE +                                                                  \
                      ((MT * __builtin_amdgcn_readfirstlane(wv) + ((py) >> 1)) * 8 + ((px) >> 1)) * WN_CP + c4;       \
    const f32x4 za = *reinterpret_cast<const f32x4*>(zb + (((py)&1) ? 2 : 0) * WN_ZPLANE);                           \
    const f32x4 zb1 = *reinterpret_cast<const f32x4*>(zb + (((py)&1) ? 4 : 2) * WN_ZPLANE);                          \
    const f32x4 zc = *reinterpret_cast<const f32x4*>(zb + (((py)&1) ? 6 : 4) * WN_ZPLANE);                           \
    _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) v[k_] = ((py)&1) ? (za[k_] - zb1[k_]) - zc[k_] : (za[k_] + zb1[k_]) + zc[k_]; \
  }
#define EPI_HEAD HEAD
#include "igemm_epilogue.inc"
#undef EPI_HEAD
#undef EPI_FETCH
#undef EPI_FULL
#undef EPI_OYW
#undef EPI_NPASS
#undef EPI_STAGE_LATE
#undef EPI_STAGE
#undef EPI_PRE_SYNC
    // the next item's first raw tile overwrites the Z planes (keeping raw buffer 0 apart from them -- 46.5 KB for 8-row
    // tiles, still three workgroups per CU -- and dropping this barrier measured neutral: 3230 -> 3218 us over the
    // twelve shapes, inside the run-to-run spread)
    if (PERS) __syncthreads();
  } while (PERS && (id += gridDim.x) < nPix * nNTall);
}

template <int MT, bool PERS>
__global__ __launch_bounds__(256, WnCfg<MT>::WGS_PER_CU) void wino_conv_kernel(const ConvArgs a) {
  wino_body<MT, PERS, false>(a);
}
template <bool PERS>
__global__ __launch_bounds__(256, 2) void wino_conv_head_kernel(const ConvArgs a) {
  wino_body<2, PERS, true>(a);
}
#ifdef DEPGAN_WINO_ABLATIONS
template <int MT, int ABL>
__global__ __launch_bounds__(256, WnCfg<MT>::WGS_PER_CU) void wino_conv_abl_kernel(const ConvArgs a) {
  wino_body<MT, true, false, ABL>(a);
}
#endif

}  // namespace

bool dg_conv_wino_supported(const ConvPlan& pl, const ConvArgs& a) {
  if (pl.variant != 9 || pl.bf16 || pl.KS != 3) return false;
  if (a.Cin != pl.Cin || (a.Cin % WN_CK) || (a.Cout % 32) || a.groups > 1 || a.dbg || ((a.H | a.W) & 1)) return false;
  if (a.cpt > 0 && ((a.Cin % (a.cpt * WN_CK)) != 0 || a.Cin / (a.cpt * WN_CK) > 4)) return false;
  return true;
}

// MT of a launch: DEPGAN_WINO_MT=1|2 forces one form for A/B runs; the fused head exists for MT = 2 only
static int wino_mt(const ConvArgs& a) {
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("DEPGAN_WINO_MT");
    forced = e ? atoi(e) : 0;
  }
  if (a.ep.head_out) return 2;
  if (forced == 1 || forced == 2) return forced;
  // 8-row tiles, three workgroups per CU: with ONE barrier per chunk (wave-private transform) the third wave per SIMD
  // pays -- sum over the step's twelve shapes 3246 us against 3335 us for 16-row tiles, never slower, up to 9 % faster
  // where 16-row tiles leave CUs without a workgroup (profiles/r03_conv_experiments.md)
  return 1;
}

template <int MT>
static void wino_geometry(ConvArgs& a, long* total, long* G, bool* pers) {
  a.lgx = cdiv(a.W, 16) * cdiv(a.H, WnCfg<MT>::TH) * a.B;
  a.lgy = a.Cout / 32;
  *total = (long)a.lgx * a.lgy;
  // resident workgroups per CU by LDS and registers; persistent when there are more items than that
  const long cap = (long)WnCfg<MT>::WGS_PER_CU * dg_cu_count();
  *pers = *total > cap;
  *G = *pers ? cap : *total;
}

const char* dg_conv_wino_name(const ConvArgs& a_in) {
  ConvArgs a = a_in;
  long total, G;
  bool pers;
  const int mt = wino_mt(a);
  if (mt == 2) wino_geometry<2>(a, &total, &G, &pers); else wino_geometry<1>(a, &total, &G, &pers);
  if (a.ep.head_out) return pers ? "wino_conv_head_kernel<true>" : "wino_conv_head_kernel<false>";
  if (mt == 2) return pers ? "wino_conv_kernel<2,true>" : "wino_conv_kernel<2,false>";
  return pers ? "wino_conv_kernel<1,true>" : "wino_conv_kernel<1,false>";
}

template <int MT>
static int wino_launch(ConvArgs a, hipStream_t st) {
  long total, G;
  bool pers;
  wino_geometry<MT>(a, &total, &G, &pers);
  constexpr size_t LDS = WnCfg<MT>::LDS;
  static DgOncePerDevice once;
  if (once.need()) {
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_conv_kernel<MT, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_conv_kernel<MT, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    if (MT == 2) {
      HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_conv_head_kernel<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
      HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_conv_head_kernel<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    }
  }
#ifdef DEPGAN_WINO_ABLATIONS
  if (const char* e = getenv("DEPGAN_WINO_ABL")) {
    const int abl = atoi(e);
    if (abl && pers && !a.ep.head_out) {
#define WN_ABL_CASE(N)                                                                                              \
  case N:                                                                                                           \
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_conv_abl_kernel<MT, N>),                       \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));                            \
    hipLaunchKernelGGL((wino_conv_abl_kernel<MT, N>), dim3((unsigned)G), dim3(256), LDS, st, a);                    \
    break;
      switch (abl) {
        WN_ABL_CASE(1) WN_ABL_CASE(2) WN_ABL_CASE(3) WN_ABL_CASE(4) WN_ABL_CASE(8) WN_ABL_CASE(10) WN_ABL_CASE(11)
        default: dg_set_error("DEPGAN_WINO_ABL=%d not instantiated", abl); return DG_ERR_ARG;
      }
      HIPCHECK(hipGetLastError());
      return DG_OK;
    }
  }
#endif
  if (a.ep.head_out) {
    if (MT != 2 || a.Cout != 32 || a.ep.pool.p) {
      dg_set_error("dg_conv_wino: the fused head needs 32 output channels and no fused pool");
      return DG_ERR_ARG;
    }
    if (pers) hipLaunchKernelGGL((wino_conv_head_kernel<true>), dim3((unsigned)G), dim3(256), LDS, st, a);
    else hipLaunchKernelGGL((wino_conv_head_kernel<false>), dim3((unsigned)G), dim3(256), LDS, st, a);
  } else {
    if (pers) hipLaunchKernelGGL((wino_conv_kernel<MT, true>), dim3((unsigned)G), dim3(256), LDS, st, a);
    else hipLaunchKernelGGL((wino_conv_kernel<MT, false>), dim3((unsigned)G), dim3(256), LDS, st, a);
  }
  HIPCHECK(hipGetLastError());
  return DG_OK;
}

int dg_conv_wino(const ConvPlan& pl, const ConvArgs& a, hipStream_t st) {
  if (!dg_conv_wino_supported(pl, a)) {
    dg_set_error("dg_conv_wino: shape not covered (3x3, Cin %% 8, Cout %% 32, even H and W, no groups)");
    return DG_ERR_UNSUPPORTED;
  }
  return wino_mt(a) == 2 ? wino_launch<2>(a, st) : wino_launch<1>(a, st);
}
