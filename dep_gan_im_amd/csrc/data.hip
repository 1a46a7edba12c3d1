// Data step in front of the path: what the reference does to the NIfTI volumes of one subject
// between nib.load and the training arrays (GT:93-118 load_data / data_prep, GT:124-146
// map_image_to_intensity_range, GT:667-723 masking, clamping, channel concatenation).
//
//   slices[z][x][y] = vol(x, y, z)                                    (data_prep: image[:, :, z], channel axis added)
//   brain_prob_1 = p1 * icv1 [* (1 - sl1)]     brain_flair_1 = f1 * icv1 [* (1 - sl1)]
//   brain_prob_2 = p2 * icv2 [* (1 - sl2)]
//   brain_flair_1 = clip((brain_flair_1 - min) / (max - min) * (1 - 0) + 0, 0, 1)   (min / max over the subject)
//   brain_prob_* [ < 0 ] = 0;   x = concat(brain_prob_1, brain_flair_1) on the channel axis when nicg = 2
//
// All of it is HBM-bound fp32 elementwise work plus one min/max reduction; the only structure is the per-slice
// transpose (file order is x fastest, the network wants NHWC with y fastest), done through a 32 x 33 LDS tile so
// that both the volume reads and the slice writes are coalesced.  Every arithmetic step is a single correctly
// rounded fp32 operation in the reference's order, so the result is bit-identical to the NumPy statements.
#include "common.h"

#include "model.h"

namespace {

constexpr int TS = 32;

__device__ __forceinline__ float masked(const float* v, const float* icv, const float* sl, size_t i) {
  float r = __fmul_rn(v[i], icv[i]);
  if (sl) r = __fmul_rn(r, __fsub_rn(1.0f, sl[i]));
  return r;
}

// grid (ceil(X/32), ceil(Y/32), Z), block (32, 8)
__global__ __launch_bounds__(256) void subject_prep_kernel(const float* __restrict__ p1, const float* __restrict__ f1,
                                                           const float* __restrict__ icv1,
                                                           const float* __restrict__ sl1,
                                                           const float* __restrict__ p2,
                                                           const float* __restrict__ icv2,
                                                           const float* __restrict__ sl2, int X, int Y, int nicg,
                                                           float* __restrict__ xo, float* __restrict__ yo,
                                                           float* __restrict__ part) {
  __shared__ float tp[TS][TS + 1], tf[TS][TS + 1], ty[TS][TS + 1];
  __shared__ float rmin[8], rmax[8];
  const int z = blockIdx.z;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const size_t vbase = (size_t)z * X * Y;
  float mn = __builtin_inff(), mx = -__builtin_inff();
  // read: x fastest (file order); threadIdx.x walks x
  for (int j = threadIdx.y; j < TS; j += 8) {
    const int x = x0 + threadIdx.x, y = y0 + j;
    if (x < X && y < Y) {
      const size_t i = vbase + (size_t)y * X + x;
      tp[j][threadIdx.x] = masked(p1, icv1, sl1, i);
      ty[j][threadIdx.x] = masked(p2, icv2, sl2, i);
      if (nicg == 2) {
        const float f = masked(f1, icv1, sl1, i);
        tf[j][threadIdx.x] = f;
        mn = fminf(mn, f);
        mx = fmaxf(mx, f);
      }
    }
  }
  __syncthreads();
  // write: y fastest (NHWC slice [z][x][y][c]); threadIdx.x walks y
  for (int j = threadIdx.y; j < TS; j += 8) {
    const int x = x0 + j, y = y0 + threadIdx.x;
    if (x < X && y < Y) {
      const size_t o = ((size_t)z * X + x) * Y + y;
      float a = tp[threadIdx.x][j], b = ty[threadIdx.x][j];
      a = (a < 0.f) ? 0.f : a;     // brain_prob[brain_prob < 0] = 0 (GT:706-707); NaN stays NaN as in NumPy
      b = (b < 0.f) ? 0.f : b;
      yo[o] = b;
      if (nicg == 2) {
        xo[2 * o] = a;
        xo[2 * o + 1] = tf[threadIdx.x][j];   // normalised in place by the second kernel
      } else {
        xo[o] = a;
      }
    }
  }
  if (nicg == 2) {
    // block min / max of the masked FLAIR -> partials (NaNs are ignored by fminf / fmaxf; np.percentile would return
    // NaN -- volumes with NaNs are outside what the reference can process either)
    for (int o = 32; o > 0; o >>= 1) {
      mn = fminf(mn, __shfl_down(mn, o));
      mx = fmaxf(mx, __shfl_down(mx, o));
    }
    const int tid = threadIdx.y * 32 + threadIdx.x;
    if ((tid & 63) == 0) {
      rmin[tid >> 6] = mn;
      rmax[tid >> 6] = mx;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w) {
        mn = fminf(mn, rmin[w]);
        mx = fmaxf(mx, rmax[w]);
      }
      const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      part[2 * b] = mn;
      part[2 * b + 1] = mx;
    }
  }
}

__global__ __launch_bounds__(256) void minmax_final_kernel(const float* __restrict__ part, size_t n,
                                                           float* __restrict__ out) {
  __shared__ float rmin[4], rmax[4];
  float mn = __builtin_inff(), mx = -__builtin_inff();
  for (size_t i = threadIdx.x; i < n; i += 256) {
    mn = fminf(mn, part[2 * i]);
    mx = fmaxf(mx, part[2 * i + 1]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_down(mn, o));
    mx = fmaxf(mx, __shfl_down(mx, o));
  }
  if ((threadIdx.x & 63) == 0) {
    rmin[threadIdx.x >> 6] = mn;
    rmax[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      mn = fminf(mn, rmin[w]);
      mx = fmaxf(mx, rmax[w]);
    }
    out[0] = mn;
    out[1] = mx;
  }
}

// channel 1 of x (npix, 2): v -> clip((v - min) / (max - min) * (max_o - min_o) + min_o, min_o, max_o), GT:140-144
__global__ __launch_bounds__(256) void flair_normalise_kernel(float* __restrict__ xo, size_t npix,
                                                              const float* __restrict__ mm, float min_o, float max_o) {
  // hipcc's __fmul_rn / __fadd_rn are plain `*` / `+` inside a header and may be contracted into an FMA; operators
  // written HERE under the pragma are what keeps every operation individually rounded, as NumPy's statement sequence
  // is (GT:140-144)
#pragma clang fp contract(off)
  const float mn = mm[0], rng = __fsub_rn(mm[1], mm[0]), span = __fsub_rn(max_o, min_o);
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
    float v = xo[2 * i + 1];
    const float q = __fdiv_rn(v - mn, rng);
    const float t = q * span;
    v = t + min_o;
    v = (v > max_o) ? max_o : v;
    v = (v < min_o) ? min_o : v;
    xo[2 * i + 1] = v;
  }
}

}  // namespace

extern "C" {

size_t depgan_data_prep_scratch_floats(int X, int Y, int Z) {
  if (X <= 0 || Y <= 0 || Z <= 0) return 0;
  return 2 * (size_t)cdiv(X, TS) * cdiv(Y, TS) * Z + 2;
}

int depgan_data_prep_subject(const float* p1, const float* f1, const float* icv1, const float* sl1, const float* p2,
                             const float* icv2, const float* sl2, int X, int Y, int Z, int nicg, float* x_out,
                             float* y2_out, float* scratch, void* stream) {
  if (!p1 || !icv1 || !p2 || !icv2 || !x_out || !y2_out || X <= 0 || Y <= 0 || Z <= 0 || (nicg != 1 && nicg != 2) ||
      (nicg == 2 && (!f1 || !scratch))) {
    dg_set_error("data_prep_subject: bad argument (X=%d Y=%d Z=%d nicg=%d)", X, Y, Z, nicg);
    return DG_ERR_ARG;
  }
  if (Z > 65535 || cdiv(Y, TS) > 65535) {
    dg_set_error("data_prep_subject: volume too large for one launch (Y=%d Z=%d)", Y, Z);
    return DG_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(cdiv(X, TS), cdiv(Y, TS), Z);
  const size_t nblk = (size_t)grid.x * grid.y * grid.z;
  hipLaunchKernelGGL(subject_prep_kernel, grid, dim3(32, 8), 0, st, p1, f1, icv1, sl1, p2, icv2, sl2, X, Y, nicg, x_out,
                     y2_out, scratch);
  HIPCHECK(hipGetLastError());
  if (nicg == 2) {
    float* mm = scratch + 2 * nblk;
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, st, scratch, nblk, mm);
    HIPCHECK(hipGetLastError());
    const size_t npix = (size_t)X * Y * Z;
    const int blocks = (int)((npix + 255) / 256 < 2048 ? (npix + 255) / 256 : 2048);
    hipLaunchKernelGGL(flair_normalise_kernel, dim3(blocks), dim3(256), 0, st, x_out, npix, mm, 0.0f, 1.0f);
    HIPCHECK(hipGetLastError());
  }
  return DG_OK;
}

}  // extern "C"
