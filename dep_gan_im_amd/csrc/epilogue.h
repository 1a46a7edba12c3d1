// Fused conv epilogue shared by the MFMA and the direct convolution kernels.
#pragma once
#include "common.h"

// per-lane constants for one output channel (and one sample)
struct EpiChan {
  float bias, scale, shift, fmul, fadd;
};

__device__ __forceinline__ EpiChan epi_load_chan(const Epilogue& e, int b, int co, int Cout) {
  EpiChan c;
  c.bias = e.bias ? e.bias[co] : 0.f;
  c.scale = e.scale ? e.scale[co] : 1.f;
  c.shift = e.shift ? e.shift[co] : 0.f;
  c.fmul = e.film_mul ? e.film_mul[(long)b * e.film_ld + co] : 1.f;
  c.fadd = e.film_add ? e.film_add[(long)b * e.film_ld + co] : 0.f;
  return c;
}

__device__ __forceinline__ long view_off(const TView& v, int b, int y, int x) {
  return (long)b * v.sB + (long)y * v.sY + (long)x * v.sX;
}

// FiLM pre-activation v = u * mul + add (GT:403-404: a Multiply layer, then an Add layer -- two roundings).  One
// function, so that the forward pass (ReLU of v) and the backward pass (sign of v recomputed from the stored u) take the
// same decision on every unit.  The contraction pragma is what guarantees it: hipcc's __fmul_rn / __fadd_rn are plain
// `x * y` / `x + y` and were contracted into v_pk_fma_f32 in film_bwd_partial but not in the convolution epilogue (found
// by the mask-pinned gradient test: one unit of 1.6e7 at 256x256 decided differently in the two passes).
__device__ __forceinline__ float film_preact(float u, float fmul, float fadd) {
#pragma clang fp contract(off)
  const float p = u * fmul;
  return p + fadd;
}

__device__ __forceinline__ void epi_store(const ConvArgs& a, const EpiChan& c, int b, int oy, int ox, int co,
                                          float acc) {
  const Epilogue& e = a.ep;
  float v = acc + c.bias;
  if (e.scale) v = __fadd_rn(__fmul_rn(v, c.scale), c.shift);
  if (e.out_pre.p) e.out_pre.p[view_off(e.out_pre, b, oy, ox) + co] = v;
  if (e.film_mul) v = film_preact(v, c.fmul, c.fadd);
  if (e.relu) v = fmaxf(v, 0.f);
  if (e.res.p) v += e.res.p[view_off(e.res, b, oy, ox) + co];
  if (e.mask.p) v = (e.mask.p[view_off(e.mask, b, oy, ox) + co] > 0.f) ? v : 0.f;
  float* o = a.out.p + view_off(a.out, b, oy, ox) + co;
  if (e.accumulate) v += *o;
  *o = v;
}

// 4 consecutive output channels at once (16-byte accesses); requires 16-byte aligned views.
__device__ __forceinline__ void epi_store4(const ConvArgs& a, int b, int oy, int ox, int co, f32x4 v) {
  const Epilogue& e = a.ep;
  if (e.bias) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(e.bias + co);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += t[k];
  }
  if (e.scale) {
    const f32x4 sc = *reinterpret_cast<const f32x4*>(e.scale + co);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(e.shift + co);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __fadd_rn(__fmul_rn(v[k], sc[k]), sh[k]);
  }
  if (e.out_pre.p) *reinterpret_cast<f32x4*>(e.out_pre.p + view_off(e.out_pre, b, oy, ox) + co) = v;
  if (e.film_mul) {
    const f32x4 fm = *reinterpret_cast<const f32x4*>(e.film_mul + (long)b * e.film_ld + co);
    const f32x4 fa = *reinterpret_cast<const f32x4*>(e.film_add + (long)b * e.film_ld + co);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = film_preact(v[k], fm[k], fa[k]);
  }
  if (e.relu) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
  }
  if (e.res.p) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(e.res.p + view_off(e.res, b, oy, ox) + co);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += r[k];
  }
  if (e.mask.p) {
    const f32x4 m = *reinterpret_cast<const f32x4*>(e.mask.p + view_off(e.mask, b, oy, ox) + co);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (m[k] > 0.f) ? v[k] : 0.f;
  }
  f32x4* o = reinterpret_cast<f32x4*>(a.out.p + view_off(a.out, b, oy, ox) + co);
  if (e.accumulate) {
    const f32x4 old = *o;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += old[k];
  }
  *o = v;
}
