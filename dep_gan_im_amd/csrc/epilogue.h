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

// FiLM pre-activation; kept as one function so forward and backward evaluate
// the sign of v with the identical instruction sequence.
__device__ __forceinline__ float film_preact(float u, float fmul, float fadd) { return __fadd_rn(__fmul_rn(u, fmul), fadd); }

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
