// Fused forward of the 2x2 / stride-2 transposed convolution (deconv_fwd.hip).
#pragma once
#include "common.h"

struct DeconvArgs {
  const float* in;     // dense NHWC [B][H][W][Cin]
  const float* w;      // Keras Conv2DTranspose kernel (kh, kw, Cout, Cin) = [4 Cout][Cin] rows
  TView out;           // view of the (2H, 2W) output grid (may be a channel slice of a concat buffer)
  const float* bias;   // [Cout] or null
  const float* scale;  // [Cout] or null (then shift is ignored)
  const float* shift;
  int relu;
  int H, W, Cin, Cout;
  int nTiles, lgW, lgH;  // filled by the launcher
  unsigned long long* stamps;  // diagnostics build (DECONV_STAMPS) only: per wave 8 x u64
  int stamp_it;                // ... taken at this tile of the workgroup's sequence
};

// true when the fused kernel covers the layer (Cin in {64, 96, 128}, Cout % 32 == 0, dense 16-byte aligned input,
// H and W powers of two with W >= 8, B H W % 32 == 0) and DEPGAN_DECONV_FUSED is not 0
bool dg_deconv_fwd_supported(int B, int H, int W, int Cin, int Cout, TView in, TView out);
int dg_deconv_fwd(DeconvArgs a, int B, hipStream_t st);
