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

// ---- weight gradient of the same layer: four taps + bias / BN-beta column sums in one launch (deconv_wgrad.hip) ----
struct DeconvWgradArgs {
  const float* in;     // dense NHWC [B][H][W][Cin] (the layer's forward input)
  TView dout;          // gradient at the (2H, 2W) output grid (may be a channel slice of a concat buffer)
  float* part;         // [nchunks][4][Cin][Cout] partial slabs (dg_deconv_wgrad_part_floats floats)
  float* colpart;      // [nchunks * 4][Cout] partial column sums of dout, or null
  int H, W, Cin, Cout;
  int steps_per_wg, lgW, lgH;  // filled by the launcher
};
// true when the fused kernel covers the layer (Cin = Cout in {64, 96, 128}, dense input, H and W powers of two, the
// k-steps of 4 pixels divisible among the workgroups) and DEPGAN_DECONV_FUSED is not 0
bool dg_deconv_wgrad_supported(int B, int H, int W, int Cin, int Cout, TView in, TView dout);
size_t dg_deconv_wgrad_part_floats(int B, int H, int W, int Cin, int Cout);
// launches the kernel; *nchunks = number of partial slabs (and nchunks * 4 partial column rows) for dg_wgrad_finish_rows
int dg_deconv_wgrad(DeconvWgradArgs a, int B, int* nchunks, hipStream_t st);
