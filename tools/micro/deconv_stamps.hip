// Phase stamps of the fused transposed-convolution kernel (dep_gan_im_amd/csrc/deconv_fwd.hip built with
// DECONV_STAMPS): s_memtime at the top of a workgroup's fourth tile, before and after its tile barrier, after its last
// MFMA, after the LDS-block writes and at the top of the fifth tile, per wave.
// hipcc -O3 --offload-arch=gfx950 -std=c++17 -DDECONV_STAMPS -I dep_gan_im_amd/csrc tools/micro/deconv_stamps.hip -o /tmp/ds
#include <stdarg.h>
#include <stdio.h>
#include <vector>
#include "deconv_fwd.hip"
void dg_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int main(int argc, char** argv) {
  const int B = 32, H = 128, W = 128, Cin = 64, Cout = 64;
  float *in, *w, *out, *bias;
  unsigned long long* stamps;
  const size_t nin = (size_t)B * H * W * Cin, nout = (size_t)B * 4 * H * W * Cout;
  hipMalloc(&in, nin * 4); hipMalloc(&w, 4 * Cin * Cout * 4); hipMalloc(&out, nout * 4); hipMalloc(&bias, Cout * 4);
  hipMalloc(&stamps, 1024 * 4 * 8 * 8);
  hipMemset(in, 0, nin * 4); hipMemset(w, 0, 4 * Cin * Cout * 4); hipMemset(bias, 0, Cout * 4);
  if (argc > 2 && atoi(argv[2])) {     // random operands: the matrix pipe's power depends on the data
    std::vector<float> hin(nin), hw(4 * Cin * Cout);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : hin) v = rnd();
    for (auto& v : hw) v = rnd() * 0.125f;
    hipMemcpy(in, hin.data(), nin * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  }
  hipMemset(stamps, 0, 1024 * 4 * 8 * 8);
  DeconvArgs a = {};
  a.in = in; a.w = w; a.out = make_view(out, 2 * H, 2 * W, Cout); a.bias = bias; a.relu = 1;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.stamps = stamps;
  a.stamp_it = argc > 1 ? atoi(argv[1]) : 3;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 10; ++rep) {
    hipEventRecord(e0, 0);
    if (dg_deconv_fwd(a, B, 0) != DG_OK) return 1;
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("kernel %.1f us, stamps at tile %d of each workgroup, %s operands\n", ms * 1e3, a.stamp_it,
         (argc > 2 && atoi(argv[2])) ? "random" : "zero");
  std::vector<unsigned long long> h(1024 * 4 * 8);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  // stamps: 0 top of the fourth tile, 4 / 5 before / after the tile barrier, 2 after the last MFMA, 3 after the
  // LDS-block writes, 1 top of the fifth tile
  const char* nm[5] = {"stream to the barrier", "barrier", "last K group", "LDS block writes", "to next top"};
  const int a0[5] = {0, 4, 5, 2, 3}, a1[5] = {4, 5, 2, 3, 1};
  for (int wg = 0; wg < 2; ++wg)
    for (int wv = 0; wv < 4; ++wv) {
      const unsigned long long* t = &h[(wg * 4 + wv) * 8];
      printf("wg %d wave %d:", wg, wv);
      for (int k = 0; k < 5; ++k) printf("  %s %llu", nm[k], t[a1[k]] - t[a0[k]]);
      printf("  | tile %llu\n", t[1] - t[0]);
    }
  double s[5] = {0, 0, 0, 0, 0}, tt = 0; int n = 0;
  for (int i = 0; i < 1024 * 4; ++i) {
    const unsigned long long* t = &h[i * 8];
    if (!t[1]) continue;
    for (int k = 0; k < 5; ++k) s[k] += (double)(t[a1[k]] - t[a0[k]]);
    tt += (double)(t[1] - t[0]);
    ++n;
  }
  printf("mean over %d waves:", n);
  for (int k = 0; k < 5; ++k) printf("  %s %.0f", nm[k], s[k] / n);
  printf("  | tile %.0f\n", tt / n);
  return 0;
}
