// Micro check: buffer-addressed LDS-DMA (16 B per lane) on gfx950: per-lane voffset, scalar soffset, and an
// out-of-range voffset returning zeros into LDS (replaces the pointer select against a zero constant).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* p, float* out, int soff, unsigned nbytes) {
  __shared__ __attribute__((aligned(16))) float s[256 * 2];
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, nbytes, 0x00020000);
  for (int i = threadIdx.x; i < 512; i += 64) s[i] = -1.f;
  __syncthreads();
  // lanes 0..47 read 16 B each at lane*16; lanes 48..63 get an out-of-range offset -> zeros
  const int voff = threadIdx.x < 48 ? (int)threadIdx.x * 16 : 0x7FFFFF00;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)s, 16, voff, soff, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(s + 256), 16, voff, soff + 1024, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = s[i];
}
int main() {
  float *d, *o; hipMalloc(&d, 8192 * 4); hipMalloc(&o, 512 * 4);
  float h[8192]; for (int i = 0; i < 8192; ++i) h[i] = (float)i;
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 4096, 8192u * 4u);
  float r[512]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) { float e = i < 192 ? 1024.f + i : 0.f; if (r[i] != e) ++bad; }
  for (int i = 0; i < 256; ++i) { float e = i < 192 ? 1280.f + i : 0.f; if (r[256 + i] != e) ++bad; }
  printf("buffer->LDS 16B: %d mismatches; r[0..3]=%g %g %g %g r[191]=%g r[192]=%g r[255]=%g r[256]=%g\n", bad, r[0], r[1], r[2], r[3], r[191], r[192], r[255], r[256]);
  return bad != 0;
}
