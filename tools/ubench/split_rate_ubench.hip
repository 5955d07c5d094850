// How fast is the fp32 -> two-fp16-pieces split (conv_f16.hpp::split2x4) and a dependent f16 MFMA chain on gfx950?
// hipcc --offload-arch=gfx950 -O3 -I lemo_amd/csrc -I include tools/ubench/split_rate_ubench.hip -o tools/ubench/split_rate_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "conv_f16.hpp"
using namespace lemo;

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters, float s) {
  float4 v[4];
  for (int i = 0; i < 4; ++i) v[i] = make_float4(threadIdx.x * 0.001f + i, 1.f + i, 2.f, 3.f);
  f32x16 acc = {0};
  unsigned acc_u = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {                 // 4 independent split2x4 (56 VALU by the compiler's count)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint2 h, l;
        split2x4(v[i], s, h, l);
        acc_u += h.x ^ l.y ^ h.y ^ l.x;
        v[i].x += 1e-3f;
      }
    }
    if (MODE == 1 || MODE == 2) {                 // 3 dependent MFMAs on one accumulator
      f16x8 a = __builtin_bit_cast(f16x8, make_uint4(acc_u | 0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + (float)acc_u + v[0].x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8 * 4096);
  unsigned long long h[4096];
  const int iters = 2000;
  for (int threads : {64, 256, 512, 1024}) {
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 256.f);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 256.f);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 256.f);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, cyc, 8 * 256, hipMemcpyDeviceToHost);
      double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
      printf("%4d threads per CU (%d waves per SIMD), mode %d (%s): %.1f cycles per iteration\n", threads, threads / 256 > 0 ? threads / 256 : 1, mode,
             mode == 0 ? "4 x split2x4" : mode == 1 ? "3 dependent f16 MFMAs" : "both", m / iters);
    }
  }
  return 0;
}
