// Epilogues shared by the 3x3 convolution kernels (conv_kernels.hip, conv_split_kernels.hip).
#pragma once
#include "kernels.hpp"

namespace lemo {

// EPI 0: out = lrelu(acc + bias)            (forward layer)
// EPI 1: out = acc * lrelu'(aux)            (backward-data; aux = saved forward activation at the
//                                             output position, same layout/channels as `out`)
// EPI 2: out = acc + bias                   (plain conv, no activation)
template <int EPI>
__device__ __forceinline__ void conv_store4(float* __restrict__ out, const float* __restrict__ bias,
                                            const float* __restrict__ aux, size_t o, int c0, float4 v) {
  if (EPI == 0 || EPI == 2) {
    const float4 bb = ld4(bias + c0);
    v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
    if (EPI == 0) { v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w); }
  } else {
    const float4 yy = ld4(aux + o);
    v.x *= lrelu_grad_from_out(yy.x); v.y *= lrelu_grad_from_out(yy.y);
    v.z *= lrelu_grad_from_out(yy.z); v.w *= lrelu_grad_from_out(yy.w);
  }
  st4(out + o, v);
}

// scalar form of the same epilogues (remainder patches of the split-bf16 kernel)
template <int EPI>
__device__ __forceinline__ void conv_store1(float* __restrict__ out, const float* __restrict__ bias,
                                            const float* __restrict__ aux, size_t o, int c, float v) {
  if (EPI == 0 || EPI == 2) {
    v += bias[c];
    if (EPI == 0) v = lrelu(v);
  } else {
    v *= lrelu_grad_from_out(aux[o]);
  }
  out[o] = v;
}

}  // namespace lemo
