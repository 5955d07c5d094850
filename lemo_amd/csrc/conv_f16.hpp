// Two-piece fp16 operand split ("split-f16", conv variant 4 and the fused layer pairs): shared device helpers.
// x * s = hi + lo with hi = f16(x s), lo = f16(x s - hi) (2 x 11 significand bits: the operand is carried to 2^-22), s an exact
// power of two taken from the maximum of the tile a workgroup holds; a b ~= a_hi b_lo + a_lo b_hi + a_hi b_hi in fp32
// (conv_split_kernels.hip header has the scheme, its error analysis and the measurements).
#pragma once
#include "common.hpp"

namespace lemo {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// x * s = hi + lo, hi = f16(x s), lo = f16(x s - hi)  (round-to-nearest-even; s an exact power of two)
__device__ __forceinline__ void split2x4(float4 v, float s, uint2& hi, uint2& lo) {
  const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
  f16x4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = (_Float16)x[e];
    l[e] = (_Float16)(x[e] - (float)h[e]);
  }
  hi = __builtin_bit_cast(uint2, h);
  lo = __builtin_bit_cast(uint2, l);
}
// max over the 64 lanes of a wave (DPP rows, then the four row leaders), same value in every lane
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  v = fmaxf(v, dpp_move<0x141>(v));
  v = fmaxf(v, dpp_move<0x140>(v));
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// power-of-two scale that puts a tile maximum m into [2^14, 2^15) (fp16 overflows at 65504) and its exact inverse;
// m = 0 / denormal / huge: exponent clamped, scale * inverse == 1 always
__device__ __forceinline__ void f16_scale_for(float m, float& scale, float& inv) {
  const int biased = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu);
  int bs = 268 - biased;                                      // 127 + 14 - (biased - 127)
  bs = bs < 1 ? 1 : (bs > 253 ? 253 : bs);
  scale = __builtin_bit_cast(float, (unsigned)bs << 23);
  inv = __builtin_bit_cast(float, (unsigned)(254 - bs) << 23);
}
// scale of a LATER staging phase of the same accumulation: the accumulators are rescaled by scale / prev when the phase changes, so the
// ratio is bounded (2^40: sums of <= 2^39 stay finite) -- a phase that is all zero, or > 2^40 below the previous one, takes prev * 2^40
// instead of 2^126 (its operands are then carried to an absolute 2^-64 of the previous phase's: far below that phase's own rounding).
// Without the bound an input whose channels 16-31 / 48-63 vanish on a tile turned every output of that tile into Inf * 0.
__device__ __forceinline__ void f16_scale_after(float m, float prev_scale, float& scale, float& inv) {
  const int biased = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu);
  const int bp = (int)((__builtin_bit_cast(unsigned, prev_scale) >> 23) & 0xffu);
  int bs = 268 - biased;
  bs = bs > bp + 40 ? bp + 40 : bs;
  bs = bs < 1 ? 1 : (bs > 253 ? 253 : bs);
  scale = __builtin_bit_cast(float, (unsigned)bs << 23);
  inv = __builtin_bit_cast(float, (unsigned)(254 - bs) << 23);
}
__device__ __forceinline__ float absmax4(float4 v, float m) {
  return fmaxf(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))), m);
}

}  // namespace lemo
