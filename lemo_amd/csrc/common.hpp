// Shared device helpers for the LEMO hot-path kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// dynamic LDS, 16-byte aligned base (guide G17); tests/hipemu pre-defines its own host version
#ifndef LEMO_DYN_SMEM
#define LEMO_DYN_SMEM(var) extern __shared__ __attribute__((aligned(16))) float var[]
#endif

#define LEMO_WAVE 64
#define LEMO_LRELU_SLOPE 0.2f

// max(v, 0.2 v) == (v > 0 ? v : 0.2 v) bit for bit (slope in (0,1)); v_mul + v_max instead of mul + cmp + cndmask
__device__ __forceinline__ float lrelu(float v) { return fmaxf(v, LEMO_LRELU_SLOPE * v); }
// derivative selected from the *output* sign (slope > 0 keeps the sign; y == 0 <=> x == 0 -> slope,
// matching torch's `x > 0 ? g : g * slope`)
__device__ __forceinline__ float lrelu_grad_from_out(float y) { return y > 0.f ? 1.f : LEMO_LRELU_SLOPE; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---- exact fp32 -> 3 x bf16 operand split (conv_split_kernels.hip header has the error analysis) ---------
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// x = hi + mid + lo exactly: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)  (round-to-nearest-even)
__device__ __forceinline__ void split3x4(float4 v, uint2& hi, uint2& mid, uint2& lo) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  bf16x4 h, m, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = (__bf16)x[e];
    float r = x[e] - (float)h[e];
    m[e] = (__bf16)r;
    r -= (float)m[e];
    l[e] = (__bf16)r;
  }
  hi = __builtin_bit_cast(uint2, h);
  mid = __builtin_bit_cast(uint2, m);
  lo = __builtin_bit_cast(uint2, l);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Deterministic block sum (fixed tree order).  `red` must hold blockDim.x/64 floats.  All threads
// must call; result valid in every thread.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

// ---- tiny 3-vector algebra ------------------------------------------------------------------
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
