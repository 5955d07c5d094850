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


#ifdef LEMO_CENSUS
static __device__ unsigned long long* g_lemo_census = nullptr;   // one copy per translation unit
#define CENSUS_SETTER(NAME) extern "C" int NAME(unsigned long long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lemo_census), &buf, sizeof(buf)); }          // [kernel 4][32 stamps], block 60 thread 0
#define CENSUS_DECL(K) int cz_ = 0; const bool cz_on_ = g_lemo_census && blockIdx.x == 60 && threadIdx.x == 0; unsigned long long* cz_p_ = g_lemo_census + (K) * 32;
#define CENSUS() if (cz_on_ && cz_ < 32) cz_p_[cz_++] = __builtin_amdgcn_s_memtime();
#else
#define CENSUS_DECL(K)
#define CENSUS()
#define CENSUS_SETTER(NAME)
#endif

// keep a loaded value where it was loaded: hipcc sinks a load whose only use sits in a later branch into that branch
#ifndef LEMO_PIN
#define LEMO_PIN(x) asm volatile("" : "+v"(x))
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

// Reductions over DPP rows: v_add_f32 with a DPP source runs at full VALU rate; __shfl_xor compiles to ds_bpermute,
// one LDS-crossbar round trip (~100+ cycles) per step, which dominated the short latency-bound kernels that reduce.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a row, bitwise identical in every lane of the row (xor-1, xor-2 butterflies via quad_perm,
// then half-row and row mirrors: the mirrored partner holds the same value as the xor-4 / xor-8 partner)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_move<0xB1>(v);             // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);             // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);            // row_half_mirror
  v += dpp_move<0x140>(v);            // row_mirror
  return v;
}
// sum over the 64 lanes of a full wave, same value in every lane (all 64 lanes must be active)
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return (r0 + r1) + (r2 + r3);
}

// 16 values per lane -> lane L ends with the sum over its 16-lane row of value (L & 15): each butterfly level keeps the half of the
// values selected by one bit of the lane index (17 DPP moves instead of 64 for 16 separate row sums); fixed order.
__device__ __forceinline__ float row16_transposed_sum(const float (&a)[16], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
  float w8[8], x4[4], y2[2];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const float keep = b0 ? a[2 * m + 1] : a[2 * m], send = b0 ? a[2 * m] : a[2 * m + 1];
    w8[m] = keep + dpp_move<0xB1>(send);                              // lane ^ 1
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float keep = b1 ? w8[2 * m + 1] : w8[2 * m], send = b1 ? w8[2 * m] : w8[2 * m + 1];
    x4[m] = keep + dpp_move<0x4E>(send);                              // lane ^ 2
  }
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const float keep = b2 ? x4[2 * m + 1] : x4[2 * m], send = b2 ? x4[2 * m] : x4[2 * m + 1];
    const float dn = dpp_move<0x124>(send), up = dpp_move<0x12C>(send);   // row_ror 4 / 12: from lane - 4 / lane + 4
    y2[m] = keep + (b2 ? dn : up);                                    // lane ^ 4
  }
  const float keep = b3 ? y2[1] : y2[0], send = b3 ? y2[0] : y2[1];
  return keep + dpp_move<0x128>(send);                                // row_ror 8: lane ^ 8
}
// ... and over the whole wave: every lane L ends with the sum over all 64 lanes of value (L & 15) (two cross-row exchanges through
// the LDS crossbar on top of the row butterfly; all 64 lanes must be active)
__device__ __forceinline__ float wave_transposed_sum16(const float (&a)[16], int lane) {
  float t = row16_transposed_sum(a, lane);
  t += __shfl_xor(t, 16);
  t += __shfl_xor(t, 32);
  return t;
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

// ---- torch.optim.Adam (defaults), one element, OPERATION FOR OPERATION as torch's CPU kernels evaluate it -- probed bit for bit
// against torch 2.10 (tests/test_adam_bits.py; rounds 1-3 used (1.f - 0.999f) for 1 - beta2, 1.3e-5 away from torch's 0.001f):
//   exp_avg.lerp_(g, 1 - b1)                        m' = fma(g - m, 0.1f, m)
//   exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)      v' = fma(0.001f * g, g, v * 0.999f)
//   denom = sqrt(v') / sqrt(1 - b2^t) + eps         (the bias correction's square root in double, rounded once)
//   p.addcdiv_(m', denom, value = -lr / (1 - b1^t))  p' = p + ((-step) * m') / denom  (step in double from the DECIMAL lr)
struct AdamCoef { float neg_step, bc2s; };
__host__ __device__ __forceinline__ AdamCoef adam_coef_t(int t /*1-based*/, double lr) {
  AdamCoef c;
  c.neg_step = (float)(-(lr / (1.0 - pow(0.9, (double)t))));
  c.bc2s = (float)sqrt(1.0 - pow(0.999, (double)t));
  return c;
}
__device__ __forceinline__ void adam_update_torch(float& p, float& m, float& v, float g, AdamCoef c) {
  m = __builtin_fmaf(g - m, 0.1f, m);
  v = __builtin_fmaf(0.001f * g, g, v * 0.999f);
  const float denom = sqrtf(v) / c.bc2s + 1e-8f;
  p = p + (c.neg_step * m) / denom;
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
