// Fixed-order reduction of the split-K GEMM's slab partials (gemm_kernels.hip), as a device body so that another reduction of the same
// stage can share its launch (lbs_kernels.hip: the all-vertex LBS backward's chunk partials).
#pragma once
#include "common.hpp"

namespace lemo {

#define GEMM_SK_NT 8     // frames per workgroup / 16 of the split-K kernels: partial tiles are [slab][128 frames][M]

// C = sum over the S slab partials, in a FIXED order (deterministic).  32 outputs (float4) x 8 slab groups per workgroup: group g adds
// slabs g, g + 8, ... in order with eight loads in flight, the eight group sums are added in group order through LDS.  (Rounds 1-2:
// one thread per output walking all S slabs -- 50 workgroups reading 25 MB: 11.7 us.)
__device__ __forceinline__ void gemm_splitk_reduce_body(const float* __restrict__ part, int M, int N, int S, float* __restrict__ C, int ldc, int blk) {
  __shared__ float4 red[8][32];
  const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int i = blk * 32 + o;                              // float4 index over [N][M / 4]
  const int m4 = M >> 2, tot = N * m4;
  const int ic = i < tot ? i : tot - 1;
  const int n = ic / m4, mq = ic - n * m4;
  const size_t stride = (size_t)(GEMM_SK_NT * 16) * M;
  const float* p = part + (size_t)n * M + mq * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = g; s0 < S; s0 += 64) {                     // slabs g, g + 8, ..., eight in flight
    float4 r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = ld4(p + (size_t)(s0 + 8 * u < S ? s0 + 8 * u : g) * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) if (s0 + 8 * u < S) { v.x += r[u].x; v.y += r[u].y; v.z += r[u].z; v.w += r[u].w; }
  }
  red[g][o] = v;
  __syncthreads();
  if (g == 0 && i < tot) {
#pragma unroll
    for (int k = 1; k < 8; ++k) { const float4 w = red[k][o]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
    st4(C + (size_t)n * ldc + mq * 4, v);
  }
}
static inline int gemm_splitk_reduce_blocks(int M, int N) { return (N * (M >> 2) + 31) / 32; }

}  // namespace lemo
