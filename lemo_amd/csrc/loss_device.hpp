// Device bodies of the two loss kernels that follow the encoder's forward pass; shared so that the fitting engine
// can run both in ONE launch (fit_losses_kernel in loss_kernels.hip: they are independent, ~6 us each, and a kernel
// boundary costs ~3 us).
#pragma once
#include "kernels.hpp"

namespace lemo {

// ---- latent smoothness loss (opt_amass_temp.py:390-391) + its gradient, fused ------------------
//   loss = mean_{c,y,x<W-1} (z[c,y,x+1]-z[c,y,x])^2
//   dpre[c,y,x] = coef * 2 * ((z[x]-z[x-1])[x>=1] - (z[x+1]-z[x])[x<=W-2]) * lrelu'(z[c,y,x])
// with coef = weight / (C*H*(W-1)).  Per-block partial sums of the squared differences go to
// `partial[blockIdx.x]` (fixed-order final reduction elsewhere -> deterministic).
__device__ __forceinline__ void smooth_loss_body(int blk, const float* __restrict__ z, float* __restrict__ dpre, float* __restrict__ partial,
                                                 int H, int W, int C, float coef2, double* __restrict__ acc) {
  __shared__ float red[4];
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const int idx = blk * blockDim.x + threadIdx.x;
  const int n = P * (C >> 3) * 2;                                   // one thread per float4
  float sq = 0.f;
  if (idx < n) {
    const int half = idx & 1, rest = idx >> 1;
    const int g = rest / P, p = rest - g * P;
    const int y = p / W, x = p - y * W;
    const size_t o = ((size_t)g * HWp + (y + 1) * Wp + (x + 1)) * 8 + 4 * half;
    const float4 c = ld4(z + o);
    float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x >= 1) {
      const float4 l = ld4(z + o - 8);
      gr.x += c.x - l.x; gr.y += c.y - l.y; gr.z += c.z - l.z; gr.w += c.w - l.w;
    }
    if (x <= W - 2) {
      const float4 r = ld4(z + o + 8);
      const float d0 = r.x - c.x, d1 = r.y - c.y, d2 = r.z - c.z, d3 = r.w - c.w;
      sq = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
      gr.x -= d0; gr.y -= d1; gr.z -= d2; gr.w -= d3;
    }
    st4(dpre + o, make_float4(coef2 * gr.x * lrelu_grad_from_out(c.x), coef2 * gr.y * lrelu_grad_from_out(c.y),
                              coef2 * gr.z * lrelu_grad_from_out(c.z), coef2 * gr.w * lrelu_grad_from_out(c.w)));
  }
  const float s = block_sum(sq, red);
  if (threadIdx.x == 0) {
    // f64 accumulation (order effects ~1e-16, invisible after the f32 cast), spread over 32 slots of 16
    // doubles: 2000+ blocks on ONE address serialise in L2 (measured 28 us)
    if (acc) atomicAdd(acc + (blk & 31) * 16, (double)s);
    else partial[blk] = s;
  }
}

// Loss accumulators (f64 [32 slots][16], zeroed at the start of every iteration by the pose-stage kernel;
// a block adds to slot blockIdx & 31 to keep the L2 atomics from serialising on one address):
//   [0] marker L1 sum ; [1+k] contact-velocity sum, [5+k] count (k = 4 foot sets) ; [9] smoothness sum of
//   squares ; [10] sum z^2 ; [11] sum betas^2 ; [12] sum hands^2.
// Accumulating f32 block sums into f64 with atomics is order-dependent only at ~1e-16 relative, far
// below the f32 value that is finally reported.
__device__ __forceinline__ void vertex_loss_body(int b, FitConst fc, const float* __restrict__ verts, int nrows, const float* __restrict__ target,
                              const float* __restrict__ contact, const float* __restrict__ shape,
                              const float* __restrict__ other, int B, double* __restrict__ accg) {
  __shared__ float red[4][12];
  const int t = threadIdx.x;
  float acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  for (int w = t; w < fc.n67 * 3; w += 256) {
    const int m = w / 3, c = w % 3;
    acc[0] += fabsf(verts[((size_t)b * nrows + fc.row67[m]) * 3 + c] - target[((size_t)b * fc.n67 + m) * 3 + c]);
  }
  if (b < B - 1) {
    for (int k = 0; k < 4; ++k) {
      if (contact[(size_t)b * 4 + k] != 1.f) continue;
      for (int q = fc.foot_start[k] + t; q < fc.foot_start[k + 1]; q += 256) {
        const float* v0 = verts + ((size_t)b * nrows + fc.foot_row[q]) * 3;
        const float* v1 = v0 + (size_t)nrows * 3;
        const float vx = (v1[0] - v0[0]) * 30.f, vy = (v1[1] - v0[1]) * 30.f, vz = (v1[2] - v0[2]) * 30.f;
        const float sp = sqrtf(vx * vx + vy * vy + vz * vz);
        if (sp - 0.1f > 0.f) { acc[1 + k] += sp; acc[5 + k] += 1.f; }
      }
    }
  }
  // L2 priors of this frame (opt_amass_temp.py:397-404): z (32), betas (10), hands (24)
  if (t < 32) { const float v = other[(size_t)b * 56 + t]; acc[9] = v * v; }
  else if (t >= 64 && t < 74) { const float v = shape[(size_t)b * 10 + (t - 64)]; acc[10] = v * v; }
  else if (t >= 128 && t < 152) { const float v = other[(size_t)b * 56 + 32 + (t - 128)]; acc[11] = v * v; }
  // 12 wave sums, ONE barrier, then 12 threads finish and publish in parallel (was: 12 block sums = 24 barriers,
  // 12 serial atomics from thread 0); the order of the adds is fixed -> deterministic
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = wave_sum(acc[i]);
  if ((t & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) red[t >> 6][i] = acc[i];
  }
  __syncthreads();
  if (t < 12) {
    const float v = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
    if (v != 0.f) atomicAdd(accg + (b & 31) * 16 + (t < 9 ? t : t + 1), (double)v);
  }
}

}  // namespace lemo
