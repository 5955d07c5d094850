// 3x3 / stride 1 / pad 1 convolutions of the motion-smoothness encoder (models/AE_sep.py:11-30,
// 77-99, `Enc(downsample=False)`: 10 x [conv3x3 + bias + LeakyReLU(0.2)], no pooling) and their
// backward-data passes, as fp32 implicit GEMMs on the CDNA4 matrix cores.
//
// Why fp32 MFMA: the smoothness loss is weighted by 1e6 (opt_amass_temp.py:49) and the parity
// budget is 1e-5 relative on the loss scalar -> exact-f32 v_mfma_f32_32x32x2_f32 (a k-ordered fmaf
// chain), no bf16.
//
// Data layout in HBM ("CG8P"): an activation with C channels over an H x W image is stored as
//     act[C/8][(H+2)*(W+2)][8]          (channel-group major, zero border of 1 pixel, 8 channels
//                                        of one pixel contiguous = 32 B)
// so that, for one (tap, channel-group), the 32 pixels of an MFMA tile x 32 B are one contiguous
// 1 KiB run: every operand load below is a fully coalesced global_load_dwordx4 and the 3x3 halo
// needs no bounds checks (the border is never written).  Weights are pre-packed on the host as
//     wt[tap][Cin/8][Cout][8]
// GEMM roles: M = Cout (A operand = weights), N = pixels (B operand = activations), K = 9*Cin.
// With the 32x32x2 lane map (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]) one dwordx4 per lane feeds four
// consecutive MFMA k-steps (lane half h consumes channels 4h..4h+3 of the 8-group), and the
// accumulator map (col=l&31 -> pixel, row=(r&3)+8*(r>>2)+4*(l>>5) -> cout) makes the epilogue a
// dwordx4 store per (lane, 8-cout group): 1 KiB contiguous per wave again.
#include "kernels.hpp"

namespace lemo {

// EPI 0: out = lrelu(acc + bias)            (forward layer)
// EPI 1: out = acc * lrelu'(aux)            (backward-data; aux = saved forward activation at the
//                                             output position, same layout/channels as `out`)
// EPI 2: out = acc + bias                   (plain conv, no activation)
template <int MT, int EPI>
__global__ void __launch_bounds__(256)
conv3x3_mfma_kernel(const float* __restrict__ in, const float* __restrict__ wt,
                    const float* __restrict__ bias, const float* __restrict__ aux,
                    float* __restrict__ out, int H, int W, int cin_g, int cout) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const int tile = blockIdx.x * 4 + wave;
  const int m_base = blockIdx.y * (MT * 32);                 // first cout of this block
  const int p = tile * 32 + j;
  if (tile * 32 >= P) return;                                // whole wave out of range (uniform)
  const int pc = p < P ? p : P - 1;
  const int y = pc / W, x = pc - y * W;
  const int poff = (y + 1) * Wp + (x + 1);

  const float* in_l = in + (size_t)poff * 8 + 4 * h;
  const float* wt_l = wt + ((size_t)(m_base + j)) * 8 + 4 * h;
  const size_t in_gstride = (size_t)HWp * 8;
  const size_t wt_itstride = (size_t)cout * 8;

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  const int nit = 9 * cin_g;
  float4 a_cur[MT], b_cur;
  {
    const int tapoff = -Wp - 1;
    b_cur = ld4(in_l + (std::ptrdiff_t)tapoff * 8);
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[m] = ld4(wt_l + (size_t)m * 256);
  }
  int tap = 0, g = 0;
  for (int it = 0; it < nit; ++it) {
    float4 a_nxt[MT], b_nxt;
    int g2 = g + 1, tap2 = tap;
    if (g2 == cin_g) { g2 = 0; tap2 = tap + 1; }
    if (it + 1 < nit) {
      const int dy = tap2 / 3 - 1, dx = tap2 - (tap2 / 3) * 3 - 1;
      b_nxt = ld4(in_l + (std::ptrdiff_t)(dy * Wp + dx) * 8 + (size_t)g2 * in_gstride);
#pragma unroll
      for (int m = 0; m < MT; ++m) a_nxt[m] = ld4(wt_l + (size_t)(it + 1) * wt_itstride + (size_t)m * 256);
    } else {
      b_nxt = b_cur;
#pragma unroll
      for (int m = 0; m < MT; ++m) a_nxt[m] = a_cur[m];
    }
    const float bs[4] = {b_cur.x, b_cur.y, b_cur.z, b_cur.w};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float as = s == 0 ? a_cur[m].x : s == 1 ? a_cur[m].y : s == 2 ? a_cur[m].z : a_cur[m].w;
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, bs[s], acc[m], 0, 0, 0);
      }
    }
    b_cur = b_nxt;
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
    g = g2; tap = tap2;
  }

  if (p < P) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = m_base + m * 32 + q * 8 + 4 * h;          // first of 4 consecutive couts
        const size_t o = ((size_t)(c0 >> 3) * HWp + poff) * 8 + (c0 & 7);
        float4 v = make_float4(acc[m][4 * q + 0], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]);
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
    }
  }
}

int conv3x3_mfma(const float* in, const float* wt, const float* bias, const float* aux, float* out,
                 int H, int W, int cin, int cout, int epi, hipStream_t s) {
  if (cin % 8 || cout % 32 || H <= 0 || W <= 0) return LEMO_ERR_SHAPE;
  const int P = H * W;
  const int mt = (cout % 64 == 0) ? 2 : 1;
  dim3 grid((P + 127) / 128, cout / (mt * 32));
#define LAUNCH(MT_, EPI_) hipLaunchKernelGGL((conv3x3_mfma_kernel<MT_, EPI_>), grid, dim3(256), 0, s, in, wt, bias, aux, out, H, W, cin / 8, cout)
  if (mt == 2) { if (epi == 0) LAUNCH(2, 0); else if (epi == 1) LAUNCH(2, 1); else LAUNCH(2, 2); }
  else         { if (epi == 0) LAUNCH(1, 0); else if (epi == 1) LAUNCH(1, 1); else LAUNCH(1, 2); }
#undef LAUNCH
  return (int)hipGetLastError();
}

// ---- first layer: 1 -> Cout channels (K = 9: VALU) --------------------------------------------
// x0: plain padded single-channel image [(H+2)*(W+2)] (zero border); w: [Cout][9]; out CG8P.
// One thread per (pixel, 8-cout group).
__global__ void __launch_bounds__(256)
conv3x3_c1_kernel(const float* __restrict__ x0, const float* __restrict__ w, const float* __restrict__ bias,
                  float* __restrict__ out, int H, int W, int cout) {
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int ng = cout >> 3;
  if (idx >= P * ng) return;
  const int g = idx / P, p = idx - g * P;
  const int y = p / W, x = p - y * W;
  const int poff = (y + 1) * Wp + (x + 1);
  float xin[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) xin[t] = x0[poff + (t / 3 - 1) * Wp + (t % 3 - 1)];
  float r[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float* wc = w + (size_t)(g * 8 + c) * 9;
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) a = fmaf(wc[t], xin[t], a);
    r[c] = lrelu(a + bias[g * 8 + c]);
  }
  float* o = out + ((size_t)g * HWp + poff) * 8;
  st4(o, make_float4(r[0], r[1], r[2], r[3]));
  st4(o + 4, make_float4(r[4], r[5], r[6], r[7]));
}

int conv3x3_c1(const float* x0, const float* w, const float* bias, float* out, int H, int W, int cout,
               hipStream_t s) {
  if (cout % 8) return LEMO_ERR_SHAPE;
  const int n = H * W * (cout / 8);
  hipLaunchKernelGGL(conv3x3_c1_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x0, w, bias, out, H, W, cout);
  return (int)hipGetLastError();
}

// ---- backward-data of the first layer: Cin(=32) channels of d(pre-activation) -> 1 channel ----
// dx0[y][x] = sum_co sum_tap dpre[co][y-dy][x-dx] * w[co][tap(dy,dx)]   (dx0: unpadded [H*W])
__global__ void __launch_bounds__(256)
conv3x3_c1_bwd_kernel(const float* __restrict__ dpre, const float* __restrict__ w, float* __restrict__ dx0,
                      int H, int W, int cout) {
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int y = p / W, x = p - y * W;
  const int poff = (y + 1) * Wp + (x + 1);
  float a = 0.f;
  for (int g = 0; g < (cout >> 3); ++g) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const float* q = dpre + ((size_t)g * HWp + poff - dy * Wp - dx) * 8;
      const float4 v0 = ld4(q), v1 = ld4(q + 4);
      const float* wc = w + (size_t)(g * 8) * 9 + t;
      a = fmaf(v0.x, wc[0], a); a = fmaf(v0.y, wc[9], a); a = fmaf(v0.z, wc[18], a); a = fmaf(v0.w, wc[27], a);
      a = fmaf(v1.x, wc[36], a); a = fmaf(v1.y, wc[45], a); a = fmaf(v1.z, wc[54], a); a = fmaf(v1.w, wc[63], a);
    }
  }
  dx0[p] = a;
}

int conv3x3_c1_bwd(const float* dpre, const float* w, float* dx0, int H, int W, int cout, hipStream_t s) {
  if (cout % 8) return LEMO_ERR_SHAPE;
  hipLaunchKernelGGL(conv3x3_c1_bwd_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, dpre, w, dx0, H, W, cout);
  return (int)hipGetLastError();
}

// ---- latent smoothness loss (opt_amass_temp.py:390-391) + its gradient, fused ------------------
//   loss = mean_{c,y,x<W-1} (z[c,y,x+1]-z[c,y,x])^2
//   dpre[c,y,x] = coef * 2 * ((z[x]-z[x-1])[x>=1] - (z[x+1]-z[x])[x<=W-2]) * lrelu'(z[c,y,x])
// with coef = weight / (C*H*(W-1)).  Per-block partial sums of the squared differences go to
// `partial[blockIdx.x]` (fixed-order final reduction elsewhere -> deterministic).
__global__ void __launch_bounds__(256)
smooth_loss_kernel(const float* __restrict__ z, float* __restrict__ dpre, float* __restrict__ partial,
                   int H, int W, int C, float coef2) {
  __shared__ float red[4];
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
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
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

int smooth_loss_blocks(int H, int W, int C) { return (H * W * (C / 8) * 2 + 255) / 256; }

int smooth_loss(const float* z, float* dpre, float* partial, int H, int W, int C, float coef2, hipStream_t s) {
  if (C % 8) return LEMO_ERR_SHAPE;
  hipLaunchKernelGGL(smooth_loss_kernel, dim3(smooth_loss_blocks(H, W, C)), dim3(256), 0, s, z, dpre, partial, H, W, C, coef2);
  return (int)hipGetLastError();
}

}  // namespace lemo
