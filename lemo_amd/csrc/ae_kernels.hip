// Building blocks of the motion-infilling autoencoder (models/AE.py:11-108: EncBlock = conv,lrelu,conv,lrelu,
// MaxPool2d(3,2,1); DecBlock = ConvTranspose2d(k3,s2,p1,output_size), lrelu, ConvTranspose2d(k3,s1,p1)[, lrelu])
// and of its per-clip self-supervised finetune step (opt_amass_temp.py:154-214: forward, L1 on the unmasked rows,
// backward incl. WEIGHT gradients, Adam lr 3e-6).  All activations are CG8P (conv_kernels.hip); the 3x3 / stride-1
// convolutions themselves (and the transposed ones, as convolutions with flipped-transposed weights) run on
// conv3x3_mfma.  A stride-2 transposed convolution is "zero-stuff to the output size, then a stride-1 one".
#include "kernels.hpp"

namespace lemo {

// ---- MaxPool2d(kernel 3, stride 2, padding 1) on CG8P; idx = winning tap 0..8 (first maximum in
// row-major window order, like torch) or 255 -------------------------------------------------------
__global__ void __launch_bounds__(256)
maxpool3s2_fwd_kernel(const float* __restrict__ in, int H, int W, float* __restrict__ out, unsigned char* __restrict__ idx,
                      int Ho, int Wo, int C, size_t cs) {
  { const size_t o_ = (size_t)blockIdx.y * cs; in += o_; out += o_; idx += 4 * o_; }         // clip = blockIdx.y (cs in floats; idx is bytes)
  const int Wp = W + 2, HWp = (H + 2) * Wp, Wop = Wo + 2, HWop = (Ho + 2) * Wop;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = (C >> 3) * Ho * Wo;
  if (t >= n) return;
  const int g = t / (Ho * Wo), p = t - g * Ho * Wo, yo = p / Wo, xo = p - yo * Wo;
  float best[8];
  unsigned char bi[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { best[c] = -3.402823466e38f; bi[c] = 255; }
  for (int ky = 0; ky < 3; ++ky) {
    const int y = 2 * yo - 1 + ky;
    if (y < 0 || y >= H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int x = 2 * xo - 1 + kx;
      if (x < 0 || x >= W) continue;
      const float* q = in + ((size_t)g * HWp + (y + 1) * Wp + (x + 1)) * 8;
      const float4 v0 = ld4(q), v1 = ld4(q + 4);
      const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (v[c] > best[c]) { best[c] = v[c]; bi[c] = (unsigned char)(ky * 3 + kx); }
    }
  }
  float* o = out + ((size_t)g * HWop + (yo + 1) * Wop + (xo + 1)) * 8;
  st4(o, make_float4(best[0], best[1], best[2], best[3]));
  st4(o + 4, make_float4(best[4], best[5], best[6], best[7]));
  uint2 iw = make_uint2(0u, 0u);
#pragma unroll
  for (int c = 0; c < 4; ++c) { iw.x |= (unsigned)bi[c] << (8 * c); iw.y |= (unsigned)bi[4 + c] << (8 * c); }
  *reinterpret_cast<uint2*>(idx + ((size_t)g * Ho * Wo + p) * 8) = iw;            // (little-endian: byte c = channel c)
}

// din[y][x] = sum over the <= 4 windows covering (y,x) whose argmax is (y,x) of dout ; optionally
// multiplied by lrelu'(act[y][x]) (act = the pooled layer's input = a LeakyReLU output)
__global__ void __launch_bounds__(256)
maxpool3s2_bwd_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ idx, int Ho, int Wo,
                      const float* __restrict__ act, float* __restrict__ din, int H, int W, int C, size_t cs) {
  { const size_t o_ = (size_t)blockIdx.y * cs; dout += o_; idx += 4 * o_; din += o_; if (act) act += o_; }
  const int Wp = W + 2, HWp = (H + 2) * Wp, Wop = Wo + 2, HWop = (Ho + 2) * Wop;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = (C >> 3) * H * W;
  if (t >= n) return;
  const int g = t / (H * W), p = t - g * H * W, y = p / W, x = p - y * W;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  for (int yo = (y >> 1); yo <= ((y + 1) >> 1); ++yo) {            // windows with 2*yo-1 <= y <= 2*yo+1
    if (yo < 0 || yo >= Ho) continue;
    const int ky = y - (2 * yo - 1);
    if (ky < 0 || ky > 2) continue;
    for (int xo = (x >> 1); xo <= ((x + 1) >> 1); ++xo) {
      if (xo < 0 || xo >= Wo) continue;
      const int kx = x - (2 * xo - 1);
      if (kx < 0 || kx > 2) continue;
      const unsigned want = (unsigned)(ky * 3 + kx);
      // the window's 8 winners and 8 gradients in three loads (8 byte loads + 8 dword loads measured 8.6 us per launch)
      const uint2 iw = *reinterpret_cast<const uint2*>(idx + ((size_t)g * Ho * Wo + yo * Wo + xo) * 8);
      const float* q = dout + ((size_t)g * HWop + (yo + 1) * Wop + (xo + 1)) * 8;
      const float4 q0 = ld4(q), q1 = ld4(q + 4);
      const float qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const unsigned w8 = ((c < 4 ? iw.x : iw.y) >> (8 * (c & 3))) & 0xffu;
        if (w8 == want) acc[c] += qv[c];
      }
    }
  }
  const size_t o = ((size_t)g * HWp + (y + 1) * Wp + (x + 1)) * 8;
  if (act) {
    const float4 a0 = ld4(act + o), a1 = ld4(act + o + 4);
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] *= lrelu_grad_from_out(av[c]);
  }
  st4(din + o, make_float4(acc[0], acc[1], acc[2], acc[3]));
  st4(din + o + 4, make_float4(acc[4], acc[5], acc[6], acc[7]));
}

int maxpool3s2_fwd(const float* in, int H, int W, float* out, unsigned char* idx, int C, hipStream_t s, int nclip, size_t cs) {
  if (C % 8 || H < 1 || W < 1) return LEMO_ERR_SHAPE;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;             // floor((H + 2 - 3) / 2) + 1
  const int n = (C / 8) * Ho * Wo;
  hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3((n + 255) / 256, nclip), dim3(256), 0, s, in, H, W, out, idx, Ho, Wo, C, cs);
  return (int)hipGetLastError();
}
int maxpool3s2_bwd(const float* dout, const unsigned char* idx, const float* act, float* din, int H, int W, int C, hipStream_t s, int nclip, size_t cs) {
  if (C % 8) return LEMO_ERR_SHAPE;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int n = (C / 8) * H * W;
  hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3((n + 255) / 256, nclip), dim3(256), 0, s, dout, idx, Ho, Wo, act, din, H, W, C, cs);
  return (int)hipGetLastError();
}

// ---- zero-stuffing for ConvTranspose2d(stride 2): out[2i][2j] = in[i][j], everything else 0 (out: H x W given
// by output_size); backward = the gather out[2i][2j] (optionally times lrelu'(act[i][j])) ----------------
__global__ void __launch_bounds__(256)
stuff2_fwd_kernel(const float* __restrict__ in, int h, int w, float* __restrict__ out, int H, int W, int C, size_t cs) {
  { const size_t o_ = (size_t)blockIdx.y * cs; in += o_; out += o_; }
  const int Wp = W + 2, HWp = (H + 2) * Wp, wp = w + 2, hwp = (h + 2) * wp;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = (C >> 3) * H * W;
  if (t >= n) return;
  const int g = t / (H * W), p = t - g * H * W, y = p / W, x = p - y * W;
  float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
  if (!(y & 1) && !(x & 1) && (y >> 1) < h && (x >> 1) < w) {
    const float* q = in + ((size_t)g * hwp + ((y >> 1) + 1) * wp + ((x >> 1) + 1)) * 8;
    v0 = ld4(q); v1 = ld4(q + 4);
  }
  float* o = out + ((size_t)g * HWp + (y + 1) * Wp + (x + 1)) * 8;
  st4(o, v0); st4(o + 4, v1);
}
__global__ void __launch_bounds__(256)
stuff2_bwd_kernel(const float* __restrict__ dout, int H, int W, const float* __restrict__ act, float* __restrict__ din,
                  int h, int w, int C) {
  const int Wp = W + 2, HWp = (H + 2) * Wp, wp = w + 2, hwp = (h + 2) * wp;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = (C >> 3) * h * w;
  if (t >= n) return;
  const int g = t / (h * w), p = t - g * h * w, i = p / w, j = p - i * w;
  const float* q = dout + ((size_t)g * HWp + (2 * i + 1) * Wp + (2 * j + 1)) * 8;
  float4 v0 = ld4(q), v1 = ld4(q + 4);
  const size_t o = ((size_t)g * hwp + (i + 1) * wp + (j + 1)) * 8;
  if (act) {
    const float4 a0 = ld4(act + o), a1 = ld4(act + o + 4);
    v0.x *= lrelu_grad_from_out(a0.x); v0.y *= lrelu_grad_from_out(a0.y); v0.z *= lrelu_grad_from_out(a0.z); v0.w *= lrelu_grad_from_out(a0.w);
    v1.x *= lrelu_grad_from_out(a1.x); v1.y *= lrelu_grad_from_out(a1.y); v1.z *= lrelu_grad_from_out(a1.z); v1.w *= lrelu_grad_from_out(a1.w);
  }
  st4(din + o, v0); st4(din + o + 4, v1);
}
int stuff2_fwd(const float* in, int h, int w, float* out, int H, int W, int C, hipStream_t s, int nclip, size_t cs) {
  if (C % 8 || 2 * (h - 1) > H - 1 || 2 * (w - 1) > W - 1) return LEMO_ERR_SHAPE;
  const int n = (C / 8) * H * W;
  hipLaunchKernelGGL(stuff2_fwd_kernel, dim3((n + 255) / 256, nclip), dim3(256), 0, s, in, h, w, out, H, W, C, cs);
  return (int)hipGetLastError();
}
int stuff2_bwd(const float* dout, int H, int W, const float* act, float* din, int h, int w, int C, hipStream_t s) {
  if (C % 8) return LEMO_ERR_SHAPE;
  const int n = (C / 8) * h * w;
  hipLaunchKernelGGL(stuff2_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dout, H, W, act, din, h, w, C);
  return (int)hipGetLastError();
}

// ---- weight gradient of a 3x3 / stride-1 / pad-1 convolution on the matrix cores --------------------------
//   dW[co][ci][tap] = sum_p dY[co][p] * X[ci][p + tap]      (X zero-padded: CG8P border)
// GEMM per tap: M = co (A = dY), N = ci (B = X shifted), K = pixels.  One workgroup = one 32(co) x 32(ci) tile
// of one tap over a slab of 512 pixels, its four waves a quarter of the slab each (summed through LDS in wave order);
// v_mfma_f32_32x32x2_f32 consumes 2 pixels per step (lane half = pixel parity), 8 steps in flight.  Slabs write partial
// tiles; a second kernel reduces them in a fixed order (deterministic) and also produces the bias gradient sum_p dY[co][p].
// (One wave per tile over the whole slab -- round 1 -- left the 20 launches of an AE step at 126-144 workgroups of a
// 256-MFMA dependent chain each: 29 us per launch, 0.59 of the step's 1.7 ms, on two waves per CU.)
#define WG_SLAB 512               // pixels per slab
__global__ void __launch_bounds__(256)
conv3x3_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int H, int W, unsigned wmagic, int cin, int cout,
                     float* __restrict__ partial /*[nslab][9][cout][cin]*/, int nslab) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, kk = lane >> 5;
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const int cot = cout >> 5, cit = (cin + 31) >> 5;
  __shared__ float red[3][16][64];
  int tile = blockIdx.x;                                           // (slab, tap, co tile, ci tile); grid = their count
  const int ct = tile % cit; tile /= cit;
  const int mt = tile % cot; tile /= cot;
  const int tap = tile % 9, slab = tile / 9;
  const int dyo = tap / 3 - 1, dxo = tap % 3 - 1;
  const int co = mt * 32 + i;
  int ci = ct * 32 + i;
  const bool ci_ok = ci < cin;
  if (!ci_ok) ci = cin - 1;
  const float* ap = dy + ((size_t)(co >> 3) * HWp) * 8 + (co & 7);
  const float* bp = x + ((size_t)(ci >> 3) * HWp) * 8 + (ci & 7);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int q0 = slab * WG_SLAB + wave * (WG_SLAB / 4);            // this wave's quarter (empty past the image end)
  const int p0 = q0 < P ? q0 : P, p1 = (q0 + WG_SLAB / 4 < P) ? q0 + WG_SLAB / 4 : P;
  // operands of step pb + 16 are requested before the 8 MFMAs of step pb (two register sets): without the prefetch
  // every step exposed a full memory round trip of 16 strided dword loads (72 us per launch, 40 % of the AE step)
  float a[2][8], b[2][8];
#define WG_LOAD(SET, PB)                                                                           \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                  \
    const int p = (PB) + 2 * u + kk;                                                               \
    const bool ok = p < p1;                                                                        \
    const int pc = ok ? p : p1 - 1;                                                                \
    const int y = (int)__umulhi((unsigned)pc, wmagic), xx = pc - y * W;   /* p / W, host-made magic */ \
    const int q = (y + 1) * Wp + (xx + 1);                                                         \
    const float av = ap[(size_t)q * 8];                                                            \
    const float bv = bp[(size_t)(q + dyo * Wp + dxo) * 8];                                         \
    a[SET][u] = ok ? av : 0.f;                                                                     \
    b[SET][u] = (ok && ci_ok) ? bv : 0.f;                                                          \
  }
#define WG_MFMA(SET)                                                                               \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[SET][u], b[SET][u], acc, 0, 0, 0);
  WG_LOAD(0, p0)
  for (int pb = p0; pb < p1; pb += 32) {
    if (pb + 16 < p1) { WG_LOAD(1, pb + 16) }
    __builtin_amdgcn_sched_barrier(0);
    WG_MFMA(0)
    __builtin_amdgcn_sched_barrier(0);
    if (pb + 16 < p1) {
      if (pb + 32 < p1) { WG_LOAD(0, pb + 32) }
      __builtin_amdgcn_sched_barrier(0);
      WG_MFMA(1)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef WG_LOAD
#undef WG_MFMA
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[w][r][lane];
  // D: col = lane&31 -> ci (B index), rows -> co (A index)
  float* out = partial + (((size_t)slab * 9 + tap) * cout) * cin;
  if (ci_ok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
      out[(size_t)(mt * 32 + row) * cin + (ct * 32 + i)] = acc[r];
    }
  }
}

// dw[co][ci][tap] (torch conv layout) = sum_slab partial ; db[co] = sum_p dy[co][p]
__global__ void __launch_bounds__(256)
conv3x3_wgrad_reduce_kernel(const float* __restrict__ partial, int nslab, int cin, int cout, int cin_real, int cout_real,
                            float* __restrict__ dw, const float* __restrict__ dy, int H, int W, float* __restrict__ db) {
  __shared__ float red[4];
  const int n = cout_real * cin_real * 9;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x < (unsigned)((n + 255) / 256)) {
    if (t < n) {
      const int tap = t % 9, ci = (t / 9) % cin_real, co = t / (9 * cin_real);
      float a = 0.f;
      for (int s = 0; s < nslab; ++s) a += partial[(((size_t)s * 9 + tap) * cout + co) * cin + ci];
      dw[t] = a;
    }
  } else if (db) {                                                  // one block per output channel
    const int co = blockIdx.x - (n + 255) / 256;
    const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
    float a = 0.f;
    for (int p = threadIdx.x; p < P; p += 256) {
      const int y = p / W, xx = p - y * W;
      a += dy[((size_t)(co >> 3) * HWp + (y + 1) * Wp + (xx + 1)) * 8 + (co & 7)];
    }
    a = block_sum(a, red);
    if (threadIdx.x == 0) db[co] = a;
  }
}

int conv3x3_wgrad_nslab(int H, int W) { return (H * W + WG_SLAB - 1) / WG_SLAB; }

// All weight-gradient reductions of a training step in ONE launch (the AE has 20 layers: 20 reduce launches of ~12 us, most of
// it launch latency, become one).  Same per-element arithmetic as conv3x3_wgrad_reduce_kernel (slab order), so the
// gradients have the same bits.  jobs: by value in the kernel argument (<= LEMO_WGRAD_MAX_JOBS).
struct WgradJobs { lemo_wgrad_job j[LEMO_WGRAD_MAX_JOBS]; int first_block[LEMO_WGRAD_MAX_JOBS + 1]; int n; };
__global__ void __launch_bounds__(256)
conv3x3_wgrad_reduce_multi_kernel(WgradJobs J) {
  __shared__ float red[4];
  int k = 0;
  while (k + 1 < J.n && (int)blockIdx.x >= J.first_block[k + 1]) ++k;         // block -> job (uniform)
  const lemo_wgrad_job& q = J.j[k];
  const int blk = (int)blockIdx.x - J.first_block[k];
  const int n = q.cout_real * q.cin_real * 9, nb = (n + 255) / 256;
  if (blk < nb) {
    // thread -> (tap, co, ci) with ci fastest = the memory order of the partials (coalesced reads, 36-byte-strided writes of
    // a tenth of the volume; tap-fastest read every slab with a cout x cin stride between lanes: 71 us for 40 MB)
    const int t = blk * 256 + (int)threadIdx.x;
    if (t < n) {
      const int ci = t % q.cin_real, co = (t / q.cin_real) % q.cout_real, tap = t / (q.cin_real * q.cout_real);
      float a = 0.f;
      for (int s = 0; s < q.nslab; ++s) a += q.partial[(((size_t)s * 9 + tap) * q.cout + co) * q.cin + ci];
      q.dw[((size_t)co * q.cin_real + ci) * 9 + tap] = a;
    }
  } else if (q.db) {
    // bias gradient of one channel: eight independent loads in flight per thread (one load per trip was a chain of 110
    // exposed round trips at 210 x 135: 55 of this launch's 71 us)
    const int co = blk - nb;
    const int Wp = q.W + 2, HWp = (q.H + 2) * Wp, P = q.H * q.W;
    const float* base = q.dy + (size_t)(co >> 3) * HWp * 8 + (co & 7);
    float a = 0.f;
    for (int p0 = threadIdx.x; p0 < P; p0 += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + 256 * u, pc = p < P ? p : P - 1;
        const int y = pc / q.W, xx = pc - y * q.W;
        v[u] = base[(size_t)((y + 1) * Wp + (xx + 1)) * 8];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) a += (p0 + 256 * u < P) ? v[u] : 0.f;
    }
    a = block_sum(a, red);
    if (threadIdx.x == 0) q.db[co] = a;
  }
}

int conv3x3_wgrad_partial(const float* dy, const float* x, int H, int W, int cin, int cout, float* partial, hipStream_t s) {
  if (cin % 8 || cout % 32 || H < 1 || W < 1 || (long)H * W > (1l << 24)) return LEMO_ERR_SHAPE;
  const int nslab = conv3x3_wgrad_nslab(H, W);
  const unsigned wmagic = (unsigned)((1ull << 32) / (unsigned)W + 1);
  const int ntile = nslab * 9 * (cout / 32) * ((cin + 31) / 32);
  hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3(ntile), dim3(256), 0, s, dy, x, H, W, wmagic, cin, cout, partial, nslab);
  return (int)hipGetLastError();
}

int conv3x3_wgrad_reduce_multi(const lemo_wgrad_job* jobs, int n, hipStream_t s) {
  if (!jobs || n < 1 || n > LEMO_WGRAD_MAX_JOBS) return LEMO_ERR_ARG;
  WgradJobs J;
  J.n = n;
  int nb = 0;
  for (int k = 0; k < n; ++k) {
    const lemo_wgrad_job& q = jobs[k];
    if (!q.partial || !q.dw || !q.dy || q.cin % 8 || q.cout % 32 || q.cin_real > q.cin || q.cout_real > q.cout || q.H < 1 || q.W < 1 ||
        q.nslab != conv3x3_wgrad_nslab(q.H, q.W)) return LEMO_ERR_SHAPE;
    J.j[k] = q;
    J.first_block[k] = nb;
    nb += (q.cout_real * q.cin_real * 9 + 255) / 256 + (q.db ? q.cout_real : 0);
  }
  J.first_block[n] = nb;
  hipLaunchKernelGGL(conv3x3_wgrad_reduce_multi_kernel, dim3(nb), dim3(256), 0, s, J);
  return (int)hipGetLastError();
}

// dy: CG8P with `cout` (multiple of 32) channels, x: CG8P with `cin` (multiple of 8) channels; dw [cout_real][cin_real][3][3]
int conv3x3_wgrad(const float* dy, const float* x, int H, int W, int cin, int cout, int cin_real, int cout_real,
                  float* partial, float* dw, float* db, hipStream_t s) {
  if (cin % 8 || cout % 32 || cin_real > cin || cout_real > cout || H < 1 || W < 1 || (long)H * W > (1l << 24)) return LEMO_ERR_SHAPE;
  const int nslab = conv3x3_wgrad_nslab(H, W);
  const unsigned wmagic = (unsigned)((1ull << 32) / (unsigned)W + 1);          // exact for p < 2^32 / W
  const int ntile = nslab * 9 * (cout / 32) * ((cin + 31) / 32);
  hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3(ntile), dim3(256), 0, s, dy, x, H, W, wmagic, cin, cout, partial, nslab);
  int e = (int)hipGetLastError();
  if (e) return e;
  const int n = cout_real * cin_real * 9;
  hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((n + 255) / 256 + (db ? cout_real : 0)), dim3(256), 0, s, partial, nslab,
                     cin, cout, cin_real, cout_real, dw, dy, H, W, db);
  return (int)hipGetLastError();
}

// ---- Adam over one flat parameter buffer (torch.optim.Adam defaults; opt_amass_temp.py:162-164: lr 3e-6) ----
// `step_dev` (optional): the 1-based step is read from device memory (completed steps + 1) so that a captured
// graph of the training step can be replayed; the counter is advanced by adam_count_kernel AFTER this kernel.
__global__ void __launch_bounds__(256)
adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int n,
                 double lr, int step /*1-based*/, const int* __restrict__ step_dev) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (step_dev) step = step_dev[0] + 1;
  float pv = p[i], mi = m[i], vi = v[i];
  adam_update_torch(pv, mi, vi, g[i], adam_coef_t(step, lr));       // common.hpp: torch.optim.Adam's own evaluation order
  m[i] = mi; v[i] = vi;
  p[i] = pv;
}
__global__ void adam_count_kernel(int* ctr) { if (threadIdx.x == 0 && blockIdx.x == 0) ctr[0] += 1; }
int adam_flat(float* p, const float* g, float* m, float* v, int n, float lr, int step, int* step_dev, hipStream_t s) {
  if (n <= 0 || (!step_dev && step < 1)) return LEMO_ERR_ARG;
  hipLaunchKernelGGL(adam_flat_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, g, m, v, n, lr_decimal(lr), step, (const int*)step_dev);
  if (step_dev) hipLaunchKernelGGL(adam_count_kernel, dim3(1), dim3(64), 0, s, step_dev);
  return (int)hipGetLastError();
}

}  // namespace lemo
