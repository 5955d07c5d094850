// Native training-step engine of the motion-infilling autoencoder (models/AE.py:11-108) for its per-clip self-supervised
// finetune (opt_amass_temp.py:154-214, temp_prox/fitting_temp_slide.py:861-893: 60 x [forward, L1 on the unmasked rows,
// backward, Adam lr 3e-6] + one eval forward at [1,4,210,135]).
//
// Why an engine: the step is 20 convolutions of 126 .. 28350 pixels with 32 .. 256 channels -- 17 GFLOP, ~0.1 ms of fp32
// MFMA time -- so its cost is the NUMBER of dependent launches and what each leaves idle.  Round 2 ran it as ~150 launches
// (autograd function + packing gathers, 1.3 ms); here one step is 53:
//   * one convolution kernel for every layer and direction: the workgroup's 4 .. 16 waves split K = 9 Cin between them and
//     reduce through LDS in wave order (no split-K partials in HBM, no combine launch), epilogue fused; the kernel takes
//     separate input / output / epilogue-operand geometries, so that a decoder block's output is written straight into the
//     zero-stuffed input of the next block's stride-2 transposed convolution, and the adjoint of the stuffing (gather of the
//     even pixels, times lrelu') is the epilogue of a backward convolution that only enumerates those pixels;
//   * the 20 weight gradients need d(pre-activation) of every layer and nothing else: they run after the backward-data
//     chain as ONE launch over all layers' (slab, tap, tile) work items (the chip is full; 20 launches of 126..576
//     workgroups each were not), writing slab partials in the layout of the parameter vector;
//   * parameters, Adam moments and gradients live in the kernels' packed forward layout (Adam is elementwise: the order
//     is free), so the optimizer launch sums the slab partials in slab order, updates, and scatters the new value into the
//     backward pack: no gathers between "the model's tensors" and "the kernels' operands" inside the loop;
//   * loss gradient sign(rec - x) * mask / count in closed form (one launch, which also advances Adam's step counter and
//     bias corrections on the device so that a captured step can be replayed).
// Arithmetic: fp32 MFMA (v_mfma_f32_32x32x2_f32), every sum in a fixed order -> results do not depend on concurrency.
#include "conv_common.hpp"
#include "conv_f16.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#define CHK_(e) do { int _e = (e); if (_e) return _e; } while (0)

// slots of AeEngine::amax.  0 .. AE_SLOT_DYN - 1 are rewritten every step (zeroed by the optimiser launch and at load)
#define AE_SLOT_ACT(i) (i)                /* max |output of layer i| (forward) */
#define AE_SLOT_DP(i) (20 + (i))          /* max |d(pre-activation) of layer i| where a convolution wrote it */
#define AE_SLOT_DPOOL(b) (40 + (b))       /* max |d(pooled output of encoder block b)| */
#define AE_SLOT_DYN 45
#define AE_SLOT_X8 45                     /* the clip image */
#define AE_SLOT_LOSS 46                   /* max (mask / count): the loss gradient is +-1 x that */
#define AE_SLOT_W(i) (48 + (i))           /* max |weights of layer i| (forward and backward pack hold the same values) */
#define AE_SLOT_SCRATCH 70
#define AE_NSLOT 128

namespace lemo {

// ---------------------------------------------------------------------------------------------------------------------
// convolution, K split over the waves of the workgroup
// ---------------------------------------------------------------------------------------------------------------------
// A pixel (y, x) of the H x W grid the launch enumerates sits at padded pixel ((s y + 1) Wp + s x + 1) of a buffer with
// row pitch Wp and stride s (1: plain CG8P of an H x W image; 2: the even pixels of a twice finer image = zero-stuffing
// geometry).  Taps always address neighbouring pixels of the INPUT buffer.
struct AeGeo { int H, W; int in_Wp, in_HWp, in_s; int out_Wp, out_HWp, out_s; int aux_Wp, aux_HWp, aux_s; };

// Clips side by side (round 4): an engine of K clips carves K identical workspaces, `cs` floats apart, and every
// launch of a training step carries the clip as its last grid dimension -- the same kernels, the same launch shapes, K times the
// blocks.  Each clip has its OWN parameters, Adam state and step counter (the reference finetunes a fresh copy per clip), so a
// clip's arithmetic does not depend on its neighbours (only on HOW MANY ride along: ae_conv_shape picks the launch shape for the clips in flight).  Null operands (an unused bias / aux) stay null.
#define AE_CLIP_OFFSET5(in_, wt_, bias_, aux_, out_, cs_)                                             \
  { const size_t o_ = (size_t)blockIdx.z * (cs_); in_ += o_; wt_ += o_; out_ += o_;                   \
    if (bias_) bias_ += o_; if (aux_) aux_ += o_; }

// maximum of |v| over the wave's stored lanes -> the launch's slot (bit pattern of a non-negative float orders like the float)
// (one atomic per wave on ONE word cost a launch ~11 ns each -- 2.7 k waves = the whole 29 us of a launch, which is how the first version
// of these kernels ran no faster than the fp32-input ones, and 49 vs 29 ms per solo clip: the slot is READ first and the atomic issued
// only by a wave that would raise it -- a handful per launch; a stale read only costs a redundant atomic)
__device__ __forceinline__ void ae_amax_publish(float m, float* slot) {
  m = wave_max(m);
  unsigned* us = reinterpret_cast<unsigned*>(slot);            // (non-negative floats order like their bit patterns)
  const unsigned um = __builtin_bit_cast(unsigned, m);
  if ((threadIdx.x & 63) == 0 && um > __hip_atomic_load(us, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(us, um);
}

template <int MT, int EPI>          // MT x 32 couts per workgroup; EPI: conv_common.hpp (0 lrelu(acc + bias), 1 acc * lrelu'(aux), 2 acc + bias)
__global__ void __launch_bounds__(1024)
ae_conv_kernel(const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
               const float* __restrict__ aux, float* __restrict__ out, AeGeo g, int cin_lg, int cout, int pt_lg, size_t cs,
               float* __restrict__ amax_out /* may be null: max |out| of the launch, for a split-f16 consumer (ae_conv_f16_kernel) */) {
  AE_CLIP_OFFSET5(in, wt, bias, aux, out, cs);                 // clip = blockIdx.z: every operand lives in that clip's workspace
  if (amax_out) amax_out += (size_t)blockIdx.z * cs;
  LEMO_DYN_SMEM(red);                                          // [wave][MT][4][64 lanes] float4
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;   // (scalar: see the ring below)
  // wave -> (pixel tile pt of the workgroup's 2^pt_lg, K slice ks of KS): the waves of one slice share the weights they load
  // (one L2 -> L1 fill per workgroup instead of one per 32 pixels), the waves of one tile split K
  const int PT = 1 << pt_lg, pt = wave & (PT - 1), ks = wave >> pt_lg, KS = NW >> pt_lg;
  const int j = lane & 31, h = lane >> 5;
  const int P = g.H * g.W;
  const int m_base = blockIdx.y * (MT * 32);
  const int p = (blockIdx.x * PT + pt) * 32 + j;
  const int pc = p < P ? p : P - 1;
  const int y = pc / g.W, x = pc - y * g.W;
  const float* in_l = in + (size_t)((g.in_s * y + 1) * g.in_Wp + g.in_s * x + 1) * 8 + 4 * h;
  const float* wt_l = wt + (size_t)(m_base + j) * 8 + 4 * h;
  const size_t in_gstride = (size_t)g.in_HWp * 8, wt_itstride = (size_t)cout * 8;
  const int gm = (1 << cin_lg) - 1, nit = 9 << cin_lg;
  const int lo = nit * ks / KS, hi = nit * (ks + 1) / KS;        // this wave's (tap, channel group) steps; host: KS <= nit

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  // Operands D - 1 steps ahead through a D-deep register ring (slots are compile-time indices: the step loop is unrolled by D).
  // A step is 4 MT MFMAs = 256 MT cycles of pipe time but its operands come from L2 / the fabric (every workgroup reads its own
  // slice of the weights once): ~1.4 us per round trip measured -- with two loads in flight (round 3's first version) a wave
  // advanced one step per 0.7 us whatever its MFMAs cost, 16 us for the 18-step waves of the 256-channel layers.
  constexpr int D = MT == 1 ? 8 : 5;
  float4 ra[D][MT], rb[D];
#define AE_LD(IT, SLOT)                                                                                     \
  {                                                                                                         \
    const int it_ = (IT);                                                                                   \
    const int tap_ = it_ >> cin_lg, g_ = it_ & gm;                                                          \
    const int dy_ = (tap_ * 11 >> 5) - 1, dx_ = tap_ - (dy_ + 1) * 3 - 1;                                   \
    rb[SLOT] = ld4(in_l + (std::ptrdiff_t)(dy_ * g.in_Wp + dx_) * 8 + (size_t)g_ * in_gstride);             \
    _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_) ra[SLOT][m_] = ld4(wt_l + (size_t)it_ * wt_itstride + (size_t)m_ * 256); \
  }
#define AE_MF(SLOT)                                                                                         \
  _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_) {                                                      \
    acc[m_] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[SLOT][m_].x, rb[SLOT].x, acc[m_], 0, 0, 0);           \
    acc[m_] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[SLOT][m_].y, rb[SLOT].y, acc[m_], 0, 0, 0);           \
    acc[m_] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[SLOT][m_].z, rb[SLOT].z, acc[m_], 0, 0, 0);           \
    acc[m_] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[SLOT][m_].w, rb[SLOT].w, acc[m_], 0, 0, 0);           \
  }
  // Every load is unconditional (indices past the wave's last step re-read that step: at most D - 1 redundant L1 hits) and the
  // branches are scalar, so that hipcc can count the loads in flight and wait for exactly the oldest (vmcnt(N)); loads under
  // an exec-mask branch made it drain the whole ring (vmcnt(0)) once per trip.
#pragma unroll
  for (int d = 0; d < D - 1; ++d) AE_LD(lo + d < hi ? lo + d : hi - 1, d)
  int it = lo;
  for (; it + D <= hi; it += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      AE_LD(it + u + D - 1 < hi ? it + u + D - 1 : hi - 1, (u + D - 1) % D)
      __builtin_amdgcn_sched_barrier(0);                         // the load above is issued before these MFMAs, not sunk below them
      AE_MF(u)
    }
  }
#pragma unroll
  for (int u = 0; u < D - 1; ++u)                                // the last < D steps: their operands are already in the ring
    if (it + u < hi) AE_MF(u)
#undef AE_LD
#undef AE_MF

  float4* r4 = reinterpret_cast<float4*>(red);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      r4[((wave * MT + m) * 4 + q) * 64 + lane] = make_float4(acc[m][4 * q], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]);
  __syncthreads();
  float mloc = 0.f;
  // unit u -> (cout tile m, row quad q) of this lane's pixel; the KS waves of the pixel tile share its 4 MT units
  if (p < P)
  for (int u = ks; u < 4 * MT; u += KS) {
    const int m = u >> 2, q = u & 3;
    float4 v = r4[((pt * MT + m) * 4 + q) * 64 + lane];
    for (int k2 = 1; k2 < KS; ++k2) {                            // slice order: deterministic
      const float4 t = r4[((((k2 << pt_lg) + pt) * MT + m) * 4 + q) * 64 + lane];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const int c0 = m_base + m * 32 + q * 8 + 4 * h;              // first of 4 consecutive couts
    const size_t o = ((size_t)(c0 >> 3) * g.out_HWp + (size_t)((g.out_s * y + 1) * g.out_Wp + g.out_s * x + 1)) * 8 + (c0 & 7);
    if (EPI == 0 || EPI == 2) {
      const float4 bb = ld4(bias + c0);
      v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
      if (EPI == 0) { v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w); }
    } else {
      const size_t oa = ((size_t)(c0 >> 3) * g.aux_HWp + (size_t)((g.aux_s * y + 1) * g.aux_Wp + g.aux_s * x + 1)) * 8 + (c0 & 7);
      const float4 yy = ld4(aux + oa);
      v.x *= lrelu_grad_from_out(yy.x); v.y *= lrelu_grad_from_out(yy.y);
      v.z *= lrelu_grad_from_out(yy.z); v.w *= lrelu_grad_from_out(yy.w);
    }
    st4(out + o, v);
    mloc = absmax4(v, mloc);
  }
  if (amax_out) ae_amax_publish(mloc, amax_out);              // (uniform: every lane of every wave arrives here)
}

// The same convolution on 16 px x 16 cout tiles (v_mfma_f32_16x16x4_f32, same flops per cycle): the 256-channel layers at
// 27 x 17 and 14 x 9 pixels are 120 and 32 tiles of 32 x 32 -- their waves share 120 / 32 of the 256 CUs' MFMA pipes however K
// is cut (measured 15 us per layer, 7.7 of them pipe time on an eighth of the chip) -- and 464 / 128 tiles of 16 x 16.
// Lane map as conv_kernels.hip's tail units: one step = two 8-channel groups, lane quarter q4 reads group 2 gp + (q4 >> 1),
// floats 4 (q4 & 1)..; D: col = lane & 15 -> pixel, rows 4 q4 + r -> cout.
template <int EPI>
__global__ void __launch_bounds__(1024)
ae_conv16_kernel(const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
                 const float* __restrict__ aux, float* __restrict__ out, AeGeo g, int cin_lg, int cout, int pt_lg, size_t cs) {
  AE_CLIP_OFFSET5(in, wt, bias, aux, out, cs);
  LEMO_DYN_SMEM(red);                                          // [wave][64 lanes] float4
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;
  const int PT = 1 << pt_lg, pt = wave & (PT - 1), ks = wave >> pt_lg, KS = NW >> pt_lg;
  const int j = lane & 15, q4 = lane >> 4;
  const int P = g.H * g.W;
  const int m_base = blockIdx.y * 16;
  const int p = (blockIdx.x * PT + pt) * 16 + j;
  const int pc = p < P ? p : P - 1;
  const int y = pc / g.W, x = pc - y * g.W;
  const size_t in_gstride = (size_t)g.in_HWp * 8, wt_itstride = (size_t)cout * 8;
  const float* in_l = in + (size_t)((g.in_s * y + 1) * g.in_Wp + g.in_s * x + 1) * 8 + (size_t)(q4 >> 1) * in_gstride + 4 * (q4 & 1);
  const float* wt_l = wt + ((size_t)(q4 >> 1) * cout + (m_base + j)) * 8 + 4 * (q4 & 1);
  const int gp_lg = cin_lg - 1, gm = (1 << gp_lg) - 1, nit = 9 << gp_lg;         // steps of 16 channels
  const int lo = nit * ks / KS, hi = nit * (ks + 1) / KS;

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int D = 10;
  float4 ra[D], rb[D];
#define AE_LD(IT, SLOT)                                                                                     \
  {                                                                                                         \
    const int it_ = (IT);                                                                                   \
    const int tap_ = it_ >> gp_lg, gp_ = it_ & gm;                                                          \
    const int dy_ = (tap_ * 11 >> 5) - 1, dx_ = tap_ - (dy_ + 1) * 3 - 1;                                   \
    rb[SLOT] = ld4(in_l + (std::ptrdiff_t)(dy_ * g.in_Wp + dx_) * 8 + (size_t)(2 * gp_) * in_gstride);      \
    ra[SLOT] = ld4(wt_l + (size_t)((tap_ << cin_lg) + 2 * gp_) * wt_itstride);                              \
  }
#define AE_MF(SLOT)                                                                                         \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[SLOT].x, rb[SLOT].x, acc, 0, 0, 0);                         \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[SLOT].y, rb[SLOT].y, acc, 0, 0, 0);                         \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[SLOT].z, rb[SLOT].z, acc, 0, 0, 0);                         \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[SLOT].w, rb[SLOT].w, acc, 0, 0, 0);
#pragma unroll
  for (int d = 0; d < D - 1; ++d) AE_LD(lo + d < hi ? lo + d : hi - 1, d)
  int it = lo;
  for (; it + D <= hi; it += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      AE_LD(it + u + D - 1 < hi ? it + u + D - 1 : hi - 1, (u + D - 1) % D)
      __builtin_amdgcn_sched_barrier(0);
      AE_MF(u)
    }
  }
#pragma unroll
  for (int u = 0; u < D - 1; ++u)
    if (it + u < hi) { AE_MF(u) }
#undef AE_LD
#undef AE_MF

  float4* r4 = reinterpret_cast<float4*>(red);
  r4[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  if (ks != 0 || p >= P) return;
  float4 v = r4[pt * 64 + lane];
  for (int k2 = 1; k2 < KS; ++k2) {                              // slice order: deterministic
    const float4 t = r4[((k2 << pt_lg) + pt) * 64 + lane];
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  const int c0 = m_base + 4 * q4;
  const size_t o = ((size_t)(c0 >> 3) * g.out_HWp + (size_t)((g.out_s * y + 1) * g.out_Wp + g.out_s * x + 1)) * 8 + (c0 & 7);
  if (EPI == 0 || EPI == 2) {
    const float4 bb = ld4(bias + c0);
    v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
    if (EPI == 0) { v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w); }
  } else {
    const size_t oa = ((size_t)(c0 >> 3) * g.aux_HWp + (size_t)((g.aux_s * y + 1) * g.aux_Wp + g.aux_s * x + 1)) * 8 + (c0 & 7);
    const float4 yy = ld4(aux + oa);
    v.x *= lrelu_grad_from_out(yy.x); v.y *= lrelu_grad_from_out(yy.y);
    v.z *= lrelu_grad_from_out(yy.z); v.w *= lrelu_grad_from_out(yy.w);
  }
  st4(out + o, v);
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// ---------------------------------------------------------------------------------------------------------------------
// the same convolutions on the f16 matrix cores ("split-f16", round 6): built, parity-tested, SELECTABLE, not the
// default -- see AeEngine::f16 for the measurement
// ---------------------------------------------------------------------------------------------------------------------
// The fp32-input MFMA above issues at 1/16 of the f16 rate.  Here a step is 16 channels x 1 tap: both operands are read as fp32 (the
// activations and the parameter vector stay fp32 -- Adam, the weight gradients and the layouts are untouched), split IN REGISTERS into
// two error-compensated fp16 pieces (conv_f16.hpp: x s = hi + lo, 2 x 11 significand bits) and multiplied with three
// v_mfma_f32_32x32x16_f16 per cout tile (hi lo + lo hi + hi hi, fp32 accumulate): 96 matrix-pipe cycles per 16 channels and tile
// instead of 512.  The power-of-two scales come from TENSOR maxima instead of a workgroup's tile (there is no LDS staging here):
//   activations  every convolution's epilogue leaves max |out| of its launch in a per-clip slot (atomicMax on the bit pattern, one per
//                wave); the consumer scales by it.  Inputs that are not convolution outputs carry a bound instead: max-pool / zero
//                stuffing keep the maximum, the max-pool adjoint sums at most four window gradients (x 4), the clip image and the loss
//                gradient's magnitude (mask / count) are reduced once at load.  A bound that is 2^k too large costs nothing while
//                k <= ~10: an element keeps its full 2^-22 relative precision down to 2^-17 of the scaled maximum, below that its error is
//                2^-40 of the maximum (conv_f16.hpp).
//   weights      max |w| per layer, reduced when a clip's parameters are loaded, with one bit of headroom: the finetune moves a weight by
//                <= 60 x lr 3e-6 x ~10 = 2e-3, far from the factor 2 (fp16 itself overflows another factor 4 later).
struct AeF16 { const float* amax_in; const float* wmax; float* amax_out; float in_fac; };       // per-clip slots (offset by the clip stride like every operand)

// Both operands of a 16-channel step are 32 contiguous bytes per lane (8 fp32 channels of one pixel / one cout) and the wave's two halves
// read two different channel groups.  Read as [16 B | 16 B] per lane, each dwordx4 instruction touches every line of both groups half --
// twice the L1 requests of the fp32-input kernel for the same bytes (the first version of these kernels: no faster than fp32, 45 vs 29 ms
// per solo clip).  Instead instruction g reads group g with lane j taking the FIRST 16 bytes and lane j + 32 the SECOND (1 KiB contiguous
// per instruction), and one v_permlane32_swap per register puts the halves where the MFMA wants them:
//   before: lanes 0-31 hold {G0[0:4], G1[0:4]}, lanes 32-63 {G0[4:8], G1[4:8]}   after: lanes 0-31 {G0[0:4], G0[4:8]}, lanes 32-63 {G1[0:4], G1[4:8]}
__device__ __forceinline__ void ae_swap_halves(float4& g0, float4& g1) {
#define AE_SW(c)                                                                                             \
  { const auto r_ = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, g0.c), __builtin_bit_cast(unsigned, g1.c), false, false); \
    g0.c = __builtin_bit_cast(float, (unsigned)r_[0]); g1.c = __builtin_bit_cast(float, (unsigned)r_[1]); }
  AE_SW(x) AE_SW(y) AE_SW(z) AE_SW(w)
#undef AE_SW
}
__device__ __forceinline__ void ae_split8(float4 lo4, float4 hi4, float s, f16x8& ph, f16x8& pl) {
  uint2 h0, l0, h1, l1;
  split2x4(lo4, s, h0, l0);
  split2x4(hi4, s, h1, l1);
  ph = __builtin_bit_cast(f16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
  pl = __builtin_bit_cast(f16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
}

template <int MT, int EPI>          // MT x 32 couts per workgroup (MT 2: at most 8 waves)
__global__ void __launch_bounds__(MT == 2 ? 512 : 1024)
ae_conv_f16_kernel(const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
                   const float* __restrict__ aux, float* __restrict__ out, AeGeo g, int cin_lg, int cout, int pt_lg, size_t cs, AeF16 q) {
  AE_CLIP_OFFSET5(in, wt, bias, aux, out, cs);
  { const size_t o_ = (size_t)blockIdx.z * cs; q.amax_in += o_; q.wmax += o_; q.amax_out += o_; }
  LEMO_DYN_SMEM(red);                                          // [wave][MT][4][64 lanes] float4
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;
  const int PT = 1 << pt_lg, pt = wave & (PT - 1), ks = wave >> pt_lg, KS = NW >> pt_lg;
  const int j = lane & 31, h = lane >> 5;
  const int P = g.H * g.W;
  const int m_base = blockIdx.y * (MT * 32);
  const int p = (blockIdx.x * PT + pt) * 32 + j;
  const int pc = p < P ? p : P - 1;
  const int y = pc / g.W, x = pc - y * g.W;
  // a step = one tap x 16 channels: lane half h carries channel group 2 gp + h (8 channels = 32 contiguous bytes in both operands)
  const size_t in_gstride = (size_t)g.in_HWp * 8, wt_itstride = (size_t)cout * 8;
  const float* in_l = in + (size_t)((g.in_s * y + 1) * g.in_Wp + g.in_s * x + 1) * 8 + 4 * h;       // lane half h: bytes 16 h .. 16 h + 15 of a group (ae_swap_halves)
  const float* wt_l = wt + (size_t)(m_base + j) * 8 + 4 * h;
  const int gp_lg = cin_lg - 1, gm = (1 << gp_lg) - 1, nit = 9 << gp_lg;
  const int lo = nit * ks / KS, hi = nit * (ks + 1) / KS;
  float sA, sAi, sB, sBi;
  f16_scale_for(q.wmax[0] * 2.f, sA, sAi);
  f16_scale_for(q.amax_in[0] * q.in_fac, sB, sBi);

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  constexpr int D = MT == 1 ? 5 : 4;
  float4 ra[D][MT][2], rb[D][2];
#define AE_LD(IT, SLOT)                                                                                     \
  {                                                                                                         \
    const int it_ = (IT);                                                                                   \
    const int tap_ = it_ >> gp_lg, gp_ = it_ & gm;                                                          \
    const int dy_ = (tap_ * 11 >> 5) - 1, dx_ = tap_ - (dy_ + 1) * 3 - 1;                                   \
    const float* bq_ = in_l + (std::ptrdiff_t)(dy_ * g.in_Wp + dx_) * 8 + (size_t)(2 * gp_) * in_gstride;   \
    rb[SLOT][0] = ld4(bq_); rb[SLOT][1] = ld4(bq_ + in_gstride);                                            \
    const float* aq_ = wt_l + (size_t)((tap_ << cin_lg) + 2 * gp_) * wt_itstride;                           \
    _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_) { ra[SLOT][m_][0] = ld4(aq_ + (size_t)m_ * 256); ra[SLOT][m_][1] = ld4(aq_ + (size_t)m_ * 256 + wt_itstride); } \
  }
#define AE_MF(SLOT)                                                                                         \
  {                                                                                                         \
    f16x8 bh_, bl_;                                                                                         \
    ae_swap_halves(rb[SLOT][0], rb[SLOT][1]);                                                               \
    ae_split8(rb[SLOT][0], rb[SLOT][1], sB, bh_, bl_);                                                      \
    _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_) {                                                     \
      f16x8 ah_, al_;                                                                                       \
      ae_swap_halves(ra[SLOT][m_][0], ra[SLOT][m_][1]);                                                     \
      ae_split8(ra[SLOT][m_][0], ra[SLOT][m_][1], sA, ah_, al_);                                            \
      acc[m_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, bl_, acc[m_], 0, 0, 0);                          \
      acc[m_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_, bh_, acc[m_], 0, 0, 0);                          \
      acc[m_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, bh_, acc[m_], 0, 0, 0);                          \
    }                                                                                                       \
  }
#pragma unroll
  for (int d = 0; d < D - 1; ++d) AE_LD(lo + d < hi ? lo + d : hi - 1, d)
  int it = lo;
  for (; it + D <= hi; it += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      AE_LD(it + u + D - 1 < hi ? it + u + D - 1 : hi - 1, (u + D - 1) % D)
      __builtin_amdgcn_sched_barrier(0);
      AE_MF(u)
    }
  }
#pragma unroll
  for (int u = 0; u < D - 1; ++u)
    if (it + u < hi) AE_MF(u)
#undef AE_LD
#undef AE_MF

  float4* r4 = reinterpret_cast<float4*>(red);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
      r4[((wave * MT + m) * 4 + qd) * 64 + lane] = make_float4(acc[m][4 * qd], acc[m][4 * qd + 1], acc[m][4 * qd + 2], acc[m][4 * qd + 3]);
  __syncthreads();
  const float unscale = sAi * sBi;                               // exact: powers of two
  float mloc = 0.f;
  if (p < P)
  for (int u = ks; u < 4 * MT; u += KS) {
    const int m = u >> 2, qd = u & 3;
    float4 v = r4[((pt * MT + m) * 4 + qd) * 64 + lane];
    for (int k2 = 1; k2 < KS; ++k2) {                            // slice order: deterministic
      const float4 t = r4[((((k2 << pt_lg) + pt) * MT + m) * 4 + qd) * 64 + lane];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    v.x *= unscale; v.y *= unscale; v.z *= unscale; v.w *= unscale;
    const int c0 = m_base + m * 32 + qd * 8 + 4 * h;
    const size_t o = ((size_t)(c0 >> 3) * g.out_HWp + (size_t)((g.out_s * y + 1) * g.out_Wp + g.out_s * x + 1)) * 8 + (c0 & 7);
    if (EPI == 0 || EPI == 2) {
      const float4 bb = ld4(bias + c0);
      v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
      if (EPI == 0) { v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w); }
    } else {
      const size_t oa = ((size_t)(c0 >> 3) * g.aux_HWp + (size_t)((g.aux_s * y + 1) * g.aux_Wp + g.aux_s * x + 1)) * 8 + (c0 & 7);
      const float4 yy = ld4(aux + oa);
      v.x *= lrelu_grad_from_out(yy.x); v.y *= lrelu_grad_from_out(yy.y);
      v.z *= lrelu_grad_from_out(yy.z); v.w *= lrelu_grad_from_out(yy.w);
    }
    st4(out + o, v);
    mloc = absmax4(v, mloc);
  }
  ae_amax_publish(mloc, q.amax_out);
}

// ... and on 16 px x 16 cout tiles (v_mfma_f32_16x16x32_f16: a step = one tap x 32 channels, lane quarter q4 carries channel group 4 gq + q4)
template <int EPI>
__global__ void __launch_bounds__(1024)
ae_conv16_f16_kernel(const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
                     const float* __restrict__ aux, float* __restrict__ out, AeGeo g, int cin_lg, int cout, int pt_lg, size_t cs, AeF16 q) {
  AE_CLIP_OFFSET5(in, wt, bias, aux, out, cs);
  { const size_t o_ = (size_t)blockIdx.z * cs; q.amax_in += o_; q.wmax += o_; q.amax_out += o_; }
  LEMO_DYN_SMEM(red);                                          // [wave][64 lanes] float4
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;
  const int PT = 1 << pt_lg, pt = wave & (PT - 1), ks = wave >> pt_lg, KS = NW >> pt_lg;
  const int j = lane & 15, q4 = lane >> 4;
  const int P = g.H * g.W;
  const int m_base = blockIdx.y * 16;
  const int p = (blockIdx.x * PT + pt) * 16 + j;
  const int pc = p < P ? p : P - 1;
  const int y = pc / g.W, x = pc - y * g.W;
  const size_t in_gstride = (size_t)g.in_HWp * 8, wt_itstride = (size_t)cout * 8;
  const float* in_l = in + (size_t)((g.in_s * y + 1) * g.in_Wp + g.in_s * x + 1) * 8 + (size_t)q4 * in_gstride;
  const float* wt_l = wt + (size_t)(m_base + j) * 8 + (size_t)q4 * wt_itstride;
  const int gq_lg = cin_lg - 2, gm = (1 << gq_lg) - 1, nit = 9 << gq_lg;           // steps of 32 channels (host: cin >= 32)
  const int lo = nit * ks / KS, hi = nit * (ks + 1) / KS;
  float sA, sAi, sB, sBi;
  f16_scale_for(q.wmax[0] * 2.f, sA, sAi);
  f16_scale_for(q.amax_in[0] * q.in_fac, sB, sBi);

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int D = 6;
  float4 ra[D][2], rb[D][2];
#define AE_LD(IT, SLOT)                                                                                     \
  {                                                                                                         \
    const int it_ = (IT);                                                                                   \
    const int tap_ = it_ >> gq_lg, gq_ = it_ & gm;                                                          \
    const int dy_ = (tap_ * 11 >> 5) - 1, dx_ = tap_ - (dy_ + 1) * 3 - 1;                                   \
    const float* bq_ = in_l + (std::ptrdiff_t)(dy_ * g.in_Wp + dx_) * 8 + (size_t)(4 * gq_) * in_gstride;   \
    rb[SLOT][0] = ld4(bq_); rb[SLOT][1] = ld4(bq_ + 4);                                                     \
    const float* aq_ = wt_l + (size_t)((tap_ << cin_lg) + 4 * gq_) * wt_itstride;                           \
    ra[SLOT][0] = ld4(aq_); ra[SLOT][1] = ld4(aq_ + 4);                                                     \
  }
#define AE_MF(SLOT)                                                                                         \
  {                                                                                                         \
    f16x8 bh_, bl_, ah_, al_;                                                                               \
    ae_split8(rb[SLOT][0], rb[SLOT][1], sB, bh_, bl_);                                                      \
    ae_split8(ra[SLOT][0], ra[SLOT][1], sA, ah_, al_);                                                      \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah_, bl_, acc, 0, 0, 0);                                    \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al_, bh_, acc, 0, 0, 0);                                    \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah_, bh_, acc, 0, 0, 0);                                    \
  }
#pragma unroll
  for (int d = 0; d < D - 1; ++d) AE_LD(lo + d < hi ? lo + d : hi - 1, d)
  int it = lo;
  for (; it + D <= hi; it += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      AE_LD(it + u + D - 1 < hi ? it + u + D - 1 : hi - 1, (u + D - 1) % D)
      __builtin_amdgcn_sched_barrier(0);
      AE_MF(u)
    }
  }
#pragma unroll
  for (int u = 0; u < D - 1; ++u)
    if (it + u < hi) AE_MF(u)
#undef AE_LD
#undef AE_MF

  float4* r4 = reinterpret_cast<float4*>(red);
  r4[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  float mloc = 0.f;
  if (p < P && ks == 0) {
    float4 v = r4[pt * 64 + lane];
    for (int k2 = 1; k2 < KS; ++k2) {
      const float4 t = r4[((k2 << pt_lg) + pt) * 64 + lane];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const float unscale = sAi * sBi;
    v.x *= unscale; v.y *= unscale; v.z *= unscale; v.w *= unscale;
    const int c0 = m_base + 4 * q4;                              // D rows 4 q4 + r -> four consecutive couts
    const size_t o = ((size_t)(c0 >> 3) * g.out_HWp + (size_t)((g.out_s * y + 1) * g.out_Wp + g.out_s * x + 1)) * 8 + (c0 & 7);
    if (EPI == 0 || EPI == 2) {
      const float4 bb = ld4(bias + c0);
      v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
      if (EPI == 0) { v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w); }
    } else {
      const size_t oa = ((size_t)(c0 >> 3) * g.aux_HWp + (size_t)((g.aux_s * y + 1) * g.aux_Wp + g.aux_s * x + 1)) * 8 + (c0 & 7);
      const float4 yy = ld4(aux + oa);
      v.x *= lrelu_grad_from_out(yy.x); v.y *= lrelu_grad_from_out(yy.y);
      v.z *= lrelu_grad_from_out(yy.z); v.w *= lrelu_grad_from_out(yy.w);
    }
    st4(out + o, v);
    mloc = absmax4(v, mloc);
  }
  ae_amax_publish(mloc, q.amax_out);
}


static int ae_conv_init() {
  static int rc = -1;
  if (rc >= 0) return rc;
  rc = 0;
#define OPTIN(K_) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&K_), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); if (e != hipSuccess) rc = (int)e; }
  OPTIN((ae_conv_kernel<1, 0>)) OPTIN((ae_conv_kernel<1, 1>)) OPTIN((ae_conv_kernel<1, 2>)) OPTIN((ae_conv_kernel<2, 0>)) OPTIN((ae_conv_kernel<2, 1>)) OPTIN((ae_conv_kernel<2, 2>))
  OPTIN((ae_conv_f16_kernel<1, 0>)) OPTIN((ae_conv_f16_kernel<1, 1>)) OPTIN((ae_conv_f16_kernel<1, 2>))
  OPTIN((ae_conv_f16_kernel<2, 0>)) OPTIN((ae_conv_f16_kernel<2, 1>)) OPTIN((ae_conv_f16_kernel<2, 2>))
#undef OPTIN
  return rc;
}

// Launch shape: tile (mt = 2: 32 px x 64 cout, 1: 32 x 32, 3: 16 x 16), pixel tiles per workgroup PT and K slices KS
// (PT KS waves), from the sweep tools/ae_conv_tune.py (profiles/r03_ae_conv_tune.txt).  The waves of a workgroup share one
// CU's four MFMA pipes, so a layer's makespan is (workgroups per CU) x (a workgroup's MFMA cycles / the pipes its waves
// cover): the tile decides how many CUs a small layer reaches -- pick the tile with the smallest estimate, a 16 x 16 tile only
// if it wins by 15 % (it loads twice the operands per flop).  PT: as many tiles per workgroup (<= 4) as leave >= 200
// workgroups (they share their weight loads).  KS: four waves per workgroup (one per pipe) -- with the 8-deep operand ring a
// wave keeps its pipe busy on its own, and every further slice only adds to the LDS reduction (8 or 16 waves measured 10-25 %
// slower) -- unless a wave's chain would exceed ~36 steps' worth of MFMAs, which is what the deep 256-channel layers need.
// nclip (round 5): the launch carries nclip clips (blockIdx.z), so the tiles that share the CUs are nclip x a clip's -- the shape is
// chosen for what is actually in flight (a small layer of ONE clip needs 16 x 16 tiles and K slices to reach enough CUs; eight clips
// of it do not).  A clip's results then depend on the engine's clip count through the summation order of the K slices: bit-identical
// for equal nclip (and between the slots of one engine), equal to a solo run to rounding (tests/test_infill_emu.py, test_gpu_r2.py).
static void ae_conv_shape(int P, int cin, int cout, int nclip, int* mt_out, int* pt_out, int* ks_out, bool f16 = false) {
  static const bool solo_shapes = getenv("LEMO_AE_SHAPE_SOLO") && atoi(getenv("LEMO_AE_SHAPE_SOLO")) != 0;      // A/B knob: round 4's rule
  if (solo_shapes || nclip < 1) nclip = 1;
  double best = -1;
  for (int mt = 3; mt >= 1; --mt) {
    const int px = mt == 3 ? 16 : 32, co = mt == 3 ? 16 : 32 * mt;
    if (cout % co || (mt == 3 && cin < (f16 ? 32 : 16))) continue;
    // (f16: a step is 32 / 16 channels instead of 16 / 8, but the launch is bound by per-wave latency, not by the matrix pipe (round 6:
    // SQ_VALU_MFMA_BUSY 9 % of the launch, 2.7 waves per SIMD) -- the shape is chosen as for the fp32-input kernel, i.e. the SAME K
    // slices with half the steps each; choosing from the halved step count gave half the waves and 49 vs 29 ms per solo clip)
    const int nit = mt == 3 ? 9 * (cin / 16) : 9 * (cin / 8);
    const long tiles = (long)((P + px - 1) / px) * (cout / co) * nclip;
    int pt = 1;
    while (pt < 4 && tiles / (2 * pt) >= 200) pt *= 2;
    const int max_nw = mt == 2 ? 8 : 16;
    int ks = 1;
    while (pt * ks < 4 && pt * ks * 2 <= max_nw && ks * 2 <= nit) ks *= 2;
    const int chain = mt == 3 ? 36 : 36 / mt;                    // steps per wave worth ~36 x 4 fp32 32x32x2 MFMAs
    while ((nit + ks - 1) / ks > chain && pt * ks * 2 <= max_nw) ks *= 2;
    const int nw = pt * ks;
    const double per_wg = pt * (mt == 3 ? nit * 4.0 * 32 : nit * 4.0 * mt * 64) / (nw < 4 ? nw : 4);
    const long wgs = (tiles + pt - 1) / pt;
    const double est = (wgs > 256 ? wgs / 256.0 : 1.0) * per_wg * (mt == 3 ? 1.15 : 1.0);
    if (best < 0 || est < best) { best = est; *mt_out = mt; *pt_out = pt; *ks_out = ks; }
  }
}

// f16 != nullptr (and cin >= 16): the split-f16 kernels with the scales of *f16; else the fp32-input MFMA kernels
int ae_conv(const float* in, const float* wt, const float* bias, const float* aux, float* out, const AeGeo& g, int cin, int cout,
            int epi, hipStream_t s, int force_mt = 0, int force_pt = 0, int force_ks = 0, int nclip = 1, size_t cs = 0, const AeF16* f16 = nullptr) {
  if (cin % 8 || (cin & (cin - 1)) || cout % 32 || g.H < 1 || g.W < 1 || epi < 0 || epi > 2) return LEMO_ERR_SHAPE;
  float* amax_only = nullptr;                                 // fp32-input kernel that still publishes max |out| for its split-f16 consumer
  if (f16 && cin < 16) { amax_only = f16->amax_out; f16 = nullptr; }      // the 8-channel first layer keeps the fp32-input kernel (2 % of a step's flops)
  if (f16 && (!f16->amax_in || !f16->wmax || !f16->amax_out)) return LEMO_ERR_ARG;
  int mt = 1, pt = 1, ks = 1;
  ae_conv_shape(g.H * g.W, cin, cout, nclip, &mt, &pt, &ks, f16 != nullptr);
  if (force_mt) { mt = force_mt; pt = force_pt; ks = force_ks; }
  const int lg = ilog2(cin / 8), nw = pt * ks;
  if (pt < 1 || (pt & (pt - 1)) || ks < 1 || nw > 16) return LEMO_ERR_ARG;
  const int pt_lg = ilog2(pt);
  if (f16) {
    if (!force_mt) { const int nit16 = mt == 3 ? 9 * (cin / 32) : 9 * (cin / 16); while (ks > 1 && ks > nit16) ks >>= 1; }
    const int nw = pt * ks;
    if (mt == 3) {
      if (cin < 32 || ks > 9 * (cin / 32)) return LEMO_ERR_ARG;
      const dim3 grid(((g.H * g.W + 15) / 16 + pt - 1) / pt, cout / 16, nclip);
#define LAUNCH16(EPI_) hipLaunchKernelGGL((ae_conv16_f16_kernel<EPI_>), grid, dim3(64 * nw), (size_t)nw * 1024, s, in, wt, bias, aux, out, g, lg, cout, pt_lg, cs, *f16)
      if (epi == 0) LAUNCH16(0); else if (epi == 1) LAUNCH16(1); else LAUNCH16(2);
#undef LAUNCH16
      return (int)hipGetLastError();
    }
    if ((mt != 1 && mt != 2) || cout % (32 * mt) || nw > (mt == 2 ? 8 : 16) || ks > 9 * (cin / 16)) return LEMO_ERR_ARG;
    const dim3 grid(((g.H * g.W + 31) / 32 + pt - 1) / pt, cout / (32 * mt), nclip);
#define LAUNCH(MT_, EPI_) hipLaunchKernelGGL((ae_conv_f16_kernel<MT_, EPI_>), grid, dim3(64 * nw), (size_t)nw * mt * 4096, s, in, wt, bias, aux, out, g, lg, cout, pt_lg, cs, *f16)
    if (mt == 2) { if (epi == 0) LAUNCH(2, 0); else if (epi == 1) LAUNCH(2, 1); else LAUNCH(2, 2); }
    else         { if (epi == 0) LAUNCH(1, 0); else if (epi == 1) LAUNCH(1, 1); else LAUNCH(1, 2); }
#undef LAUNCH
    return (int)hipGetLastError();
  }
  if (mt == 3) {
    if (cin < 16 || ks > 9 * (cin / 16)) return LEMO_ERR_ARG;
    const dim3 grid(((g.H * g.W + 15) / 16 + pt - 1) / pt, cout / 16, nclip);
    const size_t lds = (size_t)nw * 1024;
#define LAUNCH16(EPI_) hipLaunchKernelGGL((ae_conv16_kernel<EPI_>), grid, dim3(64 * nw), lds, s, in, wt, bias, aux, out, g, lg, cout, pt_lg, cs)
    if (epi == 0) LAUNCH16(0); else if (epi == 1) LAUNCH16(1); else LAUNCH16(2);
#undef LAUNCH16
    return (int)hipGetLastError();
  }
  if ((mt != 1 && mt != 2) || cout % (32 * mt) || nw * mt > 16 || ks > 9 * (cin / 8)) return LEMO_ERR_ARG;
  const dim3 grid(((g.H * g.W + 31) / 32 + pt - 1) / pt, cout / (32 * mt), nclip);
  const size_t lds = (size_t)nw * mt * 4096;
#define LAUNCH(MT_, EPI_) hipLaunchKernelGGL((ae_conv_kernel<MT_, EPI_>), grid, dim3(64 * nw), lds, s, in, wt, bias, aux, out, g, lg, cout, pt_lg, cs, amax_only)
  if (mt == 2) { if (epi == 0) LAUNCH(2, 0); else if (epi == 1) LAUNCH(2, 1); else LAUNCH(2, 2); }
  else         { if (epi == 0) LAUNCH(1, 0); else if (epi == 1) LAUNCH(1, 1); else LAUNCH(1, 2); }
#undef LAUNCH
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// all weight gradients of a step in one launch
// ---------------------------------------------------------------------------------------------------------------------
//   dW[co][ci][tap] = sum_p dY[co][p] X[ci][p + tap]        (X zero-padded: CG8P border)
// GEMM per tap: M = ci (A = X shifted), N = co (B = dY), K = pixels.  K runs over the PADDED linear pixel index q of the
// interior rows (border columns included: dY is zero there, so they add nothing and the taps of q are simply q + dy Wp + dx
// with no row arithmetic).  One WAVE = one workgroup = one 32 (ci) x 32 (co) tile, one kernel row dy, one slab of <= 576 padded
// pixels: three accumulators (dx = -1, 0, +1) and, per step of two pixels, ONE dY operand times three X operands out of a
// sliding 17-pixel window per 8 steps -- one scalar load per MFMA, every one of them `base + immediate` (consecutive pixels are
// 32 bytes apart; nothing is clamped: the two window positions that can fall outside a channel-group plane belong to border
// pixels (dY = 0) and land in the neighbouring plane or in the zeroed guard floats every engine buffer is wrapped in).  No LDS,
// no barrier; the wave of the centre row and first ci tile also sums its dY operands: the bias gradient's slab partial.
// How it got here (tools/ae_wgrad_probe.py, profiles/r03_ae_wgrad_probe.txt): one tap per 4-wave workgroup with per-load index
// arithmetic took 134 us for 70 us of MFMA pipe time; all taps per 6-wave workgroup (3 rows x 2 halves, LDS reduction) 150 us --
// 120 us of it with the loads REMOVED: six waves on a CU's four matrix pipes load two of them twice as much as the others,
// and 36 KB of LDS + 112 registers held a CU at two such workgroups.  Single-wave workgroups leave the placement to the
// dispatcher, which balances waves over the pipes.
// With ci on the MFMA's row axis a lane holds 4 consecutive ci of one co per accumulator quad: the slab partial is stored with
// dwordx4 stores, 1 KiB contiguous per wave, directly in the forward pack wt[tap][ci/8][co][8] -- the layout of the parameter
// vector, so the optimizer reads it with unit stride.
#define AE_SLAB 576               // padded pixels per slab, at most (a layer's slabs are equal parts: ae_slabs)
struct AeWgradJob {
  const float* dy; const float* x; float* partial; float* dbp;       // dbp: [nslab][2][cout] bias-gradient partials (pixel parity kept apart)
  int H, W; int cin, cout, nslab, slab_len, nwave;
};
#define AE_NLAYER 20
// the grid is the layers' waves in the order the host wants them DISPATCHED: the long ones first (full 576-pixel slabs run 23 us),
// the short ones of the 14 x 9 layers last, so that the launch does not end on a few long waves
struct AeWgradJobs { AeWgradJob j[AE_NLAYER]; int first[AE_NLAYER + 1]; int n; size_t cs; };       // cs: clip stride (clip = blockIdx.y)

template <int MODE>        // 0: the product; diagnostics (tools/ae_wgrad_probe.py): 1 = operands loaded once per wave, 2 = no MFMAs
__global__ void __launch_bounds__(64)
ae_wgrad_multi_kernel(AeWgradJobs J) {
  int k = 0;
  while (k + 1 < J.n && (int)blockIdx.x >= J.first[k + 1]) ++k;               // block -> layer (uniform)
  const AeWgradJob& q = J.j[k];
  int tile = (int)blockIdx.x - J.first[k];                         // (slab, co tile, ci tile, kernel row)
  const int lane = threadIdx.x;
  const int Wp = q.W + 2, HWp = (q.H + 2) * Wp;
  const int i = lane & 31, kk = lane >> 5;
  const int cin = q.cin, cout = q.cout;
  const int cot = cout >> 5, cit = (cin + 31) >> 5;
  const int r = tile % 3; tile /= 3;                               // kernel row dy = r - 1
  const int ct = tile % cit; tile /= cit;
  const int mt = tile % cot;
  const int slab = tile / cot;
  const int co = mt * 32 + i;
  int ci = ct * 32 + i;
  if (ci >= cin) ci = cin - 1;                                     // rows past cin are computed and never stored
  const size_t clip_off = (size_t)blockIdx.y * J.cs;
  const float* bp = q.dy + clip_off + ((size_t)(co >> 3) * HWp) * 8 + (co & 7);
  const float* ap = q.x + clip_off + ((size_t)(ci >> 3) * HWp) * 8 + (ci & 7);
  const int Q1 = (q.H + 1) * Wp;                                   // interior rows: padded pixels [Wp, (H + 1) Wp)
  const int qs = Wp + slab * q.slab_len;
  const int qe = qs + q.slab_len < Q1 ? qs + q.slab_len : Q1;      // (host: qs < Q1)
  const int off = (r - 1) * Wp + kk - 1;                           // window origin of this lane relative to the group's first pixel
  f32x16 acc0, acc1, acc2;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; acc2[e] = 0.f; }
  float bsum = 0.f;
  // a group = 8 steps = 16 pixels: dY of pixel qb + 2u + kk, X of pixels qb + off + 0..16.  Group g + 1 is requested before
  // the 24 MFMAs of group g (two register sets).
  float b[2][8], xw[2][17];
#define WG_LOAD(SET, QB)                                                                           \
  {                                                                                                \
    const float* bq = bp + (std::ptrdiff_t)((QB) + kk) * 8;                                        \
    const float* aq = ap + (std::ptrdiff_t)((QB) + off) * 8;                                       \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) b[SET][u] = bq[16 * u];                          \
    _Pragma("unroll") for (int t = 0; t < 17; ++t) xw[SET][t] = aq[8 * t];                         \
    if ((QB) + 16 > qe) {                      /* last group of the slab: pixels past its end */    \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) if ((QB) + 2 * u + kk >= qe) b[SET][u] = 0.f;  \
    }                                                                                              \
  }
#define WG_MFMA(SET)                                                                               \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                  \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xw[SET][2 * u], b[SET][u], acc0, 0, 0, 0);         \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xw[SET][2 * u + 1], b[SET][u], acc1, 0, 0, 0);     \
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xw[SET][2 * u + 2], b[SET][u], acc2, 0, 0, 0);     \
    bsum += b[SET][u];                                                                             \
  }
#define WG_FAKE(SET)                                                                               \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) { acc0[u] += b[SET][u] + xw[SET][2 * u]; acc1[u] += xw[SET][2 * u + 1]; } \
  acc2[0] += xw[SET][16];
  if (MODE == 4) {
    // no scheduling fences: the compiler is free to spread group g + 1's loads between group g's MFMAs
    WG_LOAD(0, qs)
    for (int qb = qs; qb < qe; qb += 32) {
      if (qb + 16 < qe) { WG_LOAD(1, qb + 16) }
      WG_MFMA(0)
      if (qb + 16 < qe) {
        if (qb + 32 < qe) { WG_LOAD(0, qb + 32) }
        WG_MFMA(1)
      }
    }
  } else if (MODE == 3) {
    // single register set: more waves per SIMD (83 instead of 112 registers) cover each other's load phases
    for (int qb = qs; qb < qe; qb += 16) {
      WG_LOAD(0, qb)
      __builtin_amdgcn_sched_barrier(0);
      WG_MFMA(0)
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
  WG_LOAD(0, qs)
  if (MODE == 1) { WG_LOAD(1, qs) }
  for (int qb = qs; qb < qe; qb += 32) {
    if (MODE != 1 && qb + 16 < qe) { WG_LOAD(1, qb + 16) }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE != 2) { WG_MFMA(0) } else { WG_FAKE(0) }
    __builtin_amdgcn_sched_barrier(0);
    if (qb + 16 < qe) {
      if (MODE != 1 && qb + 32 < qe) { WG_LOAD(0, qb + 32) }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 2) { WG_MFMA(1) } else { WG_FAKE(1) }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  }
#undef WG_LOAD
#undef WG_MFMA
#undef WG_FAKE
  // D: col = lane & 31 -> co, rows (e & 3) + 8 (e >> 2) + 4 kk -> ci: quad qd = 4 consecutive ci of channel group ct*4 + qd
  float* outp = q.partial + clip_off + (size_t)slab * (9 * (size_t)cin * cout);
  const int CG = cin >> 3;
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const int cg = ct * 4 + qd;
    if (cg < CG) {
      float* o = outp + (((size_t)(3 * r) * CG + cg) * cout + co) * 8 + 4 * kk;
      const size_t tap_stride = (size_t)CG * cout * 8;
      st4(o, make_float4(acc0[4 * qd], acc0[4 * qd + 1], acc0[4 * qd + 2], acc0[4 * qd + 3]));
      st4(o + tap_stride, make_float4(acc1[4 * qd], acc1[4 * qd + 1], acc1[4 * qd + 2], acc1[4 * qd + 3]));
      st4(o + 2 * tap_stride, make_float4(acc2[4 * qd], acc2[4 * qd + 1], acc2[4 * qd + 2], acc2[4 * qd + 3]));
    }
  }
  if (r == 1 && ct == 0) q.dbp[clip_off + ((size_t)slab * 2 + kk) * cout + co] = bsum;   // sum over this slab's pixels of parity kk, in pixel order
}

// ---------------------------------------------------------------------------------------------------------------------
// optimizer: slab reduction + Adam + backward pack, one launch over the packed parameter vector
// ---------------------------------------------------------------------------------------------------------------------
// theta = [weights of layer 0 | ... | weights of layer 19 | biases of layer 0 | ... ], weights in the forward pack
// wt[tap][cin_pad/8][cout_pad][8] (each a multiple of 256 floats: a block never straddles layers), biases padded to
// cout_pad.  Padded entries have zero gradient (their operands are zero) and are masked here as well, so they stay zero.
struct AeAdamLayer { const float* partial; const float* dbp; int nslab, w_off, b_off, wb_off, cin_lg /*log2(cin_pad/8)*/, cout_lg, cin, cout; };
struct AeAdamArgs {
  AeAdamLayer L[AE_NLAYER];
  float* theta; float* m; float* v; float* wb; const float* ctr;    // ctr: [1] = -lr / (1 - b1^t), [2] = sqrt(1 - b2^t) (floats)
  float* amax;                                                      // the step's tensor maxima: slots 0 .. AE_SLOT_DYN - 1 zeroed here for the next step
  int n_w, n_all; float lr;
  size_t cs;                                                        // clip stride (clip = blockIdx.y)
};

__device__ __forceinline__ float adam_update(float p, float g, float& m, float& v, AdamCoef c) {
  adam_update_torch(p, m, v, g, c);                                 // common.hpp: torch.optim.Adam's own evaluation order
  return p;
}

__global__ void __launch_bounds__(256)
ae_adam_kernel(AeAdamArgs A) {
  // one thread = four consecutive entries (one half of an 8-channel group of one (tap, cout)): dwordx4 everywhere but the
  // scatter into the backward pack
  const int idx = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (blockIdx.x == 0 && threadIdx.x < AE_SLOT_DYN) A.amax[(size_t)blockIdx.y * A.cs + threadIdx.x] = 0.f;     // (last launch of the step: every reader is done)
  if (idx >= A.n_all) return;
  const size_t clip_off = (size_t)blockIdx.y * A.cs;                // (the argument struct itself stays read-only: a kernel that
  float* const theta = A.theta + clip_off;                          //  writes to it gets a private copy in scratch -- 30 -> 250 us)
  float* const am = A.m + clip_off;
  float* const av = A.v + clip_off;
  float* const wb = A.wb + clip_off;
  const float* const ctr = A.ctr + clip_off;
  const AdamCoef ac{ctr[1], ctr[2]};                            // (-lr / (1 - b1^t), sqrt(1 - b2^t)) of this step
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  int wb_idx = -1;
  if (idx < A.n_w) {
    int k = 0;
    while (k + 1 < AE_NLAYER && idx >= A.L[k + 1].w_off) ++k;       // uniform per block
    const AeAdamLayer& q = A.L[k];
    const int i = idx - q.w_off;
    const int c8 = i & 7, co = (i >> 3) & ((1 << q.cout_lg) - 1);
    const int t = i >> (3 + q.cout_lg);
    const int cg = t & ((1 << q.cin_lg) - 1), tap = t >> q.cin_lg;
    const int ci = cg * 8 + c8;
    const int n_w = 9 << (q.cin_lg + 3 + q.cout_lg);
    if (ci < q.cin && co < q.cout) {
      for (int s0 = 0; s0 < q.nslab; s0 += 8) {                     // slab order: deterministic; 8 loads in flight
        float4 pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pv[u] = ld4(q.partial + clip_off + (size_t)(s0 + u < q.nslab ? s0 + u : q.nslab - 1) * n_w + i);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (s0 + u < q.nslab) { g.x += pv[u].x; g.y += pv[u].y; g.z += pv[u].z; g.w += pv[u].w; }
      }
      if (ci + 1 >= q.cin) g.y = 0.f;                               // padded input channels keep a zero gradient
      if (ci + 2 >= q.cin) g.z = 0.f;
      if (ci + 3 >= q.cin) g.w = 0.f;
    }
    if (q.wb_off >= 0)        // backward-data pack of the same convolution: wtb[8 - tap][co/8][ci][co%8]
      wb_idx = q.wb_off + ((((8 - tap) << (q.cout_lg - 3)) + (co >> 3)) << (q.cin_lg + 3)) * 8 + ci * 8 + (co & 7);
  } else {
    const int bi = idx - A.n_w;
    int k = 0;
    while (k + 1 < AE_NLAYER && bi >= A.L[k + 1].b_off) ++k;
    const AeAdamLayer& q = A.L[k];
    const int co = bi - q.b_off, cop = 1 << q.cout_lg;
    if (co < q.cout) {
      const int ns = 2 * q.nslab;
      for (int s0 = 0; s0 < ns; s0 += 8) {                          // slab order, even pixels then odd: deterministic
        float4 pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pv[u] = ld4(q.dbp + clip_off + (size_t)(s0 + u < ns ? s0 + u : ns - 1) * cop + co);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (s0 + u < ns) { g.x += pv[u].x; g.y += pv[u].y; g.z += pv[u].z; g.w += pv[u].w; }
      }
      if (co + 1 >= q.cout) g.y = 0.f;
      if (co + 2 >= q.cout) g.z = 0.f;
      if (co + 3 >= q.cout) g.w = 0.f;
    }
  }
  float4 m = ld4(am + idx), v = ld4(av + idx), p = ld4(theta + idx);
  p.x = adam_update(p.x, g.x, m.x, v.x, ac);
  p.y = adam_update(p.y, g.y, m.y, v.y, ac);
  p.z = adam_update(p.z, g.z, m.z, v.z, ac);
  p.w = adam_update(p.w, g.w, m.w, v.w, ac);
  st4(am + idx, m); st4(av + idx, v); st4(theta + idx, p);
  if (wb_idx >= 0) { wb[wb_idx] = p.x; wb[wb_idx + 8] = p.y; wb[wb_idx + 16] = p.z; wb[wb_idx + 24] = p.w; }
}

// ---------------------------------------------------------------------------------------------------------------------
// small kernels: loss gradient, (un)packing
// ---------------------------------------------------------------------------------------------------------------------
// d(loss)/d(rec) of loss = sum(|rec - x| * mask) / count (opt_amass_temp.py:199-203) = sign(rec - x) * (mask / count), into
// channel 0 of the last layer's d(pre-activation) (the layer has no activation); thread 0 advances the step counter and
// the bias corrections Adam reads later in the same step.
__global__ void __launch_bounds__(256)
ae_loss_grad_kernel(const float* __restrict__ rec, const float* __restrict__ x8, const float* __restrict__ moc,
                    float* __restrict__ dpre, int H, int W, float* __restrict__ ctr, double lr, size_t cs) {
  { const size_t o_ = (size_t)blockIdx.y * cs; rec += o_; x8 += o_; moc += o_; dpre += o_; ctr += o_; }       // clip = blockIdx.y
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p == 0) {
    int* ci = reinterpret_cast<int*>(ctr);
    const int step = ci[0] + 1;
    ci[0] = step;
    const AdamCoef ac = adam_coef_t(step, lr);
    ctr[1] = ac.neg_step;
    ctr[2] = ac.bc2s;
  }
  if (p >= H * W) return;
  const int y = p / W, x = p - y * W;
  const size_t o = (size_t)((y + 1) * (W + 2) + x + 1) * 8;
  const float d = rec[o] - x8[o];
  const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);            // torch.sign; NaN -> 0 like the comparison chain
  dpre[o] = sg * moc[p];
}

// max |src[0 .. n)| per job into its slot (load time: the clip image, mask / count, every layer's weights)
struct AeAbsJobs { const float* src[24]; int n[24]; float* dst[24]; };
__global__ void __launch_bounds__(256)
ae_absmax_kernel(AeAbsJobs J) {
  const float* src = J.src[blockIdx.y];
  const int n = J.n[blockIdx.y];
  float m = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) m = fmaxf(m, fabsf(src[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(J.dst[blockIdx.y]), __builtin_bit_cast(unsigned, m));
}

// plain [C][H][W] -> CG8P (channels >= C of the last group stay as they are: zero)
__global__ void __launch_bounds__(256)
ae_to_cg8p_kernel(const float* __restrict__ src, int C, int H, int W, float* __restrict__ dst) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= C * H * W) return;
  const int c = t / (H * W), p = t - c * H * W, y = p / W, x = p - y * W;
  dst[((size_t)(c >> 3) * (H + 2) * (W + 2) + (size_t)((y + 1) * (W + 2) + x + 1)) * 8 + (c & 7)] = src[t];
}
__global__ void __launch_bounds__(256)
ae_from_cg8p_kernel(const float* __restrict__ src, int C, int H, int W, float* __restrict__ dst) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= C * H * W) return;
  const int c = t / (H * W), p = t - c * H * W, y = p / W, x = p - y * W;
  dst[t] = src[((size_t)(c >> 3) * (H + 2) * (W + 2) + (size_t)((y + 1) * (W + 2) + x + 1)) * 8 + (c & 7)];
}

// the model's own tensors (state_dict order: per layer weight then bias; Conv2d weight [out][in][3][3], ConvTranspose2d
// weight [in][out][3][3]) <-> theta (+ the backward pack).  The conv-equivalent weight of a stride-1 transposed convolution
// is the flipped transpose: cw[co][ci][tap] = w[ci][co][8 - tap].
struct AePackLayer { int w_off, wb_off, b_off, flat_w, flat_b, cin_lg, cout_lg, cin, cout, deconv; };
struct AePackArgs { AePackLayer L[AE_NLAYER]; int n_w, n_all; };

__device__ __forceinline__ int ae_flat_index(const AePackLayer& q, int tap, int ci, int co) {
  return q.deconv ? q.flat_w + (ci * q.cout + co) * 9 + (8 - tap) : q.flat_w + (co * q.cin + ci) * 9 + tap;
}

template <bool UNPACK>
__global__ void __launch_bounds__(256)
ae_pack_kernel(AePackArgs A, const float* __restrict__ src, float* __restrict__ theta, float* __restrict__ wb) {
  // UNPACK = false: src = flat -> theta, wb.  UNPACK = true: src = theta -> `theta` argument = flat output
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= A.n_all) return;
  if (idx < A.n_w) {
    int k = 0;
    while (k + 1 < AE_NLAYER && idx >= A.L[k + 1].w_off) ++k;
    const AePackLayer& q = A.L[k];
    const int i = idx - q.w_off;
    const int c8 = i & 7, co = (i >> 3) & ((1 << q.cout_lg) - 1);
    const int t = i >> (3 + q.cout_lg);
    const int cg = t & ((1 << q.cin_lg) - 1), tap = t >> q.cin_lg;
    const int ci = cg * 8 + c8;
    const bool real = ci < q.cin && co < q.cout;
    if (UNPACK) {
      if (real) theta[ae_flat_index(q, tap, ci, co)] = src[idx];
    } else {
      const float v = real ? src[ae_flat_index(q, tap, ci, co)] : 0.f;
      theta[idx] = v;
      if (q.wb_off >= 0)
        wb[q.wb_off + ((((8 - tap) << (q.cout_lg - 3)) + (co >> 3)) << (q.cin_lg + 3)) * 8 + ci * 8 + (co & 7)] = v;
    }
  } else {
    const int b = idx - A.n_w;
    int k = 0;
    while (k + 1 < AE_NLAYER && b >= A.L[k + 1].b_off) ++k;
    const AePackLayer& q = A.L[k];
    const int co = b - q.b_off;
    if (UNPACK) { if (co < q.cout) theta[q.flat_b + co] = src[idx]; }
    else theta[idx] = co < q.cout ? src[q.flat_b + co] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// the engine
// ---------------------------------------------------------------------------------------------------------------------
static const int AE_ENC[5][2] = {{4, 32}, {32, 64}, {64, 128}, {128, 256}, {256, 256}};      // models/AE.py:81-91, in_channel = 4
static const int AE_DEC[5][2] = {{256, 256}, {256, 128}, {128, 64}, {64, 32}, {32, 1}};

struct AeLayer { int cin, cout, cin_pad, cout_pad, deconv, level, w_off, b_off, wb_off, flat_w, flat_b, nslab, slab_len; size_t part_off, dbp_off; };

// slabs of padded interior pixels for the weight gradients: equal parts of at most AE_SLAB pixels, a multiple of 32 long (27 x 19
// = 513 padded pixels is ONE slab, not 512 + 1)
// `scale` (round 5, measured and NOT adopted): with k clips side by side the launch has k x the waves, so a clip's slabs could be k x
// longer for the same number of waves in flight (fewer partial tiles written by the weight-gradient launch and read back by the
// optimiser launch).  8 clips per engine, ms per clip: x1 16.40, x2 16.69, x4 17.24, x8 18.07, x16 20.51; 16 clips: x1 15.43, x4 15.76,
// x8 16.32 (profiles/r05_ab_ae_slab_scale.txt) -- the launch wants MANY SHORT waves (its cost is the spread between a level-0 wave
// and a level-4 one, not the 40 MB of partials per clip), so the scale stays 1; LEMO_AE_SLAB_SCALE keeps the switch
static int ae_slabs(int H, int W, int* len, int scale = 1) {
  const int cap = AE_SLAB * (scale < 1 ? 1 : scale);
  const int n_px = H * (W + 2), n = (n_px + cap - 1) / cap;
  *len = ((n_px + n - 1) / n + 31) / 32 * 32;
  return (n_px + *len - 1) / *len;
}

static int pad8(int c) { return (c + 7) / 8 * 8; }
static int pad32(int c) { return (c + 31) / 32 * 32; }
static size_t cg8p_floats(int C, int H, int W) { return (size_t)(C / 8 > 0 ? C / 8 : 1) * (H + 2) * (W + 2) * 8; }

struct AeEngine {
  int H[6], W[6];
  AeLayer L[AE_NLAYER];
  int n_w = 0, n_b = 0, n_wb = 0, n_flat = 0;
  size_t n_part = 0, n_dbp = 0;
  float lr = 0.f;
  // carved from the caller's workspace
  float *theta = nullptr, *m = nullptr, *v = nullptr, *wb = nullptr, *dbp = nullptr, *zero_bias = nullptr, *ctr = nullptr,
        *part = nullptr, *moc = nullptr, *x8 = nullptr;
  float* act[AE_NLAYER];       // output of layer i (decoder blocks 0..3: their second layer's output lives stuffed in S[b + 1])
  float* xin[AE_NLAYER];       // input of layer i
  float* dp[AE_NLAYER];        // d(pre-activation) of layer i
  float *P[5], *dP[5], *S[5];
  unsigned char* idx[5];
  hipGraphExec_t exec[2] = {nullptr, nullptr};      // 5 steps, 1 step
  unsigned long long loaded = 0;   // bit c: clip c has been loaded (every clip must be before a step / forward: ADVICE r04)
  int forward_valid = 0;           // an eval forward of all clips ran after the last step / load (what forward_clip(c > 0) copies from)
  int nclip = 1;               // clips side by side: clip c's buffers are the pointers above + c * cs floats
  size_t cs = 0;
  float* amax = nullptr;       // [AE_NSLOT] per clip: tensor maxima / bounds the split-f16 convolutions scale by (slots below)
  int f16 = 0;                 // 0: fp32-input MFMA convolutions (shipped) ; 1: split-f16 convolutions (round 6; LEMO_AE_ARITH=f16).  Measured
                               // (profiles/r06_ae_f16_vs_fp32.txt): 16.05 vs 16.4 ms per clip at 8 clips per engine, 14.4 vs 15.45 at 16, but 44.5 vs
                               // 28.8 for a solo clip -- the launches are bound by L2 re-reads of the direct-from-global operands (every input element
                               // 9 taps x cout / 32 MT times) and per-wave latency, not by the matrix pipe: SQ_VALU_MFMA_BUSY 9 % (f16) / 48 % (fp32)
};


// every buffer is wrapped in AE_GUARD zeroed floats that nothing writes: the weight-gradient kernel's operand windows may
// reach up to 17 pixels (136 floats) past either end of a CG8P buffer, at positions whose products are multiplied by zero
#define AE_GUARD 256
struct Bump {
  float* base; size_t off = 0;
  float* take(size_t n) { float* p = base ? base + off + AE_GUARD : nullptr; off += (n + 2 * AE_GUARD + 63) / 64 * 64; return p; }
};

// the same routine sizes the workspace (base == nullptr) and carves it
static int ae_slab_scale(int nclip) {
  static const char* ev = getenv("LEMO_AE_SLAB_SCALE");        // A/B switch: slab length in units of AE_SLAB pixels, whatever the clip count
  if (ev && atoi(ev) >= 1) return atoi(ev);
  (void)nclip;
  return 1;
}

static void ae_layout(AeEngine* e, int H0, int W0, float* base, size_t* total, int slab_scale = 1) {
  e->H[0] = H0; e->W[0] = W0;
  for (int k = 0; k < 5; ++k) { e->H[k + 1] = (e->H[k] - 1) / 2 + 1; e->W[k + 1] = (e->W[k] - 1) / 2 + 1; }
  int n = 0;
  for (int b = 0; b < 5; ++b) {
    const int ci = AE_ENC[b][0], co = AE_ENC[b][1];
    e->L[n++] = AeLayer{ci, co, pad8(ci), pad32(co), 0, b};
    e->L[n++] = AeLayer{co, co, pad32(co), pad32(co), 0, b};
  }
  for (int b = 0; b < 5; ++b) {
    const int ci = AE_DEC[b][0], co = AE_DEC[b][1];
    e->L[n++] = AeLayer{ci, co, pad32(ci), pad32(co), 1, 4 - b};
    e->L[n++] = AeLayer{co, co, pad32(co), pad32(co), 1, 4 - b};
  }
  int w = 0, wbo = 0, flat = 0;
  size_t part = 0, dbp = 0;
  for (int i = 0; i < AE_NLAYER; ++i) {
    AeLayer& l = e->L[i];
    l.w_off = w; w += 9 * l.cin_pad * l.cout_pad;
    l.wb_off = i == 0 ? -1 : wbo; if (i) wbo += 9 * l.cin_pad * l.cout_pad;      // the first layer needs no backward-data
    l.flat_w = flat; flat += 9 * l.cin * l.cout;
    l.flat_b = flat; flat += l.cout;
    l.nslab = ae_slabs(e->H[l.level], e->W[l.level], &l.slab_len, slab_scale);
    l.part_off = part; part += (size_t)l.nslab * 9 * l.cin_pad * l.cout_pad;
    l.dbp_off = dbp; dbp += (size_t)l.nslab * 2 * l.cout_pad;
  }
  e->n_w = w; e->n_wb = wbo; e->n_flat = flat; e->n_part = part; e->n_dbp = dbp;
  int b = 0;
  for (int i = 0; i < AE_NLAYER; ++i) { e->L[i].b_off = b; b += e->L[i].cout_pad; }
  e->n_b = b;
  Bump B{base};
  e->theta = B.take(e->n_w + e->n_b); e->m = B.take(e->n_w + e->n_b); e->v = B.take(e->n_w + e->n_b);
  e->wb = B.take(e->n_wb); e->dbp = B.take(e->n_dbp); e->zero_bias = B.take(256); e->ctr = B.take(64); e->amax = B.take(AE_NSLOT);
  e->part = B.take(e->n_part); e->moc = B.take((size_t)H0 * W0);
  e->x8 = B.take(cg8p_floats(8, H0, W0));
  for (int bk = 0; bk < 5; ++bk) {
    const int lv = bk, i0 = 2 * bk, i2 = 2 * bk + 1;
    e->act[i0] = B.take(cg8p_floats(e->L[i0].cout_pad, e->H[lv], e->W[lv]));
    e->act[i2] = B.take(cg8p_floats(e->L[i2].cout_pad, e->H[lv], e->W[lv]));
    e->P[bk] = B.take(cg8p_floats(e->L[i2].cout_pad, e->H[lv + 1], e->W[lv + 1]));
    e->dP[bk] = B.take(cg8p_floats(e->L[i2].cout_pad, e->H[lv + 1], e->W[lv + 1]));
    e->idx[bk] = reinterpret_cast<unsigned char*>(B.take(((size_t)e->L[i2].cout_pad * e->H[lv + 1] * e->W[lv + 1] + 3) / 4));
    e->xin[i0] = bk == 0 ? e->x8 : e->P[bk - 1];
    e->xin[i2] = e->act[i0];
  }
  for (int bk = 0; bk < 5; ++bk) {
    const int lv = 4 - bk, i1 = 10 + 2 * bk;
    e->S[bk] = B.take(cg8p_floats(e->L[i1].cin_pad, e->H[lv], e->W[lv]));
  }
  for (int bk = 0; bk < 5; ++bk) {
    const int lv = 4 - bk, i1 = 10 + 2 * bk, i2 = 11 + 2 * bk;
    e->act[i1] = B.take(cg8p_floats(e->L[i1].cout_pad, e->H[lv], e->W[lv]));
    e->act[i2] = bk < 4 ? e->S[bk + 1] : B.take(cg8p_floats(e->L[i2].cout_pad, e->H[lv], e->W[lv]));
    e->xin[i1] = e->S[bk];
    e->xin[i2] = e->act[i1];
  }
  for (int i = 0; i < AE_NLAYER; ++i) e->dp[i] = B.take(cg8p_floats(e->L[i].cout_pad, e->H[e->L[i].level], e->W[e->L[i].level]));
  *total = B.off;
}

static AeGeo geo_plain(int H, int W) { const int Wp = W + 2, HWp = (H + 2) * Wp; return AeGeo{H, W, Wp, HWp, 1, Wp, HWp, 1, Wp, HWp, 1}; }

// engine launches carry every clip: ae_conv(..., s) -> ae_conv(..., s, 0, 0, 0, e->nclip, e->cs)
#define AE_CONV(e_, ...) ae_conv(__VA_ARGS__, 0, 0, 0, (e_)->nclip, (e_)->cs)
// ... with the scale slots of a split-f16 launch: input maximum (x fac: a bound, see the kernel's header), weights of layer LW, output slot
#define AE_CONVS(e_, SIN, FAC, LW, SOUT, ...)                                                                                   \
  [&]() { const AeF16 q_{(e_)->amax + (SIN), (e_)->amax + AE_SLOT_W(LW), (e_)->amax + (SOUT), (FAC)};                           \
          return ae_conv(__VA_ARGS__, 0, 0, 0, (e_)->nclip, (e_)->cs, (e_)->f16 ? &q_ : nullptr); }()

static int ae_forward(AeEngine* e, hipStream_t s) {
  for (int b = 0; b < 5; ++b) {
    const int H = e->H[b], W = e->W[b], i0 = 2 * b, i2 = 2 * b + 1;
    const AeGeo g = geo_plain(H, W);
    // (pooling and zero stuffing keep the maximum: the pooled / stuffed inputs scale by their source's slot)
    CHK_(AE_CONVS(e, b == 0 ? AE_SLOT_X8 : AE_SLOT_ACT(i0 - 1), 1.f, i0, AE_SLOT_ACT(i0),
                  e->xin[i0], e->theta + e->L[i0].w_off, e->theta + e->n_w + e->L[i0].b_off, nullptr, e->act[i0], g, e->L[i0].cin_pad, e->L[i0].cout_pad, 0, s));
    CHK_(AE_CONVS(e, AE_SLOT_ACT(i0), 1.f, i2, AE_SLOT_ACT(i2),
                  e->xin[i2], e->theta + e->L[i2].w_off, e->theta + e->n_w + e->L[i2].b_off, nullptr, e->act[i2], g, e->L[i2].cin_pad, e->L[i2].cout_pad, 0, s));
    CHK_(maxpool3s2_fwd(e->act[i2], H, W, e->P[b], e->idx[b], e->L[i2].cout_pad, s, e->nclip, e->cs));
  }
  CHK_(stuff2_fwd(e->P[4], e->H[5], e->W[5], e->S[0], e->H[4], e->W[4], e->L[10].cin_pad, s, e->nclip, e->cs));
  for (int b = 0; b < 5; ++b) {
    const int lv = 4 - b, H = e->H[lv], W = e->W[lv], i1 = 10 + 2 * b, i2 = 11 + 2 * b;
    AeGeo g = geo_plain(H, W);
    CHK_(AE_CONVS(e, AE_SLOT_ACT(i1 - 1), 1.f, i1, AE_SLOT_ACT(i1),
                  e->xin[i1], e->theta + e->L[i1].w_off, e->theta + e->n_w + e->L[i1].b_off, nullptr, e->act[i1], g, e->L[i1].cin_pad, e->L[i1].cout_pad, 0, s));
    if (b < 4) {       // straight into the stuffed input of the next block: pixel (y, x) -> (2y, 2x) of the next finer level
      g.out_Wp = e->W[lv - 1] + 2; g.out_HWp = (e->H[lv - 1] + 2) * g.out_Wp; g.out_s = 2;
    }
    CHK_(AE_CONVS(e, AE_SLOT_ACT(i1), 1.f, i2, AE_SLOT_ACT(i2),
                  e->xin[i2], e->theta + e->L[i2].w_off, e->theta + e->n_w + e->L[i2].b_off, nullptr, e->act[i2], g, e->L[i2].cin_pad, e->L[i2].cout_pad, b < 4 ? 0 : 2, s));
  }
  return 0;
}

static int ae_wgrad_launch(AeEngine* e, hipStream_t s, int mode);

static int ae_train_step(AeEngine* e, hipStream_t s) {
  CHK_(ae_forward(e, s));
  const int H0 = e->H[0], W0 = e->W[0];
  hipLaunchKernelGGL(ae_loss_grad_kernel, dim3((H0 * W0 + 255) / 256, e->nclip), dim3(256), 0, s, (const float*)e->act[19], (const float*)e->x8,
                     (const float*)e->moc, e->dp[19], H0, W0, e->ctr, lr_decimal(e->lr), e->cs);
  CHK_((int)hipGetLastError());
  // ---- decoder, last block first
  for (int b = 4; b >= 0; --b) {
    const int lv = 4 - b, H = e->H[lv], W = e->W[lv], i1 = 10 + 2 * b, i2 = 11 + 2 * b;
    const AeGeo g = geo_plain(H, W);
    CHK_(AE_CONVS(e, i2 == 19 ? AE_SLOT_LOSS : AE_SLOT_DP(i2), 1.f, i2, AE_SLOT_DP(i1),
                  e->dp[i2], e->wb + e->L[i2].wb_off, nullptr, e->act[i1], e->dp[i1], g, e->L[i2].cout_pad, e->L[i2].cin_pad, 1, s));     // * lrelu'(act[i1])
    // adjoint of (stuffing, transposed conv): only the even pixels of d(stuffed input) exist downstream -> enumerate the
    // coarse grid, centre taps at (2i, 2j); times lrelu' of the previous block's output (read where it lives: stuffed in S[b])
    const int h = e->H[lv + 1], w = e->W[lv + 1];
    AeGeo gs = geo_plain(h, w);
    gs.in_Wp = W + 2; gs.in_HWp = (H + 2) * (W + 2); gs.in_s = 2;
    gs.aux_Wp = gs.in_Wp; gs.aux_HWp = gs.in_HWp; gs.aux_s = 2;
    float* dst = b > 0 ? e->dp[i1 - 1] : e->dP[4];
    if (b > 0) CHK_(AE_CONVS(e, AE_SLOT_DP(i1), 1.f, i1, AE_SLOT_DP(i1 - 1),
                             e->dp[i1], e->wb + e->L[i1].wb_off, nullptr, e->S[b], dst, gs, e->L[i1].cout_pad, e->L[i1].cin_pad, 1, s));
    else       CHK_(AE_CONVS(e, AE_SLOT_DP(i1), 1.f, i1, AE_SLOT_DPOOL(4),
                             e->dp[i1], e->wb + e->L[i1].wb_off, e->zero_bias, nullptr, dst, gs, e->L[i1].cout_pad, e->L[i1].cin_pad, 2, s));   // the latent has no activation
  }
  // ---- encoder, last block first
  for (int b = 4; b >= 0; --b) {
    const int H = e->H[b], W = e->W[b], i0 = 2 * b, i2 = 2 * b + 1;
    const AeGeo g = geo_plain(H, W);
    CHK_(maxpool3s2_bwd(e->dP[b], e->idx[b], e->act[i2], e->dp[i2], H, W, e->L[i2].cout_pad, s, e->nclip, e->cs));
    // (the max-pool adjoint adds at most four window gradients into one pixel, times lrelu' <= 1: dp[i2] is bounded by 4 x the pooled gradient's maximum)
    CHK_(AE_CONVS(e, AE_SLOT_DPOOL(b), 4.f, i2, AE_SLOT_DP(i0),
                  e->dp[i2], e->wb + e->L[i2].wb_off, nullptr, e->act[i0], e->dp[i0], g, e->L[i2].cout_pad, e->L[i2].cin_pad, 1, s));
    if (b > 0) CHK_(AE_CONVS(e, AE_SLOT_DP(i0), 1.f, i0, AE_SLOT_DPOOL(b - 1),
                             e->dp[i0], e->wb + e->L[i0].wb_off, e->zero_bias, nullptr, e->dP[b - 1], g, e->L[i0].cout_pad, e->L[i0].cin_pad, 2, s));
  }
  // ---- all weight and bias gradients
  CHK_(ae_wgrad_launch(e, s, 0));
  // ---- Adam
  AeAdamArgs A;
  for (int i = 0; i < AE_NLAYER; ++i) {
    const AeLayer& l = e->L[i];
    A.L[i] = AeAdamLayer{e->part + l.part_off, e->dbp + l.dbp_off, l.nslab, l.w_off, l.b_off, l.wb_off, ilog2(l.cin_pad / 8), ilog2(l.cout_pad), l.cin, l.cout};
  }
  A.theta = e->theta; A.m = e->m; A.v = e->v; A.wb = e->wb; A.ctr = e->ctr; A.amax = e->amax;
  A.n_w = e->n_w; A.n_all = e->n_w + e->n_b; A.lr = e->lr; A.cs = e->cs;
  hipLaunchKernelGGL(ae_adam_kernel, dim3((A.n_all / 4 + 255) / 256, e->nclip), dim3(256), 0, s, A);        // (n_w and n_b are multiples of 4)
  return (int)hipGetLastError();
}

static int ae_wgrad_launch(AeEngine* e, hipStream_t s, int mode) {
  AeWgradJobs J;
  int nb = 0, n = 0;
  for (int lv = 0; lv < 5; ++lv)                                // dispatch order: big images (long waves) first
    for (int i = 0; i < AE_NLAYER; ++i) {
      const AeLayer& l = e->L[i];
      if (l.level != lv) continue;
      AeWgradJob& q = J.j[n];
      q.dy = e->dp[i]; q.x = e->xin[i]; q.partial = e->part + l.part_off; q.dbp = e->dbp + l.dbp_off;
      q.H = e->H[l.level]; q.W = e->W[l.level];
      q.cin = l.cin_pad; q.cout = l.cout_pad; q.nslab = l.nslab; q.slab_len = l.slab_len;
      q.nwave = 3 * l.nslab * (l.cout_pad / 32) * ((l.cin_pad + 31) / 32);
      J.first[n++] = nb;
      nb += q.nwave;
    }
  J.first[n] = nb; J.n = n; J.cs = e->cs;
  if (mode == 1) hipLaunchKernelGGL(ae_wgrad_multi_kernel<1>, dim3(nb, e->nclip), dim3(64), 0, s, J);
  else if (mode == 2) hipLaunchKernelGGL(ae_wgrad_multi_kernel<2>, dim3(nb, e->nclip), dim3(64), 0, s, J);
  else if (mode == 3) hipLaunchKernelGGL(ae_wgrad_multi_kernel<3>, dim3(nb, e->nclip), dim3(64), 0, s, J);
  else if (mode == 4) hipLaunchKernelGGL(ae_wgrad_multi_kernel<4>, dim3(nb, e->nclip), dim3(64), 0, s, J);
  else hipLaunchKernelGGL(ae_wgrad_multi_kernel<0>, dim3(nb, e->nclip), dim3(64), 0, s, J);
  return (int)hipGetLastError();
}

static AePackArgs ae_pack_args(const AeEngine* e) {
  AePackArgs A;
  for (int i = 0; i < AE_NLAYER; ++i) {
    const AeLayer& l = e->L[i];
    A.L[i] = AePackLayer{l.w_off, l.wb_off, l.b_off, l.flat_w, l.flat_b, ilog2(l.cin_pad / 8), ilog2(l.cout_pad), l.cin, l.cout, l.deconv};
  }
  A.n_w = e->n_w; A.n_all = e->n_w + e->n_b;
  return A;
}

static int ae_capture(AeEngine* e, hipStream_t s, int steps, hipGraphExec_t* out) {
  hipGraph_t g = nullptr;
  int rc = (int)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  if (rc) return rc;
  for (int i = 0; i < steps && !rc; ++i) rc = ae_train_step(e, s);
  const int ec = (int)hipStreamEndCapture(s, &g);
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (ec) return ec;
  const int ic = (int)hipGraphInstantiate(out, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (!ic) (void)hipGraphUpload(*out, s);
  return ic;
}

}  // namespace lemo

using namespace lemo;

extern "C" {

long long lemo_ae_ws_floats(int H, int W) {
  if (H < 2 || W < 2 || (long)H * W > (1l << 22)) return 0;
  AeEngine e;
  size_t total = 0;
  ae_layout(&e, H, W, nullptr, &total);
  return (long long)total;
}

int lemo_ae_n_param(void) {
  AeEngine e;
  size_t total = 0;
  ae_layout(&e, 64, 64, nullptr, &total);
  return e.n_flat;
}

void* lemo_ae_create(const lemo_ae_desc* d) {
  if (!d || !d->ws || d->H < 2 || d->W < 2 || (long)d->H * d->W > (1l << 22) || !(d->lr > 0.f)) return nullptr;
  if (ae_conv_init()) return nullptr;
  AeEngine* e = new (std::nothrow) AeEngine();
  if (!e) return nullptr;
  size_t total = 0;
  e->nclip = d->clips > 1 ? d->clips : 1;
  ae_layout(e, d->H, d->W, d->ws, &total, ae_slab_scale(e->nclip));     // (longer slabs only shrink the layout: lemo_ae_ws_floats stays the bound)
  e->cs = total;                                            // clip c = the same layout, c * total floats further
  if (e->nclip > 64 || (long long)(total * (size_t)e->nclip) > d->ws_floats) { delete e; return nullptr; }
  e->lr = d->lr;
  if (const char* a = getenv("LEMO_AE_ARITH")) e->f16 = strcmp(a, "f16") == 0;        // A/B and parity tests: the split-f16 convolutions
  return e;
}

void lemo_ae_destroy(void* h) {
  AeEngine* e = (AeEngine*)h;
  if (!e) return;
  for (int l = 0; l < 2; ++l) if (e->exec[l]) (void)hipGraphExecDestroy(e->exec[l]);
  delete e;
}

int lemo_ae_load_clip(void* h, int clip, const float* flat, const float* x, const float* moc, void* stream) {
  AeEngine* e = (AeEngine*)h;
  if (!e || !flat || !x || !moc || clip < 0 || clip >= e->nclip) return LEMO_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int n_all = e->n_w + e->n_b, H = e->H[0], W = e->W[0];
  const size_t o = (size_t)clip * e->cs;
  hipLaunchKernelGGL((ae_pack_kernel<false>), dim3((n_all + 255) / 256), dim3(256), 0, s, ae_pack_args(e), flat, e->theta + o, e->wb + o);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  if ((rc = (int)hipMemsetAsync(e->m + o, 0, sizeof(float) * n_all, s))) return rc;        // a fresh optimizer (opt_amass_temp.py:160-164)
  if ((rc = (int)hipMemsetAsync(e->v + o, 0, sizeof(float) * n_all, s))) return rc;
  if ((rc = (int)hipMemsetAsync(e->ctr + o, 0, sizeof(float) * 64, s))) return rc;
  if ((rc = (int)hipMemcpyAsync(e->moc + o, moc, sizeof(float) * H * W, hipMemcpyDeviceToDevice, s))) return rc;
  hipLaunchKernelGGL(ae_to_cg8p_kernel, dim3((4 * H * W + 255) / 256), dim3(256), 0, s, x, 4, H, W, e->x8 + o);
  // tensor maxima the split-f16 convolutions scale by: everything that no convolution of a step writes
  if ((rc = (int)hipMemsetAsync(e->amax + o, 0, sizeof(float) * AE_NSLOT, s))) return rc;
  {
    AeAbsJobs J{};
    J.src[0] = x; J.n[0] = 4 * H * W; J.dst[0] = e->amax + o + AE_SLOT_X8;
    J.src[1] = e->moc + o; J.n[1] = H * W; J.dst[1] = e->amax + o + AE_SLOT_LOSS;
    for (int i = 0; i < AE_NLAYER; ++i) {
      J.src[2 + i] = e->theta + o + e->L[i].w_off; J.n[2 + i] = 9 * e->L[i].cin_pad * e->L[i].cout_pad; J.dst[2 + i] = e->amax + o + AE_SLOT_W(i);
    }
    hipLaunchKernelGGL(ae_absmax_kernel, dim3(64, 2 + AE_NLAYER), dim3(256), 0, s, J);
  }
  e->loaded |= 1ull << clip;
  e->forward_valid = 0;
  return (int)hipGetLastError();
}
static bool ae_all_loaded(const AeEngine* e) {
  return e->loaded == (e->nclip >= 64 ? ~0ull : (1ull << e->nclip) - 1ull);
}
int lemo_ae_load(void* h, const float* flat, const float* x, const float* moc, void* stream) { return lemo_ae_load_clip(h, 0, flat, x, moc, stream); }

int lemo_ae_step(void* h, int n, int use_graph, void* stream) {
  AeEngine* e = (AeEngine*)h;
  if (!e || n < 0) return LEMO_ERR_ARG;
  if (!ae_all_loaded(e)) return LEMO_ERR_STATE;            // a step advances EVERY clip: one that was never loaded would train from stale parameters
  hipStream_t s = (hipStream_t)stream;
  if (n > 0) e->forward_valid = 0;
  if (!use_graph) {
    for (int i = 0; i < n; ++i) { const int rc = ae_train_step(e, s); if (rc) return rc; }
    return 0;
  }
  const int unroll[2] = {5, 1};
  int plan[2], left = n;
  for (int l = 0; l < 2; ++l) { plan[l] = left / unroll[l]; left -= plan[l] * unroll[l]; }
  for (int l = 0; l < 2; ++l)
    if (plan[l] && !e->exec[l]) { const int rc = ae_capture(e, s, unroll[l], &e->exec[l]); if (rc) return rc; }
  for (int l = 0; l < 2; ++l)
    for (int i = 0; i < plan[l]; ++i) { const int rc = (int)hipGraphLaunch(e->exec[l], s); if (rc) return rc; }
  return 0;
}

// eval forward of ALL clips (one set of launches), then clip `clip`'s reconstruction / latent are copied out; clip < 0: forward only
int lemo_ae_forward_clip(void* h, int clip, float* rec, float* z, void* stream) {
  AeEngine* e = (AeEngine*)h;
  if (!e || clip >= e->nclip || (clip >= 0 && !rec)) return LEMO_ERR_ARG;
  if (!ae_all_loaded(e)) return LEMO_ERR_STATE;
  hipStream_t s = (hipStream_t)stream;
  if (clip <= 0) { const int rc = ae_forward(e, s); if (rc) return rc; e->forward_valid = 1; }      // clip 0 (or -1) runs the forward; later clips read its results
  else if (!e->forward_valid) return LEMO_ERR_STATE;       // no eval forward since the last step / load: nothing valid to copy out
  if (clip < 0) return 0;
  const int H = e->H[0], W = e->W[0];
  const size_t o = (size_t)clip * e->cs;
  hipLaunchKernelGGL(ae_from_cg8p_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, (const float*)e->act[19] + o, 1, H, W, rec);
  if (z) hipLaunchKernelGGL(ae_from_cg8p_kernel, dim3((256 * e->H[5] * e->W[5] + 255) / 256), dim3(256), 0, s, (const float*)e->P[4] + o, 256, e->H[5], e->W[5], z);
  return (int)hipGetLastError();
}
int lemo_ae_forward(void* h, float* rec, float* z, void* stream) { return lemo_ae_forward_clip(h, 0, rec, z, stream); }

int lemo_ae_params_clip(void* h, int clip, float* flat_out, void* stream) {
  AeEngine* e = (AeEngine*)h;
  if (!e || !flat_out || clip < 0 || clip >= e->nclip) return LEMO_ERR_ARG;
  if (!(e->loaded >> clip & 1ull)) return LEMO_ERR_STATE;
  const int n_all = e->n_w + e->n_b;
  hipLaunchKernelGGL((ae_pack_kernel<true>), dim3((n_all + 255) / 256), dim3(256), 0, (hipStream_t)stream, ae_pack_args(e),
                     (const float*)e->theta + (size_t)clip * e->cs, flat_out, (float*)nullptr);
  return (int)hipGetLastError();
}
int lemo_ae_params(void* h, float* flat_out, void* stream) { return lemo_ae_params_clip(h, 0, flat_out, stream); }

/* diagnostics: the weight-gradient launch alone on the engine's current buffers (mode 0 product, 1 operands loaded once per wave,
   2 no MFMAs) */
int lemo_ae_wgrad_probe(void* h, int mode, void* stream) {
  AeEngine* e = (AeEngine*)h;
  if (!e || mode < 0 || mode > 4) return LEMO_ERR_ARG;
  if (!ae_all_loaded(e)) return LEMO_ERR_STATE;
  return ae_wgrad_launch(e, (hipStream_t)stream, mode);
}

/* one convolution of the engine on its own (tests, tools): plain geometry unless in_s / out_s = 2 (see AeGeo) */
int lemo_ae_conv(const float* in, const float* wt, const float* bias, const float* aux, float* out, int H, int W, int fineH, int fineW,
                 int in_s, int out_s, int cin, int cout, int epi, int mt, int pt, int ks, void* stream) {
  if (!in || !wt || !out || (epi != 1 && !bias) || (epi == 1 && !aux) || (in_s != 1 && in_s != 2) || (out_s != 1 && out_s != 2)) return LEMO_ERR_ARG;
  if (ae_conv_init()) return LEMO_ERR_STATE;
  AeGeo g = geo_plain(H, W);
  if (in_s == 2) { g.in_Wp = fineW + 2; g.in_HWp = (fineH + 2) * (fineW + 2); g.in_s = 2; g.aux_Wp = g.in_Wp; g.aux_HWp = g.in_HWp; g.aux_s = 2; }
  if (out_s == 2) { g.out_Wp = fineW + 2; g.out_HWp = (fineH + 2) * (fineW + 2); g.out_s = 2; }
  return ae_conv(in, wt, bias, aux, out, g, cin, cout, epi, (hipStream_t)stream, mt, pt, ks);
}

/* the same on the split-f16 kernels (round 6): amax_in [1] = max |in| (or a bound: x in_fac), wmax [1] = max |wt|, amax_out [1] receives
   max |out| of the launch by atomicMax (zero it first); cin >= 16 (mt 3: >= 32) */
int lemo_ae_conv_f16(const float* in, const float* wt, const float* bias, const float* aux, float* out, int H, int W, int fineH, int fineW,
                     int in_s, int out_s, int cin, int cout, int epi, int mt, int pt, int ks, const float* amax_in, float in_fac,
                     const float* wmax, float* amax_out, void* stream) {
  if (!in || !wt || !out || (epi != 1 && !bias) || (epi == 1 && !aux) || (in_s != 1 && in_s != 2) || (out_s != 1 && out_s != 2)) return LEMO_ERR_ARG;
  if (!amax_in || !wmax || !amax_out || !(in_fac >= 1.f) || cin < 16) return LEMO_ERR_ARG;
  if (ae_conv_init()) return LEMO_ERR_STATE;
  AeGeo g = geo_plain(H, W);
  if (in_s == 2) { g.in_Wp = fineW + 2; g.in_HWp = (fineH + 2) * (fineW + 2); g.in_s = 2; g.aux_Wp = g.in_Wp; g.aux_HWp = g.in_HWp; g.aux_s = 2; }
  if (out_s == 2) { g.out_Wp = fineW + 2; g.out_HWp = (fineH + 2) * (fineW + 2); g.out_s = 2; }
  const AeF16 q{amax_in, wmax, amax_out, in_fac};
  return ae_conv(in, wt, bias, aux, out, g, cin, cout, epi, (hipStream_t)stream, mt, pt, ks, 1, 0, &q);
}

}  // extern "C"
