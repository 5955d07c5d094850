// 3x3 convolutions of the smoothness encoder with 32 / 64 input and output channels (models/AE_sep.py:
// 11-30, 77-99) on the 16-bit matrix cores with fp32-accurate SPLIT operands.  Two arithmetic variants share
// one kernel body (template parameter NP = pieces per operand):
//
// NP = 3, "split-bf16" (conv variant 3): x = hi + mid + lo EXACTLY (hi = bf16(x), mid = bf16(x - hi),
//   lo = bf16(x - hi - mid); 3 x 8 significand bits) and
//     a*b  ~=  a_hi*b_lo + a_lo*b_hi + a_mid*b_mid + a_hi*b_mid + a_mid*b_hi + a_hi*b_hi
//   = six v_mfma_f32_32x32x16_bf16 per 16-deep k-chunk (each bf16 x bf16 product is exact in fp32; the three
//   dropped terms are < 2^-24 |ab|).  6/16 of the fp32 matrix pipe per MAC; error vs float64 5.5e-7 of max|out|
//   (fp32-MFMA kernel: 1.07e-6).
//
// NP = 2, "split-f16" (conv variant 4, the engines' default since round 3): the error-compensated two-piece
//   fp16 scheme (Markidis et al. 2018; Ootomo & Yokota 2022) -- x*s = hi + lo with hi = f16(x*s),
//   lo = f16(x*s - hi) (2 x 11 significand bits: the operand is carried to 2^-22 relative, rounding error
//   of the pair <= 2^-23 |x| typ.) and
//     a*b  ~=  a_hi*b_lo + a_lo*b_hi + a_hi*b_hi          (the dropped lo*lo term is < 2^-22 |ab|)
//   = THREE v_mfma_f32_32x32x16_f16 per k-chunk, fp32 accumulate: half the matrix work of NP = 3.  fp16 has
//   5 exponent bits, so operands are scaled by exact powers of two: the weights once on the host (pack header),
//   the activations PER WORKGROUP AND STAGING PHASE from the maximum of the tile the workgroup has just loaded
//   (max -> [2^14, 2^15): no overflow; every element within 2^-17 of the tile maximum keeps its 22 bits, smaller
//   ones carry an absolute error < 2^-39 of that maximum -- far below the fp32 accumulation rounding of the
//   sum they enter).  Scales never mix inside an accumulator: the accumulators are rescaled (exactly) between
//   the two k-chunk phases and un-scaled before the epilogue.  Nothing crosses kernels: any fp32 CG8P tensor
//   is a valid input.  Measured ON THE GPU through the encoder's 10 layers on a real marker image, every layer against torch
//   float64 (tools/enc_layer_numerics.py, profiles/r04_enc_layer_numerics.txt; round 3 cited a CPU emulation with a per-TENSOR
//   scale here, which the kernels never used): forward activations 2.4-6.3e-7 of the layer maximum (torch's own fp32 convolution
//   on the CPU: 3.0-9.2e-7; NP = 3: 3.2-8.2e-7; fp32-input MFMA: 5.6e-7-1.4e-6), backward-data maps 1.8-6.5e-7 per layer with
//   float64 inputs to every layer, d(image) 6.1e-7 (NP = 3: 7.5e-7, fp32 MFMA: 6.5e-7).
//
// Work decomposition, written for the 64 -> 64 layers (one 512-thread block per CU, all 256 CUs, no second
// wave of blocks; Cout 32 halves the waves, Cin 32 halves the k-chunks and needs one staging phase only):
//   block  = 128 consecutive pixels x 64 couts x K = 9 taps x 64 cin
//   wave w = pixel half (w&1: 64 px = 2 MFMA N-tiles) x cout half (w>>1&1: 32 = 1 M-tile) x K half
//            (w>>2: channel groups 4kh..4kh+3) -> 18 (chunk, tap) steps x 12 (NP 3) / 6 (NP 2) MFMAs, 32 accumulators
//   roles  : A = weights (M = cout), B = activations (N = pixel), lane half h takes channel group 2kc+h
// Activations: the block stages its 128-pixel tile + 3x3 halo of all 64 channels ONCE, converting
// fp32 CG8P -> NP 16-bit planes in LDS ([group 8][piece NP][pixel][8 x 16 bit]: 153 / 102 KB), in two phases so
// that the MFMAs of the first k-chunk overlap the global loads of the second.  Every B fragment is one
// conflict-free ds_read_b128.  Weights are pre-split on the host (w[kc][tap][mt][piece][lane][8]) and
// read straight from global memory/L2 as one coalesced 1 KB dwordx4 per fragment (only 2 waves share each
// fragment).  The two K halves are summed through LDS (4 KB per wave).
// The P % 128 remainder pixels are cut into 4 px x 4 cout patches, one per block, computed with plain
// fp32 FMAs (one output pair per wave) inside the latency shadow of the first staging loads - no tail
// blocks, exactly one block per CU.
//
// (Round 1-2 also carried a persistent multi-layer "chain" kernel -- per-tile flags + system-coherent activation
// traffic instead of kernel boundaries.  It measured 18.8 us per layer against 16.6 us for launches and was removed
// in round 3; DESIGN 8f keeps the measurement.)
#include "conv_common.hpp"
#include "conv_f16.hpp"

namespace lemo {

#define CV3_NPX 408
template <int CIN, int COUT, int NP> struct Cv3Cfg {
  static_assert((CIN == 32 || CIN == 64) && (COUT == 32 || COUT == 64) && (NP == 2 || NP == 3), "encoder layer shapes");
  static constexpr int PLANE = CV3_NPX * 16;                    // bytes of one (group, piece) plane
  static constexpr int GRP = NP * PLANE;
  static constexpr int MT = COUT / 32;                          // cout tiles = waves along cout
  static constexpr int NT = 256 * MT;                           // threads: (2 pixel halves) x MT x (2 K halves) waves
  static constexpr int NCC = CIN / 32;                          // 16-channel k-chunks per K half = staging phases
  static constexpr int GH = 2 * NCC;                            // channel groups per K half
  static constexpr int TILE_BYTES = (CIN / 8) * GRP;            // NP 3: 156,672 (Cin 64) / 78,336 (Cin 32); NP 2: 104,448 / 52,224
  static constexpr int SMEM_BYTES = TILE_BYTES + (NP == 2 ? 256 : 0);   // + per-wave maxima of the two staging phases
  static constexpr int NB = (4 * 2 * CV3_NPX + NT - 1) / NT;    // staging slots per thread and phase (4 groups)
  static constexpr int PPX = NT / 128;                          // pixels of a remainder patch (one (px, cout pair) per wave)
  static constexpr int CQ = COUT / 4;                           // cout quads
  static constexpr int NTI = CIN == 64 ? 9 : 5;                 // patch: taps per lane (Cin 32: lane half = tap parity)
};

// p / W for 0 <= p < 2^24 with magic = 2^32 / W + 1 (host): one v_mul_hi instead of the ~40-instruction
// runtime division (the prologue had 13 of them per thread: 3.6k of its 6.9k cycles were address math)
__device__ __forceinline__ int div_w(int p, unsigned magic) { return (int)__umulhi((unsigned)p, magic); }

// ---- split-f16 helpers (NP = 2): conv_f16.hpp (shared with the fused layer pairs, conv_pair_kernels.hip) ----

// One remainder patch = PPX px x 4 couts over K = 9 Cin, no LDS, no barrier: wave w owns pixel w>>1 and the
// cout pair 2*(w&1), +1; lane l takes channel l of every tap (Cin 64; Cin 32: channel l&31 of the taps of
// parity l>>5): 9 (5) activation + 18 (10) weight dwords in coalesced runs, fp32 FMAs, two wave sums.
// `load` and `finish` are separate so that the kernel can put the loads several taps ahead of their use.
template <int CIN> struct SplitPatch {
  static constexpr int NTI = CIN == 64 ? 9 : 5;
  float a[NTI], w0[NTI], w1[NTI];
  float e0, e1;                                                 // epilogue operands (bias or saved activation)
  int poff, co, valid;
};
template <int EPI, int CIN, int COUT, int NP>
__device__ __forceinline__ void split_patch_load(SplitPatch<CIN>& pt, const float* __restrict__ in, const float* __restrict__ wt,
                                                 const float* __restrict__ bias, const float* __restrict__ aux,
                                                 int W, unsigned wmagic, int Wp, int HWp, int P, int rem0, int patch) {
  typedef Cv3Cfg<CIN, COUT, NP> Cfg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pq = patch / Cfg::CQ, cq = patch - pq * Cfg::CQ;
  int p = rem0 + pq * Cfg::PPX + (wave >> 1);
  pt.valid = p < P;
  p = p < P ? p : P - 1;
  const int y = div_w(p, wmagic), x = p - y * W;
  pt.poff = (y + 1) * Wp + (x + 1);
  pt.co = cq * 4 + 2 * (wave & 1);
  const int c = lane & (CIN - 1);
  const unsigned ia = (unsigned)(c >> 3) * ((unsigned)HWp * 8u) + (unsigned)pt.poff * 8u + (c & 7);
  const float* wa = wt + ((unsigned)(c >> 3) * COUT + pt.co) * 8u + (c & 7);     // wt[tap][Cin/8][Cout][8]
#pragma unroll
  for (int m = 0; m < Cfg::NTI; ++m) {
    int t = CIN == 64 ? m : 2 * m + (lane >> 5);
    const bool live = t < 9;
    t = live ? t : 8;
    const float av = in[ia + (unsigned)(((t / 3 - 1) * Wp + (t % 3 - 1)) * 8)];
    pt.a[m] = live ? av : 0.f;
    pt.w0[m] = wa[t * (CIN * COUT)];
    pt.w1[m] = wa[t * (CIN * COUT) + 8];
  }
  // wave-uniform address -> scalar loads (lgkmcnt): as vector loads hipcc sinks them to their first use,
  // behind whatever vector loads were issued in between
  const int cou = __builtin_amdgcn_readfirstlane(pt.co), pou = __builtin_amdgcn_readfirstlane(pt.poff);
  const float* ep = EPI == 1 ? aux + ((size_t)(cou >> 3) * HWp + pou) * 8 + (cou & 7) : bias + cou;
  pt.e0 = ep[0];
  pt.e1 = ep[1];
}
template <int EPI, int CIN>
__device__ __forceinline__ void split_patch_finish(const SplitPatch<CIN>& pt, float* __restrict__ out, int HWp) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int t = 0; t < SplitPatch<CIN>::NTI; ++t) { s0 = fmaf(pt.a[t], pt.w0[t], s0); s1 = fmaf(pt.a[t], pt.w1[t], s1); }
  // bias / lrelu' enter BEFORE the wave sums (in straight-line code): used only inside the lane-0 store
  // block, hipcc sinks their loads into it
  const bool l0 = (threadIdx.x & 63) == 0;
  if (EPI == 1) { s0 *= lrelu_grad_from_out(pt.e0); s1 *= lrelu_grad_from_out(pt.e1); }
  else { s0 += l0 ? pt.e0 : 0.f; s1 += l0 ? pt.e1 : 0.f; }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (EPI == 0) { s0 = lrelu(s0); s1 = lrelu(s1); }
  if ((threadIdx.x & 63) == 0 && pt.valid) {                    // co is even: both couts in one 8-group
    float* o = out + ((unsigned)(pt.co >> 3) * HWp + pt.poff) * 8u + (pt.co & 7);
    o[0] = s0; o[1] = s1;
  }
}

// hand `give` to the partner wave through LDS, add the partner's tile to `keep` (fixed order: K half 0 +
// K half 1), epilogue, store 32 couts x 32 px
template <int EPI>
__device__ __forceinline__ void split_reduce_store(const f32x16& give, const f32x16& keep, float* mine, const float* theirs,
                                                   bool second, float* __restrict__ out, const float* __restrict__ bias,
                                                   const float* __restrict__ aux, int HWp, int poff, int m_base, int h) {
  // epilogue operands (bias, or the saved activation for lrelu') are requested before the LDS exchange:
  // their L2 round trip hides behind the barrier instead of following it
  float4 eo[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c0 = m_base + q * 8 + 4 * h;
    eo[q] = EPI == 1 ? ld4(aux + ((size_t)(c0 >> 3) * HWp + poff) * 8 + (c0 & 7)) : ld4(bias + c0);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) st4(mine + q * 256, make_float4(give[4 * q], give[4 * q + 1], give[4 * q + 2], give[4 * q + 3]));
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 t = ld4(theirs + q * 256);
    const int c0 = m_base + q * 8 + 4 * h;
    float4 v;
    v.x = second ? t.x + keep[4 * q] : keep[4 * q] + t.x;
    v.y = second ? t.y + keep[4 * q + 1] : keep[4 * q + 1] + t.y;
    v.z = second ? t.z + keep[4 * q + 2] : keep[4 * q + 2] + t.z;
    v.w = second ? t.w + keep[4 * q + 3] : keep[4 * q + 3] + t.w;
    if (EPI == 1) {
      v.x *= lrelu_grad_from_out(eo[q].x); v.y *= lrelu_grad_from_out(eo[q].y);
      v.z *= lrelu_grad_from_out(eo[q].z); v.w *= lrelu_grad_from_out(eo[q].w);
    } else {
      v.x += eo[q].x; v.y += eo[q].y; v.z += eo[q].z; v.w += eo[q].w;
      if (EPI == 0) { v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w); }
    }
    st4(out + ((unsigned)(c0 >> 3) * HWp + poff) * 8u + (c0 & 7), v);   // (an `nt` store doubles the next layer's fabric fetches for no gain: measured)
  }
}

// One layer of one tile: the whole kernel body.  w3: NP 3 -> bf16 pieces; NP 2 -> f16 pieces of (weight * 2^k), `winv` = 2^-k.
template <int EPI, int CIN, int COUT, int NP, bool DBG>
__device__ __forceinline__ void split_layer(const float* __restrict__ in, const uint4* __restrict__ w3, float winv, const float* __restrict__ wt,
                                            const float* __restrict__ bias, const float* __restrict__ aux, float* __restrict__ out,
                                            int H, int W, unsigned wmagic, int full_blocks, int tile,
                                            unsigned long long* __restrict__ dbg) {
  unsigned long long t_start = 0, t_pro = 0, t_loop = 0, t_mid0 = 0, t_mid1 = 0;
  if (DBG) t_start = __builtin_amdgcn_s_memtime();
  typedef Cv3Cfg<CIN, COUT, NP> Cfg;
  constexpr int PLANE = Cfg::PLANE, GRP = Cfg::GRP, MT = Cfg::MT, NT = Cfg::NT, NCC = Cfg::NCC, GH = Cfg::GH, NB = Cfg::NB;
  constexpr int NU = 9 * NCC;                                   // (chunk, tap) steps of one wave
  constexpr int NW = NT / 64;
  LEMO_DYN_SMEM(smem_f);
  unsigned char* smem = reinterpret_cast<unsigned char*>(smem_f);
  float* wmax = reinterpret_cast<float*>(smem + Cfg::TILE_BYTES);   // NP 2: [phase 2][wave <= 16] tile maxima
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int ph = wave & 1, ch = MT == 2 ? (wave >> 1) & 1 : 0, kh = wave >> (MT == 2 ? 2 : 1);
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const unsigned in_gstride = (unsigned)HWp * 8u;

  // Operand pipeline over the 9 NCC (chunk, tap) steps u: weight fragments come from L2 (~1 us away) and are
  // requested TWO steps ahead through a ring of three register sets; activation fragments come from LDS
  // one step ahead (two sets; not across the phase barrier).  The sched_barriers keep hipcc from sinking
  // the loads next to their uses.
#ifndef CV3_RA
#define CV3_RA 3
#endif
  constexpr int RA = CV3_RA;                                    // weight ring: requested RA - 1 steps ahead
  uint4 ra[RA][NP], rb[2][2][NP];
#define CV3_LOAD_A(SET, U)                                                                         \
  _Pragma("unroll") for (int s_ = 0; s_ < NP; ++s_)                                                \
    ra[SET][s_] = w3[(unsigned)((((NCC * kh + (U) / 9) * 9 + (U) % 9) * MT + ch) * NP + s_) * 64u + lane];
#define CV3_LOAD_B(SET, U)                                                                         \
  _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_)                                              \
    _Pragma("unroll") for (int s_ = 0; s_ < NP; ++s_)                                              \
      rb[SET][nt_][s_] = *reinterpret_cast<const uint4*>(                                         \
          smem + (2 * (NCC * kh + (U) / 9) + h) * GRP + s_ * PLANE + (li[nt_] + (((U) % 9) / 3 - 1) * Wp + (((U) % 9) % 3 - 1)) * 16);
#define CV3_MFMA1(SA, SETA, SB, SETB)                                                              \
  _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_) {                                            \
    if (NP == 3)                                                                                   \
      acc[nt_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[SETA][SA]), \
                                                         __builtin_bit_cast(bf16x8, rb[SETB][nt_][SB]), acc[nt_], 0, 0, 0); \
    else                                                                                           \
      acc[nt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[SETA][SA]),   \
                                                        __builtin_bit_cast(f16x8, rb[SETB][nt_][SB]), acc[nt_], 0, 0, 0); \
  }
  // smallest products first
#define CV3_MFMA(SETA, SETB)                                                                       \
  if (NP == 3) {                                                                                   \
    CV3_MFMA1(0, SETA, NP - 1, SETB) CV3_MFMA1(NP - 1, SETA, 0, SETB) CV3_MFMA1(1, SETA, 1, SETB)  \
    CV3_MFMA1(0, SETA, 1, SETB) CV3_MFMA1(1, SETA, 0, SETB) CV3_MFMA1(0, SETA, 0, SETB)            \
  } else {                                                                                         \
    CV3_MFMA1(0, SETA, 1, SETB) CV3_MFMA1(1, SETA, 0, SETB) CV3_MFMA1(0, SETA, 0, SETB)            \
  }

  // the first two weight fragments are requested before anything else (nothing depends on them; issuing the staging
  // loads first instead shortens the prologue in isolation but measured 0.5 % slower inside the engine, and moving
  // the remainder patch's loads into the prologue 1.5 % slower: same-box A/B with tools/gpu_ab_bench.sh)
#pragma unroll
  for (int u0 = 0; u0 < RA - 1; ++u0)
    if (u0 < NU) { CV3_LOAD_A(u0, u0) }

  // ---- remainder patch of this block: PPX px x 4 couts ---------------------------------------------
  const int rem0 = full_blocks * 128;
  const int npatch = ((P - rem0 + Cfg::PPX - 1) / Cfg::PPX) * Cfg::CQ;
  const bool has_patch = tile < npatch;                          // uniform
  // Its loads are issued at the first tap of the main loop and consumed four taps later: the VALU work
  // rides between MFMAs.  (Unconditional loads + straight-line code: inside an `if` hipcc merges the FMAs
  // back into the load block and waits there.)
  SplitPatch<CIN> pt;

  // ---- staging plan ---------------------------------------------------------------------------------
  const int pfirst = tile * 128, plast = pfirst + 127;
  const int yf = div_w(pfirst, wmagic), yl = div_w(plast, wmagic);
  const int q0 = (yf + 1) * Wp + (pfirst - yf * W + 1), q1 = (yl + 1) * Wp + (plast - yl * W + 1);
  const int qin = q0 - Wp - 1;                                   // first staged padded pixel
  const int npx = q1 - q0 + 2 * Wp + 3;                          // staged pixels (<= CV3_NPX, checked on host)
  const int nchB = 2 * npx;                                      // 16-B chunks (4 floats) per channel group
  // phase f stages groups {2f, 2f+1, GH+2f, GH+2f+1}: the f-th k-chunk of both K halves.
  // Slot c = tid + NT k covers chunk c % 816 of group c / 816 (constant stride = the LDS plane size, so the
  // map costs a handful of VALU ops: every prologue instruction is paid twice per SIMD with the MFMA pipe
  // idle); chunks past the tile's own nchB re-read its last one.  No predicates anywhere: a load whose only
  // use sits inside an `if` is sunk into it by hipcc and then waited for with vmcnt(0).
  unsigned offB[NB];
  int dstB[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    int c0 = (int)threadIdx.x + k * NT;
    c0 = c0 < 4 * 2 * CV3_NPX ? c0 : 4 * 2 * CV3_NPX - 1;        // surplus slots redo the last chunk (same data, same place)
    const int gg = c0 / (2 * CV3_NPX), c = c0 - gg * (2 * CV3_NPX);
    const int g0 = (gg >> 1) * GH + (gg & 1);
    const int cl = c < nchB ? c : nchB - 1;
    offB[k] = (unsigned)g0 * in_gstride + (unsigned)qin * 8u + (unsigned)cl * 4u;
    dstB[k] = g0 * GRP + c * 8;
  }
  float4 stB[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) stB[k] = ld4(in + offB[k]);

  // NP 2: the phase's power-of-two scale from the maximum of what this workgroup has just loaded (tile + halo, the
  // phase's 32 channels); sc[f] / sci[f] = scale / exact inverse of phase f
  float sc[2] = {1.f, 1.f}, sci[2] = {1.f, 1.f};
  if (NP == 2) {
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < NB; ++k) m = absmax4(stB[k], m);
    m = wave_max(m);
    if (lane == 0) wmax[wave] = m;
    __syncthreads();
    float mm = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) mm = fmaxf(mm, wmax[i]);
    f16_scale_for(mm, sc[0], sci[0]);
  }
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    if (NP == 3) {
      uint2 s0, s1, s2;
      split3x4(stB[k], s0, s1, s2);
      *reinterpret_cast<uint2*>(smem + dstB[k]) = s0;
      *reinterpret_cast<uint2*>(smem + dstB[k] + PLANE) = s1;
      *reinterpret_cast<uint2*>(smem + dstB[k] + (NP - 1) * PLANE) = s2;
    } else {
      uint2 s0, s1;
      split2x4(stB[k], sc[0], s0, s1);
      *reinterpret_cast<uint2*>(smem + dstB[k]) = s0;
      *reinterpret_cast<uint2*>(smem + dstB[k] + PLANE) = s1;
    }
  }
  // second phase (Cin 64): loads in flight during the first k-chunk's MFMAs
  if (NCC == 2) {
#pragma unroll
    for (int k = 0; k < NB; ++k) stB[k] = ld4(in + 2u * in_gstride + offB[k]);
  }
  __syncthreads();
  if (DBG) t_pro = __builtin_amdgcn_s_memtime();

  // ---- main loop -------------------------------------------------------------------------------------
  int li[2];
  int poffn[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int p = pfirst + ph * 64 + nt * 32 + j;
    const int y = div_w(p, wmagic), x = p - y * W;
    poffn[nt] = (y + 1) * Wp + (x + 1);
    li[nt] = poffn[nt] - qin;
  }
  f32x16 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

#pragma unroll
  for (int cc = 0; cc < NCC; ++cc) {
    CV3_LOAD_B((cc * 9) & 1, cc * 9)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int u = cc * 9 + tap;
      if (u + RA - 1 < NU) { CV3_LOAD_A((u + RA - 1) % RA, u + RA - 1) }
      if (tap + 1 < 9) { CV3_LOAD_B((u + 1) & 1, u + 1) }
      if (u == 0) {
        split_patch_load<EPI, CIN, COUT, NP>(pt, in, wt, bias, aux, W, wmagic, Wp, HWp, P, rem0, has_patch ? tile : 0);
        pt.valid = pt.valid && has_patch;
      }
      __builtin_amdgcn_sched_barrier(0);
      CV3_MFMA(u % RA, u & 1)
      if (u == 4) split_patch_finish<EPI, CIN>(pt, out, HWp);
      // NP 2, second phase: its maximum is published at tap 0 (the loads have had the first step to land) and
      // collected behind a workgroup barrier at tap 1, with this step's MFMAs already queued
      if (NP == 2 && cc == 0 && NCC == 2) {
        if (tap == 0) {
          float m = 0.f;
#pragma unroll
          for (int k = 0; k < NB; ++k) m = absmax4(stB[k], m);
          m = wave_max(m);
          if (lane == 0) wmax[16 + wave] = m;
        }
        if (tap == 1) {
          __syncthreads();
          float mm = 0.f;
#pragma unroll
          for (int i = 0; i < NW; ++i) mm = fmaxf(mm, wmax[16 + i]);
          f16_scale_after(mm, sc[0], sc[1], sci[1]);
        }
      }
      // second-phase staging spread over taps 2..8 (slot k at tap 2 + 7k/NB): its VALU ops and LDS writes
      // issue between MFMAs instead of in one MFMA-idle burst before the phase barrier
      if (cc == 0 && NCC == 2) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          if (2 + (7 * k) / NB != tap) continue;
          if (NP == 3) {
            uint2 s0, s1, s2;
            split3x4(stB[k], s0, s1, s2);
            *reinterpret_cast<uint2*>(smem + 2 * GRP + dstB[k]) = s0;
            *reinterpret_cast<uint2*>(smem + 2 * GRP + dstB[k] + PLANE) = s1;
            *reinterpret_cast<uint2*>(smem + 2 * GRP + dstB[k] + (NP - 1) * PLANE) = s2;
          } else {
            uint2 s0, s1;
            split2x4(stB[k], sc[1], s0, s1);
            *reinterpret_cast<uint2*>(smem + 2 * GRP + dstB[k]) = s0;
            *reinterpret_cast<uint2*>(smem + 2 * GRP + dstB[k] + PLANE) = s1;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (DBG && cc == 0) t_mid0 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (DBG && cc == 0) t_mid1 = __builtin_amdgcn_s_memtime();
    if (NP == 2 && NCC == 2 && cc == 0) {        // the accumulators hold scale-0 sums; the next k-chunk arrives in scale 1 (exact: powers of two)
      const float f = sc[1] * sci[0];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] *= f;
    }
  }
#undef CV3_LOAD_A
#undef CV3_LOAD_B
#undef CV3_MFMA1
#undef CV3_MFMA
  if (DBG) t_loop = __builtin_amdgcn_s_memtime();
  if (NP == 2) {                                 // back to the operands' own scale (exact)
    const float f = sci[NCC - 1] * winv;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] *= f;
  }

  // ---- sum the two K halves: wave kh keeps pixel tile nt = kh and hands the other one over ---------
  // (kh is wave-uniform: a scalar branch instead of 32 v_cndmask on the accumulators)
  {
    float* red = smem_f + (((ph * MT + ch) * 2) * 4) * 256 + lane * 4;  // [ph*MT+ch][kh][4][64][4] floats (<= 32 KB), tile LDS is dead
    const int m_base = ch * 32;
    if (__builtin_amdgcn_readfirstlane(kh))
      split_reduce_store<EPI>(acc[0], acc[1], red + 1024, red, true, out, bias, aux, HWp, poffn[1], m_base, h);
    else
      split_reduce_store<EPI>(acc[1], acc[0], red, red + 1024, false, out, bias, aux, HWp, poffn[0], m_base, h);
  }
  // shapes with more remainder patches than blocks: the rest, round-robin (not on the headline shapes)
  for (int patch = tile + full_blocks; patch < npatch; patch += full_blocks) {
    split_patch_load<EPI, CIN, COUT, NP>(pt, in, wt, bias, aux, W, wmagic, Wp, HWp, P, rem0, patch);
    split_patch_finish<EPI, CIN>(pt, out, HWp);
  }
  if (DBG && lane == 0) {           // census record, same format as conv3x3_mfma_v2_kernel
    unsigned long long* r = dbg + ((size_t)blockIdx.x * (NT / 64) + wave) * 8;
    r[0] = __builtin_amdgcn_s_getreg(63492);
    r[1] = __builtin_amdgcn_s_getreg(63508);
    r[2] = t_start; r[3] = __builtin_amdgcn_s_memtime(); r[4] = t_pro; r[5] = t_loop; r[6] = t_mid0; r[7] = t_mid1;
  }
}

template <int EPI, int CIN, int COUT, int NP, bool DBG>
__global__ void __launch_bounds__((Cv3Cfg<CIN, COUT, NP>::NT))
conv3x3_split_kernel(const float* __restrict__ in, const uint4* __restrict__ w3, float winv, const float* __restrict__ wt,
                     const float* __restrict__ bias, const float* __restrict__ aux, float* __restrict__ out,
                     int H, int W, unsigned wmagic, int full_blocks, unsigned long long* __restrict__ dbg) {
  // XCD-aware tile order (workgroup b runs on XCD b % 8): XCD x owns a contiguous run of tiles so that
  // vertically adjacent tiles share their halo rows in one L2
  int tile = (int)blockIdx.x;
  {
    const int q = full_blocks >> 3, r = full_blocks & 7, xcd = tile & 7, k = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  split_layer<EPI, CIN, COUT, NP, DBG>(in, w3, winv, wt, bias, aux, out, H, W, wmagic, full_blocks, tile, dbg);
}

int conv_split_init() {
  static int rc = -1;
  if (rc >= 0) return rc;
  rc = 0;
#define OPTIN(EPI_, CI_, CO_, NP_, DBG_) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_kernel<EPI_, CI_, CO_, NP_, DBG_>), hipFuncAttributeMaxDynamicSharedMemorySize, (Cv3Cfg<CI_, CO_, NP_>::SMEM_BYTES)); if (e != hipSuccess) rc = (int)e; }
#define OPTIN3(CI_, CO_, NP_) OPTIN(0, CI_, CO_, NP_, false) OPTIN(1, CI_, CO_, NP_, false) OPTIN(2, CI_, CO_, NP_, false)
  OPTIN3(64, 64, 3) OPTIN3(64, 32, 3) OPTIN3(32, 64, 3) OPTIN3(32, 32, 3) OPTIN(0, 64, 64, 3, true)
  OPTIN3(64, 64, 2) OPTIN3(64, 32, 2) OPTIN3(32, 64, 2) OPTIN3(32, 32, 2) OPTIN(0, 64, 64, 2, true)
#undef OPTIN3
#undef OPTIN
  return rc;
}

// shapes the split kernel takes; everything else stays on conv3x3_mfma_lds
bool conv3x3_split_supported(int H, int W, int cin, int cout) {
  if ((cin != 32 && cin != 64) || (cout != 32 && cout != 64) || H <= 0 || W <= 0) return false;
  const long P = (long)H * W;
  const long full = P / 128;
  if (full < 1 || P > (1l << 24)) return false;
  if (127 + 2 * (127 / W + 1) + 2 * (W + 2) + 3 > CV3_NPX) return false;
  return true;
}

// pieces = 3: w3 = bf16 pack (pack_conv3x3_split), winv ignored.  pieces = 2: w3 = f16 pack of weight * 2^k, winv = 2^-k.
int conv3x3_mfma_split(const float* in, const void* w3, const float* wt, const float* bias, const float* aux, float* out,
                       int H, int W, int cin, int cout, int epi, hipStream_t s, unsigned long long* dbg, int pieces, float winv) {
  if (!conv3x3_split_supported(H, W, cin, cout) || epi < 0 || epi > 2 || (pieces != 2 && pieces != 3)) return LEMO_ERR_SHAPE;
  if (pieces == 2 && !(winv > 0.f)) return LEMO_ERR_ARG;
  const int full = (H * W) / 128;
  if (int rc = conv_split_init()) return rc;
  const uint4* w3v = reinterpret_cast<const uint4*>(w3);
  const unsigned wmagic = (unsigned)((1ull << 32) / (unsigned)W + 1);       // exact for p < 2^32 / W (P <= 2^24 checked)
  if (dbg) {
    if (epi != 0 || cin != 64 || cout != 64) return LEMO_ERR_ARG;
    if (pieces == 3) hipLaunchKernelGGL((conv3x3_split_kernel<0, 64, 64, 3, true>), dim3(full), dim3(512), (Cv3Cfg<64, 64, 3>::SMEM_BYTES), s, in, w3v, winv, wt, bias, aux, out, H, W, wmagic, full, dbg);
    else hipLaunchKernelGGL((conv3x3_split_kernel<0, 64, 64, 2, true>), dim3(full), dim3(512), (Cv3Cfg<64, 64, 2>::SMEM_BYTES), s, in, w3v, winv, wt, bias, aux, out, H, W, wmagic, full, dbg);
    return (int)hipGetLastError();
  }
#define LAUNCH3(EPI_, CI_, CO_, NP_) hipLaunchKernelGGL((conv3x3_split_kernel<EPI_, CI_, CO_, NP_, false>), dim3(full), dim3((Cv3Cfg<CI_, CO_, NP_>::NT)), (Cv3Cfg<CI_, CO_, NP_>::SMEM_BYTES), s, in, w3v, winv, wt, bias, aux, out, H, W, wmagic, full, (unsigned long long*)nullptr)
#define LAUNCH_E(CI_, CO_, NP_) { if (epi == 0) LAUNCH3(0, CI_, CO_, NP_); else if (epi == 1) LAUNCH3(1, CI_, CO_, NP_); else LAUNCH3(2, CI_, CO_, NP_); }
#define LAUNCH_S(NP_) { if (cin == 64 && cout == 64) LAUNCH_E(64, 64, NP_) else if (cin == 64) LAUNCH_E(64, 32, NP_) else if (cout == 64) LAUNCH_E(32, 64, NP_) else LAUNCH_E(32, 32, NP_) }
  if (pieces == 3) LAUNCH_S(3) else LAUNCH_S(2)
#undef LAUNCH_S
#undef LAUNCH_E
#undef LAUNCH3
  return (int)hipGetLastError();
}

}  // namespace lemo
