// 3x3 convolutions of the smoothness encoder with 32 / 64 input and output channels (models/AE_sep.py:
// 11-30, 77-99) on the bf16 matrix cores with EXACT fp32 operands ("split-bf16", conv variant 3).
//
// Every fp32 operand x is split into three bf16 pieces x = hi + mid + lo (hi = bf16(x),
// mid = bf16(x - hi), lo = bf16(x - hi - mid); 3 x 8 significand bits, the sum is exact) and
//     a*b  ~=  a_hi*b_lo + a_lo*b_hi + a_mid*b_mid + a_hi*b_mid + a_mid*b_hi + a_hi*b_hi
// is accumulated in fp32 by six v_mfma_f32_32x32x16_bf16 per 16-deep k-chunk.  Each bf16 x bf16
// product is exact in fp32; the three dropped terms are < 2^-24 |ab| each.  Measured on the encoder's
// own weights: dropped-term error 3e-9 of max|out| vs 3e-7 for the fp32 accumulation rounding that any
// fp32 convolution (the fp32 MFMA, cuDNN, an fmaf chain) carries; on gfx950 the 6-product sum of a
// K = 576 dot product has max error 7.4e-7 vs 8.2e-7 for v_mfma_f32_32x32x2_f32
// (tools/ubench/split_ubench.hip).  Six bf16 MFMAs run at 16x the fp32-MFMA rate, so one fp32-exact
// multiply-accumulate costs 6/16 of the fp32 matrix pipe: the 1e-5 loss-parity budget stays on fp32
// numerics while the MFMA floor of the layer drops from 18 us to 6.9 us.
//
// Work decomposition, written for the 64 -> 64 layers (one 512-thread block per CU, all 256 CUs, no second
// wave of blocks; Cout 32 halves the waves, Cin 32 halves the k-chunks and needs one staging phase only):
//   block  = 128 consecutive pixels x 64 couts x K = 9 taps x 64 cin
//   wave w = pixel half (w&1: 64 px = 2 MFMA N-tiles) x cout half (w>>1&1: 32 = 1 M-tile) x K half
//            (w>>2: channel groups 4kh..4kh+3) -> 18 (chunk, tap) steps x 12 MFMAs, 32 accumulators
//   roles  : A = weights (M = cout), B = activations (N = pixel), lane half h takes channel group 2kc+h
// Activations: the block stages its 128-pixel tile + 3x3 halo of all 64 channels ONCE, converting
// fp32 CG8P -> three bf16 planes in LDS ([group 8][split 3][pixel][8 bf16]: 153 KB), in two phases so
// that the MFMAs of the first k-chunk overlap the global loads of the second.  Every B fragment is one
// conflict-free ds_read_b128.  Weights are pre-split on the host (w3[kc][tap][mt][split][lane][8]) and
// read straight from global memory/L2 as one coalesced 1 KB dwordx4 per fragment (no LDS room left,
// and only 2 waves share each fragment).  The two K halves are summed through LDS (4 KB per wave).
// The P % 128 remainder pixels are cut into 4 px x 4 cout patches, one per block, computed with plain
// fp32 FMAs (one output pair per wave) inside the latency shadow of the first staging loads - no tail
// blocks, exactly one block per CU.
#include "conv_common.hpp"

namespace lemo {

#define CV3_NPX 408
template <int CIN, int COUT> struct Cv3Cfg {
  static_assert((CIN == 32 || CIN == 64) && (COUT == 32 || COUT == 64), "encoder layer shapes");
  static constexpr int PLANE = CV3_NPX * 16;                    // bytes of one (group, split) plane
  static constexpr int GRP = 3 * PLANE;
  static constexpr int MT = COUT / 32;                          // cout tiles = waves along cout
  static constexpr int NT = 256 * MT;                           // threads: (2 pixel halves) x MT x (2 K halves) waves
  static constexpr int NCC = CIN / 32;                          // 16-channel k-chunks per K half = staging phases
  static constexpr int GH = 2 * NCC;                            // channel groups per K half
  static constexpr int SMEM_BYTES = (CIN / 8) * GRP;            // 156,672 (Cin 64) / 78,336 (Cin 32)
  static constexpr int NB = (4 * 2 * CV3_NPX + NT - 1) / NT;    // staging slots per thread and phase (4 groups)
  static constexpr int PPX = NT / 128;                          // pixels of a remainder patch (one (px, cout pair) per wave)
  static constexpr int CQ = COUT / 4;                           // cout quads
  static constexpr int NTI = CIN == 64 ? 9 : 5;                 // patch: taps per lane (Cin 32: lane half = tap parity)
};

// p / W for 0 <= p < 2^24 with magic = 2^32 / W + 1 (host): one v_mul_hi instead of the ~40-instruction
// runtime division (the prologue had 13 of them per thread: 3.6k of its 6.9k cycles were address math)
__device__ __forceinline__ int div_w(int p, unsigned magic) { return (int)__umulhi((unsigned)p, magic); }

// Activation accessor.  COH = false: plain cached loads / stores (one kernel launch per layer; the kernel boundary
// makes the producer's writes visible).  COH = true: system-coherent (sc0 sc1) buffer loads / stores that go around
// the per-XCD L2s, for the persistent chain kernel below where layer l+1 reads what other workgroups (possibly on
// other XCDs) wrote in layer l without a kernel boundary or an L2 write-back / invalidate in between.
#define LEMO_AUX_SC 17                                          // gfx940+ cache policy: bit 0 = sc0, bit 4 = sc1
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <bool COH> struct ActIO {
  const float* p;
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ ActIO(const float* base, int bytes) : p(base) {
    if (COH) r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
  }
  __device__ __forceinline__ float4 ld4(unsigned foff) const {
    if (COH) return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(foff * 4u), 0, LEMO_AUX_SC));
    return ::ld4(p + foff);
  }
  __device__ __forceinline__ float ld1(unsigned foff) const {
    if (COH) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(foff * 4u), 0, LEMO_AUX_SC));
    return p[foff];
  }
  __device__ __forceinline__ void st4(unsigned foff, float4 v) const {
    if (COH) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, (int)(foff * 4u), 0, LEMO_AUX_SC);
    else ::st4(const_cast<float*>(p) + foff, v);
  }
  __device__ __forceinline__ void st2(unsigned foff, float a, float b) const {
    if (COH) { u32x2_t v = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)}; __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)(foff * 4u), 0, LEMO_AUX_SC); }
    else { float* o = const_cast<float*>(p) + foff; o[0] = a; o[1] = b; }
  }
};

// Synchronisation state of one layer inside the chain kernel (all pointers into the caller's `sync` buffer).
// flag[t] == epoch: tile t (and its remainder patch) of that layer is written and visible ; done == epoch * nblk:
// every workgroup has finished that layer.  Spins are bounded: on a timeout `err` is raised and the wait ends.
struct ChainCtx {
  const int* flag_prev; int* flag_cur;
  const int* done_prev; int* done_cur;
  int* err;
  int epoch, layer, nblk;
};
__device__ __forceinline__ void chain_spin(const int* p, int want, bool geq, int* err) {
  int spins = 0;
  for (;;) {
    const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (geq ? v >= want : v == want) break;
    __builtin_amdgcn_s_sleep(4);
    if (++spins > (1 << 21)) { *err = 1; break; }                // ~0.3 s: never hang the GPU
  }
}

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
template <int EPI, int CIN, int COUT, bool COH>
__device__ __forceinline__ void split_patch_load(SplitPatch<CIN>& pt, const ActIO<COH>& in, const float* __restrict__ wt,
                                                 const float* __restrict__ bias, const float* __restrict__ aux,
                                                 int W, unsigned wmagic, int Wp, int HWp, int P, int rem0, int patch) {
  typedef Cv3Cfg<CIN, COUT> Cfg;
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
    const float av = in.ld1(ia + (unsigned)(((t / 3 - 1) * Wp + (t % 3 - 1)) * 8));
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
template <int EPI, int CIN, bool COH>
__device__ __forceinline__ void split_patch_finish(const SplitPatch<CIN>& pt, const ActIO<COH>& out, int HWp) {
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
  if ((threadIdx.x & 63) == 0 && pt.valid)                      // co is even: both couts in one 8-group
    out.st2(((unsigned)(pt.co >> 3) * HWp + pt.poff) * 8u + (pt.co & 7), s0, s1);
}

// hand `give` to the partner wave through LDS, add the partner's tile to `keep` (fixed order: K half 0 +
// K half 1), epilogue, store 32 couts x 32 px
template <int EPI, bool COH>
__device__ __forceinline__ void split_reduce_store(const f32x16& give, const f32x16& keep, float* mine, const float* theirs,
                                                   bool second, const ActIO<COH>& out, const float* __restrict__ bias,
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
    out.st4(((unsigned)(c0 >> 3) * HWp + poff) * 8u + (c0 & 7), v);   // (an `nt` store doubles the next layer's fabric fetches for no gain: measured)
  }
}

// One layer of one tile: the whole kernel body.  COH = false: called once per launch (conv3x3_split_kernel).
// COH = true: called once per layer by the persistent chain kernel; activations go through coherent accesses
// and `cx` carries the neighbour-tile flags.
template <int EPI, int CIN, int COUT, bool DBG, bool COH>
__device__ __forceinline__ void split_layer(const float* __restrict__ in_p, const uint4* __restrict__ w3, const float* __restrict__ wt,
                                            const float* __restrict__ bias, const float* __restrict__ aux, float* __restrict__ out_p,
                                            int H, int W, unsigned wmagic, int full_blocks, int tile, const ChainCtx& cx,
                                            unsigned long long* __restrict__ dbg) {
  unsigned long long t_start = 0, t_pro = 0, t_loop = 0, t_mid0 = 0, t_mid1 = 0;
  if (DBG) t_start = __builtin_amdgcn_s_memtime();
  typedef Cv3Cfg<CIN, COUT> Cfg;
  constexpr int PLANE = Cfg::PLANE, GRP = Cfg::GRP, MT = Cfg::MT, NT = Cfg::NT, NCC = Cfg::NCC, GH = Cfg::GH, NB = Cfg::NB;
  constexpr int NU = 9 * NCC;                                   // (chunk, tap) steps of one wave
  LEMO_DYN_SMEM(smem_f);
  unsigned char* smem = reinterpret_cast<unsigned char*>(smem_f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int ph = wave & 1, ch = MT == 2 ? (wave >> 1) & 1 : 0, kh = wave >> (MT == 2 ? 2 : 1);
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const unsigned in_gstride = (unsigned)HWp * 8u;
  const ActIO<COH> in(in_p, (CIN / 8) * HWp * 32), out(out_p, (COUT / 8) * HWp * 32);

  // Operand pipeline over the 9 NCC (chunk, tap) steps u: weight fragments come from L2 (~1 us away) and are
  // requested TWO steps ahead through a ring of three register sets; activation fragments come from LDS
  // one step ahead (two sets; not across the phase barrier).  The sched_barriers keep hipcc from sinking
  // the loads next to their uses.
  uint4 ra[3][3], rb[2][2][3];
#define CV3_LOAD_A(SET, U)                                                                         \
  _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_)                                                 \
    ra[SET][s_] = w3[(unsigned)((((NCC * kh + (U) / 9) * 9 + (U) % 9) * MT + ch) * 3 + s_) * 64u + lane];
#define CV3_LOAD_B(SET, U)                                                                         \
  _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_)                                              \
    _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_)                                               \
      rb[SET][nt_][s_] = *reinterpret_cast<const uint4*>(                                         \
          smem + (2 * (NCC * kh + (U) / 9) + h) * GRP + s_ * PLANE + (li[nt_] + (((U) % 9) / 3 - 1) * Wp + (((U) % 9) % 3 - 1)) * 16);
#define CV3_MFMA1(SA, SETA, SB, SETB)                                                              \
  _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_)                                              \
    acc[nt_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[SETA][SA]),   \
                                                       __builtin_bit_cast(bf16x8, rb[SETB][nt_][SB]), acc[nt_], 0, 0, 0);
  // smallest products first
#define CV3_MFMA(SETA, SETB)                                                                       \
  CV3_MFMA1(0, SETA, 2, SETB) CV3_MFMA1(2, SETA, 0, SETB) CV3_MFMA1(1, SETA, 1, SETB)              \
  CV3_MFMA1(0, SETA, 1, SETB) CV3_MFMA1(1, SETA, 0, SETB) CV3_MFMA1(0, SETA, 0, SETB)

  // the first two weight fragments are requested before anything else (nothing depends on them; issuing the staging
  // loads first instead shortens the prologue in isolation but measured 0.5 % slower inside the engine, and moving
  // the remainder patch's loads into the prologue 1.5 % slower: same-box A/B with tools/gpu_ab_bench.sh)
  CV3_LOAD_A(0, 0)
  CV3_LOAD_A(1, 1)

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
  if (COH && cx.layer > 0) {
    // chain: the tiles this one reads (t-2 .. t+2: the halo is 137 > 128 pixels) must be written; tiles whose halo
    // reaches into the remainder pixels wait for every workgroup (the patches are spread over all of them)
    if (threadIdx.x < 5) {
      const int t = tile + (int)threadIdx.x - 2;
      if (t >= 0 && t < cx.nblk) chain_spin(cx.flag_prev + t, cx.epoch, false, cx.err);
    } else if (threadIdx.x == 64 && tile * 128 + 127 + W + 3 >= cx.nblk * 128) {
      chain_spin(cx.done_prev, cx.epoch * cx.nblk, true, cx.err);
    }
    __syncthreads();
  }
  float4 stB[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) stB[k] = in.ld4(offB[k]);

#pragma unroll
  for (int k = 0; k < NB; ++k) {
    uint2 s0, s1, s2;
    split3x4(stB[k], s0, s1, s2);
    *reinterpret_cast<uint2*>(smem + dstB[k]) = s0;
    *reinterpret_cast<uint2*>(smem + dstB[k] + PLANE) = s1;
    *reinterpret_cast<uint2*>(smem + dstB[k] + 2 * PLANE) = s2;
  }
  // second phase (Cin 64): loads in flight during the first k-chunk's MFMAs
  if (NCC == 2) {
#pragma unroll
    for (int k = 0; k < NB; ++k) stB[k] = in.ld4(2u * in_gstride + offB[k]);
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
      if (u + 2 < NU) { CV3_LOAD_A((u + 2) % 3, u + 2) }
      if (tap + 1 < 9) { CV3_LOAD_B((u + 1) & 1, u + 1) }
      if (u == 0 && !COH) {
        split_patch_load<EPI, CIN, COUT>(pt, in, wt, bias, aux, W, wmagic, Wp, HWp, P, rem0, has_patch ? tile : 0);
        pt.valid = pt.valid && has_patch;
      }
      __builtin_amdgcn_sched_barrier(0);
      CV3_MFMA(u % 3, u & 1)
      if (u == 4 && !COH) split_patch_finish<EPI, CIN>(pt, out, HWp);
      // second-phase staging spread over taps 2..8 (slot k at tap 2 + 7k/NB): its VALU ops and LDS writes
      // issue between MFMAs instead of in one MFMA-idle burst before the phase barrier
      if (cc == 0 && NCC == 2) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          if (2 + (7 * k) / NB != tap) continue;
          uint2 s0, s1, s2;
          split3x4(stB[k], s0, s1, s2);
          *reinterpret_cast<uint2*>(smem + 2 * GRP + dstB[k]) = s0;
          *reinterpret_cast<uint2*>(smem + 2 * GRP + dstB[k] + PLANE) = s1;
          *reinterpret_cast<uint2*>(smem + 2 * GRP + dstB[k] + 2 * PLANE) = s2;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (DBG && cc == 0) t_mid0 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (DBG && cc == 0) t_mid1 = __builtin_amdgcn_s_memtime();
  }
#undef CV3_LOAD_A
#undef CV3_LOAD_B
#undef CV3_MFMA1
#undef CV3_MFMA
  if (DBG) t_loop = __builtin_amdgcn_s_memtime();

  // chain: the remainder patch (of layers > 0) reads pixels written by the last tiles AND by every other workgroup's
  // patch of the previous layer, so it waits for that whole layer -- here, after the main loop, where the wait is
  // (almost always) already satisfied and the patch loads fly during the K-half exchange below
  if (COH) {
    if (cx.layer > 0) {
      if (threadIdx.x == 0) chain_spin(cx.done_prev, cx.epoch * cx.nblk, true, cx.err);
      __syncthreads();
    }
    split_patch_load<EPI, CIN, COUT>(pt, in, wt, bias, aux, W, wmagic, Wp, HWp, P, rem0, has_patch ? tile : 0);
    pt.valid = pt.valid && has_patch;
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
  if (COH) split_patch_finish<EPI, CIN>(pt, out, HWp);
  // shapes with more remainder patches than blocks: the rest, round-robin (not on the headline shapes)
  for (int patch = tile + full_blocks; patch < npatch; patch += full_blocks) {
    split_patch_load<EPI, CIN, COUT>(pt, in, wt, bias, aux, W, wmagic, Wp, HWp, P, rem0, patch);
    split_patch_finish<EPI, CIN>(pt, out, HWp);
  }
  if (COH) {
    // publish: every write-through store of this workgroup is acknowledged (vmcnt counts stores on gfx9), then
    // the tile flag and the layer counter; the barrier also protects the LDS tile against the next layer's staging
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_store(cx.flag_cur + tile, cx.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(cx.done_cur, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (DBG && lane == 0) {           // census record, same format as conv3x3_mfma_v2_kernel
    unsigned long long* r = dbg + ((size_t)blockIdx.x * (NT / 64) + wave) * 8;
    r[0] = __builtin_amdgcn_s_getreg(63492);
    r[1] = __builtin_amdgcn_s_getreg(63508);
    r[2] = t_start; r[3] = __builtin_amdgcn_s_memtime(); r[4] = t_pro; r[5] = t_loop; r[6] = t_mid0; r[7] = t_mid1;
  }
}

template <int EPI, int CIN, int COUT, bool DBG>
__global__ void __launch_bounds__((Cv3Cfg<CIN, COUT>::NT))
conv3x3_split_kernel(const float* __restrict__ in, const uint4* __restrict__ w3, const float* __restrict__ wt,
                     const float* __restrict__ bias, const float* __restrict__ aux, float* __restrict__ out,
                     int H, int W, unsigned wmagic, int full_blocks, unsigned long long* __restrict__ dbg) {
  // XCD-aware tile order (workgroup b runs on XCD b % 8): XCD x owns a contiguous run of tiles so that
  // vertically adjacent tiles share their halo rows in one L2
  int tile = (int)blockIdx.x;
  {
    const int q = full_blocks >> 3, r = full_blocks & 7, xcd = tile & 7, k = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  ChainCtx cx{};
  split_layer<EPI, CIN, COUT, DBG, false>(in, w3, wt, bias, aux, out, H, W, wmagic, full_blocks, tile, cx, dbg);
}

// ---- persistent chain: up to LEMO_CHAIN_MAX consecutive 64 -> 64 layers in ONE launch ------------------------
// One workgroup per CU keeps its tile position through all layers.  A kernel boundary between two layers costs
// ~6 us here (dispatch latency, end-of-kernel L2 write-back, a cold first read: tools/ubench/chain_ubench.hip
// measures 14.5 us per layer for launches vs 8.1 us for this scheme on a memory-only stand-in); layer l+1 of a tile
// only needs tiles t-2 .. t+2 of layer l, so the boundary is replaced by per-tile flags in global memory.
// Activations are read / written with system-coherent (sc0 sc1) buffer accesses -- no L2 write-back or invalidate,
// the cached weights stay cached -- and flags are polled with relaxed agent-scope loads.  Requires every workgroup
// to be resident at once (grid <= number of CUs; 153 KB of LDS pins one workgroup per CU); spins are bounded.
// MEASURED (tools/chain_check.py, 7 layers at 245x134): bit-identical to 7 launches, but 18.8 us per layer against
// 16.6 us for the launches (17.3 without the remainder patches): a flag makes a store -> memory -> poll round trip
// (~2 us) on top of the acknowledged write-through of the tile, the halo is re-read 3x from the fabric instead of
// partly from L2, and that is as much as the kernel boundary costs.  The fitting engine therefore launches per
// layer by default (lemo_fit_desc.conv_chain_sync = NULL); the chain stays as a tested option.
// sync layout (ints): [0] epoch of the last completed launch, [1] error flag, [2] finished workgroups (all epochs),
// [3 .. 3+n) layer counters, then n x nblk tile flags.  A sync buffer belongs to one (n, nblk) and is zero-initialised by the caller.
template <int EPI>
__global__ void __launch_bounds__(512)
conv3x3_split_chain_kernel(lemo_conv_chain c, int H, int W, unsigned wmagic, int full_blocks, int* __restrict__ sync) {
  int tile = (int)blockIdx.x;
  {
    const int q = full_blocks >> 3, r = full_blocks & 7, xcd = tile & 7, k = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int epoch = sync[0] + 1;                               // written by the previous launch's last workgroup
  int* done = sync + 3;
  int* flags = sync + 3 + c.n;
  for (int l = 0; l < c.n; ++l) {
    ChainCtx cx;
    cx.flag_prev = flags + (l > 0 ? l - 1 : 0) * full_blocks; cx.flag_cur = flags + l * full_blocks;
    cx.done_prev = done + (l > 0 ? l - 1 : 0); cx.done_cur = done + l;
    cx.err = sync + 1; cx.epoch = epoch; cx.layer = l; cx.nblk = full_blocks;
    split_layer<EPI, 64, 64, false, true>(c.in[l], reinterpret_cast<const uint4*>(c.w3[l]), c.wt[l], c.bias[l], c.aux[l], c.out[l],
                                         H, W, wmagic, full_blocks, tile, cx, nullptr);
  }
  // the last workgroup to finish closes the epoch for the next launch
  if (threadIdx.x == 0 &&
      __hip_atomic_fetch_add(sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch * full_blocks - 1)
    __hip_atomic_store(sync, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int conv_split_init() {
  static int rc = -1;
  if (rc >= 0) return rc;
  rc = 0;
#define OPTIN(EPI_, CI_, CO_, DBG_) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_kernel<EPI_, CI_, CO_, DBG_>), hipFuncAttributeMaxDynamicSharedMemorySize, (Cv3Cfg<CI_, CO_>::SMEM_BYTES)); if (e != hipSuccess) rc = (int)e; }
#define OPTIN3(CI_, CO_) OPTIN(0, CI_, CO_, false) OPTIN(1, CI_, CO_, false) OPTIN(2, CI_, CO_, false)
  OPTIN3(64, 64) OPTIN3(64, 32) OPTIN3(32, 64) OPTIN3(32, 32) OPTIN(0, 64, 64, true)
#undef OPTIN3
#undef OPTIN
#define OPTINC(EPI_) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_chain_kernel<EPI_>), hipFuncAttributeMaxDynamicSharedMemorySize, (Cv3Cfg<64, 64>::SMEM_BYTES)); if (e != hipSuccess) rc = (int)e; }
  OPTINC(0) OPTINC(1) OPTINC(2)
#undef OPTINC
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

int conv3x3_mfma_split(const float* in, const void* w3, const float* wt, const float* bias, const float* aux, float* out,
                       int H, int W, int cin, int cout, int epi, hipStream_t s, unsigned long long* dbg) {
  if (!conv3x3_split_supported(H, W, cin, cout) || epi < 0 || epi > 2) return LEMO_ERR_SHAPE;
  const int full = (H * W) / 128;
  if (int rc = conv_split_init()) return rc;
  const uint4* w3v = reinterpret_cast<const uint4*>(w3);
  const unsigned wmagic = (unsigned)((1ull << 32) / (unsigned)W + 1);       // exact for p < 2^32 / W (P <= 2^24 checked)
  if (dbg) {
    if (epi != 0 || cin != 64 || cout != 64) return LEMO_ERR_ARG;
    hipLaunchKernelGGL((conv3x3_split_kernel<0, 64, 64, true>), dim3(full), dim3(512), (Cv3Cfg<64, 64>::SMEM_BYTES), s, in, w3v, wt, bias, aux, out, H, W, wmagic, full, dbg);
    return (int)hipGetLastError();
  }
#define LAUNCH3(EPI_, CI_, CO_) hipLaunchKernelGGL((conv3x3_split_kernel<EPI_, CI_, CO_, false>), dim3(full), dim3((Cv3Cfg<CI_, CO_>::NT)), (Cv3Cfg<CI_, CO_>::SMEM_BYTES), s, in, w3v, wt, bias, aux, out, H, W, wmagic, full, (unsigned long long*)nullptr)
#define LAUNCH_E(CI_, CO_) { if (epi == 0) LAUNCH3(0, CI_, CO_); else if (epi == 1) LAUNCH3(1, CI_, CO_); else LAUNCH3(2, CI_, CO_); }
  if (cin == 64 && cout == 64) LAUNCH_E(64, 64)
  else if (cin == 64) LAUNCH_E(64, 32)
  else if (cout == 64) LAUNCH_E(32, 64)
  else LAUNCH_E(32, 32)
#undef LAUNCH_E
#undef LAUNCH3
  return (int)hipGetLastError();
}

// ---- chain launcher ---------------------------------------------------------------------------------------------
int conv3x3_split_chain_sync_ints(int H, int W, int n) {
  if (H <= 0 || W <= 0 || n < 1 || n > LEMO_CHAIN_MAX) return 0;
  return 3 + n + n * ((H * W) / 128);
}

// can the chain kernel run this shape on the current device?  (every workgroup must be resident at once)
bool conv3x3_split_chain_supported(int H, int W) {
  if (!conv3x3_split_supported(H, W, 64, 64)) return false;
  static int cus = -1;
  if (cus < 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) cus = 0;
    else cus = prop.multiProcessorCount;
  }
  return (H * W) / 128 <= cus;
}

int conv3x3_split_chain(const lemo_conv_chain& c, int H, int W, int epi, int* sync, bool force, hipStream_t s) {
  if (c.n < 1 || c.n > LEMO_CHAIN_MAX || epi < 0 || epi > 2 || !sync) return LEMO_ERR_ARG;
  if (!conv3x3_split_supported(H, W, 64, 64)) return LEMO_ERR_SHAPE;
  if (!force && !conv3x3_split_chain_supported(H, W)) return LEMO_ERR_STATE;
  for (int l = 0; l < c.n; ++l) {
    if (!c.in[l] || !c.w3[l] || !c.wt[l] || !c.out[l] || (epi != 1 && !c.bias[l]) || (epi == 1 && !c.aux[l])) return LEMO_ERR_ARG;
    if (l > 0 && c.in[l] != c.out[l - 1]) return LEMO_ERR_ARG;                 // a chain: layer l reads what l-1 wrote
  }
  const int full = (H * W) / 128;
  if (int rc = conv_split_init()) return rc;
  const unsigned wmagic = (unsigned)((1ull << 32) / (unsigned)W + 1);
#define LAUNCHC(EPI_) hipLaunchKernelGGL((conv3x3_split_chain_kernel<EPI_>), dim3(full), dim3(512), (Cv3Cfg<64, 64>::SMEM_BYTES), s, c, H, W, wmagic, full, sync)
  if (epi == 0) LAUNCHC(0); else if (epi == 1) LAUNCHC(1); else LAUNCHC(2);
#undef LAUNCHC
  return (int)hipGetLastError();
}

}  // namespace lemo
