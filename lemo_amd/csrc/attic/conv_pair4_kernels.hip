// Fused encoder layer pairs as FOUR-wave workgroups, two per CU ("conv variant 6"; VERDICT r04 next #1a: "two 4-wave workgroups per CU
// ... non-matrix phases of one hide behind the other's MFMAs" -- built and MEASURED in round 5; it loses to variant 5, DESIGN 11 has the
// census; kept selectable, never the default).
//
// Variant 5 (conv_pair_kernels.hip) runs ONE 8-wave workgroup per CU on a 10 x 14 tile: its census says the matrix pipe is busy 19.0 k of
// 36.1 k cycles -- in the other 17 k all eight waves stage, exchange K halves, find the tile maximum and convert together, three workgroup
// barriers with nothing else resident to issue MFMAs.  Here the tile is halved (5 x 14 outputs) and the workgroup is four waves, one per
// SIMD, 74 KB of LDS and <= 256 VGPRs, so that TWO workgroups share a CU and one's barrier phases can run under the other's MFMAs.
//   in   9 x 18 = 162 px  [group 8][piece 2][px][8 x f16] = 41,472 B   (two staging phases: the f-th k-chunk of both K halves)
//   mid  7 x 16 on pitch 18 (one zero pad column each side) = 126 slots, 32,256 B
//   49 x 10 = 490 workgroups for 245 x 134 (two per CU on 234 of the 256 CUs); 7 N-tile steps per 70 outputs against variant 5's 11 per
//   140: 1.27 x the matrix work per pixel (the halo of the shorter tile; half of the last mid and out N-tiles is idle).
// Waves: w = ch + 2 kh -- ch: cout half (M-tile), kh: K half.  Layer 1: ALL four mid N-tiles over k-chunks 2 kh, 2 kh + 1 (18 steps x
//   12 MFMAs); the K halves are summed through LDS (the dead input planes) in the fixed order kh 0 + kh 1, K half kh finishing mid
//   tiles 2 kh, 2 kh + 1.  Layer 2: all three out N-tiles (18 x 9 MFMAs), K half kh finishing out tile kh and quads 2 kh, 2 kh + 1 of
//   tile 2.  Same arithmetic, packs, lane mapping, epilogues AND summation order as variant 5: the results are bit-identical to it
//   (power-of-two tile scales do not change an fp16 split), tested at full size.
// What was measured (tools/pair4_check.py, profiles/r05_pair4_check.txt, r05_pair4_fullk_sweeps.txt): 25.5-26.5 us per pair against
//   variant 5's 22.4-23.5 on the same box, interleaved.  (1) The second workgroup of a CU (wave slot 1 = second half of the grid) is
//   starved, not interleaved: layer 1 takes 10.2 k cycles in slot 0 and 22.6 k in slot 1, lifetimes 29.5 k / 40.9 k -- the older wave
//   wins the issue arbitration whenever it has anything ready; s_setprio, start delays of 4-12 k cycles for the second half, weight
//   rings of 3-8 steps and one accumulator per product change nothing beyond +-2 %.  (2) A wave ALONE on its SIMD reaches 68-70 % of
//   the MFMA rate in these loops whatever the ring depth or dependency distance: 12 MFMAs (384 cycles) need 10 KB of operand
//   fragments (8 ds_read_b128 + 2 global_load_dwordx4 per lane group), and a step takes 384 + ~140 cycles -- the returning fragments
//   and the MFMAs of one wave do not overlap; two waves of ONE workgroup in the same loop (variant 5) overlap them to 82 %, two waves
//   of different workgroups do not get the chance.  (3) A first form in which every wave walked the full K of layer 1 (no exchange)
//   measured the same 26-27 us with twice the weight-fragment traffic.
#include "conv_common.hpp"
#include <cstdlib>
#include <type_traits>
#include "conv_f16.hpp"

namespace lemo {

constexpr int Q_TH = 5, Q_TW = 14;
constexpr int Q_INW = Q_TW + 4, Q_INH = Q_TH + 4, Q_NIN = Q_INW * Q_INH;               // 18 x 9 = 162
constexpr int Q_MIDW = Q_TW + 2, Q_MIDH = Q_TH + 2;                                    // 16 x 7
constexpr int Q_MIDP = Q_MIDW + 2, Q_NMIDP = Q_MIDP * Q_MIDH;                          // pitch 18, 126 slots
constexpr int Q_PL_IN = Q_NIN * 16, Q_GRP_IN = 2 * Q_PL_IN;
constexpr int Q_PL_MID = Q_NMIDP * 16, Q_GRP_MID = 2 * Q_PL_MID;
constexpr int Q_MID_OFF = 8 * Q_GRP_IN;                                                // 41,472
constexpr int Q_WMAX_OFF = Q_MID_OFF + 8 * Q_GRP_MID;                                  // 73,728
constexpr int Q_SMEM = Q_WMAX_OFF + 3 * 4 * 4;
constexpr int Q_NSLOT = 6;                                                             // staging slots per thread and phase
#ifndef LEMO_Q_RA
#define LEMO_Q_RA 3
#endif
constexpr int Q_RA = LEMO_Q_RA;          // weight-fragment ring: requested Q_RA - 1 steps ahead.  A wave ALONE on its SIMD issues a step (6 MFMAs) every 192 cycles:
                                         // depth 3 (variant 5: two waves per SIMD, 576 cycles per step pair) covers 384 cycles of an L2 round trip of ~700
static_assert(Q_MIDW == 16 && Q_INW == Q_MIDP, "N-tiles = 2 rows x 16 columns on grids of row pitch 18");
static_assert(4 * 2 * Q_NIN <= Q_NSLOT * 256, "staging slots");
static_assert(2 * Q_SMEM <= 160 * 1024, "two workgroups per CU");
static_assert(4 * 8 * 256 * 4 <= Q_MID_OFF, "the K-half exchange (8 quads per wave) fits the dead input planes");

struct Pair4Args {
  const float* in;
  const uint4* wA;
  const uint4* wB;
  const float *biasA, *biasB;
  const float *auxA, *auxB;
  float* mid;
  float* out;
  float winvA, winvB;
  int H, W, ntx, ntiles;
  unsigned long long* dbg;
  int prio_mode, delay;            // experiment knobs (LEMO_PAIR4_PRIO / LEMO_PAIR4_DELAY): see the kernel body
};

template <int U, int END> struct QSteps {
  template <class F> static __device__ __forceinline__ void run(F&& f) {
    f(std::integral_constant<int, U>{});
    QSteps<U + 1, END>::run(f);
  }
};
template <int END> struct QSteps<END, END> {
  template <class F> static __device__ __forceinline__ void run(F&&) {}
};

__device__ __forceinline__ int q_lane_col(int j) { return j < 16 ? j : ((j - 18) & 15); }

// K loop of one layer over ONE K half (k-chunks 2 kh, 2 kh + 1: 18 (chunk, tap) steps) for NT N-tiles: A = weight fragments from L2
// through a ring of Q_RA steps, B = activation fragments from the LDS planes at `bbase`, one step ahead.  Steps are compile-time
// constants (template recursion: with `#pragma unroll` and a deeper ring the unroller once left the loop rolled and the ring went to
// scratch).  mid_fn(u) runs after the MFMAs of step u were issued; end_fn(u) after the fence of step u.
template <int NT, int GRP, int PL, int PITCH, bool PHASED, typename MidFn, typename EndFn>
__device__ __forceinline__ void q_kloop(f32x16 (&acc)[NT], uint4 (&ra)[Q_RA][2], const uint4* __restrict__ w, const unsigned char* bbase,
                                        const int (&li)[NT], int kh, int ch, int lane, MidFn&& mid_fn, EndFn&& end_fn) {
  const int h = lane >> 5;
  uint4 rb[2][NT][2];
#define Q_LOAD_A(SET, U)                                                                               \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                     \
    ra[SET][s_] = w[(unsigned)((((2 * kh + (U) / 9) * 9 + (U) % 9) * 2 + ch) * 2 + s_) * 64u + lane];
#define Q_LOAD_B(SET, U)                                                                               \
  _Pragma("unroll") for (int nt_ = 0; nt_ < NT; ++nt_)                                                 \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                   \
      rb[SET][nt_][s_] = *reinterpret_cast<const uint4*>(                                              \
          bbase + (2 * (2 * kh + (U) / 9) + h) * GRP + s_ * PL + (li[nt_] + (((U) % 9) / 3 - 1) * PITCH + (((U) % 9) % 3 - 1)) * 16);
#define Q_MFMA1(U, SA, SB)                                                                             \
  _Pragma("unroll") for (int nt_ = 0; nt_ < NT; ++nt_)                                                 \
    acc[nt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[(U) % Q_RA][SA]),   \
                                                      __builtin_bit_cast(f16x8, rb[(U) & 1][nt_][SB]), acc[nt_], 0, 0, 0);
  Q_LOAD_B(0, 0)
  QSteps<0, 18>::run([&](auto uc) {
    constexpr int u = decltype(uc)::value;
    if (u + Q_RA - 1 < 18) { Q_LOAD_A((u + Q_RA - 1) % Q_RA, (u + Q_RA - 1 < 18 ? u + Q_RA - 1 : 17)) }
    // PHASED (layer 1): the second k-chunk's planes are written by OTHER waves during the first chunk and published by the barrier in
    // end_fn(8) -- its first fragments may only be requested after that barrier, not one step ahead (on the GPU a wave at step 8 read
    // planes a slower wave had not converted yet: 8e-2 errors that the emulator, which runs waves to each barrier in turn, cannot show)
    if (u + 1 < 18 && !(PHASED && u + 1 == 9)) { Q_LOAD_B((u + 1) & 1, (u + 1 < 18 ? u + 1 : 17)) }
    __builtin_amdgcn_sched_barrier(0);
    Q_MFMA1(u, 0, 1) Q_MFMA1(u, 1, 0) Q_MFMA1(u, 0, 0)            // smallest products first; NT >= 3 tiles between two uses of an accumulator
    mid_fn(u);
    __builtin_amdgcn_sched_barrier(0);
    end_fn(u);
    if (PHASED && u + 1 == 9) { Q_LOAD_B(1, 9) }
  });
#undef Q_LOAD_A
#undef Q_LOAD_B
#undef Q_MFMA1
}
// the first Q_RA - 1 weight fragments of a K loop over K half kh: requested as early as the caller knows the layer
__device__ __forceinline__ void q_preload_a(uint4 (&ra)[Q_RA][2], const uint4* __restrict__ w, int kh, int ch, int lane) {
#pragma unroll
  for (int u0 = 0; u0 < Q_RA - 1; ++u0)
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) ra[u0][s_] = w[(unsigned)((((2 * kh) * 9 + u0) * 2 + ch) * 2 + s_) * 64u + lane];
}

template <int EPI>
__device__ __forceinline__ float4 q_epilogue(float4 r, float4 eo) {
  if (EPI == 1) {
    r.x *= lrelu_grad_from_out(eo.x); r.y *= lrelu_grad_from_out(eo.y);
    r.z *= lrelu_grad_from_out(eo.z); r.w *= lrelu_grad_from_out(eo.w);
  } else {
    r.x = lrelu(r.x + eo.x); r.y = lrelu(r.y + eo.y); r.z = lrelu(r.z + eo.z); r.w = lrelu(r.w + eo.w);
  }
  return r;
}
__device__ __forceinline__ float4 q_quad(const f32x16& a, int q) { return make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]); }
__device__ __forceinline__ float4 q_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// K-half exchange, layer 1: K half KH keeps mid tiles 2 KH, 2 KH + 1 and hands the other two to its partner (8 quads each way)
template <int KH> __device__ __forceinline__ void q_give1(const f32x16 (&acc)[4], float* mine) {
#pragma unroll
  for (int i = 0; i < 8; ++i) st4(mine + i * 256, q_quad(acc[2 * (1 - KH) + (i >> 2)], i & 3));
}
template <int KH> __device__ __forceinline__ void q_keep1(const f32x16 (&acc)[4], const float* theirs, float4 (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float4 o = ld4(theirs + i * 256), m = q_quad(acc[2 * KH + (i >> 2)], i & 3);
    v[i] = KH ? q_add(o, m) : q_add(m, o);                 // fixed order kh 0 + kh 1
  }
}
// layer 2: K half KH keeps out tile KH and quads 2 KH, 2 KH + 1 of out tile 2 (6 quads each way)
template <int KH> __device__ __forceinline__ void q_give2(const f32x16 (&acc)[3], float* mine) {
#pragma unroll
  for (int i = 0; i < 4; ++i) st4(mine + i * 256, q_quad(acc[1 - KH], i));
#pragma unroll
  for (int i = 0; i < 2; ++i) st4(mine + (4 + i) * 256, q_quad(acc[2], 2 * (1 - KH) + i));
}
template <int KH> __device__ __forceinline__ void q_keep2(const f32x16 (&acc)[3], const float* theirs, float4 (&v)[6]) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float4 o = ld4(theirs + i * 256), m = i < 4 ? q_quad(acc[KH], i) : q_quad(acc[2], 2 * KH + (i - 4));
    v[i] = KH ? q_add(o, m) : q_add(m, o);
  }
}

template <int EPI, bool DBG>
__global__ void __launch_bounds__(256, 2)
conv3x3_pair4_kernel(Pair4Args a) {
  unsigned long long t_start = 0, t_pro = 0, t_l1 = 0, t_mx = 0, t_mid = 0, t_l2 = 0;
  if (DBG) t_start = __builtin_amdgcn_s_memtime();
  LEMO_DYN_SMEM(smem_f);
  unsigned char* smem = reinterpret_cast<unsigned char*>(smem_f);
  float* wmax = reinterpret_cast<float*>(smem + Q_WMAX_OFF);          // [phase 3][wave 4]
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int ch = wave & 1, kh = wave >> 1;
  const int KHu = __builtin_amdgcn_readfirstlane(kh);
  const int H = a.H, W = a.W, Wp = W + 2, HWp = (H + 2) * Wp;
  const unsigned in_gstride = (unsigned)HWp * 8u;
  int tile = (int)blockIdx.x;                                          // XCD-aware order: XCD x owns a contiguous run of tiles
  {
    const int q = a.ntiles >> 3, r = a.ntiles & 7, xcd = tile & 7, k = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int ty = tile / a.ntx, tx = tile - ty * a.ntx;
  const int y0 = ty * Q_TH, x0 = tx * Q_TW;
  {                                                                    // experiment knobs (tools/pair4_check.py; 0 / 0 in the product)
    const bool second = (int)blockIdx.x >= (a.ntiles + 1) / 2;          // (dispatch order: the second half of the grid takes the second slots)
    if (a.prio_mode == 2) { if (!second) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
    if (second) for (int i = 0; i < a.delay; ++i) __builtin_amdgcn_s_sleep(64);      // 64 x 64 = 4096 cycles per unit
  }
  uint4 ra[Q_RA][2];
  q_preload_a(ra, a.wA, kh, ch, lane);

  // ---- staging: phase f = channel groups {2 f, 2 f + 1, 4 + 2 f, 4 + 2 f + 1} (the f-th k-chunk of BOTH K halves) of the 9 x 18 input
  // tile; clamped coordinates land on the zero border ring (CG8P contract), pixels two steps out only feed mid pixels masked below
  unsigned offB[Q_NSLOT];
  int dstB[Q_NSLOT];
#pragma unroll
  for (int k = 0; k < Q_NSLOT; ++k) {
    int c0 = tid + k * 256;
    c0 = c0 < 4 * 2 * Q_NIN ? c0 : 4 * 2 * Q_NIN - 1;              // surplus slots redo the last chunk
    const int gg = c0 / (2 * Q_NIN), c = c0 - gg * (2 * Q_NIN);
    const int px = c >> 1, half = c & 1;
    const int r = px / Q_INW, col = px - r * Q_INW;
    int gy = y0 - 2 + r, gx = x0 - 2 + col;
    gy = (gy < -1 ? -1 : (gy > H ? H : gy)) + 1;
    gx = (gx < -1 ? -1 : (gx > W ? W : gx)) + 1;
    const int g0 = (gg >> 1) * 4 + (gg & 1);
    offB[k] = (unsigned)g0 * in_gstride + (unsigned)(gy * Wp + gx) * 8u + 4u * half;
    dstB[k] = g0 * Q_GRP_IN + px * 16 + 8 * half;
  }
  float4 stB[Q_NSLOT];
#pragma unroll
  for (int k = 0; k < Q_NSLOT; ++k) stB[k] = ld4(a.in + offB[k]);
  // the two pad columns of every mid plane hold zeros: 7 rows x 2 x 16 planes of 16 B
  if (tid < Q_MIDH * 2 * 16) {
    const int pl = tid / (Q_MIDH * 2), rc = tid - pl * (Q_MIDH * 2);
    *reinterpret_cast<uint4*>(smem + Q_MID_OFF + pl * Q_PL_MID + ((rc >> 1) * Q_MIDP + (rc & 1) * (Q_MIDP - 1)) * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  float sc[2] = {1.f, 1.f}, sci[2] = {1.f, 1.f};
  {
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < Q_NSLOT; ++k) m = absmax4(stB[k], m);
    m = wave_max(m);
    if (lane == 0) wmax[wave] = m;
    __syncthreads();
    float mm = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) mm = fmaxf(mm, wmax[i]);
    f16_scale_for(mm, sc[0], sci[0]);
  }
#pragma unroll
  for (int k = 0; k < Q_NSLOT; ++k) {
    uint2 s0, s1;
    split2x4(stB[k], sc[0], s0, s1);
    *reinterpret_cast<uint2*>(smem + dstB[k]) = s0;
    *reinterpret_cast<uint2*>(smem + dstB[k] + Q_PL_IN) = s1;
  }
#pragma unroll
  for (int k = 0; k < Q_NSLOT; ++k) stB[k] = ld4(a.in + 2u * in_gstride + offB[k]);      // second phase: in flight during the first k-chunk
  __syncthreads();
  if (DBG) t_pro = __builtin_amdgcn_s_memtime();

  // ---- layer 1: all four mid N-tiles (rows 2 t, 2 t + 1; row 7 does not exist: its lanes redo row 6 and are dropped) over this wave's K half
  int li[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int my_raw = 2 * nt + (j >> 4), my = my_raw < Q_MIDH ? my_raw : Q_MIDH - 1;
    li[nt] = (my + 1) * Q_INW + q_lane_col(j) + 1;
  }
  // geometry of the two tiles this wave FINISHES (2 kh, 2 kh + 1)
  int mp[2], mpoff[2];
  bool inimg[2], inner[2], exists[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int my_raw = 2 * (2 * kh + nt) + (j >> 4), mx = q_lane_col(j);
    const int my = my_raw < Q_MIDH ? my_raw : Q_MIDH - 1;
    const int y = y0 - 1 + my, x = x0 - 1 + mx;
    mp[nt] = my * Q_MIDP + mx + 1;
    exists[nt] = my_raw < Q_MIDH;
    inimg[nt] = exists[nt] && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    inner[nt] = inimg[nt] && my >= 1 && my <= Q_TH && mx >= 1 && mx <= Q_TW;
    const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y), xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
    mpoff[nt] = (yc + 1) * Wp + (xc + 1);
  }
  f32x16 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  float4 eo[8];                                            // epilogue operands of layer 1: requested near the end of the K loop
  q_kloop<4, Q_GRP_IN, Q_PL_IN, Q_INW, true>(
      acc, ra, a.wA, smem, li, kh, ch, lane,
      [&](int u) {
        // second staging phase: maximum published at step 0, collected behind a barrier at step 1, conversion spread over steps 2 .. 7
        if (u == 0) {
          float m = 0.f;
#pragma unroll
          for (int k = 0; k < Q_NSLOT; ++k) m = absmax4(stB[k], m);
          m = wave_max(m);
          if (lane == 0) wmax[4 + wave] = m;
        }
        if (u == 1) {
          __syncthreads();
          float mm = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) mm = fmaxf(mm, wmax[4 + i]);
          f16_scale_after(mm, sc[0], sc[1], sci[1]);
        }
#pragma unroll
        for (int k = 0; k < Q_NSLOT; ++k) {
          if (2 + k != u) continue;
          uint2 s0, s1;
          split2x4(stB[k], sc[1], s0, s1);
          *reinterpret_cast<uint2*>(smem + 2 * Q_GRP_IN + dstB[k]) = s0;
          *reinterpret_cast<uint2*>(smem + 2 * Q_GRP_IN + dstB[k] + Q_PL_IN) = s1;
        }
        if (u == 13) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int c0 = ch * 32 + (i & 3) * 8 + 4 * h;
            eo[i] = EPI == 1 ? ld4(a.auxA + ((size_t)(c0 >> 3) * HWp + mpoff[i >> 2]) * 8 + (c0 & 7)) : ld4(a.biasA + c0);
          }
        }
      },
      [&](int u) {
        if (u == 8) {
          __syncthreads();                 // the second phase's planes are complete; the accumulators hold scale-0 sums, the second k-chunk
          const float f = sc[1] * sci[0];  // arrives in scale 1 (exact: powers of two, ratio bounded by f16_scale_after)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] *= f;
        }
        if (u == 17) __syncthreads();      // every wave has read its last input fragment: the input planes become the exchange scratch
      });
  if (DBG) t_l1 = __builtin_amdgcn_s_memtime();
  q_preload_a(ra, a.wB, kh, ch, lane);                     // layer 2's first weight fragments travel during the exchange / epilogue
  {
    const float f = sci[1] * a.winvA;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] *= f;
  }
  // ---- K halves summed (through the dead input planes: [ch 2][writer kh 2][8 quads][64 lanes][4 floats] = 32 KB), epilogue, mask, tile maximum
  float* red = smem_f + lane * 4;
  float* mine = red + ((ch * 2 + KHu) * 8) * 256;
  const float* theirs = red + ((ch * 2 + (1 - KHu)) * 8) * 256;
  if (KHu) q_give1<1>(acc, mine); else q_give1<0>(acc, mine);
  __syncthreads();
  float4 v[8];
  if (KHu) q_keep1<1>(acc, theirs, v); else q_keep1<0>(acc, theirs, v);
  float mloc = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float4 r = q_epilogue<EPI>(v[i], eo[i]);
    if (!inimg[i >> 2]) r = make_float4(0.f, 0.f, 0.f, 0.f);          // zero padding of the second layer (and the row that does not exist)
    v[i] = r;
    mloc = absmax4(r, mloc);
  }
  mloc = wave_max(mloc);
  if (lane == 0) wmax[8 + wave] = mloc;
  __syncthreads();
  float sm, smi;
  {
    float mm = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) mm = fmaxf(mm, wmax[8 + i]);
    f16_scale_for(mm, sm, smi);
  }
  if (DBG) t_mx = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int nt = i >> 2, q = i & 3;
    if (!exists[nt]) continue;                              // (the lanes that redid row 6 do not store it a second time)
    uint2 s0, s1;
    split2x4(v[i], sm, s0, s1);
    unsigned char* d = smem + Q_MID_OFF + (ch * 4 + q) * Q_GRP_MID + mp[nt] * 16 + 8 * h;
    *reinterpret_cast<uint2*>(d) = s0;
    *reinterpret_cast<uint2*>(d + Q_PL_MID) = s1;
  }
  __syncthreads();
  if (DBG) t_mid = __builtin_amdgcn_s_memtime();

  // ---- layer 2: all three out N-tiles (rows 2 T, 2 T + 1; row 5 does not exist) over this wave's K half
  int lo[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    const int oy_raw = 2 * nt + (j >> 4), oy = oy_raw < Q_TH ? oy_raw : Q_TH - 1;
    lo[nt] = (oy + 1) * Q_MIDP + q_lane_col(j) + 1;
  }
  // geometry of what this wave finishes: slot 0 = out tile kh (4 quads), slot 1 = out tile 2 (quads 2 kh, 2 kh + 1)
  int poff[2];
  bool ok[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int T = nt == 0 ? kh : 2;
    const int oy_raw = 2 * T + (j >> 4), c = q_lane_col(j), ox = c - 1;
    const int oy = oy_raw < Q_TH ? oy_raw : Q_TH - 1;
    const int y = y0 + oy, x = x0 + ox;
    ok[nt] = oy_raw < Q_TH && ox >= 0 && ox < Q_TW && y < H && x < W;
    const int yc = y < H ? y : H - 1, xc = x < 0 ? 0 : (x < W ? x : W - 1);
    poff[nt] = (yc + 1) * Wp + (xc + 1);
  }
  f32x16 acc2[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[nt][r] = 0.f;
  float4 eo2[6];
  q_kloop<3, Q_GRP_MID, Q_PL_MID, Q_MIDP, false>(
      acc2, ra, a.wB, smem + Q_MID_OFF, lo, kh, ch, lane,
      [&](int u) {
        if (u != 13) return;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int q = i < 4 ? i : 2 * kh + (i - 4);
          const int c0 = ch * 32 + q * 8 + 4 * h;
          eo2[i] = EPI == 1 ? ld4(a.auxB + ((size_t)(c0 >> 3) * HWp + poff[i >> 2]) * 8 + (c0 & 7)) : ld4(a.biasB + c0);
        }
      },
      [](int) {});
  if (DBG) t_l2 = __builtin_amdgcn_s_memtime();
  {
    const float f2 = smi * a.winvB;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[nt][r] *= f2;
  }
  // (no barrier before these writes: the scratch aliases the INPUT planes, dead since layer 1, and two workgroup barriers lie between
  // layer 1's exchange reads and here)
  if (KHu) q_give2<1>(acc2, mine); else q_give2<0>(acc2, mine);
  __syncthreads();
  float4 r2[6];
  if (KHu) q_keep2<1>(acc2, theirs, r2); else q_keep2<0>(acc2, theirs, r2);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = i < 4 ? i : 2 * kh + (i - 4);
    const int c0 = ch * 32 + q * 8 + 4 * h;
    const float4 r = q_epilogue<EPI>(r2[i], eo2[i]);
    if (ok[i >> 2]) st4(a.out + ((size_t)(c0 >> 3) * HWp + poff[i >> 2]) * 8 + (c0 & 7), r);
  }
  if (EPI != 1) {                                          // the saved activation of the first layer (its 5 x 14 interior), at the very end:
#pragma unroll                                             // a store issued before the barriers would be waited for by each of them
    for (int i = 0; i < 8; ++i) {
      const int nt = i >> 2, q = i & 3;
      const int c0 = ch * 32 + q * 8 + 4 * h;
      if (inner[nt]) st4(a.mid + ((size_t)(c0 >> 3) * HWp + mpoff[nt]) * 8 + (c0 & 7), v[i]);
    }
  }
  if (DBG && lane == 0) {
    unsigned long long* r = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
    r[0] = __builtin_amdgcn_s_getreg(63492);
    r[1] = t_l2;
    r[2] = t_start; r[3] = __builtin_amdgcn_s_memtime(); r[4] = t_pro; r[5] = t_l1; r[6] = t_mid; r[7] = t_mx;
  }
}

static int conv_pair4_init() {
  static int rc = -1;
  if (rc >= 0) return rc;
  rc = 0;
#define OPTIN(EPI_, DBG_) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pair4_kernel<EPI_, DBG_>), hipFuncAttributeMaxDynamicSharedMemorySize, Q_SMEM); if (e != hipSuccess) rc = (int)e; }
  OPTIN(0, false) OPTIN(1, false) OPTIN(0, true)
#undef OPTIN
  return rc;
}

// same contract as conv3x3_pair_f16 (conv_pair_kernels.hip); `dbg` (forward only): 8 stamps per wave, 4 waves per workgroup
int conv3x3_pair4_f16(const float* in, const void* wA, float winvA, const float* biasA, const float* auxA, float* mid, const void* wB,
                      float winvB, const float* biasB, const float* auxB, float* out, int H, int W, int epi, hipStream_t s,
                      unsigned long long* dbg) {
  if (!conv3x3_pair_supported(H, W, 64, 64, 64) || (epi != 0 && epi != 1)) return LEMO_ERR_SHAPE;
  if (!in || !wA || !wB || !out || !(winvA > 0.f) || !(winvB > 0.f)) return LEMO_ERR_ARG;
  if (epi == 0 ? (!biasA || !biasB || !mid) : (!auxA || !auxB)) return LEMO_ERR_ARG;
  if (dbg && epi != 0) return LEMO_ERR_ARG;
  if (int rc = conv_pair4_init()) return rc;
  Pair4Args a{};
  a.in = in; a.wA = reinterpret_cast<const uint4*>(wA); a.wB = reinterpret_cast<const uint4*>(wB);
  a.biasA = biasA; a.biasB = biasB; a.auxA = auxA; a.auxB = auxB; a.mid = mid; a.out = out;
  a.winvA = winvA; a.winvB = winvB; a.H = H; a.W = W;
  a.ntx = (W + Q_TW - 1) / Q_TW;
  a.ntiles = a.ntx * ((H + Q_TH - 1) / Q_TH);
  a.dbg = dbg;
  {
    static const int pm = getenv("LEMO_PAIR4_PRIO") ? atoi(getenv("LEMO_PAIR4_PRIO")) : 0;
    static const int dl = getenv("LEMO_PAIR4_DELAY") ? atoi(getenv("LEMO_PAIR4_DELAY")) : 0;
    a.prio_mode = pm; a.delay = dl;
  }
  if (dbg) hipLaunchKernelGGL((conv3x3_pair4_kernel<0, true>), dim3(a.ntiles), dim3(256), Q_SMEM, s, a);
  else if (epi == 0) hipLaunchKernelGGL((conv3x3_pair4_kernel<0, false>), dim3(a.ntiles), dim3(256), Q_SMEM, s, a);
  else hipLaunchKernelGGL((conv3x3_pair4_kernel<1, false>), dim3(a.ntiles), dim3(256), Q_SMEM, s, a);
  return (int)hipGetLastError();
}

}  // namespace lemo
