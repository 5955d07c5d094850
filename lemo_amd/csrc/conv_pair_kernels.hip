// TWO 64 -> 64 3x3 layers of the smoothness encoder (models/AE_sep.py:11-30, 77-99) in ONE launch, forward or backward-data,
// on 2-D tiles with the intermediate activation kept in LDS ("conv variant 5").
//
// Why: a per-layer launch (conv_split_kernels.hip) stages a 1-D 128-pixel tile + halo = 404 px x 64 ch (3.1x duplication),
// pays a 1.9 us kernel boundary, one fabric fill and one write-back per layer, and its 3 us of matrix work only half overlaps the
// fill (11.6 us per layer where the matrix floor is 3).  Here a workgroup owns a TH x TW = 10 x 14 output tile:
//   in   (TH+4) x (TW+4) = 14 x 18 = 252 px staged once (1.8x), fp32 -> two fp16 pieces, [group 8][piece 2][px][8 x f16] = 63 KB
//   mid  (TH+2) x (TW+2) = 12 x 16 = 192 px = 6 MFMA N-tiles, computed with the first layer's weights, epilogue applied, zeroed
//        outside the image (the second layer's zero padding), split again with its own power-of-two scale, kept in LDS (48 KB);
//        the forward also writes its own 10 x 14 interior to HBM (the saved activation the backward pass needs)
//   out  10 x 14 = 140 px = 5 N-tiles (20 columns of the last one idle) from the mid planes with the second layer's weights.
// ceil(245 / 10) x ceil(134 / 14) = 250 workgroups of 8 waves: one per CU, one round.  Matrix work per pair 11 N-tiles x 2 M-tiles
// x 36 k-steps x 3 products = 594 MFMAs per SIMD (19 k cycles) against 2 x 216 for two single-layer launches: 1.375x the flops
// for one boundary, one fill, one drain and (backward) one write-back + one read fewer.
//
// Arithmetic = conv variant 4 (conv_f16.hpp): two error-compensated fp16 pieces per operand, three v_mfma_f32_32x32x16_f16 per
// 16-deep k-chunk, fp32 accumulate; weights pre-split on the host (the same packs w[kc][tap][mt][piece][lane][8] the single-layer
// kernel reads), activations scaled per workgroup (input: per staging phase; intermediate: one scale for the tile).
//
// Waves: w = ng + 2 ch + 4 kh -- ng: N-tile group, ch: cout half (M-tile), kh: K half (channel groups 4 kh .. 4 kh + 3).
//   layer 1: wave (ng, ch, kh) accumulates mid N-tiles 3 ng .. 3 ng + 2 over its K half; the two K halves are summed through LDS
//            (fixed order kh 0 + kh 1), each half finishing 1.5 tiles: bias + LeakyReLU / x lrelu'(saved activation), mask, store.
//   layer 2: N-tile group ng2 = ng ^ kh (tiles 0-2 | 3-4), so that the two waves of a SIMD (w, w + 4) carry 3 + 2 tiles.
#include "conv_common.hpp"
#include "conv_f16.hpp"

namespace lemo {

constexpr int CP_TH = 10, CP_TW = 14;
constexpr int CP_INW = CP_TW + 4, CP_INH = CP_TH + 4, CP_NIN = CP_INW * CP_INH;        // 18 x 14 = 252
constexpr int CP_MIDW = CP_TW + 2, CP_MIDH = CP_TH + 2, CP_NMID = CP_MIDW * CP_MIDH;   // 16 x 12 = 192
constexpr int CP_MIDP = CP_MIDW + 2, CP_NMIDP = CP_MIDP * CP_MIDH;                     // mid planes: row pitch 18 (one pad column each side), 216 slots
constexpr int CP_NOUT = CP_TH * CP_TW;                                                 // 140 real outputs of the 160 computed
constexpr int CP_PL_IN = CP_NIN * 16, CP_GRP_IN = 2 * CP_PL_IN;                        // bytes of a (group, piece) plane / of a group
constexpr int CP_PL_MID = CP_NMIDP * 16, CP_GRP_MID = 2 * CP_PL_MID;
constexpr int CP_MID_OFF = 8 * CP_GRP_IN;                                              // 64,512
constexpr int CP_WMAX_OFF = CP_MID_OFF + 8 * CP_GRP_MID;                               // 113,664
constexpr int CP_SMEM = CP_WMAX_OFF + 3 * 8 * 4;
constexpr int CP_NSLOT = 4;                                                            // staging slots per thread and phase
static_assert(CP_MIDW == 16 && CP_NMID == 6 * 32 && CP_INW == CP_MIDP, "N-tiles = 2 rows x 16 columns on grids of row pitch 18");
static_assert(4 * 2 * CP_NIN <= CP_NSLOT * 512 && CP_NOUT == 140, "staging slots");
static_assert(8 * 6 * 256 * 4 <= CP_MID_OFF, "the K-half exchange (6 quads per wave) fits the dead input planes");

struct PairArgs {
  const float* in;                 // CG8P, 64 channels
  const uint4* wA;                 // first layer of the launch: split-f16 pack, its inverse host scale
  const uint4* wB;                 // second layer
  const float *biasA, *biasB;      // EPI 0
  const float *auxA, *auxB;        // EPI 1: saved forward activations at the mid / out positions
  float* mid;                      // EPI 0: receives the intermediate activation (all 64 channels); EPI 1: unused
  float* out;
  float winvA, winvB;
  int H, W, ntx, ntiles;
  unsigned long long* dbg;
};

// which (tile, quad) pairs of its partial sums K-half KH finishes itself ("keep") and which it hands to the other half ("give"):
// NT tiles x 4 quads (a quad = 4 consecutive couts of one column = accumulator registers 4 q .. 4 q + 3).
// NT 3: KH keeps tile KH and half of tile 2 (quads 2 KH, 2 KH + 1); NT 2: KH keeps tile KH.
template <int NT> struct PairSplit {
  static constexpr int NQ = NT == 3 ? 6 : 4;
  static constexpr int keep_tile(int KH, int i) { return i < 4 ? KH : 2; }
  static constexpr int keep_quad(int KH, int i) { return i < 4 ? i : (i - 4) + 2 * KH; }
  static constexpr int give_tile(int KH, int i) { return i < 4 ? 1 - KH : 2; }
  static constexpr int give_quad(int KH, int i) { return i < 4 ? i : (i - 4) + 2 * (1 - KH); }
};

// hand the "give" quads to the partner wave through LDS, add the partner's to the "keep" quads in the fixed order kh 0 + kh 1
// (The __syncthreads() below is reached through differently specialised inlined copies of this function -- NT 3 | 2, KH 0 | 1 -- i.e. the
// waves of a workgroup meet at different s_barrier instructions.  gfx9's s_barrier counts waves, not call sites, and the host emulator
// does the same; ADVICE r04 asked for ONE call site at kernel scope.  Round 5 built that (exchange split into give / keep around a
// barrier in the kernel body, layer 2 split into two halves) and measured it: bit-identical results, but +1.0 us per pair on the same
// box, interleaved (fwd 25.6 vs 24.5, bwd 23.7 vs 22.8 us) and SQ_LDS_BANK_CONFLICT 5.2e5 -> 1.09e6 per launch (profiles/r05_pmc_summary.txt
// of commit 1121d81 vs r04) -- the merged control flow costs the layer-2 epilogue its schedule.  Reverted; the form below stays.)
template <int NT, int KH>
__device__ __forceinline__ void pair_exchange(const f32x16 (&acc)[3], float4 (&v)[6], float* red_mine, const float* red_theirs) {
  typedef PairSplit<NT> S;
#pragma unroll
  for (int i = 0; i < S::NQ; ++i) {
    const int t = S::give_tile(KH, i), q = S::give_quad(KH, i);
    st4(red_mine + i * 256, make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]));
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < S::NQ; ++i) {
    const int t = S::keep_tile(KH, i), q = S::keep_quad(KH, i);
    const float4 o = ld4(red_theirs + i * 256);
    const float4 m = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
    v[i] = KH ? make_float4(o.x + m.x, o.y + m.y, o.z + m.z, o.w + m.w) : make_float4(m.x + o.x, m.y + o.y, m.z + o.z, m.w + o.w);
  }
}

// An MFMA N-tile = 2 rows x 16 columns of a grid with row pitch 18 (layer 1: the 12 x 16 mid grid read from the 14 x 18 input
// planes; layer 2: the 10 x 16 "virtual" out grid -- columns -1 .. 14 of the out tile, two of them padding -- read from the mid planes,
// stored with the same pitch).  ds_read_b128 serves a wave in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32), and a
// group is conflict-free iff its 16 lanes hit 16 distinct 16-byte slots mod 16 (MI355X_MICROARCH.md, LDS).  With lane j <-> column j
// in both rows the second row sits 18 = 16 + 2 slots further and two slots of every group collide (measured: SQ_LDS_BANK_CONFLICT =
// 47 % of SQ_LDS_IDX_ACTIVE).  Rotating the second row's columns by 2 -- lane 16 + i <-> column (i - 2) mod 16 -- makes lane j's
// slot == j + const (mod 16) for all 32 lanes: conflict-free for every tap.
__device__ __forceinline__ int cp_lane_col(int j) { return j < 16 ? j : ((j - 18) & 15); }

#define CP_RA 3          // weight-fragment ring: requested CP_RA - 1 steps ahead (conv_split_kernels.hip: deeper measured slower)

// K loop of one layer over one K half (2 k-chunks x 9 taps = 18 steps) for NT N-tiles: A = weights from L2 through the ring,
// B = activation fragments from LDS planes at `bbase` (group stride GRP, piece stride PL, row pitch PITCH), one step ahead.
// STAGE (layer 1 only): the second staging phase rides inside the first k-chunk (see the kernel body).
// the first CP_RA - 1 weight fragments of a K loop: requested by the caller as early as it knows the layer (they come from L2,
// ~700 cycles away: layer 2's are requested before the K-half exchange of layer 1, not at the top of its own loop)
__device__ __forceinline__ void pair_preload_a(uint4 (&ra)[CP_RA][2], const uint4* __restrict__ w, int kh, int ch, int lane) {
#pragma unroll
  for (int u0 = 0; u0 < CP_RA - 1; ++u0)
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) ra[u0][s_] = w[(unsigned)((((2 * kh) * 9 + u0) * 2 + ch) * 2 + s_) * 64u + lane];
}

template <int NT, int GRP, int PL, int PITCH, typename MidFn, typename EndFn>
__device__ __forceinline__ void pair_kloop(f32x16 (&acc)[3], uint4 (&ra)[CP_RA][2], const uint4* __restrict__ w, const unsigned char* bbase,
                                           const int (&li)[3], int kh, int ch, int lane, MidFn&& mid_fn, EndFn&& end_fn) {
  const int h = lane >> 5;
  uint4 rb[2][3][2];
#define CP_LOAD_A(SET, U)                                                                              \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                     \
    ra[SET][s_] = w[(unsigned)((((2 * kh + (U) / 9) * 9 + (U) % 9) * 2 + ch) * 2 + s_) * 64u + lane];
#define CP_LOAD_B(SET, U)                                                                              \
  _Pragma("unroll") for (int nt_ = 0; nt_ < NT; ++nt_)                                                 \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                   \
      rb[SET][nt_][s_] = *reinterpret_cast<const uint4*>(                                              \
          bbase + (2 * (2 * kh + (U) / 9) + h) * GRP + s_ * PL + (li[nt_] + (((U) % 9) / 3 - 1) * PITCH + (((U) % 9) % 3 - 1)) * 16);
#define CP_MFMA1(SA, SETA, SB, SETB)                                                                   \
  _Pragma("unroll") for (int nt_ = 0; nt_ < NT; ++nt_)                                                 \
    acc[nt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[SETA][SA]),         \
                                                      __builtin_bit_cast(f16x8, rb[SETB][nt_][SB]), acc[nt_], 0, 0, 0);
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    CP_LOAD_B((cc * 9) & 1, cc * 9)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int u = cc * 9 + tap;
      if (u + CP_RA - 1 < 18) { CP_LOAD_A((u + CP_RA - 1) % CP_RA, u + CP_RA - 1) }
      if (tap + 1 < 9) { CP_LOAD_B((u + 1) & 1, u + 1) }
      __builtin_amdgcn_sched_barrier(0);
      CP_MFMA1(0, u % CP_RA, 1, u & 1) CP_MFMA1(1, u % CP_RA, 0, u & 1) CP_MFMA1(0, u % CP_RA, 0, u & 1)   // smallest products first
      mid_fn(cc, tap);
      __builtin_amdgcn_sched_barrier(0);
    }
    end_fn(cc);
  }
#undef CP_LOAD_A
#undef CP_LOAD_B
#undef CP_MFMA1
}

// second layer of the pair for a wave that carries NT (3 | 2) out N-tiles starting at tile T0
template <int EPI, int NT>
__device__ __forceinline__ void pair_layer2(const PairArgs& a, unsigned char* smem, float* smem_f, int y0, int x0, int T0, int ng2, int ch,
                                            int kh, int lane, float smi, uint4 (&ra)[CP_RA][2]) {
  const int j = lane & 31, h = lane >> 5;
  const int Wp = a.W + 2, HWp = (a.H + 2) * Wp;
  int lo[3] = {0, 0, 0}, poff[3] = {0, 0, 0};
  bool ok[3] = {false, false, false};
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    // out N-tile T = rows 2T, 2T + 1 of the tile x 16 virtual columns c <-> ox = c - 1 (ox = -1, 14: padding, computed and dropped)
    const int oy = 2 * (T0 + nt) + (j >> 4), c = cp_lane_col(j), ox = c - 1;
    lo[nt] = (oy + 1) * CP_MIDP + c + 1;
    const int y = y0 + oy, x = x0 + ox;
    ok[nt] = ox >= 0 && ox < CP_TW && y < a.H && x < a.W;
    const int yc = y < a.H ? y : a.H - 1, xc = x < 0 ? 0 : (x < a.W ? x : a.W - 1);
    poff[nt] = (yc + 1) * Wp + (xc + 1);
  }
  f32x16 acc[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  pair_kloop<NT, CP_GRP_MID, CP_PL_MID, CP_MIDP>(acc, ra, a.wB, smem + CP_MID_OFF, lo, kh, ch, lane, [](int, int) {}, [](int) {});
  {
    const float f = smi * a.winvB;                          // back to the operands' own scale (exact: powers of two)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] *= f;
  }
  typedef PairSplit<NT> S;
  // geometry of the quads this K half finishes: quads 0..3 belong to tile KH, quads 4..5 (NT 3 only) to tile 2
  const int KHu = __builtin_amdgcn_readfirstlane(kh);
  const int poA = KHu ? poff[1] : poff[0], poB = poff[2];
  const bool okA = KHu ? ok[1] : ok[0], okB = ok[2];
  // epilogue operands: requested before the exchange barrier (their round trip hides behind it)
  float4 eo[6];
#pragma unroll
  for (int i = 0; i < S::NQ; ++i) {
    const int q = KHu ? S::keep_quad(1, i) : S::keep_quad(0, i);
    const int c0 = ch * 32 + q * 8 + 4 * h;
    const int po = i < 4 ? poA : poB;
    eo[i] = EPI == 1 ? ld4(a.auxB + ((size_t)(c0 >> 3) * HWp + po) * 8 + (c0 & 7)) : ld4(a.biasB + c0);
  }
  float* red = smem_f + ((ng2 * 2 + ch) * 2) * 1536 + lane * 4;      // [slot (ng2, ch)][writer kh][6 quads][64 lanes][4]: input planes are dead
  float4 v[6];
  // (no barrier needed before the exchange: its scratch aliases the INPUT planes, dead since layer 1, and two workgroup barriers
  // lie between layer 1's exchange reads and these writes)
  if (KHu) pair_exchange<NT, 1>(acc, v, red + 1536, red);
  else pair_exchange<NT, 0>(acc, v, red, red + 1536);
#pragma unroll
  for (int i = 0; i < S::NQ; ++i) {
    const int q = KHu ? S::keep_quad(1, i) : S::keep_quad(0, i);
    const int c0 = ch * 32 + q * 8 + 4 * h;
    const int po = i < 4 ? poA : poB;
    const bool st = i < 4 ? okA : okB;
    float4 r = v[i];
    if (EPI == 1) {
      r.x *= lrelu_grad_from_out(eo[i].x); r.y *= lrelu_grad_from_out(eo[i].y);
      r.z *= lrelu_grad_from_out(eo[i].z); r.w *= lrelu_grad_from_out(eo[i].w);
    } else {
      r.x = lrelu(r.x + eo[i].x); r.y = lrelu(r.y + eo[i].y); r.z = lrelu(r.z + eo[i].z); r.w = lrelu(r.w + eo[i].w);
    }
    if (st) st4(a.out + ((size_t)(c0 >> 3) * HWp + po) * 8 + (c0 & 7), r);
  }
}

template <int EPI, bool DBG>
__global__ void __launch_bounds__(512)
conv3x3_pair_kernel(PairArgs a) {
  unsigned long long t_start = 0, t_pro = 0, t_l1 = 0, t_mid = 0, t_ex = 0, t_mx = 0;
  if (DBG) t_start = __builtin_amdgcn_s_memtime();
  LEMO_DYN_SMEM(smem_f);
  unsigned char* smem = reinterpret_cast<unsigned char*>(smem_f);
  float* wmax = reinterpret_cast<float*>(smem + CP_WMAX_OFF);        // [phase 3][wave 8] tile maxima
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int ng = wave & 1, ch = (wave >> 1) & 1, kh = wave >> 2;
  const int H = a.H, W = a.W, Wp = W + 2, HWp = (H + 2) * Wp;
  const unsigned in_gstride = (unsigned)HWp * 8u;
  // XCD-aware tile order (workgroup b runs on XCD b % 8): XCD x owns a contiguous run of tiles (neighbours share halos in one L2)
  int tile = (int)blockIdx.x;
  {
    const int q = a.ntiles >> 3, r = a.ntiles & 7, xcd = tile & 7, k = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int ty = tile / a.ntx, tx = tile - ty * a.ntx;
  const int y0 = ty * CP_TH, x0 = tx * CP_TW;

  uint4 ra[CP_RA][2];
  pair_preload_a(ra, a.wA, kh, ch, lane);                  // nothing depends on them: requested before anything else

  // ---- staging plan: phase f stages groups {2f, 2f+1, 4+2f, 4+2f+1} (the f-th k-chunk of both K halves) of the 14 x 18 input tile.
  // Pixels outside the image land on the zero border ring through CLAMPED coordinates (the ring is zero by the CG8P contract; a
  // pixel two steps out only feeds mid pixels that are masked below): no predicates, every load unconditional.
  unsigned offB[CP_NSLOT];
  int dstB[CP_NSLOT];
#pragma unroll
  for (int k = 0; k < CP_NSLOT; ++k) {
    int c0 = tid + k * 512;
    c0 = c0 < 4 * 2 * CP_NIN ? c0 : 4 * 2 * CP_NIN - 1;          // surplus slots redo the last chunk (same data, same place)
    const int gg = c0 / (2 * CP_NIN), c = c0 - gg * (2 * CP_NIN);
    const int px = c >> 1, half = c & 1;
    const int r = px / CP_INW, col = px - r * CP_INW;
    int gy = y0 - 2 + r, gx = x0 - 2 + col;
    gy = (gy < -1 ? -1 : (gy > H ? H : gy)) + 1;
    gx = (gx < -1 ? -1 : (gx > W ? W : gx)) + 1;
    const int g0 = (gg >> 1) * 4 + (gg & 1);
    offB[k] = (unsigned)g0 * in_gstride + (unsigned)(gy * Wp + gx) * 8u + 4u * half;
    dstB[k] = g0 * CP_GRP_IN + px * 16 + 8 * half;
  }
  float4 stB[CP_NSLOT];
#pragma unroll
  for (int k = 0; k < CP_NSLOT; ++k) stB[k] = ld4(a.in + offB[k]);
  // the two pad columns of every mid plane (read by layer 2's padding outputs only) hold zeros: 12 rows x 2 x 16 planes of 16 B
  if (tid < CP_MIDH * 2 * 16) {
    const int pl = tid / (CP_MIDH * 2), rc = tid - pl * (CP_MIDH * 2);
    *reinterpret_cast<uint4*>(smem + CP_MID_OFF + pl * CP_PL_MID + ((rc >> 1) * CP_MIDP + (rc & 1) * (CP_MIDP - 1)) * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  float sc[2] = {1.f, 1.f}, sci[2] = {1.f, 1.f};
  {
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < CP_NSLOT; ++k) m = absmax4(stB[k], m);
    m = wave_max(m);
    if (lane == 0) wmax[wave] = m;
    __syncthreads();
    float mm = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) mm = fmaxf(mm, wmax[i]);
    f16_scale_for(mm, sc[0], sci[0]);
  }
#pragma unroll
  for (int k = 0; k < CP_NSLOT; ++k) {
    uint2 s0, s1;
    split2x4(stB[k], sc[0], s0, s1);
    *reinterpret_cast<uint2*>(smem + dstB[k]) = s0;
    *reinterpret_cast<uint2*>(smem + dstB[k] + CP_PL_IN) = s1;
  }
#pragma unroll
  for (int k = 0; k < CP_NSLOT; ++k) stB[k] = ld4(a.in + 2u * in_gstride + offB[k]);      // second phase: in flight during the first k-chunk
  __syncthreads();
  if (DBG) t_pro = __builtin_amdgcn_s_memtime();

  // ---- layer 1: mid N-tiles 3 ng .. 3 ng + 2 over this wave's K half -----------------------------------------------------
  int li[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    const int my = 2 * (3 * ng + nt) + (j >> 4), mx = cp_lane_col(j);      // mid N-tile t = rows 2t, 2t + 1 x 16 columns (rotated)
    li[nt] = (my + 1) * CP_INW + mx + 1;
  }
  f32x16 acc[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  pair_kloop<3, CP_GRP_IN, CP_PL_IN, CP_INW>(
      acc, ra, a.wA, smem, li, kh, ch, lane,
      [&](int cc, int tap) {
        if (cc != 0) return;
        // the second phase's maximum is published at tap 0 and collected behind a workgroup barrier at tap 1 (that step's MFMAs are
        // already queued); its conversion is spread over taps 2.. so that the VALU work and LDS writes issue between MFMAs
        if (tap == 0) {
          float m = 0.f;
#pragma unroll
          for (int k = 0; k < CP_NSLOT; ++k) m = absmax4(stB[k], m);
          m = wave_max(m);
          if (lane == 0) wmax[8 + wave] = m;
        }
        if (tap == 1) {
          __syncthreads();
          float mm = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) mm = fmaxf(mm, wmax[8 + i]);
          f16_scale_after(mm, sc[0], sc[1], sci[1]);
        }
#pragma unroll
        for (int k = 0; k < CP_NSLOT; ++k) {
          if (2 + (7 * k) / CP_NSLOT != tap) continue;
          uint2 s0, s1;
          split2x4(stB[k], sc[1], s0, s1);
          *reinterpret_cast<uint2*>(smem + 2 * CP_GRP_IN + dstB[k]) = s0;
          *reinterpret_cast<uint2*>(smem + 2 * CP_GRP_IN + dstB[k] + CP_PL_IN) = s1;
        }
      },
      [&](int cc) {
        __syncthreads();
        if (cc == 0) {                   // the accumulators hold scale-0 sums; the next k-chunk arrives in scale 1 (exact: powers of two)
          const float f = sc[1] * sci[0];
#pragma unroll
          for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] *= f;
        }
      });
  if (DBG) t_l1 = __builtin_amdgcn_s_memtime();
  pair_preload_a(ra, a.wB, kh, ch, lane);                  // layer 2's first weight fragments travel during the exchange / epilogue
  {
    const float f = sci[1] * a.winvA;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] *= f;
  }

  // ---- K halves summed, layer-1 epilogue, mask, split, into the mid planes -------------------------------------------------
  const int KHu = __builtin_amdgcn_readfirstlane(kh);
  typedef PairSplit<3> S3;
  // geometry of the two tiles this wave finishes: slot 0 = tile KH, slot 1 = tile 2 (of its group)
  // (scalars, not arrays: a runtime-indexed private array goes to scratch)
  int mp0, mp1, mpoff0, mpoff1;
  bool inimg0, inimg1, inner0, inner1;
  {
    auto geom = [&](int t, int& p_, int& poff_, bool& inimg_, bool& inner_) {
      const int my = 2 * (3 * ng + t) + (j >> 4), mx = cp_lane_col(j);
      const int y = y0 - 1 + my, x = x0 - 1 + mx;
      p_ = my * CP_MIDP + mx + 1;                                    // slot in the mid planes (pitch 18, columns shifted by the pad)
      inimg_ = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
      inner_ = inimg_ && my >= 1 && my <= CP_TH && mx >= 1 && mx <= CP_TW;
      const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y), xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
      poff_ = (yc + 1) * Wp + (xc + 1);
    };
    geom(KHu, mp0, mpoff0, inimg0, inner0);
    geom(2, mp1, mpoff1, inimg1, inner1);
  }
  float4 eo[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = KHu ? S3::keep_quad(1, i) : S3::keep_quad(0, i);
    const int c0 = ch * 32 + q * 8 + 4 * h;
    const int po = i < 4 ? mpoff0 : mpoff1;
    eo[i] = EPI == 1 ? ld4(a.auxA + ((size_t)(c0 >> 3) * HWp + po) * 8 + (c0 & 7)) : ld4(a.biasA + c0);
  }
  float4 v[6];
  {
    float* red = smem_f + ((ng * 2 + ch) * 2) * 1536 + lane * 4;     // the input planes are dead (barrier at the end of the k loop)
    if (KHu) pair_exchange<3, 1>(acc, v, red + 1536, red);
    else pair_exchange<3, 0>(acc, v, red, red + 1536);
  }
  if (DBG) t_ex = __builtin_amdgcn_s_memtime();
  float mloc = 0.f;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const bool in_i = i < 4 ? inimg0 : inimg1;
    float4 r = v[i];
    if (EPI == 1) {
      r.x *= lrelu_grad_from_out(eo[i].x); r.y *= lrelu_grad_from_out(eo[i].y);
      r.z *= lrelu_grad_from_out(eo[i].z); r.w *= lrelu_grad_from_out(eo[i].w);
    } else {
      r.x = lrelu(r.x + eo[i].x); r.y = lrelu(r.y + eo[i].y); r.z = lrelu(r.z + eo[i].z); r.w = lrelu(r.w + eo[i].w);
    }
    if (!in_i) r = make_float4(0.f, 0.f, 0.f, 0.f);                    // zero padding of the second layer
    v[i] = r;          // (EPI 0: also the saved activation, written to HBM at the END of the kernel: a store issued here would be
                       // waited for -- vmcnt(0) -- by the workgroup barriers below, an HBM round trip per barrier)
    mloc = absmax4(r, mloc);
  }
  mloc = wave_max(mloc);
  if (lane == 0) wmax[16 + wave] = mloc;
  __syncthreads();
  float sm, smi;
  {
    float mm = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) mm = fmaxf(mm, wmax[16 + i]);
    f16_scale_for(mm, sm, smi);
  }
  if (DBG) t_mx = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = KHu ? S3::keep_quad(1, i) : S3::keep_quad(0, i);
    uint2 s0, s1;
    split2x4(v[i], sm, s0, s1);
    unsigned char* d = smem + CP_MID_OFF + (ch * 4 + q) * CP_GRP_MID + (i < 4 ? mp0 : mp1) * 16 + 8 * h;
    *reinterpret_cast<uint2*>(d) = s0;
    *reinterpret_cast<uint2*>(d + CP_PL_MID) = s1;
  }
  __syncthreads();
  if (DBG) t_mid = __builtin_amdgcn_s_memtime();

  // ---- layer 2 -------------------------------------------------------------------------------------------------------------------
  const int ng2 = __builtin_amdgcn_readfirstlane(ng ^ kh);
  if (ng2) pair_layer2<EPI, 2>(a, smem, smem_f, y0, x0, 3, 1, ch, kh, lane, smi, ra);
  else pair_layer2<EPI, 3>(a, smem, smem_f, y0, x0, 0, 0, ch, kh, lane, smi, ra);
  if (EPI != 1) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int q = KHu ? S3::keep_quad(1, i) : S3::keep_quad(0, i);
      const int c0 = ch * 32 + q * 8 + 4 * h;
      if (i < 4 ? inner0 : inner1) st4(a.mid + ((size_t)(c0 >> 3) * HWp + (i < 4 ? mpoff0 : mpoff1)) * 8 + (c0 & 7), v[i]);
    }
  }
  if (DBG && lane == 0) {
    unsigned long long* r = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
    r[0] = __builtin_amdgcn_s_getreg(63492);
    r[1] = t_ex;
    r[2] = t_start; r[3] = __builtin_amdgcn_s_memtime(); r[4] = t_pro; r[5] = t_l1; r[6] = t_mid; r[7] = t_mx;
  }
}

static int conv_pair_init() {
  static int rc = -1;
  if (rc >= 0) return rc;
  rc = 0;
#define OPTIN(EPI_, DBG_) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pair_kernel<EPI_, DBG_>), hipFuncAttributeMaxDynamicSharedMemorySize, CP_SMEM); if (e != hipSuccess) rc = (int)e; }
  OPTIN(0, false) OPTIN(1, false) OPTIN(0, true)
#undef OPTIN
  return rc;
}

bool conv3x3_pair_supported(int H, int W, int c0, int c1, int c2) {
  return c0 == 64 && c1 == 64 && c2 == 64 && H >= 1 && W >= 1 && (long)H * W <= (1l << 24);
}

// in -> [layer A] -> mid -> [layer B] -> out.  epi 0: both layers lrelu(conv + bias), `mid` receives the intermediate activation;
// epi 1: both layers conv * lrelu'(aux) (backward-data: wA / wB are the backward packs, auxA / auxB the saved activations at the
// mid / out positions), `mid` unused.  Packs and inverse scales: pack_conv3x3_split_f16 / pack_conv3x3_bwd_split_f16.
int conv3x3_pair_f16(const float* in, const void* wA, float winvA, const float* biasA, const float* auxA, float* mid, const void* wB,
                     float winvB, const float* biasB, const float* auxB, float* out, int H, int W, int epi, hipStream_t s,
                     unsigned long long* dbg) {
  if (!conv3x3_pair_supported(H, W, 64, 64, 64) || (epi != 0 && epi != 1)) return LEMO_ERR_SHAPE;
  if (!in || !wA || !wB || !out || !(winvA > 0.f) || !(winvB > 0.f)) return LEMO_ERR_ARG;
  if (epi == 0 ? (!biasA || !biasB || !mid) : (!auxA || !auxB)) return LEMO_ERR_ARG;
  if (dbg && epi != 0) return LEMO_ERR_ARG;
  if (int rc = conv_pair_init()) return rc;
  PairArgs a{};
  a.in = in; a.wA = reinterpret_cast<const uint4*>(wA); a.wB = reinterpret_cast<const uint4*>(wB);
  a.biasA = biasA; a.biasB = biasB; a.auxA = auxA; a.auxB = auxB; a.mid = mid; a.out = out;
  a.winvA = winvA; a.winvB = winvB; a.H = H; a.W = W;
  a.ntx = (W + CP_TW - 1) / CP_TW;
  a.ntiles = a.ntx * ((H + CP_TH - 1) / CP_TH);
  a.dbg = dbg;
  if (dbg) hipLaunchKernelGGL((conv3x3_pair_kernel<0, true>), dim3(a.ntiles), dim3(512), CP_SMEM, s, a);
  else if (epi == 0) hipLaunchKernelGGL((conv3x3_pair_kernel<0, false>), dim3(a.ntiles), dim3(512), CP_SMEM, s, a);
  else hipLaunchKernelGGL((conv3x3_pair_kernel<1, false>), dim3(a.ntiles), dim3(512), CP_SMEM, s, a);
  return (int)hipGetLastError();
}

}  // namespace lemo
