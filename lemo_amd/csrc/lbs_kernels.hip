// Vertex stage of SMPL-X linear blend skinning (human_body_prior/body_model/lbs.py:81,94-99,108-117;
// smplx SMPLX.forward "+ transl") and its backward.
//
// Forward, fused in one kernel per 32-vertex tile:
//     v_posed = v_template + [shape coefs | pose feature] . D          (lbs.py:81 + :94-99; one
//                                                                       fp32-MFMA GEMM, K = 20+486 -> 512)
//     verts   = (sum_j W[v,j] A[b,j]) . [v_posed, 1] + transl           (lbs.py:108-117)
// GEMM roles: M = columns (3 per vertex; A operand = blend directions D, streamed ONCE from HBM,
// 64 MB, KG8 layout [K/8][3V][8] -> every load a contiguous 1 KiB dwordx4 run), N = frames
// (B operand = per-frame features Xg, L2 resident), so no (B,V,3) intermediate ever round-trips
// through HBM between the blend and the skinning: the GEMM tile is handed over through LDS.
// Skinning weights are held as ELL (<= KW non-zeros per vertex; real SMPL-X rows are sparse).
#include "kernels.hpp"
#include "loss_device.hpp"
#include "gemm_reduce.hpp"

namespace lemo {

#define LBS_VPB 42                // vertices per block -> 126 of 128 columns (4 MFMA M-tiles): V=10475 -> 250 blocks ~ 1 per CU
#define LBS_COLS 128
#define LBS_FR 128                // frames per block pass (4 N-tiles)
#define LBS_KC 8                  // 8-feature groups staged per LDS stage (64 features)
#define LBS_PITCH 129
#define LBS_STAGE_FLOATS (LBS_KC * 2 * 128 * 4)                   // one operand, one buffer: [KC][2 planes][128][4]
#define LBS_SMEM_BYTES (4 * LBS_STAGE_FLOATS * 4)                 // A,B x double buffer = 128 KB
#define LBS_SMEM_MAX 163840
// GEMM staging (128 KB), later aliased by [blend tile 128 x 129] + [2 x 12 frames of A]
static inline int lbs_smem_bytes(int nj) { const int b = (LBS_FR * LBS_PITCH + 2 * 12 * nj * 12) * 4; return b > LBS_SMEM_BYTES ? b : LBS_SMEM_BYTES; }

// 8 waves: wave w = (frame tile w&3, column-tile pair w>>2) -> 2 waves per SIMD.  Both GEMM operands go
// through LDS (register-staged, double-buffered, all loads of a stage issued up front); operand
// reads run one 2-group chunk ahead of the MFMAs that consume them.
// PRE (with SPLIT): the B operand (per-frame features, identical for every workgroup) was split into its three bf16
// pieces ONCE by the pose-stage kernel, in MFMA-fragment order (XgS[k-chunk][piece][frame][lane half][8 bf16]); the waves read
// their B fragments straight from L2 (1 KiB coalesced per fragment, one stage ahead) and only the streamed D operand goes
// through registers -> split -> LDS.  Without it every one of the 250 workgroups converted the same 128 x 512 features and
// both operands shared the LDS port: 196 KB of LDS traffic per 32-feature stage = as many cycles as its 48 MFMAs.
typedef _Float16 lbs_f16x8 __attribute__((ext_vector_type(8)));

// F16 (with SPLIT and PRE): BOTH operands arrive pre-split into two fp16 pieces -- D once on the host (SkinConst.DgH: the same 32
// bytes per (group, column) as the fp32 copy, [hi 4 | lo 4] per lane half, scaled by a power of two), the features by the pose
// kernel (XgS in its fp16 form) -- so the staging threads only move bytes (no conversion: the 3-piece bf16 split of the streamed
// operand was ~130 vector-ALU instructions per thread and 64-feature stage) and a 16-deep k-chunk is THREE fp16 MFMA products
// (hi hi + hi lo + lo hi, operands carried to 2^-22, fp32 accumulate) instead of six bf16 ones.
template <bool DBG, bool SPLIT, bool PRE = false, bool F16 = false, bool LOOP = false>
__global__ void __launch_bounds__(512)
lbs_verts_fwd_kernel(SkinConst c, const float* __restrict__ Xg, int Bp, const float* __restrict__ A, int nj,
                     const float* __restrict__ transl, const int* __restrict__ ids, int n, int B,
                     float* __restrict__ verts, float* __restrict__ v_posed, unsigned long long* __restrict__ dbg,
                     const unsigned short* __restrict__ XgS) {
  LEMO_DYN_SMEM(smem);
  unsigned long long t_start = 0, t_pro = 0, t_gemm = 0;
  if (DBG) t_start = __builtin_amdgcn_s_memtime();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  // a workgroup takes vertex tiles blockIdx.x, blockIdx.x + gridDim.x, ... (one tile each in the in-line launch; the fit engine's
  // side-branch launch runs 125 workgroups x 2 tiles so that the per-frame launches it overlaps keep half of the CUs: round 6)
  // (LOOP is a template flag: the in-line instantiations keep the straight-line body and its register allocation -- the looping bf16
  // instantiation spilled 56 registers, tests/test_resource_usage.py)
  const int ntile_x = (n + LBS_VPB - 1) / LBS_VPB;
  int tile_x = blockIdx.x;
  do {                                                 // `while (LOOP && ...)`: no loop at all in the in-line instantiations
  const int s0 = tile_x * LBS_VPB;                     // first vertex slot of this tile
  const int f0 = blockIdx.y * LBS_FR;                  // first frame of this pass
  const int nt = wave & 3, mp = wave >> 2;
  // ---- staging plan: thread -> fixed (column, half) of A and (frame, half) of B; groups (tid>>8) + 2k
  const int rem = tid & 255, scol = rem >> 1, shalf = rem & 1, sg0 = tid >> 8;
  size_t a_off;                                        // float offset of this thread's column in one group of Dg
  {
    int slot = s0 + scol / 3;
    if (slot >= n) slot = n - 1;
    const int vid = ids ? ids[slot] : slot;
    a_off = ((size_t)vid * 3 + (scol % 3)) * 8 + 4 * shalf;
  }
  const int fr_s = (f0 + scol < Bp) ? f0 + scol : Bp - 1;
  const size_t b_off = (size_t)fr_s * 8 + 4 * shalf;
  const size_t dg_stride = (size_t)c.NC * 8, xg_stride = (size_t)Bp * 8;
  const int lds_dst = ((sg0 * 2 + shalf) * 128 + scol) * 4;     // + k * (2 groups) ; same for A and B tiles
  f32x16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  if (SPLIT) {
    // ---- blend GEMM on the bf16 matrix cores with exact fp32 operands (3-way bf16 split, 6 products per
    // 16-deep k-chunk, fp32 accumulate: conv_split_kernels.hip has the error analysis).  The MFMA floor of
    // the 128 x 128 x 512 tile drops from 65.5k to 24.6k cycles; D (64 MB, streamed once) is converted on
    // the fly while it is staged, so HBM still sees 4 B per element.
    // LDS per buffer: A [4 groups][3 pieces][128 cols][8 bf16] (24 KB) + B likewise over frames; 2 buffers.
    constexpr int SG = 4, PL = 128 * 16, OPB = SG * 3 * PL, BUFB = 2 * OPB, NSTS = 64 / SG;
    unsigned char* sm = reinterpret_cast<unsigned char*>(smem);
    const int dstb = scol * 16 + shalf * 8;                       // + (group * 3 + piece) * PL
    if (PRE && F16) {
      const int fr = (f0 + nt * 32 + j < Bp) ? f0 + nt * 32 + j : Bp - 1;
      const unsigned char* xb = reinterpret_cast<const unsigned char*>(XgS) + ((size_t)fr * 2 + h) * 16;
      const size_t piece_b = (size_t)Bp * 32, chunk_b = 2 * piece_b;          // bytes per piece / per 16-feature chunk
      const float* DgH = reinterpret_cast<const float*>(c.DgH);               // same offsets as Dg: 32 B per (group, column)
      // one LDS stage = 64 features (8 groups x 2 pieces x 2 KB = 32 KB per buffer); D loads one stage ahead, B fragments one
      // 32-feature half ahead -- the schedule of the bf16 path below with a third fewer planes and half the MFMAs
      constexpr int SGB = 8, BUFA = SGB * 2 * PL, NSTB = 64 / SGB;
      float4 sa[4];
      uint4 rbg[2][2][2];                                                     // [half parity][chunk of the half][piece]
#define LBS_H_LOAD_A(ST)                                                                           \
      _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                \
        sa[k] = ld4(DgH + (size_t)((ST) * SGB + sg0 + 2 * k) * dg_stride + a_off);
#define LBS_H_LOAD_B(SET, G)                                                                       \
      _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_)                                             \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                           \
          rbg[SET][c_][s_] = *reinterpret_cast<const uint4*>(xb + (size_t)((G) * 2 + c_) * chunk_b + s_ * piece_b);
#define LBS_H_STORE_A(BUF)                                                                         \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                              \
        unsigned char* d = sm + (BUF) * BUFA + (sg0 + 2 * k) * 2 * PL + dstb;                      \
        *reinterpret_cast<float2*>(d) = make_float2(sa[k].x, sa[k].y);                             \
        *reinterpret_cast<float2*>(d + PL) = make_float2(sa[k].z, sa[k].w);                        \
      }
      LBS_H_LOAD_A(0)
      LBS_H_LOAD_B(0, 0)
      LBS_H_STORE_A(0)
      LBS_H_LOAD_A(1)
      __syncthreads();
      if (DBG) t_pro = __builtin_amdgcn_s_memtime();
      const int a_rdb = (mp * 64 + j) * 16;
#define LBS_H_READ(P, C)                                                                           \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                         \
          ra[C][0][s_] = *reinterpret_cast<const uint4*>(base + ((P) * 2 + (C)) * 4 * PL + s_ * PL + a_rdb);   \
          ra[C][1][s_] = *reinterpret_cast<const uint4*>(base + ((P) * 2 + (C)) * 4 * PL + s_ * PL + a_rdb + 32 * 16); \
        }
#define LBS_H_MFMA1(P, C, SA, SB)                                                                  \
        _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                           \
          acc[m_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(lbs_f16x8, ra[C][m_][SA]), \
                                                           __builtin_bit_cast(lbs_f16x8, rbg[P][C][SB]), acc[m_], 0, 0, 0);
#define LBS_H_MFMA(P, C) LBS_H_MFMA1(P, C, 1, 0) LBS_H_MFMA1(P, C, 0, 1) LBS_H_MFMA1(P, C, 0, 0)      /* small terms first */
#define LBS_H_HALF(P, ST) {                                                                        \
        if (2 * (ST) + (P) + 1 < 2 * NSTB) { LBS_H_LOAD_B(1 - (P), 2 * (ST) + (P) + 1) }           \
        uint4 ra[2][2][2];                                                                         \
        LBS_H_READ(P, 0)                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        LBS_H_READ(P, 1)                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        LBS_H_MFMA(P, 0)                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        LBS_H_MFMA(P, 1)                                                                           \
      }
      for (int st = 0; st < NSTB; ++st) {
        const unsigned char* base = sm + (st & 1) * BUFA + h * 2 * PL;
        LBS_H_HALF(0, st)
        __builtin_amdgcn_sched_barrier(0);
        LBS_H_HALF(1, st)
        if (st + 1 < NSTB) {
          LBS_H_STORE_A((st + 1) & 1)
          if (st + 2 < NSTB) { LBS_H_LOAD_A(st + 2) }
        }
        __syncthreads();
      }
#undef LBS_H_HALF
#undef LBS_H_READ
#undef LBS_H_MFMA1
#undef LBS_H_MFMA
#undef LBS_H_LOAD_A
#undef LBS_H_LOAD_B
#undef LBS_H_STORE_A
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] *= c.dgh_inv;       // D was scaled by a power of two: exact
    } else if (PRE) {
      // ---- D through LDS (as below), B fragments from the pre-split copy in L2 ----
      const int fr = (f0 + nt * 32 + j < Bp) ? f0 + nt * 32 + j : Bp - 1;
      const unsigned char* xb = reinterpret_cast<const unsigned char*>(XgS) + ((size_t)fr * 2 + h) * 16;
      const size_t piece_b = (size_t)Bp * 32, chunk_b = 3 * piece_b;          // bytes per piece / per 16-feature chunk
      // One LDS stage = 64 features (8 groups, 48 KB per buffer now that only D is staged) = two 32-feature halves: the block
      // barrier, the split + LDS store burst and the exposed LDS read latency behind it are paid 8 times instead of 16
      // (a 32-feature stage took 2.7 k cycles for 1.5 k cycles of MFMAs; three stages of D loads in flight instead of two
      // changed nothing, so it was not HBM latency).  D loads run one stage (= 64 features, as many bytes as before) ahead,
      // B fragments one half ahead.
      constexpr int SGB = 8, BUFA = SGB * 3 * PL, NSTB = 64 / SGB;
      float4 sa[4];
      uint4 rbg[2][2][3];                                                     // [half parity][chunk of the half][piece]
#define LBS_PRE_LOAD_A(ST)                                                                         \
      _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                \
        sa[k] = ld4(c.Dg + (size_t)((ST) * SGB + sg0 + 2 * k) * dg_stride + a_off);
#define LBS_PRE_LOAD_B(SET, G)                                                                     \
      _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_)                                             \
        _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_)                                           \
          rbg[SET][c_][s_] = *reinterpret_cast<const uint4*>(xb + (size_t)((G) * 2 + c_) * chunk_b + s_ * piece_b);
#define LBS_PRE_STORE_A(BUF)                                                                       \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                              \
        uint2 p0, p1, p2;                                                                          \
        unsigned char* d = sm + (BUF) * BUFA + (sg0 + 2 * k) * 3 * PL + dstb;                      \
        split3x4(sa[k], p0, p1, p2);                                                               \
        *reinterpret_cast<uint2*>(d) = p0; *reinterpret_cast<uint2*>(d + PL) = p1; *reinterpret_cast<uint2*>(d + 2 * PL) = p2; \
      }
      LBS_PRE_LOAD_A(0)
      LBS_PRE_LOAD_B(0, 0)
      LBS_PRE_STORE_A(0)
      LBS_PRE_LOAD_A(1)
      __syncthreads();
      if (DBG) t_pro = __builtin_amdgcn_s_memtime();
      const int a_rdb = (mp * 64 + j) * 16;
#define LBS_PRE_READ(P, C)                                                                         \
        _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) {                                         \
          ra[C][0][s_] = *reinterpret_cast<const uint4*>(base + ((P) * 2 + (C)) * 6 * PL + s_ * PL + a_rdb);   \
          ra[C][1][s_] = *reinterpret_cast<const uint4*>(base + ((P) * 2 + (C)) * 6 * PL + s_ * PL + a_rdb + 32 * 16); \
        }
#define LBS_PRE_MFMA1(P, C, SA, SB)                                                                \
        _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                           \
          acc[m_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[C][m_][SA]), \
                                                            __builtin_bit_cast(bf16x8, rbg[P][C][SB]), acc[m_], 0, 0, 0);
#define LBS_PRE_MFMA(P, C) LBS_PRE_MFMA1(P, C, 0, 2) LBS_PRE_MFMA1(P, C, 2, 0) LBS_PRE_MFMA1(P, C, 1, 1) \
                           LBS_PRE_MFMA1(P, C, 0, 1) LBS_PRE_MFMA1(P, C, 1, 0) LBS_PRE_MFMA1(P, C, 0, 0)
      // half P (literal 0 / 1) of stage ST: 32 features; the global half index 2 ST + P has parity P
#define LBS_PRE_HALF(P, ST) {                                                                      \
        if (2 * (ST) + (P) + 1 < 2 * NSTB) { LBS_PRE_LOAD_B(1 - (P), 2 * (ST) + (P) + 1) }         \
        uint4 ra[2][2][3];                                                                         \
        LBS_PRE_READ(P, 0)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        LBS_PRE_READ(P, 1)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        LBS_PRE_MFMA(P, 0)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        LBS_PRE_MFMA(P, 1)                                                                         \
      }
      for (int st = 0; st < NSTB; ++st) {
        const unsigned char* base = sm + (st & 1) * BUFA + h * 3 * PL;
        LBS_PRE_HALF(0, st)
        __builtin_amdgcn_sched_barrier(0);
        LBS_PRE_HALF(1, st)
        if (st + 1 < NSTB) {
          LBS_PRE_STORE_A((st + 1) & 1)
          if (st + 2 < NSTB) { LBS_PRE_LOAD_A(st + 2) }
        }
        __syncthreads();
      }
#undef LBS_PRE_HALF
#undef LBS_PRE_READ
#undef LBS_PRE_MFMA1
#undef LBS_PRE_MFMA
#undef LBS_PRE_LOAD_A
#undef LBS_PRE_LOAD_B
#undef LBS_PRE_STORE_A
    } else {
    // D streams from HBM (~2 us away under load) and one stage is only ~0.75 us of MFMA work: the global
    // loads run TWO stages ahead (two register sets), the LDS image one stage ahead
    // (LBS_PFA = stages the D loads run ahead: 2 in the product; 3 / 4 are A/B builds, tools/ab_build.sh)
#ifndef LBS_PFA
#define LBS_PFA 2
#endif
    constexpr int PA = LBS_PFA;
    float4 sa[PA][2], sb[2][2];
#define LBS_SPLIT_LOAD_A(SET, ST)                                                                  \
    _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                  \
      sa[SET][k] = ld4(c.Dg + (size_t)((ST) * SG + sg0 + 2 * k) * dg_stride + a_off);
#define LBS_SPLIT_LOAD_B(SET, ST)                                                                  \
    _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                  \
      sb[SET][k] = ld4(Xg + (size_t)((ST) * SG + sg0 + 2 * k) * xg_stride + b_off);
#define LBS_SPLIT_STORE(SETA, SETB, BUF)                                                           \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                \
      uint2 p0, p1, p2;                                                                            \
      unsigned char* d = sm + (BUF) * BUFB + (sg0 + 2 * k) * 3 * PL + dstb;                        \
      split3x4(sa[SETA][k], p0, p1, p2);                                                           \
      *reinterpret_cast<uint2*>(d) = p0; *reinterpret_cast<uint2*>(d + PL) = p1; *reinterpret_cast<uint2*>(d + 2 * PL) = p2; \
      split3x4(sb[SETB][k], p0, p1, p2);                                                           \
      *reinterpret_cast<uint2*>(d + OPB) = p0; *reinterpret_cast<uint2*>(d + OPB + PL) = p1;       \
      *reinterpret_cast<uint2*>(d + OPB + 2 * PL) = p2;                                            \
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) { LBS_SPLIT_LOAD_A(i, i) }
    LBS_SPLIT_LOAD_B(0, 0)
    LBS_SPLIT_LOAD_B(1, 1)
    LBS_SPLIT_STORE(0, 0, 0)
    __syncthreads();
    if (DBG) t_pro = __builtin_amdgcn_s_memtime();
    const int a_rdb = (mp * 64 + j) * 16, b_rdb = OPB + (nt * 32 + j) * 16;
#if LBS_PFA == 2
#pragma unroll 2
#else
#pragma unroll
#endif
    for (int st = 0; st < NSTS; ++st) {
      const int buf = st & 1;
      // register set (st % PA) held stage st (already in LDS): refill it with stage st + PA; same for B with 2 sets
      if (st + PA < NSTS) { LBS_SPLIT_LOAD_A(st % PA, st + PA) }
      if (st + 2 < NSTS) { LBS_SPLIT_LOAD_B(st & 1, st + 2) }
      const unsigned char* base = sm + buf * BUFB + h * 3 * PL;   // lane half h takes group 2c + h of chunk c
      uint4 ra[2][2][3], rb[2][3];                                // [set][m-tile][piece]
#define LBS_SREAD(SET, C)                                                                          \
      _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) {                                           \
        ra[SET][0][s_] = *reinterpret_cast<const uint4*>(base + (C) * 6 * PL + s_ * PL + a_rdb);   \
        ra[SET][1][s_] = *reinterpret_cast<const uint4*>(base + (C) * 6 * PL + s_ * PL + a_rdb + 32 * 16); \
        rb[SET][s_] = *reinterpret_cast<const uint4*>(base + (C) * 6 * PL + s_ * PL + b_rdb);      \
      }
#define LBS_SMFMA1(SET, SA, SB)                                                                    \
      _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                             \
        acc[m_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[SET][m_][SA]), \
                                                          __builtin_bit_cast(bf16x8, rb[SET][SB]), acc[m_], 0, 0, 0);
#define LBS_SMFMA(SET) LBS_SMFMA1(SET, 0, 2) LBS_SMFMA1(SET, 2, 0) LBS_SMFMA1(SET, 1, 1)           \
                       LBS_SMFMA1(SET, 0, 1) LBS_SMFMA1(SET, 1, 0) LBS_SMFMA1(SET, 0, 0)
      LBS_SREAD(0, 0)
      __builtin_amdgcn_sched_barrier(0);
      LBS_SREAD(1, 1)
      __builtin_amdgcn_sched_barrier(0);
      LBS_SMFMA(0)
      __builtin_amdgcn_sched_barrier(0);
      LBS_SMFMA(1)
#undef LBS_SREAD
#undef LBS_SMFMA1
#undef LBS_SMFMA
      // (running the two waves of a SIMD out of phase -- one converting while the other owns the MFMA pipe --
      // measured slower: 49.9k vs 43.1k cycles)
      if (st + 1 < NSTS) { LBS_SPLIT_STORE((st + 1) % PA, (st + 1) & 1, (st + 1) & 1) }
      __syncthreads();
    }
#undef LBS_SPLIT_STORE
#undef LBS_SPLIT_LOAD_A
#undef LBS_SPLIT_LOAD_B
    }
  } else {
  float* As0 = smem;                                   // [2 buffers] of LBS_STAGE_FLOATS
  float* Bs0 = smem + 2 * LBS_STAGE_FLOATS;
  float4 sa[4], sb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sa[k] = ld4(c.Dg + (size_t)(sg0 + 2 * k) * dg_stride + a_off);
    sb[k] = ld4(Xg + (size_t)(sg0 + 2 * k) * xg_stride + b_off);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    st4(As0 + lds_dst + k * (2 * 2 * 128 * 4), sa[k]);
    st4(Bs0 + lds_dst + k * (2 * 2 * 128 * 4), sb[k]);
  }
  __syncthreads();
  if (DBG) t_pro = __builtin_amdgcn_s_memtime();
  const int a_rd = (h * 128 + mp * 64 + j) * 4, b_rd = (h * 128 + nt * 32 + j) * 4;
  constexpr int NST = 64 / LBS_KC;
  for (int st = 0; st < NST; ++st) {
    const int buf = st & 1;
    if (st + 1 < NST) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sa[k] = ld4(c.Dg + (size_t)((st + 1) * LBS_KC + sg0 + 2 * k) * dg_stride + a_off);
        sb[k] = ld4(Xg + (size_t)((st + 1) * LBS_KC + sg0 + 2 * k) * xg_stride + b_off);
      }
    }
    const float* ar = As0 + buf * LBS_STAGE_FLOATS + a_rd;
    const float* br = Bs0 + buf * LBS_STAGE_FLOATS + b_rd;
    float4 ra[2][2][2], rb[2][2];                      // [set][group in chunk][m-tile]
#define LBS_READ2(SET, CH)                                                                        \
    _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) {                                            \
      const int go_ = ((CH) * 2 + g_) * (2 * 128 * 4);                                            \
      ra[SET][g_][0] = ld4(ar + go_); ra[SET][g_][1] = ld4(ar + go_ + 32 * 4);                    \
      rb[SET][g_] = ld4(br + go_);                                                                \
    }
#define LBS_MFMA2(SET)                                                                            \
    _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) {                                            \
      _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_) {                                          \
        acc[m_] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[SET][g_][m_].x, rb[SET][g_].x, acc[m_], 0, 0, 0);  \
        acc[m_] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[SET][g_][m_].y, rb[SET][g_].y, acc[m_], 0, 0, 0);  \
        acc[m_] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[SET][g_][m_].z, rb[SET][g_].z, acc[m_], 0, 0, 0);  \
        acc[m_] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[SET][g_][m_].w, rb[SET][g_].w, acc[m_], 0, 0, 0);  \
      }                                                                                           \
    }
    LBS_READ2(0, 0)
    __builtin_amdgcn_sched_barrier(0);
    LBS_READ2(1, 1)
    __builtin_amdgcn_sched_barrier(0);
    LBS_MFMA2(0)
    __builtin_amdgcn_sched_barrier(0);
    LBS_READ2(0, 2)
    __builtin_amdgcn_sched_barrier(0);
    LBS_MFMA2(1)
    __builtin_amdgcn_sched_barrier(0);
    LBS_READ2(1, 3)
    __builtin_amdgcn_sched_barrier(0);
    LBS_MFMA2(0)
    __builtin_amdgcn_sched_barrier(0);
    LBS_MFMA2(1)
#undef LBS_READ2
#undef LBS_MFMA2
    if (st + 1 < NST) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        st4(As0 + (buf ^ 1) * LBS_STAGE_FLOATS + lds_dst + k * (2 * 2 * 128 * 4), sa[k]);
        st4(Bs0 + (buf ^ 1) * LBS_STAGE_FLOATS + lds_dst + k * (2 * 2 * 128 * 4), sb[k]);
      }
    }
    __syncthreads();
  }
  }
  if (DBG) t_gemm = __builtin_amdgcn_s_memtime();
  // ---- hand the blend tile over through LDS (aliases the staging buffers: everyone is past the last read)
  float* vp = smem;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;             // D: col = frame j, row -> column of the M-tile
      vp[(nt * 32 + j) * LBS_PITCH + (mp * 2 + m) * 32 + row] = acc[m][r];
    }
  __syncthreads();
  // ---- skinning: thread = (vertex lv < 42, frame group of 12)
  const int lv = tid % LBS_VPB, fg = tid / LBS_VPB;
  const int slot = s0 + lv;
  const bool skin = tid < 12 * LBS_VPB && slot < n;
  // (Three chunks in flight instead of one were tried in round 2 and measured SLOWER -- 33.9k vs 28.7k cycles for the
  // phase, tools/lbs_census.py: the phase is bound by LDS issue + the strided global stores, not by the A loads.)
  // The per-joint transforms A[f][nj][12] are gathered (KW joints per vertex) for every (vertex, frame)
  // pair: straight from L2 that is a chain of exposed round trips (measured 51k cycles per block), so
  // they are staged through LDS in chunks of 12 frames (one frame per thread group), double-buffered
  // behind the blend tile, the next chunk loading while the current one is consumed.
  const int fpf = nj * 12;                                        // floats per frame of A
  float* Ab = smem + LBS_FR * LBS_PITCH;                           // [2][12 * fpf]
  const int nfr = (B - f0 < LBS_FR) ? B - f0 : LBS_FR;
  const int nchunk = (nfr + 11) / 12;
  const int n4 = 3 * fpf;                                          // float4 per chunk (12 * fpf / 4)
  const int a_last4 = B * fpf / 4 - 1;                             // clamp: last float4 of A
  float4 sA[5];
#define LBS_ALOAD(CH)                                                                             \
  _Pragma("unroll") for (int k = 0; k < 5; ++k) {                                                 \
    int i4 = (f0 + (CH) * 12) * (fpf / 4) + tid + k * 512;                                        \
    if (i4 > a_last4) i4 = a_last4;                                                               \
    sA[k] = ld4(A + (size_t)i4 * 4);                                                              \
  }
#define LBS_ASTORE(BUF)                                                                           \
  _Pragma("unroll") for (int k = 0; k < 5; ++k) {                                                 \
    const int l4 = tid + k * 512;                                                                 \
    if (l4 < n4) st4(Ab + (BUF) * 12 * fpf + l4 * 4, sA[k]);                                      \
  }
  int vid = 0;
  float tx = 0.f, ty = 0.f, tz = 0.f;
  if (skin) {
    vid = ids ? ids[slot] : slot;
    tx = c.v_template[(size_t)vid * 3]; ty = c.v_template[(size_t)vid * 3 + 1]; tz = c.v_template[(size_t)vid * 3 + 2];
  }
  const int* wi = c.w_idx + (size_t)vid * c.KW;
  const float* wv = c.w_val + (size_t)vid * c.KW;
  // the vertex's (joint, weight) pairs are read ONCE, up front and all together (they were re-read from global memory
  // inside every 12-frame chunk: KW dependent L1 round trips x 11 chunks per thread); rows beyond KW carry weight 0
  constexpr int KWR = 8;
  int jk[KWR];
  float wk[KWR];
#pragma unroll
  for (int k = 0; k < KWR; ++k) {
    const int kk = k < c.KW ? k : c.KW - 1;
    jk[k] = wi[kk] * 12;
    wk[k] = k < c.KW ? wv[kk] : 0.f;
  }
  const int kwr = c.KW < KWR ? c.KW : KWR;
  LBS_ALOAD(0)
  LBS_ASTORE(0)
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunk) { LBS_ALOAD(ch + 1) }
    const int fl = ch * 12 + fg, f = f0 + fl;
    if (skin && fl < nfr) {
      const float px = vp[fl * LBS_PITCH + 3 * lv] + tx, py = vp[fl * LBS_PITCH + 3 * lv + 1] + ty,
                  pz = vp[fl * LBS_PITCH + 3 * lv + 2] + tz;
      float T[12];
#pragma unroll
      for (int e = 0; e < 12; ++e) T[e] = 0.f;
      const float* Af = Ab + buf * 12 * fpf + fg * fpf;
#define LBS_SKIN_K(W_, AJ_) {                                                                     \
        const float w = (W_);                                                                     \
        const float* Aj = (AJ_);                                                                  \
        const float4 r0 = ld4(Aj), r1 = ld4(Aj + 4), r2 = ld4(Aj + 8);                            \
        T[0] = fmaf(w, r0.x, T[0]); T[1] = fmaf(w, r0.y, T[1]); T[2] = fmaf(w, r0.z, T[2]); T[3] = fmaf(w, r0.w, T[3]);   \
        T[4] = fmaf(w, r1.x, T[4]); T[5] = fmaf(w, r1.y, T[5]); T[6] = fmaf(w, r1.z, T[6]); T[7] = fmaf(w, r1.w, T[7]);   \
        T[8] = fmaf(w, r2.x, T[8]); T[9] = fmaf(w, r2.y, T[9]); T[10] = fmaf(w, r2.z, T[10]); T[11] = fmaf(w, r2.w, T[11]); }
#pragma unroll
      for (int k = 0; k < KWR; ++k)
        if (k < kwr) LBS_SKIN_K(wk[k], Af + jk[k])
      for (int k = KWR; k < c.KW; ++k) LBS_SKIN_K(wv[k], Af + wi[k] * 12)     // models with more than 8 weights per vertex
#undef LBS_SKIN_K
      float ox = T[0] * px + T[1] * py + T[2] * pz + T[3];
      float oy = T[4] * px + T[5] * py + T[6] * pz + T[7];
      float oz = T[8] * px + T[9] * py + T[10] * pz + T[11];
      if (transl) { ox += transl[(size_t)f * 3]; oy += transl[(size_t)f * 3 + 1]; oz += transl[(size_t)f * 3 + 2]; }
      // one 12-byte store per output row (global_store_dwordx3) instead of three dword stores with a 12-byte lane stride
      *reinterpret_cast<float3*>(verts + ((size_t)f * n + slot) * 3) = make_float3(ox, oy, oz);
      if (v_posed) *reinterpret_cast<float3*>(v_posed + ((size_t)f * n + slot) * 3) = make_float3(px, py, pz);
    }
    if (ch + 1 < nchunk) { LBS_ASTORE(buf ^ 1) }
    __syncthreads();
  }
#undef LBS_ALOAD
#undef LBS_ASTORE
  } while (LOOP && (tile_x += (int)gridDim.x) < ntile_x);      // (the body's last statement is a workgroup barrier: the next tile may overwrite LDS)
  if (DBG && lane == 0) {
    unsigned long long* r = dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 4;
    r[0] = t_start; r[1] = t_pro; r[2] = t_gemm; r[3] = __builtin_amdgcn_s_memtime();
  }
}

int lbs_init() {
  static int rc = -1;
  if (rc >= 0) return rc;
  rc = 0;
#define OPTIN(DBG_, SPLIT_, PRE_) if (!rc) rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&lbs_verts_fwd_kernel<DBG_, SPLIT_, PRE_>), hipFuncAttributeMaxDynamicSharedMemorySize, LBS_SMEM_MAX);
  OPTIN(false, false, false) OPTIN(true, false, false) OPTIN(false, true, false) OPTIN(true, true, false) OPTIN(false, true, true) OPTIN(true, true, true)
#undef OPTIN
#define OPTINH(DBG_) if (!rc) rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&lbs_verts_fwd_kernel<DBG_, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LBS_SMEM_MAX);
  OPTINH(false) OPTINH(true)
#undef OPTINH
  if (!rc) rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&lbs_verts_fwd_kernel<false, true, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LBS_SMEM_MAX);
  return rc;
}

int lbs_verts_fwd(const SkinConst& c, const float* Xg, int Bp, const float* A, int nj, const float* transl,
                  const int* ids, int n, int B, float* verts, float* v_posed, hipStream_t s, unsigned long long* dbg,
                  const unsigned short* XgS, int max_blocks_x) {
  if (n <= 0 || B <= 0 || B > Bp || (Bp % 32) || (!ids && n != c.V)) return LEMO_ERR_SHAPE;
  if (lbs_init()) return LEMO_ERR_STATE;
  if (nj > 64 || 3 * nj * 12 > 5 * 512) return LEMO_ERR_SHAPE;
  const int smem_bytes = lbs_smem_bytes(nj);
  dim3 grid((n + LBS_VPB - 1) / LBS_VPB, (B + LBS_FR - 1) / LBS_FR);
  if (max_blocks_x > 0 && (int)grid.x > max_blocks_x) {      // fewer workgroups, each looping over its tiles (same arithmetic per tile: same bits)
    if (c.blend_fp32 || !XgS || !c.DgH || dbg) return LEMO_ERR_ARG;      // only the shipped arithmetic (pre-split fp16 operands) has the looping instantiation
    grid.x = max_blocks_x;
    hipLaunchKernelGGL((lbs_verts_fwd_kernel<false, true, true, true, true>), grid, dim3(512), smem_bytes, s, c, Xg, Bp, A, nj, transl, ids, n, B, verts, v_posed, dbg, XgS);
    return (int)hipGetLastError();
  }
#define LAUNCH(DBG_, SPLIT_, PRE_) hipLaunchKernelGGL((lbs_verts_fwd_kernel<DBG_, SPLIT_, PRE_>), grid, dim3(512), smem_bytes, s, c, Xg, Bp, A, nj, transl, ids, n, B, verts, v_posed, dbg, XgS)
#define LAUNCHH(DBG_) hipLaunchKernelGGL((lbs_verts_fwd_kernel<DBG_, true, true, true>), grid, dim3(512), smem_bytes, s, c, Xg, Bp, A, nj, transl, ids, n, B, verts, v_posed, dbg, XgS)
  if (!c.blend_fp32 && XgS && c.DgH) { if (dbg) LAUNCHH(true); else LAUNCHH(false); }      // (XgS in its fp16 form: lemo_pose_ws.xgs_f16)
  else if (!c.blend_fp32 && XgS) { if (dbg) LAUNCH(true, true, true); else LAUNCH(false, true, true); }
  else if (!c.blend_fp32) { if (dbg) LAUNCH(true, true, false); else LAUNCH(false, true, false); }
  else { if (dbg) LAUNCH(true, false, false); else LAUNCH(false, false, false); }
#undef LAUNCH
#undef LAUNCHH
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Forward restricted to a small vertex set U (the AMASS losses touch 253 of 10475 vertices).  The tiled
// kernel above is built for all vertices (42-vertex tiles x the full K on one CU each: 7 busy CUs and ~37 us for
// n = 253); here the blend is a small MFMA GEMM over U's compact transposed directions DkT [3n -> NCs][512]
// (384 workgroups, split-K inside each) followed by a thread-per-(frame, vertex) skinning pass.
//   blend [B][NCs] scratch (the engine passes its d(v_posed) buffer, free during the forward pass)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
lbs_skin_active_kernel(SkinConst c, VertexSetBwd u, const float* __restrict__ blend, const float* __restrict__ A, int nj,
                       const float* __restrict__ transl, int B, float* __restrict__ verts, float* __restrict__ v_posed,
                       float* __restrict__ transl_copy) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * u.n) return;
  const int b = idx / u.n, s = idx - b * u.n;
  if (transl_copy && transl && s < 3) transl_copy[b * 3 + s] = transl[b * 3 + s];      // the forward's translation, for the fit engine's side-branch launch
  const int vid = u.ids[s];
  const float* bl = blend + (size_t)b * u.NCs + 3 * s;
  const float px = bl[0] + c.v_template[(size_t)vid * 3], py = bl[1] + c.v_template[(size_t)vid * 3 + 1],
              pz = bl[2] + c.v_template[(size_t)vid * 3 + 2];
  const int* wi = c.w_idx + (size_t)vid * c.KW;
  const float* wv = c.w_val + (size_t)vid * c.KW;
  const float* Af = A + (size_t)b * nj * 12;
  float T[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) T[e] = 0.f;
  for (int k0 = 0; k0 < c.KW; k0 += 4) {                       // 4 (joint, weight) pairs per round trip
    int ji[4]; float wk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int kk = k0 + k < c.KW ? k0 + k : c.KW - 1;
      ji[k] = wi[kk];
      wk[k] = k0 + k < c.KW ? wv[kk] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float w = wk[k];
      const float* Aj = Af + ji[k] * 12;
      const float4 r0 = ld4(Aj), r1 = ld4(Aj + 4), r2 = ld4(Aj + 8);
      T[0] = fmaf(w, r0.x, T[0]); T[1] = fmaf(w, r0.y, T[1]); T[2] = fmaf(w, r0.z, T[2]); T[3] = fmaf(w, r0.w, T[3]);
      T[4] = fmaf(w, r1.x, T[4]); T[5] = fmaf(w, r1.y, T[5]); T[6] = fmaf(w, r1.z, T[6]); T[7] = fmaf(w, r1.w, T[7]);
      T[8] = fmaf(w, r2.x, T[8]); T[9] = fmaf(w, r2.y, T[9]); T[10] = fmaf(w, r2.z, T[10]); T[11] = fmaf(w, r2.w, T[11]);
    }
  }
  float ox = T[0] * px + T[1] * py + T[2] * pz + T[3];
  float oy = T[4] * px + T[5] * py + T[6] * pz + T[7];
  float oz = T[8] * px + T[9] * py + T[10] * pz + T[11];
  if (transl) { ox += transl[(size_t)b * 3]; oy += transl[(size_t)b * 3 + 1]; oz += transl[(size_t)b * 3 + 2]; }
  float* o = verts + ((size_t)b * u.n + s) * 3;
  o[0] = ox; o[1] = oy; o[2] = oz;
  if (v_posed) { float* q = v_posed + ((size_t)b * u.n + s) * 3; q[0] = px; q[1] = py; q[2] = pz; }
}

int lbs_verts_fwd_active(const SkinConst& c, const VertexSetBwd& u, const float* Xg, int Bp, const float* A, int nj,
                         const float* transl, int B, float* blend, float* verts, float* v_posed, hipStream_t s, float* transl_copy) {
  if (u.n <= 0 || B <= 0 || B > Bp || (u.NCs % 16) || u.NCs < 3 * u.n || !u.DkT || !blend || (transl_copy && u.n < 3)) return LEMO_ERR_SHAPE;
  if (int e = gemm_nt16_kg8(u.DkT, 512, Xg, Bp, u.NCs, B, 512, blend, u.NCs, s)) return e;
  hipLaunchKernelGGL(lbs_skin_active_kernel, dim3((B * u.n + 255) / 256), dim3(256), 0, s, c, u, blend, A, nj, transl, B, verts, v_posed, transl_copy);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Backward over a vertex set U (all vertices, or only those that carry gradient: the AMASS losses
// touch 81 markers + 172 heel/toe vertices, opt_amass_temp.py:359,366,414-425 -- every other row of
// d(verts) is exactly zero, so restricting the backward to U is exact).
//   kernel 1 (block per frame): dvp = T_R^T g ; dA[j] = sum_u W[u,j] g (x) [v_posed,1] ; dtransl = sum_u g
//   kernel 2: dX[b][k] = sum_col Dk[k][col] dvp[b][col]        (fp32-MFMA NT GEMM)
// ------------------------------------------------------------------------------------------------
#define LBS_BWD_STAGE 1024          // vertex sets up to this size keep g / v_posed of the frame in LDS
#define LBS_BWD_NNZ 4096            // ... and the joint-major CSR (vertex, weight) lists up to this many entries
// The staged variant runs 1024 threads per frame: every phase is a short dependent chain (global -> LDS -> LDS), and
// with one wave per SIMD each of its ~100 VALU instructions and each LDS round trip is fully exposed; four waves per
// SIMD overlap them (the joint loop below becomes one batch per wave).
// FUSED (fitting engine): d(verts) of the frame is computed here (loss_device.hpp::dverts_vertex) by the first four
// waves, straight into the LDS copy the other phases read, while the other twelve stage the frame -- the separate
// dverts_assemble launch (one cold dependent chain + a dispatch, ~9 us) and the global round trip of d(verts) go.
template <bool STAGE, bool FUSED>
__global__ void __launch_bounds__(STAGE ? 1024 : 256)
lbs_bwd_frame_kernel(SkinConst c, VertexSetBwd u, const float* __restrict__ A, int nj,
                     const float* __restrict__ v_posed, int vp_rows, const float* __restrict__ dverts,
                     float* __restrict__ dvp, float* __restrict__ dA, float* __restrict__ dtransl, FitFuse ff) {
  static_assert(STAGE || !FUSED, "the fused variant stages the frame in LDS");
  CENSUS_DECL(2)
  CENSUS()
  __shared__ float gs[STAGE ? LBS_BWD_STAGE * 3 : 1];
  __shared__ float vs[STAGE ? LBS_BWD_STAGE * 3 : 1];
  __shared__ float As[STAGE ? 64 * 12 : 1];
  __shared__ int cu[STAGE ? LBS_BWD_NNZ : 1];
  __shared__ float cw[STAGE ? LBS_BWD_NNZ : 1];
  __shared__ int js[STAGE ? 65 : 1];   // jcsr_start (nj <= 64: the host checks)
  constexpr int NT = STAGE ? 1024 : 256, NW = NT / 64;
  const int b = blockIdx.x, t = threadIdx.x;
  const float* Af = A + (size_t)b * nj * 12;
  const float* g = FUSED ? nullptr : dverts + (size_t)b * u.n * 3;
  int vid0 = 0, ji0[4] = {0, 0, 0, 0};
  float wk0[4] = {0.f, 0.f, 0.f, 0.f};
  if (STAGE) {                        // coalesced / gathered once, then every inner loop reads LDS
    // loads are issued in batches of 4 per thread with clamped (never predicated) addresses so that they are all in
    // flight together; with one load -> one LDS store per loop trip the prologue was a chain of ~20 L2 round trips
    constexpr int DV = FUSED ? 256 : 0, NS = NT - DV;    // threads [0, DV): loss record + d(verts); the rest stage
    __shared__ float losses[FUSED ? 12 : 1];
    __shared__ double tots[FUSED ? 13 : 1];
    const int ts = t - DV;
    DvIdx ix0 = {0, -1, 0, -1};
    DvRegs dv0;
    const int n3 = u.n * 3, na = nj * 12, nnz = u.jcsr_start[nj];
    vid0 = u.ids[min(t, u.n - 1)];   // first vertex of the dvp loop below: its (index, weight) reads ride along
    if (!FUSED || t >= DV) {
      for (int i0 = 0; i0 < n3; i0 += 4 * NS) {
        float a[4], v[4]; int row[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = min(i0 + ts + NS * k, n3 - 1);
          a[k] = FUSED ? 0.f : g[i];
          row[k] = u.vp_row[i / 3];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = min(i0 + ts + NS * k, n3 - 1);
          v[k] = v_posed[((size_t)b * vp_rows + row[k]) * 3 + (i % 3)];
        }
        if (i0 == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int kk = min(k, c.KW - 1);
            ji0[k] = c.w_idx[(size_t)vid0 * c.KW + kk];
            wk0[k] = k < c.KW ? c.w_val[(size_t)vid0 * c.KW + kk] : 0.f;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + ts + NS * k;
          if (i < n3) { if (!FUSED) gs[i] = a[k]; vs[i] = v[k]; }
        }
      }
      for (int i0 = 0; i0 < nnz; i0 += 4 * NS) {
        int cu4[4]; float cw4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = min(i0 + ts + NS * k, nnz - 1);
          cu4[k] = u.jcsr_u[i];
          cw4[k] = u.jcsr_w[i];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + ts + NS * k;
          if (i < nnz) { cu[i] = cu4[k]; cw[i] = cw4[k]; }
        }
      }
      for (int i = ts; i < na; i += NS) As[i] = Af[i];
      if (ts <= nj) js[ts] = u.jcsr_start[ts];
    } else {
      // loss record: 13 lanes of wave 0 total the accumulator slots, lane 0 finalises (same wave: program order is
      // enough between the LDS write and the read); every block does it, block 0 publishes
      ix0 = dverts_indices(ff.fc, min(t, u.n - 1));       // indices of this lane's first vertex: in flight with the rest
      // loss record: d(verts) only needs 1 / count of the four foot sets (accumulators 5..8): lanes 5..8 total their
      // slots and divide, in parallel.  The full record (a dozen f64 divisions on one lane, ~4 k cycles) is only for
      // reporting: block 0 totals all 13 and an otherwise idle staging wave finalises it after the barrier.
      if (t < 13 && (b == 0 || (t >= 5 && t < 9))) {
        const double tot = loss_slot_total(ff.acc, t);
        tots[t] = tot;
        if (t >= 5 && t < 9) losses[8 + (t - 5)] = tot >= 1.0 ? (float)(1.0 / tot) : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int kk = min(k, c.KW - 1);
        ji0[k] = c.w_idx[(size_t)vid0 * c.KW + kk];
        wk0[k] = k < c.KW ? c.w_val[(size_t)vid0 * c.KW + kk] : 0.f;
      }
      dverts_load(ff.fc, ff.in, b, ix0, dv0);               // second-level reads of the first vertex, before the barrier
    }
    __syncthreads();
    if (FUSED) {
      if (b == 0 && t == DV) {
        float rec[12];
        finalize_losses(tots, ff.in.B, ff.fc.n67, ff.smooth_count, ff.in.weights, rec);
        for (int i = 0; i < 12; ++i) ff.losses_out[i] = rec[i];
      }
      if (t < DV)
        for (int uu = t; uu < u.n; uu += DV) {
          float gx, gy, gz;
          if (uu == t) dverts_compute(ff.fc, ff.in, losses, b, dv0, gx, gy, gz);
          else dverts_vertex(ff.fc, ff.in, losses, b, dverts_indices(ff.fc, uu), gx, gy, gz);
          gs[3 * uu] = gx; gs[3 * uu + 1] = gy; gs[3 * uu + 2] = gz;
        }
      __syncthreads();
    }
    CENSUS()
  }
  const float* gp = STAGE ? gs : g;
  const float* Ap = STAGE ? As : Af;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int s = t; s < u.n; s += NT) {
    const bool pre = STAGE && s == t;                    // block-uniform: the first trip uses the prefetched reads
    const int vid = pre ? vid0 : u.ids[s];
    const float gx = gp[3 * s], gy = gp[3 * s + 1], gz = gp[3 * s + 2];
    sx += gx; sy += gy; sz += gz;
    float T[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) T[e] = 0.f;
    const int* wi = c.w_idx + (size_t)vid * c.KW;
    const float* wv = c.w_val + (size_t)vid * c.KW;
    for (int k0 = 0; k0 < c.KW; k0 += 4) {               // 4 (index, weight) pairs per round trip, not one
      int ji[4]; float wk[4];
      if (pre && k0 == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { ji[k] = ji0[k]; wk[k] = wk0[k]; }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int kk = k0 + k < c.KW ? k0 + k : c.KW - 1;
          ji[k] = wi[kk];
          wk[k] = k0 + k < c.KW ? wv[kk] : 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float w = wk[k];
        const float* Aj = Ap + ji[k] * 12;
        T[0] = fmaf(w, Aj[0], T[0]); T[1] = fmaf(w, Aj[1], T[1]); T[2] = fmaf(w, Aj[2], T[2]);
        T[3] = fmaf(w, Aj[4], T[3]); T[4] = fmaf(w, Aj[5], T[4]); T[5] = fmaf(w, Aj[6], T[5]);
        T[6] = fmaf(w, Aj[8], T[6]); T[7] = fmaf(w, Aj[9], T[7]); T[8] = fmaf(w, Aj[10], T[8]);
      }
    }
    float* d = dvp + (size_t)b * u.NCs + 3 * s;
    d[0] = T[0] * gx + T[3] * gy + T[6] * gz;
    d[1] = T[1] * gx + T[4] * gy + T[7] * gz;
    d[2] = T[2] * gx + T[5] * gy + T[8] * gz;
  }
  CENSUS()
  // zero the padding columns of this frame's dvp row
  for (int cidx = 3 * u.n + t; cidx < u.NCs; cidx += NT) dvp[(size_t)b * u.NCs + cidx] = 0.f;
  // dA via the joint-major CSR (deterministic gather)
  if (STAGE) {
    // the lists are very uneven (the heel / toe vertices hang ~170 entries on each foot joint, most joints have a
    // handful): one (joint, element) list per thread left 250 threads waiting for the 4 longest walks (23 k cycles).
    // Here a wave takes a joint; lane = 16 * row + segment walks every 16th entry for one row of dA (4 outputs),
    // and the 16 segments are combined with a fixed DPP tree -> deterministic, ~16x shorter critical path.
    // Four joints of the wave are in flight at a time: one joint after
    // the other is a chain of ~4 dependent LDS round trips + the reduction per joint, 14 times per wave.
    const int wave = t >> 6, lane = t & 63, seg = lane & 15, r = lane >> 4, rr = min(r, 2);
    const int qmax = max(js[nj] - 1, 0);
    // list bounds of the wave's joints wave, wave + NW, ...: lane L holds joint wave + NW L, handed out by v_readlane
    // (wave-uniform q0 / q1 in scalar registers; one LDS round trip instead of one per joint)
    const int jl = min(wave + NW * (lane & 15), nj - 1);
    const int js_lo = js[jl], js_hi = wave + NW * (lane & 15) < nj ? js[jl + 1] : js_lo;
    const bool b0 = seg & 1, b1 = seg & 2, b2 = seg & 4, b3 = seg & 8;
    for (int j0 = wave, m0 = 0; j0 < nj; j0 += 4 * NW, m0 += 4) {
      int q0[4], q1[4], sv[4];
      float wq[4], acc[16];             // acc[4 k + e]: joint j0 + NW k, element e of row r
      int len = 0;                      // wave-uniform: the longest of the four lists
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        q0[k] = __builtin_amdgcn_readlane(js_lo, m0 + k);
        q1[k] = __builtin_amdgcn_readlane(js_hi, m0 + k);
        len = max(len, q1[k] - q0[k]);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      for (int base = 0; base < len; base += 16) {      // every trip: 4 joints x (index, weight) -> (g, v) reads in flight
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int q = q0[k] + base + seg, qc = min(q, qmax);
          sv[k] = cu[qc];
          wq[k] = q < q1[k] ? cw[qc] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float gv = gs[3 * sv[k] + rr] * wq[k];
          acc[4 * k] = fmaf(gv, vs[3 * sv[k]], acc[4 * k]); acc[4 * k + 1] = fmaf(gv, vs[3 * sv[k] + 1], acc[4 * k + 1]);
          acc[4 * k + 2] = fmaf(gv, vs[3 * sv[k] + 2], acc[4 * k + 2]);
          acc[4 * k + 3] += gv;
        }
      }
      // 16 values x 16 segments -> lane seg ends with the row sum of value seg: each butterfly level keeps the half of
      // the values selected by one bit of the lane index (17 DPP moves instead of 64 for 16 separate row sums)
      float w8[8], x4[4], y2[2];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const float keep = b0 ? acc[2 * m + 1] : acc[2 * m], send = b0 ? acc[2 * m] : acc[2 * m + 1];
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
      const float tot = keep + dpp_move<0x128>(send);                     // row_ror 8: lane ^ 8
      const int jj = j0 + NW * (seg >> 2);
      if (r < 3 && jj < nj) dA[((size_t)b * nj + jj) * 12 + 4 * r + (seg & 3)] = tot;
    }
  } else {
    for (int w = t; w < nj * 12; w += 256) {
      const int jj = w / 12, e = w % 12, r = e >> 2, cc = e & 3;
      float acc = 0.f;
      const int q0 = u.jcsr_start[jj], q1 = u.jcsr_start[jj + 1];
      for (int q = q0; q < q1; ++q) {
        const int sv = u.jcsr_u[q];
        const float gv = gp[3 * sv + r] * u.jcsr_w[q];
        if (cc < 3) acc += gv * v_posed[((size_t)b * vp_rows + u.vp_row[sv]) * 3 + cc];
        else acc += gv;
      }
      dA[((size_t)b * nj) * 12 + w] = acc;
    }
  }
  CENSUS()
  if (dtransl) {                      // one barrier pair for the three sums (fixed order: wave tree, then the waves in order)
    __shared__ float red3[3 * NW];
    sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
    if ((t & 63) == 0) { red3[3 * (t >> 6)] = sx; red3[3 * (t >> 6) + 1] = sy; red3[3 * (t >> 6) + 2] = sz; }
    __syncthreads();
    if (t < 3) {
      float a = red3[t];
      for (int w = 1; w < NW; ++w) a += red3[3 * w + t];
      dtransl[(size_t)b * 3 + t] = a;
    }
  }
  CENSUS()
}

// ---- dense variant (large vertex sets, e.g. the PROX window whose scene terms touch all 10475 vertices) ------------
// lbs_bwd_frame_kernel walks each frame with ONE block: 41 vertices per thread and joint lists of ~760 entries per
// thread -> 1.3 ms per launch at n = 10475.  Here a block takes a chunk of 512 vertices of one frame: vertex-major,
// dA accumulated in LDS with float atomics and added to global dA with one atomic per entry (order of the adds is
// not fixed: results vary in the last bits between runs, like the torch scatter ops of the same path).
#define LBS_DENSE_CHUNK 512
__global__ void __launch_bounds__(256)
lbs_bwd_zero_kernel(float* __restrict__ dvp, int NCs, int n3, float* __restrict__ dA, int na, float* __restrict__ dtransl) {
  const int b = blockIdx.x, t = threadIdx.x;
  for (int i = t; i < na; i += 256) dA[(size_t)b * na + i] = 0.f;
  for (int i = n3 + t; i < NCs; i += 256) dvp[(size_t)b * NCs + i] = 0.f;       // padding columns of the GEMM operand
  if (dtransl && t < 3) dtransl[(size_t)b * 3 + t] = 0.f;
}
__global__ void __launch_bounds__(256)
lbs_bwd_dense_kernel(SkinConst c, VertexSetBwd u, const float* __restrict__ A, int nj, const float* __restrict__ v_posed,
                     int vp_rows, const float* __restrict__ dverts, float* __restrict__ dvp, float* __restrict__ dA,
                     float* __restrict__ dtransl) {
  __shared__ float As[64 * 12];
  __shared__ float dAs[64 * 12];
  __shared__ float red[4];
  const int b = blockIdx.y, t = threadIdx.x;
  const float* Af = A + (size_t)b * nj * 12;
  for (int i = t; i < nj * 12; i += 256) { As[i] = Af[i]; dAs[i] = 0.f; }
  __syncthreads();
  float sx = 0.f, sy = 0.f, sz = 0.f;
  const int s_end = min((int)(blockIdx.x + 1) * LBS_DENSE_CHUNK, u.n);
  for (int s = blockIdx.x * LBS_DENSE_CHUNK + t; s < s_end; s += 256) {
    const int vid = u.ids[s];
    const float* g = dverts + ((size_t)b * u.n + s) * 3;
    const float* vp = v_posed + ((size_t)b * vp_rows + u.vp_row[s]) * 3;
    const float gx = g[0], gy = g[1], gz = g[2];
    const float vx = vp[0], vy = vp[1], vz = vp[2];
    sx += gx; sy += gy; sz += gz;
    float T[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) T[e] = 0.f;
    const int* wi = c.w_idx + (size_t)vid * c.KW;
    const float* wv = c.w_val + (size_t)vid * c.KW;
    for (int k = 0; k < c.KW; ++k) {
      const float w = wv[k];
      if (w == 0.f) continue;                                  // ELL padding
      const int j = wi[k];
      const float* Aj = As + j * 12;
      T[0] = fmaf(w, Aj[0], T[0]); T[1] = fmaf(w, Aj[1], T[1]); T[2] = fmaf(w, Aj[2], T[2]);
      T[3] = fmaf(w, Aj[4], T[3]); T[4] = fmaf(w, Aj[5], T[4]); T[5] = fmaf(w, Aj[6], T[5]);
      T[6] = fmaf(w, Aj[8], T[6]); T[7] = fmaf(w, Aj[9], T[7]); T[8] = fmaf(w, Aj[10], T[8]);
      float* d = dAs + j * 12;
      const float wx = w * gx, wy = w * gy, wz = w * gz;       // dA[j][r][:] += w g_r (x) [v, 1]
      atomicAdd(d + 0, wx * vx); atomicAdd(d + 1, wx * vy); atomicAdd(d + 2, wx * vz); atomicAdd(d + 3, wx);
      atomicAdd(d + 4, wy * vx); atomicAdd(d + 5, wy * vy); atomicAdd(d + 6, wy * vz); atomicAdd(d + 7, wy);
      atomicAdd(d + 8, wz * vx); atomicAdd(d + 9, wz * vy); atomicAdd(d + 10, wz * vz); atomicAdd(d + 11, wz);
    }
    float* d = dvp + (size_t)b * u.NCs + 3 * s;
    d[0] = T[0] * gx + T[3] * gy + T[6] * gz;
    d[1] = T[1] * gx + T[4] * gy + T[7] * gz;
    d[2] = T[2] * gx + T[5] * gy + T[8] * gz;
  }
  __syncthreads();
  for (int i = t; i < nj * 12; i += 256) {
    const float v = dAs[i];
    if (v != 0.f) atomicAdd(dA + (size_t)b * nj * 12 + i, v);
  }
  if (dtransl) {
    const float tx = block_sum(sx, red), ty = block_sum(sy, red), tz = block_sum(sz, red);
    if (t == 0) { atomicAdd(dtransl + (size_t)b * 3, tx); atomicAdd(dtransl + (size_t)b * 3 + 1, ty); atomicAdd(dtransl + (size_t)b * 3 + 2, tz); }
  }
}

// ---- dense variant, deterministic (round 2): the same (frame, 512-vertex chunk) decomposition, but dA is gathered
// joint-major inside the chunk (the joint lists are sorted by set position: chunk c of joint j is a contiguous run,
// u.jcsr_chunk) by one wave per joint with a fixed reduction tree, written as a per-chunk partial, and the partials are
// added in chunk order by lbs_bwd_reduce_kernel.  No atomics anywhere: two runs give identical bits.
#define LBS_PART_STRIDE(nj) ((nj) * 12 + 4)
#define LBS_CHUNK_NNZ 2560       // staged (position, weight) pairs per chunk: 512 vertices x <= 5 weights (20 KB); above: global reads
__global__ void __launch_bounds__(256)
lbs_bwd_chunk_kernel(SkinConst c, VertexSetBwd u, const float* __restrict__ A, int nj, const float* __restrict__ v_posed,
                     int vp_rows, const float* __restrict__ dverts, float* __restrict__ dvp) {
  __shared__ __attribute__((aligned(16))) float As[64 * 12];
  __shared__ float gs[LBS_DENSE_CHUNK * 3], vs[LBS_DENSE_CHUNK * 3];
  __shared__ float red[3 * 4];
  const int b = blockIdx.y, ch = blockIdx.x, t = threadIdx.x;
  const int nchunk = gridDim.x;
  const int s0 = ch * LBS_DENSE_CHUNK, s_end = min(s0 + LBS_DENSE_CHUNK, u.n), cn = s_end - s0;
  const float* Af = A + (size_t)b * nj * 12;
  // the all-vertex set is the identity (ids[s] == vp_row[s] == s): its staging reads are then plain contiguous runs
  // instead of index -> row chains; every read of the kernel is issued here, before the barrier (the first version
  // chased ids -> weights -> LDS per vertex after it: 51 us for a kernel that moves 38 MB)
  const bool ident = u.n == c.V && vp_rows == c.V;
  // this chunk's joint lists: offsets per joint + the (position, weight) pairs themselves, one contiguous run of the
  // chunk-major arrays, staged before the barrier (read per joint from global memory in the wave loops below they were a
  // dependent L2 round trip per joint: with weights spread over many joints, ~14 per wave)
  __shared__ int tabs[65];
  __shared__ int cus[LBS_CHUNK_NNZ];
  __shared__ float cws[LBS_CHUNK_NNZ];
  const int* tabg = u.jcsr_chunk + (size_t)ch * (nj + 1);
  if (t <= nj) tabs[t] = tabg[t];
  const int e0 = tabg[0], e1 = tabg[nj];
  const bool staged = e1 - e0 <= LBS_CHUNK_NNZ;
  if (staged) {                                               // same rule: every read in flight before the first store
    int ru[LBS_CHUNK_NNZ / 256]; float rw[LBS_CHUNK_NNZ / 256];
    const int ne = e1 - e0;
#pragma unroll
    for (int k = 0; k < LBS_CHUNK_NNZ / 256; ++k) {
      const int i = e0 + min(t + 256 * k, max(ne - 1, 0));
      ru[k] = u.jc_u[i]; rw[k] = u.jc_w[i];
    }
#pragma unroll
    for (int k = 0; k < LBS_CHUNK_NNZ / 256; ++k) if (t + 256 * k < ne) { cus[t + 256 * k] = ru[k]; cws[t + 256 * k] = rw[k]; }
  }
  // all 12 + 3 staging reads of a thread are issued before the first LDS store (a load inside a rolled loop is waited
  // for in the same trip: six dependent round trips)
  float rg[6], rv[6], ra[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) ra[k] = Af[min(t + 256 * k, nj * 12 - 1)];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int i = min(t + 256 * k, cn * 3 - 1);
    const int s = s0 + i / 3, e = i % 3;
    rg[k] = dverts[((size_t)b * u.n + s) * 3 + e];
    rv[k] = v_posed[((size_t)b * vp_rows + (ident ? s : u.vp_row[s])) * 3 + e];
  }
  constexpr int KWF = 4;                                     // skinning weights of a vertex read together (real SMPL-X: <= 4)
  int wj[2][KWF]; float wk[2][KWF]; int vids[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int l = min(t + 256 * r, cn - 1);
    vids[r] = ident ? s0 + l : u.ids[s0 + l];
  }
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int k = 0; k < KWF; ++k) {
      const int kk = k < c.KW ? k : c.KW - 1;
      wj[r][k] = c.w_idx[(size_t)vids[r] * c.KW + kk] * 12;
      wk[r][k] = k < c.KW ? c.w_val[(size_t)vids[r] * c.KW + kk] : 0.f;
    }
#pragma unroll
  for (int k = 0; k < 3; ++k) if (t + 256 * k < nj * 12) As[t + 256 * k] = ra[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) if (t + 256 * k < cn * 3) { gs[t + 256 * k] = rg[k]; vs[t + 256 * k] = rv[k]; }
  __syncthreads();
  // ---- vertex-major: d(v_posed) = T^T g
  float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int l = t + 256 * r;
    if (l >= cn) continue;
    const int vid = vids[r];
    const float gx = gs[3 * l], gy = gs[3 * l + 1], gz = gs[3 * l + 2];
    sx += gx; sy += gy; sz += gz;
    float T[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) T[e] = 0.f;
    const int* wi = c.w_idx + (size_t)vid * c.KW;
    const float* wv = c.w_val + (size_t)vid * c.KW;
#pragma unroll
    for (int k = 0; k < KWF; ++k) {
      const float w = wk[r][k];
      const float* Aj = As + wj[r][k];
      const float4 a0 = ld4(Aj), a1 = ld4(Aj + 4), a2 = ld4(Aj + 8);   // three 16-byte LDS reads per joint (48-byte rows)
      T[0] = fmaf(w, a0.x, T[0]); T[1] = fmaf(w, a0.y, T[1]); T[2] = fmaf(w, a0.z, T[2]);
      T[3] = fmaf(w, a1.x, T[3]); T[4] = fmaf(w, a1.y, T[4]); T[5] = fmaf(w, a1.z, T[5]);
      T[6] = fmaf(w, a2.x, T[6]); T[7] = fmaf(w, a2.y, T[7]); T[8] = fmaf(w, a2.z, T[8]);
    }
    for (int k = KWF; k < c.KW; ++k) {                         // models with more than 4 weights per vertex
      const float w = wv[k];
      const float* Aj = As + wi[k] * 12;                        // ELL padding rows carry weight 0
      T[0] = fmaf(w, Aj[0], T[0]); T[1] = fmaf(w, Aj[1], T[1]); T[2] = fmaf(w, Aj[2], T[2]);
      T[3] = fmaf(w, Aj[4], T[3]); T[4] = fmaf(w, Aj[5], T[4]); T[5] = fmaf(w, Aj[6], T[5]);
      T[6] = fmaf(w, Aj[8], T[6]); T[7] = fmaf(w, Aj[9], T[7]); T[8] = fmaf(w, Aj[10], T[8]);
    }
    float* d = dvp + (size_t)b * u.NCs + 3 * (s0 + l);
    d[0] = T[0] * gx + T[3] * gy + T[6] * gz;
    d[1] = T[1] * gx + T[4] * gy + T[7] * gz;
    d[2] = T[2] * gx + T[5] * gy + T[8] * gz;
  }
  if (ch == nchunk - 1)                                                                          // padding columns of the GEMM operand: any width
    for (int i = 3 * u.n + t; i < u.NCs; i += 256) dvp[(size_t)b * u.NCs + i] = 0.f;          // (a C-API caller may pad wider than 16)
  float* part = u.part + ((size_t)b * nchunk + ch) * LBS_PART_STRIDE(nj);
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
  if ((t & 63) == 0) { red[3 * (t >> 6)] = sx; red[3 * (t >> 6) + 1] = sy; red[3 * (t >> 6) + 2] = sz; }
  __syncthreads();
  if (t < 3) part[nj * 12 + t] = ((red[t] + red[3 + t]) + red[6 + t]) + red[9 + t];
  // ---- joint-major: dA[j][r][:] = sum over the chunk's entries of joint j of  w g_r (x) [v, 1]
  // A wave per joint; joints without entries in this chunk store zeros.  (Round 5 also built a thread-per-output form for joints with
  // few entries and measured it slower -- `lbs_bwd_chunk` 41 -> 69 us, a thread's loop is a chain of dependent LDS reads --
  // profiles/r05_ab_lbs_bwd_joint_threads.txt; the code was deleted in round 6.)
  for (int o = t; o < nj * 12; o += 256)
    if (tabs[o / 12 + 1] == tabs[o / 12]) part[o] = 0.f;
  const int wave = t >> 6, lane = t & 63;
  for (int j = wave; j < nj; j += 4) {
    const int q0 = tabs[j], q1 = tabs[j + 1];
    if (q1 == q0) continue;                                      // wave-uniform: zeroed above
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int q = q0 + lane; q < q1; q += 64) {
      const int l = (staged ? cus[q - e0] : u.jc_u[q]) - s0;
      const float w = staged ? cws[q - e0] : u.jc_w[q];
      const float vx = vs[3 * l], vy = vs[3 * l + 1], vz = vs[3 * l + 2];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float gv = gs[3 * l + r] * w;
        acc[4 * r] = fmaf(gv, vx, acc[4 * r]); acc[4 * r + 1] = fmaf(gv, vy, acc[4 * r + 1]);
        acc[4 * r + 2] = fmaf(gv, vz, acc[4 * r + 2]); acc[4 * r + 3] += gv;
      }
    }
    // the twelve sums over the wave in one transposed butterfly (lane e ends with entry e) and ONE store by twelve lanes: twelve
    // separate wave sums + twelve stores by lane 0 were ~170 of the ~220 vector instructions a joint costs, and the kernel is bound
    // by exactly those (round 4: 4 instruction-issue cycles each; frames per workgroup, occupancy and table staging made no difference)
    const float tot = wave_transposed_sum16(acc, lane);
    if (lane < 12) part[j * 12 + lane] = tot;
  }
}
// frame b's chunk partials added in chunk order; `pad`: also zero the padding columns of the GEMM operand d(v_posed)
__device__ __forceinline__ void lbs_bwd_reduce_body(const VertexSetBwd& u, int nj, int nchunk, float* __restrict__ dvp, float* __restrict__ dA,
                                                    float* __restrict__ dtransl, int b, bool pad) {
  const int t = threadIdx.x, st = LBS_PART_STRIDE(nj);
  const float* p = u.part + (size_t)b * nchunk * st;
  // one thread per output, its chunk partials loaded 8 at a time (a dependent load per addend was 21 L2 round trips:
  // 18 us for a kernel that moves 5 MB) and added in chunk order
  const int nout = nj * 12 + (dtransl ? 3 : 0);
  for (int i = t; i < nout; i += 256) {
    const int col = i < nj * 12 ? i : nj * 12 + (i - nj * 12);
    float a = 0.f;
    for (int c0 = 0; c0 < nchunk; c0 += 8) {
      float r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = p[(size_t)(c0 + u < nchunk ? c0 + u : nchunk - 1) * st + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) if (c0 + u < nchunk) a += r[u];
    }
    if (i < nj * 12) dA[(size_t)b * nj * 12 + i] = a;
    else dtransl[(size_t)b * 3 + (i - nj * 12)] = a;
  }
  if (pad) for (int i = 3 * u.n + t; i < u.NCs; i += 256) dvp[(size_t)b * u.NCs + i] = 0.f;     // padding columns of the GEMM operand
}
__global__ void __launch_bounds__(256)
lbs_bwd_reduce_kernel(VertexSetBwd u, int nj, int nchunk, float* __restrict__ dvp, float* __restrict__ dA, float* __restrict__ dtransl) {
  lbs_bwd_reduce_body(u, nj, nchunk, dvp, dA, dtransl, (int)blockIdx.x, true);
}
// ONE launch for the two fixed-order reductions that end the all-vertex backward (round 5): workgroups 0 .. B - 1 add the chunk partials
// of d(A) / d(transl), the rest the slab partials of the feature-gradient GEMM dX = Dk . d(v_posed) -- the GEMM's partial kernel runs
// BEFORE this launch, on d(v_posed) whose padding columns the chunk kernel has zeroed (one launch and one kernel boundary less)
__global__ void __launch_bounds__(256)
lbs_gemm_reduce_kernel(VertexSetBwd u, int nj, int nchunk, float* __restrict__ dA, float* __restrict__ dtransl, int B,
                       const float* __restrict__ part, int M, int S, float* __restrict__ C, int ldc) {
  const int id = (int)blockIdx.x;
  if (id < B) lbs_bwd_reduce_body(u, nj, nchunk, nullptr, dA, dtransl, id, false);
  else gemm_splitk_reduce_body(part, M, B, S, C, ldc, id - B);
}

// the staged (frame in LDS) kernel takes the set; only that one has the fused d(verts) form
bool lbs_verts_bwd_fusable(const SkinConst& c, const VertexSetBwd& u, int nj) {
  return u.n <= LBS_BWD_STAGE && nj <= 64 && (long)u.n * c.KW <= LBS_BWD_NNZ;
}

int lbs_verts_bwd(const SkinConst& c, const VertexSetBwd& u, const float* A, int nj, const float* v_posed, int vp_rows,
                  const float* dverts, int B, int Bp, float* dvp, float* dA, float* dtransl, float* dX, hipStream_t s,
                  const FitFuse* fuse) {
  if (u.n <= 0 || B <= 0 || (u.NCs % 16) || u.NCs < 3 * u.n) return LEMO_ERR_SHAPE;
  (void)Bp;
  if (fuse && (!lbs_verts_bwd_fusable(c, u, nj) || fuse->fc.n != u.n)) return LEMO_ERR_SHAPE;
  if (!fuse && !dverts) return LEMO_ERR_ARG;
  if (lbs_verts_bwd_fusable(c, u, nj))
    {
    if (fuse) hipLaunchKernelGGL((lbs_bwd_frame_kernel<true, true>), dim3(B), dim3(1024), 0, s, c, u, A, nj, v_posed, vp_rows, dverts, dvp, dA, dtransl, *fuse);
    else hipLaunchKernelGGL((lbs_bwd_frame_kernel<true, false>), dim3(B), dim3(1024), 0, s, c, u, A, nj, v_posed, vp_rows, dverts, dvp, dA, dtransl, FitFuse{});
  }
  else if (nj <= 64 && u.jcsr_chunk && u.jc_u && u.jc_w && u.part && B <= u.part_frames) {
    const int nchunk = (u.n + LBS_DENSE_CHUNK - 1) / LBS_DENSE_CHUNK;
    hipLaunchKernelGGL(lbs_bwd_chunk_kernel, dim3(nchunk, B), dim3(256), 0, s, c, u, A, nj, v_posed, vp_rows, dverts, dvp);
    static const bool two = getenv("LEMO_LBS_TWO_REDUCES") != nullptr;       // A/B switch: the form up to round 4 (reduce, GEMM, reduce)
    if (!two && u.gemm_part && u.gemm_slabs > 0 && B <= 128) {               // chunk partials and GEMM slab partials reduced in ONE launch
      int e = (int)hipGetLastError();
      if (e) return e;
      e = gemm_nt16_splitk_partials(u.Dk, u.NCs, dvp, u.NCs, 512, B, u.NCs, u.gemm_part, u.gemm_slabs, s, u.DkG);
      if (e) return e;
      hipLaunchKernelGGL(lbs_gemm_reduce_kernel, dim3(B + gemm_splitk_reduce_blocks(512, B)), dim3(256), 0, s, u, nj, nchunk, dA, dtransl, B,
                         u.gemm_part, 512, u.gemm_slabs, dX, 512);
      return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(lbs_bwd_reduce_kernel, dim3(B), dim3(256), 0, s, u, nj, nchunk, dvp, dA, dtransl);
  }
  else if (nj <= 64) {
    hipLaunchKernelGGL(lbs_bwd_zero_kernel, dim3(B), dim3(256), 0, s, dvp, u.NCs, 3 * u.n, dA, nj * 12, dtransl);
    hipLaunchKernelGGL(lbs_bwd_dense_kernel, dim3((u.n + LBS_DENSE_CHUNK - 1) / LBS_DENSE_CHUNK, B), dim3(256), 0, s, c, u, A, nj,
                       v_posed, vp_rows, dverts, dvp, dA, dtransl);
  } else
    hipLaunchKernelGGL((lbs_bwd_frame_kernel<false, false>), dim3(B), dim3(256), 0, s, c, u, A, nj, v_posed, vp_rows, dverts, dvp, dA, dtransl, FitFuse{});
  int e = (int)hipGetLastError();
  if (e) return e;
  // dX[b][k] = sum_col Dk[k][col] dvp[b][col]  : A = Dk (M = 512 features), B = dvp (N = B frames)
  if (u.gemm_part && u.gemm_slabs > 0 && B <= 128)
    return gemm_nt16_splitk(u.Dk, u.NCs, dvp, u.NCs, 512, B, u.NCs, dX, 512, u.gemm_part, u.gemm_slabs, s, u.DkG);
  return gemm_nt16(u.Dk, u.NCs, dvp, u.NCs, 512, B, u.NCs, dX, 512, nullptr, nullptr, 0, 0, s);
}

// ------------------------------------------------------------------------------------------------
// joints = cat[posed joints, vertex picks, barycentric landmarks] (+ transl)
// (smplx VertexJointSelector + vertices2landmarks; 55 + 21 + 51 = 127 for SMPL-X)
// `verts` already contain transl when it is applied; Jtr does not.
// ------------------------------------------------------------------------------------------------
__global__ void joints_assemble_kernel(const float* __restrict__ Jtr, int nj, const float* __restrict__ verts, int vrows,
                                       const int* __restrict__ extra_rows, int n_extra, const int* __restrict__ lmk_rows,
                                       const float* __restrict__ lmk_bary, int n_lmk, const float* __restrict__ transl,
                                       float* __restrict__ joints) {
  const int b = blockIdx.x, nt = nj + n_extra + n_lmk;
  for (int w = threadIdx.x; w < nt * 3; w += blockDim.x) {
    const int i = w / 3, cc = w % 3;
    float v;
    if (i < nj) v = Jtr[((size_t)b * nj + i) * 3 + cc] + (transl ? transl[(size_t)b * 3 + cc] : 0.f);
    else if (i < nj + n_extra) v = verts[((size_t)b * vrows + extra_rows[i - nj]) * 3 + cc];
    else {
      const int l = i - nj - n_extra;
      v = 0.f;
      for (int f = 0; f < 3; ++f) v += verts[((size_t)b * vrows + lmk_rows[3 * l + f]) * 3 + cc] * lmk_bary[3 * l + f];
    }
    joints[((size_t)b * nt + i) * 3 + cc] = v;
  }
}

int joints_assemble(const float* Jtr, int nj, const float* verts, int vrows, const int* extra_rows, int n_extra,
                    const int* lmk_rows, const float* lmk_bary, int n_lmk, const float* transl, int B, float* joints,
                    hipStream_t s) {
  hipLaunchKernelGGL(joints_assemble_kernel, dim3(B), dim3(128), 0, s, Jtr, nj, verts, vrows, extra_rows, n_extra,
                     lmk_rows, lmk_bary, n_lmk, transl, joints);
  return (int)hipGetLastError();
}

}  // namespace lemo

CENSUS_SETTER(lemo_census_set_lbs)
