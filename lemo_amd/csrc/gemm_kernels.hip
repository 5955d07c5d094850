// Small dense layers on the fitting path as fp32-MFMA "NT" GEMMs:
//     C[n][m] = epi( sum_k A[m][k] * B[n][k] )        A: [M][lda], B: [N][ldb], both K-contiguous
// Used for the VPoser decoder MLP (vposer_smpl.py:107-115: 32->512->512->126, frames as N) and for
// d(pose feature) = Dk . dvp in the LBS backward.  These GEMMs are tiny (<= 0.1 GFLOP) and would be
// pure latency chains if one wave walked K alone, so one 16x16 output tile is owned by a whole
// workgroup: its 4 waves split K, each issues ALL its operand loads (<= 16 dwordx4 per lane) before
// its v_mfma_f32_16x16x4_f32 chain (one exposed memory round trip), and the partial tiles are
// reduced through LDS in a fixed order (deterministic).
#include <cstdlib>
#include "kernels.hpp"
#include "gemm_reduce.hpp"

namespace lemo {

#define GEMM_MAXCH 8            // 16-k chunks per wave per round

// epi 0: none | 1: lrelu(v + bias[m]) | 2: v + bias[m] | 3: v * lrelu'(aux[n][m])
template <int EPI>
__global__ void __launch_bounds__(256)
gemm_nt16_kernel(const float* __restrict__ Am, int lda, const float* __restrict__ Bm, int ldb, int M, int N, int K,
                 float* __restrict__ C, int ldc, const float* __restrict__ bias, const float* __restrict__ aux, int ldaux,
                 int b_kg8 /* 0: B row-major [N][ldb] ; else B in KG8 layout [K/8][b_kg8 rows][8] (the pose stage's Xg) */) {
  __shared__ __attribute__((aligned(16))) float red[4][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int mtiles = M >> 4;
  const int mt = blockIdx.x % mtiles, nt = blockIdx.x / mtiles;
  const int nrow = nt * 16 + i;
  const float* ap = Am + (size_t)(mt * 16 + i) * lda + 4 * q;
  // KG8: element (n, k) sits at ((k >> 3) * rows + n) * 8 + (k & 7); lane quarter q reads k = 16 c + 4 q .. + 3,
  // i.e. group 2 c + (q >> 1), floats 4 (q & 1) ..: affine in the chunk index c with step 16 * rows
  const int nclamp = nrow < N ? nrow : N - 1;
  const float* bp = b_kg8 ? Bm + ((size_t)(q >> 1) * b_kg8 + nclamp) * 8 + 4 * (q & 1) : Bm + (size_t)nclamp * ldb + 4 * q;
  const int bstep = b_kg8 ? 16 * b_kg8 : 16;
  const int k16 = K >> 4, per = (k16 + 3) >> 2;
  const int c0 = wave * per, c1 = (c0 + per < k16) ? c0 + per : k16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int cb = c0; cb < c1; cb += GEMM_MAXCH) {
    float4 a[GEMM_MAXCH], b[GEMM_MAXCH];
#pragma unroll
    for (int u = 0; u < GEMM_MAXCH; ++u) {
      const int c = (cb + u < c1) ? cb + u : c1 - 1;             // clamp: unconditional loads
      a[u] = ld4(ap + c * 16);
      b[u] = ld4(bp + (size_t)c * bstep);
    }
    __builtin_amdgcn_sched_barrier(0);                           // all loads in flight before the MFMA chain
#pragma unroll
    for (int u = 0; u < GEMM_MAXCH; ++u) {
      if (cb + u < c1) {                                         // wave-uniform
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc, 0, 0, 0);
      }
    }
  }
  st4(&red[wave][lane][0], make_float4(acc[0], acc[1], acc[2], acc[3]));
  __syncthreads();
  if (wave == 0 && nrow < N) {
    float4 v = ld4(&red[0][lane][0]);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 p = ld4(&red[w][lane][0]);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    const int m0 = mt * 16 + 4 * q;                               // D: col = n (lane&15), rows = 4q + r -> m
    if (EPI == 1 || EPI == 2) {
      const float4 bb = ld4(bias + m0);
      v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
      if (EPI == 1) { v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w); }
    } else if (EPI == 3) {
      const float4 y = ld4(aux + (size_t)nrow * ldaux + m0);
      v.x *= lrelu_grad_from_out(y.x); v.y *= lrelu_grad_from_out(y.y);
      v.z *= lrelu_grad_from_out(y.z); v.w *= lrelu_grad_from_out(y.w);
    }
    st4(C + (size_t)nrow * ldc + m0, v);
  }
}

int gemm_nt16(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc,
              const float* bias, const float* aux, int ldaux, int epi, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0 || (M & 15) || (K & 15) || (lda & 3) || (ldb & 3) || (ldc & 3)) return LEMO_ERR_SHAPE;
  if ((epi == 1 || epi == 2) && !bias) return LEMO_ERR_ARG;
  if (epi == 3 && (!aux || (ldaux & 3))) return LEMO_ERR_ARG;
  const dim3 grid((M >> 4) * ((N + 15) >> 4));
#define GL(E) hipLaunchKernelGGL((gemm_nt16_kernel<E>), grid, dim3(256), 0, s, A, lda, B, ldb, M, N, K, C, ldc, bias, aux, ldaux, 0)
  if (epi == 0) GL(0); else if (epi == 1) GL(1); else if (epi == 2) GL(2); else if (epi == 3) GL(3); else return LEMO_ERR_ARG;
#undef GL
  return (int)hipGetLastError();
}

// ---- long-K form: C[n][m] = sum_k A[m][k] B[n][k] with K in the tens of thousands and only M x N <= 512 x 128 outputs (the
// feature gradient of the all-vertex LBS backward: A = Dk [512][3V], B = d(v_posed) [frames][3V]).  gemm_nt16 gives every
// 16 x 16 output tile its own workgroup: 224 workgroups that each walk the whole K and re-read A once per frame tile
// (7 x 64 MB) -- 107 us.  Here a workgroup owns 64 rows of A x ALL frames x one K slab (A is streamed exactly once, the
// B slab is shared by the 8 row blocks through L2), writes its partial tile, and a second launch adds the slabs in slab
// order (deterministic).
// GEMM_SK_NT (gemm_reduce.hpp): frame tiles of 16 (N <= 128)
// Measured (rocprofv3, B = 100, K = 31440).  v1: 32 slabs, every wave loading its own A and all 8 B fragments of a step and
// waiting for them: 101.8 us (no better than the 107 us it replaced).  v2: 96 slabs + operands of the next step requested
// before the MFMAs of the current one: 76 us -- still 9 KB of global loads per wave and step for 32 MFMAs, one step
// (0.43 us of MFMA work) of look-ahead against ~2 us of memory latency.  v3 (this): the 4 waves of a workgroup need the SAME
// B fragments, so the B tile of a step (128 frames x 16 k = 8 KB) is staged once per workgroup through LDS (double
// buffered, global -> register -> LDS two steps ahead) and only the wave's own A fragment (1 KB) comes straight from
// global memory, three steps ahead: 3 KB of global loads per wave and step instead of 9.
__global__ void __launch_bounds__(256)
gemm_nt16_splitk_kernel(const float* __restrict__ Am, int lda, const float* __restrict__ Bm, int ldb, int M, int N, int K, int S,
                        float* __restrict__ part) {
  constexpr int NTHR = 256, NSLOT = 2;                          // (same staging roles as the bf16 kernel with MW = 2)
  __shared__ __attribute__((aligned(16))) float Bs[2][4][GEMM_SK_NT * 16][4];      // [buffer][k quarter][frame][4 k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int mblocks = M >> 6;
  const int mb = blockIdx.x % mblocks, slab = blockIdx.x / mblocks;
  const int mt = mb * 4 + wave;
  const int k16 = K >> 4, per = (k16 + S - 1) / S;
  const int c0 = slab * per, c1 = (c0 + per < k16) ? c0 + per : k16;
  const int nst = c1 - c0;
  const float* ap = Am + (size_t)(mt * 16 + i) * lda + 4 * q;
  // staging role of this thread: float4 idx = tid + 256 r -> frame idx >> 2, k quarter idx & 3
  const float* bsrc[NSLOT];
  int bdst[NSLOT];
#pragma unroll
  for (int r = 0; r < NSLOT; ++r) {
    const int idx = tid + NTHR * r, n = idx >> 2, kq = idx & 3;
    bsrc[r] = Bm + (size_t)(n < N ? n : N - 1) * ldb + 4 * kq;
    bdst[r] = (kq * (GEMM_SK_NT * 16) + n) * 4;
  }
  f32x4 acc[GEMM_SK_NT];
#pragma unroll
  for (int t = 0; t < GEMM_SK_NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (nst > 0) {
    auto clampc = [&](int c) { return c < c1 ? c : c1 - 1; };
    float4 a[3], bg[2][2];                                     // A of steps s, s+1, s+2 ; B (global) of steps s+1, s+2
#pragma unroll
    for (int u = 0; u < 3; ++u) a[u] = ld4(ap + (size_t)clampc(c0 + u) * 16);
    {                                                          // step 0 straight into LDS buffer 0, steps 1 and 2 into registers
      float4 b0[NSLOT];
#pragma unroll
      for (int r = 0; r < NSLOT; ++r) b0[r] = ld4(bsrc[r] + (size_t)c0 * 16);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < NSLOT; ++r) bg[u][r] = ld4(bsrc[r] + (size_t)clampc(c0 + 1 + u) * 16);
#pragma unroll
      for (int r = 0; r < 2; ++r) st4(&Bs[0][0][0][0] + bdst[r], b0[r]);
    }
    __syncthreads();
    for (int s_ = 0; s_ < nst; ++s_) {
      const int buf = s_ & 1;
      const float4 av = a[0];
      // B fragments of this step from LDS, then the MFMAs
      float4 bf[GEMM_SK_NT];
#pragma unroll
      for (int t = 0; t < GEMM_SK_NT; ++t) bf[t] = ld4(&Bs[buf][q][t * 16 + i][0]);
      // publish step s+1 (held in registers since two steps ago) to the other buffer, refill the registers with step s+3
      if (s_ + 1 < nst) {
#pragma unroll
        for (int r = 0; r < 2; ++r) st4(&Bs[buf ^ 1][0][0][0] + bdst[r], bg[0][r]);
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) { bg[0][r] = bg[1][r]; bg[1][r] = ld4(bsrc[r] + (size_t)clampc(c0 + s_ + 3) * 16); }
      a[0] = a[1]; a[1] = a[2]; a[2] = ld4(ap + (size_t)clampc(c0 + s_ + 3) * 16);
#pragma unroll
      for (int t = 0; t < GEMM_SK_NT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bf[t].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bf[t].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bf[t].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bf[t].w, acc[t], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  // D: col = n (lane & 15), rows 4 q + r -> m ;  part[slab][n][m]
  float* pp = part + (size_t)slab * (GEMM_SK_NT * 16) * M;
#pragma unroll
  for (int t = 0; t < GEMM_SK_NT; ++t)
    st4(pp + (size_t)(t * 16 + i) * M + mt * 16 + 4 * q, make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]));
}
// v4: the same decomposition on the bf16 matrix cores with exact fp32 operands (3 bf16 pieces per operand, 6 products per 16-deep
// k-step, fp32 accumulate: conv_split_kernels.hip has the error analysis) -- the fp32 MFMA floor of this GEMM (4.1 GFLOP at
// 157 TFLOP/s = 26 us) was above its HBM floor (64 MB of A: ~12 us).  Wave = one 32-row M-tile x two 32-frame N-tiles; A fragments
// (8 consecutive k of one row per lane) straight from global memory three steps ahead and split in registers, the B tile of a
// step split ONCE per workgroup on its way into LDS ([piece][k half][frame][8 bf16]: every fragment one conflict-free b128).
// MW = 32-row M-tiles per workgroup (round 3).  The workgroup stages the B tile of a step (128 frames x 16 k = 8 KB) once and
// every M-tile wave pair consumes it: with MW = 2 (64 rows, rounds 1-2) the eight M-blocks of the PROX GEMM (M = 512) each
// re-read all of dvp -- 8 x 12.6 MB against 64 MB of A -- and a workgroup pulled 16 KB per step into its CU for 64 rows; a CU
// takes in cold data at ~10 B/clk whatever is asked of it, so the launch lasted 42 us for a 12 us HBM floor.
// MW = 4: 128 rows per workgroup, 8 waves, dvp read 4 times, half the partial slabs for the same number of workgroups.
#ifndef LEMO_SK_RA
#define LEMO_SK_RA 3
#endif
constexpr int SK_RA = LEMO_SK_RA;      // steps of A (and B) in flight per lane: the operands of step s + RA are requested in step s
// (round 5, same box interleaved, PROX window it/s: RA = 3 2283 / 2273, RA = 5 2272 / 2285, RA = 7 2238 / 2249; 128 slabs instead of 64
// (two workgroups per CU, 109 VGPRs allow it) 2260 / 2255, with RA = 5 2213 / 2211: the launch is not waiting for operands it could have
// asked for earlier, and more partial-sum traffic costs more than a second resident workgroup hides -- profiles/r05_ab_variant9_gemm_ring.txt)
template <int MW>
__global__ void __launch_bounds__(128 * MW)
gemm_nt16_splitk_bf16_kernel(const float* __restrict__ Am, int lda, const float* __restrict__ Bm, int ldb, int M, int N, int K, int S,
                             float* __restrict__ part, int a_grouped) {
  constexpr int NTHR = 128 * MW, NSLOT = 512 / NTHR;             // staging slots per thread: 512 float4 per step
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][3][2][GEMM_SK_NT * 16][16];   // [buffer][piece][k half][frame][8 bf16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int mblocks = M / (32 * MW);
  const int mb = blockIdx.x % mblocks, slab = blockIdx.x / mblocks;
  const int mtile = wave % MW, npair = wave / MW;                // rows mb*32*MW + mtile*32 .. +31 ; frames npair*64 .. +63
  const int k16 = K >> 4, per = (k16 + S - 1) / S;
  const int c0 = slab * per, c1 = (c0 + per < k16) ? c0 + per : k16;
  const int nst = c1 - c0;
  // A element (row, k): row-major [M][lda], or k-chunk major [K/16][M][16] (a_grouped: a step's 64 x 16 tile is contiguous)
  const int arow = mb * (32 * MW) + mtile * 32 + j;
  const float* ap = a_grouped ? Am + (size_t)arow * 16 + 8 * h : Am + (size_t)arow * lda + 8 * h;
  const size_t astep = a_grouped ? (size_t)M * 16 : 16;
  // staging role: float4 idx = tid + 256 r -> frame idx >> 2, k quarter idx & 3 (4 k each)
  const float* bsrc[NSLOT];
  int bdst[NSLOT];
#pragma unroll
  for (int r = 0; r < NSLOT; ++r) {
    const int idx = tid + NTHR * r, n = idx >> 2, kq = idx & 3;
    bsrc[r] = Bm + (size_t)(n < N ? n : N - 1) * ldb + 4 * kq;
    bdst[r] = (((kq >> 1) * (GEMM_SK_NT * 16) + n) * 16) + (kq & 1) * 8;       // + piece * (2 * 128 * 16)
  }
  constexpr int PIECE = 2 * GEMM_SK_NT * 16 * 16, BUF = 3 * PIECE;
  unsigned char* bs = &Bs[0][0][0][0][0];
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  if (nst > 0) {
    auto clampc = [&](int c) { return c < c1 ? c : c1 - 1; };
    float4 a[SK_RA][2], bg[SK_RA - 1][NSLOT];                  // A of steps s .. s+RA-1 ; B (global) of steps s+1 .. s+RA-1
#pragma unroll
    for (int u = 0; u < SK_RA; ++u) { a[u][0] = ld4(ap + (size_t)clampc(c0 + u) * astep); a[u][1] = ld4(ap + (size_t)clampc(c0 + u) * astep + 4); }
#define SK_STORE_B(BUFI, V)                                                                          \
    _Pragma("unroll") for (int r = 0; r < NSLOT; ++r) {                                            \
      uint2 p0, p1, p2;                                                                            \
      split3x4(V[r], p0, p1, p2);                                                                  \
      unsigned char* d = bs + (BUFI) * BUF + bdst[r];                                              \
      *reinterpret_cast<uint2*>(d) = p0; *reinterpret_cast<uint2*>(d + PIECE) = p1; *reinterpret_cast<uint2*>(d + 2 * PIECE) = p2; \
    }
    {
      float4 b0[NSLOT];
#pragma unroll
      for (int r = 0; r < NSLOT; ++r) b0[r] = ld4(bsrc[r] + (size_t)c0 * 16);
#pragma unroll
      for (int u = 0; u < SK_RA - 1; ++u)
#pragma unroll
        for (int r = 0; r < NSLOT; ++r) bg[u][r] = ld4(bsrc[r] + (size_t)clampc(c0 + 1 + u) * 16);
      SK_STORE_B(0, b0)
    }
    __syncthreads();
    for (int s_ = 0; s_ < nst; ++s_) {
      const int buf = s_ & 1;
      // this step's A fragment: 8 consecutive k of row j (lane half h: k 8h .. 8h+7) as three bf16 pieces
      uint4 af[3];
      {
        uint2 l0, l1, l2, u0, u1, u2;
        split3x4(a[0][0], l0, l1, l2);
        split3x4(a[0][1], u0, u1, u2);
        af[0] = make_uint4(l0.x, l0.y, u0.x, u0.y); af[1] = make_uint4(l1.x, l1.y, u1.x, u1.y); af[2] = make_uint4(l2.x, l2.y, u2.x, u2.y);
      }
      uint4 bf[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bf[t][p] = *reinterpret_cast<const uint4*>(bs + buf * BUF + p * PIECE + ((h * (GEMM_SK_NT * 16)) + npair * 64 + t * 32 + j) * 16);
      // publish step s+1 (held in registers since RA-1 steps ago) to the other buffer, refill the registers with step s+RA
      if (s_ + 1 < nst) { SK_STORE_B(buf ^ 1, bg[0]) }
#pragma unroll
      for (int u = 0; u + 1 < SK_RA - 1; ++u)
#pragma unroll
        for (int r = 0; r < NSLOT; ++r) bg[u][r] = bg[u + 1][r];
#pragma unroll
      for (int r = 0; r < NSLOT; ++r) bg[SK_RA - 2][r] = ld4(bsrc[r] + (size_t)clampc(c0 + s_ + SK_RA) * 16);
#pragma unroll
      for (int u = 0; u + 1 < SK_RA; ++u) { a[u][0] = a[u + 1][0]; a[u][1] = a[u + 1][1]; }
      a[SK_RA - 1][0] = ld4(ap + (size_t)clampc(c0 + s_ + SK_RA) * astep); a[SK_RA - 1][1] = ld4(ap + (size_t)clampc(c0 + s_ + SK_RA) * astep + 4);
#define SK_MFMA1(SA, SB)                                                                             \
      _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                \
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[SA]), __builtin_bit_cast(bf16x8, bf[t][SB]), acc[t], 0, 0, 0);
      SK_MFMA1(0, 2) SK_MFMA1(2, 0) SK_MFMA1(1, 1) SK_MFMA1(0, 1) SK_MFMA1(1, 0) SK_MFMA1(0, 0)      // smallest products first
#undef SK_MFMA1
      __syncthreads();
    }
#undef SK_STORE_B
  }
  // D: col = lane & 31 -> frame, rows (r & 3) + 8 (r >> 2) + 4 h -> m ;  part[slab][n][m]
  float* pp = part + (size_t)slab * (GEMM_SK_NT * 16) * M;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq)
      if (npair * 64 + t * 32 + j < N)                       // frames past N are padding: nobody reads their partials
        st4(pp + (size_t)(npair * 64 + t * 32 + j) * M + mb * (32 * MW) + mtile * 32 + 8 * rq + 4 * h,
            make_float4(acc[t][4 * rq], acc[t][4 * rq + 1], acc[t][4 * rq + 2], acc[t][4 * rq + 3]));
}
__global__ void __launch_bounds__(256)
gemm_splitk_reduce_kernel(const float* __restrict__ part, int M, int N, int S, float* __restrict__ C, int ldc) {
  gemm_splitk_reduce_body(part, M, N, S, C, ldc, (int)blockIdx.x);
}

int gemm_nt16_splitk_part_floats(int M, int S) { return S * GEMM_SK_NT * 16 * M; }

// the partial tiles only: the caller reduces them (gemm_splitk_reduce_body) in a launch of its own
int gemm_nt16_splitk_partials(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* part, int S, hipStream_t s,
                              const float* A_grouped) {
  if (M <= 0 || N <= 0 || N > GEMM_SK_NT * 16 || K <= 0 || (M & 63) || (K & 15) || (lda & 3) || (ldb & 3) || S < 1 || S > (K >> 4) || !part)
    return LEMO_ERR_SHAPE;
  static const bool fp32_mfma = getenv("LEMO_SPLITK_FP32") != nullptr;       // A/B switch (diagnostics): v3, the fp32-MFMA kernel
  if (fp32_mfma) hipLaunchKernelGGL(gemm_nt16_splitk_kernel, dim3((M >> 6) * S), dim3(256), 0, s, A, lda, B, ldb, M, N, K, S, part);
  else if ((M & 127) == 0) {                                 // 128 rows per workgroup (M % 128 == 0: the PROX feature-gradient GEMM)
    if (A_grouped) hipLaunchKernelGGL((gemm_nt16_splitk_bf16_kernel<4>), dim3((M >> 7) * S), dim3(512), 0, s, A_grouped, lda, B, ldb, M, N, K, S, part, 1);
    else hipLaunchKernelGGL((gemm_nt16_splitk_bf16_kernel<4>), dim3((M >> 7) * S), dim3(512), 0, s, A, lda, B, ldb, M, N, K, S, part, 0);
  }
  else if (A_grouped) hipLaunchKernelGGL((gemm_nt16_splitk_bf16_kernel<2>), dim3((M >> 6) * S), dim3(256), 0, s, A_grouped, lda, B, ldb, M, N, K, S, part, 1);
  else hipLaunchKernelGGL((gemm_nt16_splitk_bf16_kernel<2>), dim3((M >> 6) * S), dim3(256), 0, s, A, lda, B, ldb, M, N, K, S, part, 0);
  return (int)hipGetLastError();
}

int gemm_nt16_splitk(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc, float* part, int S,
                     hipStream_t s, const float* A_grouped) {
  if (ldc & 3) return LEMO_ERR_SHAPE;
  if (int rc = gemm_nt16_splitk_partials(A, lda, B, ldb, M, N, K, part, S, s, A_grouped)) return rc;
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(gemm_splitk_reduce_blocks(M, N)), dim3(256), 0, s, part, M, N, S, C, ldc);
  return (int)hipGetLastError();
}

// C[n][m] = sum_k A[m][k] * X(n, k) with X in the KG8 layout [K/8][rows][8]
int gemm_nt16_kg8(const float* A, int lda, const float* Xg, int rows, int M, int N, int K, float* C, int ldc, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0 || (M & 15) || (K & 15) || (lda & 3) || (ldc & 3) || N > rows) return LEMO_ERR_SHAPE;
  const dim3 grid((M >> 4) * ((N + 15) >> 4));
  hipLaunchKernelGGL((gemm_nt16_kernel<0>), grid, dim3(256), 0, s, A, lda, Xg, 0, M, N, K, C, ldc, (const float*)nullptr,
                     (const float*)nullptr, 0, rows);
  return (int)hipGetLastError();
}

}  // namespace lemo
