// Small dense layers on the fitting path as fp32-MFMA "NT" GEMMs:
//     C[n][m] = epi( sum_k A[m][k] * B[n][k] )        A: [M][lda], B: [N][ldb], both K-contiguous
// Used for the VPoser decoder MLP (vposer_smpl.py:107-115: 32->512->512->126, frames as N) and for
// d(pose feature) = Dk . dvp in the LBS backward.  These GEMMs are tiny (<= 0.1 GFLOP) and would be
// pure latency chains if one wave walked K alone, so one 16x16 output tile is owned by a whole
// workgroup: its 4 waves split K, each issues ALL its operand loads (<= 16 dwordx4 per lane) before
// its v_mfma_f32_16x16x4_f32 chain (one exposed memory round trip), and the partial tiles are
// reduced through LDS in a fixed order (deterministic).
#include "kernels.hpp"

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

// C[n][m] = sum_k A[m][k] * X(n, k) with X in the KG8 layout [K/8][rows][8]
int gemm_nt16_kg8(const float* A, int lda, const float* Xg, int rows, int M, int N, int K, float* C, int ldc, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0 || (M & 15) || (K & 15) || (lda & 3) || (ldc & 3) || N > rows) return LEMO_ERR_SHAPE;
  const dim3 grid((M >> 4) * ((N + 15) >> 4));
  hipLaunchKernelGGL((gemm_nt16_kernel<0>), grid, dim3(256), 0, s, A, lda, Xg, 0, M, N, K, C, ldc, (const float*)nullptr,
                     (const float*)nullptr, 0, rows);
  return (int)hipGetLastError();
}

}  // namespace lemo
