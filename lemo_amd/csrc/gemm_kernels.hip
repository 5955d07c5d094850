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
                 float* __restrict__ C, int ldc, const float* __restrict__ bias, const float* __restrict__ aux, int ldaux) {
  __shared__ __attribute__((aligned(16))) float red[4][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int mtiles = M >> 4;
  const int mt = blockIdx.x % mtiles, nt = blockIdx.x / mtiles;
  const int nrow = nt * 16 + i;
  const float* ap = Am + (size_t)(mt * 16 + i) * lda + 4 * q;
  const float* bp = Bm + (size_t)(nrow < N ? nrow : N - 1) * ldb + 4 * q;
  const int k16 = K >> 4, per = (k16 + 3) >> 2;
  const int c0 = wave * per, c1 = (c0 + per < k16) ? c0 + per : k16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int cb = c0; cb < c1; cb += GEMM_MAXCH) {
    float4 a[GEMM_MAXCH], b[GEMM_MAXCH];
#pragma unroll
    for (int u = 0; u < GEMM_MAXCH; ++u) {
      const int c = (cb + u < c1) ? cb + u : c1 - 1;             // clamp: unconditional loads
      a[u] = ld4(ap + c * 16);
      b[u] = ld4(bp + c * 16);
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
#define GL(E) hipLaunchKernelGGL((gemm_nt16_kernel<E>), grid, dim3(256), 0, s, A, lda, B, ldb, M, N, K, C, ldc, bias, aux, ldaux)
  if (epi == 0) GL(0); else if (epi == 1) GL(1); else if (epi == 2) GL(2); else if (epi == 3) GL(3); else return LEMO_ERR_ARG;
#undef GL
  return (int)hipGetLastError();
}

}  // namespace lemo
