// accuracy + lane-map check: fp32 product via 3-way bf16 split on v_mfma_f32_32x32x16_bf16 (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define K 576
__device__ inline void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x; float r = x - (float)h; m = (__bf16)r; r = r - (float)m; l = (__bf16)r;
}
// A [32][K] row-major, B [K][32] row-major, out [mode][32][32]
__global__ void k(const float* A, const float* B, float* out) {
  int l = threadIdx.x, i = l & 31, hh = l >> 5;
  f32x16 acc = {0}, accs = {0}, acc32 = {0};
  for (int kc = 0; kc < K; kc += 16) {
    bf16x8 a[3], b[3];
    for (int j = 0; j < 8; ++j) {
      __bf16 h, m, lo;
      split3(A[i * K + kc + 8 * hh + j], h, m, lo); a[0][j] = h; a[1][j] = m; a[2][j] = lo;
      split3(B[(kc + 8 * hh + j) * 32 + i], h, m, lo); b[0][j] = h; b[1][j] = m; b[2][j] = lo;
    }
    // mode 0: all into one accumulator, small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    // mode 1: hi*hi separately from the rest
    accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], accs, 0, 0, 0);
  }
  for (int kc = 0; kc < K; kc += 2)
    acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + kc + hh], B[(kc + hh) * 32 + i], acc32, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
    out[0 * 1024 + row * 32 + i] = acc[r];
    out[1 * 1024 + row * 32 + i] = accs[r];
    out[2 * 1024 + row * 32 + i] = acc32[r];
  }
}
int main() {
  std::mt19937 g(1); std::normal_distribution<float> nd;
  std::vector<float> A(32 * K), B(K * 32), o(3 * 1024);
  for (int pass = 0; pass < 2; ++pass) {
    for (auto& v : A) v = pass ? std::fabs(nd(g)) : nd(g);
    for (auto& v : B) v = pass ? std::fabs(nd(g)) * 0.05f : nd(g) * 0.05f;
    float *dA, *dB, *dO; (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dO, o.size() * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dO); (void)hipMemcpy(o.data(), dO, o.size() * 4, hipMemcpyDeviceToHost);
    double e6 = 0, e1 = 0, e32 = 0, s6 = 0, s32 = 0, mx = 0, ef = 0, sf = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double r = 0; float f = 0; for (int kk = 0; kk < K; ++kk) { r += (double)A[i * K + kk] * B[kk * 32 + j]; f = fmaf(A[i * K + kk], B[kk * 32 + j], f); }
      mx = fmax(mx, fabs(r));
      double d6 = o[i * 32 + j] - r, d1 = o[1024 + i * 32 + j] - r, d32 = o[2048 + i * 32 + j] - r, df = f - r;
      e6 = fmax(e6, fabs(d6)); e1 = fmax(e1, fabs(d1)); e32 = fmax(e32, fabs(d32)); ef = fmax(ef, fabs(df)); s6 += d6; s32 += d32; sf += df;
    }
    printf("%s data: max|ref| %.3f  err/max: split6 %.3e (mean signed %.2e)  hi*hi only %.3e  f32 mfma %.3e (mean signed %.2e)  host fmaf chain %.3e (mean %.2e)\n",
           pass ? "positive" : "signed", mx, e6 / mx, s6 / 1024 / mx, e1 / mx, e32 / mx, s32 / 1024 / mx, ef / mx, sf / 1024 / mx);
  }
  return 0;
}
