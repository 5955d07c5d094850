// What this box actually sustains ("confirm the peaks with a probe rather than trusting the sheet"):
//   (1) v_mfma_f32_32x32x16_bf16 issue rate with every CU busy for a few milliseconds (clock included) -> the bf16 matrix
//       roof the split convolution is priced against (sheet: 2.5 PFLOP/s dense = 256 CUs x 4096 FLOP/clk x 2.4 GHz)
//   (2) HBM stream: read-only sum, write-only fill and copy of 1 GiB (sheet: ~8 TB/s)
// hipcc --offload-arch=gfx950 -O3 peak_ubench.hip -o peak_ubench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(512) mfma_k(float* out, unsigned long long* cyc, int n) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(1.0f + threadIdx.x * 1e-3f); y[e] = (__bf16)(0.5f + e * 1e-2f); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

__global__ void __launch_bounds__(256) rd_k(const float4* __restrict__ p, size_t n4, float* out) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = p[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;                      // keeps the loads alive
}
__global__ void __launch_bounds__(256) wr_k(float4* __restrict__ p, size_t n4, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(v, v, v, v);
}
__global__ void __launch_bounds__(256) cp_k(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

template <typename F> static float time_ms(F f, int reps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) f();
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  printf("%s: %d CUs, clock %d MHz (sheet)\n", pr.name, cus, pr.clockRate / 1000);
  // NACC independent accumulator chains per wave x waves per SIMD: how many chains the pipe needs to stay full
  // (the split convolution runs 2 waves x 2 accumulators per SIMD)
#define RUN(NACC, THREADS)                                                                                          \
  {                                                                                                                 \
    const int threads = THREADS, blocks = cus, n = 16384 / NACC;                                                    \
    float* out; unsigned long long* cyc;                                                                            \
    (void)hipMalloc(&out, sizeof(float) * threads * blocks); (void)hipMalloc(&cyc, 8 * (threads / 64) * blocks);    \
    float ms = 1e9f;                                                                                                \
    for (int rep = 0; rep < 3; ++rep) ms = fminf(ms, time_ms([&] { mfma_k<NACC><<<blocks, threads>>>(out, cyc, n); }, 5)); \
    std::vector<unsigned long long> h((threads / 64) * blocks);                                                     \
    (void)hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);                                            \
    double mean = 0; for (auto v : h) mean += v; mean /= h.size();                                                  \
    const double nm = (double)n * 8 * NACC, flop = nm * 32768.0 * h.size();                                         \
    printf("bf16 MFMA 32x32x16: %d wave(s)/SIMD x %d accumulator chain(s): %5.1f s_memtime ticks per MFMA per SIMD, %.2f ms, %4.0f TFLOP/s sustained\n", \
           threads / 256, NACC, mean / nm / (threads / 256), ms, flop / (ms * 1e-3) / 1e12);                        \
    (void)hipFree(out); (void)hipFree(cyc);                                                                         \
  }
  RUN(4, 512) RUN(2, 512) RUN(1, 512) RUN(4, 256) RUN(2, 256) RUN(1, 256)
  printf("(sheet: 32 shader cycles per MFMA per SIMD = 2516 TFLOP/s at 2.4 GHz; ticks/MFMA x TFLOP/s tells the s_memtime rate)\n");
  {
    const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
    float4 *a, *b; float* o;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&o, 4);
    (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes);
    for (int grid : {cus * 8, cus * 32}) {
      const float r = time_ms([&] { rd_k<<<grid, 256>>>(a, n4, o); }, 10);
      const float w = time_ms([&] { wr_k<<<grid, 256>>>(b, n4, 1.f); }, 10);
      const float c = time_ms([&] { cp_k<<<grid, 256>>>(a, b, n4); }, 10);
      printf("HBM stream over 1 GiB, grid %5d x 256: read %.2f TB/s   write %.2f TB/s   copy %.2f TB/s (read + write bytes)\n", grid,
             bytes / (r * 1e-3) / 1e12, bytes / (w * 1e-3) / 1e12, 2.0 * bytes / (c * 1e-3) / 1e12);
    }
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(o);
  }
  return 0;
}
