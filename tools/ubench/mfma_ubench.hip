// fp32 MFMA issue-rate microbenchmark (diagnostic): dependent chains, 1 vs 2 waves per SIMD,
// 1 vs 2 accumulators.  hipcc --offload-arch=gfx950 -O3 mfma_ubench.hip -o mfma_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int SHAPE>
__global__ void k(float* out, unsigned long long* cyc, int n) {
  f32x16 acc[NACC]; f32x4 acc4[NACC];
  for (int a = 0; a < NACC; ++a) { for (int r = 0; r < 16; ++r) acc[a][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[a][r] = 0.f; }
  float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        if (SHAPE == 32) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
        else acc4[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc4[a], 0, 0, 0);
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) { for (int r = 0; r < 16; ++r) s += acc[a][r]; for (int r = 0; r < 4; ++r) s += acc4[a][r]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int NACC, int SHAPE>
void run(const char* name, int threads, int blocks) {
  const int n = 256;                               // n*8*NACC MFMAs per wave
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * threads * blocks); hipMalloc(&cyc, 8 * (threads / 64) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, SHAPE><<<blocks, threads>>>(out, cyc, n); hipDeviceSynchronize();
  hipEventRecord(e0); k<NACC, SHAPE><<<blocks, threads>>>(out, cyc, n); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((threads / 64) * blocks);
  hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= h.size();
  const double nm = (double)n * 8 * NACC;
  const double flop = nm * (SHAPE == 32 ? 4096.0 : 2048.0) * h.size();
  printf("%-34s waves/SIMD %d  cycles/MFMA(per wave) %.1f  wall %.1f us  %.1f TFLOP/s  clock~%.2f GHz\n", name, threads / 256,
         mean / nm, ms * 1e3, flop / (ms * 1e-3) / 1e12, mean / (ms * 1e-3) / 1e9);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1, 32>("32x32x2 1 acc", 256, 256);
  run<1, 32>("32x32x2 1 acc", 512, 256);
  run<2, 32>("32x32x2 2 acc", 256, 256);
  run<2, 32>("32x32x2 2 acc", 512, 256);
  run<4, 32>("32x32x2 4 acc", 256, 256);
  run<1, 16>("16x16x4 1 acc", 256, 256);
  run<1, 16>("16x16x4 1 acc", 512, 256);
  run<2, 16>("16x16x4 2 acc", 512, 256);
  run<4, 16>("16x16x4 4 acc", 256, 256);
  run<4, 16>("16x16x4 4 acc", 512, 256);
  run<1, 32>("32x32x2 1 acc, 4 w/SIMD", 1024, 256);
  return 0;
}
