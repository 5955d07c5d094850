// What does a dependent kernel boundary cost for the conv kernel's launch geometry?  (diagnostic)
// 40 dependent launches in a captured graph, replayed 20 times: us per kernel for
//   grid 256 x {64, 256, 512} threads, dynamic LDS {0, 52, 104, 153} KB, and a body that writes {0, 8.4 MB} (dirty L2 lines
//   that the boundary has to write back) after reading what the previous launch wrote.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WRITE>
__global__ void body(const float* __restrict__ in, float* __restrict__ out, int n_per_block) {
  extern __shared__ float sm[];
  const int t = threadIdx.x, b = blockIdx.x;
  if (WRITE) {
    for (int i = t; i < n_per_block; i += blockDim.x) out[(size_t)b * n_per_block + i] = in[(size_t)b * n_per_block + i] + 1.f;
  } else if (t == 0) {
    out[b] = in[b] + 1.f;
  }
  if (t == 1000000) sm[0] = 0.f;
}
int main() {
  const int nblk = 256, npb = 8400000 / 4 / nblk;       // 8.4 MB per kernel in total
  float *a, *b; (void)hipMalloc(&a, (size_t)nblk * npb * 4); (void)hipMalloc(&b, (size_t)nblk * npb * 4);
  (void)hipMemset(a, 0, (size_t)nblk * npb * 4); (void)hipMemset(b, 0, (size_t)nblk * npb * 4);
  hipStream_t s; (void)hipStreamCreate(&s);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&body<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&body<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int wr = 0; wr < 2; ++wr)
    for (int thr : {64, 256, 512})
      for (int lds : {0, 52, 104, 153}) {
        const int N = 40;
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < N; ++i) {
          const float* in = (i & 1) ? b : a; float* out = (i & 1) ? a : b;
          if (wr) body<1><<<nblk, thr, lds * 1024, s>>>(in, out, npb); else body<0><<<nblk, thr, lds * 1024, s>>>(in, out, npb);
        }
        (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int w = 0; w < 3; ++w) (void)hipGraphLaunch(ge, s);
        (void)hipStreamSynchronize(s);
        (void)hipEventRecord(e0, s);
        const int R = 20;
        for (int r = 0; r < R; ++r) (void)hipGraphLaunch(ge, s);
        (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s, 256 x %3d threads, %3d KB LDS: %.2f us per kernel\n", wr ? "copy 8.4 MB" : "trivial     ", thr, lds, ms * 1e3 / R / N);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
      }
  return 0;
}
