// cost of a dependent kernel boundary inside a hipGraph replay (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void medium(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
  int* d; float* f; (void)hipMalloc(&d, 4); (void)hipMalloc(&f, 4 << 20);
  hipStream_t s; (void)hipStreamCreate(&s);
  for (int mode = 0; mode < 3; ++mode) {
    for (int N : {10, 40, 160}) {
      hipGraph_t g; hipGraphExec_t ge;
      (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
      for (int i = 0; i < N; ++i) {
        if (mode == 0) tiny<<<1, 64, 0, s>>>(d);
        else if (mode == 1) tiny<<<119, 256, 0, s>>>(d);
        else medium<<<4096, 256, 0, s>>>(f, 1 << 20);
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
      printf("mode %d (%s) N=%3d: %.2f us per replay, %.3f us per kernel\n", mode, mode == 0 ? "1x64 trivial" : mode == 1 ? "119x256 trivial" : "4 MB elementwise", N, ms * 1e3 / R, ms * 1e3 / R / N);
    }
  }
  return 0;
}
