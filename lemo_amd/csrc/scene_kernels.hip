// Scene terms of the PROX fitting loss (temp_prox/fitting_temp_slide.py:685-739): trilinear lookup of a
// signed-distance volume at body vertices == F.grid_sample(sdf, grid, padding_mode='border')
// (mode bilinear, align_corners=False, the torch>=1.3 default the reference runs with), plus its
// gradient w.r.t. the sampled point.
#include "scene_device.hpp"

namespace lemo {

// pts: [N][3] world coordinates; sdf: [D][H][W].  val[N] ; dval[N][3] = d val / d p (may be null).
__global__ void __launch_bounds__(256)
sdf_sample_kernel(SdfVol v, const float* __restrict__ pts, int N, float* __restrict__ val, float* __restrict__ dval) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float g[3];
  val[i] = sdf_at(v, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], dval ? g : nullptr);
  if (dval) { dval[3 * i] = g[0]; dval[3 * i + 1] = g[1]; dval[3 * i + 2] = g[2]; }
}

int sdf_sample(const float* sdf, int D, int H, int W, const float* pts, int N, const float* gmin, const float* gmax,
               float* val, float* dval, hipStream_t s) {
  if (D < 1 || H < 1 || W < 1 || N <= 0) return LEMO_ERR_SHAPE;
  hipLaunchKernelGGL(sdf_sample_kernel, dim3((N + 255) / 256), dim3(256), 0, s, make_sdf_vol(sdf, D, H, W, gmin, gmax), pts, N, val, dval);
  return (int)hipGetLastError();
}

}  // namespace lemo
