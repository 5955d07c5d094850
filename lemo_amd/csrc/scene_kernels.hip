// Scene terms of the PROX fitting loss (temp_prox/fitting_temp_slide.py:685-739): trilinear lookup of a
// signed-distance volume at body vertices == F.grid_sample(sdf, grid, padding_mode='border')
// (mode bilinear, align_corners=False, the torch>=1.3 default the reference runs with), plus its
// gradient w.r.t. the sampled point.
#include "kernels.hpp"

namespace lemo {

// pts: [N][3] world coordinates.  norm = (p - gmin)/(gmax - gmin)*2 - 1 ; the reference feeds
// norm[..., [2,1,0]] as the grid, i.e. grid x (-> W index) = norm z, grid y (-> H) = norm y, grid z (-> D) = norm x.
// sdf: [D][H][W].  val[N] ; dval[N][3] = d val / d p (may be null).
__global__ void __launch_bounds__(256)
sdf_sample_kernel(const float* __restrict__ sdf, int D, int H, int W, const float* __restrict__ pts, int N,
                  float gx0, float gy0, float gz0, float sx, float sy, float sz,      // gmin, 2/(gmax-gmin)
                  float* __restrict__ val, float* __restrict__ dval) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float nx = (pts[3 * i] - gx0) * sx - 1.f, ny = (pts[3 * i + 1] - gy0) * sy - 1.f, nz = (pts[3 * i + 2] - gz0) * sz - 1.f;
  // unnormalise (align_corners = False) and clamp to the border
  float fw = ((nz + 1.f) * W - 1.f) * 0.5f, fh = ((ny + 1.f) * H - 1.f) * 0.5f, fd = ((nx + 1.f) * D - 1.f) * 0.5f;
  float mw = 1.f, mh = 1.f, md = 1.f;                    // gradient multipliers (0 where clamped)
  if (fw < 0.f) { fw = 0.f; mw = 0.f; } else if (fw > (float)(W - 1)) { fw = (float)(W - 1); mw = 0.f; }
  if (fh < 0.f) { fh = 0.f; mh = 0.f; } else if (fh > (float)(H - 1)) { fh = (float)(H - 1); mh = 0.f; }
  if (fd < 0.f) { fd = 0.f; md = 0.f; } else if (fd > (float)(D - 1)) { fd = (float)(D - 1); md = 0.f; }
  const float w0f = floorf(fw), h0f = floorf(fh), d0f = floorf(fd);
  const int w0 = (int)w0f, h0 = (int)h0f, d0 = (int)d0f;
  const float tw = fw - w0f, th = fh - h0f, td = fd - d0f;
  const int w1 = w0 + 1 < W ? w0 + 1 : w0, h1 = h0 + 1 < H ? h0 + 1 : h0, d1 = d0 + 1 < D ? d0 + 1 : d0;
  // out-of-range corners (index == size) contribute zero weight in torch; with border clamping the
  // weight t is exactly 0 there, so re-using the in-range index is equivalent
#define SDF_AT(d_, h_, w_) sdf[((size_t)(d_) * H + (h_)) * W + (w_)]
  const float c000 = SDF_AT(d0, h0, w0), c001 = SDF_AT(d0, h0, w1), c010 = SDF_AT(d0, h1, w0), c011 = SDF_AT(d0, h1, w1);
  const float c100 = SDF_AT(d1, h0, w0), c101 = SDF_AT(d1, h0, w1), c110 = SDF_AT(d1, h1, w0), c111 = SDF_AT(d1, h1, w1);
#undef SDF_AT
  const float a00 = c000 * (1.f - tw) + c001 * tw, a01 = c010 * (1.f - tw) + c011 * tw;
  const float a10 = c100 * (1.f - tw) + c101 * tw, a11 = c110 * (1.f - tw) + c111 * tw;
  const float b0 = a00 * (1.f - th) + a01 * th, b1 = a10 * (1.f - th) + a11 * th;
  val[i] = b0 * (1.f - td) + b1 * td;
  if (dval) {
    const float gw = ((c001 - c000) * (1.f - th) + (c011 - c010) * th) * (1.f - td) +
                     ((c101 - c100) * (1.f - th) + (c111 - c110) * th) * td;
    const float gh = (a01 - a00) * (1.f - td) + (a11 - a10) * td;
    const float gd = b1 - b0;
    // d fw / d p_z = sz * W / 2 ; d fh / d p_y = sy * H / 2 ; d fd / d p_x = sx * D / 2
    dval[3 * i] = gd * md * sx * 0.5f * D;
    dval[3 * i + 1] = gh * mh * sy * 0.5f * H;
    dval[3 * i + 2] = gw * mw * sz * 0.5f * W;
  }
}

int sdf_sample(const float* sdf, int D, int H, int W, const float* pts, int N, const float* gmin, const float* gmax,
               float* val, float* dval, hipStream_t s) {
  if (D < 1 || H < 1 || W < 1 || N <= 0) return LEMO_ERR_SHAPE;
  hipLaunchKernelGGL(sdf_sample_kernel, dim3((N + 255) / 256), dim3(256), 0, s, sdf, D, H, W, pts, N, gmin[0], gmin[1], gmin[2],
                     2.f / (gmax[0] - gmin[0]), 2.f / (gmax[1] - gmin[1]), 2.f / (gmax[2] - gmin[2]), val, dval);
  return (int)hipGetLastError();
}

}  // namespace lemo
