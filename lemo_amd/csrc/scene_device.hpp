// Trilinear lookup of a signed-distance volume == F.grid_sample(sdf, grid, padding_mode='border', align_corners=False)
// as the reference calls it (temp_prox/fitting_temp_slide.py:685-688, 700-703), with the gradient w.r.t. the point.
// Shared by the stand-alone sampler (scene_kernels.hip) and the PROX engine's loss kernels (prox_kernels.hip).
#pragma once
#include "kernels.hpp"

namespace lemo {

struct SdfVol {
  const float* sdf;            // [D][H][W]
  int D, H, W;
  float g0[3], sc[3];          // grid_min, 2 / (grid_max - grid_min)
};

// p: world coordinates.  norm = (p - gmin) / (gmax - gmin) * 2 - 1 ; the reference feeds norm[..., [2,1,0]] as the grid:
// grid x (-> W index) = norm z, grid y (-> H) = norm y, grid z (-> D) = norm x.  grad (may be null): d val / d p.
__device__ __forceinline__ float sdf_at(const SdfVol& v, float px, float py, float pz, float* grad) {
  const int D = v.D, H = v.H, W = v.W;
  const float nx = (px - v.g0[0]) * v.sc[0] - 1.f, ny = (py - v.g0[1]) * v.sc[1] - 1.f, nz = (pz - v.g0[2]) * v.sc[2] - 1.f;
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
#define SDF_AT(d_, h_, w_) v.sdf[((size_t)(d_) * H + (h_)) * W + (w_)]
  const float c000 = SDF_AT(d0, h0, w0), c001 = SDF_AT(d0, h0, w1), c010 = SDF_AT(d0, h1, w0), c011 = SDF_AT(d0, h1, w1);
  const float c100 = SDF_AT(d1, h0, w0), c101 = SDF_AT(d1, h0, w1), c110 = SDF_AT(d1, h1, w0), c111 = SDF_AT(d1, h1, w1);
#undef SDF_AT
  const float a00 = c000 * (1.f - tw) + c001 * tw, a01 = c010 * (1.f - tw) + c011 * tw;
  const float a10 = c100 * (1.f - tw) + c101 * tw, a11 = c110 * (1.f - tw) + c111 * tw;
  const float b0 = a00 * (1.f - th) + a01 * th, b1 = a10 * (1.f - th) + a11 * th;
  if (grad) {
    const float gw = ((c001 - c000) * (1.f - th) + (c011 - c010) * th) * (1.f - td) +
                     ((c101 - c100) * (1.f - th) + (c111 - c110) * th) * td;
    const float gh = (a01 - a00) * (1.f - td) + (a11 - a10) * td;
    const float gd = b1 - b0;
    // d fw / d p_z = sz * W / 2 ; d fh / d p_y = sy * H / 2 ; d fd / d p_x = sx * D / 2
    grad[0] = gd * md * v.sc[0] * 0.5f * D;
    grad[1] = gh * mh * v.sc[1] * 0.5f * H;
    grad[2] = gw * mw * v.sc[2] * 0.5f * W;
  }
  return b0 * (1.f - td) + b1 * td;
}

static inline SdfVol make_sdf_vol(const float* sdf, int D, int H, int W, const float* gmin, const float* gmax) {
  SdfVol v;
  v.sdf = sdf; v.D = D; v.H = H; v.W = W;
  for (int k = 0; k < 3; ++k) { v.g0[k] = gmin[k]; v.sc[k] = 2.f / (gmax[k] - gmin[k]); }
  return v;
}

}  // namespace lemo
