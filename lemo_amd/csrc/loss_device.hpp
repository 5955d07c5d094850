// Device bodies of the two loss kernels that follow the encoder's forward pass; shared so that the fitting engine
// can run both in ONE launch (fit_losses_kernel in loss_kernels.hip: they are independent, ~6 us each, and a kernel
// boundary costs ~3 us).
#pragma once
#include "kernels.hpp"

namespace lemo {

__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// canon[0..8] = R0 (row-major, columns x,y,z axes), canon[9..11] = marker 0 of frame 0
__device__ __forceinline__ void canonical_frame(const float* verts, int nrows, const int* row81, const float* Jtr,
                                                int nj, const float* transl, float* canon, const float* cam2world = nullptr) {
  // joints[0, 1:3] (posed joints + transl), opt_amass_temp.py:368-375
  float j1[3], j2[3];
  for (int k = 0; k < 3; ++k) {
    const float tr = transl ? transl[k] : 0.f;
    j1[k] = Jtr[3 * 1 + k] + tr;
    j2[k] = Jtr[3 * 2 + k] + tr;
  }
  if (cam2world) {                 // PROX: the frame is built in world coordinates (fitting_temp_slide.py:1001-1010)
    float w1[3], w2[3];
    for (int i = 0; i < 3; ++i) {
      w1[i] = cam2world[3 * i] * j1[0] + cam2world[3 * i + 1] * j1[1] + cam2world[3 * i + 2] * j1[2] + cam2world[9 + i];
      w2[i] = cam2world[3 * i] * j2[0] + cam2world[3 * i + 1] * j2[1] + cam2world[3 * i + 2] * j2[2] + cam2world[9 + i];
    }
    for (int i = 0; i < 3; ++i) { j1[i] = w1[i]; j2[i] = w2[i]; }
  }
  (void)nj;
  float xx = j2[0] - j1[0], xy = j2[1] - j1[1];
  const float nx = sqrtf(xx * xx + xy * xy);
  xx /= nx; xy /= nx;
  // y = cross(z, x) = (-x.y, x.x, 0), normalised
  float yx = -xy, yy = xx;
  const float ny = sqrtf(yx * yx + yy * yy);
  yx /= ny; yy /= ny;
  canon[0] = xx; canon[1] = yx; canon[2] = 0.f;
  canon[3] = xy; canon[4] = yy; canon[5] = 0.f;
  canon[6] = 0.f; canon[7] = 0.f; canon[8] = 1.f;
  if (cam2world) {                 // world marker differences are R (v - v0): fold R into the matrix, canon' = R^T R0
    float m[9];
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 3; ++c)
        m[3 * k + c] = cam2world[k] * canon[c] + cam2world[3 + k] * canon[3 + c] + cam2world[6 + k] * canon[6 + c];
    for (int e = 0; e < 9; ++e) canon[e] = m[e];
  }
  const float* m0 = verts + (size_t)row81[0] * 3;       // frame 0, marker 0
  (void)nrows;
  canon[9] = m0[0]; canon[10] = m0[1]; canon[11] = m0[2];
}



// ---- latent smoothness loss (opt_amass_temp.py:390-391) + its gradient, fused ------------------
//   loss = mean_{c,y,x<W-1} (z[c,y,x+1]-z[c,y,x])^2
//   dpre[c,y,x] = coef * 2 * ((z[x]-z[x-1])[x>=1] - (z[x+1]-z[x])[x<=W-2]) * lrelu'(z[c,y,x])
// with coef = weight / (C*H*(W-1)).  Per-block partial sums of the squared differences go to
// `partial[blockIdx.x]` (fixed-order final reduction elsewhere -> deterministic).
#define LEMO_SMOOTH_ITEMS 4          // float4s per thread: all 12 loads of a thread are issued before the first use
__device__ __forceinline__ void smooth_loss_body(int blk, const float* __restrict__ z, float* __restrict__ dpre, float* __restrict__ partial,
                                                 int H, int W, int C, float coef2, double* __restrict__ acc) {
  __shared__ float red[4];
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const int n = P * (C >> 3) * 2;                                   // one item per float4
  float sq = 0.f;
  float4 cv[LEMO_SMOOTH_ITEMS], lv[LEMO_SMOOTH_ITEMS], rv[LEMO_SMOOTH_ITEMS];
  size_t ov[LEMO_SMOOTH_ITEMS];
  bool hasl[LEMO_SMOOTH_ITEMS], hasr[LEMO_SMOOTH_ITEMS], valid[LEMO_SMOOTH_ITEMS];
#pragma unroll
  for (int k = 0; k < LEMO_SMOOTH_ITEMS; ++k) {
    const int idx = (blk * LEMO_SMOOTH_ITEMS + k) * (int)blockDim.x + (int)threadIdx.x;
    valid[k] = idx < n;
    const int ic = valid[k] ? idx : n - 1;
    const int half = ic & 1, rest = ic >> 1;
    const int g = rest / P, p = rest - g * P;
    const int y = p / W, x = p - y * W;
    ov[k] = ((size_t)g * HWp + (y + 1) * Wp + (x + 1)) * 8 + 4 * half;
    hasl[k] = x >= 1; hasr[k] = x <= W - 2;
    cv[k] = ld4(z + ov[k]);
    lv[k] = ld4(z + ov[k] - 8);                                     // x = 0 / W-1: the zero border column, switched off below
    rv[k] = ld4(z + ov[k] + 8);
  }
#pragma unroll
  for (int k = 0; k < LEMO_SMOOTH_ITEMS; ++k) {
    const float4 c = cv[k], l = lv[k], r = rv[k];
    float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hasl[k]) { gr.x += c.x - l.x; gr.y += c.y - l.y; gr.z += c.z - l.z; gr.w += c.w - l.w; }
    if (hasr[k]) {
      const float d0 = r.x - c.x, d1 = r.y - c.y, d2 = r.z - c.z, d3 = r.w - c.w;
      if (valid[k]) sq += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
      gr.x -= d0; gr.y -= d1; gr.z -= d2; gr.w -= d3;
    }
    if (valid[k])
      st4(dpre + ov[k], make_float4(coef2 * gr.x * lrelu_grad_from_out(c.x), coef2 * gr.y * lrelu_grad_from_out(c.y),
                                    coef2 * gr.z * lrelu_grad_from_out(c.z), coef2 * gr.w * lrelu_grad_from_out(c.w)));
  }
  const float s = block_sum(sq, red);
  if (threadIdx.x == 0) {
    // f64 accumulation (order effects ~1e-16, invisible after the f32 cast), spread over 32 slots of 16
    // doubles: 2000+ blocks on ONE address serialise in L2 (measured 28 us)
    if (acc) atomicAdd(acc + (blk & 31) * 16, (double)s);
    else partial[blk] = s;
  }
}

// Loss accumulators (f64 [32 slots][16], zeroed at the start of every iteration by the pose-stage kernel;
// a block adds to slot blockIdx & 31 to keep the L2 atomics from serialising on one address):
//   [0] marker L1 sum ; [1+k] contact-velocity sum, [5+k] count (k = 4 foot sets) ; [9] smoothness sum of
//   squares ; [10] sum z^2 ; [11] sum betas^2 ; [12] sum hands^2.
// Accumulating f32 block sums into f64 with atomics is order-dependent only at ~1e-16 relative, far
// below the f32 value that is finally reported.
__device__ __forceinline__ void vertex_loss_body(int b, FitConst fc, const float* __restrict__ verts, int nrows, const float* __restrict__ target,
                              const float* __restrict__ contact, const float* __restrict__ shape,
                              const float* __restrict__ other, int B, double* __restrict__ accg) {
  __shared__ float red[4][12];
  const int t = threadIdx.x;
  float acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  for (int w = t; w < fc.n67 * 3; w += 256) {
    const int m = w / 3, c = w % 3;
    acc[0] += fabsf(verts[((size_t)b * nrows + fc.row67[m]) * 3 + c] - target[((size_t)b * fc.n67 + m) * 3 + c]);
  }
  if (b < B - 1) {
    for (int k = 0; k < 4; ++k) {
      if (contact[(size_t)b * 4 + k] != 1.f) continue;
      for (int q = fc.foot_start[k] + t; q < fc.foot_start[k + 1]; q += 256) {
        const float* v0 = verts + ((size_t)b * nrows + fc.foot_row[q]) * 3;
        const float* v1 = v0 + (size_t)nrows * 3;
        const float vx = (v1[0] - v0[0]) * 30.f, vy = (v1[1] - v0[1]) * 30.f, vz = (v1[2] - v0[2]) * 30.f;
        const float sp = sqrtf(vx * vx + vy * vy + vz * vz);
        if (sp - 0.1f > 0.f) { acc[1 + k] += sp; acc[5 + k] += 1.f; }
      }
    }
  }
  // L2 priors of this frame (opt_amass_temp.py:397-404): z (32), betas (10), hands (24)
  if (t < 32) { const float v = other[(size_t)b * 56 + t]; acc[9] = v * v; }
  else if (t >= 64 && t < 74) { const float v = shape[(size_t)b * 10 + (t - 64)]; acc[10] = v * v; }
  else if (t >= 128 && t < 152) { const float v = other[(size_t)b * 56 + 32 + (t - 128)]; acc[11] = v * v; }
  // 12 wave sums, ONE barrier, then 12 threads finish and publish in parallel (was: 12 block sums = 24 barriers,
  // 12 serial atomics from thread 0); the order of the adds is fixed -> deterministic
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = wave_sum(acc[i]);
  if ((t & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) red[t >> 6][i] = acc[i];
  }
  __syncthreads();
  if (t < 12) {
    const float v = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
    if (v != 0.f) atomicAdd(accg + (b & 31) * 16 + (t < 9 ? t : t + 1), (double)v);
  }
}

// ---- d(total)/d(verts) pieces (opt_amass_temp.py:359-425 differentiated by hand) --------------------------------
// losses[0..6] = marker, vposer, shape, hand, contact, smooth, total ; losses[8..11] = 1/count per foot set
// weights[0..5] = rec_markers, vposer, shape, hand, contact_vel, smooth   (opt_amass_temp.py:47-52)
__device__ __forceinline__ void finalize_losses(const double* tot, int B, int n67, double smooth_count,
                                                const float* weights, float* losses) {
  const float l_marker = (float)(tot[0] / ((double)B * n67 * 3));
  float l_contact = 0.f;
  for (int k = 0; k < 4; ++k) {
    const double cnt = tot[5 + k];
    const float part = cnt >= 1.0 ? (float)(tot[1 + k] / cnt) : 0.f;
    l_contact = l_contact + part;
    losses[8 + k] = cnt >= 1.0 ? (float)(1.0 / cnt) : 0.f;
  }
  const float l_smooth = (float)(tot[9] / smooth_count);
  const float l_vposer = (float)(tot[10] / ((double)B * 32));
  const float l_shape = (float)(tot[11] / ((double)B * 10));
  const float l_hand = (float)(tot[12] / ((double)B * 24));
  float total = weights[0] * l_marker + weights[1] * l_vposer;
  total = total + weights[2] * l_shape;
  total = total + weights[3] * l_hand;
  total = total + weights[4] * l_contact;
  total = total + weights[5] * l_smooth;
  losses[0] = l_marker; losses[1] = l_vposer; losses[2] = l_shape; losses[3] = l_hand;
  losses[4] = l_contact; losses[5] = l_smooth; losses[6] = total; losses[7] = 0.f;
}


// total of accumulator `i` over its 32 slots
__device__ __forceinline__ double loss_slot_total(const double* __restrict__ acc, int i) {
  double v = 0.0;
  for (int sl = 0; sl < 32; ++sl) v += acc[sl * 16 + i];
  return v;
}


// gradient of the weighted total w.r.t. vertex u of the active set, frame b.  `losses` = the finalised record
// (losses[8 + k] = 1 / count of foot set k).
// Every global read is issued up front with clamped addresses and the terms are switched by selects: written the
// natural way (a load inside each `if`), the function was a chain of 5-6 dependent cold round trips (~7 us per block).
struct DvIdx { int row, m67, fm, m81; };
__device__ __forceinline__ DvIdx dverts_indices(const FitConst& fc, int u) {
  DvIdx ix;
  ix.row = fc.u_row[u]; ix.m67 = fc.u_m67[u]; ix.fm = fc.u_foot_mask[u]; ix.m81 = fc.u_m81[u];
  return ix;
}

// everything dverts_compute reads from global memory, loaded by dverts_load (so a caller can put a barrier -- the
// loss record -- between the two without exposing the reads behind it)
struct DvRegs {
  float p[3], pn[3], pm[3], tgv[3], xstd[3], ct[4], ctm[4], cn[9], wm, wc;
  float dx[3][2][4];
  bool on[3][2][4];
  bool fast;
  int m67, fm, m81;
};

__device__ __forceinline__ void dverts_load(const FitConst& fc, const DvertsIn& in, int b, const DvIdx ix, DvRegs& r) {
  const float* __restrict__ verts = in.verts; const float* __restrict__ target = in.target;
  const float* __restrict__ contact = in.contact; const float* __restrict__ dx0 = in.dx0;
  const float* __restrict__ canon = in.canon; const float* __restrict__ weights = in.weights;
  const int nrows = in.nrows, B = in.B;
  const int D = 3 * fc.n81, W = B - 1 + 16, nd = B - 1;
  const int row = ix.row, m67 = ix.m67, fm = ix.fm, m81 = ix.m81;
  // ---- reads
  const int bn = min(b + 1, B - 1), bp = max(b - 1, 0), bc = max(min(b, B - 2), 0);
  const float* v = verts + ((size_t)b * nrows + row) * 3;
  const float* v1 = verts + ((size_t)bn * nrows + row) * 3;
  const float* vm = verts + ((size_t)bp * nrows + row) * 3;
  const float* tg = target + ((size_t)b * fc.n67 + max(m67, 0)) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) { r.p[c] = v[c]; r.pn[c] = v1[c]; r.pm[c] = vm[c]; r.tgv[c] = tg[c]; r.xstd[c] = fc.Xstd[3 * max(m81, 0) + c]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { r.ct[k] = contact[(size_t)bc * 4 + k]; r.ctm[k] = contact[(size_t)bp * 4 + k]; }
#pragma unroll
  for (int e = 0; e < 9; ++e) r.cn[e] = canon[e];
  r.wm = weights[0] / ((float)in.Bn * fc.n67 * 3); r.wc = weights[4];
  // smoothness-image gradient: feature row d = 3 m81 + c is read by padded rows y = d + 1 (+ one reflected copy for
  // d == 1 or d == D - 2) and, for each of the two time differences, padded columns t' + 8 (+ one reflected copy near
  // either end).  2 x 2 x 2 candidate reads per component, absent ones point at the main one and are switched off.
  r.fast = D >= 5 && nd >= 18;                   // the reflected copies are then mutually exclusive
  r.m67 = m67; r.fm = fm; r.m81 = m81;
  if (r.fast) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int d = 3 * max(m81, 0) + c;
      const int yA = d + 1, yB = d == 1 ? 0 : (d == D - 2 ? D + 1 : -1);
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        const int tp = side == 0 ? b - 1 : b;
        const bool tv = tp >= 0 && tp <= nd - 1;
        const int tq = min(max(tp, 0), nd - 1);
        const int xA = tq + 8, xB = (tq >= 1 && tq <= 8) ? 8 - tq : ((tq >= nd - 9 && tq <= nd - 2) ? 2 * (nd - 1) - tq + 8 : -1);
        r.on[c][side][0] = tv;               r.on[c][side][1] = tv && xB >= 0;
        r.on[c][side][2] = tv && yB >= 0;    r.on[c][side][3] = tv && yB >= 0 && xB >= 0;
        const int yb = yB >= 0 ? yB : yA, xb = xB >= 0 ? xB : xA;
        r.dx[c][side][0] = dx0[(size_t)yA * W + xA]; r.dx[c][side][1] = dx0[(size_t)yA * W + xb];
        r.dx[c][side][2] = dx0[(size_t)yb * W + xA]; r.dx[c][side][3] = dx0[(size_t)yb * W + xb];
      }
    }
  }
}

__device__ __forceinline__ void dverts_compute(const FitConst& fc, const DvertsIn& in, const float* losses, int b, const DvRegs& r,
                                               float& gx_out, float& gy_out, float& gz_out) {
  const float* __restrict__ dx0 = in.dx0;
  const int B = in.B, D = 3 * fc.n81, W = B - 1 + 16, nd = B - 1;
  const int m67 = r.m67, fm = r.fm, m81 = r.m81;
  const float wm = r.wm, wc = r.wc;
  const bool fast = r.fast;
  // ---- marker term: d|v - target| (opt_amass_temp.py:359)
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (m67 >= 0) {
    const float d0 = r.p[0] - r.tgv[0], d1 = r.p[1] - r.tgv[1], d2 = r.p[2] - r.tgv[2];
    gx += wm * (d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f));
    gy += wm * (d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f));
    gz += wm * (d2 > 0.f ? 1.f : (d2 < 0.f ? -1.f : 0.f));
  }
  // ---- contact-velocity term (:414-425): speeds above 0.1 of the foot sets in contact, both neighbours
  if (fm) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!((fm >> k) & 1)) continue;
      const float coef = wc * losses[8 + k] * 30.f;
      if (b < B - 1 && r.ct[k] == 1.f) {
        const float vx = (r.pn[0] - r.p[0]) * 30.f, vy = (r.pn[1] - r.p[1]) * 30.f, vz = (r.pn[2] - r.p[2]) * 30.f;
        const float sp = sqrtf(vx * vx + vy * vy + vz * vz);
        if (sp - 0.1f > 0.f) { const float q = coef / sp; gx -= q * vx; gy -= q * vy; gz -= q * vz; }
      }
      if (b >= 1 && r.ctm[k] == 1.f) {
        const float vx = (r.p[0] - r.pm[0]) * 30.f, vy = (r.p[1] - r.pm[1]) * 30.f, vz = (r.p[2] - r.pm[2]) * 30.f;
        const float sp = sqrtf(vx * vx + vy * vy + vz * vz);
        if (sp - 0.1f > 0.f) { const float q = coef / sp; gx += q * vx; gy += q * vy; gz += q * vz; }
      }
    }
  }
  // ---- smoothness term through the marker image (:376-391)
  if (m81 >= 0) {
    float dg[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int d = 3 * m81 + c;
      float acc = 0.f;
      if (fast) {
#pragma unroll
        for (int side = 0; side < 2; ++side) {             // side 0: difference t' = b - 1 (+), side 1: t' = b (-)
          float sv = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) sv += r.on[c][side][q] ? r.dx[c][side][q] : 0.f;
          acc += side == 0 ? sv : -sv;
        }
      } else {
        int ys[3]; int ny = 0;
        ys[ny++] = d + 1;
        if (d == 1) ys[ny++] = 0;
        if (d == D - 2) ys[ny++] = D + 1;
        for (int side = 0; side < 2; ++side) {
          const int tp = side == 0 ? b - 1 : b;
          if (tp < 0 || tp > nd - 1) continue;
          int xs[3]; int nx = 0;
          xs[nx++] = tp + 8;
          if (tp >= 1 && tp <= 8) xs[nx++] = 8 - tp;
          if (tp >= nd - 9 && tp <= nd - 2) xs[nx++] = 2 * (nd - 1) - tp + 8;
          float sv = 0.f;
          for (int iy = 0; iy < ny; ++iy)
            for (int jx = 0; jx < nx; ++jx) sv += dx0[(size_t)ys[iy] * W + xs[jx]];
          acc += side == 0 ? sv : -sv;
        }
      }
      dg[c] = acc / r.xstd[c];
    }
    gx += r.cn[0] * dg[0] + r.cn[1] * dg[1] + r.cn[2] * dg[2];
    gy += r.cn[3] * dg[0] + r.cn[4] * dg[1] + r.cn[5] * dg[2];
    gz += r.cn[6] * dg[0] + r.cn[7] * dg[1] + r.cn[8] * dg[2];
  }
  gx_out = gx; gy_out = gy; gz_out = gz;
}

__device__ __forceinline__ void dverts_vertex(const FitConst& fc, const DvertsIn& in, const float* losses, int b, const DvIdx ix,
                                              float& gx_out, float& gy_out, float& gz_out) {
  DvRegs r;
  dverts_load(fc, in, b, ix, r);
  dverts_compute(fc, in, losses, b, r, gx_out, gy_out, gz_out);
}

}  // namespace lemo
