// Loss assembly of the AMASS temporal-fitting iteration (opt_amass_temp.py:366-453), its gradient
// w.r.t. the vertices that carry loss, and the Adam update (opt_amass_temp.py:342-352,454-455).
// All data-dependent branches of the reference (`.item()` host syncs at :431-443) are evaluated on
// the device: counts go through a tiny finalize kernel, nothing ever returns to the host.
#include "loss_device.hpp"

namespace lemo {

// Encoder input: canonicalise the 81 smoothness markers, normalise, temporal difference, reflect-pad
// (8,8,1,1)  -> x0 padded single-channel image [(H+2)][(W+2)], H = 3*n81+2, W = B-1+16.
__global__ void __launch_bounds__(256)
marker_feature_kernel(FitConst fc, const float* __restrict__ verts, int nrows, const float* __restrict__ Jtr, int nj,
                      const float* __restrict__ transl, int B, float* __restrict__ x0, float* __restrict__ canon_out) {
  __shared__ float cn[12];
  // this thread's reads first (they do not depend on the canonical frame): they are in flight while thread 0 walks
  // its own index -> vertex -> joint chain below
  const int D = 3 * fc.n81, H = D + 2, W = B - 1 + 16, Wp = W + 2;
  const int p = blockIdx.x * blockDim.x + threadIdx.x, pc = min(p, H * W - 1);
  const int y = pc / W, x = pc - y * W;
  const int d = reflect_idx(y - 1, D), tp = reflect_idx(x - 8, B - 1);
  const int m = d / 3, c = d - 3 * m;
  const float* v0 = verts + ((size_t)tp * nrows + fc.row81[m]) * 3;
  const float* v1 = v0 + (size_t)nrows * 3;
  const float a0 = v0[0], a1 = v0[1], a2 = v0[2], b0 = v1[0], b1 = v1[1], b2 = v1[2];
  const float xm = fc.Xmean[d], xs = fc.Xstd[d];
  if (threadIdx.x == 0) {
    canonical_frame(verts, nrows, fc.row81, Jtr, nj, transl, cn, fc.cam2world);
    if (blockIdx.x == 0) for (int i = 0; i < 12; ++i) canon_out[i] = cn[i];
  }
  __syncthreads();
  if (p >= H * W) return;
  const float g0 = (a0 - cn[9]) * cn[c] + (a1 - cn[10]) * cn[3 + c] + (a2 - cn[11]) * cn[6 + c];
  const float g1 = (b0 - cn[9]) * cn[c] + (b1 - cn[10]) * cn[3 + c] + (b2 - cn[11]) * cn[6 + c];
  const float n0 = (g0 - xm) / xs, n1 = (g1 - xm) / xs;
  x0[(size_t)(y + 1) * Wp + (x + 1)] = n1 - n0;
}

int marker_feature(const FitConst& fc, const float* verts, int nrows, const float* Jtr, int nj, const float* transl, int B,
                   float* x0, float* canon, hipStream_t s) {
  if (B < 10) return LEMO_ERR_SHAPE;                       // reflect pad of 8 needs >= 9 differences
  const int P = (3 * fc.n81 + 2) * (B - 1 + 16);
  hipLaunchKernelGGL(marker_feature_kernel, dim3((P + 255) / 256), dim3(256), 0, s, fc, verts, nrows, Jtr, nj, transl, B, x0, canon);
  return (int)hipGetLastError();
}

// ---- marker image + first encoder layer in one launch (fitting engine) ---------------------------------------
// marker_feature_kernel writes the single-channel image x0 and conv3x3_c1_kernel (conv_kernels.hip) reads it back nine
// times per output: two launches of ~5 us whose work is a few microseconds of latency.  Here a block owns a tile of
// MC_TY feature rows x MC_TX time columns: it computes the tile of x0 with its one-pixel halo into LDS (same formula,
// same order of operations as marker_feature_kernel), publishes the interior to the global x0 (kept for the parity
// tests and the C-ABI contract of the engine's buffers), and applies the 1 -> cout 3x3 layer from LDS.  The layer's
// weights are uniform across the block (every thread loops over all channel groups): scalar loads.
#define MC_TX 32
#define MC_TY 8
#define MC_PITCH (MC_TX + 2)
__global__ void __launch_bounds__(MC_TX * MC_TY)
marker_c1_kernel(FitConst fc, const float* __restrict__ verts, int nrows, const float* __restrict__ Jtr, int nj,
                 const float* __restrict__ transl, int B, const float* __restrict__ w, const float* __restrict__ bias,
                 float* __restrict__ x0, float* __restrict__ canon_out, float* __restrict__ out, int cout) {
  __shared__ float cn[12];
  __shared__ float xs[(MC_TY + 2) * MC_PITCH];
  constexpr int NT = MC_TX * MC_TY, NH = (MC_TY + 2) * MC_PITCH;     // 256 threads, 340 halo-tile values
  const int D = 3 * fc.n81, H = D + 2, W = B - 1 + 16, Wp = W + 2, HWp = (H + 2) * Wp;
  const int t = threadIdx.x;
  const int x00 = blockIdx.x * MC_TX, y00 = blockIdx.y * MC_TY;      // tile origin (image coordinates)
  // ---- reads of this thread's (up to two) halo-tile values, before anything that waits
  float a[2][3], b[2][3], xm[2], xsd[2];
  int cc[2];
  bool inside[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = min(t + NT * k, NH - 1);
    const int yy = y00 + i / MC_PITCH - 1, xx = x00 + i % MC_PITCH - 1;   // image coordinates of the value (-1 .. H / W)
    inside[k] = yy >= 0 && yy < H && xx >= 0 && xx < W;
    const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
    const int d = reflect_idx(yc - 1, D), tp = reflect_idx(xc - 8, B - 1);
    const int m = d / 3;
    cc[k] = d - 3 * m;
    const float* v0 = verts + ((size_t)tp * nrows + fc.row81[m]) * 3;
    const float* v1 = v0 + (size_t)nrows * 3;
#pragma unroll
    for (int e = 0; e < 3; ++e) { a[k][e] = v0[e]; b[k][e] = v1[e]; }
    xm[k] = fc.Xmean[d]; xsd[k] = fc.Xstd[d];
  }
  if (t == 0) {
    canonical_frame(verts, nrows, fc.row81, Jtr, nj, transl, cn, fc.cam2world);
    if (blockIdx.x == 0 && blockIdx.y == 0) for (int i = 0; i < 12; ++i) canon_out[i] = cn[i];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = t + NT * k;
    if (i < NH) {
      const int c = cc[k];
      const float g0 = (a[k][0] - cn[9]) * cn[c] + (a[k][1] - cn[10]) * cn[3 + c] + (a[k][2] - cn[11]) * cn[6 + c];
      const float g1 = (b[k][0] - cn[9]) * cn[c] + (b[k][1] - cn[10]) * cn[3 + c] + (b[k][2] - cn[11]) * cn[6 + c];
      const float n0 = (g0 - xm[k]) / xsd[k], n1 = (g1 - xm[k]) / xsd[k];
      const float v = inside[k] ? n1 - n0 : 0.f;                      // outside the image: the zero border of the padded x0
      xs[i] = v;
      const int ly = i / MC_PITCH, lx = i % MC_PITCH;
      if (inside[k] && ly >= 1 && ly <= MC_TY && lx >= 1 && lx <= MC_TX)   // the tile's own pixels
        x0[(size_t)(y00 + ly) * Wp + (x00 + lx)] = v;
    }
  }
  __syncthreads();
  const int ly = t / MC_TX, lx = t % MC_TX, y = y00 + ly, x = x00 + lx;
  if (y >= H || x >= W) return;
  float xin[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) xin[tp] = xs[(ly + tp / 3) * MC_PITCH + lx + tp % 3];
  const int poff = (y + 1) * Wp + (x + 1);
  for (int g = 0; g < (cout >> 3); ++g) {
    float r[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float* wc = w + (size_t)(g * 8 + c) * 9;
      float acc = 0.f;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) acc = fmaf(wc[tp], xin[tp], acc);
      r[c] = lrelu(acc + bias[g * 8 + c]);
    }
    float* o = out + ((size_t)g * HWp + poff) * 8;
    st4(o, make_float4(r[0], r[1], r[2], r[3]));
    st4(o + 4, make_float4(r[4], r[5], r[6], r[7]));
  }
}

int marker_c1(const FitConst& fc, const float* verts, int nrows, const float* Jtr, int nj, const float* transl, int B,
              const float* w, const float* bias, float* x0, float* canon, float* out, int cout, hipStream_t s) {
  if (B < 10 || (cout % 8)) return LEMO_ERR_SHAPE;
  const int H = 3 * fc.n81 + 2, W = B - 1 + 16;
  hipLaunchKernelGGL(marker_c1_kernel, dim3((W + MC_TX - 1) / MC_TX, (H + MC_TY - 1) / MC_TY), dim3(MC_TX * MC_TY), 0, s, fc, verts, nrows,
                     Jtr, nj, transl, B, w, bias, x0, canon, out, cout);
  return (int)hipGetLastError();
}

// (bodies: loss_device.hpp)
__global__ void __launch_bounds__(256)
vertex_loss_accumulate_kernel(FitConst fc, const float* __restrict__ verts, int nrows, const float* __restrict__ target,
                              const float* __restrict__ contact, const float* __restrict__ shape,
                              const float* __restrict__ other, int B, double* __restrict__ accg) {
  vertex_loss_body((int)blockIdx.x, fc, verts, nrows, target, contact, shape, other, B, accg);
}

// both post-encoder loss kernels in one launch: blocks [0, B) = per-frame vertex losses and priors, the rest =
// smoothness loss + d(pre-activation)
__global__ void __launch_bounds__(256)
fit_losses_kernel(int nsm, const float* __restrict__ z, float* __restrict__ dpre, int H, int W, int C, float coef2, double* __restrict__ sm_acc,
                  FitConst fc, const float* __restrict__ verts, int nrows, const float* __restrict__ target,
                              const float* __restrict__ contact, const float* __restrict__ shape,
                              const float* __restrict__ other, int B, double* __restrict__ accg) {
  // the per-frame blocks are a chain of dependent gathers (index -> vertex rows): first in the grid, so that the
  // streaming smoothness blocks run under their latency instead of the other way round
  if ((int)blockIdx.x < B) vertex_loss_body((int)blockIdx.x, fc, verts, nrows, target, contact, shape, other, B, accg);
  else smooth_loss_body((int)blockIdx.x - B, z, dpre, nullptr, H, W, C, coef2, sm_acc);
  (void)nsm;
}

int fit_losses(const float* z, float* dpre, int H, int W, int C, float coef2, double* sm_acc, const FitConst& fc, const float* verts,
               int nrows, const float* target, const float* contact, const float* shape, const float* other, int B, double* acc,
               hipStream_t s) {
  if (C % 8 || !sm_acc) return LEMO_ERR_SHAPE;
  const int nsm = smooth_loss_blocks(H, W, C);
  hipLaunchKernelGGL(fit_losses_kernel, dim3(nsm + B), dim3(256), 0, s, nsm, z, dpre, H, W, C, coef2, sm_acc, fc, verts, nrows, target,
                     contact, shape, other, B, acc);
  return (int)hipGetLastError();
}

int vertex_loss_accumulate(const FitConst& fc, const float* verts, int nrows, const float* target, const float* contact,
                           const float* shape, const float* other, int B, double* acc, hipStream_t s) {
  hipLaunchKernelGGL(vertex_loss_accumulate_kernel, dim3(B), dim3(256), 0, s, fc, verts, nrows, target, contact, shape, other, B, acc);
  return (int)hipGetLastError();
}

__global__ void loss_finalize_kernel(const double* __restrict__ acc, int B, int n67, double smooth_count,
                                     const float* __restrict__ weights, float* __restrict__ losses) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double tot[13];
    for (int i = 0; i < 13; ++i) { double v = 0.0; for (int sl = 0; sl < 32; ++sl) v += acc[sl * 16 + i]; tot[i] = v; }
    finalize_losses(tot, B, n67, smooth_count, weights, losses);
  }
}

int loss_finalize(const double* acc, int B, int n67, double smooth_count, const float* weights, float* losses, hipStream_t s) {
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, s, acc, B, n67, smooth_count, weights, losses);
  return (int)hipGetLastError();
}

// d(total)/d(verts) on the active vertex set U (block per frame; bodies in loss_device.hpp, shared with the fused
// LBS backward of the fitting engine).
__global__ void __launch_bounds__(256)
dverts_assemble_kernel(FitConst fc, const float* __restrict__ verts, int nrows, const float* __restrict__ target,
                       const float* __restrict__ contact, const float* __restrict__ dx0, const float* __restrict__ canon,
                       const float* __restrict__ weights, const double* __restrict__ acc, double smooth_count,
                       float* __restrict__ losses_out, int B, int Bn, float* __restrict__ dverts) {
  __shared__ float losses[12];
  __shared__ double tots[13];
  const int b = blockIdx.x;
  // every block finalises the (tiny) loss record itself; block 0 publishes it
  if (threadIdx.x < 13) tots[threadIdx.x] = loss_slot_total(acc, threadIdx.x);
  __syncthreads();
  if (threadIdx.x == 0) {
    finalize_losses(tots, B, fc.n67, smooth_count, weights, losses);
    if (b == 0) for (int i = 0; i < 12; ++i) losses_out[i] = losses[i];
  }
  __syncthreads();
  const DvertsIn in = {verts, nrows, target, contact, dx0, canon, weights, B, Bn};
  for (int u = threadIdx.x; u < fc.n; u += 256) {
    float gx, gy, gz;
    dverts_vertex(fc, in, losses, b, dverts_indices(fc, u), gx, gy, gz);
    float* o = dverts + ((size_t)b * fc.n + u) * 3;
    o[0] = gx; o[1] = gy; o[2] = gz;
  }
}

int dverts_assemble(const FitConst& fc, const float* verts, int nrows, const float* target, const float* contact,
                    const float* dx0, const float* canon, const float* weights, const double* acc, double smooth_count,
                    float* losses, int B, float* dverts, hipStream_t s, int Bn) {
  hipLaunchKernelGGL(dverts_assemble_kernel, dim3(B), dim3(256), 0, s, fc, verts, nrows, target, contact, dx0, canon, weights, acc, smooth_count, losses, B, Bn > 0 ? Bn : B, dverts);
  return (int)hipGetLastError();
}

// Prior gradients + Adam (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay).
// state[0] = step counter (as float bits of an int), incremented here.
// lr = step > lr_switch ? lr1 : lr0   with step counted from 0 (opt_amass_temp.py:349-352).
struct AdamGroup { float* p; const float* g; float* m; float* v; int n; };
// gradient of the L2 priors on "other" = [z 32 | hands 24] (opt_amass_temp.py:397-404)
__device__ __forceinline__ float adam_prior_grad(int col, float pv, const float* __restrict__ weights, int B) {
  return col < 32 ? weights[1] * 2.f * pv / ((float)B * 32.f) : weights[3] * 2.f * pv / ((float)B * 24.f);
}
// step-dependent scalars of one Adam update (common.hpp adam_coef_t: torch's own evaluation order) at the loop's lr level:
// lr = step > lr_switch ? lr1 : lr0 with step counted from 0 (opt_amass_temp.py:349-352; a third level for opt_amass_perframe.py)
__device__ __forceinline__ AdamCoef adam_coef(int step, double lr0, double lr1, int lr_switch, double lr2, int lr_switch2) {
  const double lr = (lr_switch2 > 0 && step > lr_switch2) ? lr2 : (step > lr_switch ? lr1 : lr0);
  return adam_coef_t(step + 1, lr);
}
// one element of torch.optim.Adam (shared by adam_kernel and fit_tail_kernel: the same arithmetic, bit for bit)
__device__ __forceinline__ void adam_update_one(float* p, float* mp, float* vp, float grad, AdamCoef c) {
  float pv = *p, m = *mp, v = *vp;
  adam_update_torch(pv, m, v, grad, c);
  *mp = m; *vp = v; *p = pv;
}
__global__ void __launch_bounds__(256)
adam_kernel(AdamGroup g0, AdamGroup g1, AdamGroup g2, int B, const float* __restrict__ weights, int* __restrict__ step_ctr,
            const int* __restrict__ step_cur, double lr0, double lr1, int lr_switch, double lr2, int lr_switch2,
            float* __restrict__ snap, int* __restrict__ nonfinite, const float* __restrict__ losses) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // 0-based index of this iteration, latched into step_cur at the start of the iteration (every thread
  // reads the latch; one thread advances the counter -> no read/write race, no extra launch)
  const int step = *step_cur;
  if (i == 0) *step_ctr = step + 1;
  // non-finite total loss (FittingMonitor.run_fitting, fitting_temp_slide.py:198-204): thread 0 records the first
  // offending iteration in nonfinite[0]; every thread reads nonfinite[1], the copy latched by the pose-stage kernel at
  // the START of this iteration (written one kernel boundary ago: no race), and skips its update once it is set
  bool frozen = false;
  if (nonfinite) {
    frozen = nonfinite[1] != 0;
    if (i == 0 && nonfinite[0] == 0) {
      const float tot = losses[6];
      if (!(fabsf(tot) <= 3.402823466e38f)) nonfinite[0] = step + 1;
    }
  }
  const int ntot = g0.n + g1.n + g2.n;
  if (i < ntot && snap) {
    const float* src = i < g0.n ? g0.p + i : (i < g0.n + g1.n ? g1.p + (i - g0.n) : g2.p + (i - g0.n - g1.n));
    snap[i] = *src;
  }
  if (i < ntot && !frozen) {
    AdamGroup G = g0; int k = i;
    if (k >= g0.n) { k -= g0.n; G = g1; if (k >= g1.n) { k -= g1.n; G = g2; } }
    float grad = G.g[k];
    if (G.p == g2.p) {                       // "other" = [z 32 | hands 24]: L2 priors (opt_amass_temp.py:397-404)
      const int col = k % 56;
      grad += adam_prior_grad(col, G.p[k], weights, B);
    }
    adam_update_one(G.p + k, G.m + k, G.v + k, grad, adam_coef(step, lr0, lr1, lr_switch, lr2, lr_switch2));
  }
}
int adam_step(float* transl, const float* g_transl, float* m0, float* v0, float* rot6d, const float* g_rot, float* m1,
              float* v1, float* other, const float* g_other, float* m2, float* v2, int B, const float* weights,
              int* step_ctr, const int* step_cur, float lr0, float lr1, int lr_switch, hipStream_t s, float lr2, int lr_switch2,
              float* snap, int* nonfinite, const float* losses) {
  AdamGroup a{transl, g_transl, m0, v0, B * 3}, b{rot6d, g_rot, m1, v1, B * 6}, c{other, g_other, m2, v2, B * 56};
  const int n = B * 65;
  if (nonfinite && !losses) return LEMO_ERR_ARG;
  hipLaunchKernelGGL(adam_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, c, B, weights, step_ctr, step_cur, lr_decimal(lr0),
                     lr_decimal(lr1), lr_switch, lr_decimal(lr2), lr_switch2, snap, nonfinite, losses);
  return (int)hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// Tail of one fitting iteration, one workgroup per frame (three launches -> one):
//   dz   = dh1 . W1                 the last layer of the VPoser decoder backward (vposer_smpl.py:107-115 transposed)
//   Adam on this frame's 65 parameters (adam_kernel above: same arithmetic, same snapshot / latch / counter protocol)
//   h1'  = lrelu(W1 z' + b1)        the FIRST layer of the NEXT iteration's decoder forward, on the updated latent
// Plain fp32 FMAs in a fixed order (K = 512 and K = 32 dot products of one frame: nothing for the matrix cores to do);
// the stand-alone first-layer launch that opens a graph runs this same kernel with do_dz = do_adam = 0, so h1 has the
// same bits whichever launch produced it (graph replay == eager launches).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
fit_tail_kernel(FitTail a) {
  __shared__ __attribute__((aligned(16))) float dh[512];
  __shared__ float zs[32], dzs[32];
  const int b = blockIdx.x, t = threadIdx.x, B = a.B;
  // ---- every global read of the kernel is issued here, before the first barrier, with clamped (never predicated)
  // addresses: the phases below depend on each other and would otherwise each begin with an exposed L2 round trip
  // (pose_kernels.hip has the measurements behind this rule).  Launch-uniform switches only select whether a value is used.
  const int m_dz = t >> 3, part = t & 7;
  float dh_a = 0.f, dh_b = 0.f;
  float4 wv[16];
  if (a.do_dz) {
    dh_a = a.dh1[(size_t)b * 512 + t];
    dh_b = a.dh1[(size_t)b * 512 + t + 256];
    const float* wr = a.w1t + (size_t)m_dz * 512 + 4 * part;
#pragma unroll
    for (int i = 0; i < 16; ++i) wv[i] = ld4(wr + 32 * i);
  }
  float4 w8[2][8];
  float bias[2] = {0.f, 0.f};
  if (a.h1) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float* wr = a.w1 + (size_t)(t + 256 * r) * 32;
#pragma unroll
      for (int i = 0; i < 8; ++i) w8[r][i] = ld4(wr + 4 * i);
      bias[r] = a.b1[t + 256 * r];
    }
  }
  // this thread's parameter (threads >= 65 shadow element 64: same addresses, nothing stored)
  const int te = t < 65 ? t : 64;
  float *p, *mp, *vp;
  const float* gp;
  int flat, col = -1;
  if (te < 3) { const int k = b * 3 + te; p = a.transl + k; mp = a.m0 + k; vp = a.v0 + k; gp = a.g_transl + k; flat = k; }
  else if (te < 9) { const int k = b * 6 + (te - 3); p = a.rot6d + k; mp = a.m1 + k; vp = a.v1 + k; gp = a.g_rot6d + k; flat = 3 * B + k; }
  else { col = te - 9; const int k = b * 56 + col; p = a.other + k; mp = a.m2 + k; vp = a.v2 + k; gp = a.g_other + k; flat = 9 * B + k; }
  float p_old = *p, m_old = 0.f, v_old = 0.f, g_in = 0.f, w_v = 0.f, w_h = 0.f, tot = 0.f;
  int step = 0, nf0 = 0, nf1 = 0;
  if (a.do_adam) {
    m_old = *mp; v_old = *vp; g_in = *gp;                         // (columns 0..31 of g_other are replaced by dz below)
    w_v = a.weights[1]; w_h = a.weights[3];
    step = *a.step_cur;
    if (a.nonfinite) { nf0 = a.nonfinite[0]; nf1 = a.nonfinite[1]; tot = a.losses[6]; }
  }
  // the two double-precision pow() of the bias corrections depend on the (scalar-loaded) step only: they run here, in the
  // shadow of the vector loads above
  AdamCoef coef{0.f, 1.f};        // (neg_step, bc2s)
  if (a.do_adam) coef = adam_coef(step, a.lr0, a.lr1, a.lr_switch, a.lr2, a.lr_switch2);
  LEMO_PIN(p_old); LEMO_PIN(m_old); LEMO_PIN(v_old); LEMO_PIN(g_in); LEMO_PIN(w_v); LEMO_PIN(w_h); LEMO_PIN(tot);
  LEMO_PIN(bias[0]); LEMO_PIN(bias[1]);
  if (a.do_dz) {
    // ---- dz[m] = sum_k w1t[m][k] dh1[b][k] : thread = (m = t >> 3, part = t & 7) takes k = 4 part + 32 i .. + 3
    dh[t] = dh_a;
    dh[t + 256] = dh_b;
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 d4 = ld4(&dh[4 * part + 32 * i]);
      acc = fmaf(wv[i].x, d4.x, acc); acc = fmaf(wv[i].y, d4.y, acc); acc = fmaf(wv[i].z, d4.z, acc); acc = fmaf(wv[i].w, d4.w, acc);
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    if (part == 0) { a.g_other[(size_t)b * 56 + m_dz] = acc; dzs[m_dz] = acc; }
  }
  __syncthreads();                              // dz of this frame is in dzs
  float p_new = p_old;
  if (a.do_adam) {
    if (b == 0 && t == 0) {
      *a.step_ctr = step + 1;
      // non-finite total loss: record the first offending iteration; updates stop from the NEXT iteration on (the
      // pose-stage kernel latches nonfinite[0] into nonfinite[1] at the start of every iteration)
      if (a.nonfinite && nf0 == 0 && !(fabsf(tot) <= 3.402823466e38f)) a.nonfinite[0] = step + 1;
    }
    if (t < 65) {
      float grad = g_in;
      if (col >= 0) {
        if (a.do_dz && col < 32) grad = dzs[col];
        grad += col < 32 ? w_v * 2.f * p_old / ((float)a.Bn * 32.f) : w_h * 2.f * p_old / ((float)a.Bn * 24.f);
      }
      if (a.snap) a.snap[flat] = p_old;
      if (nf1 == 0) {
        float mm = m_old, vv = v_old;
        adam_update_one(&p_new, &mm, &vv, grad, coef);
        *p = p_new; *mp = mm; *vp = vv;
      }
    }
  }
  if (!a.h1) return;
  if (t >= 9 && t < 41) zs[t - 9] = p_new;      // the latent of this frame, as updated above
  __syncthreads();
  // ---- h1[m] = lrelu(b1[m] + sum_k w1[m][k] z[k]) : thread t takes rows t and t + 256
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    float acc = bias[r];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc = fmaf(w8[r][i].x, zs[4 * i], acc); acc = fmaf(w8[r][i].y, zs[4 * i + 1], acc);
      acc = fmaf(w8[r][i].z, zs[4 * i + 2], acc); acc = fmaf(w8[r][i].w, zs[4 * i + 3], acc);
    }
    a.h1[(size_t)b * 512 + t + 256 * r] = lrelu(acc);
  }
}
int fit_tail(const FitTail& a, hipStream_t s) {
  if (a.B <= 0 || (a.do_dz && (!a.dh1 || !a.g_other || !a.w1t)) || (a.h1 && (!a.w1 || !a.b1 || !a.other))) return LEMO_ERR_ARG;
  if (a.do_adam && (!a.step_ctr || !a.step_cur || (a.nonfinite && !a.losses))) return LEMO_ERR_ARG;
  hipLaunchKernelGGL(fit_tail_kernel, dim3(a.B), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// optimiser-state hand-over (lemo_fit_load_state / lemo_fit_save_state and the PROX twins): up to STATE_MAX_JOBS flat float
// copies + the step counter in ONE launch, device to device, no host value involved (the call can be captured)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
state_copy_kernel(StateCopy a) {
  const int j = blockIdx.y;
  if (j < a.njobs) {
    const float* src = a.src[j];
    float* dst = a.dst[j];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n[j]; i += gridDim.x * blockDim.x) dst[i] = src[i];
  }
  if (blockIdx.x == 0 && j == 0 && threadIdx.x == 0) {
    if (a.step_dst && a.step_src) *a.step_dst = *a.step_src;
    if (a.nonfinite) { a.nonfinite[0] = 0; a.nonfinite[1] = 0; }       // a loaded state is a fresh run for the NaN / Inf latch
  }
}
int state_copy(const StateCopy& a, hipStream_t s) {
  if (a.njobs < 0 || a.njobs > STATE_MAX_JOBS) return LEMO_ERR_ARG;
  int nmax = 1;
  for (int j = 0; j < a.njobs; ++j) {
    if (!a.src[j] || !a.dst[j] || a.n[j] < 0) return LEMO_ERR_ARG;
    if (a.n[j] > nmax) nmax = a.n[j];
  }
  const int bx = (nmax + 255) / 256 < 64 ? (nmax + 255) / 256 : 64;
  hipLaunchKernelGGL(state_copy_kernel, dim3(bx, a.njobs ? a.njobs : 1), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace lemo
