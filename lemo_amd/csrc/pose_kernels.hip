// Per-frame "pose stage" of the fitting iteration: everything between the optimised parameters and
// the per-joint rigid transforms.  One workgroup per frame (the frames of a window are independent
// here), per-frame state staged in LDS.
//   * VPoser decoder            human_body_prior/train/vposer_smpl.py:107-121 (+49-62, 153-161)
//   * 6-D -> axis-angle         utils/utils.py:111-123
//   * SMPL-X pose assembly, Rodrigues, joint regression, kinematic chain
//                               smplx==0.1.26 SMPLX.forward ; human_body_prior/body_model/lbs.py:81-106,
//                               166-263 (vendored smplx.lbs)
// and the analytic backward of each.
#include "kernels.hpp"
#include "rot.hpp"

namespace lemo {

#define VP_H 512
#define VP_Z 32
#define VP_O 126
#define VP_NJ 21

// ------------------------------------------------------------------------------------------------
// VPoser.decode: z[32] -> lrelu(fc1) -> lrelu(fc2) -> out[126] -> 21 x (6D -> R -> aa)
// (dropout is identity in eval(), model_loader.py:70).  The three Linear layers run as fp32-MFMA
// GEMMs over all frames (gemm_kernels.hip); the per-joint rotation conversions are below.
// `o` is [B][128] (126 used; the out layer is packed to 128 rows of zeros-padded weights).
// ------------------------------------------------------------------------------------------------
__global__ void vposer_rot_fwd_kernel(const float* __restrict__ og, int B, float* __restrict__ matrot, float* __restrict__ aa) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * VP_NJ) return;
  const int b = idx / VP_NJ, t = idx - b * VP_NJ;
  float o6[6], R[9], a3[3];
  for (int i = 0; i < 6; ++i) o6[i] = og[(size_t)b * 128 + 6 * t + i];
  rot6d_fwd(o6, R);
  if (matrot) for (int i = 0; i < 9; ++i) matrot[((size_t)b * VP_NJ + t) * 9 + i] = R[i];
  if (aa) {
    rotmat_to_aa_fwd(R, a3);
    for (int i = 0; i < 3; ++i) aa[(size_t)b * 63 + 3 * t + i] = a3[i];
  }
}

int vposer_decode_fwd(const VPoserW& w, const float* z, int z_stride, int B, float* h1, float* h2, float* o,
                      float* matrot, float* aa, hipStream_t s) {
  if (B <= 0) return LEMO_ERR_SHAPE;
  int e;
  if ((e = gemm_nt16(w.w1, VP_Z, z, z_stride, VP_H, B, VP_Z, h1, VP_H, w.b1, nullptr, 0, 1, s))) return e;
  if ((e = gemm_nt16(w.w2, VP_H, h1, VP_H, VP_H, B, VP_H, h2, VP_H, w.b2, nullptr, 0, 1, s))) return e;
  if ((e = gemm_nt16(w.w3, VP_H, h2, VP_H, 128, B, VP_H, o, 128, w.b3, nullptr, 0, 2, s))) return e;
  const int n = B * VP_NJ;
  hipLaunchKernelGGL(vposer_rot_fwd_kernel, dim3((n + 63) / 64), dim3(64), 0, s, o, B, matrot, aa);
  return (int)hipGetLastError();
}

// d_aa / d_matrot -> d(out layer) [B][128]
__global__ void vposer_rot_bwd_kernel(const float* __restrict__ og, const float* __restrict__ d_aa,
                                      const float* __restrict__ d_matrot, int B, float* __restrict__ dout) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * 22) return;
  const int b = idx / 22, t = idx - b * 22;
  if (t == VP_NJ) { dout[(size_t)b * 128 + 126] = 0.f; dout[(size_t)b * 128 + 127] = 0.f; return; }
  float o6[6], R[9], dR[9], d6[6];
  for (int i = 0; i < 6; ++i) o6[i] = og[(size_t)b * 128 + 6 * t + i];
  rot6d_fwd(o6, R);
  for (int i = 0; i < 9; ++i) dR[i] = 0.f;
  if (d_aa) {
    float g[3];
    for (int i = 0; i < 3; ++i) g[i] = d_aa[(size_t)b * 63 + 3 * t + i];
    rotmat_to_aa_bwd(R, g, dR);
  }
  if (d_matrot) for (int i = 0; i < 9; ++i) dR[i] += d_matrot[((size_t)b * VP_NJ + t) * 9 + i];
  rot6d_bwd(o6, dR, d6);
  for (int i = 0; i < 6; ++i) dout[(size_t)b * 128 + 6 * t + i] = d6[i];
}

// scratch: dout [B][128], dh2 [B][512], dh1 [B][512]
int vposer_decode_bwd(const VPoserW& w, const float* h1, const float* h2, const float* o, const float* matrot,
                      const float* d_aa, const float* d_matrot, int B, float* dz, int dz_stride, float* scratch, hipStream_t s) {
  (void)matrot;
  if (B <= 0 || !scratch) return LEMO_ERR_SHAPE;
  float* dout = scratch;
  float* dh2 = scratch + (size_t)B * 128;
  float* dh1 = dh2 + (size_t)B * VP_H;
  const int n = B * 22;
  hipLaunchKernelGGL(vposer_rot_bwd_kernel, dim3((n + 63) / 64), dim3(64), 0, s, o, d_aa, d_matrot, B, dout);
  int e = (int)hipGetLastError();
  if (e) return e;
  (void)dh2; (void)dh1;
  return vposer_mlp_bwd(w, h1, h2, B, dz, dz_stride, scratch, s);
}

// dout = scratch[0 .. B*128) ; dh2 = (dout . W3) * lrelu'(h2) ; dh1 = (dh2 . W2) * lrelu'(h1) ; dz = dh1 . W1
int vposer_mlp_bwd(const VPoserW& w, const float* h1, const float* h2, int B, float* dz, int dz_stride, float* scratch,
                   hipStream_t s) {
  if (B <= 0 || !scratch) return LEMO_ERR_SHAPE;
  float* dout = scratch;
  float* dh2 = scratch + (size_t)B * 128;
  float* dh1 = dh2 + (size_t)B * VP_H;
  int e;
  if ((e = gemm_nt16(w.w3t, 128, dout, 128, VP_H, B, 128, dh2, VP_H, nullptr, h2, VP_H, 3, s))) return e;
  if ((e = gemm_nt16(w.w2t, VP_H, dh2, VP_H, VP_H, B, VP_H, dh1, VP_H, nullptr, h1, VP_H, 3, s))) return e;
  return gemm_nt16(w.w1t, VP_H, dh1, VP_H, VP_Z, B, VP_H, dz, dz_stride, nullptr, nullptr, 0, 0, s);
}

// ------------------------------------------------------------------------------------------------
// 6-D -> axis-angle rows (convert_to_3D_rot, utils/utils.py:111-123)
// ------------------------------------------------------------------------------------------------
__global__ void rot6d_to_aa_fwd_kernel(const float* __restrict__ x6, int stride, int N, float* __restrict__ aa) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float x[6], R[9], a[3];
  for (int k = 0; k < 6; ++k) x[k] = x6[(size_t)i * stride + k];
  rot6d_fwd(x, R);
  rotmat_to_aa_fwd(R, a);
  for (int k = 0; k < 3; ++k) aa[(size_t)i * 3 + k] = a[k];
}
__global__ void rot6d_to_aa_bwd_kernel(const float* __restrict__ x6, int stride, const float* __restrict__ d_aa,
                                       int N, float* __restrict__ dx6) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float x[6], R[9], dR[9], g[3], d6[6];
  for (int k = 0; k < 6; ++k) x[k] = x6[(size_t)i * stride + k];
  for (int k = 0; k < 3; ++k) g[k] = d_aa[(size_t)i * 3 + k];
  rot6d_fwd(x, R);
  rotmat_to_aa_bwd(R, g, dR);
  rot6d_bwd(x, dR, d6);
  for (int k = 0; k < 6; ++k) dx6[(size_t)i * 6 + k] = d6[k];
}
int rot6d_to_aa_fwd(const float* x6, int stride, int N, float* aa, hipStream_t s) {
  hipLaunchKernelGGL(rot6d_to_aa_fwd_kernel, dim3((N + 63) / 64), dim3(64), 0, s, x6, stride, N, aa);
  return (int)hipGetLastError();
}
int rot6d_to_aa_bwd(const float* x6, int stride, const float* d_aa, int N, float* dx6, hipStream_t s) {
  hipLaunchKernelGGL(rot6d_to_aa_bwd_kernel, dim3((N + 63) / 64), dim3(64), 0, s, x6, stride, d_aa, N, dx6);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// SMPL-X pose stage forward (one block of 256 threads per frame)
// ------------------------------------------------------------------------------------------------
#define MAXJ 64

__global__ void __launch_bounds__(256)
smplx_pose_fwd_kernel(BodyConst c, PoseIn in, PoseWs ws) {
  __shared__ float fp[MAXJ * 3];
  __shared__ float Rs[MAXJ * 9];
  __shared__ float Js[MAXJ * 3];
  __shared__ float Ts[MAXJ * 12];
  __shared__ float shp[32];
  __shared__ float gob[66];           // global_orient (3) | body_pose (63) of this frame
  const int b = blockIdx.x, t = threadIdx.x;
  const int nj = c.nj, np = nj * 3;
  // ---- per-iteration bookkeeping of the fitting engine (block 0 only)
  if (b == 0) {
    if (in.zero_f64) for (int i = t; i < in.n_zero; i += 256) in.zero_f64[i] = 0.0;
    if (in.step_cur && t == 0) *in.step_cur = *in.step_ctr;
  }
  // ---- global_orient / body_pose: given, or derived from the 6-D rotation / VPoser out layer
  if (in.rot6d) {
    if (t == 0) {
      float x6[6], R[9], a3[3];
      for (int k = 0; k < 6; ++k) x6[k] = in.rot6d[(size_t)b * 6 + k];
      rot6d_fwd(x6, R);
      rotmat_to_aa_fwd(R, a3);
      for (int k = 0; k < 3; ++k) { gob[k] = a3[k]; if (in.go_out) in.go_out[(size_t)b * 3 + k] = a3[k]; }
    }
  } else if (t < 3) gob[t] = in.global_orient[(size_t)b * 3 + t];
  if (in.vposer_o) {
    if (t >= 64 && t < 64 + VP_NJ) {
      const int jn = t - 64;
      float o6[6], R[9], a3[3];
      for (int k = 0; k < 6; ++k) o6[k] = in.vposer_o[(size_t)b * 128 + 6 * jn + k];
      rot6d_fwd(o6, R);
      rotmat_to_aa_fwd(R, a3);
      for (int k = 0; k < 3; ++k) gob[3 + 3 * jn + k] = a3[k];
    }
  } else if (t >= 64 && t < 64 + 63) gob[3 + t - 64] = in.body_pose[(size_t)b * 63 + (t - 64)];
  __syncthreads();
  // ---- full pose (SMPLX.forward: cat[go, body, jaw, leye, reye, lhand45, rhand45] + pose_mean)
  for (int i = t; i < np; i += 256) {
    float v;
    if (i < 66) v = gob[i];
    else if (i < 69) v = in.jaw ? in.jaw[(size_t)b * 3 + (i - 66)] : 0.f;
    else if (i < 72) v = in.leye ? in.leye[(size_t)b * 3 + (i - 69)] : 0.f;
    else if (i < 75) v = in.reye ? in.reye[(size_t)b * 3 + (i - 72)] : 0.f;
    else {
      const int hidx = i - 75, side = hidx / 45, cc = hidx - side * 45;
      const float* hp = (side == 0 ? in.lh : in.rh) + (size_t)b * in.hand_stride;
      if (c.ncomp > 0) {
        const float* comp = side == 0 ? c.lh_comp : c.rh_comp;
        float a = 0.f;
        for (int k = 0; k < c.ncomp; ++k) a = fmaf(hp[k], comp[k * 45 + cc], a);
        v = a;
      } else v = hp[cc];
    }
    v += c.pose_mean[i];
    fp[i] = v;
    ws.full_pose[(size_t)b * np + i] = v;
  }
  if (t < c.nshape) {
    const int nb = c.nshape / 2;
    shp[t] = t < nb ? in.betas[(size_t)b * in.betas_stride + t] : (in.expr ? in.expr[(size_t)b * nb + (t - nb)] : 0.f);
  }
  __syncthreads();
  // ---- Rodrigues
  if (t < nj) {
    float R[9];
    rodrigues_fwd(&fp[3 * t], R);
    for (int i = 0; i < 9; ++i) { Rs[9 * t + i] = R[i]; ws.R[((size_t)b * nj + t) * 9 + i] = R[i]; }
  }
  // ---- rest joints  J = J_template + J_dirs . shape
  for (int i = t; i < np; i += 256) {
    float a = c.J_template[i];
    for (int k = 0; k < c.nshape; ++k) a = fmaf(c.J_dirs[(size_t)i * c.nshape + k], shp[k], a);
    Js[i] = a;
    ws.J[(size_t)b * np + i] = a;
  }
  __syncthreads();
  // ---- GEMM features, KG8 layout: Xg[k>>3][b][k&7]; k < nshape: shape coefs; then (R[1:] - I)
  {
    const int nfeat = c.nshape + (nj - 1) * 9;
    for (int k = t; k < nfeat; k += 256) {
      float v;
      if (k < c.nshape) v = shp[k];
      else {
        const int f = k - c.nshape, e = f % 9;
        v = Rs[9 + f] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
      }
      ws.Xg[((size_t)(k >> 3) * ws.Bp + b) * 8 + (k & 7)] = v;
    }
  }
  // ---- kinematic chain, level-synchronous: T[i] = T[parent] * [R_i | J_i - J_parent]
  for (int lev = 0; lev < c.nlev; ++lev) {
    const int s0 = c.level_start[lev], s1 = c.level_start[lev + 1];
    for (int w = t; w < (s1 - s0) * 12; w += 256) {
      const int i = c.level_joints[s0 + w / 12], e = w % 12, r = e >> 2, cc = e & 3;
      const int p = c.parents[i];
      float v;
      if (p < 0) v = cc < 3 ? Rs[9 * i + 3 * r + cc] : Js[3 * i + r];
      else {
        const float* Tp = &Ts[12 * p + 4 * r];
        if (cc < 3) v = Tp[0] * Rs[9 * i + cc] + Tp[1] * Rs[9 * i + 3 + cc] + Tp[2] * Rs[9 * i + 6 + cc];
        else v = Tp[0] * (Js[3 * i] - Js[3 * p]) + Tp[1] * (Js[3 * i + 1] - Js[3 * p + 1]) +
                 Tp[2] * (Js[3 * i + 2] - Js[3 * p + 2]) + Tp[3];
      }
      Ts[12 * i + e] = v;
    }
    __syncthreads();
  }
  // ---- relative transforms A = [T_R | T_t - T_R J], posed joints
  for (int w = t; w < nj * 12; w += 256) {
    const int i = w / 12, e = w % 12, r = e >> 2, cc = e & 3;
    const float* Ti = &Ts[12 * i + 4 * r];
    float v = Ti[cc];
    if (cc == 3) {
      v -= Ti[0] * Js[3 * i] + Ti[1] * Js[3 * i + 1] + Ti[2] * Js[3 * i + 2];
      ws.Jtr[((size_t)b * nj + i) * 3 + r] = Ti[3];
    }
    ws.A[((size_t)b * nj + i) * 12 + e] = v;
    ws.T[((size_t)b * nj + i) * 12 + e] = Ti[cc];
  }
}

int smplx_pose_fwd(const BodyConst& c, const PoseIn& in, const PoseWs& ws, int B, hipStream_t s) {
  if (c.nj > MAXJ || c.nshape > 32 || c.ncomp > 45 || B <= 0 || B > ws.Bp) return LEMO_ERR_SHAPE;
  if (c.nshape + (c.nj - 1) * 9 > 512) return LEMO_ERR_SHAPE;
  hipLaunchKernelGGL(smplx_pose_fwd_kernel, dim3(B), dim3(256), 0, s, c, in, ws);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// SMPL-X pose stage backward
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
smplx_pose_bwd_kernel(BodyConst c, PoseWs ws, PoseGradIn gi, PoseGradOut go) {
  __shared__ float G[MAXJ * 12];      // d(global transform) [dT_R | dT_t]
  __shared__ float Rs[MAXJ * 9];
  __shared__ float Ts[MAXJ * 12];
  __shared__ float Js[MAXJ * 3];
  __shared__ float dJ[MAXJ * 3];
  __shared__ float drel[MAXJ * 3];
  __shared__ float dRl[MAXJ * 9];
  __shared__ float dfp[MAXJ * 3];
  // tree tables and hand PCA components: staged once (coalesced, independent loads) instead of being chased
  // element by element through L2 inside the level loops / the 45-term component sums
  __shared__ int cstart[MAXJ + 1], clist[MAXJ], par[MAXJ];
  __shared__ float hcomp[2 * 45 * 45];
  const int b = blockIdx.x, t = threadIdx.x;
  const int nj = c.nj, np = nj * 3;
  if (t <= nj) cstart[t] = c.child_start[t];
  if (t < nj) { par[t] = c.parents[t]; if (t < nj - 1) clist[t] = c.child_list[t]; }
  if (c.ncomp > 0) for (int i = t; i < c.ncomp * 45; i += 256) { hcomp[i] = c.lh_comp[i]; hcomp[45 * 45 + i] = c.rh_comp[i]; }
  for (int i = t; i < nj * 9; i += 256) Rs[i] = ws.R[(size_t)b * nj * 9 + i];
  for (int i = t; i < nj * 12; i += 256) Ts[i] = ws.T[(size_t)b * nj * 12 + i];
  for (int i = t; i < np; i += 256) Js[i] = ws.J[(size_t)b * np + i];
  __syncthreads();
  // own terms: A_R = T_R ; A_t = T_t - T_R J ; Jtr = T_t
  for (int w = t; w < nj * 12; w += 256) {
    const int i = w / 12, e = w % 12, r = e >> 2, cc = e & 3;
    const float* dAi = gi.dA + ((size_t)b * nj + i) * 12;
    float v;
    if (cc < 3) v = dAi[4 * r + cc] - dAi[4 * r + 3] * Js[3 * i + cc];
    else v = dAi[4 * r + 3] + (gi.dJtr ? gi.dJtr[((size_t)b * nj + i) * 3 + r] : 0.f);
    G[w] = v;
  }
  for (int w = t; w < np; w += 256) {          // dJ_i = -T_R^T dA_t
    const int i = w / 3, cc = w % 3;
    const float* dAi = gi.dA + ((size_t)b * nj + i) * 12;
    dJ[w] = -(Ts[12 * i + cc] * dAi[3] + Ts[12 * i + 4 + cc] * dAi[7] + Ts[12 * i + 8 + cc] * dAi[11]);
  }
  __syncthreads();
  // reverse levels: G[p] += sum_children [G_R[ch] R_ch^T + G_t[ch] (x) rel_ch | G_t[ch]]
  for (int lev = c.nlev - 2; lev >= 0; --lev) {
    const int s0 = c.level_start[lev], s1 = c.level_start[lev + 1];
    for (int w = t; w < (s1 - s0) * 12; w += 256) {
      const int i = c.level_joints[s0 + w / 12], e = w % 12, r = e >> 2, cc = e & 3;
      float acc = 0.f;
      for (int q = cstart[i]; q < cstart[i + 1]; ++q) {
        const int ch = clist[q];
        const float* Gc = &G[12 * ch + 4 * r];
        if (cc < 3) {
          acc += Gc[0] * Rs[9 * ch + 3 * cc] + Gc[1] * Rs[9 * ch + 3 * cc + 1] + Gc[2] * Rs[9 * ch + 3 * cc + 2] +
                 Gc[3] * (Js[3 * ch + cc] - Js[3 * i + cc]);
        } else acc += Gc[3];
      }
      G[12 * i + e] += acc;
    }
    __syncthreads();
  }
  // local grads: dR_i = T_R[p]^T G_R[i] ; drel_i = T_R[p]^T G_t[i]
  for (int w = t; w < nj * 12; w += 256) {
    const int i = w / 12, e = w % 12, r = e >> 2, cc = e & 3;     // (r,cc): element of dR_i (cc<3) or drel (cc==3 -> comp r)
    const int p = par[i];
    float v;
    if (p < 0) v = G[12 * i + 4 * r + cc];
    else v = Ts[12 * p + r] * G[12 * i + cc] + Ts[12 * p + 4 + r] * G[12 * i + 4 + cc] + Ts[12 * p + 8 + r] * G[12 * i + 8 + cc];
    if (cc < 3) {
      float d = v;
      if (i >= 1 && gi.dX) d += gi.dX[(size_t)b * 512 + c.nshape + (i - 1) * 9 + 3 * r + cc];
      dRl[9 * i + 3 * r + cc] = d;
    } else drel[3 * i + r] = v;
  }
  __syncthreads();
  // dJ: J_i enters rel_i (+) and rel_children (-)
  for (int w = t; w < np; w += 256) {
    const int i = w / 3, cc = w % 3;
    float v = dJ[w] + drel[w];
    for (int q = cstart[i]; q < cstart[i + 1]; ++q) v -= drel[3 * clist[q] + cc];
    dJ[w] = v;
  }
  // Rodrigues backward
  if (t < nj) {
    float fp3[3], d3[3];
    for (int k = 0; k < 3; ++k) fp3[k] = ws.full_pose[(size_t)b * np + 3 * t + k];
    rodrigues_bwd(fp3, &dRl[9 * t], d3);
    for (int k = 0; k < 3; ++k) dfp[3 * t + k] = d3[k];
  }
  __syncthreads();
  // fused consumers: d(global_orient) -> d(rot6d) ; d(body_pose) -> d(VPoser out layer)
  if (go.d_rot6d && t == 0) {
    float x6[6], R[9], dR[9], d6[6];
    for (int k = 0; k < 6; ++k) x6[k] = go.rot6d[(size_t)b * 6 + k];
    rot6d_fwd(x6, R);
    rotmat_to_aa_bwd(R, &dfp[0], dR);
    rot6d_bwd(x6, dR, d6);
    for (int k = 0; k < 6; ++k) go.d_rot6d[(size_t)b * 6 + k] = d6[k];
  }
  if (go.d_vposer_o && t >= 64 && t < 64 + VP_NJ + 1) {
    const int jn = t - 64;
    if (jn == VP_NJ) { go.d_vposer_o[(size_t)b * 128 + 126] = 0.f; go.d_vposer_o[(size_t)b * 128 + 127] = 0.f; }
    else {
      float o6[6], R[9], dR[9], d6[6];
      for (int k = 0; k < 6; ++k) o6[k] = go.vposer_o[(size_t)b * 128 + 6 * jn + k];
      rot6d_fwd(o6, R);
      rotmat_to_aa_bwd(R, &dfp[3 + 3 * jn], dR);
      rot6d_bwd(o6, dR, d6);
      for (int k = 0; k < 6; ++k) go.d_vposer_o[(size_t)b * 128 + 6 * jn + k] = d6[k];
    }
  }
  // scatter d(full_pose)
  for (int i = t; i < 75; i += 256) {
    const float v = dfp[i];
    if (i < 3) { if (go.d_global_orient) go.d_global_orient[(size_t)b * 3 + i] = v; }
    else if (i < 66) { if (go.d_body_pose) go.d_body_pose[(size_t)b * 63 + (i - 3)] = v; }
    else if (i < 69) { if (go.d_jaw) go.d_jaw[(size_t)b * 3 + (i - 66)] = v; }
    else if (i < 72) { if (go.d_leye) go.d_leye[(size_t)b * 3 + (i - 69)] = v; }
    else { if (go.d_reye) go.d_reye[(size_t)b * 3 + (i - 72)] = v; }
  }
  {
    const int nh = c.ncomp > 0 ? c.ncomp : 45;
    for (int w = t; w < 2 * nh; w += 256) {
      const int side = w / nh, k = w - side * nh;
      float* dst = side == 0 ? go.d_lh : go.d_rh;
      if (!dst) continue;
      float v;
      if (c.ncomp > 0) {
        const float* comp = hcomp + side * 45 * 45;
        v = 0.f;
        for (int cc = 0; cc < 45; ++cc) v = fmaf(comp[k * 45 + cc], dfp[75 + side * 45 + cc], v);
      } else v = dfp[75 + side * 45 + k];
      dst[(size_t)b * go.hand_stride + k] = v;
    }
  }
  // d(shape coefs) = dX[:nshape] + J_dirs^T dJ
  if (t < c.nshape && (go.d_betas || go.d_expr)) {
    float v = gi.dX ? gi.dX[(size_t)b * 512 + t] : 0.f;
    for (int i = 0; i < np; ++i) v = fmaf(c.J_dirs[(size_t)i * c.nshape + t], dJ[i], v);
    const int nb = c.nshape / 2;
    if (t < nb) { if (go.d_betas) go.d_betas[(size_t)b * nb + t] = v; }
    else { if (go.d_expr) go.d_expr[(size_t)b * nb + (t - nb)] = v; }
  }
}

int smplx_pose_bwd(const BodyConst& c, const PoseWs& ws, const PoseGradIn& gi, const PoseGradOut& go, int B, hipStream_t s) {
  if (c.nj > MAXJ || B <= 0) return LEMO_ERR_SHAPE;
  hipLaunchKernelGGL(smplx_pose_bwd_kernel, dim3(B), dim3(256), 0, s, c, ws, gi, go);
  return (int)hipGetLastError();
}

}  // namespace lemo
