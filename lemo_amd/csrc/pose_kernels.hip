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
  float* dout = vposer_scratch_dout(scratch, B);
  const int n = B * 22;
  hipLaunchKernelGGL(vposer_rot_bwd_kernel, dim3((n + 63) / 64), dim3(64), 0, s, o, d_aa, d_matrot, B, dout);
  int e = (int)hipGetLastError();
  if (e) return e;
  return vposer_mlp_bwd(w, h1, h2, B, dz, dz_stride, scratch, s);
}

// dout = scratch[0 .. B*128) ; dh2 = (dout . W3) * lrelu'(h2) ; dh1 = (dh2 . W2) * lrelu'(h1) ; dz = dh1 . W1
int vposer_mlp_bwd(const VPoserW& w, const float* h1, const float* h2, int B, float* dz, int dz_stride, float* scratch,
                   hipStream_t s) {
  if (B <= 0 || !scratch) return LEMO_ERR_SHAPE;
  static_assert(VP_H == VP_HIDDEN, "scratch layout helpers (kernels.hpp)");
  float* dout = vposer_scratch_dout(scratch, B);
  float* dh2 = vposer_scratch_dh2(scratch, B);
  float* dh1 = vposer_scratch_dh1(scratch, B);
  int e;
  if ((e = gemm_nt16(w.w3t, 128, dout, 128, VP_H, B, 128, dh2, VP_H, nullptr, h2, VP_H, 3, s))) return e;
  if ((e = gemm_nt16(w.w2t, VP_H, dh2, VP_H, VP_H, B, VP_H, dh1, VP_H, nullptr, h1, VP_H, 3, s))) return e;
  if (!dz) return 0;                     // the fitting engine computes the last layer in its fused tail launch (fit_tail)
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
  CENSUS_DECL(0)
  CENSUS()
  __shared__ float fp[MAXJ * 3];
  __shared__ float Rs[MAXJ * 9];
  __shared__ float Js[MAXJ * 3];
  __shared__ float Ts[MAXJ * 12];
  __shared__ float shp[32];
  __shared__ float gob[66];           // global_orient (3) | body_pose (63) of this frame
  __shared__ float Jd[MAXJ * 3 * 32];
  __shared__ int par[MAXJ];
  const int b = blockIdx.x, t = threadIdx.x;
  const int nj = c.nj, np = nj * 3;
  if (t >= 192 && t - 192 < nj) par[t - 192] = c.parents[t - 192];
  // ---- per-iteration bookkeeping of the fitting engine (block 0 only)
  if (b == 0) {
    if (in.zero_f64) for (int i = t; i < in.n_zero; i += 256) in.zero_f64[i] = 0.0;
    if (in.step_cur && t == 0) *in.step_cur = *in.step_ctr;
    if (in.nonfinite && t == 1) in.nonfinite[1] = in.nonfinite[0];
  }
  // ---- every global read of the kernel is issued here, before the first barrier, with clamped (never predicated)
  // addresses; the phases below only touch registers and LDS.  (Read where they were consumed, three phases in a
  // row began with an exposed L2 round trip: ~12 k of the kernel's 20 k cycles.)
  float r6[6], o6[6];                 // operands of the rotation conversions first: they gate the longest chain
  if (in.rot6d) for (int k = 0; k < 6; ++k) r6[k] = in.rot6d[(size_t)b * 6 + k];
  const int jn = min(max(t - 64, 0), VP_NJ - 1);
  if (in.vposer_o) for (int k = 0; k < 6; ++k) o6[k] = in.vposer_o[(size_t)b * 128 + 6 * jn + k];
  const int ic = min(t, np - 1);      // thread t owns full_pose[t] (np <= 192)
  float r_own;                        // its entry of [global_orient | body_pose | jaw | leye | reye] when read directly
  {
    const float* q = nullptr;
    if (ic < 3) q = in.rot6d ? nullptr : in.global_orient + (size_t)b * 3 + ic;
    else if (ic < 66) q = in.vposer_o ? nullptr : in.body_pose + (size_t)b * 63 + (ic - 3);
    else if (ic < 69) q = in.jaw ? in.jaw + (size_t)b * 3 + (ic - 66) : nullptr;
    else if (ic < 72) q = in.leye ? in.leye + (size_t)b * 3 + (ic - 69) : nullptr;
    else if (ic < 75) q = in.reye ? in.reye + (size_t)b * 3 + (ic - 72) : nullptr;
    r_own = q ? *q : 0.f;
  }
  const float r_pm = c.pose_mean[ic];
  float r_shp = 0.f;
  if (t < c.nshape) {
    const int nb = c.nshape / 2;
    r_shp = t < nb ? in.betas[(size_t)b * in.betas_stride + t] : (in.expr ? in.expr[(size_t)b * nb + (t - nb)] : 0.f);
  }
  const float rJt = c.J_template[ic];
  // J_dirs [np][nshape]: read coalesced (a row per lane is one cache line per lane and instruction) and staged flat
  float rJd[24];
  const int njd = np * c.nshape;      // <= 192 * 32
#pragma unroll
  for (int m = 0; m < 24; ++m) rJd[m] = c.J_dirs[min(t + 256 * m, njd - 1)];
  // ---- global_orient / body_pose derived from the 6-D rotation / VPoser out layer: every lane runs the conversion
  // (same cost as one lane); LEMO_PIN keeps the loads above from being sunk into these branches
  // (the pins sit after every load has been issued: an asm operand is a use, and waits for its load)
  if (in.rot6d) for (int k = 0; k < 6; ++k) LEMO_PIN(r6[k]);
  if (in.vposer_o) for (int k = 0; k < 6; ++k) LEMO_PIN(o6[k]);
  if (in.rot6d && (t >> 6) == 2) {    // wave 2 (wave-uniform branch): the two conversions run side by side
    float R[9], a3[3];
    rot6d_fwd(r6, R);
    rotmat_to_aa_fwd(R, a3);
    if (t == 128) for (int k = 0; k < 3; ++k) { gob[k] = a3[k]; if (in.go_out) in.go_out[(size_t)b * 3 + k] = a3[k]; }
  }
  if (in.vposer_o && (t >> 6) == 1) {
    float R[9], a3[3];
    rot6d_fwd(o6, R);
    rotmat_to_aa_fwd(R, a3);
    if (t < 64 + VP_NJ) for (int k = 0; k < 3; ++k) gob[3 + 3 * jn + k] = a3[k];
  }
  // ---- hand pose of lane t = 75 + 45 side + cc (PCA components -> 45 axis-angle values per hand)
  float vh = 0.f;
  if (np > 75) {
    const int hidx = min(max(t - 75, 0), 89), side = hidx >= 45 ? 1 : 0, cc = hidx - 45 * side;
    const float* hp = (side == 0 ? in.lh : in.rh) + (size_t)b * in.hand_stride;
    if (c.ncomp > 0) {
      const float* comp = side == 0 ? c.lh_comp : c.rh_comp;
      for (int k0 = 0; k0 < c.ncomp; k0 += 12) {
        float h[12], cm[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) { const int kk = min(k0 + k, c.ncomp - 1); h[k] = hp[kk]; cm[k] = comp[kk * 45 + cc]; }
#pragma unroll
        for (int k = 0; k < 12; ++k) vh = fmaf(k0 + k < c.ncomp ? h[k] : 0.f, cm[k], vh);
      }
    } else vh = hp[cc];
  }
  if (t < c.nshape) shp[t] = r_shp;
#pragma unroll
  for (int m = 0; m < 24; ++m) if (t + 256 * m < njd) Jd[t + 256 * m] = rJd[m];
  __syncthreads(); CENSUS()
  // ---- full pose (SMPLX.forward: cat[go, body, jaw, leye, reye, lhand45, rhand45] + pose_mean)
  if (t < np) {
    float v;
    if (t < 3) v = in.rot6d ? gob[t] : r_own;
    else if (t < 66) v = in.vposer_o ? gob[t] : r_own;
    else if (t < 75) v = r_own;
    else v = vh;
    v += r_pm;
    fp[t] = v;
    ws.full_pose[(size_t)b * np + t] = v;
  }
  __syncthreads(); CENSUS()
  // ---- Rodrigues
  if (t < nj) {
    float R[9];
    rodrigues_fwd(&fp[3 * t], R);
    for (int i = 0; i < 9; ++i) { Rs[9 * t + i] = R[i]; ws.R[((size_t)b * nj + t) * 9 + i] = R[i]; }
  }
  // ---- rest joints  J = J_template + J_dirs . shape
  if (t < np) {
    float a = rJt;
    for (int k = 0; k < c.nshape; ++k) a = fmaf(Jd[t * c.nshape + k], shp[k], a);
    Js[t] = a;
    ws.J[(size_t)b * np + t] = a;
  }
  __syncthreads(); CENSUS()
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
      if (ws.XgS && ws.xgs_f16) {
        // the same value as two fp16 pieces (hi = f16(v), lo = f16(v - hi): 22 significand bits; |v| <= ~3, no scaling needed),
        // same fragment order with two planes per chunk: XgS[k >> 4][piece][frame][(k >> 3) & 1][k & 7]
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        _Float16* q = reinterpret_cast<_Float16*>(ws.XgS) + ((((size_t)(k >> 4) * 2) * ws.Bp + b) * 2 + ((k >> 3) & 1)) * 8 + (k & 7);
        q[0] = hi; q[(size_t)ws.Bp * 16] = lo;
      } else if (ws.XgS) {
        // the same value as its three exact bf16 pieces, in the B-fragment order of the blend GEMM of lbs_verts_fwd
        // (chunk = 16 features, lane half = 8-feature group parity): XgS[k >> 4][piece][frame][(k >> 3) & 1][k & 7]
        const __bf16 hi = (__bf16)v;
        float r = v - (float)hi;
        const __bf16 mid = (__bf16)r;
        r -= (float)mid;
        const __bf16 lo = (__bf16)r;
        __bf16* q = reinterpret_cast<__bf16*>(ws.XgS) + ((((size_t)(k >> 4) * 3) * ws.Bp + b) * 2 + ((k >> 3) & 1)) * 8 + (k & 7);
        const size_t ps = (size_t)ws.Bp * 16;
        q[0] = hi; q[ps] = mid; q[2 * ps] = lo;
      }
    }
  }
  // ---- kinematic chain: T[i] = L[root] * ... * L[parent(i)] * L[i],  L[j] = [R_j | J_j - J_parent(j)]
  // Four threads per joint (one column of T each) walk up the ancestor list and left-multiply as they go.  The
  // level-synchronous form (T[i] = T[parent] * L[i], one barrier per tree level) spent 11 x 1250 cycles on barriers
  // and dependent table loads for ~40 FMAs of work per thread; the walk is <= 10 steps of one LDS round trip each.
  if (t < nj * 4) {
    const int i = t >> 2, cc = t & 3;
    int p = par[i];
    float x, y, z;
    if (cc < 3) { x = Rs[9 * i + cc]; y = Rs[9 * i + 3 + cc]; z = Rs[9 * i + 6 + cc]; }
    else {
      x = Js[3 * i]; y = Js[3 * i + 1]; z = Js[3 * i + 2];
      if (p >= 0) { x -= Js[3 * p]; y -= Js[3 * p + 1]; z -= Js[3 * p + 2]; }
    }
    const float tsel = cc == 3 ? 1.f : 0.f;
    while (p >= 0) {                    // every lane reads the translation too (x 0 for the rotation columns): a
      const int pp = par[p], pq = max(pp, 0);   // branch on the column would put a second LDS round trip in each step
      const float* R = &Rs[9 * p];
      const float rs = pp >= 0 ? tsel : 0.f;
      const float tx = tsel * Js[3 * p] - rs * Js[3 * pq], ty = tsel * Js[3 * p + 1] - rs * Js[3 * pq + 1],
                  tz = tsel * Js[3 * p + 2] - rs * Js[3 * pq + 2];
      const float nx = R[0] * x + R[1] * y + R[2] * z + tx;
      const float ny = R[3] * x + R[4] * y + R[5] * z + ty;
      const float nz = R[6] * x + R[7] * y + R[8] * z + tz;
      x = nx; y = ny; z = nz;
      p = pp;
    }
    Ts[12 * i + cc] = x; Ts[12 * i + 4 + cc] = y; Ts[12 * i + 8 + cc] = z;
  }
  __syncthreads(); CENSUS()
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
  CENSUS()
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
  CENSUS_DECL(1)
  CENSUS()
  __shared__ __attribute__((aligned(16))) float G[MAXJ * 12];      // d(global transform) [dT_R | dT_t]
  __shared__ __attribute__((aligned(16))) float Cb[MAXJ * 12];     // what each joint adds to its parent's G
  __shared__ float Rs[MAXJ * 9];
  __shared__ float Ts[MAXJ * 12];
  __shared__ float Js[MAXJ * 3];
  __shared__ float dJ[MAXJ * 3];
  __shared__ float drel[MAXJ * 3];
  __shared__ float dRl[MAXJ * 9];
  __shared__ float dfp[MAXJ * 3];
  __shared__ float dAs[MAXJ * 12], dJt[MAXJ * 3], fps[MAXJ * 3], dXs[512], o6s[128], x6s[8], part[8 * 32];
  // tree tables and hand PCA components: staged once (coalesced, independent loads) instead of being chased
  // element by element through L2 inside the level loops / the 45-term component sums
  __shared__ int cstart[MAXJ + 1], clist[MAXJ], par[MAXJ], lvl[MAXJ];
  __shared__ float hcomp[2 * 45 * 45];
  const int b = blockIdx.x, t = threadIdx.x;
  const int nj = c.nj, np = nj * 3, n9 = nj * 9, n12 = nj * 12;
  // ---- every global read of the kernel is issued here, before the first barrier, with clamped (never predicated)
  // addresses: the loads of one thread are all in flight together and the later phases only touch LDS.  (Issued
  // where they were consumed, each phase started with an exposed L2 round trip: 37 k cycles per block, of which
  // ~15 k were load latency.)
  float rR[3], rT[3], rA[3], rX[2], rJd[24];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rR[k] = ws.R[(size_t)b * n9 + min(t + 256 * k, n9 - 1)];
    rT[k] = ws.T[(size_t)b * n12 + min(t + 256 * k, n12 - 1)];
    rA[k] = gi.dA[(size_t)b * n12 + min(t + 256 * k, n12 - 1)];
  }
  const int tc = min(t, np - 1);
  const float rJ = ws.J[(size_t)b * np + tc], rF = ws.full_pose[(size_t)b * np + tc];
  const float rJt = gi.dJtr ? gi.dJtr[(size_t)b * np + tc] : 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) rX[k] = gi.dX ? gi.dX[(size_t)b * 512 + t + 256 * k] : 0.f;
  const float rO = go.d_vposer_o ? go.vposer_o[(size_t)b * 128 + (t & 127)] : 0.f;
  const float r6 = go.d_rot6d ? go.rot6d[(size_t)b * 6 + min(t, 5)] : 0.f;
  const int tj = min(t, nj - 1);
  const int rcs = c.child_start[min(t, nj)], rpar = c.parents[tj], rcl = nj > 1 ? c.child_list[min(t, nj - 2)] : 0;
  const int rlj = c.level_joints[tj];                  // joint at position t of the level-ordered list
  int rlev = 0;                                        // ... and its level: #{l : level_start[l + 1] <= t}
  for (int l = 1; l < c.nlev; ++l) rlev += c.level_start[l] <= t ? 1 : 0;
  // J_dirs^T dJ at the end of the kernel: thread (k = t & 31, group = t >> 5) owns rows group, group + 8, ...
  const int sk = max(min(t & 31, c.nshape - 1), 0), sg = t >> 5;
#pragma unroll
  for (int m = 0; m < 24; ++m) rJd[m] = c.J_dirs[(size_t)min(sg + 8 * m, np - 1) * c.nshape + sk];
  if (c.ncomp > 0) {
    const int nh = c.ncomp * 45;
    for (int i0 = 0; i0 < nh; i0 += 1024) {
      float hl[4], hr[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int i = min(i0 + t + 256 * k, nh - 1); hl[k] = c.lh_comp[i]; hr[k] = c.rh_comp[i]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int i = i0 + t + 256 * k; if (i < nh) { hcomp[i] = hl[k]; hcomp[45 * 45 + i] = hr[k]; } }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int i = t + 256 * k;
    if (i < n9) Rs[i] = rR[k];
    if (i < n12) { Ts[i] = rT[k]; dAs[i] = rA[k]; }
  }
  if (t < np) { Js[t] = rJ; fps[t] = rF; dJt[t] = rJt; }
  dXs[t] = rX[0]; dXs[t + 256] = rX[1];
  if (t < 128) o6s[t] = rO;
  if (t < 6) x6s[t] = r6;
  if (t <= nj) cstart[t] = rcs;
  if (t < nj) { par[t] = rpar; lvl[rlj] = rlev; if (t < nj - 1) clist[t] = rcl; }
  __syncthreads(); CENSUS()
  for (int w = t; w < np; w += 256) {          // dJ_i = -T_R^T dA_t
    const int i = w / 3, cc = w % 3;
    const float* dAi = dAs + 12 * i;
    dJ[w] = -(Ts[12 * i + cc] * dAi[3] + Ts[12 * i + 4 + cc] * dAi[7] + Ts[12 * i + 8 + cc] * dAi[11]);
  }
  // G = d(global transforms).  Own terms (A_R = T_R ; A_t = T_t - T_R J ; Jtr = T_t), then children into parents,
  // deepest level first:  G[p] += sum_children [G_R[ch] R_ch^T + G_t[ch] (x) rel_ch | G_t[ch]],  rel_ch = J_ch - J_p.
  // Row r of G[p] only needs row r of the children, so wave r carries row r of every joint (lane = joint) and the
  // levels are ordered by the wave's own program order -- no block barrier per level.  (The block-wide form, one
  // barrier and ~6 dependent LDS reads per level, took 10 x 1.0-1.6 k cycles.)
  if (t < 192) {
    const int r = t >> 6, i = min(t & 63, nj - 1);
    const float* dAi = dAs + 12 * i + 4 * r;
    const float jx = Js[3 * i], jy = Js[3 * i + 1], jz = Js[3 * i + 2];
    float g0 = dAi[0] - dAi[3] * jx, g1 = dAi[1] - dAi[3] * jy, g2 = dAi[2] - dAi[3] * jz;
    float g3 = dAi[3] + dJt[3 * i + r];
    const int mylev = (t & 63) < nj ? lvl[i] : -1, q0 = cstart[i], q1 = cstart[i + 1], p = par[i];
    // what this lane hands to its parent is a function of its own finished row and its own R / rel (in registers);
    // the parent only adds the published 4-vectors of its children: one LDS write -> read per level
    float Rc[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) Rc[e] = Rs[9 * i + e];
    const int pc = max(p, 0);
    const float rx = jx - Js[3 * pc], ry = jy - Js[3 * pc + 1], rz = jz - Js[3 * pc + 2];
    int ch[5];                          // SMPL-X: at most 5 children (the wrists); longer lists take the loop below
#pragma unroll
    for (int k = 0; k < 5; ++k) ch[k] = clist[min(q0 + k, max(nj - 2, 0))];
    for (int lev = c.nlev - 1; lev >= 0; --lev) {
      if (mylev == lev) {
        float4 cv[5];                     // five independent reads, then selects: no branch (and wait) per child
#pragma unroll
        for (int k = 0; k < 5; ++k) cv[k] = ld4(&Cb[12 * ch[k] + 4 * r]);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const bool on = q0 + k < q1;
          g0 += on ? cv[k].x : 0.f; g1 += on ? cv[k].y : 0.f; g2 += on ? cv[k].z : 0.f; g3 += on ? cv[k].w : 0.f;
        }
        for (int q = q0 + 5; q < q1; ++q) { const float4 cv = ld4(&Cb[12 * clist[q] + 4 * r]); g0 += cv.x; g1 += cv.y; g2 += cv.z; g3 += cv.w; }
        st4(&G[12 * i + 4 * r], make_float4(g0, g1, g2, g3));
        st4(&Cb[12 * i + 4 * r], make_float4(g0 * Rc[0] + g1 * Rc[1] + g2 * Rc[2] + g3 * rx,
                                              g0 * Rc[3] + g1 * Rc[4] + g2 * Rc[5] + g3 * ry,
                                              g0 * Rc[6] + g1 * Rc[7] + g2 * Rc[8] + g3 * rz, g3));
      }
      __builtin_amdgcn_wave_barrier();    // (emulation: lanes of the wave rendezvous; hardware: lockstep, scheduling fence)
    }
  }
  __syncthreads(); CENSUS()
  // local grads: dR_i = T_R[p]^T G_R[i] ; drel_i = T_R[p]^T G_t[i]
  for (int w = t; w < nj * 12; w += 256) {
    const int i = w / 12, e = w % 12, r = e >> 2, cc = e & 3;     // (r,cc): element of dR_i (cc<3) or drel (cc==3 -> comp r)
    const int p = par[i];
    float v;
    if (p < 0) v = G[12 * i + 4 * r + cc];
    else v = Ts[12 * p + r] * G[12 * i + cc] + Ts[12 * p + 4 + r] * G[12 * i + 4 + cc] + Ts[12 * p + 8 + r] * G[12 * i + 8 + cc];
    if (cc < 3) {
      float d = v;
      if (i >= 1) d += dXs[c.nshape + (i - 1) * 9 + 3 * r + cc];
      dRl[9 * i + 3 * r + cc] = d;
    } else drel[3 * i + r] = v;
  }
  __syncthreads(); CENSUS()
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
    for (int k = 0; k < 3; ++k) fp3[k] = fps[3 * t + k];
    rodrigues_bwd(fp3, &dRl[9 * t], d3);
    for (int k = 0; k < 3; ++k) dfp[3 * t + k] = d3[k] + (gi.d_full_pose ? gi.d_full_pose[(size_t)b * np + 3 * t + k] : 0.f);
  }
  __syncthreads(); CENSUS()
  // fused consumers: d(global_orient) -> d(rot6d) ; d(body_pose) -> d(VPoser out layer)
  if (go.d_rot6d && t == 0) {
    float x6[6], R[9], dR[9], d6[6];
    for (int k = 0; k < 6; ++k) x6[k] = x6s[k];
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
      for (int k = 0; k < 6; ++k) o6[k] = o6s[6 * jn + k];
      rot6d_fwd(o6, R);
      rotmat_to_aa_bwd(R, &dfp[3 + 3 * jn], dR);
      rot6d_bwd(o6, dR, d6);
      for (int k = 0; k < 6; ++k) go.d_vposer_o[(size_t)b * 128 + 6 * jn + k] = d6[k];
    }
  }
  // scatter d(full_pose)
  for (int i = t - 192; i >= 0 && i < 75; i += 64) {      // wave 3 (waves 0 / 1 are busy with the conversions above)
    const float v = dfp[i];
    if (i < 3) { if (go.d_global_orient) go.d_global_orient[(size_t)b * 3 + i] = v; }
    else if (i < 66) { if (go.d_body_pose) go.d_body_pose[(size_t)b * 63 + (i - 3)] = v; }
    else if (i < 69) { if (go.d_jaw) go.d_jaw[(size_t)b * 3 + (i - 66)] = v; }
    else if (i < 72) { if (go.d_leye) go.d_leye[(size_t)b * 3 + (i - 69)] = v; }
    else { if (go.d_reye) go.d_reye[(size_t)b * 3 + (i - 72)] = v; }
  }
  {
    const int nh = c.ncomp > 0 ? c.ncomp : 45;
    for (int w = t - 128; w >= 0 && w < 2 * nh; w += 128) { // waves 2 and 3
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
  // d(shape coefs) = dX[:nshape] + J_dirs^T dJ   (8 row groups x 32 coefficients, combined in fixed order)
  {
    float a = 0.f;
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      const int i = sg + 8 * m;
      a = fmaf(rJd[m], i < np ? dJ[min(i, np - 1)] : 0.f, a);
    }
    part[t] = a;
  }
  __syncthreads();
  if (t < c.nshape && (go.d_betas || go.d_expr)) {
    float v = dXs[t];
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) v += part[32 * gq + t];
    const int nb = c.nshape / 2;
    if (t < nb) { if (go.d_betas) go.d_betas[(size_t)b * nb + t] = v; }
    else { if (go.d_expr) go.d_expr[(size_t)b * nb + (t - nb)] = v; }
  }
  CENSUS()
}

int smplx_pose_bwd(const BodyConst& c, const PoseWs& ws, const PoseGradIn& gi, const PoseGradOut& go, int B, hipStream_t s) {
  if (c.nj > MAXJ || B <= 0) return LEMO_ERR_SHAPE;
  hipLaunchKernelGGL(smplx_pose_bwd_kernel, dim3(B), dim3(256), 0, s, c, ws, gi, go);
  return (int)hipGetLastError();
}

}  // namespace lemo

CENSUS_SETTER(lemo_census_set_pose)
