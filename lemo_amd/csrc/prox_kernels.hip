// Loss kernels of the PROX sliding-window iteration (temp_prox/fitting_temp_slide.py, SMPLifyLoss.forward as configured by
// PROXD_temp_S2.yaml / S3.yaml) and its Adam update -- the native twin of loss_kernels.hip.  Four launches replace the
// ~250 torch ops of the module-level path (lemo_amd/prox.py):
//   prox_frame_kernel   block per frame: 127 smplx joints, 2-D keypoint loss through the fixed perspective camera
//                       (:573-580, camera.py:88-116) with its gradient gathered back to the smplx joints, the L2 / angle
//                       priors (:586-615), and the pairwise sums of the friction (:699-739), infill L1 and contact
//                       velocity terms (:944-992)
//   prox_dense_kernel   thread per (frame, vertex): cam -> world (:676-680), trilinear SDF lookup, penetration loss and
//                       its gradient for ALL vertices (:685-694) -- the dense part of d(loss)/d(verts)
//   prox_sparse_kernel  block per frame, thread per "special" vertex: friction / infill / contact / smoothness / vertex-
//                       joint gradients added onto the dense d(verts); block 0 finalises the 14 loss_dict entries
//   prox_adam_kernel    priors' own gradients, erase of the first int(0.15 B) frames (:282-289), torch.optim.Adam
// prox_frame and prox_dense share ONE launch (prox_frame_dense_kernel: workgroup roles by index).
// Every data-dependent branch of the reference (`.item()` at :690,:719,:730,:736,:974-987) is a device-side count.
#include "scene_device.hpp"
#include "loss_device.hpp"

namespace lemo {

typedef lemo_prox_const ProxConst;

// accumulator slots: f64 [32][32], zeroed by the pose-stage kernel at the start of every iteration
enum { PA_JOINT = 0, PA_PE, PA_BETAS, PA_ANGLE, PA_LH, PA_RH, PA_EXPR, PA_JAW, PA_SDF, PA_FT, PA_FT_N, PA_FN, PA_FN_N,
       PA_INF, PA_INF_N, PA_CT, PA_CT_N = PA_CT + 4, PA_SMOOTH = PA_CT_N + 4, PA_COUNT };
#define PROX_ACC_STRIDE 32
// weights[] indices (LEMO_PROX_NW)
enum { PW_DATA = 0, PW_BODY_POSE, PW_SHAPE, PW_BENDING, PW_HAND, PW_EXPR, PW_JAW, PW_SDF, PW_SMOOTH, PW_FRIC_N, PW_FRIC_T,
       PW_INFILL, PW_INFILL_CONTACT };
// Adam parameter order: global_orient 3 | transl 3 | lh 12 | rh 12 | jaw 3 | leye 3 | reye 3 | expression 10 | pose_embedding 32
#define PROX_NP 81

struct Cam2World { float R[9], t[3]; };
__device__ __forceinline__ void to_world(const Cam2World& c, const float* p, float* w) {
#pragma unroll
  for (int i = 0; i < 3; ++i) w[i] = c.R[3 * i] * p[0] + c.R[3 * i + 1] * p[1] + c.R[3 * i + 2] * p[2] + c.t[i];
}
__device__ __forceinline__ void to_cam_grad(const Cam2World& c, const float* gw, float* gc) {       // R^T gw
#pragma unroll
  for (int k = 0; k < 3; ++k) gc[k] = c.R[k] * gw[0] + c.R[3 + k] * gw[1] + c.R[6 + k] * gw[2];
}

struct ProxFrameIn {
  const float *Jtr, *transl, *verts, *full_pose, *pose_embedding, *betas, *lh, *rh, *expr, *jaw, *lh_comp, *rh_comp;
  const float *gt, *w2, *weights, *marker_mask, *rec, *clbl;
  int nj, ncomp, V, B, T, use_infill;
  float fx, fy, cx, cy;
};
struct ProxFrameOut { double* acc; float *dJtr, *dJv, *dtr_j, *gp, *dfp_add; };

#define PROX_MAXJ 160
// body of a frame-role workgroup: frame b, part `part` of `nparts` (see below)
__device__ __forceinline__ void prox_frame_body(const ProxConst& pc, const ProxFrameIn& in, const ProxFrameOut& out, const Cam2World& cw,
                                                const SdfVol& vol, int b, int part, int nparts) {
  __shared__ float J[PROX_MAXJ * 3], gj[PROX_MAXJ * 3], dJ[PROX_MAXJ * 3];
  __shared__ float h45[90];
  __shared__ float red[4][PA_COUNT];
  const int t = threadIdx.x;
  const int nj = in.nj, V = in.V, B = in.B, nsj = pc.n_sj;
  float acc[PA_COUNT];
#pragma unroll
  for (int i = 0; i < PA_COUNT; ++i) acc[i] = 0.f;
  // the frame's work is split over two workgroups: part 0 = joints, priors, infill and contact sums; part 1 = the
  // friction sums (~1000 vertices x 2 world transforms + an SDF lookup: half of the kernel's time when one block did both)
  const bool do_main = part == 0, do_fric = part == nparts - 1;
  if (do_main) {
  // ---- the 127 smplx joints of this frame: posed skeleton + transl | vertex picks | barycentric landmarks
  for (int w = t; w < nsj * 3; w += 256) {
    const int i = w / 3, c = w % 3;
    float v;
    if (i < nj) v = in.Jtr[((size_t)b * nj + i) * 3 + c] + in.transl[(size_t)b * 3 + c];
    else if (i < nj + pc.n_extra) v = in.verts[((size_t)b * V + pc.extra_rows[i - nj]) * 3 + c];
    else {
      const int l = i - nj - pc.n_extra;
      v = 0.f;
      for (int f = 0; f < 3; ++f) v += in.verts[((size_t)b * V + pc.lmk_rows[3 * l + f]) * 3 + c] * pc.lmk_bary[3 * l + f];
    }
    J[w] = v;
  }
  __syncthreads();
  // ---- 2-D keypoints: |gt - (f X / Z + c)| weighted by (joint_weight conf)^2, mean over B x 118 x 2  (:573-580)
  const float cj = in.weights[PW_DATA] / ((float)B * pc.n_op * 2.f);
  for (int i = t; i < pc.n_op; i += 256) {
    const int j = pc.joint_map[i];
    const float x = J[3 * j], y = J[3 * j + 1], z = J[3 * j + 2];
    const float px = in.fx * (x / z) + in.cx, py = in.fy * (y / z) + in.cy;
    const float dx = in.gt[((size_t)b * pc.n_op + i) * 2] - px, dy = in.gt[((size_t)b * pc.n_op + i) * 2 + 1] - py;
    const float w2 = in.w2[(size_t)b * pc.n_op + i];
    acc[PA_JOINT] += w2 * fabsf(dx) + w2 * fabsf(dy);
    const float gpx = -w2 * cj * (dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f));
    const float gpy = -w2 * cj * (dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f));
    gj[3 * i] = gpx * in.fx / z;
    gj[3 * i + 1] = gpy * in.fy / z;
    gj[3 * i + 2] = -(gpx * in.fx * x + gpy * in.fy * y) / (z * z);
  }
  __syncthreads();
  // gather back to the smplx joints (index_select backward as a fixed-order gather: deterministic)
  for (int w = t; w < nsj * 3; w += 256) {
    const int j = w / 3, c = w % 3;
    float a = 0.f;
    for (int q = pc.jm_start[j]; q < pc.jm_start[j + 1]; ++q) a += gj[3 * pc.jm_list[q] + c];
    dJ[w] = a;
    if (j < nj) out.dJtr[((size_t)b * nj + j) * 3 + c] = a;
    else out.dJv[((size_t)b * (nsj - nj) + (j - nj)) * 3 + c] = a;
  }
  __syncthreads();
  if (t < 3) {       // d(transl) through the posed skeleton joints (Jtr + transl); the vertex joints reach it through d(verts)
    float a = 0.f;
    for (int j = 0; j < nj; ++j) a += dJ[3 * j + t];
    out.dtr_j[(size_t)b * 3 + t] = a;
  }
  // ---- priors (:586-615) and their own gradients (added to the back-propagated ones by the Adam kernel)
  float* gp = out.gp + (size_t)b * PROX_NP;
  const float wbp = in.weights[PW_BODY_POSE], wh = in.weights[PW_HAND], we = in.weights[PW_EXPR], wj = in.weights[PW_JAW];
  if (t < 32) { const float v = in.pose_embedding[(size_t)b * 32 + t]; acc[PA_PE] = v * v; gp[49 + t] = 2.f * wbp * wbp * v; }
  else if (t < 42) { const float v = in.betas[(size_t)b * 10 + (t - 32)]; acc[PA_BETAS] = v * v; }
  else if (t >= 64 && t < 68) {                            // SMPLifyAnglePrior: exp(+-pose) on elbows / knees (prior.py:50-81)
    const int k = t - 64;
    const int idx = k == 0 ? 55 : (k == 1 ? 58 : (k == 2 ? 12 : 15));
    const float sg = k == 0 ? 1.f : -1.f;
    const float e = expf(in.full_pose[(size_t)b * nj * 3 + idx] * sg);
    acc[PA_ANGLE] = e;
    const float wb = in.weights[PW_BENDING];
    out.dfp_add[(size_t)b * nj * 3 + idx] = wb * wb * sg * e;
  } else if (t >= 96 && t < 106) { const float v = in.expr[(size_t)b * 10 + (t - 96)]; acc[PA_EXPR] = v * v; gp[39 + (t - 96)] = 2.f * we * we * v; }
  else if (t >= 112 && t < 115) { const float v = in.jaw[(size_t)b * 3 + (t - 112)]; acc[PA_JAW] = (v * wj) * (v * wj); gp[30 + (t - 112)] = 2.f * wj * wj * v; }
  else if (t >= 128 && t < 128 + 90) {                     // hand prior on the 45-D PCA-expanded poses (smplx output)
    const int side = (t - 128) / 45, c = (t - 128) % 45;
    const float* hp = (side == 0 ? in.lh : in.rh) + (size_t)b * 12;
    const float* comp = side == 0 ? in.lh_comp : in.rh_comp;
    float v = 0.f;
    if (in.ncomp > 0) { for (int k = 0; k < in.ncomp; ++k) v = fmaf(hp[k], comp[k * 45 + c], v); }
    else v = hp[c];
    h45[t - 128] = v;
    acc[side == 0 ? PA_LH : PA_RH] = v * v;
  }
  if (t < 6 || (t >= 33 && t < 39)) gp[t] = 0.f;          // global_orient, transl, leye, reye: no prior
  __syncthreads();
  if (t < 24) {
    const int side = t / 12, k = t % 12;
    const float* comp = side == 0 ? in.lh_comp : in.rh_comp;
    float v = 0.f;
    if (in.ncomp > 0) { for (int c = 0; c < 45; ++c) v = fmaf(h45[side * 45 + c], comp[k * 45 + c], v); }
    else v = h45[side * 45 + k];
    gp[6 + t] = 2.f * wh * wh * v;
  }
  }   // do_main
  // ---- pairwise terms between frame b and b + 1
  if (b < B - 1) {
    // friction (:699-739): foot / gluteus vertices with sdf < 0.01 at frame b; floor normal n = (0,0,1)
    for (int q = do_fric ? t : pc.n_fric; q < pc.n_fric; q += 256) {
      const int vid = pc.fric_vid[q];
      float w0[3], w1[3];
      to_world(cw, in.verts + ((size_t)b * V + vid) * 3, w0);
      to_world(cw, in.verts + ((size_t)(b + 1) * V + vid) * 3, w1);
      if (sdf_at(vol, w0[0], w0[1], w0[2], nullptr) < 0.01f) {
        const float vx = w1[0] - w0[0], vy = w1[1] - w0[1], vz = w1[2] - w0[2];
        const float gt_ = sqrtf(vx * vx + vy * vy);
        if (gt_ - 0.0001f > 0.f) { acc[PA_FT] += gt_; acc[PA_FT_N] += 1.f; }
        if (vz < 0.f) { acc[PA_FN] += fabsf(vz); acc[PA_FN_N] += 1.f; }
      }
    }
    if (do_main && in.use_infill && b < in.T) {            // contact velocity of the heel / toe sets (:953-992)
      for (int k = 0; k < 4; ++k) {
        if (in.clbl[(size_t)b * 4 + k] != 1.f) continue;
        for (int q = pc.foot_start[k] + t; q < pc.foot_start[k + 1]; q += 256) {
          const float* v0 = in.verts + ((size_t)b * V + pc.foot_vid[q]) * 3;
          const float* v1 = v0 + (size_t)V * 3;
          float a0[3], a1[3];
          to_world(cw, v0, a0); to_world(cw, v1, a1);
          const float vx = (a1[0] - a0[0]) * 30.f, vy = (a1[1] - a0[1]) * 30.f, vz = (a1[2] - a0[2]) * 30.f;
          const float sp = sqrtf(vx * vx + vy * vy + vz * vz);
          if (sp - 0.1f > 0.f) { acc[PA_CT + k] += sp; acc[PA_CT_N + k] += 1.f; }
        }
      }
    }
  }
  if (do_main && in.use_infill && b < in.T) {              // infill L1 on the occluded markers (:944-951)
    for (int w = t; w < pc.n67 * 3; w += 256) {
      const int m = w / 3, c = w % 3;
      float mw[3];
      to_world(cw, in.verts + ((size_t)b * V + pc.m67_vid[m]) * 3, mw);
      const float d = fabsf(in.rec[((size_t)b * pc.n67 + m) * 3 + c] - mw[c]) * (1.f - in.marker_mask[(size_t)b * pc.n67 + m]);
      if (d > 0.f) { acc[PA_INF] += d; acc[PA_INF_N] += 1.f; }
    }
  }
  // ---- publish: wave sums, one barrier, fixed-order combine, f64 atomics into slot b & 31
#pragma unroll
  for (int i = 0; i < PA_COUNT; ++i) acc[i] = wave_sum(acc[i]);
  if ((t & 63) == 0) {
#pragma unroll
    for (int i = 0; i < PA_COUNT; ++i) red[t >> 6][i] = acc[i];
  }
  __syncthreads();
  if (t < PA_COUNT) {
    const float v = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
    if (v != 0.f) atomicAdd(out.acc + (b & 31) * PROX_ACC_STRIDE + t, (double)v);
  }
}

// ---- dense part: SDF penetration for every vertex ------------------------------------------------------------------------------
struct ProxDenseIn { const float* verts; int N; const float* weights; float* dverts; double* acc; };
__device__ __forceinline__ void prox_dense_body(const ProxDenseIn& dn, const Cam2World& cw, const SdfVol& vol, int blk) {
  __shared__ float red[4];
  const float* __restrict__ verts = dn.verts; const float* __restrict__ weights = dn.weights;
  float* __restrict__ dverts = dn.dverts; double* __restrict__ acc = dn.acc;
  const int N = dn.N;
  const int i = blk * 256 + threadIdx.x;
  float s = 0.f;
  if (i < N) {
    float w[3], g[3], gc[3] = {0.f, 0.f, 0.f};
    to_world(cw, verts + (size_t)i * 3, w);
    const float val = sdf_at(vol, w[0], w[1], w[2], g);
    if (val < 0.f) {                                       // loss = w sum |sdf| over sdf < 0  ->  d/d world = -w grad
      s = fabsf(val);
      const float wp = -weights[PW_SDF];
      const float gw[3] = {wp * g[0], wp * g[1], wp * g[2]};
      to_cam_grad(cw, gw, gc);
    }
    dverts[(size_t)i * 3] = gc[0]; dverts[(size_t)i * 3 + 1] = gc[1]; dverts[(size_t)i * 3 + 2] = gc[2];
  }
  const float tot = block_sum(s, red);
  if (threadIdx.x == 0 && tot != 0.f) atomicAdd(acc + (blk & 31) * PROX_ACC_STRIDE + PA_SDF, (double)tot);
}
// ONE launch for the two loss stages that read only the vertices and joints (round 5): workgroups 0 .. 2 B - 1 are the frame roles
// (latency chains on 2 B of the 256 CUs for ~16 us when launched alone), the rest the dense SDF roles, which fill the other CUs
// meanwhile -- no fork / join (the graph-branch form of this overlap lost 3 %), one kernel boundary less
__global__ void __launch_bounds__(256)
prox_frame_dense_kernel(ProxConst pc, ProxFrameIn in, ProxFrameOut out, ProxDenseIn dn, Cam2World cw, SdfVol vol, int nparts) {
  const int id = (int)blockIdx.x, nf = in.B * nparts;
  if (id < nf) prox_frame_body(pc, in, out, cw, vol, id % in.B, id / in.B, nparts);
  else prox_dense_body(dn, cw, vol, id - nf);
}

// ---- loss record ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double prox_slot_total(const double* __restrict__ acc, int i) {
  double v = 0.0;
  for (int sl = 0; sl < 32; ++sl) v += acc[sl * PROX_ACC_STRIDE + i];
  return v;
}
// coefficients the gradient needs: weight / count of the count-normalised terms (0 when the selection is empty)
struct ProxInv { float ft, fn, inf, ct[4]; };

__device__ __forceinline__ void prox_finalize(const double* tot, const float* w, int B, int n_op, double smooth_count, int use_infill,
                                              float* losses) {
  const float joint = (float)(tot[PA_JOINT] / ((double)B * n_op * 2)) * w[PW_DATA];
  const float pprior = (float)tot[PA_PE] * (w[PW_BODY_POSE] * w[PW_BODY_POSE]);
  const float shape = (float)tot[PA_BETAS] * (w[PW_SHAPE] * w[PW_SHAPE]);
  const float angle = (float)tot[PA_ANGLE] * (w[PW_BENDING] * w[PW_BENDING]);
  const float lhand = (float)tot[PA_LH] * (w[PW_HAND] * w[PW_HAND]), rhand = (float)tot[PA_RH] * (w[PW_HAND] * w[PW_HAND]);
  const float expr = (float)tot[PA_EXPR] * (w[PW_EXPR] * w[PW_EXPR]);
  const float jaw = (float)tot[PA_JAW];
  const float sdf_pen = w[PW_SDF] > 0.f ? w[PW_SDF] * (float)tot[PA_SDF] : 0.f;
  const float fric_t = tot[PA_FT_N] >= 1.0 ? (float)(tot[PA_FT] / tot[PA_FT_N]) * w[PW_FRIC_T] : 0.f;
  const float fric_n = tot[PA_FN_N] >= 1.0 ? (float)(tot[PA_FN] / tot[PA_FN_N]) * w[PW_FRIC_N] : 0.f;
  float infill = 0.f, infill_c = 0.f;
  if (use_infill) {
    infill = tot[PA_INF_N] >= 1.0 ? w[PW_INFILL] * (float)(tot[PA_INF] / tot[PA_INF_N]) : 0.f;
    float c = 0.f;
    for (int k = 0; k < 4; ++k) c = c + (tot[PA_CT_N + k] >= 1.0 ? (float)(tot[PA_CT + k] / tot[PA_CT_N + k]) : 0.f);
    infill_c = w[PW_INFILL_CONTACT] * c;
  }
  const float smooth = (float)(tot[PA_SMOOTH] / smooth_count) * w[PW_SMOOTH];
  // the reference's order of additions (:1036-1044); the disabled terms are exact zeros
  float total = joint + pprior; total += shape; total += angle; total += 0.f; total += jaw; total += expr; total += lhand;
  total += rhand; total += 0.f; total += 0.f; total += sdf_pen; total += 0.f; total += 0.f; total += 0.f; total += smooth;
  total += fric_t; total += fric_n; total += infill; total += infill_c;
  losses[0] = total; losses[1] = joint; losses[2] = 0.f; losses[3] = 0.f; losses[4] = 0.f; losses[5] = sdf_pen; losses[6] = 0.f;
  losses[7] = 0.f; losses[8] = 0.f; losses[9] = smooth; losses[10] = fric_t; losses[11] = fric_n; losses[12] = infill;
  losses[13] = infill_c; losses[14] = pprior + angle; losses[15] = lhand + rhand + expr + jaw + shape;
}

struct ProxSparseIn {
  const float *verts, *dJv, *weights, *marker_mask, *rec, *clbl, *dx0, *canon;
  const double* acc;
  int V, B, T, use_infill, nvj;
  double smooth_count;
};

__global__ void __launch_bounds__(256)
prox_sparse_kernel(ProxConst pc, FitConst fc, ProxSparseIn in, Cam2World cw, SdfVol vol, float* __restrict__ dverts,
                   float* __restrict__ losses_out) {
  __shared__ double tots[PA_COUNT];
  __shared__ float inv[8];
  __shared__ float dummy_losses[12];
  const int b = blockIdx.x, t = threadIdx.x, V = in.V, B = in.B;
  if (t < 12) dummy_losses[t] = 0.f;
  // target / contact are only dereferenced (clamped reads), never used: m67 = -1, fm = 0 -> any buffer of >= B * max(n67 * 3, 4) floats
  const DvertsIn din = {in.verts, V, in.verts, in.verts, in.dx0, in.canon, in.weights, B, B};
  const bool has_next = b < B - 1, has_prev = b >= 1;
  // ---- everything of a vertex that does not need the loss totals: positions, the two contact decisions of the friction term (SDF
  // lookups), the smoothness gradient through the marker image.  The first vertex of a thread is taken through this BEFORE the totals
  // are read: the kernel is one dependent chain per thread (index -> vertex -> SDF gathers -> image gradient -> joint gradients ->
  // read-modify-write), and the totals' own chain (32 slots -> barrier -> 1 / count -> barrier) used to sit in front of it.
  struct Pre { int vid; float w0[3], wn[3], wp[3], s[3]; bool c0, cp; };
  auto vertex_pre = [&](int u) {
    Pre r;
    r.vid = pc.s_vid[u];
    const float* p = in.verts + ((size_t)b * V + r.vid) * 3;
    to_world(cw, p, r.w0);
    r.wn[0] = r.wn[1] = r.wn[2] = 0.f; r.wp[0] = r.wp[1] = r.wp[2] = 0.f;
    if (has_next) to_world(cw, p + (size_t)V * 3, r.wn);
    if (has_prev) to_world(cw, p - (size_t)V * 3, r.wp);
    const bool fric = pc.s_fric[u] >= 0;
    r.c0 = fric && has_next && sdf_at(vol, r.w0[0], r.w0[1], r.w0[2], nullptr) < 0.01f;
    r.cp = fric && has_prev && sdf_at(vol, r.wp[0], r.wp[1], r.wp[2], nullptr) < 0.01f;
    r.s[0] = r.s[1] = r.s[2] = 0.f;
    const int m81 = pc.s_m81[u];
    if (m81 >= 0) {                                          // smoothness prior through the marker image (canon already folds the cam -> world rotation)
      const DvIdx ix = {r.vid, -1, 0, m81};
      dverts_vertex(fc, din, dummy_losses, b, ix, r.s[0], r.s[1], r.s[2]);
    }
    return r;
  };
  const int u_first = blockIdx.y * 256 + t;
  Pre first;
  if (u_first < pc.n_s) first = vertex_pre(u_first);
  if (t < PA_COUNT) {
    if (t == PA_SMOOTH) {          // the smoothness kernel (smooth_loss_body) adds into its own [32][16] block behind the main one
      double v = 0.0;
      for (int sl = 0; sl < 32; ++sl) v += in.acc[32 * PROX_ACC_STRIDE + sl * 16];
      tots[t] = v;
    } else tots[t] = prox_slot_total(in.acc, t);
  }
  __syncthreads();
  if (t == 0) inv[0] = tots[PA_FT_N] >= 1.0 ? in.weights[PW_FRIC_T] / (float)tots[PA_FT_N] : 0.f;
  if (t == 1) inv[1] = tots[PA_FN_N] >= 1.0 ? in.weights[PW_FRIC_N] / (float)tots[PA_FN_N] : 0.f;
  if (t == 2) inv[2] = (in.use_infill && tots[PA_INF_N] >= 1.0) ? in.weights[PW_INFILL] / (float)tots[PA_INF_N] : 0.f;
  if (t >= 3 && t < 7) inv[t] = (in.use_infill && tots[PA_CT_N + (t - 3)] >= 1.0) ? in.weights[PW_INFILL_CONTACT] / (float)tots[PA_CT_N + (t - 3)] : 0.f;
  if (b == 0 && blockIdx.y == 0 && t == 64) {
    float rec[16];
    prox_finalize(tots, in.weights, B, pc.n_op, in.smooth_count, in.use_infill, rec);
    for (int i = 0; i < 16; ++i) losses_out[i] = rec[i];
  }
  __syncthreads();
  for (int u = u_first; u < pc.n_s; u += 256 * gridDim.y) {     // (frame, quarter of S) per workgroup
    const Pre r = u == u_first ? first : vertex_pre(u);
    const int vid = r.vid;
    const float *w0 = r.w0, *wn = r.wn, *wp = r.wp;
    float g[3] = {0.f, 0.f, 0.f}, gw[3] = {0.f, 0.f, 0.f};
    // friction: this vertex is the `b` end of pair (b, b+1) [contact decided at frame b] and the `b+1` end of (b-1, b)
    if (r.c0) {
      const float vx = wn[0] - w0[0], vy = wn[1] - w0[1], vz = wn[2] - w0[2];
      const float gt_ = sqrtf(vx * vx + vy * vy);
      if (gt_ - 0.0001f > 0.f) { gw[0] -= inv[0] * vx / gt_; gw[1] -= inv[0] * vy / gt_; }
      if (vz < 0.f) gw[2] += inv[1];                       // d mean(-v.z) / d z_b = +1 / count
    }
    if (r.cp) {
      const float vx = w0[0] - wp[0], vy = w0[1] - wp[1], vz = w0[2] - wp[2];
      const float gt_ = sqrtf(vx * vx + vy * vy);
      if (gt_ - 0.0001f > 0.f) { gw[0] += inv[0] * vx / gt_; gw[1] += inv[0] * vy / gt_; }
      if (vz < 0.f) gw[2] -= inv[1];
    }
    // infill L1 on occluded markers
    const int m67 = pc.s_m67[u];
    if (m67 >= 0 && in.use_infill && b < in.T) {
      const float om = 1.f - in.marker_mask[(size_t)b * pc.n67 + m67];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float rr = in.rec[((size_t)b * pc.n67 + m67) * 3 + c] - w0[c];
        if (fabsf(rr) * om > 0.f) gw[c] -= inv[2] * om * (rr > 0.f ? 1.f : -1.f);
      }
    }
    // contact velocity of the heel / toe sets
    const int fm = pc.s_foot_mask[u];
    if (fm && in.use_infill) {
      for (int k = 0; k < 4; ++k) {
        if (!((fm >> k) & 1)) continue;
        const float coef = inv[3 + k] * 30.f;
        if (has_next && b < in.T && in.clbl[(size_t)b * 4 + k] == 1.f) {
          const float vx = (wn[0] - w0[0]) * 30.f, vy = (wn[1] - w0[1]) * 30.f, vz = (wn[2] - w0[2]) * 30.f;
          const float sp = sqrtf(vx * vx + vy * vy + vz * vz);
          if (sp - 0.1f > 0.f) { const float q = coef / sp; gw[0] -= q * vx; gw[1] -= q * vy; gw[2] -= q * vz; }
        }
        if (has_prev && b - 1 < in.T && in.clbl[(size_t)(b - 1) * 4 + k] == 1.f) {
          const float vx = (w0[0] - wp[0]) * 30.f, vy = (w0[1] - wp[1]) * 30.f, vz = (w0[2] - wp[2]) * 30.f;
          const float sp = sqrtf(vx * vx + vy * vy + vz * vz);
          if (sp - 0.1f > 0.f) { const float q = coef / sp; gw[0] += q * vx; gw[1] += q * vy; gw[2] += q * vz; }
        }
      }
    }
    to_cam_grad(cw, gw, g);
    g[0] += r.s[0]; g[1] += r.s[1]; g[2] += r.s[2];          // (the smoothness gradient: zeros when the vertex is no marker of the image)
    // vertex-pick joints and landmarks that read this vertex
    for (int q = pc.s_jstart[u]; q < pc.s_jstart[u + 1]; ++q) {
      const float w = pc.s_jw[q];
      const float* dj = in.dJv + ((size_t)b * in.nvj + pc.s_jidx[q]) * 3;
      g[0] = fmaf(w, dj[0], g[0]); g[1] = fmaf(w, dj[1], g[1]); g[2] = fmaf(w, dj[2], g[2]);
    }
    float* o = dverts + ((size_t)b * V + vid) * 3;
    o[0] += g[0]; o[1] += g[1]; o[2] += g[2];
  }
}

// ---- Adam over the 81 parameters of a frame --------------------------------------------------------------------------------
struct ProxParams {
  float* p[9];                 // global_orient, transl, lh, rh, jaw, leye, reye, expression, pose_embedding
  const float* g[9];           // back-propagated gradients (transl: dtr_v; its joint part comes in dtr_j)
  const float* dtr_j;
  const float* gp;             // [B][81] priors' own gradients
  float *m, *v;
};
__global__ void __launch_bounds__(256)
prox_adam_kernel(ProxParams P, int B, int erase_n, double lr, int* __restrict__ step_ctr, const int* __restrict__ step_cur,
                 int* __restrict__ nonfinite, const float* __restrict__ losses) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int step = *step_cur;
  if (i == 0) *step_ctr = step + 1;
  bool frozen = false;
  if (nonfinite) {
    frozen = nonfinite[1] != 0;
    if (i == 0 && nonfinite[0] == 0) {
      const float tot = losses[0];
      if (!(fabsf(tot) <= 3.402823466e38f)) nonfinite[0] = step + 1;
    }
  }
  if (i >= B * PROX_NP || frozen) return;
  const int b = i / PROX_NP, k = i - b * PROX_NP;
  // segment of the parameter vector
  const int off[10] = {0, 3, 6, 18, 30, 33, 36, 39, 49, 81};
  int sgm = 0;
#pragma unroll
  for (int q = 1; q < 9; ++q) sgm += k >= off[q] ? 1 : 0;
  const int dim = off[sgm + 1] - off[sgm], e = k - off[sgm];
  float* pp = P.p[sgm] + (size_t)b * dim + e;
  float grad = P.g[sgm][(size_t)b * dim + e] + P.gp[(size_t)b * PROX_NP + k];
  if (sgm == 1) grad += P.dtr_j[(size_t)b * 3 + e];
  if (b < erase_n) grad = 0.f;                             // "erase gradient for first 15 frames" (:282-289)
  float pv = *pp, m = P.m[i], v = P.v[i];
  adam_update_torch(pv, m, v, grad, adam_coef_t(step + 1, lr));      // common.hpp: torch's own evaluation order
  P.m[i] = m; P.v[i] = v;
  *pp = pv;
}

// ---- tail of one PROX iteration, one workgroup per frame (three launches -> one; fit_tail_kernel of loss_kernels.hip is the AMASS twin):
//   dz   = dh1 . W1              the last layer of the VPoser decoder backward (vposer_smpl.py:107-115 transposed) = d(loss)/d(pose_embedding)
//   Adam on this frame's 81 parameters (prox_adam_kernel above: same arithmetic, same erase / latch / counter protocol)
//   h1'  = lrelu(W1 z' + b1)     the FIRST layer of the NEXT iteration's decoder forward, on the updated embedding
// The launch that opens a graph (or an eager run) runs this kernel with do_dz = do_adam = 0: h1 has the same bits whichever launch made it.
struct ProxTail {
  ProxParams P;
  const float *w1, *w1t, *b1, *dh1;
  float *g_pe, *h1;
  int B, erase_n, do_dz, do_adam;
  double lr;
  int* step_ctr; const int* step_cur; int* nonfinite; const float* losses;
};
__global__ void __launch_bounds__(256)
prox_tail_kernel(ProxTail a) {
  __shared__ __attribute__((aligned(16))) float dh[512];
  __shared__ float zs[32], dzs[32];
  const int b = blockIdx.x, t = threadIdx.x;
  // every global read first, with clamped (never predicated) addresses: the phases below depend on each other
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
  // this thread's parameter (threads >= 81 shadow element 80: same addresses, nothing stored)
  const int k = t < PROX_NP ? t : PROX_NP - 1;
  const int off[10] = {0, 3, 6, 18, 30, 33, 36, 39, 49, 81};
  int sgm = 0;
#pragma unroll
  for (int q = 1; q < 9; ++q) sgm += k >= off[q] ? 1 : 0;
  const int dim = off[sgm + 1] - off[sgm], e = k - off[sgm];
  float* pp = a.P.p[sgm] + (size_t)b * dim + e;
  const int i = b * PROX_NP + k;
  float p_old = *pp, m_old = 0.f, v_old = 0.f, g_in = 0.f, g_prior = 0.f, g_trj = 0.f, tot = 0.f;
  int step = 0, nf0 = 0, nf1 = 0;
  if (a.do_adam) {
    m_old = a.P.m[i]; v_old = a.P.v[i];
    // pose_embedding under do_dz: its gradient is the dz computed below and g_pe has not been written yet (vposer_mlp_bwd ran with
    // dz = nullptr) -- the load stays unconditional (one address select, no branch) but reads an initialised word
    const float* g_src = (a.do_dz && sgm == 8) ? a.P.gp + i : a.P.g[sgm] + (size_t)b * dim + e;
    g_in = *g_src;
    g_prior = a.P.gp[i];
    g_trj = a.P.dtr_j[(size_t)b * 3 + (sgm == 1 ? e : 0)];
    step = *a.step_cur;
    if (a.nonfinite) { nf0 = a.nonfinite[0]; nf1 = a.nonfinite[1]; tot = a.losses[0]; }
  }
  AdamCoef coef{0.f, 1.f};
  if (a.do_adam) coef = adam_coef_t(step + 1, a.lr);
  LEMO_PIN(p_old); LEMO_PIN(m_old); LEMO_PIN(v_old); LEMO_PIN(g_in); LEMO_PIN(g_prior); LEMO_PIN(g_trj); LEMO_PIN(tot);
  LEMO_PIN(bias[0]); LEMO_PIN(bias[1]);
  if (a.do_dz) {
    // dz[m] = sum_k w1t[m][k] dh1[b][k] : thread = (m = t >> 3, part = t & 7) takes k = 4 part + 32 i .. + 3
    dh[t] = dh_a;
    dh[t + 256] = dh_b;
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int i2 = 0; i2 < 16; ++i2) {
      const float4 d4 = ld4(&dh[4 * part + 32 * i2]);
      acc = fmaf(wv[i2].x, d4.x, acc); acc = fmaf(wv[i2].y, d4.y, acc); acc = fmaf(wv[i2].z, d4.z, acc); acc = fmaf(wv[i2].w, d4.w, acc);
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    if (part == 0) { a.g_pe[(size_t)b * 32 + m_dz] = acc; dzs[m_dz] = acc; }
  }
  __syncthreads();
  float p_new = p_old;
  if (a.do_adam) {
    if (b == 0 && t == 0) {
      *a.step_ctr = step + 1;
      if (a.nonfinite && nf0 == 0 && !(fabsf(tot) <= 3.402823466e38f)) a.nonfinite[0] = step + 1;
    }
    if (t < PROX_NP && nf1 == 0) {
      float grad = ((a.do_dz && sgm == 8) ? dzs[e] : g_in) + g_prior;
      if (sgm == 1) grad += g_trj;
      if (b < a.erase_n) grad = 0.f;                         // "erase gradient for first 15 frames" (:282-289)
      float mm = m_old, vv = v_old;
      adam_update_torch(p_new, mm, vv, grad, coef);
      a.P.m[i] = mm; a.P.v[i] = vv;
      *pp = p_new;
    }
  }
  if (!a.h1) return;
  if (t >= 49 && t < PROX_NP) zs[t - 49] = p_new;            // the embedding of this frame, as updated above
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    float acc = bias[r];
#pragma unroll
    for (int i2 = 0; i2 < 8; ++i2) {
      acc = fmaf(w8[r][i2].x, zs[4 * i2], acc); acc = fmaf(w8[r][i2].y, zs[4 * i2 + 1], acc);
      acc = fmaf(w8[r][i2].z, zs[4 * i2 + 2], acc); acc = fmaf(w8[r][i2].w, zs[4 * i2 + 3], acc);
    }
    a.h1[(size_t)b * 512 + t + 256 * r] = lrelu(acc);
  }
}

// ---- launchers ---------------------------------------------------------------------------------------------------------------
static Cam2World make_cw(const float* c) { Cam2World w; for (int i = 0; i < 9; ++i) w.R[i] = c[i]; for (int i = 0; i < 3; ++i) w.t[i] = c[9 + i]; return w; }

// the two roles as launches of their own (A/B switch LEMO_PROX_TWO_LAUNCHES: the form up to round 4)
__global__ void __launch_bounds__(256)
prox_frame_kernel(ProxConst pc, ProxFrameIn in, ProxFrameOut out, Cam2World cw, SdfVol vol) {
  prox_frame_body(pc, in, out, cw, vol, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}
__global__ void __launch_bounds__(256)
prox_dense_kernel(ProxDenseIn dn, Cam2World cw, SdfVol vol) { prox_dense_body(dn, cw, vol, (int)blockIdx.x); }

int prox_frame_dense(const lemo_prox_desc& d, hipStream_t s) {
  if (d.pc.n_sj > PROX_MAXJ || d.pc.n_op > PROX_MAXJ) return LEMO_ERR_SHAPE;
  ProxFrameIn in{};
  in.Jtr = d.pose.Jtr; in.transl = d.transl; in.verts = d.verts; in.full_pose = d.pose.full_pose; in.pose_embedding = d.pose_embedding;
  in.betas = d.betas; in.lh = d.left_hand_pose; in.rh = d.right_hand_pose; in.expr = d.expression; in.jaw = d.jaw_pose;
  in.lh_comp = d.body.lh_comp; in.rh_comp = d.body.rh_comp; in.gt = d.gt_joints; in.w2 = d.w2; in.weights = d.weights;
  in.marker_mask = d.marker_mask; in.rec = d.body_markers_rec; in.clbl = d.contact_lbl_rec;
  in.nj = d.body.nj; in.ncomp = d.body.ncomp; in.V = d.V; in.B = d.B; in.T = d.T; in.use_infill = d.use_infill;
  in.fx = d.cam[0]; in.fy = d.cam[1]; in.cx = d.cam[2]; in.cy = d.cam[3];
  ProxFrameOut out{d.loss_acc, d.dJtr, d.dJv, d.dtr_j, d.gp, d.dfp_add};
  const int N = d.B * d.V;
  ProxDenseIn dn{d.verts, N, d.weights, d.dverts, d.loss_acc};
  static const bool two = getenv("LEMO_PROX_TWO_LAUNCHES") != nullptr;
  const SdfVol vol = make_sdf_vol(d.sdf, d.sdf_dim[0], d.sdf_dim[1], d.sdf_dim[2], d.grid_min, d.grid_max);
  if (two) {
    hipLaunchKernelGGL(prox_frame_kernel, dim3(d.B, 2), dim3(256), 0, s, d.pc, in, out, make_cw(d.cam2world), vol);
    hipLaunchKernelGGL(prox_dense_kernel, dim3((N + 255) / 256), dim3(256), 0, s, dn, make_cw(d.cam2world), vol);
  } else
    hipLaunchKernelGGL(prox_frame_dense_kernel, dim3(2 * d.B + (N + 255) / 256), dim3(256), 0, s, d.pc, in, out, dn, make_cw(d.cam2world), vol, 2);
  return (int)hipGetLastError();
}

int prox_sparse(const lemo_prox_desc& d, double smooth_count, hipStream_t s) {
  ProxSparseIn in{};
  in.verts = d.verts; in.dJv = d.dJv; in.weights = d.weights; in.marker_mask = d.marker_mask; in.rec = d.body_markers_rec;
  in.clbl = d.contact_lbl_rec; in.dx0 = d.dx0; in.canon = d.canon; in.acc = d.loss_acc; in.V = d.V; in.B = d.B; in.T = d.T;
  in.use_infill = d.use_infill; in.nvj = d.pc.n_sj - d.body.nj; in.smooth_count = smooth_count;
  hipLaunchKernelGGL(prox_sparse_kernel, dim3(d.B, (d.pc.n_s + 255) / 256), dim3(256), 0, s, d.pc, d.fit, in, make_cw(d.cam2world),
                     make_sdf_vol(d.sdf, d.sdf_dim[0], d.sdf_dim[1], d.sdf_dim[2], d.grid_min, d.grid_max), d.dverts, d.losses);
  return (int)hipGetLastError();
}

int prox_adam(const lemo_prox_desc& d, hipStream_t s) {
  ProxParams P{};
  float* p[9] = {d.global_orient, d.transl, d.left_hand_pose, d.right_hand_pose, d.jaw_pose, d.leye_pose, d.reye_pose, d.expression, d.pose_embedding};
  const float* g[9] = {d.g_go, d.dtr_v, d.g_lh, d.g_rh, d.g_jaw, d.g_leye, d.g_reye, d.g_expr, d.g_pe};
  for (int i = 0; i < 9; ++i) { P.p[i] = p[i]; P.g[i] = g[i]; }
  P.dtr_j = d.dtr_j; P.gp = d.gp; P.m = d.adam_m; P.v = d.adam_v;
  const int n = d.B * PROX_NP;
  const int erase_n = d.first_batch_flag ? 0 : (int)(d.B * 0.15);
  hipLaunchKernelGGL(prox_adam_kernel, dim3((n + 255) / 256), dim3(256), 0, s, P, d.B, erase_n, lr_decimal(d.lr), d.step_ctr, d.step_cur, d.nonfinite, d.losses);
  return (int)hipGetLastError();
}

// update = false: only h1 = first VPoser layer of the current embedding (opens a graph / an eager run); update = true: the whole tail
int prox_tail(const lemo_prox_desc& d, bool update, hipStream_t s) {
  ProxTail a{};
  float* p[9] = {d.global_orient, d.transl, d.left_hand_pose, d.right_hand_pose, d.jaw_pose, d.leye_pose, d.reye_pose, d.expression, d.pose_embedding};
  const float* g[9] = {d.g_go, d.dtr_v, d.g_lh, d.g_rh, d.g_jaw, d.g_leye, d.g_reye, d.g_expr, d.g_pe};
  for (int i = 0; i < 9; ++i) { a.P.p[i] = p[i]; a.P.g[i] = g[i]; }
  a.P.dtr_j = d.dtr_j; a.P.gp = d.gp; a.P.m = d.adam_m; a.P.v = d.adam_v;
  a.w1 = d.vposer.w1; a.w1t = d.vposer.w1t; a.b1 = d.vposer.b1;
  a.dh1 = vposer_scratch_dh1(d.vp_scratch, d.B);             // kernels.hpp: the layout vposer_mlp_bwd writes
  a.g_pe = d.g_pe; a.h1 = d.h1;
  a.B = d.B; a.erase_n = d.first_batch_flag ? 0 : (int)(d.B * 0.15);
  a.do_dz = a.do_adam = update ? 1 : 0;
  a.lr = lr_decimal(d.lr); a.step_ctr = d.step_ctr; a.step_cur = d.step_cur; a.nonfinite = d.nonfinite; a.losses = d.losses;
  if (!a.w1 || !a.w1t || !a.b1 || !a.h1 || !d.vp_scratch || !a.g_pe) return LEMO_ERR_ARG;
  hipLaunchKernelGGL(prox_tail_kernel, dim3(d.B), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace lemo
