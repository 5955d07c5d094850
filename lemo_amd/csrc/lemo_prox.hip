// PROX sliding-window fitting engine (C ABI lemo_prox_*): one iteration of temp_prox/fitting_temp_slide.py's
// `optimizer.step(closure)` as a fixed sequence of kernel launches, captured once into hipGraphs and replayed.
// The twin of the AMASS engine in lemo_hip.hip; kernels: prox_kernels.hip + the shared pose / LBS / encoder kernels.
#include "kernels.hpp"
#include "enc_chain.hpp"
#include <new>

namespace lemo {
int prox_frame_dense(const lemo_prox_desc& d, hipStream_t s);
int prox_sparse(const lemo_prox_desc& d, double smooth_count, hipStream_t s);
int prox_adam(const lemo_prox_desc& d, hipStream_t s);
int prox_tail(const lemo_prox_desc& d, bool update, hipStream_t s);
}
using namespace lemo;

#define CHK(x) do { int e_ = (x); if (e_) return e_; } while (0)
static inline hipStream_t S(void* s) { return (hipStream_t)s; }

static const int PROX_LEVELS = 2;
static const int PROX_UNROLL[PROX_LEVELS] = {10, 1};

struct ProxEngine {
  lemo_prox_desc d;
  hipGraphExec_t exec[PROX_LEVELS] = {nullptr, nullptr};
};

// forward + backward of one iteration (fitting_func :239-311 without the erase, which the update applies)
// `fused_tail`: h1 (first VPoser layer) is already there and the last layer of the VPoser backward is left to prox_tail
static int prox_closure(const lemo_prox_desc& d, hipStream_t s, bool fused_tail = false) {
  const int B = d.B, nj = d.body.nj;
  const int H = 3 * d.fit.n81 + 2, W = B - 1 + 16;
  // ---- body: VPoser decode (:243), ONE SMPL-X forward for both joint sets (:248, :253-258)
  if (!fused_tail) CHK(gemm_nt16(d.vposer.w1, 32, d.pose_embedding, 32, 512, B, 32, d.h1, 512, d.vposer.b1, nullptr, 0, 1, s));
  CHK(gemm_nt16(d.vposer.w2, 512, d.h1, 512, 512, B, 512, d.h2, 512, d.vposer.b2, nullptr, 0, 1, s));
  CHK(gemm_nt16(d.vposer.w3, 512, d.h2, 512, 128, B, 512, d.vo, 128, d.vposer.b3, nullptr, 0, 2, s));
  lemo_pose_in in{};
  in.global_orient = d.global_orient; in.vposer_o = d.vo;
  in.jaw = d.jaw_pose; in.leye = d.leye_pose; in.reye = d.reye_pose;
  in.lh = d.left_hand_pose; in.rh = d.right_hand_pose; in.hand_stride = 12;
  in.betas = d.betas; in.betas_stride = 10; in.expr = d.expression;
  in.zero_f64 = d.loss_acc; in.n_zero = 32 * 32 + 32 * 16; in.step_ctr = d.step_ctr; in.step_cur = d.step_cur;
  in.nonfinite = d.nonfinite;
  CHK(smplx_pose_fwd(d.body, in, d.pose, B, s));
  CHK(lbs_verts_fwd(d.skin, d.pose.Xg, d.Bp, d.pose.A, nj, d.transl, nullptr, d.V, B, d.verts, d.v_posed, s, nullptr, d.pose.XgS));
  // ---- loss: per-frame terms, dense SDF term, smoothness prior through the encoder
  CHK(prox_frame_dense(d, s));
  if (enc_fused_head3(d))
    CHK(enc_head3(d.fit, d.verts, d.V, d.pose.Jtr, nj, d.transl, B, d.enc_w[0], d.enc_b[0], d.enc_w3[1], d.enc_w3_inv[1], d.enc_b[1], d.enc_w3[2],
                  d.enc_w3_inv[2], d.enc_b[2], d.x0, d.canon, d.act[1], d.act[2], d.act[3], s));
  else if (enc_fused_head(d))
    CHK(enc_head(d.fit, d.verts, d.V, d.pose.Jtr, nj, d.transl, B, d.enc_w[0], d.enc_b[0], d.enc_w3[1], d.enc_w3_inv[1], d.enc_b[1], d.x0, d.canon,
                 d.act[1], d.act[2], s));
  else
    CHK(marker_c1(d.fit, d.verts, d.V, d.pose.Jtr, nj, d.transl, B, d.enc_w[0], d.enc_b[0], d.x0, d.canon, d.act[1], d.enc_ch[1], s));
  CHK(enc_chain_fwd(d, H, W, s, enc_fused_head3(d) ? 3 : (enc_fused_head(d) ? 2 : 1)));
  const double cnt = (double)d.enc_ch[10] * H * (W - 1);
  const float coef2 = (float)((double)d.weights_host[8] * 2.0 / cnt);
  CHK(smooth_loss(d.act[10], d.dact[0], nullptr, H, W, d.enc_ch[10], coef2, s, d.loss_acc + 32 * 32));
  // ---- backward
  int cur = 0;
  CHK(enc_chain_bwd(d, H, W, s, &cur, enc_bwd_l_last(d)));
  CHK(enc_bwd_tail(d, cur, H, W, s));
  CHK(prox_sparse(d, cnt, s));
  CHK(lbs_verts_bwd(d.skin, d.uset, d.pose.A, nj, d.v_posed, d.V, d.dverts, B, d.Bp, d.dvp, d.dA, d.dtr_v, d.dX, s));
  lemo_pose_grad_in gi{d.dA, d.dJtr, d.dX, d.dfp_add};
  lemo_pose_grad_out go{};
  go.d_global_orient = d.g_go; go.d_jaw = d.g_jaw; go.d_leye = d.g_leye; go.d_reye = d.g_reye;
  go.d_lh = d.g_lh; go.d_rh = d.g_rh; go.hand_stride = 12; go.d_expr = d.g_expr;
  go.vposer_o = d.vo; go.d_vposer_o = d.vp_scratch;
  CHK(smplx_pose_bwd(d.body, d.pose, gi, go, B, s));
  CHK(vposer_mlp_bwd(d.vposer, d.h1, d.h2, B, fused_tail ? nullptr : d.g_pe, 32, d.vp_scratch, s));
  return 0;
}

// one optimiser iteration.  Default (round 5): the last VPoser backward layer, Adam and the NEXT iteration's first VPoser layer share
// one launch (prox_tail; `first`: the run / graph opens with the h1-only form of it) -- 3 launches and 2 kernel boundaries less per
// iteration; LEMO_PROX_SEPARATE_ADAM: closure + prox_adam as up to round 4 (A/B switch)
static int prox_iteration(const lemo_prox_desc& d, hipStream_t s, bool first) {
  static const bool separate = getenv("LEMO_PROX_SEPARATE_ADAM") != nullptr;
  if (separate || !d.vposer.w1t) {
    CHK(prox_closure(d, s));
    return prox_adam(d, s);
  }
  if (first) CHK(prox_tail(d, false, s));
  CHK(prox_closure(d, s, true));
  return prox_tail(d, true, s);
}

static int prox_capture(ProxEngine* e, hipStream_t s, int iters, hipGraphExec_t* out) {
  hipGraph_t g = nullptr;
  CHK((int)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  for (int i = 0; i < iters && !rc; ++i) rc = prox_iteration(e->d, s, i == 0);
  const int ec = (int)hipStreamEndCapture(s, &g);
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
  CHK(ec);
  const int ic = (int)hipGraphInstantiate(out, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (!ic) (void)hipGraphUpload(*out, s);
  return ic;
}

extern "C" {

void* lemo_prox_create(const lemo_prox_desc* d) {
  if (!d || d->B < 10 || d->B > d->Bp || !d->verts || !d->dverts || !d->sdf || d->uset.n != d->V || !d->fit.cam2world) return nullptr;
  // the all-vertex backward must be one of the deterministic forms: the staged per-frame kernel (small models) or the
  // chunked gather with its partial-sum scratch
  if (!lbs_verts_bwd_fusable(d->skin, d->uset, d->body.nj) && (!d->uset.jcsr_chunk || !d->uset.part || d->uset.part_frames < d->B))
    return nullptr;
  if (d->use_infill && (!d->marker_mask || !d->body_markers_rec || !d->contact_lbl_rec || d->T > d->B - 1 || d->T < 1)) return nullptr;
  if (conv_lds_init() || conv_split_init() || lbs_init()) return nullptr;
  if (d->pose.XgS && (d->skin.DgH != nullptr) != (d->pose.xgs_f16 != 0)) return nullptr;     // both operands of the blend GEMM in one form
  ProxEngine* e = new (std::nothrow) ProxEngine();
  if (e) e->d = *d;
  return e;
}

void lemo_prox_destroy(void* h) {
  ProxEngine* e = (ProxEngine*)h;
  if (!e) return;
  for (int l = 0; l < PROX_LEVELS; ++l) if (e->exec[l]) (void)hipGraphExecDestroy(e->exec[l]);
  delete e;
}

int lemo_prox_closure(void* h, void* stream) {
  ProxEngine* e = (ProxEngine*)h;
  if (!e) return LEMO_ERR_ARG;
  return prox_closure(e->d, S(stream));
}

int lemo_prox_step(void* h, int n, int use_graph, void* stream) {
  ProxEngine* e = (ProxEngine*)h;
  if (!e || n < 0) return LEMO_ERR_ARG;
  hipStream_t s = S(stream);
  if (!use_graph) {
    for (int i = 0; i < n; ++i) CHK(prox_iteration(e->d, s, i == 0));
    return 0;
  }
  // graphs are stream-agnostic once instantiated: kept across stream changes (see fit_graphs)
  // open with single-iteration replays (the device starts while the host still enqueues), then the 10-iteration graph
  int left = n;
  const int head = left < 3 ? left : 3;
  if (head && !e->exec[1]) CHK(prox_capture(e, s, 1, &e->exec[1]));
  if (left - head >= PROX_UNROLL[0] && !e->exec[0]) CHK(prox_capture(e, s, PROX_UNROLL[0], &e->exec[0]));
  for (int i = 0; i < head; ++i) CHK((int)hipGraphLaunch(e->exec[1], s));
  left -= head;
  for (; left >= PROX_UNROLL[0]; left -= PROX_UNROLL[0]) CHK((int)hipGraphLaunch(e->exec[0], s));
  if (left && !e->exec[1]) CHK(prox_capture(e, s, 1, &e->exec[1]));
  for (; left > 0; --left) CHK((int)hipGraphLaunch(e->exec[1], s));
  return 0;
}

static int prox_state_io(ProxEngine* e, const lemo_prox_state* st, bool load, hipStream_t s) {
  if (!e || !st || !st->adam_m || !st->adam_v || !st->step) return LEMO_ERR_ARG;
  const lemo_prox_desc& d = e->d;
  StateCopy a{};
  float* eng[11] = {d.global_orient, d.transl, d.left_hand_pose, d.right_hand_pose, d.jaw_pose, d.leye_pose, d.reye_pose,
                    d.expression, d.pose_embedding, d.adam_m, d.adam_v};
  float* usr[11] = {st->global_orient, st->transl, st->left_hand_pose, st->right_hand_pose, st->jaw_pose, st->leye_pose,
                    st->reye_pose, st->expression, st->pose_embedding, st->adam_m, st->adam_v};
  const int width[11] = {3, 3, 12, 12, 3, 3, 3, 10, 32, 81, 81};
  for (int i = 0; i < 11; ++i) {
    if (!eng[i] || !usr[i]) return LEMO_ERR_ARG;
    a.src[i] = load ? usr[i] : eng[i];
    a.dst[i] = load ? eng[i] : usr[i];
    a.n[i] = d.B * width[i];
  }
  a.njobs = 11;
  a.step_src = load ? st->step : d.step_ctr;
  a.step_dst = load ? d.step_ctr : st->step;
  a.nonfinite = load ? d.nonfinite : nullptr;
  return state_copy(a, s);
}
int lemo_prox_load_state(void* h, const lemo_prox_state* st, void* stream) { return prox_state_io((ProxEngine*)h, st, true, S(stream)); }
int lemo_prox_save_state(void* h, const lemo_prox_state* st, void* stream) { return prox_state_io((ProxEngine*)h, st, false, S(stream)); }

}  // extern "C"
