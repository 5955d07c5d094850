// C ABI of liblemo_hip.so (see include/lemo_hip.h) and the native fitting engine: the launch
// sequence of one AMASS temporal-fitting iteration (opt_amass_temp.py:349-455), optionally captured
// once into a hipGraph and replayed (the iteration is ~40 short kernels: launch-bound without it).
#include "kernels.hpp"
#include "enc_chain.hpp"

#include <new>

using namespace lemo;

#define S(x) ((hipStream_t)(x))
#define CHK(e) do { int _e = (e); if (_e) return _e; } while (0)

extern "C" {

int lemo_abi_version(void) { return 5; }
int lemo_build_flags(void) {
#ifdef LEMO_NO_PACKED_FP32
  return 1;
#else
  return 0;
#endif
}

int lemo_conv3x3_mfma(const float* in, const float* wt, const float* bias, const float* aux, float* out, int H, int W,
                      int cin, int cout, int epi, int variant, void* stream) {
  if (!in || !wt || !out || (epi != 1 && !bias) || (epi == 1 && !aux)) return LEMO_ERR_ARG;
  return conv3x3_mfma(in, wt, bias, aux, out, H, W, cin, cout, epi, variant, S(stream));
}
int lemo_conv3x3_mfma_splitk(const float* in, const float* wt, const float* bias, const float* aux, float* out,
                             float* partial, int ks, int H, int W, int cin, int cout, int epi, void* stream) {
  if (!in || !wt || !out || !partial || (epi != 1 && !bias) || (epi == 1 && !aux)) return LEMO_ERR_ARG;
  return conv3x3_mfma_splitk(in, wt, bias, aux, out, partial, ks, H, W, cin, cout, epi, S(stream));
}
int lemo_conv3x3_mfma_lds(const float* in, const float* wt, const float* wt2, const float* bias, const float* aux,
                          float* out, int H, int W, int cin, int cout, int epi, void* stream) {
  if (!in || !wt || !wt2 || !out || (epi != 1 && !bias) || (epi == 1 && !aux)) return LEMO_ERR_ARG;
  return conv3x3_mfma_lds(in, wt, wt2, bias, aux, out, H, W, cin, cout, epi, S(stream));
}
int lemo_conv3x3_split_supported(int H, int W, int cin, int cout) { return conv3x3_split_supported(H, W, cin, cout) ? 1 : 0; }
int lemo_conv3x3_mfma_split(const float* in, const void* w3, const float* wt, const float* bias, const float* aux,
                            float* out, int H, int W, int cin, int cout, int epi, void* stream) {
  if (!in || !w3 || !wt || !out || (epi != 1 && !bias) || (epi == 1 && !aux)) return LEMO_ERR_ARG;
  return conv3x3_mfma_split(in, w3, wt, bias, aux, out, H, W, cin, cout, epi, S(stream));
}
int lemo_conv3x3_mfma_split_census(const float* in, const void* w3, const float* wt, const float* bias, float* out,
                                   int H, int W, int cin, int cout, unsigned long long* dbg, void* stream) {
  if (!in || !w3 || !wt || !bias || !out || !dbg) return LEMO_ERR_ARG;
  return conv3x3_mfma_split(in, w3, wt, bias, nullptr, out, H, W, cin, cout, 0, S(stream), dbg);
}
int lemo_conv3x3_mfma_split_f16(const float* in, const void* w2, float winv, const float* wt, const float* bias, const float* aux,
                                float* out, int H, int W, int cin, int cout, int epi, void* stream) {
  if (!in || !w2 || !wt || !out || (epi != 1 && !bias) || (epi == 1 && !aux)) return LEMO_ERR_ARG;
  return conv3x3_mfma_split(in, w2, wt, bias, aux, out, H, W, cin, cout, epi, S(stream), nullptr, 2, winv);
}
int lemo_conv3x3_wino_supported(int H, int W, int cin, int cout) { return conv3x3_wino_supported(H, W, cin, cout) ? 1 : 0; }
int lemo_conv3x3_wino_f16(const float* in, const void* wU, float winv, const float* wt, const float* bias, const float* aux, float* out,
                          int H, int W, int epi, unsigned long long* dbg, void* stream) {
  return conv3x3_wino_f16(in, wU, winv, wt, bias, aux, out, H, W, epi, S(stream), dbg);
}
int lemo_conv3x3_pair_supported(int H, int W, int c0, int c1, int c2) { return conv3x3_pair_supported(H, W, c0, c1, c2) ? 1 : 0; }
int lemo_conv3x3_pair_f16(const float* in, const void* wA, float winvA, const float* biasA, const float* auxA, float* mid,
                          const void* wB, float winvB, const float* biasB, const float* auxB, float* out, int H, int W, int epi,
                          unsigned long long* dbg, void* stream) {
  return conv3x3_pair_f16(in, wA, winvA, biasA, auxA, mid, wB, winvB, biasB, auxB, out, H, W, epi, S(stream), dbg);
}
int lemo_enc_head(const lemo_fit_const* fc, const float* verts, int nrows, const float* Jtr, int nj, const float* transl, int B, const float* w0,
                  const float* b0, const void* w1pack, float w1inv, const float* b1, float* x0, float* canon, float* act1, float* act2,
                  void* stream) {
  if (!fc) return LEMO_ERR_ARG;
  return enc_head(*fc, verts, nrows, Jtr, nj, transl, B, w0, b0, w1pack, w1inv, b1, x0, canon, act1, act2, S(stream));
}
int lemo_enc_tail(const float* din, const void* w1bpack, float w1binv, const float* act1, const float* w0, float* dx0, int H, int W,
                  void* stream) {
  return enc_tail(din, w1bpack, w1binv, act1, w0, dx0, H, W, S(stream));
}
int lemo_enc_tail3(const float* din, const void* w2bpack, float w2binv, const float* act2, const void* w1bpack, float w1binv, const float* act1,
                   const float* w0, float* dx0, int H, int W, void* stream) {
  return enc_tail3(din, w2bpack, w2binv, act2, w1bpack, w1binv, act1, w0, dx0, H, W, S(stream));
}
int lemo_conv3x3_mfma_split_census2(const float* in, const void* w, float winv, int pieces, const float* wt, const float* bias, float* out,
                                    int H, int W, int cin, int cout, unsigned long long* dbg, void* stream) {
  if (!in || !w || !wt || !bias || !out || !dbg) return LEMO_ERR_ARG;
  return conv3x3_mfma_split(in, w, wt, bias, nullptr, out, H, W, cin, cout, 0, S(stream), dbg, pieces, winv);
}
int lemo_conv3x3_mfma_lds_census(const float* in, const float* wt, const float* wt2, const float* bias, float* out,
                                 int H, int W, int cin, int cout, unsigned long long* dbg, void* stream) {
  if (!in || !wt || !wt2 || !out || !bias || !dbg) return LEMO_ERR_ARG;
  return conv3x3_mfma_lds(in, wt, wt2, bias, nullptr, out, H, W, cin, cout, 0, S(stream), dbg);
}
int lemo_conv3x3_c1(const float* x0, const float* w, const float* bias, float* out, int H, int W, int cout, void* stream) {
  return conv3x3_c1(x0, w, bias, out, H, W, cout, S(stream));
}
int lemo_conv3x3_c1_bwd(const float* dpre, const float* w, float* dx0, int H, int W, int cout, void* stream) {
  return conv3x3_c1_bwd(dpre, w, dx0, H, W, cout, S(stream));
}
int lemo_smooth_loss_blocks(int H, int W, int C) { return smooth_loss_blocks(H, W, C); }
int lemo_smooth_loss(const float* z, float* dpre, float* partial, int H, int W, int C, float coef2, void* stream) {
  return smooth_loss(z, dpre, partial, H, W, C, coef2, S(stream));
}

int lemo_vposer_decode_fwd(const lemo_vposer_w* w, const float* z, int z_stride, int B, float* h1, float* h2, float* o,
                           float* matrot, float* aa, void* stream) {
  if (!w || !z || !h1 || !h2 || !o) return LEMO_ERR_ARG;
  return vposer_decode_fwd(*w, z, z_stride, B, h1, h2, o, matrot, aa, S(stream));
}
int lemo_vposer_decode_bwd(const lemo_vposer_w* w, const float* h1, const float* h2, const float* o, const float* d_aa,
                           const float* d_matrot, int B, float* dz, int dz_stride, float* scratch, void* stream) {
  if (!w || !dz || !scratch || (!d_aa && !d_matrot)) return LEMO_ERR_ARG;
  return vposer_decode_bwd(*w, h1, h2, o, nullptr, d_aa, d_matrot, B, dz, dz_stride, scratch, S(stream));
}
int lemo_vposer_mlp_bwd(const lemo_vposer_w* w, const float* h1, const float* h2, int B, float* dz, int dz_stride,
                        float* scratch, void* stream) {
  if (!w || !h1 || !h2 || !dz || !scratch) return LEMO_ERR_ARG;
  return vposer_mlp_bwd(*w, h1, h2, B, dz, dz_stride, scratch, S(stream));
}
int lemo_gemm_nt16(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc,
                   const float* bias, const float* aux, int ldaux, int epi, void* stream) {
  if (!A || !B || !C) return LEMO_ERR_ARG;
  return gemm_nt16(A, lda, B, ldb, M, N, K, C, ldc, bias, aux, ldaux, epi, S(stream));
}
int lemo_gemm_nt16_splitk_part_floats(int M, int S) { return gemm_nt16_splitk_part_floats(M, S); }
int lemo_gemm_nt16_splitk(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc, float* part, int S,
                          const float* A_grouped, void* stream) {
  if (!A || !B || !C || !part) return LEMO_ERR_ARG;
  return gemm_nt16_splitk(A, lda, B, ldb, M, N, K, C, ldc, part, S, S(stream), A_grouped);
}
int lemo_rot6d_to_aa_fwd(const float* x6, int stride, int N, float* aa, void* stream) {
  return rot6d_to_aa_fwd(x6, stride, N, aa, S(stream));
}
int lemo_rot6d_to_aa_bwd(const float* x6, int stride, const float* d_aa, int N, float* dx6, void* stream) {
  return rot6d_to_aa_bwd(x6, stride, d_aa, N, dx6, S(stream));
}
int lemo_smplx_pose_fwd(const lemo_body_const* c, const lemo_pose_in* in, const lemo_pose_ws* ws, int B, void* stream) {
  if (!c || !in || !ws) return LEMO_ERR_ARG;
  return smplx_pose_fwd(*c, *in, *ws, B, S(stream));
}
int lemo_smplx_pose_bwd(const lemo_body_const* c, const lemo_pose_ws* ws, const lemo_pose_grad_in* gi,
                        const lemo_pose_grad_out* go, int B, void* stream) {
  if (!c || !ws || !gi || !go || !gi->dA) return LEMO_ERR_ARG;
  return smplx_pose_bwd(*c, *ws, *gi, *go, B, S(stream));
}
int lemo_lbs_verts_fwd_active(const lemo_skin_const* c, const lemo_vertex_set_bwd* u, const float* Xg, int Bp, const float* A,
                              int nj, const float* transl, int B, float* blend, float* verts, float* v_posed, void* stream) {
  if (!c || !u || !Xg || !A || !blend || !verts) return LEMO_ERR_ARG;
  return lbs_verts_fwd_active(*c, *u, Xg, Bp, A, nj, transl, B, blend, verts, v_posed, S(stream));
}
int lemo_lbs_verts_fwd(const lemo_skin_const* c, const float* Xg, int Bp, const float* A, int nj, const float* transl,
                       const int* ids, int n, int B, float* verts, float* v_posed, void* stream) {
  if (!c || !Xg || !A || !verts) return LEMO_ERR_ARG;
  return lbs_verts_fwd(*c, Xg, Bp, A, nj, transl, ids, n, B, verts, v_posed, S(stream));
}
int lemo_lbs_verts_fwd_xs(const lemo_skin_const* c, const float* Xg, const unsigned short* XgS, int Bp, const float* A, int nj,
                          const float* transl, const int* ids, int n, int B, float* verts, float* v_posed, void* stream) {
  if (!c || !Xg || !A || !verts) return LEMO_ERR_ARG;
  return lbs_verts_fwd(*c, Xg, Bp, A, nj, transl, ids, n, B, verts, v_posed, S(stream), nullptr, XgS);
}
int lemo_lbs_verts_fwd_census(const lemo_skin_const* c, const float* Xg, int Bp, const float* A, int nj, const float* transl,
                              int n, int B, float* verts, float* v_posed, unsigned long long* dbg, void* stream,
                              const unsigned short* XgS) {
  if (!c || !Xg || !A || !verts || !dbg) return LEMO_ERR_ARG;
  return lbs_verts_fwd(*c, Xg, Bp, A, nj, transl, nullptr, n, B, verts, v_posed, S(stream), dbg, XgS);
}
int lemo_lbs_verts_bwd(const lemo_skin_const* c, const lemo_vertex_set_bwd* u, const float* A, int nj, const float* v_posed,
                       int vp_rows, const float* dverts, int B, int Bp, float* dvp, float* dA, float* dtransl, float* dX,
                       void* stream) {
  if (!c || !u || !A || !v_posed || !dverts || !dvp || !dA || !dX) return LEMO_ERR_ARG;
  return lbs_verts_bwd(*c, *u, A, nj, v_posed, vp_rows, dverts, B, Bp, dvp, dA, dtransl, dX, S(stream));
}
int lemo_joints_assemble(const float* Jtr, int nj, const float* verts, int vrows, const int* extra_rows, int n_extra,
                         const int* lmk_rows, const float* lmk_bary, int n_lmk, const float* transl, int B, float* joints,
                         void* stream) {
  return joints_assemble(Jtr, nj, verts, vrows, extra_rows, n_extra, lmk_rows, lmk_bary, n_lmk, transl, B, joints, S(stream));
}

int lemo_maxpool3s2_fwd(const float* in, int H, int W, float* out, unsigned char* idx, int C, void* stream) {
  if (!in || !out || !idx) return LEMO_ERR_ARG;
  return maxpool3s2_fwd(in, H, W, out, idx, C, S(stream));
}
int lemo_maxpool3s2_bwd(const float* dout, const unsigned char* idx, const float* act, float* din, int H, int W, int C, void* stream) {
  if (!dout || !idx || !din) return LEMO_ERR_ARG;
  return maxpool3s2_bwd(dout, idx, act, din, H, W, C, S(stream));
}
int lemo_stuff2_fwd(const float* in, int h, int w, float* out, int H, int W, int C, void* stream) {
  if (!in || !out) return LEMO_ERR_ARG;
  return stuff2_fwd(in, h, w, out, H, W, C, S(stream));
}
int lemo_stuff2_bwd(const float* dout, int H, int W, const float* act, float* din, int h, int w, int C, void* stream) {
  if (!dout || !din) return LEMO_ERR_ARG;
  return stuff2_bwd(dout, H, W, act, din, h, w, C, S(stream));
}
int lemo_conv3x3_wgrad_nslab(int H, int W) { return conv3x3_wgrad_nslab(H, W); }
int lemo_conv3x3_wgrad(const float* dy, const float* x, int H, int W, int cin, int cout, int cin_real, int cout_real,
                       float* partial, float* dw, float* db, void* stream) {
  if (!dy || !x || !partial || !dw) return LEMO_ERR_ARG;
  return conv3x3_wgrad(dy, x, H, W, cin, cout, cin_real, cout_real, partial, dw, db, S(stream));
}
int lemo_conv3x3_wgrad_partial(const float* dy, const float* x, int H, int W, int cin, int cout, float* partial, void* stream) {
  if (!dy || !x || !partial) return LEMO_ERR_ARG;
  return conv3x3_wgrad_partial(dy, x, H, W, cin, cout, partial, S(stream));
}
int lemo_conv3x3_wgrad_reduce_multi(const lemo_wgrad_job* jobs, int n, void* stream) {
  return conv3x3_wgrad_reduce_multi(jobs, n, S(stream));
}
int lemo_adam_flat(float* p, const float* g, float* m, float* v, int n, float lr, int step, void* stream) {
  if (!p || !g || !m || !v) return LEMO_ERR_ARG;
  return adam_flat(p, g, m, v, n, lr, step, nullptr, S(stream));
}
int lemo_adam_flat_ctr(float* p, const float* g, float* m, float* v, int n, float lr, int* step_ctr, void* stream) {
  if (!p || !g || !m || !v || !step_ctr) return LEMO_ERR_ARG;
  return adam_flat(p, g, m, v, n, lr, 0, step_ctr, S(stream));
}
int lemo_capture_begin(void* stream) {
  if (!stream) return LEMO_ERR_ARG;                      // the legacy default stream cannot be captured
  return (int)hipStreamBeginCapture(S(stream), hipStreamCaptureModeRelaxed);
}
int lemo_capture_end(void* stream, void** graph_exec) {
  if (!stream || !graph_exec) return LEMO_ERR_ARG;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(S(stream), &g);
  if (e != hipSuccess || !g) return e != hipSuccess ? (int)e : LEMO_ERR_STATE;
  hipGraphExec_t x = nullptr;
  e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return (int)e;
  *graph_exec = (void*)x;
  return 0;
}
int lemo_graph_launch(void* graph_exec, void* stream) {
  if (!graph_exec) return LEMO_ERR_ARG;
  return (int)hipGraphLaunch((hipGraphExec_t)graph_exec, S(stream));
}
int lemo_graph_destroy(void* graph_exec) {
  if (!graph_exec) return 0;
  return (int)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
}
#ifdef LEMO_CENSUS
extern "C" int lemo_census_set_pose(unsigned long long*);
extern "C" int lemo_census_set_lbs(unsigned long long*);
extern "C" int lemo_census_set(unsigned long long* buf) { return lemo_census_set_pose(buf) | lemo_census_set_lbs(buf); }
#endif
int lemo_sdf_sample(const float* sdf, int D, int H, int W, const float* pts, int N, const float* gmin, const float* gmax,
                    float* val, float* dval, void* stream) {
  if (!sdf || !pts || !gmin || !gmax || !val) return LEMO_ERR_ARG;
  return sdf_sample(sdf, D, H, W, pts, N, gmin, gmax, val, dval, S(stream));
}

int lemo_reconstruct_global_body(const float* in, int T, int J, double rot_0_pivot, float* out, void* stream) {
  if (!in || !out) return LEMO_ERR_ARG;
  return reconstruct_global_body(in, T, J, rot_0_pivot, out, S(stream));
}
int lemo_reconstruct_global_body_dev(const float* in, int T, int J, const double* rot_0_pivot, float* out, void* stream) {
  if (!in || !out || !rot_0_pivot) return LEMO_ERR_ARG;
  return reconstruct_global_body(in, T, J, 0.0, out, S(stream), rot_0_pivot);
}
int lemo_decode_clip(const float* rec, const float* traj, const double* stats, const double* rot_0_pivot, const float* post, int T,
                     int J, float* contact_lbl, float* markers, void* stream) {
  if (!rec || !traj || !stats || !rot_0_pivot || !contact_lbl || !markers) return LEMO_ERR_ARG;
  return decode_clip(rec, traj, stats, rot_0_pivot, post, T, J, contact_lbl, markers, S(stream));
}
int lemo_local_markers_4chan(const float* body, const float* contact, int T, int M1, float* image, double* rot_0_pivot,
                             void* stream) {
  if (!body || !contact || !image || !rot_0_pivot) return LEMO_ERR_ARG;
  return local_markers_4chan(body, contact, T, M1, image, rot_0_pivot, S(stream));
}

// ------------------------------------------------------------------------------------------------
// fitting engine
// ------------------------------------------------------------------------------------------------
// a replay costs ~8 us of device idle time around the graph (tools/ubench/launch_ubench.hip) on top of its nodes, and every
// graph re-launches the first VPoser layer: as few, as large graphs as the host can enqueue without starving the device.
// Graphs of 1 .. FIT_MAXG iterations are captured on first use (lemo_fit_prepare) and kept.
static const int FIT_MAXG = 20;     // iterations per graph, at most (20: 372.4 vs 374.4 us per iteration with 5 only, same box)

struct FitSide { hipStream_t side = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool pending = false; };      // (see fit_side_launch below)
struct FitEngine {
  lemo_fit_desc d;
  hipGraphExec_t exec[FIT_MAXG + 1] = {};      // exec[k] = k iterations
  int head = 5;     // > 0: a call opens with a 1-iteration and a `head`-iteration graph (LEMO_FIT_HEAD overrides; 0 = largest graphs first)
  FitSide fs;       // side stream + events of the all-vertex side launch (lemo_fit_desc.verts_side); unused otherwise
};

static int fit_iteration(const lemo_fit_desc& d, hipStream_t s, bool first, bool last, FitSide* fs);

static int capture_iterations(FitEngine* e, hipStream_t s, int iters, hipGraphExec_t* out) {
  hipGraph_t g = nullptr;
  CHK((int)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  for (int i = 0; i < iters && !rc; ++i) rc = fit_iteration(e->d, s, i == 0, i == iters - 1, &e->fs);
  const int ec = (int)hipStreamEndCapture(s, &g);
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
  CHK(ec);
  const int ic = (int)hipGraphInstantiate(out, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (!ic) (void)hipGraphUpload(*out, s);       // the first replay then does not pay for the upload (lemo_fit_prepare)
  return ic;
}

// n iterations as replays; launch = false only captures what is missing.
// hipGraphLaunch hands a graph to the queue only after the host has written all of its nodes (measured: a 20-step call that
// opens with the 621-node graph starts ~0.7 ms late, 9 % of the call; later replays are enqueued while the device is busy and
// cost nothing).  A call therefore opens with the 32-node graph (the device starts within ~40 us), then a 5-iteration one
// (enqueued in 0.17 ms while the first runs 0.33 ms), then 20-iteration graphs (0.7 ms of host time against 6.6 ms of device
// time each) and ONE graph for what is left.  Round 3 first used only the 20 / 5 / 1 sizes -- [1 x 5, 5 x 3] for the driver's
// 20-step call: 8 launches where [1, 5, 14] is 3, each launch ~13 us of gap + repeated first layer.
static int fit_graphs(FitEngine* e, hipStream_t s, int n, bool launch) {
  // a hipGraphExec is not bound to the stream it was captured on: the graphs are kept when the caller changes streams
  // (round 2 destroyed and re-captured all three -- ~600 nodes -- per stream change, possibly under a replay still in flight)
  int open[2] = {0, 0}, left = n;
  if (e->head > 0) {
    if (left > 0) { open[0] = 1; left -= 1; }
    if (left > 0) { open[1] = left < e->head ? left : (e->head < FIT_MAXG ? e->head : FIT_MAXG); left -= open[1]; }
  }
  const int nfull = left / FIT_MAXG, rem = left - nfull * FIT_MAXG;
  const int need[4] = {open[0], open[1], nfull ? FIT_MAXG : 0, rem};
  for (int i = 0; i < 4; ++i)
    if (need[i] && !e->exec[need[i]]) CHK(capture_iterations(e, s, need[i], &e->exec[need[i]]));
  if (!launch) return 0;
  for (int i = 0; i < 2; ++i) if (open[i]) CHK((int)hipGraphLaunch(e->exec[open[i]], s));
  for (int i = 0; i < nfull; ++i) CHK((int)hipGraphLaunch(e->exec[FIT_MAXG], s));
  if (rem) CHK((int)hipGraphLaunch(e->exec[rem], s));
  return 0;
}

// arguments of the fused tail launch (kernels.hpp FitTail) from the descriptor
static FitTail fit_tail_args(const lemo_fit_desc& d, bool dz, bool adam, bool h1) {
  FitTail a{};
  a.w1 = d.vposer.w1; a.w1t = d.vposer.w1t; a.b1 = d.vposer.b1;
  a.dh1 = vposer_scratch_dh1(d.vp_scratch, d.B);                      // kernels.hpp: the layout vposer_mlp_bwd writes
  a.g_other = d.g_other; a.h1 = h1 ? d.h1 : nullptr;
  a.transl = d.transl; a.rot6d = d.rot6d; a.other = d.other; a.g_transl = d.g_transl; a.g_rot6d = d.g_rot6d;
  a.m0 = d.adam_m[0]; a.v0 = d.adam_v[0]; a.m1 = d.adam_m[1]; a.v1 = d.adam_v[1]; a.m2 = d.adam_m[2]; a.v2 = d.adam_v[2];
  a.weights = d.weights; a.step_ctr = d.step_ctr; a.step_cur = d.step_cur;
  a.lr0 = lr_decimal(d.lr0); a.lr1 = lr_decimal(d.lr1); a.lr_switch = d.lr_switch; a.lr2 = lr_decimal(d.lr2); a.lr_switch2 = d.lr_switch2;
  a.snap = d.snap; a.nonfinite = d.nonfinite; a.losses = d.losses;
  a.B = d.B; a.do_dz = dz; a.do_adam = adam;
  a.Bn = d.per_frame ? 1 : d.B;       // per_frame: every row is a fit of its own (B > 1 = several clips' frames in lockstep)
  return a;
}

// The all-vertex forward off the critical path (lemo_fit_desc.verts_side, round 6).  Only the 253 vertices of the set U feed the losses; the
// other 10222 rows of the reference's `vertices` output have no consumer inside the iteration.  The launch that regresses them (30 us on all
// 250 CUs, a tenth of the iteration) is issued on the engine's own side stream behind the encoder's backward tail, on 125 workgroups of two
// tiles each, so that it runs BESIDE the eight per-frame launches that close the iteration (one workgroup per frame: 119 of 256 CUs, ~56 us)
// instead of in front of the encoder; it is joined before the next iteration's pose kernel overwrites its operands (Xg, XgS, A; the
// translation it adds is the copy the set-U forward left in transl_side, because the Adam launch rewrites `transl` meanwhile), and at the
// end of every run / captured graph.  Round 5 measured the same idea with the 250-workgroup launch and lost 3-8 %: its one-per-CU
// workgroups took the CUs the per-frame launches needed (DESIGN_HISTORY 11.7).
// MEASURED (round 6, profiles/r06_side_forward.txt): bit-identical vertices, the launches DO overlap (side launch 50.9 us beside the tail's
// 68 us) -- and the iteration is 333 us instead of 309 (3001-3016 against 3222-3254 it/s; 3507 without the all-vertex forward at all): the
// graph runs the two branches on two hardware queues and every cross-queue edge costs ~6-12 us of idle time on the branch that waits (fork:
// the per-frame chain starts 12 us behind the encoder's tail; join: the next pose kernel 10 us behind the last GEMM although the side launch
// had finished 24 us earlier), while the per-frame launches run 15 us longer beside it.  Two edges per iteration cost what the overlap
// buys.  The path is kept -- tested, selectable (AmassTemporalFitter(side_full_forward=True), LEMO_SIDE_FULL_FORWARD=1) -- and OFF.
#define FIT_SIDE_BLOCKS 125
static int fit_side_launch(const lemo_fit_desc& d, hipStream_t s) {      // the launch itself, on whatever stream
  static const int blocks = getenv("LEMO_FIT_SIDE_BLOCKS") ? atoi(getenv("LEMO_FIT_SIDE_BLOCKS")) : FIT_SIDE_BLOCKS;      // A/B knob
  return lbs_verts_fwd(d.skin, d.pose.Xg, d.Bp, d.pose.A, d.body.nj, d.transl_side, nullptr, d.V, d.B, d.verts_side, nullptr, s, nullptr, d.pose.XgS,
                       blocks);
}
static int fit_side_join(FitSide* fs, hipStream_t s) {
  if (!fs || !fs->pending) return 0;
  fs->pending = false;
  return fs->side ? (int)hipStreamWaitEvent(s, fs->join, 0) : 0;
}

// compute_h1: the first VPoser layer is launched here (a bare forward, or the first iteration of a graph / call); inside
// a run of iterations the previous iteration's tail launch has already produced it from the updated latent
// `stages` (diagnostics, lemo_fit_census): bit 0 VPoser + pose stage, 1 vertex stage, 2 marker image + encoder forward, 3 losses
// fs: the side-branch context of a run of iterations (null: a bare forward -- the all-vertex forward of verts_side then runs in line)
static int fit_forward(const lemo_fit_desc& d, hipStream_t s, bool finalize, bool compute_h1 = true, unsigned stages = ~0u, FitSide* fs = nullptr) {
  const int B = d.B, nj = d.body.nj;
  const int H = 3 * d.fit.n81 + 2, W = B - 1 + 16;
  // VPoser MLP (3 MFMA GEMMs); its rotation head and the 6-D -> axis-angle conversion of the global
  // orientation are fused into the pose-stage kernel, which also zeroes the loss accumulators and
  // latches the step counter for this iteration.
  if (stages & 1u) {
  if (compute_h1) CHK(fit_tail(fit_tail_args(d, false, false, true), s));
  CHK(gemm_nt16(d.vposer.w2, 512, d.h1, 512, 512, B, 512, d.h2, 512, d.vposer.b2, nullptr, 0, 1, s));
  CHK(gemm_nt16(d.vposer.w3, 512, d.h2, 512, 128, B, 512, d.vo, 128, d.vposer.b3, nullptr, 0, 2, s));
  lemo_pose_in in{};
  in.rot6d = d.rot6d; in.vposer_o = d.vo; in.go_out = d.go_aa;
  in.lh = d.other + 32; in.rh = d.other + 44; in.hand_stride = 56;
  in.betas = d.shape; in.betas_stride = 10;
  in.zero_f64 = d.loss_acc; in.n_zero = 512; in.step_ctr = d.step_ctr; in.step_cur = d.step_cur;
  in.nonfinite = d.nonfinite;
  CHK(fit_side_join(fs, s));                               // the previous iteration's side launch still reads Xg / XgS / A
  CHK(smplx_pose_fwd(d.body, in, d.pose, B, s));
  }
  if (stages & 2u) {
  if (d.full_vertices) CHK(lbs_verts_fwd(d.skin, d.pose.Xg, d.Bp, d.pose.A, nj, d.transl, nullptr, d.V, B, d.verts, d.v_posed, s, nullptr, d.pose.XgS));
  else if (d.uset.DkT && d.uset.n == d.fit.n)      // the loss-carrying set IS the backward set U (same order): small-set path
    CHK(lbs_verts_fwd_active(d.skin, d.uset, d.pose.Xg, d.Bp, d.pose.A, nj, d.transl, B, d.dvp, d.verts, d.v_posed, s, d.verts_side ? d.transl_side : nullptr));
  else {
    CHK(lbs_verts_fwd(d.skin, d.pose.Xg, d.Bp, d.pose.A, nj, d.transl, d.fwd_ids, d.fit.n, B, d.verts, d.v_posed, s, nullptr, d.pose.XgS));
    if (d.verts_side && !d.full_vertices) CHK((int)hipMemcpyAsync(d.transl_side, d.transl, sizeof(float) * 3 * B, hipMemcpyDeviceToDevice, s));
  }
  if (d.verts_side && !d.full_vertices && !fs) CHK(fit_side_launch(d, s));      // bare forward: in line
  }
  if (d.per_frame) {       // opt_amass_perframe.py:324-351: marker L1 + the three L2 priors, nothing temporal
    CHK(vertex_loss_accumulate(d.fit, d.verts, d.nrows, d.target, d.contact, d.shape, d.other, B, d.loss_acc, s));
    if (finalize) CHK(loss_finalize(d.loss_acc, B, d.fit.n67, 1.0, d.weights, d.losses, s));
    return 0;
  }
  // marker image + first encoder layer in one launch (x0 is still written: parity tests read it)
  if (stages & 4u) {
  if (enc_fused_head3(d))
    CHK(enc_head3(d.fit, d.verts, d.nrows, d.pose.Jtr, nj, d.transl, B, d.enc_w[0], d.enc_b[0], d.enc_w3[1], d.enc_w3_inv[1], d.enc_b[1], d.enc_w3[2],
                  d.enc_w3_inv[2], d.enc_b[2], d.x0, d.canon, d.act[1], d.act[2], d.act[3], s));
  else if (enc_fused_head(d))
    CHK(enc_head(d.fit, d.verts, d.nrows, d.pose.Jtr, nj, d.transl, B, d.enc_w[0], d.enc_b[0], d.enc_w3[1], d.enc_w3_inv[1], d.enc_b[1], d.x0, d.canon,
                 d.act[1], d.act[2], s));
  else
    CHK(marker_c1(d.fit, d.verts, d.nrows, d.pose.Jtr, nj, d.transl, B, d.enc_w[0], d.enc_b[0], d.x0, d.canon, d.act[1], d.enc_ch[1], s));
  CHK(enc_chain_fwd(d, H, W, s, enc_fused_head3(d) ? 3 : (enc_fused_head(d) ? 2 : 1)));
  }
  const double cnt = (double)d.enc_ch[10] * H * (W - 1);
  const float coef2 = (float)((double)d.weights_host[5] * 2.0 / cnt);
  if (stages & 8u)
  CHK(fit_losses(d.act[10], d.dact[0], H, W, d.enc_ch[10], coef2, d.loss_acc + 9, d.fit, d.verts, d.nrows, d.target, d.contact,
                 d.shape, d.other, B, d.loss_acc, s));
  if (finalize) CHK(loss_finalize(d.loss_acc, B, d.fit.n67, cnt, d.weights, d.losses, s));
  return 0;
}

// update = false: gradients only (lemo_fit_backward).  update = true: the tail launch also runs Adam and, with next_h1,
// the first VPoser layer of the next iteration.
// `stages`: bit 4 encoder backward-data + first-layer adjoint, 5 vertex-stage backward, 6 pose / VPoser backward + tail
static int fit_backward(const lemo_fit_desc& d, hipStream_t s, bool update = false, bool next_h1 = false, unsigned stages = ~0u, FitSide* fs = nullptr) {
  const int B = d.B, nj = d.body.nj;
  const int H = 3 * d.fit.n81 + 2, W = B - 1 + 16;
  const double cnt = d.per_frame ? 1.0 : (double)d.enc_ch[10] * H * (W - 1);
  int cur = 0;
  if (d.per_frame || !(stages & 16u)) goto vertex_stage;           // per_frame: no encoder (d.fit.u_m81 is all -1, dx0 is never read)
  CHK(enc_chain_bwd(d, H, W, s, &cur, enc_bwd_l_last(d)));          // d(pre-act of layer 10) -> ... -> d(pre-act of layer 1)
  CHK(enc_bwd_tail(d, cur, H, W, s));
  if (fs && d.verts_side && !d.full_vertices) {            // fork: the all-vertex forward beside the per-frame launches below
    if (fs->side) {
      CHK((int)hipEventRecord(fs->fork, s));
      CHK((int)hipStreamWaitEvent(fs->side, fs->fork, 0));
      CHK(fit_side_launch(d, fs->side));
      CHK((int)hipEventRecord(fs->join, fs->side));
    } else CHK(fit_side_launch(d, s));                     // (host emulation: no second stream)
    fs->pending = true;
  }
vertex_stage:
  if (!(stages & 32u)) goto pose_stage;
  if (lbs_verts_bwd_fusable(d.skin, d.uset, nj) && d.fit.n == d.uset.n) {
    // d(total)/d(verts) is computed inside the LBS backward (block per frame in both): one launch instead of two
    const FitFuse ff{d.fit, DvertsIn{d.verts, d.nrows, d.target, d.contact, d.dx0, d.canon, d.weights, B, d.per_frame ? 1 : B}, d.loss_acc, cnt, d.losses};
    CHK(lbs_verts_bwd(d.skin, d.uset, d.pose.A, nj, d.v_posed, d.nrows, nullptr, B, d.Bp, d.dvp, d.dA, d.g_transl, d.dX, s, &ff));
  } else {
    CHK(dverts_assemble(d.fit, d.verts, d.nrows, d.target, d.contact, d.dx0, d.canon, d.weights, d.loss_acc, cnt, d.losses, B, d.dverts, s, d.per_frame ? 1 : B));
    CHK(lbs_verts_bwd(d.skin, d.uset, d.pose.A, nj, d.v_posed, d.nrows, d.dverts, B, d.Bp, d.dvp, d.dA, d.g_transl, d.dX, s));
  }
pose_stage:
  if (!(stages & 64u)) return 0;
  lemo_pose_grad_in gi{d.dA, nullptr, d.dX};
  lemo_pose_grad_out go{};
  go.d_lh = d.g_other + 32; go.d_rh = d.g_other + 44; go.hand_stride = 56;
  go.rot6d = d.rot6d; go.d_rot6d = d.g_rot6d;            // d(global_orient) -> d(rot6d), fused
  go.vposer_o = d.vo; go.d_vposer_o = d.vp_scratch;      // d(body_pose) -> d(VPoser out layer), fused
  CHK(smplx_pose_bwd(d.body, d.pose, gi, go, B, s));
  CHK(vposer_mlp_bwd(d.vposer, d.h1, d.h2, B, nullptr, 56, d.vp_scratch, s));      // dh2, dh1; the last layer is in the tail
  CHK(fit_tail(fit_tail_args(d, true, update, update && next_h1), s));
  return 0;
}

// first / last: position inside the run of iterations issued together (one graph, or one eager call)
static int fit_iteration(const lemo_fit_desc& d, hipStream_t s, bool first, bool last, FitSide* fs) {
  const bool side = d.verts_side && !d.full_vertices && !d.per_frame;
  CHK(fit_forward(d, s, false, first, ~0u, side ? fs : nullptr));
  if (side && !fs) return LEMO_ERR_STATE;
  CHK(fit_backward(d, s, true, !last, ~0u, side ? fs : nullptr));
  if (last) CHK(fit_side_join(fs, s));                     // a run / a captured graph ends with everything on the caller's stream
  return 0;
}

void* lemo_fit_create(const lemo_fit_desc* d) {
  if (!d || d->B < (d->per_frame ? 1 : 10) || d->B > d->Bp || !d->verts || !d->transl) return nullptr;
  if (conv_lds_init() || conv_split_init() || lbs_init()) return nullptr;
  if (d->pose.XgS && (d->skin.DgH != nullptr) != (d->pose.xgs_f16 != 0)) return nullptr;     // both operands of the blend GEMM in one form
  FitEngine* e = new (std::nothrow) FitEngine();
  if (e) {
    e->d = *d;
    // stage 1 fits frame after frame on one stream: the device is never idle when a call starts, so the small head graphs
    // (there to get an idle device going within ~40 us) only cost host time -- 12 hipGraphLaunch per 100-step frame instead of 5,
    // and the host thread is what limits several clips in lockstep (tools/perframe_concurrent.py)
    if (d->per_frame) e->head = 0;
    if (const char* h = getenv("LEMO_FIT_HEAD")) e->head = atoi(h);       // diagnostics: A/B of the replay schedule
    if (d->verts_side && !d->full_vertices) {
      if (!d->transl_side || d->per_frame) { delete e; return nullptr; }
      // (the host emulation hands back a null stream: FitSide::side == nullptr = "no second stream", the launch then runs in line)
      // (default priority: on the LOWEST priority the captured branch starves -- 1930 instead of 3000 it/s, profiles/r06_side_forward.txt)
      if (hipStreamCreateWithPriority(&e->fs.side, hipStreamNonBlocking, getenv("LEMO_FIT_SIDE_PRIO") ? atoi(getenv("LEMO_FIT_SIDE_PRIO")) : 0) != hipSuccess ||
          hipEventCreateWithFlags(&e->fs.fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&e->fs.join, hipEventDisableTiming) != hipSuccess) { lemo_fit_destroy(e); return nullptr; }
    }
  }
  return e;
}

void lemo_fit_destroy(void* h) {
  FitEngine* e = (FitEngine*)h;
  if (!e) return;
  for (int l = 0; l <= FIT_MAXG; ++l) if (e->exec[l]) (void)hipGraphExecDestroy(e->exec[l]);
  if (e->fs.fork) (void)hipEventDestroy(e->fs.fork);
  if (e->fs.join) (void)hipEventDestroy(e->fs.join);
  if (e->fs.side) { (void)hipStreamSynchronize(e->fs.side); (void)hipStreamDestroy(e->fs.side); }
  delete e;
}

int lemo_fit_forward(void* h, void* stream) {
  FitEngine* e = (FitEngine*)h;
  if (!e) return LEMO_ERR_ARG;
  return fit_forward(e->d, S(stream), true);
}

int lemo_fit_backward(void* h, void* stream) {       // after lemo_fit_forward: gradients only, no Adam
  FitEngine* e = (FitEngine*)h;
  if (!e) return LEMO_ERR_ARG;
  return fit_backward(e->d, S(stream));
}

int lemo_fit_step(void* h, int n, int use_graph, void* stream) {
  FitEngine* e = (FitEngine*)h;
  if (!e || n < 0) return LEMO_ERR_ARG;
  hipStream_t s = S(stream);
  if (!use_graph) {
    for (int i = 0; i < n; ++i) CHK(fit_iteration(e->d, s, i == 0, i == n - 1, &e->fs));
    return 0;
  }
  CHK(fit_graphs(e, s, n, false));
  return fit_graphs(e, s, n, true);
}

int lemo_fit_prepare(void* h, int n, void* stream) {
  FitEngine* e = (FitEngine*)h;
  if (!e || n < 0) return LEMO_ERR_ARG;
  return fit_graphs(e, S(stream), n, false);
}

// Diagnostics: where an iteration's time goes, stage by stage.  Every stage (and the whole forward + backward) is captured `reps`
// times back to back into its own graph and replayed between two events -- consecutive graph nodes, like inside the iteration.
// Buffers are left in a consistent state by a clean forward + backward at the end.  SYNCHRONISES (events): not for the hot path.
int lemo_fit_census(void* h, int reps, float* ms_out, void* stream) {
  FitEngine* e = (FitEngine*)h;
  hipStream_t s = S(stream);
  if (!e || !ms_out || reps < 1 || !s) return LEMO_ERR_ARG;
  const lemo_fit_desc& d = e->d;
  CHK(fit_forward(d, s, true));
  CHK(fit_backward(d, s));
  hipEvent_t e0, e1;
  CHK((int)hipEventCreate(&e0));
  CHK((int)hipEventCreate(&e1));
  int rc = 0;
  for (int k = 0; k <= LEMO_FIT_NSTAGE && !rc; ++k) {           // k == LEMO_FIT_NSTAGE: the whole forward + backward (no update)
    const unsigned mask = k == LEMO_FIT_NSTAGE ? ~0u : (1u << k);
    hipGraph_t g = nullptr;
    hipGraphExec_t x = nullptr;
    rc = (int)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int r = 0; r < reps && !rc; ++r) {
      if (mask & 15u) rc = fit_forward(d, s, false, true, mask);
      if (!rc && (mask & 112u)) rc = fit_backward(d, s, false, false, mask);
    }
    const int ec = (int)hipStreamEndCapture(s, &g);
    if (!rc) rc = ec;
    if (!rc) rc = (int)hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (g) (void)hipGraphDestroy(g);
    if (!rc) {
      rc = (int)hipGraphLaunch(x, s);                              // once untimed (upload, caches), once timed
      if (!rc) rc = (int)hipEventRecord(e0, s);
      if (!rc) rc = (int)hipGraphLaunch(x, s);
      if (!rc) rc = (int)hipEventRecord(e1, s);
      if (!rc) rc = (int)hipEventSynchronize(e1);
      float ms = 0.f;
      if (!rc) rc = (int)hipEventElapsedTime(&ms, e0, e1);
      ms_out[k] = ms / (float)reps;
    }
    if (x) (void)hipGraphExecDestroy(x);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (!rc) rc = fit_forward(d, s, true);
  if (!rc) rc = fit_backward(d, s);
  return rc;
}

// engine <-> caller copies of the optimiser state (parameters, Adam moments, completed-step count): ONE launch
static int fit_state_io(FitEngine* e, const lemo_fit_state* st, bool load, hipStream_t s) {
  if (!e || !st || !st->transl || !st->rot6d || !st->other || !st->step) return LEMO_ERR_ARG;
  const lemo_fit_desc& d = e->d;
  if (!d.rot6d || !d.other || !d.step_ctr) return LEMO_ERR_STATE;
  StateCopy a{};
  float* eng[9] = {d.transl, d.rot6d, d.other, d.adam_m[0], d.adam_m[1], d.adam_m[2], d.adam_v[0], d.adam_v[1], d.adam_v[2]};
  float* usr[9] = {st->transl, st->rot6d, st->other, st->adam_m[0], st->adam_m[1], st->adam_m[2], st->adam_v[0], st->adam_v[1], st->adam_v[2]};
  const int width[3] = {3, 6, 56};
  for (int i = 0; i < 9; ++i) {
    if (!eng[i] || !usr[i]) return LEMO_ERR_ARG;
    a.src[i] = load ? usr[i] : eng[i];
    a.dst[i] = load ? eng[i] : usr[i];
    a.n[i] = d.B * width[i % 3];
  }
  a.njobs = 9;
  a.step_src = load ? st->step : d.step_ctr;
  a.step_dst = load ? d.step_ctr : st->step;
  a.nonfinite = load ? d.nonfinite : nullptr;
  return state_copy(a, s);
}
int lemo_fit_load_state(void* h, const lemo_fit_state* st, void* stream) { return fit_state_io((FitEngine*)h, st, true, S(stream)); }
int lemo_fit_save_state(void* h, const lemo_fit_state* st, void* stream) { return fit_state_io((FitEngine*)h, st, false, S(stream)); }

}  // extern "C"
