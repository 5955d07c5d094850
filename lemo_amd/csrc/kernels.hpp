// Internal (C++) launcher declarations shared by the kernel translation units and the C-ABI layer.
// Every launcher enqueues on `s`, never allocates, never synchronises, and returns 0 or an error.
#pragma once
#include "common.hpp"
#include "lemo_hip.h"

#include <cmath>

namespace lemo {

// The reference writes its learning rates as decimal literals (0.01, 0.005, 0.1, 0.003, 3e-6: Python doubles) and torch divides
// THAT double by the bias correction; the C ABI carries them as float.  The shortest decimal (<= 6 significant digits) that
// rounds to the given float is that literal again; a float that is no short decimal is taken as it is.
static inline double lr_decimal(float lr) {
  if (!(lr > 0.f) || !std::isfinite(lr)) return (double)lr;
  const double k = std::pow(10.0, 5.0 - std::floor(std::log10((double)lr)));
  const double d = std::nearbyint((double)lr * k) / k;
  return (float)d == lr ? d : (double)lr;
}

// ---------------- conv_kernels.hip ----------------
int conv3x3_mfma(const float* in, const float* wt, const float* bias, const float* aux, float* out,
                 int H, int W, int cin, int cout, int epi, int variant, hipStream_t s);
int conv3x3_mfma_splitk(const float* in, const float* wt, const float* bias, const float* aux, float* out, float* partial, int ks,
                        int H, int W, int cin, int cout, int epi, hipStream_t s);
int conv3x3_mfma_lds(const float* in, const float* wt, const float* wt2, const float* bias, const float* aux, float* out,
                     int H, int W, int cin, int cout, int epi, hipStream_t s, unsigned long long* dbg = nullptr);
int conv_lds_init();
// ---------------- conv_split_kernels.hip ----------------
// variant 3: fp32-exact 64->64 conv on the bf16 matrix cores (3-way bf16 operand split, 6 products)
bool conv3x3_split_supported(int H, int W, int cin, int cout);
int conv3x3_mfma_split(const float* in, const void* w3, const float* wt, const float* bias, const float* aux, float* out,
                       int H, int W, int cin, int cout, int epi, hipStream_t s, unsigned long long* dbg = nullptr, int pieces = 3,
                       float winv = 1.f);
int conv_split_init();
// ---------------- conv_pair_kernels.hip ----------------
// variant 5: TWO 64 -> 64 layers per launch on 10 x 14 tiles, intermediate in LDS (split-f16 arithmetic of variant 4)
bool conv3x3_pair_supported(int H, int W, int c0, int c1, int c2);
int conv3x3_pair_f16(const float* in, const void* wA, float winvA, const float* biasA, const float* auxA, float* mid, const void* wB,
                     float winvB, const float* biasB, const float* auxB, float* out, int H, int W, int epi, hipStream_t s,
                     unsigned long long* dbg = nullptr);
// ---------------- conv_wino_kernels.hip ----------------
// variant 10: ONE 64 -> 64 layer as a Winograd F(2x2, 3x3) convolution, its 16 GEMMs in the split-f16 arithmetic of variant 4
bool conv3x3_wino_supported(int H, int W, int cin, int cout);
int conv3x3_wino_f16(const float* in, const void* wU, float winv, const float* wt, const float* bias, const float* aux, float* out,
                     int H, int W, int epi, hipStream_t s, unsigned long long* dbg = nullptr);
int conv3x3_c1(const float* x0, const float* w, const float* bias, float* out, int H, int W, int cout, hipStream_t s);
int conv3x3_c1_bwd(const float* dpre, const float* w, float* dx0, int H, int W, int cout, hipStream_t s);
int smooth_loss_blocks(int H, int W, int C);
int smooth_loss(const float* z, float* dpre, float* partial, int H, int W, int C, float coef2, hipStream_t s,
                double* acc = nullptr);

// ---------------- gemm_kernels.hip ----------------
// C[n][m] = epi(sum_k A[m][k] B[n][k]); epi 0 none | 1 lrelu(v+bias[m]) | 2 v+bias[m] | 3 v*lrelu'(aux[n][m])
int gemm_nt16(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc,
              const float* bias, const float* aux, int ldaux, int epi, hipStream_t s);

// long-K, few-outputs form (N <= 128, M % 64 == 0): S K-slabs, partial tiles in `part` (gemm_nt16_splitk_part_floats), fixed-order sum
int gemm_nt16_splitk_part_floats(int M, int S);
int gemm_nt16_splitk_partials(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* part, int S, hipStream_t s,
                              const float* A_grouped = nullptr);
int gemm_nt16_splitk(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc, float* part, int S,
                     hipStream_t s, const float* A_grouped = nullptr);   // A_grouped: A again as [K/16][M][16]
int gemm_nt16_kg8(const float* A, int lda, const float* Xg, int rows, int M, int N, int K, float* C, int ldc, hipStream_t s);

// ---------------- pose_kernels.hip ----------------
typedef lemo_vposer_w VPoserW;
typedef lemo_body_const BodyConst;
typedef lemo_pose_in PoseIn;
typedef lemo_pose_ws PoseWs;
typedef lemo_pose_grad_in PoseGradIn;
typedef lemo_pose_grad_out PoseGradOut;
typedef lemo_skin_const SkinConst;
typedef lemo_vertex_set_bwd VertexSetBwd;
typedef lemo_fit_const FitConst;
// inputs of d(total)/d(verts) (loss_device.hpp::dverts_vertex)
struct DvertsIn {
  const float* verts; int nrows; const float* target; const float* contact; const float* dx0; const float* canon;
  const float* weights; int B;
  int Bn;       // frames the marker mean runs over: B, or 1 when every row is a fit of its own (lemo_fit_desc.per_frame)
};
// the fitting engine's LBS backward computes d(verts) itself instead of reading it (one launch less per iteration)
struct FitFuse { FitConst fc; DvertsIn in; const double* acc; double smooth_count; float* losses_out; };

// z: [B] rows with stride z_stride floats.  Saves h1,h2 [B][512], o [B][128] for backward.
int vposer_decode_fwd(const VPoserW& w, const float* z, int z_stride, int B, float* h1, float* h2, float* o,
                      float* matrot /*[B][21][9] or null*/, float* aa /*[B][63] or null*/, hipStream_t s);
// d_aa [B][63] and/or d_matrot [B][21][9] (either may be null) -> dz [B] rows with stride dz_stride
int vposer_decode_bwd(const VPoserW& w, const float* h1, const float* h2, const float* o, const float* matrot,
                      const float* d_aa, const float* d_matrot, int B, float* dz, int dz_stride, float* scratch /*[B][1152]*/,
                      hipStream_t s);
// layout of the VPoser backward's scratch [B][1152]: d(out) [B][128] | d(h2) [B][512] | d(h1) [B][512].  The fused tail launches of the
// two engines (fit_tail, prox_tail) read d(h1) out of it: they and vposer_mlp_bwd share THESE helpers.
constexpr int VP_HIDDEN = 512, VP_OUT_PAD = 128;
static inline float* vposer_scratch_dout(float* scratch, int) { return scratch; }
static inline float* vposer_scratch_dh2(float* scratch, int B) { return scratch + (size_t)B * VP_OUT_PAD; }
static inline float* vposer_scratch_dh1(float* scratch, int B) { return scratch + (size_t)B * (VP_OUT_PAD + VP_HIDDEN); }
int vposer_mlp_bwd(const VPoserW& w, const float* h1, const float* h2, int B, float* dz, int dz_stride, float* scratch,
                   hipStream_t s);
int rot6d_to_aa_fwd(const float* x6, int stride, int N, float* aa, hipStream_t s);
int rot6d_to_aa_bwd(const float* x6, int stride, const float* d_aa, int N, float* dx6, hipStream_t s);
int smplx_pose_fwd(const BodyConst& c, const PoseIn& in, const PoseWs& ws, int B, hipStream_t s);
int smplx_pose_bwd(const BodyConst& c, const PoseWs& ws, const PoseGradIn& gi, const PoseGradOut& go, int B, hipStream_t s);

// ---------------- lbs_kernels.hip ----------------
int lbs_init();
// verts[b][slot] for slot < n ; ids == null => slot == vertex id, n == V
int lbs_verts_fwd_active(const SkinConst& c, const VertexSetBwd& u, const float* Xg, int Bp, const float* A, int nj,
                         const float* transl, int B, float* blend, float* verts, float* v_posed, hipStream_t s, float* transl_copy = nullptr);
int lbs_verts_fwd(const SkinConst& c, const float* Xg, int Bp, const float* A, int nj, const float* transl,
                  const int* ids, int n, int B, float* verts, float* v_posed, hipStream_t s, unsigned long long* dbg = nullptr,
                  const unsigned short* XgS = nullptr, int max_blocks_x = 0 /* > 0: at most this many workgroups, each looping over its tiles */);
bool lbs_verts_bwd_fusable(const SkinConst& c, const VertexSetBwd& u, int nj);
int lbs_verts_bwd(const SkinConst& c, const VertexSetBwd& u, const float* A, int nj, const float* v_posed, int vp_rows,
                  const float* dverts /*[B][n][3]*/, int B, int Bp, float* dvp /*[B][NCs] scratch*/,
                  float* dA /*[B][nj][12]*/, float* dtransl /*[B][3] or null*/, float* dX /*[B][512]*/, hipStream_t s, const FitFuse* fuse = nullptr);
int joints_assemble(const float* Jtr, int nj, const float* verts, int vrows, const int* extra_rows, int n_extra,
                    const int* lmk_rows /*[n_lmk][3]*/, const float* lmk_bary, int n_lmk, const float* transl,
                    int B, float* joints /*[B][nj+n_extra+n_lmk][3]*/, hipStream_t s);

// ---------------- scene_kernels.hip ----------------
int sdf_sample(const float* sdf, int D, int H, int W, const float* pts, int N, const float* gmin /*host[3]*/,
               const float* gmax /*host[3]*/, float* val, float* dval, hipStream_t s);

// ---------------- ae_kernels.hip ----------------
// nclip / cs (AE step engine): the same launch for `nclip` clips whose buffers lie `cs` floats apart (clip = blockIdx.y)
int maxpool3s2_fwd(const float* in, int H, int W, float* out, unsigned char* idx, int C, hipStream_t s, int nclip = 1, size_t cs = 0);
int maxpool3s2_bwd(const float* dout, const unsigned char* idx, const float* act, float* din, int H, int W, int C, hipStream_t s,
                   int nclip = 1, size_t cs = 0);
int stuff2_fwd(const float* in, int h, int w, float* out, int H, int W, int C, hipStream_t s, int nclip = 1, size_t cs = 0);
int stuff2_bwd(const float* dout, int H, int W, const float* act, float* din, int h, int w, int C, hipStream_t s);
int conv3x3_wgrad_nslab(int H, int W);
int conv3x3_wgrad_partial(const float* dy, const float* x, int H, int W, int cin, int cout, float* partial, hipStream_t s);
int conv3x3_wgrad_reduce_multi(const lemo_wgrad_job* jobs, int n, hipStream_t s);
int conv3x3_wgrad(const float* dy, const float* x, int H, int W, int cin, int cout, int cin_real, int cout_real,
                  float* partial, float* dw, float* db, hipStream_t s);
int adam_flat(float* p, const float* g, float* m, float* v, int n, float lr, int step, int* step_dev, hipStream_t s);

// ---------------- loss_kernels.hip ----------------
int marker_c1(const FitConst& fc, const float* verts, int nrows, const float* Jtr, int nj, const float* transl, int B,
              const float* w, const float* bias, float* x0, float* canon, float* out, int cout, hipStream_t s);
// ---------------- conv_head_kernels.hip (conv variant 7: variant 5 + these) ----------------
// marker image + layer 0 (1 -> 32, fp32 FMAs) + layer 1 (32 -> 32, split-f16 MFMA) in one launch / its adjoint in one launch
int enc_head(const FitConst& fc, const float* verts, int nrows, const float* Jtr, int nj, const float* transl, int B, const float* w0,
             const float* b0, const void* w1pack, float w1inv, const float* b1, float* x0, float* canon, float* act1, float* act2,
             hipStream_t s);
int enc_head3(const FitConst& fc, const float* verts, int nrows, const float* Jtr, int nj, const float* transl, int B, const float* w0,
              const float* b0, const void* w1pack, float w1inv, const float* b1, const void* w2pack, float w2inv, const float* b2,
              float* x0, float* canon, float* act1, float* act2, float* act3, hipStream_t s);
int enc_tail(const float* din, const void* w1bpack, float w1binv, const float* act1, const float* w0, float* dx0, int H, int W, hipStream_t s);
int enc_tail3(const float* din, const void* w2bpack, float w2binv, const float* act2, const void* w1bpack, float w1binv, const float* act1,
              const float* w0, float* dx0, int H, int W, hipStream_t s);
int marker_feature(const FitConst& fc, const float* verts, int nrows, const float* Jtr, int nj, const float* transl, int B,
                   float* x0, float* canon, hipStream_t s);
// acc (f64[16], zeroed at the start of the iteration): [0] marker L1 sum, [1..4] contact sums, [5..8] contact
// counts, [9] smoothness sum of squares, [10] sum z^2, [11] sum betas^2, [12] sum hands^2
int fit_losses(const float* z, float* dpre, int H, int W, int C, float coef2, double* sm_acc, const FitConst& fc, const float* verts,
               int nrows, const float* target, const float* contact, const float* shape, const float* other, int B, double* acc,
               hipStream_t s);
int vertex_loss_accumulate(const FitConst& fc, const float* verts, int nrows, const float* target, const float* contact,
                           const float* shape, const float* other, int B, double* acc, hipStream_t s);
int loss_finalize(const double* acc, int B, int n67, double smooth_count, const float* weights, float* losses, hipStream_t s);
// finalises the losses from `acc` in its prologue (block 0 publishes them to `losses`)
int dverts_assemble(const FitConst& fc, const float* verts, int nrows, const float* target, const float* contact,
                    const float* dx0, const float* canon, const float* weights, const double* acc, double smooth_count,
                    float* losses, int B, float* dverts, hipStream_t s, int Bn = 0);
// arguments of fit_tail_kernel (loss_kernels.hip): last VPoser-backward layer + Adam + first VPoser-forward layer of the
// next iteration, one workgroup per frame
struct FitTail {
  const float *w1, *w1t, *b1;      // bodyprior_dec_fc1 [512][32], its transpose [32][512], bias
  const float* dh1;                // [B][512]
  float* g_other;                  // [B][56]: dz goes to columns 0..31
  float* h1;                       // [B][512] out (nullptr: no forward layer)
  float *transl, *rot6d, *other;
  const float *g_transl, *g_rot6d;
  float *m0, *v0, *m1, *v1, *m2, *v2;
  const float* weights;
  int* step_ctr;
  const int* step_cur;
  double lr0, lr1;                  // decimal learning rates (lr_decimal of the descriptor's floats)
  int lr_switch;
  double lr2;
  int lr_switch2;
  float* snap;
  int* nonfinite;
  const float* losses;
  int B, do_dz, do_adam;
  int Bn;                          // frames the prior means run over (B; 1 with lemo_fit_desc.per_frame: independent rows)
};
int fit_tail(const FitTail& a, hipStream_t s);
int adam_step(float* transl, const float* g_transl, float* m0, float* v0, float* rot6d, const float* g_rot, float* m1,
              float* v1, float* other, const float* g_other, float* m2, float* v2, int B, const float* weights,
              int* step_ctr, const int* step_cur, float lr0, float lr1, int lr_switch, hipStream_t s, float lr2 = 0.f,
              int lr_switch2 = 0, float* snap = nullptr, int* nonfinite = nullptr, const float* losses = nullptr);

// optimiser-state hand-over: flat device-to-device copies + step counter in one launch (lemo_*_load_state / save_state)
static const int STATE_MAX_JOBS = 12;
struct StateCopy {
  const float* src[STATE_MAX_JOBS];
  float* dst[STATE_MAX_JOBS];
  int n[STATE_MAX_JOBS];
  int njobs;
  const int* step_src;
  int* step_dst;
  int* nonfinite;                  // [2] zeroed when non-null (load)
};
int state_copy(const StateCopy& a, hipStream_t s);

// ---------------- marker_kernels.hip ----------------
int reconstruct_global_body(const float* in, int T, int J, double rot0, float* out, hipStream_t s, const double* rot0_dev = nullptr);
int local_markers_4chan(const float* body, const float* contact, int T, int M1, float* image, double* rot0, hipStream_t s);
int decode_clip(const float* rec, const float* traj, const double* stats, const double* rot0, const float* post, int T, int J,
                float* lbl, float* markers, hipStream_t s);

}  // namespace lemo
