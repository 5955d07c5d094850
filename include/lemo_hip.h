/* lemo_hip.h -- C ABI of liblemo_hip.so, the MI355X (gfx950) kernels of LEMO's temporal fitting hot path.
 *
 * The reference (sanweiliti/LEMO) has no native/FFI layer: its boundary for this path is the Python
 * object API of smplx / VPoser / Enc.  This header is the boundary the build adds
 * underneath those objects; each entry point names the reference code whose arithmetic it replaces.
 *
 * Conventions (all entry points):
 *   - plain pointers to DEVICE memory + sizes; the caller owns every buffer (the Python host side
 *     allocates them with torch); nothing here allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued asynchronously on it;
 *   - return 0 on success, a hipError_t value or LEMO_ERR_* (>= 10001) otherwise;
 *   - fp32 everywhere (the reference path is fp32; the 1e6-weighted smoothness loss forbids less).
 */
#ifndef LEMO_HIP_H
#define LEMO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif
/* liblemo_hip.so is built with -fvisibility=hidden: the declarations of this header are its ONLY exports (tests/test_abi.py
 * compares `nm -D` with the prototypes below). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define LEMO_ERR_SHAPE 10001
#define LEMO_ERR_ARG 10002
#define LEMO_ERR_STATE 10003

int lemo_abi_version(void);
/* bit 0: built with -fno-slp-vectorize -fno-vectorize (csrc/Makefile passes -DLEMO_NO_PACKED_FP32 together with them): no
 * auto-vectorised packed fp32, the co-residency hazard of DESIGN 9.3.  lemo_amd._hip refuses a library without it. */
int lemo_build_flags(void);

/* ---- motion-smoothness encoder, models/AE_sep.py:11-30,77-99 (Enc(downsample=False)) ----------
 * Activations: CG8P layout act[C/8][(H+2)*(W+2)][8], zero 1-pixel border owned by the caller.
 * Weights: wt[tap][Cin/8][Cout][8] (packed by lemo_amd.priors.pack_conv3x3 / pack_conv3x3_bwd). */
/* epi 0: out = lrelu(conv + bias) | 1: out = conv * lrelu'(aux) (backward-data) | 2: out = conv + bias
 * variant 0: 4 waves x (32 px x Cout) per block | 1: CU-balanced geometry (256 full blocks + fine tail) */
int lemo_conv3x3_mfma(const float* in, const float* wt, const float* bias, const float* aux, float* out,
                      int H, int W, int cin, int cout, int epi, int variant, void* stream);
/* LDS-tiled variant (the engine's default): additionally takes the channel-group-major pack
 * wt2[Cin/8][tap][Cout][8]; W must satisfy 127 + 2*(127/W+1) + 2*(W+2) + 3 <= 416 (W <= 139) */
/* variant 0 with the K = 9 Cin reduction split over `ks` slices (grid.z) + a combine pass; for layers with few pixels
 * and many channels (too few tiles to fill the chip).  partial: ks * (cout/8) * (H+2)*(W+2) * 8 floats of scratch. */
int lemo_conv3x3_mfma_splitk(const float* in, const float* wt, const float* bias, const float* aux, float* out,
                             float* partial, int ks, int H, int W, int cin, int cout, int epi, void* stream);
int lemo_conv3x3_mfma_lds(const float* in, const float* wt, const float* wt2, const float* bias, const float* aux,
                          float* out, int H, int W, int cin, int cout, int epi, void* stream);
/* diagnostics: same launch (epi 0, Cout % 64 == 0) that also records, per wave, {HW_ID, XCC_ID, start, end}
 * shader-clock stamps into dbg[(block*8 + wave)*4 ..] -- used by tools/conv_census.py only */
int lemo_conv3x3_mfma_lds_census(const float* in, const float* wt, const float* wt2, const float* bias, float* out,
                                 int H, int W, int cin, int cout, unsigned long long* dbg, void* stream);
/* variant 3 ("split-bf16"): the same fp32 convolution for Cin, Cout in {32, 64} on the bf16 matrix cores.  Each fp32
 * operand is split exactly into three bf16 pieces and six of the nine piece products are accumulated in fp32
 * (error of the dropped terms < 2^-24 per product, 100x below the fp32 accumulation rounding of any fp32 conv).
 * w3 = weights pre-split on the host: bf16 w3[Cin/16][tap 9][Cout/32][split 3][lane 64][8] with lane l holding
 * cout (l&31) of the 32-block and channels 8*(2*kc + (l>>5)) .. +7; wt = the tap-major fp32 pack (remainder pixels).
 * Returns LEMO_ERR_SHAPE for shapes it does not take (lemo_conv3x3_split_supported() == 0). */
int lemo_conv3x3_split_supported(int H, int W, int cin, int cout);
int lemo_conv3x3_mfma_split(const float* in, const void* w3, const float* wt, const float* bias, const float* aux,
                            float* out, int H, int W, int cin, int cout, int epi, void* stream);
int lemo_conv3x3_mfma_split_census(const float* in, const void* w3, const float* wt, const float* bias, float* out,
                                   int H, int W, int cin, int cout, unsigned long long* dbg, void* stream);
/* variant 4 ("split-f16", the engines' default since round 3): the same convolution with TWO fp16 pieces per operand and
 * three of the four piece products (the error-compensated fp16 scheme of Markidis et al. / Ootomo & Yokota: hi = f16(x s),
 * lo = f16(x s - hi); a b ~= a_hi b_lo + a_lo b_hi + a_hi b_hi, fp32 accumulate) -- half the matrix-core work of variant 3,
 * error vs float64 at the level of an fp32 convolution (conv_split_kernels.hip header has the measurements).  The
 * activations are scaled per workgroup inside the kernel (any fp32 CG8P tensor is a valid input); the weights are scaled
 * on the host: w2 = f16 w2[Cin/16][tap 9][Cout/32][piece 2][lane 64][8] of weight * 2^k, winv = 2^-k. */
int lemo_conv3x3_mfma_split_f16(const float* in, const void* w2, float winv, const float* wt, const float* bias, const float* aux,
                                float* out, int H, int W, int cin, int cout, int epi, void* stream);
/* census of either variant (pieces = 3: w = w3, winv ignored; pieces = 2: w = w2) */
int lemo_conv3x3_mfma_split_census2(const float* in, const void* w, float winv, int pieces, const float* wt, const float* bias, float* out,
                                    int H, int W, int cin, int cout, unsigned long long* dbg, void* stream);
/* variant 5 ("pairs", the engines' default since round 4): TWO consecutive 64 -> 64 layers of the encoder in ONE launch, in the
 * arithmetic of variant 4.  A workgroup owns a 10 x 14 output tile, stages the 14 x 18 input tile once, keeps the 12 x 16
 * intermediate tile in LDS (zero outside the image = the second layer's padding) and never re-reads it from HBM:
 *   epi 0 (forward, models/AE_sep.py:11-30): mid = lrelu(conv(in, A) + biasA), out = lrelu(conv(mid, B) + biasB); `mid` (all 64
 *          channels, CG8P) is ALSO written -- the backward pass needs every layer's activation;
 *   epi 1 (backward-data): mid = conv(in, A) * lrelu'(auxA), out = conv(mid, B) * lrelu'(auxB) with A / B the backward packs of the
 *          LATER / EARLIER layer and auxA / auxB the saved activations at the mid / out positions; `mid` is not written.
 * wA / wB: the variant-4 packs (pack_conv3x3_split_f16 / pack_conv3x3_bwd_split_f16), winvA / winvB their inverse host scales.
 * dbg (epi 0 only, may be NULL): per wave {HW_ID, XCC_ID, start, end, after staging, after layer 1, after the mid planes} stamps. */
int lemo_conv3x3_pair_supported(int H, int W, int c0, int c1, int c2);
int lemo_conv3x3_pair_f16(const float* in, const void* wA, float winvA, const float* biasA, const float* auxA, float* mid,
                          const void* wB, float winvB, const float* biasB, const float* auxB, float* out, int H, int W, int epi,
                          unsigned long long* dbg, void* stream);
/* variant 10 (round 6): ONE 64 -> 64 layer (forward, epi 0, models/AE_sep.py:11-30; backward-data, epi 1) as a Winograd F(2x2, 3x3)
 * convolution: 16 instead of 36 multiplies per 2 x 2 output tile and (cin, cout), the 16 [64 x 64 x tiles] GEMMs in the split-f16
 * arithmetic of variant 4 (fp32 transforms, two fp16 pieces per operand, 3 MFMA products, fp32 accumulate).  A workgroup owns 32
 * consecutive 2 x 2 tiles of the even part of the image; the last row of an odd H is a direct fp32 convolution from `wt`.
 * wU = pack_conv3x3_wino_f16 / pack_conv3x3_bwd_wino_f16 (G g G^T in float64, split on the host; winv = its inverse scale),
 * wt = the fp32 tap-major pack of the same weights (pack_conv3x3 / pack_conv3x3_bwd).  dbg (epi 0 only, may be NULL): per wave
 * {start, loads issued, transformed, planes written, GEMMs done, exchanged, end, HW_ID} shader-clock stamps (tools/wino_check.py). */
int lemo_conv3x3_wino_supported(int H, int W, int cin, int cout);
int lemo_conv3x3_wino_f16(const float* in, const void* wU, float winv, const float* wt, const float* bias, const float* aux, float* out,
                          int H, int W, int epi, unsigned long long* dbg, void* stream);
/* first layer, 1 input channel: x0 padded [(H+2)*(W+2)], w [Cout][9] */
int lemo_conv3x3_c1(const float* x0, const float* w, const float* bias, float* out, int H, int W, int cout, void* stream);
int lemo_conv3x3_c1_bwd(const float* dpre, const float* w, float* dx0, int H, int W, int cout, void* stream);
/* loss_smooth = mean((z[...,1:]-z[...,:-1])^2)  (opt_amass_temp.py:390-391) fused with d(loss)/d(pre-act):
 * partial[lemo_smooth_loss_blocks()] receives per-block sums of squares; coef2 = weight*2/count */
int lemo_smooth_loss_blocks(int H, int W, int C);
int lemo_smooth_loss(const float* z, float* dpre, float* partial, int H, int W, int C, float coef2, void* stream);

/* ---- VPoser.decode, human_body_prior/train/vposer_smpl.py:107-121 ------------------------------ */
typedef struct lemo_vposer_w {           /* *_t = transposed copy [in][out]; the out layer is zero-padded 126 -> 128 */
  const float *w1, *w1t, *b1;            /* bodyprior_dec_fc1  [512][32]  / [32][512]  / [512] */
  const float *w2, *w2t, *b2;            /* bodyprior_dec_fc2  [512][512] / [512][512] / [512] */
  const float *w3, *w3t, *b3;            /* bodyprior_dec_out  [128][512] / [512][128] / [128] */
} lemo_vposer_w;
int lemo_vposer_decode_fwd(const lemo_vposer_w* w, const float* z, int z_stride, int B, float* h1, float* h2, float* o,
                           float* matrot, float* aa, void* stream);
/* scratch: [B][1152] floats */
int lemo_vposer_decode_bwd(const lemo_vposer_w* w, const float* h1, const float* h2, const float* o, const float* d_aa,
                           const float* d_matrot, int B, float* dz, int dz_stride, float* scratch, void* stream);
/* MLP part of the VPoser backward alone: dout [B][128] (= scratch[0 .. B*128)) -> dz */
int lemo_vposer_mlp_bwd(const lemo_vposer_w* w, const float* h1, const float* h2, int B, float* dz, int dz_stride,
                        float* scratch, void* stream);
/* generic small NT GEMM on the matrix cores: C[n][m] = epi(sum_k A[m][k] B[n][k]); M, K multiples of 16 */
int lemo_gemm_nt16(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc,
                   const float* bias, const float* aux, int ldaux, int epi, void* stream);
/* long-K form of the same contraction (M x N <= 512 x 128 outputs, K in the tens of thousands: the feature gradient of the
   all-vertex LBS backward, lbs.py:94-99 transposed): K slabs on the bf16 matrix cores with exactly split fp32 operands, partial
   tiles reduced in slab order (deterministic).  part: lemo_gemm_nt16_splitk_part_floats(M, S) floats of scratch;
   A_grouped: optional copy of A as [K/16][M][16] (contiguous operand tiles); N <= 128, M % 64 == 0, K % 16 == 0 */
int lemo_gemm_nt16_splitk_part_floats(int M, int S);
int lemo_gemm_nt16_splitk(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc, float* part, int S,
                          const float* A_grouped, void* stream);
/* convert_to_3D_rot's 6-D -> axis-angle, utils/utils.py:111-123 (+63-81) */
int lemo_rot6d_to_aa_fwd(const float* x6, int stride, int N, float* aa, void* stream);
int lemo_rot6d_to_aa_bwd(const float* x6, int stride, const float* d_aa, int N, float* dx6, void* stream);

/* ---- SMPL-X pose stage: smplx==0.1.26 SMPLX.forward + lbs.py:81-106,166-263 --------------------- */
typedef struct lemo_body_const {
  int nj, nshape, ncomp, nlev;
  const int *parents, *level_start, *level_joints, *child_start, *child_list;
  const float *J_template, *J_dirs, *pose_mean, *lh_comp, *rh_comp;
} lemo_body_const;
typedef struct lemo_pose_in {
  const float *global_orient, *body_pose, *jaw, *leye, *reye, *lh, *rh;
  int hand_stride;
  const float* betas;
  int betas_stride;
  const float* expr;
  /* optional fused producers (fitting engine): when non-null they replace global_orient / body_pose */
  const float* rot6d;        /* [B][6]  -> global_orient = 6-D -> axis-angle (utils/utils.py:111-123) */
  const float* vposer_o;     /* [B][128] VPoser out layer -> body_pose = 21 x (6-D -> aa) */
  float* go_out;             /* [B][3] receives the global_orient derived from rot6d */
  /* optional per-iteration bookkeeping done by block 0: zero n_zero doubles, latch the step counter */
  double* zero_f64;
  int n_zero;
  const int* step_ctr;
  int* step_cur;
  int* nonfinite;            /* optional [2]: block 0 latches nonfinite[1] = nonfinite[0] (see lemo_fit_desc.nonfinite) */
} lemo_pose_in;
typedef struct lemo_pose_ws {
  float *full_pose, *R, *J, *T, *A, *Jtr, *Xg;
  int Bp;
  unsigned short* XgS;   /* optional [512/16][3][Bp][2][8] bf16: Xg split into its three exact bf16 pieces, in the fragment order of
                            lemo_lbs_verts_fwd_xs's blend GEMM (written by lemo_smplx_pose_fwd when non-NULL; pad entries stay 0) */
  int xgs_f16;           /* 1: XgS holds TWO fp16 pieces instead, [512/16][2][Bp][2][8] (hi = f16(x), lo = f16(x - hi)): the B operand
                            of the blend GEMM on the fp16 matrix cores; goes with lemo_skin_const.DgH (both or neither) */
} lemo_pose_ws;
typedef struct lemo_pose_grad_in {
  const float *dA, *dJtr, *dX;
  const float* d_full_pose;  /* optional [B][3 nj]: extra d(loss)/d(full_pose) added before the pose chain (PROX angle prior) */
} lemo_pose_grad_in;
typedef struct lemo_pose_grad_out {
  float *d_global_orient, *d_body_pose, *d_jaw, *d_leye, *d_reye, *d_lh, *d_rh;
  int hand_stride;
  float *d_betas, *d_expr;
  /* optional fused consumers (fitting engine) */
  const float* rot6d;        /* [B][6] with d_rot6d: chain d(global_orient) back to the 6-D parameters */
  float* d_rot6d;            /* [B][6] */
  const float* vposer_o;     /* [B][128] with d_vposer_o: chain d(body_pose) back to the VPoser out layer */
  float* d_vposer_o;         /* [B][128] */
} lemo_pose_grad_out;
int lemo_smplx_pose_fwd(const lemo_body_const* c, const lemo_pose_in* in, const lemo_pose_ws* ws, int B, void* stream);
int lemo_smplx_pose_bwd(const lemo_body_const* c, const lemo_pose_ws* ws, const lemo_pose_grad_in* gi,
                        const lemo_pose_grad_out* go, int B, void* stream);

/* ---- SMPL-X vertex stage (blend shapes + skinning + transl): lbs.py:81,94-99,108-117 ------------ */
typedef struct lemo_skin_const {
  int V, NC, KW;
  int blend_fp32;           /* blend GEMM of lemo_lbs_verts_fwd: 0 (default) = bf16 matrix cores with exact fp32 operands
                             * (3-way bf16 split, 6 products, fp32 accumulate -- same error class as an fp32 GEMM, see
                             * lemo_conv3x3_mfma_split) ; 1 = fp32-input MFMA.  Per model constant: no process-wide state */
  const float* Dg;          /* [64][NC][8]  blend directions (shape | pose), K padded to 512 */
  const float* v_template;  /* [V][3] */
  const int* w_idx;         /* [V][KW] ELL skinning weights */
  const float* w_val;
  const unsigned short* DgH; /* optional: Dg * 2^k pre-split on the host into two fp16 pieces, [64][NC][2 halves][hi 4 | lo 4] (the same
                             * 32 bytes per (8-feature group, column) as Dg: HBM sees the same traffic, the kernel converts nothing).
                             * With it (and blend_fp32 == 0, XgS in its fp16 form) the blend GEMM is 3 fp16 MFMA products per 16-deep
                             * k-chunk (hi hi + hi lo + lo hi: operands carried to 2^-22, fp32 accumulate) instead of 6 bf16 ones. */
  float dgh_inv;            /* 2^-k */
} lemo_skin_const;
typedef struct lemo_vertex_set_bwd {
  int n, NCs;
  const int *ids, *vp_row;
  const float* Dk;          /* [512][NCs] */
  const float* DkT;         /* [NCs][512]: the same directions feature-contiguous (forward over the set), or NULL */
  const int *jcsr_start, *jcsr_u;
  const float* jcsr_w;
  /* optional (large sets): deterministic dense backward.  Chunk c (512 consecutive set positions) owns the entries
   * jcsr_chunk[c * (nj + 1) + j] .. jcsr_chunk[c * (nj + 1) + j + 1] of jc_u / jc_w for joint j;
   * part: [part_frames][n_chunk][nj * 12 + 4] floats of per-chunk partial sums, reduced in chunk order. */
  const int* jcsr_chunk;          /* [n_chunk * (nj + 1)] offsets into jc_u / jc_w, or NULL */
  const int* jc_u;                /* the same (set position, weight) pairs as jcsr_u / jcsr_w, reordered CHUNK-major (chunk, */
  const float* jc_w;              /* then joint, then position): the entries of one 512-vertex chunk are one contiguous run */
  float* part;
  int part_frames;
  /* optional: K-slab partial tiles of the feature-gradient GEMM dX = Dk . d(v_posed) (K = NCs): gemm_slabs x 128 x 512
   * floats; when NULL the one-workgroup-per-output-tile GEMM is used (fine for the compact sets, 107 us for all vertices) */
  int gemm_slabs;
  float* gemm_part;
  const float* DkG;         /* optional, with gemm_part: Dk again as [NCs/16][512][16] (k-chunk major: the 64-row x 16-k operand tile of a
                               split-K step is one contiguous 4 KB run instead of 64 segments 4 NCs bytes apart) */
} lemo_vertex_set_bwd;
int lemo_lbs_verts_fwd(const lemo_skin_const* c, const float* Xg, int Bp, const float* A, int nj, const float* transl,
                       const int* ids, int n, int B, float* verts, float* v_posed, void* stream);
/* diagnostics (tools/lbs_census.py): same launch, per wave {start, after prologue, after GEMM, end} clock stamps */
/* same, with the per-frame features also given pre-split (lemo_pose_ws.XgS): the blend GEMM reads its B operand from there.
 * XgS must be in the form c selects: two fp16 pieces when c->DgH is set (lemo_pose_ws.xgs_f16 = 1), three bf16 pieces otherwise */
int lemo_lbs_verts_fwd_xs(const lemo_skin_const* c, const float* Xg, const unsigned short* XgS, int Bp, const float* A, int nj,
                          const float* transl, const int* ids, int n, int B, float* verts, float* v_posed, void* stream);
int lemo_lbs_verts_fwd_census(const lemo_skin_const* c, const float* Xg, int Bp, const float* A, int nj, const float* transl,
                              int n, int B, float* verts, float* v_posed, unsigned long long* dbg, void* stream,
                              const unsigned short* XgS /* optional, see lemo_lbs_verts_fwd_xs */);
/* forward over a small vertex set U only: verts / v_posed [B][u->n][3] in U's order, identical arithmetic
 * per vertex up to the summation order of the blend GEMM; blend: [B][u->NCs] floats of scratch; needs u->DkT */
int lemo_lbs_verts_fwd_active(const lemo_skin_const* c, const lemo_vertex_set_bwd* u, const float* Xg, int Bp, const float* A,
                              int nj, const float* transl, int B, float* blend, float* verts, float* v_posed, void* stream);
int lemo_lbs_verts_bwd(const lemo_skin_const* c, const lemo_vertex_set_bwd* u, const float* A, int nj, const float* v_posed,
                       int vp_rows, const float* dverts, int B, int Bp, float* dvp, float* dA, float* dtransl, float* dX,
                       void* stream);
int lemo_joints_assemble(const float* Jtr, int nj, const float* verts, int vrows, const int* extra_rows, int n_extra,
                         const int* lmk_rows, const float* lmk_bary, int n_lmk, const float* transl, int B, float* joints,
                         void* stream);

/* ---- marker-image decode / encode around the loop, one clip (T <= 256 frames) per call ----
 * utils/utils.py:184-203 reconstruct_global_body: in [T][J+2][3] = (ignored reference slot, J local joints, trajectory
 * (dx, dz, dr)); rot_0_pivot from the encode below; out [T][J][3] global positions. */
int lemo_reconstruct_global_body(const float* in, int T, int J, double rot_0_pivot, float* out, void* stream);
/* same, rot_0_pivot read from device memory ([1] double, as lemo_local_markers_4chan leaves it): no host round trip */
int lemo_reconstruct_global_body_dev(const float* in, int T, int J, const double* rot_0_pivot, float* out, void* stream);
/* opt_amass_temp.py:273-325 (twin fitting_temp_slide.py:895-940): decode of the infilling network's output in one launch.
 * rec [d][T] = channel 0 of the un-padded output (d = 3 J + 4: J = pelvis + markers rows, then 4 contact logits);
 * traj [3][T] = row 0 of channels 1..3 of the input image (normalised dx, dz, dr); stats [2 d + 4] doubles =
 * Xmean_local[d], Xstd_local[d], Xmean_global_xy, Xstd_global_xy, Xmean_global_r, Xstd_global_r
 * (preprocess_stats_infill_local_markers_4chan.npz); rot_0_pivot [1] double on the device; post [13] floats or NULL =
 * (z shift, M[9] row-major, t[3]): out = (p + (0,0,shift)) . M + t  (PROX: back to the scene frame, :934-939).
 * -> contact_lbl [T][4] in {0,1} (sigmoid > 0.5), markers [T][J-1][3] global (pelvis row dropped). */
int lemo_decode_clip(const float* rec, const float* traj, const double* stats, const double* rot_0_pivot, const float* post, int T,
                     int J, float* contact_lbl, float* markers, void* stream);
/* utils/utils.py:209-265 get_local_markers_4chan: body [T][1+67][3] global pelvis + markers, contact [T][4] ->
 * image [4][T-1][3*68+4] (local markers + contacts | dx | dz | dr repeated) and rot_0_pivot[1] (float64, device). */
int lemo_local_markers_4chan(const float* body, const float* contact, int T, int M1, float* image, double* rot_0_pivot,
                             void* stream);

/* ---- motion-infilling autoencoder, models/AE.py:11-108 and its finetune step, opt_amass_temp.py:154-214 ----
 * (stride-1 convs / transposed convs run on lemo_conv3x3_mfma; these are the remaining layer types) */
/* MaxPool2d(3,2,1): out is CG8P of ((H-1)/2+1) x ((W-1)/2+1); idx [C/8][Ho*Wo][8] winning tap (uint8) */
int lemo_maxpool3s2_fwd(const float* in, int H, int W, float* out, unsigned char* idx, int C, void* stream);
/* din = scatter of dout to the argmax positions (gather form), optionally times lrelu'(act) */
int lemo_maxpool3s2_bwd(const float* dout, const unsigned char* idx, const float* act, float* din, int H, int W, int C, void* stream);
/* zero-stuffing of ConvTranspose2d(stride 2, output_size = H x W): out[2i][2j] = in[i][j]; and its adjoint */
int lemo_stuff2_fwd(const float* in, int h, int w, float* out, int H, int W, int C, void* stream);
int lemo_stuff2_bwd(const float* dout, int H, int W, const float* act, float* din, int h, int w, int C, void* stream);
/* dW[co][ci][3][3] = sum_p dY[co][p] X[ci][p+tap] (+ db[co] = sum_p dY) ; partial: [nslab][9][cout][cin] scratch */
int lemo_conv3x3_wgrad_nslab(int H, int W);
int lemo_conv3x3_wgrad(const float* dy, const float* x, int H, int W, int cin, int cout, int cin_real, int cout_real,
                       float* partial, float* dw, float* db, void* stream);
/* the same in two stages for a whole network: slab partials per layer (any stream), then ONE launch that reduces the partials
   of up to LEMO_WGRAD_MAX_JOBS layers (same summation order as lemo_conv3x3_wgrad: identical bits).  models/AE.py:36-108 has
   20 convolutions; the per-layer reduce launches were 0.25 ms of its 2 ms training step. */
#define LEMO_WGRAD_MAX_JOBS 24
typedef struct lemo_wgrad_job {
  const float* partial;     /* [nslab][9][cout][cin] written by lemo_conv3x3_wgrad_partial */
  const float* dy;          /* CG8P, cout channels (bias gradient) */
  float* dw;                /* [cout_real][cin_real][3][3] */
  float* db;                /* [cout_real] or NULL */
  int nslab, cin, cout, cin_real, cout_real, H, W;
} lemo_wgrad_job;
int lemo_conv3x3_wgrad_partial(const float* dy, const float* x, int H, int W, int cin, int cout, float* partial, void* stream);
int lemo_conv3x3_wgrad_reduce_multi(const lemo_wgrad_job* jobs, int n, void* stream);
/* torch.optim.Adam (defaults) over a flat buffer; step is 1-based */
int lemo_adam_flat(float* p, const float* g, float* m, float* v, int n, float lr, int step, void* stream);
/* same with the step count on the device (step_ctr[0] = completed steps; advanced by one after the update), so that a
 * captured graph of a whole training step can be replayed */
int lemo_adam_flat_ctr(float* p, const float* g, float* m, float* v, int n, float lr, int* step_ctr, void* stream);

/* ---- native training-step engine of the infilling autoencoder (lemo_amd/csrc/ae_engine.hip) ------------------------
 * Replaces, for one clip, the block opt_amass_temp.py:160-214 (= temp_prox/fitting_temp_slide.py:861-893): "reload the
 * pretrained AE, optimizer = Adam(lr 3e-6), 60 x [rec = AE(x); loss = sum(|rec - x| * mask) / count; backward; step],
 * eval forward".  One step is 53 launches (20 + 19 convolutions with fused epilogues, pooling, ONE launch for all 20 weight
 * gradients, ONE for slab reduction + Adam + re-packing), captured into 5-step and 1-step graphs on first use.
 *   ws / ws_floats: caller-owned ZEROED device workspace of lemo_ae_ws_floats(H, W) floats, alive as long as the engine;
 *   H x W: the clip image [1,4,H,W] (H = 3 * markers + contacts + 2 pads, W = frames + 16), H, W >= 2, H * W <= 2^22.
 * lemo_ae_load: `flat` = the model's 40 tensors back to back in state_dict order per layer (weight, bias; enc_blc1.main.0,
 *   enc_blc1.main.2, ..., dec_blc5.deconv2; each in its own layout: Conv2d [out][in][3][3], ConvTranspose2d [in][out][3][3]),
 *   lemo_ae_n_param() floats; x = clip image [4][H][W]; moc = train mask / count, [H][W] (d loss / d rec = sign(rec - x) * moc).
 *   Resets the optimizer state (a fresh Adam per clip, as the reference builds one).
 * lemo_ae_step: n training steps on `stream` (use_graph: replay captured steps; same kernels, same results).
 * lemo_ae_forward: eval forward with the current parameters -> rec [H][W], z [256][h5][w5] (may be NULL; h5, w5 = five times
 *   (n - 1) / 2 + 1).   lemo_ae_params: the current parameters in `flat` order. */
typedef struct lemo_ae_desc {
  int H, W;
  float lr;
  float* ws;
  long long ws_floats;
  int clips;                      /* round 4 (appended; 0 or 1 = one clip): K clips SIDE BY SIDE in every launch of a step -- the clip is
                                   * the last grid dimension of the convolution / pooling / weight-gradient / Adam launches, each clip has
                                   * its own parameters, Adam state and step counter in its own slice of the workspace (the reference
                                   * finetunes a fresh copy of the pretrained model per clip): ws_floats >= clips * lemo_ae_ws_floats(H, W).
                                   * Round 5: the convolutions' launch shapes are chosen for the clips in flight (18.0 -> 16.4 ms per clip at
                                   * 8): a clip's results depend on `clips` through the order its K slices are summed in -- bit-identical
                                   * for equal `clips` and between the slots of one engine, equal to a one-clip engine's to rounding. */
} lemo_ae_desc;
long long lemo_ae_ws_floats(int H, int W);
int lemo_ae_n_param(void);
void* lemo_ae_create(const lemo_ae_desc* d);
void lemo_ae_destroy(void* h);
int lemo_ae_load(void* h, const float* flat, const float* x, const float* moc, void* stream);
int lemo_ae_step(void* h, int n, int use_graph, void* stream);          /* every clip of the engine advances n steps */
int lemo_ae_forward(void* h, float* rec, float* z, void* stream);
int lemo_ae_params(void* h, float* flat_out, void* stream);
/* the same per clip of a multi-clip engine (clip 0 .. desc.clips - 1; the un-suffixed forms address clip 0).  lemo_ae_forward_clip with
 * clip 0 (or -1: no copy-out) runs the eval forward of ALL clips; clips > 0 only copy that forward's reconstruction / latent out:
 * call it for clip 0 first.  State rules (LEMO_ERR_STATE otherwise): lemo_ae_step / lemo_ae_forward* need EVERY clip of the engine loaded
 * since creation (a step advances all of them; an unloaded clip would train from a zeroed workspace); lemo_ae_forward_clip(clip > 0) needs
 * an eval forward (clip 0 or -1) AFTER the last step / load; lemo_ae_params_clip needs that clip loaded. */
int lemo_ae_load_clip(void* h, int clip, const float* flat, const float* x, const float* moc, void* stream);
int lemo_ae_forward_clip(void* h, int clip, float* rec, float* z, void* stream);
int lemo_ae_params_clip(void* h, int clip, float* flat_out, void* stream);
/* diagnostics (tools/ae_wgrad_probe.py): the step's weight-gradient launch alone on the engine's current buffers; mode 0 = as in
 * a step, 1 = operands loaded once per wave, 2 = loads without MFMAs (1 and 2 leave garbage in the slab partials). */
int lemo_ae_wgrad_probe(void* h, int mode, void* stream);
/* one convolution of the engine on its own (tests, tools).  Enumerates an H x W pixel grid; in_s = 2: the input (and the
 * epi-1 operand) is a fineH x fineW image read at its even pixels; out_s = 2: the output is written to the even pixels of a
 * fineH x fineW image (zero-stuffing geometry).  mt = 0: the engine's own launch shape; else tile mt (1: 32 px x 32 cout,
 * 2: 32 x 64, 3: 16 x 16) with pt pixel tiles x ks K-slices = pt * ks <= 16 waves per workgroup. */
int lemo_ae_conv(const float* in, const float* wt, const float* bias, const float* aux, float* out, int H, int W, int fineH, int fineW,
                 int in_s, int out_s, int cin, int cout, int epi, int mt, int pt, int ks, void* stream);
/* ... on the split-f16 kernels (round 6, the engine's default arithmetic): both operands are read as fp32 and split in registers into two
 * error-compensated fp16 pieces, three f16 MFMA products, fp32 accumulate.  amax_in [1] = max |in| (or a bound, multiplied by in_fac >= 1),
 * wmax [1] = max |wt|, amax_out [1] receives max |out| of the launch (atomicMax: zero it first).  cin >= 16 (mt 3: cin >= 32). */
int lemo_ae_conv_f16(const float* in, const float* wt, const float* bias, const float* aux, float* out, int H, int W, int fineH, int fineW,
                     int in_s, int out_s, int cin, int cout, int epi, int mt, int pt, int ks, const float* amax_in, float in_fac,
                     const float* wmax, float* amax_out, void* stream);

/* ---- stream capture helpers: record everything a host-side step enqueues on `stream` (HIP kernels of this library
 * and the caller's own device work alike) into an executable graph, replay it with one call.  Relaxed capture mode;
 * the caller guarantees that the step does not synchronise and that every buffer it touches outlives the replays. */
int lemo_capture_begin(void* stream);
int lemo_capture_end(void* stream, void** graph_exec);
int lemo_graph_launch(void* graph_exec, void* stream);
int lemo_graph_destroy(void* graph_exec);

/* ---- PROX scene terms: F.grid_sample(sdf, verts, padding_mode='border') of temp_prox/fitting_temp_slide.py:685-739
 * sdf [D][H][W] device; pts [N][3] device world coordinates; gmin/gmax HOST float[3]; val [N]; dval [N][3] or NULL
 * (d val / d pts).  Grid axis order follows the reference's norm_vertices[:, :, [2,1,0]]. */
int lemo_sdf_sample(const float* sdf, int D, int H, int W, const float* pts, int N, const float* gmin, const float* gmax,
                    float* val, float* dval, void* stream);

/* ---- AMASS temporal fitting iteration, opt_amass_temp.py:349-455 -------------------------------- */
typedef struct lemo_fit_const {
  int n, n67, n81;
  const int *row67, *row81, *foot_start, *foot_row, *u_row, *u_m67, *u_m81, *u_foot_mask;
  const float *Xstd, *Xmean;
  const float* cam2world;         /* optional [12] = R row-major, t (PROX, fitting_temp_slide.py:676-680): the canonical frame
                                   * is built from WORLD joints R (J + transl) + t and applied to world markers R v + t; the
                                   * published canon matrix is R^T R0 so that it acts on camera-frame vertices directly */
} lemo_fit_const;

/* conv variant 7 (round 5) = variant 5 + the encoder's head and tail as ONE launch each (csrc/conv_head_kernels.hip):
 *   lemo_enc_head: canonicalised marker image (opt_amass_temp.py:366-392) -> layer 0 (1 -> 32, fp32 FMAs, bit-identical to lemo_conv3x3_c1 on the
 *     published x0) -> layer 1 (32 -> 32, split-f16 MFMA; w1pack / w1inv = pack_conv3x3_split_f16 of its weights) -- writes x0 (padded image),
 *     canon [12], act1 and act2 (CG8P, 32 channels each); replaces the marker_c1 launch + one single-layer launch;
 *   lemo_enc_tail: d(pre-act 2) (CG8P, 32 channels) -> layer 1 backward-data x lrelu'(act1) (w1bpack = pack_conv3x3_bwd_split_f16) -> layer 0
 *     adjoint (w0 [32][9]) -> dx0 [H * W]; replaces one single-layer launch + lemo_conv3x3_c1_bwd; d(pre-act 1) stays in LDS. */
int lemo_enc_head(const lemo_fit_const* fc, const float* verts, int nrows, const float* Jtr, int nj, const float* transl, int B, const float* w0,
                  const float* b0, const void* w1pack, float w1inv, const float* b1, float* x0, float* canon, float* act1, float* act2,
                  void* stream);
int lemo_enc_tail(const float* din, const void* w1bpack, float w1binv, const float* act1, const float* w0, float* dx0, int H, int W,
                  void* stream);
/* conv variant 9: the tail with layer 2's backward-data in front -- d(pre-act 3) (CG8P, 64 channels) -> conv^T 64 -> 32 x lrelu'(act2)
 * (w2bpack = pack_conv3x3_bwd_split_f16 of layer 2) -> lemo_enc_tail's two stages -> dx0; replaces two single-layer launches +
 * lemo_conv3x3_c1_bwd (models/AE_sep.py:16-21 backwards); d(pre-act 2) and d(pre-act 1) stay in LDS. */
int lemo_enc_tail3(const float* din, const void* w2bpack, float w2binv, const float* act2, const void* w1bpack, float w1binv, const float* act1,
                   const float* w0, float* dx0, int H, int W, void* stream);

typedef struct lemo_fit_desc {
  int B, Bp, V, nrows;            /* frames, padded frames, model vertices, rows of `verts` (V or n) */
  int full_vertices;              /* 1: regress all V vertices per frame (reference behaviour) ; 0: only the set U */
  int conv_variant;               /* kernel family of the encoder's MFMA layers (the struct has no default: 0 selects lemo_conv3x3_mfma variant 0;
                                   * lemo_amd.priors.DEFAULT_CONV_VARIANT = 9 is what the Python fitters pass and what every gate runs on):
                                   * 9 (shipped): lemo_enc_head3 (image + layers 0-2) + fused 64 -> 64 pairs (lemo_conv3x3_pair_f16) + lemo_enc_tail3
                                   *    (layers 2-0 backwards), split-f16 arithmetic throughout ; 8: 9 with layer 2's backward a launch of its own ;
                                   * 7: head / tail without layer 2 (lemo_enc_head / lemo_enc_tail) ; 5: pairs only, head and tail layer by layer ;
                                   * 10 (round 6): 9 with every 64 -> 64 layer ONE Winograd launch (lemo_conv3x3_wino_f16) instead of the pairs --
                                   *    parity-green, measured slower, enc_w3 / enc_wbwd3 of those layers then hold the Winograd packs ;
                                   * 4: split-f16, one launch per layer ; 3: split-bf16 (lemo_conv3x3_mfma_split) where it takes the shape, else 2 ;
                                   * 2: lemo_conv3x3_mfma_lds ; 0 / 1: lemo_conv3x3_mfma variants (the any-shape fallback of all the others).
                                   * (6, the pairs on four-wave workgroups, left the library in round 6: csrc/attic) */
  lemo_vposer_w vposer;
  lemo_body_const body;
  lemo_skin_const skin;
  lemo_vertex_set_bwd uset;
  lemo_fit_const fit;
  const int* fwd_ids;             /* [n] when !full_vertices */
  /* smoothness encoder: 10 layers; channels ch[0..10] = 1,32,32,64,... */
  int enc_ch[11];
  const float* enc_w[10];         /* layer 0: [Cout][9] ; others packed wt[tap][Cin/8][Cout][8] */
  const float* enc_b[10];
  const float* enc_wbwd[10];      /* backward-data packs (layer 0: same [Cout][9]) */
  const float* enc_w2[10];        /* channel-group-major packs for conv_variant 2 (layer 0 unused) */
  const float* enc_wbwd2[10];
  const void* enc_w3[10];         /* split packs (layer 0: NULL): bf16 x 3 pieces for conv_variant 3, f16 x 2 pieces for 4 */
  const void* enc_wbwd3[10];
  float enc_w3_inv[10];           /* conv_variant 4: 2^-k of each pack's host-side weight scale 2^k (lemo_conv3x3_mfma_split_f16) */
  float enc_wbwd3_inv[10];
  /* sequence data */
  const float* target;            /* [B][n67][3]  markers_rec */
  const float* contact;           /* [B][4] */
  const float* weights;           /* [6] rec_markers, vposer, shape, hand, contact_vel, smooth (device) */
  float weights_host[6];          /* same values, host copy (launch-time constants) */
  /* optimised parameters + Adam state */
  float *transl, *rot6d, *other;  /* [B][3], [B][6], [B][56] */
  const float* shape;             /* [B][10] fixed */
  float *adam_m[3], *adam_v[3];
  int* step_ctr;
  float lr0, lr1;
  int lr_switch;
  /* workspace (caller-allocated, sizes documented in lemo_amd/fitting.py) */
  float *go_aa, *body_aa, *h1, *h2, *vo, *vp_scratch;   /* vp_scratch [B][1152] */
  lemo_pose_ws pose;
  float *verts, *v_posed, *x0, *canon;
  float* act[11];                 /* act[0] unused; act[l] = output of layer l, CG8P */
  float* dact[2];                 /* ping-pong d(pre-activation) buffers, 64-channel CG8P */
  float *dx0, *spartial, *vpartial, *losses, *dverts, *dvp, *dA, *dX;
  double* loss_acc;               /* [32][16] per-iteration loss accumulators (f64 atomics, 32 slots) */
  int* step_cur;                  /* [1] step index latched at the start of the iteration */
  float *g_transl, *g_rot6d, *g_other, *g_go, *g_body;
  /* ---- round-2 additions (appended: earlier offsets unchanged; all optional / zero = previous behaviour) ---- */
  float* snap;                    /* [B*65] or NULL: every Adam update first stores the PRE-update parameters here
                                   * (transl [B][3] | rot6d [B][6] | other [B][56]): with go_aa of the same iteration's
                                   * forward this is the reference's body_params_opt_t_72 (opt_amass_temp.py:457) */
  int* nonfinite;                 /* [2] or NULL, zeroed by the caller.  [0]: 1-based index of the first iteration whose
                                   * total loss was NaN / Inf (0 = none), written by the Adam kernel; [1]: its copy latched at
                                   * the start of the next iteration.  Once latched, parameter updates are skipped: the device
                                   * side of FittingMonitor.run_fitting's "NaN/Inf loss -> stop" (fitting_temp_slide.py:198-204);
                                   * like there, the update of the offending iteration itself has already been applied. */
  int per_frame;                  /* 1: opt_amass_perframe.py:324-351 -- marker L1 + the three L2 priors only (no smoothness
                                   * encoder, no contact term), any B >= 1 */
  float lr2;                      /* third learning-rate level: lr = lr2 when step > lr_switch2 > 0 (opt_amass_perframe.py:316-321) */
  int lr_switch2;
  /* ---- round-6 addition (appended; NULL = previous behaviour): the all-vertex forward OFF the iteration's critical path ---- */
  float* verts_side;              /* [B][V][3] or NULL.  With full_vertices = 0 (the loss path forwards only the set U): every iteration ALSO regresses
                                   * all V vertices of its pose into this buffer (what the reference's smplx forward returns: utils/utils.py:152), by a
                                   * launch on the engine's own side stream that starts behind the encoder's backward tail and runs beside the per-frame
                                   * launches of the iteration's end (119 of 256 CUs); it is joined before the next iteration's pose stage overwrites its
                                   * operands, and at the end of every call / graph.  Same kernel, same bits as the in-line forward of full_vertices = 1. */
  float* transl_side;             /* [B][3], required with verts_side: the translation of the iteration's forward (the Adam launch updates `transl` while
                                   * the side launch is in flight; the set-U forward leaves a copy here) */
} lemo_fit_desc;

/* Opaque engine: holds a copy of the descriptor (pointers only) and, optionally, a captured hipGraph. */
void* lemo_fit_create(const lemo_fit_desc* d);
void lemo_fit_destroy(void* h);
/* one forward pass only (no backward / Adam): fills verts, losses, ... */
int lemo_fit_forward(void* h, void* stream);
/* after lemo_fit_forward: gradients of the total loss into g_transl / g_rot6d / g_other (priors' own
 * gradient terms are added inside the Adam kernel), no parameter update */
int lemo_fit_backward(void* h, void* stream);
/* n full iterations (forward, backward, Adam).  use_graph: the call is replayed from hipGraphs -- a 1-iteration and a
 * 5-iteration graph to get the device going, then 20-iteration graphs, then ONE graph for what is left (sizes 1 .. 20 are captured on
 * first use, on `stream`, which must not be the legacy default stream, and kept). */
int lemo_fit_step(void* h, int n, int use_graph, void* stream);
/* record (without running anything) the graphs an n-iteration lemo_fit_step(use_graph = 1) on `stream` will replay,
 * so that the first such call does not pay for capture + instantiation */
int lemo_fit_prepare(void* h, int n, void* stream);
/* Diagnostics (bench.py's per-stage figures): average duration in ms of every stage of ONE iteration, each measured as `reps`
 * back-to-back repetitions of the stage's own launches captured into a graph and replayed between two events (consecutive graph
 * nodes, like inside the iteration): ms[0] VPoser decode + pose stage, [1] vertex stage (lbs_verts_fwd), [2] marker image + first
 * layer + encoder forward, [3] losses, [4] encoder backward-data + first-layer adjoint, [5] vertex-stage backward (+ d(verts)),
 * [6] pose / VPoser backward + tail (no update), [7] = LEMO_FIT_NSTAGE: the whole forward + backward the same way.  `stream` must be
 * capturable (not the legacy default stream).  The call SYNCHRONISES (event timing) and leaves losses / gradients of a clean
 * forward + backward of the current parameters behind; parameters and optimiser state are not touched. */
#define LEMO_FIT_NSTAGE 7
int lemo_fit_census(void* h, int reps, float* ms_out /* host [LEMO_FIT_NSTAGE + 1] */, void* stream);
/* Optimiser state of a fit = what torch.optim.Adam + the three parameter tensors hold between two iterations of
 * opt_amass_temp.py:349-455 (opt_amass_perframe.py:312-355 for per_frame engines): the parameters, Adam's exp_avg /
 * exp_avg_sq and the number of completed steps (which also selects the learning-rate level, :350-352).  All pointers are
 * caller-owned DEVICE buffers shaped like the descriptor's ([B][3], [B][6], [B][56] each; step: int[1]).
 * lemo_fit_load_state: engine <- st (and the NaN / Inf latch is cleared): the next lemo_fit_step continues from that state as
 *   if the engine had produced it itself -- used to resume a fit and, in the parity tests, to run ONE step from a state the
 *   reference's own loop went through (teacher forcing) instead of comparing free-running trajectories.
 * lemo_fit_save_state: st <- engine.  One kernel launch each, asynchronous on `stream`, capturable. */
typedef struct lemo_fit_state {
  float *transl, *rot6d, *other;
  float *adam_m[3], *adam_v[3];   /* transl, rot6d, other */
  int* step;                      /* [1] completed Adam steps */
} lemo_fit_state;
int lemo_fit_load_state(void* h, const lemo_fit_state* st, void* stream);
int lemo_fit_save_state(void* h, const lemo_fit_state* st, void* stream);

/* ---- PROX sliding-window fitting iteration (the twin of the loop body above) ------------------------------------------
 * temp_prox/fitting_temp_slide.py: closure fitting_func :239-311 (VPoser decode, SMPL-X, loss, backward, erase of the
 * first int(0.15 B) frames' gradients unless first_batch_flag), SMPLifyLoss.forward as configured by
 * cfg_files/PROXD_temp_S2.yaml / S3.yaml -- 2-D keypoints :573-580 through PerspectiveCamera (camera.py:88-116, fixed
 * identity pose) and JointMapper (misc_utils.py:44-57); L2 / angle priors :586-615 (prior.py:50-90); cam -> world
 * :676-680; SDF penetration :685-694; friction :699-739; infill L1 + contact velocity :944-992 (S3); smoothness prior
 * :997-1031; sum + loss_dict :1036-1061 -- and Adam (optimizers/optim_factory.py:43-46, lr 0.005).  Every .item() branch
 * of the reference is a device-side count; nothing returns to the host. */
typedef struct lemo_prox_const {
  int n_op, n_sj;                 /* OpenPose keypoints (118); smplx joints nj + n_extra + n_lmk (127) */
  const int* joint_map;           /* [n_op] smplx joint each keypoint reads */
  const int *jm_start, *jm_list;  /* inverse of joint_map: CSR [n_sj + 1], [n_op] */
  int n_extra, n_lmk;
  const int* extra_rows;          /* [n_extra] vertex-pick joints */
  const int* lmk_rows;            /* [n_lmk][3] face vertices of the barycentric landmarks */
  const float* lmk_bary;          /* [n_lmk][3] */
  /* S: sorted unique ids of every vertex that carries a term besides the SDF penetration (friction set, 67 + 81 markers,
   * heel / toe sets, vertex-pick joints, landmark face vertices) */
  int n_s;
  const int *s_vid, *s_m67, *s_m81, *s_foot_mask, *s_fric;   /* [n_s]: position in the respective list or -1; 4-bit mask */
  const int *s_jstart, *s_jidx;   /* CSR [n_s + 1] over (vertex-joint index in [0, n_extra + n_lmk), weight) */
  const float* s_jw;
  int n_fric;
  const int* fric_vid;            /* [n_fric] contact_fric_verts_ids (fit_temp_loadprox_slide.py:343-349) */
  int n67;
  const int* m67_vid;             /* [n67] infill markers */
  const int *foot_start, *foot_vid;   /* [5], [...]: left heel, right heel, left toe, right toe */
} lemo_prox_const;

/* weights[13] (device + host copy): data, body_pose, shape, bending_prior, hand_prior, expr_prior, jaw_prior,
 * sdf_penetration, motion_prior_smooth, friction_normal, friction_tangent, motion_infill_rec, motion_infill_contact */
#define LEMO_PROX_NW 13
/* losses[16]: the 14 loss_dict entries in the reference's order (:1047-1061) -- total_loss, joint_loss, s2m_dist, m2s_dist,
 * self_penetration_loss, sdf_penetration_loss, contact_loss, smooth_acc_loss, smooth_vel_loss, motion_prior_smooth_loss,
 * loss_fric_tangent, loss_fric_normal, motion_infill_loss, motion_infill_contact_loss -- then 2 spare */
typedef struct lemo_prox_desc {
  int B, Bp, V;
  int conv_variant;
  int first_batch_flag;           /* 0: gradients of frames [0, int(0.15 B)) are erased every iteration (:282-289) */
  int use_infill;                 /* S3 terms live (marker_mask has an occluded entry, :944) */
  int T;                          /* rows of body_markers_rec / contact_lbl_rec (B - 1) */
  lemo_vposer_w vposer;
  lemo_body_const body;
  lemo_skin_const skin;
  lemo_vertex_set_bwd uset;       /* ALL vertices, with jcsr_chunk + part (deterministic dense backward) */
  lemo_fit_const fit;             /* smoothness-marker tables: n81, row81, Xstd, Xmean, cam2world */
  lemo_prox_const pc;
  int enc_ch[11];
  const float* enc_w[10]; const float* enc_b[10]; const float* enc_wbwd[10];
  const float* enc_w2[10]; const float* enc_wbwd2[10];
  const void* enc_w3[10]; const void* enc_wbwd3[10];
  float enc_w3_inv[10]; float enc_wbwd3_inv[10];      /* conv_variant 4 (see lemo_fit_desc) */
  /* scene */
  const float* sdf; int sdf_dim[3];           /* [D][H][W] */
  float grid_min[3], grid_max[3];
  float cam2world[12];            /* R row-major, t (host copy; fit.cam2world is the device copy) */
  float cam[4];                   /* fx, fy, cx, cy (PROXD_temp_S2.yaml:111-114) */
  /* window data */
  const float* gt_joints;         /* [B][n_op][2] */
  const float* w2;                /* [B][n_op] (joint_weights * joints_conf)^2 */
  const float* marker_mask;       /* [B][n67] or NULL */
  const float* body_markers_rec;  /* [T][n67][3] or NULL */
  const float* contact_lbl_rec;   /* [T][4] or NULL */
  const float* weights;           /* device [LEMO_PROX_NW] */
  float weights_host[LEMO_PROX_NW];
  /* parameters (the smplx module's nn.Parameters + pose_embedding), fixed betas */
  float *global_orient, *transl, *left_hand_pose, *right_hand_pose, *jaw_pose, *leye_pose, *reye_pose, *expression, *pose_embedding;
  const float* betas;             /* [B][10] */
  float *adam_m, *adam_v;         /* [B][81] each, parameter order as listed above */
  int* step_ctr; int* step_cur; int* nonfinite;   /* [1], [1], [2] */
  float lr;
  /* workspace */
  float *h1, *h2, *vo, *vp_scratch;
  lemo_pose_ws pose;
  float *verts, *v_posed, *dverts;            /* [B][V][3] */
  float *x0, *canon, *dx0;
  float* act[11]; float* dact[2];
  float *dJtr, *dJv, *dtr_j, *gp, *dfp_add;   /* [B][nj][3], [B][n_extra+n_lmk][3], [B][3], [B][81], [B][3 nj] (zeroed once) */
  float *dvp, *dA, *dtr_v, *dX;
  float *g_go, *g_lh, *g_rh, *g_jaw, *g_leye, *g_reye, *g_expr, *g_pe;
  double* loss_acc;               /* [32][32] */
  float* losses;                  /* [16] */
} lemo_prox_desc;
void* lemo_prox_create(const lemo_prox_desc* d);
void lemo_prox_destroy(void* h);
/* forward + backward without the update: losses[] and the g_* / dtr_* buffers (erase applied by the update only) */
int lemo_prox_closure(void* h, void* stream);
/* n iterations (closure + Adam); use_graph as lemo_fit_step */
int lemo_prox_step(void* h, int n, int use_graph, void* stream);
/* Optimiser state of a PROX window (fitting_temp_slide.py:196-204: optimizer.step(closure) x maxiters on the parameters
 * fit_temp_loadprox_slide.py:511-519 collects): the nine optimised tensors in lemo_prox_desc's shapes, Adam's moments as
 * [B][81] blocks in the same order (global_orient 3 | transl 3 | left_hand 12 | right_hand 12 | jaw 3 | leye 3 | reye 3 |
 * expression 10 | pose_embedding 32) and the completed-step count.  Same contract as lemo_fit_load_state / save_state;
 * window k + 1 of a recording is window k's saved parameters on the overlap with a FRESH Adam (data_parser_slide.py:326-331). */
typedef struct lemo_prox_state {
  float *global_orient, *transl, *left_hand_pose, *right_hand_pose, *jaw_pose, *leye_pose, *reye_pose, *expression, *pose_embedding;
  float *adam_m, *adam_v;         /* [B][81] */
  int* step;                      /* [1] */
} lemo_prox_state;
int lemo_prox_load_state(void* h, const lemo_prox_state* st, void* stream);
int lemo_prox_save_state(void* h, const lemo_prox_state* st, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
