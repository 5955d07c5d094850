// Head and tail of the smoothness encoder as ONE launch each (VERDICT r04 next #1b: "fuse c1 + 32 -> 32 into marker_c1 / c1_bwd"):
//   enc_head : canonicalised marker image (opt_amass_temp.py:366-392) -> conv 1 -> 32 + LeakyReLU (models/AE_sep.py:16-17, layer 0,
//              plain fp32 FMAs) -> conv 32 -> 32 + LeakyReLU (layer 1, split-f16 MFMA) -- replaces marker_c1_kernel + one
//              conv3x3_split_kernel<0, 32, 32> launch: one kernel boundary and one 4.2 MB read less; x0, act[1] and act[2] are all
//              still written (parity tests read x0, the backward pass reads every activation);
//   enc_tail : d(pre-act 2) -> conv^T 32 -> 32 x lrelu'(act[1]) (split-f16 MFMA) -> conv^T 32 -> 1 (fp32 FMAs) = d(loss)/d(image)
//              -- replaces conv3x3_split_kernel<1, 32, 32> + conv3x3_c1_bwd_kernel: d(pre-act 1) never leaves the CU.
// Geometry = the fused pairs' (conv_pair_kernels.hip): a workgroup of 8 waves owns a 10 x 14 tile of its OUTPUT map, the intermediate
// 12 x 16 tile (one-pixel halo) lives in LDS, N-tiles are 2 rows x 16 columns on row pitch 18 with the second row rotated by 2
// (conflict-free ds_read_b128), 250 workgroups for 245 x 134: one per CU.  The MFMA layer has ONE M-tile (32 couts) and K = 288, so
// there is no K split and no exchange: a wave owns an N-tile over the whole K (18 steps x 3 products, one accumulator per product:
// the three would otherwise wait for each other), 5 (head) / 6 (tail) of the 8 waves carry one.
// Arithmetic: layer 0 exactly as marker_c1_kernel / conv3x3_c1_bwd_kernel (same FMA order: act[1] is bit-identical to theirs, tested);
// the MFMA layer is conv variant 4's (two fp16 pieces, three products, power-of-two tile scale).
#include <type_traits>
#include "conv_f16.hpp"
#include "loss_device.hpp"

namespace lemo {

constexpr int HD_TH = 10, HD_TW = 14;
constexpr int HD_XW = HD_TW + 4, HD_XH = HD_TH + 4, HD_NX = HD_XW * HD_XH;             // 18 x 14 = 252: image / input tile with a halo of 2
constexpr int HD_MIDW = HD_TW + 2, HD_MIDH = HD_TH + 2;                                // 16 x 12 = 192 mid pixels
constexpr int HD_MIDP = HD_MIDW + 2, HD_NMIDP = HD_MIDP * HD_MIDH;                     // pitch 18 (one pad column each side): 216 slots
constexpr int HD_PL_MID = HD_NMIDP * 16, HD_GRP_MID = 2 * HD_PL_MID;                   // a (group, piece) plane: 3456 B
constexpr int HD_PL_IN = HD_NX * 16, HD_GRP_IN = 2 * HD_PL_IN;                         // tail: input planes, 4032 B each
static_assert(HD_MIDW == 16 && HD_XW == HD_MIDP, "N-tiles = 2 rows x 16 columns on grids of row pitch 18");

__device__ __forceinline__ int hd_lane_col(int j) { return j < 16 ? j : ((j - 18) & 15); }

template <int U, int END> struct HdSteps {
  template <class F> static __device__ __forceinline__ void run(F&& f) { f(std::integral_constant<int, U>{}); HdSteps<U + 1, END>::run(f); }
};
template <int END> struct HdSteps<END, END> { template <class F> static __device__ __forceinline__ void run(F&&) {} };

// 32 -> 32 layer for ONE N-tile over the whole K = 2 k-chunks x 9 taps: A = weight fragments w[kc][tap][piece][lane] (one M-tile) from
// L2 through a ring, B from the LDS planes at bbase (group stride GRP, piece stride PL, row pitch 18), one step ahead.
// acc[p]: one accumulator per product (lo x hi, hi x lo, hi x hi); the caller adds them smallest first.
#define HD_RA 4
// NST = 9 x (input channels / 16) steps; PITCH = row pitch of the B planes.
template <int GRP, int PL, int MT = 1, int NST = 18, int PITCH = HD_MIDP>
__device__ __forceinline__ void hd_kloop(f32x16 (&acc)[3], const uint4* __restrict__ w, const unsigned char* bbase, int li, int lane, int mt = 0) {
  const int h = lane >> 5;
  uint4 ra[HD_RA][2], rb[2][2];
#define HD_LOAD_A(SET, U) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) ra[SET][s_] = w[(unsigned)(((U) * MT + mt) * 2 + s_) * 64u + lane];
#define HD_LOAD_B(SET, U) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                          \
    rb[SET][s_] = *reinterpret_cast<const uint4*>(bbase + (2 * ((U) / 9) + h) * GRP + s_ * PL + (li + (((U) % 9) / 3 - 1) * PITCH + (((U) % 9) % 3 - 1)) * 16);
#pragma unroll
  for (int u0 = 0; u0 < HD_RA - 1; ++u0) { HD_LOAD_A(u0, u0) }
  HD_LOAD_B(0, 0)
  HdSteps<0, NST>::run([&](auto uc) {
    constexpr int u = decltype(uc)::value;
    if (u + HD_RA - 1 < NST) { HD_LOAD_A((u + HD_RA - 1) % HD_RA, (u + HD_RA - 1 < NST ? u + HD_RA - 1 : NST - 1)) }
    if (u + 1 < NST) { HD_LOAD_B((u + 1) & 1, (u + 1 < NST ? u + 1 : NST - 1)) }
    __builtin_amdgcn_sched_barrier(0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[u % HD_RA][0]), __builtin_bit_cast(f16x8, rb[u & 1][1]), acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[u % HD_RA][1]), __builtin_bit_cast(f16x8, rb[u & 1][0]), acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[u % HD_RA][0]), __builtin_bit_cast(f16x8, rb[u & 1][0]), acc[2], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  });
#undef HD_LOAD_A
#undef HD_LOAD_B
}

// the same for TWO N-tiles that share their weight fragments (layer 2 of enc_head3: out tiles 3 and 4 of one M-tile in one wave)
template <int GRP, int PL, int MT>
__device__ __forceinline__ void hd_kloop2(f32x16 (&acc)[2][3], const uint4* __restrict__ w, const unsigned char* bbase, const int (&li)[2], int lane, int mt) {
  const int h = lane >> 5;
  uint4 ra[HD_RA][2], rb[2][2][2];
#define HD_LOAD_A(SET, U) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) ra[SET][s_] = w[(unsigned)(((U) * MT + mt) * 2 + s_) * 64u + lane];
#define HD_LOAD_B(SET, U) _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)            \
    rb[SET][t_][s_] = *reinterpret_cast<const uint4*>(bbase + (2 * ((U) / 9) + h) * GRP + s_ * PL + (li[t_] + (((U) % 9) / 3 - 1) * HD_MIDP + (((U) % 9) % 3 - 1)) * 16);
#pragma unroll
  for (int u0 = 0; u0 < HD_RA - 1; ++u0) { HD_LOAD_A(u0, u0) }
  HD_LOAD_B(0, 0)
  HdSteps<0, 18>::run([&](auto uc) {
    constexpr int u = decltype(uc)::value;
    if (u + HD_RA - 1 < 18) { HD_LOAD_A((u + HD_RA - 1) % HD_RA, (u + HD_RA - 1 < 18 ? u + HD_RA - 1 : 17)) }
    if (u + 1 < 18) { HD_LOAD_B((u + 1) & 1, (u + 1 < 18 ? u + 1 : 17)) }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t_ = 0; t_ < 2; ++t_) {
      acc[t_][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[u % HD_RA][0]), __builtin_bit_cast(f16x8, rb[u & 1][t_][1]), acc[t_][0], 0, 0, 0);
      acc[t_][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[u % HD_RA][1]), __builtin_bit_cast(f16x8, rb[u & 1][t_][0]), acc[t_][1], 0, 0, 0);
      acc[t_][2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[u % HD_RA][0]), __builtin_bit_cast(f16x8, rb[u & 1][t_][0]), acc[t_][2], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
#undef HD_LOAD_A
#undef HD_LOAD_B
}

struct HeadArgs {
  FitConst fc;
  const float* verts; int nrows;
  const float* Jtr; int nj;
  const float* transl; int B;
  const float *w0, *b0;            // layer 0: [32][9], [32]
  const uint4* w1; float w1inv;    // layer 1: split-f16 pack (pack_conv3x3_split_f16), its inverse host scale
  const float* b1;
  float *x0, *canon_out, *act1, *act2;
  int ntx, ntiles;
};

__global__ void __launch_bounds__(512)
enc_head_kernel(HeadArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char mid[4 * HD_GRP_MID];           // [group 4][piece 2][216][8 x f16] = 27,648 B
  __shared__ float xs[HD_NX];
  __shared__ float cn[12];
  __shared__ float wmax[8];
  const FitConst& fc = a.fc;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = 3 * fc.n81, H = D + 2, W = a.B - 1 + 16, Wp = W + 2, HWp = (H + 2) * Wp;
  int tile = (int)blockIdx.x;                              // XCD-aware order (workgroup b runs on XCD b % 8): contiguous runs of tiles per XCD
  {
    const int q = a.ntiles >> 3, r = a.ntiles & 7, xcd = tile & 7, k = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int ty = tile / a.ntx, tx = tile - ty * a.ntx;
  const int y0 = ty * HD_TH, x0c = tx * HD_TW;
  // ---- image tile with a halo of 2 (same formula and order of operations as marker_feature_kernel / marker_c1_kernel): reads first
  float va[3] = {0.f, 0.f, 0.f}, vb[3] = {0.f, 0.f, 0.f}, xm = 0.f, xsd = 1.f;
  int cc = 0;
  bool inside = false;
  const int ip = tid < HD_NX ? tid : HD_NX - 1;
  const int ily = ip / HD_XW, ilx = ip - ily * HD_XW;
  {
    const int yy = y0 - 2 + ily, xx = x0c - 2 + ilx;
    inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
    const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
    const int d = reflect_idx(yc - 1, D), tp = reflect_idx(xc - 8, a.B - 1);
    const int m = d / 3;
    cc = d - 3 * m;
    const float* v0 = a.verts + ((size_t)tp * a.nrows + fc.row81[m]) * 3;
    const float* v1 = v0 + (size_t)a.nrows * 3;
#pragma unroll
    for (int e = 0; e < 3; ++e) { va[e] = v0[e]; vb[e] = v1[e]; }
    xm = fc.Xmean[d]; xsd = fc.Xstd[d];
  }
  // the two pad columns of every mid plane hold zeros: 12 rows x 2 x 8 planes of 16 B
  if (tid >= 512 - HD_MIDH * 2 * 8) {
    const int i = tid - (512 - HD_MIDH * 2 * 8), pl = i / (HD_MIDH * 2), rc = i - pl * (HD_MIDH * 2);
    *reinterpret_cast<uint4*>(mid + pl * HD_PL_MID + ((rc >> 1) * HD_MIDP + (rc & 1) * (HD_MIDP - 1)) * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  if (tid == 0) {
    canonical_frame(a.verts, a.nrows, fc.row81, a.Jtr, a.nj, a.transl, cn, fc.cam2world);
    if (blockIdx.x == 0) for (int i = 0; i < 12; ++i) a.canon_out[i] = cn[i];
  }
  __syncthreads();
  if (tid < HD_NX) {
    const float g0 = (va[0] - cn[9]) * cn[cc] + (va[1] - cn[10]) * cn[3 + cc] + (va[2] - cn[11]) * cn[6 + cc];
    const float g1 = (vb[0] - cn[9]) * cn[cc] + (vb[1] - cn[10]) * cn[3 + cc] + (vb[2] - cn[11]) * cn[6 + cc];
    const float n0 = (g0 - xm) / xsd, n1 = (g1 - xm) / xsd;
    const float v = inside ? n1 - n0 : 0.f;                // outside the image: the zero border of the padded x0
    xs[tid] = v;
    if (inside && ily >= 2 && ily < 2 + HD_TH && ilx >= 2 && ilx < 2 + HD_TW)       // the tile's own pixels
      a.x0[(size_t)(y0 + ily - 2 + 1) * Wp + (x0c + ilx - 2 + 1)] = v;
  }
  __syncthreads();
  // ---- layer 0 on the 12 x 16 mid tile: thread = (mid pixel, cout half); waves 0-2 carry couts 0-15, waves 3-5 couts 16-31
  const int halfu = __builtin_amdgcn_readfirstlane(wave / 3);       // wave-uniform: the weights below become scalar loads
  const bool worker = wave < 6;
  const int px = worker ? tid - 192 * halfu : 0;
  const int my = px >> 4, mx = px & 15;
  const int y = y0 - 1 + my, x = x0c - 1 + mx;
  const bool inimg = worker && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
  const bool inner = inimg && my >= 1 && my <= HD_TH && mx >= 1 && mx <= HD_TW;
  float r[16];
  float mloc = 0.f;
  if (worker) {
    float xin[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) xin[tp] = xs[(my + tp / 3) * HD_XW + mx + tp % 3];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float* wc = a.w0 + (size_t)(halfu * 16 + c) * 9;
      float acc = 0.f;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) acc = fmaf(wc[tp], xin[tp], acc);
      const float v = lrelu(acc + a.b0[halfu * 16 + c]);
      r[c] = inimg ? v : 0.f;                              // zero padding of layer 1
      mloc = fmaxf(mloc, fabsf(r[c]));
    }
  } else {
#pragma unroll
    for (int c = 0; c < 16; ++c) r[c] = 0.f;
  }
  mloc = wave_max(mloc);
  if (lane == 0) wmax[wave] = mloc;
  __syncthreads();
  float sm, smi;
  {
    float mm = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) mm = fmaxf(mm, wmax[i]);
    f16_scale_for(mm, sm, smi);
  }
  if (worker) {
    const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y), xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
    const int poff = (yc + 1) * Wp + (xc + 1);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float4 lo4 = make_float4(r[8 * g], r[8 * g + 1], r[8 * g + 2], r[8 * g + 3]);
      const float4 hi4 = make_float4(r[8 * g + 4], r[8 * g + 5], r[8 * g + 6], r[8 * g + 7]);
      if (inner) {                                         // the saved activation act[1] (its 10 x 14 interior)
        float* o = a.act1 + ((size_t)(2 * halfu + g) * HWp + poff) * 8;
        st4(o, lo4); st4(o + 4, hi4);
      }
      uint2 a0, a1, b0, b1;
      split2x4(lo4, sm, a0, a1);
      split2x4(hi4, sm, b0, b1);
      unsigned char* d = mid + (2 * halfu + g) * HD_GRP_MID + (my * HD_MIDP + mx + 1) * 16;
      *reinterpret_cast<uint4*>(d) = make_uint4(a0.x, a0.y, b0.x, b0.y);
      *reinterpret_cast<uint4*>(d + HD_PL_MID) = make_uint4(a1.x, a1.y, b1.x, b1.y);
    }
  }
  __syncthreads();
  // ---- layer 1 (32 -> 32): wave T < 5 owns out N-tile T = rows 2T, 2T + 1 x 16 virtual columns c <-> ox = c - 1 (ox = -1, 14: padding)
  if (wave >= 5) return;
  const int j = lane & 31, h = lane >> 5;
  const int oy = 2 * wave + (j >> 4), c = hd_lane_col(j), ox = c - 1;
  const int lo = (oy + 1) * HD_MIDP + c + 1;
  f32x16 acc[3];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
  hd_kloop<HD_GRP_MID, HD_PL_MID>(acc, a.w1, mid, lo, lane);
  const int yo = y0 + oy, xo = x0c + ox;
  const bool ok = ox >= 0 && ox < HD_TW && yo < H && xo < W;
  const int yoc = yo < H ? yo : H - 1, xoc = xo < 0 ? 0 : (xo < W ? xo : W - 1);
  const int po = (yoc + 1) * Wp + (xoc + 1);
  const float f = smi * a.w1inv;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c0 = q * 8 + 4 * h;
    const float4 bb = ld4(a.b1 + c0);
    float4 v;
    v.x = lrelu(((acc[0][4 * q] + acc[1][4 * q]) + acc[2][4 * q]) * f + bb.x);
    v.y = lrelu(((acc[0][4 * q + 1] + acc[1][4 * q + 1]) + acc[2][4 * q + 1]) * f + bb.y);
    v.z = lrelu(((acc[0][4 * q + 2] + acc[1][4 * q + 2]) + acc[2][4 * q + 2]) * f + bb.z);
    v.w = lrelu(((acc[0][4 * q + 3] + acc[1][4 * q + 3]) + acc[2][4 * q + 3]) * f + bb.w);
    if (ok) st4(a.act2 + ((size_t)(c0 >> 3) * HWp + po) * 8 + (c0 & 7), v);
  }
}

int enc_head(const FitConst& fc, const float* verts, int nrows, const float* Jtr, int nj, const float* transl, int B, const float* w0,
             const float* b0, const void* w1pack, float w1inv, const float* b1, float* x0, float* canon, float* act1, float* act2,
             hipStream_t s) {
  if (B < 10 || !w1pack || !(w1inv > 0.f) || !verts || !x0 || !canon || !act1 || !act2) return LEMO_ERR_ARG;
  const int H = 3 * fc.n81 + 2, W = B - 1 + 16;
  HeadArgs a{};
  a.fc = fc; a.verts = verts; a.nrows = nrows; a.Jtr = Jtr; a.nj = nj; a.transl = transl; a.B = B;
  a.w0 = w0; a.b0 = b0; a.w1 = reinterpret_cast<const uint4*>(w1pack); a.w1inv = w1inv; a.b1 = b1;
  a.x0 = x0; a.canon_out = canon; a.act1 = act1; a.act2 = act2;
  a.ntx = (W + HD_TW - 1) / HD_TW;
  a.ntiles = a.ntx * ((H + HD_TH - 1) / HD_TH);
  hipLaunchKernelGGL(enc_head_kernel, dim3(a.ntiles), dim3(512), 0, s, a);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// enc_head3 (conv variant 8): the head with layer 2 (32 -> 64) on top -- image tile with a halo of 3 (16 x 20), layer 0 on 14 x 18,
// layer 1 on 12 x 16 (6 N-tiles, one wave each), layer 2 on the 10 x 14 tile (5 N-tiles x 2 M-tiles = 10 blocks on 8 waves: waves 0, 1
// run two).  x0, act[1], act[2], act[3] are all written.  Five forward launches remain: head3, pairs (3,4) (5,6) (7,8), layer 9.
constexpr int H3_IW = HD_TW + 6, H3_IH = HD_TH + 6, H3_NI = H3_IW * H3_IH;             // 20 x 16 = 320 image values
constexpr int H3_PL1 = HD_NX * 16, H3_GRP1 = 2 * H3_PL1;                              // layer-0 output planes: 14 x 18 = 252 slots

struct Head3Args {
  HeadArgs h;
  const uint4* w2; float w2inv; const float* b2;          // layer 2: split-f16 pack (cin 32, cout 64), inverse host scale, bias
  float* act3;
};

__global__ void __launch_bounds__(512)
enc_head3_kernel(Head3Args A3) {
  __shared__ __attribute__((aligned(16))) unsigned char p1[4 * H3_GRP1];               // act[1] tile: [group 4][piece 2][252][8 x f16] = 32,256 B
  __shared__ __attribute__((aligned(16))) unsigned char p2[4 * HD_GRP_MID];            // act[2] tile on pitch 18: 27,648 B
  __shared__ float xs[H3_NI];
  __shared__ float cn[12];
  __shared__ float wmax[24];                              // [0..7] layer-0 maxima, [8..15] layer-1 maxima, [20] layer 0's inverse scale
  const HeadArgs& a = A3.h;
  const FitConst& fc = a.fc;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int D = 3 * fc.n81, H = D + 2, W = a.B - 1 + 16, Wp = W + 2, HWp = (H + 2) * Wp;
  int tile = (int)blockIdx.x;
  {
    const int q = a.ntiles >> 3, r = a.ntiles & 7, xcd = tile & 7, k = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int ty = tile / a.ntx, tx = tile - ty * a.ntx;
  const int y0 = ty * HD_TH, x0c = tx * HD_TW;
  // ---- image tile with a halo of 3: reads first
  float va[3] = {0.f, 0.f, 0.f}, vb[3] = {0.f, 0.f, 0.f}, xm = 0.f, xsd = 1.f;
  int cc = 0;
  bool inside = false;
  const int ip = tid < H3_NI ? tid : H3_NI - 1;
  const int ily = ip / H3_IW, ilx = ip - ily * H3_IW;
  {
    const int yy = y0 - 3 + ily, xx = x0c - 3 + ilx;
    inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
    const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
    const int d = reflect_idx(yc - 1, D), tp = reflect_idx(xc - 8, a.B - 1);
    const int m = d / 3;
    cc = d - 3 * m;
    const float* v0 = a.verts + ((size_t)tp * a.nrows + fc.row81[m]) * 3;
    const float* v1 = v0 + (size_t)a.nrows * 3;
#pragma unroll
    for (int e = 0; e < 3; ++e) { va[e] = v0[e]; vb[e] = v1[e]; }
    xm = fc.Xmean[d]; xsd = fc.Xstd[d];
  }
  if (tid >= 512 - HD_MIDH * 2 * 8) {                      // zero pad columns of the act[2] planes
    const int i = tid - (512 - HD_MIDH * 2 * 8), pl = i / (HD_MIDH * 2), rc = i - pl * (HD_MIDH * 2);
    *reinterpret_cast<uint4*>(p2 + pl * HD_PL_MID + ((rc >> 1) * HD_MIDP + (rc & 1) * (HD_MIDP - 1)) * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  if (tid == 0) {
    canonical_frame(a.verts, a.nrows, fc.row81, a.Jtr, a.nj, a.transl, cn, fc.cam2world);
    if (blockIdx.x == 0) for (int i = 0; i < 12; ++i) a.canon_out[i] = cn[i];
  }
  __syncthreads();
  if (tid < H3_NI) {
    const float g0 = (va[0] - cn[9]) * cn[cc] + (va[1] - cn[10]) * cn[3 + cc] + (va[2] - cn[11]) * cn[6 + cc];
    const float g1 = (vb[0] - cn[9]) * cn[cc] + (vb[1] - cn[10]) * cn[3 + cc] + (vb[2] - cn[11]) * cn[6 + cc];
    const float n0 = (g0 - xm) / xsd, n1 = (g1 - xm) / xsd;
    const float v = inside ? n1 - n0 : 0.f;
    xs[tid] = v;
    if (inside && ily >= 3 && ily < 3 + HD_TH && ilx >= 3 && ilx < 3 + HD_TW)
      a.x0[(size_t)(y0 + ily - 3 + 1) * Wp + (x0c + ilx - 3 + 1)] = v;
  }
  __syncthreads();
  // ---- layer 0 on the 14 x 18 tile (halo 2): thread = (pixel, cout half); waves 0-3 couts 0-15, waves 4-7 couts 16-31
  {
    const int halfu = __builtin_amdgcn_readfirstlane(wave >> 2);
    const int px = tid & 255;
    const bool worker = px < HD_NX;
    const int pc = worker ? px : 0;
    const int r1 = pc / HD_XW, c1 = pc - r1 * HD_XW;
    const int y = y0 - 2 + r1, x = x0c - 2 + c1;
    const bool inimg = worker && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const bool inner = inimg && r1 >= 2 && r1 < 2 + HD_TH && c1 >= 2 && c1 < 2 + HD_TW;
    float r[16];
    float mloc = 0.f;
    float xin[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) xin[tp] = xs[(r1 + tp / 3) * H3_IW + c1 + tp % 3];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float* wc = a.w0 + (size_t)(halfu * 16 + c) * 9;
      float acc = 0.f;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) acc = fmaf(wc[tp], xin[tp], acc);
      const float v = lrelu(acc + a.b0[halfu * 16 + c]);
      r[c] = inimg ? v : 0.f;
      mloc = fmaxf(mloc, fabsf(r[c]));
    }
    mloc = wave_max(mloc);
    if (lane == 0) wmax[wave] = mloc;
    __syncthreads();
    float sm, smi;
    {
      float mm = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) mm = fmaxf(mm, wmax[i]);
      f16_scale_for(mm, sm, smi);
    }
    if (lane == 0 && wave == 0) wmax[20] = smi;            // layer 1's waves need the inverse scale
    if (worker) {
      const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y), xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
      const int poff = (yc + 1) * Wp + (xc + 1);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const float4 lo4 = make_float4(r[8 * g], r[8 * g + 1], r[8 * g + 2], r[8 * g + 3]);
        const float4 hi4 = make_float4(r[8 * g + 4], r[8 * g + 5], r[8 * g + 6], r[8 * g + 7]);
        if (inner) {
          float* o = a.act1 + ((size_t)(2 * halfu + g) * HWp + poff) * 8;
          st4(o, lo4); st4(o + 4, hi4);
        }
        uint2 a0, a1, b0, b1;
        split2x4(lo4, sm, a0, a1);
        split2x4(hi4, sm, b0, b1);
        unsigned char* d = p1 + (2 * halfu + g) * H3_GRP1 + pc * 16;
        *reinterpret_cast<uint4*>(d) = make_uint4(a0.x, a0.y, b0.x, b0.y);
        *reinterpret_cast<uint4*>(d + H3_PL1) = make_uint4(a1.x, a1.y, b1.x, b1.y);
      }
    }
  }
  __syncthreads();
  // ---- layer 1 (32 -> 32) on the 12 x 16 tile: wave T < 6 owns N-tile T (mid rows 2T, 2T + 1)
  float4 v1[4];
  float m1 = 0.f;
  int mp2 = 0, mpoff = 0;
  bool inner2 = false;
  if (wave < 6) {
    const int my = 2 * wave + (j >> 4), mx = hd_lane_col(j);
    const int li = (my + 1) * HD_XW + mx + 1;
    f32x16 acc[3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    hd_kloop<H3_GRP1, H3_PL1>(acc, a.w1, p1, li, lane);
    const int y = y0 - 1 + my, x = x0c - 1 + mx;
    const bool inimg = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    inner2 = inimg && my >= 1 && my <= HD_TH && mx >= 1 && mx <= HD_TW;
    const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y), xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
    mpoff = (yc + 1) * Wp + (xc + 1);
    mp2 = my * HD_MIDP + mx + 1;
    const float f = wmax[20] * a.w1inv;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bb = ld4(a.b1 + q * 8 + 4 * h);
      float4 v;
      v.x = lrelu(((acc[0][4 * q] + acc[1][4 * q]) + acc[2][4 * q]) * f + bb.x);
      v.y = lrelu(((acc[0][4 * q + 1] + acc[1][4 * q + 1]) + acc[2][4 * q + 1]) * f + bb.y);
      v.z = lrelu(((acc[0][4 * q + 2] + acc[1][4 * q + 2]) + acc[2][4 * q + 2]) * f + bb.z);
      v.w = lrelu(((acc[0][4 * q + 3] + acc[1][4 * q + 3]) + acc[2][4 * q + 3]) * f + bb.w);
      if (!inimg) v = make_float4(0.f, 0.f, 0.f, 0.f);                // zero padding of layer 2
      v1[q] = v;
      m1 = absmax4(v, m1);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) v1[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  m1 = wave_max(m1);
  if (lane == 0 && wave < 6) wmax[8 + wave] = m1;
  if (lane == 0 && wave >= 6) wmax[8 + wave] = 0.f;
  __syncthreads();
  float sm2, smi2;
  {
    float mm = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) mm = fmaxf(mm, wmax[8 + i]);
    f16_scale_for(mm, sm2, smi2);
  }
  if (wave < 6) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c0 = q * 8 + 4 * h;
      if (inner2) st4(a.act2 + ((size_t)(c0 >> 3) * HWp + mpoff) * 8 + (c0 & 7), v1[q]);
      uint2 s0, s1;
      split2x4(v1[q], sm2, s0, s1);
      unsigned char* d = p2 + q * HD_GRP_MID + mp2 * 16 + 8 * h;
      *reinterpret_cast<uint2*>(d) = s0;
      *reinterpret_cast<uint2*>(d + HD_PL_MID) = s1;
    }
  }
  __syncthreads();
  // ---- layer 2 (32 -> 64) on the 10 x 14 tile: wave w = (M-tile w & 1, tile set w >> 1): out N-tiles {0}, {1}, {2}, {3, 4} -- the last
  // set as ONE K loop over two tiles that share their weight fragments (ten blocks on eight waves with waves 0 and 1 running two K
  // loops in a row measured 17.6 us for the launch: single-tile loops are latency chains)
  {
    const int mt = wave & 1, ts = wave >> 1;
    const float f = smi2 * A3.w2inv;
    auto store_tile = [&](const f32x16 (&acc)[3], int T) {
      const int oy = 2 * T + (j >> 4), c = hd_lane_col(j), ox = c - 1;
      const int yo = y0 + oy, xo = x0c + ox;
      const bool ok = ox >= 0 && ox < HD_TW && yo < H && xo < W;
      const int yoc = yo < H ? yo : H - 1, xoc = xo < 0 ? 0 : (xo < W ? xo : W - 1);
      const int po = (yoc + 1) * Wp + (xoc + 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = mt * 32 + q * 8 + 4 * h;
        const float4 bb = ld4(A3.b2 + c0);
        float4 v;
        v.x = lrelu(((acc[0][4 * q] + acc[1][4 * q]) + acc[2][4 * q]) * f + bb.x);
        v.y = lrelu(((acc[0][4 * q + 1] + acc[1][4 * q + 1]) + acc[2][4 * q + 1]) * f + bb.y);
        v.z = lrelu(((acc[0][4 * q + 2] + acc[1][4 * q + 2]) + acc[2][4 * q + 2]) * f + bb.z);
        v.w = lrelu(((acc[0][4 * q + 3] + acc[1][4 * q + 3]) + acc[2][4 * q + 3]) * f + bb.w);
        if (ok) st4(A3.act3 + ((size_t)(c0 >> 3) * HWp + po) * 8 + (c0 & 7), v);
      }
    };
    auto lo_of = [&](int T) { const int oy = 2 * T + (j >> 4), c = hd_lane_col(j); return (oy + 1) * HD_MIDP + c + 1; };
    if (ts < 3) {
      f32x16 acc[3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
      hd_kloop<HD_GRP_MID, HD_PL_MID, 2>(acc, A3.w2, p2, lo_of(ts), lane, mt);
      store_tile(acc, ts);
    } else {
      f32x16 acc[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[t][p][e] = 0.f;
      const int li2[2] = {lo_of(3), lo_of(4)};
      hd_kloop2<HD_GRP_MID, HD_PL_MID, 2>(acc, A3.w2, p2, li2, lane, mt);
      store_tile(acc[0], 3);
      store_tile(acc[1], 4);
    }
  }
}

int enc_head3(const FitConst& fc, const float* verts, int nrows, const float* Jtr, int nj, const float* transl, int B, const float* w0,
              const float* b0, const void* w1pack, float w1inv, const float* b1, const void* w2pack, float w2inv, const float* b2,
              float* x0, float* canon, float* act1, float* act2, float* act3, hipStream_t s) {
  if (B < 10 || !w1pack || !w2pack || !(w1inv > 0.f) || !(w2inv > 0.f) || !verts || !x0 || !canon || !act1 || !act2 || !act3) return LEMO_ERR_ARG;
  const int H = 3 * fc.n81 + 2, W = B - 1 + 16;
  Head3Args A{};
  HeadArgs& a = A.h;
  a.fc = fc; a.verts = verts; a.nrows = nrows; a.Jtr = Jtr; a.nj = nj; a.transl = transl; a.B = B;
  a.w0 = w0; a.b0 = b0; a.w1 = reinterpret_cast<const uint4*>(w1pack); a.w1inv = w1inv; a.b1 = b1;
  a.x0 = x0; a.canon_out = canon; a.act1 = act1; a.act2 = act2;
  a.ntx = (W + HD_TW - 1) / HD_TW;
  a.ntiles = a.ntx * ((H + HD_TH - 1) / HD_TH);
  A.w2 = reinterpret_cast<const uint4*>(w2pack); A.w2inv = w2inv; A.b2 = b2; A.act3 = act3;
  hipLaunchKernelGGL(enc_head3_kernel, dim3(a.ntiles), dim3(512), 0, s, A);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
struct TailArgs {
  const float* din;                // d(pre-act 2): CG8P, 32 channels
  const uint4* w1b; float w1binv;  // layer 1 backward pack (pack_conv3x3_bwd_split_f16)
  const float* act1;               // saved activation of layer 0 (lrelu' operand), CG8P 32 channels
  const float* w0;                 // layer 0 weights [32][9]
  float* dx0;                      // [H * W], unpadded
  int H, W, ntx, ntiles;
};

__global__ void __launch_bounds__(512)
enc_tail_kernel(TailArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char inp[4 * HD_GRP_IN];            // [group 4][piece 2][252][8 x f16] = 32,256 B
  __shared__ __attribute__((aligned(16))) float d1[32 * HD_NMIDP];                     // d(pre-act 1) [cout][12 x 18]: 27,648 B
  __shared__ float wmax[8];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, W = a.W, Wp = W + 2, HWp = (H + 2) * Wp;
  int tile = (int)blockIdx.x;
  {
    const int q = a.ntiles >> 3, r = a.ntiles & 7, xcd = tile & 7, k = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int ty = tile / a.ntx, tx = tile - ty * a.ntx;
  const int y0 = ty * HD_TH, x0c = tx * HD_TW;
  // ---- staging: 14 x 18 input tile, 4 channel groups x 2 halves x 252 px = 2016 float4 chunks, 4 slots per thread (clamped coordinates
  // land on the zero border ring of the CG8P map; a pixel two steps out only feeds mid pixels that are masked below)
  float4 st[4];
  int dst[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c0 = tid + k * 512;
    c0 = c0 < 4 * 2 * HD_NX ? c0 : 4 * 2 * HD_NX - 1;
    const int gg = c0 / (2 * HD_NX), c = c0 - gg * (2 * HD_NX);
    const int px = c >> 1, half = c & 1;
    const int r = px / HD_XW, col = px - r * HD_XW;
    int gy = y0 - 2 + r, gx = x0c - 2 + col;
    gy = (gy < -1 ? -1 : (gy > H ? H : gy)) + 1;
    gx = (gx < -1 ? -1 : (gx > W ? W : gx)) + 1;
    st[k] = ld4(a.din + ((size_t)gg * HWp + gy * Wp + gx) * 8 + 4 * half);
    dst[k] = gg * HD_GRP_IN + px * 16 + 8 * half;
  }
  float sc, sci;
  {
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) m = absmax4(st[k], m);
    m = wave_max(m);
    if (lane == 0) wmax[wave] = m;
    __syncthreads();
    float mm = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) mm = fmaxf(mm, wmax[i]);
    f16_scale_for(mm, sc, sci);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint2 s0, s1;
    split2x4(st[k], sc, s0, s1);
    *reinterpret_cast<uint2*>(inp + dst[k]) = s0;
    *reinterpret_cast<uint2*>(inp + dst[k] + HD_PL_IN) = s1;
  }
  __syncthreads();
  // ---- layer 1 backward-data (32 -> 32): wave T < 6 owns mid N-tile T = mid rows 2T, 2T + 1
  if (wave < 6) {
    const int j = lane & 31, h = lane >> 5;
    const int my = 2 * wave + (j >> 4), mx = hd_lane_col(j);
    const int li = (my + 1) * HD_XW + mx + 1;
    f32x16 acc[3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    const int y = y0 - 1 + my, x = x0c - 1 + mx;
    const bool inimg = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y), xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
    const int poff = (yc + 1) * Wp + (xc + 1);
    float4 aux[4];                                          // saved activation at the mid position: requested before the K loop
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int c0 = q * 8 + 4 * h; aux[q] = ld4(a.act1 + ((size_t)(c0 >> 3) * HWp + poff) * 8 + (c0 & 7)); }
    hd_kloop<HD_GRP_IN, HD_PL_IN>(acc, a.w1b, inp, li, lane);
    const float f = sci * a.w1binv;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c0 = q * 8 + 4 * h;
      float v[4];
      const float ax[4] = {aux[q].x, aux[q].y, aux[q].z, aux[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s = ((acc[0][4 * q + e] + acc[1][4 * q + e]) + acc[2][4 * q + e]) * f * lrelu_grad_from_out(ax[e]);
        v[e] = inimg ? s : 0.f;                             // outside the image there is no d(pre-act 1)
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) d1[(c0 + e) * HD_NMIDP + my * HD_MIDP + mx + 1] = v[e];
    }
  }
  __syncthreads();
  // ---- layer 0 adjoint on the 10 x 14 tile: dx0[y][x] = sum_co sum_tap d1[co][y - dy][x - dx] w0[co][tap] in conv3x3_c1_bwd_kernel's
  // order (groups, taps, channels); mid coordinates of the operand: (oy + 1 - dy, ox + 1 - dx)
  if (tid < HD_TH * HD_TW) {
    const int oy = tid / HD_TW, ox = tid - oy * HD_TW;
    const int y = y0 + oy, x = x0c + ox;
    if (y < H && x < W) {
      float acc = 0.f;
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int dy = t / 3 - 1, dx = t % 3 - 1;
          const float* q = d1 + (size_t)(g * 8) * HD_NMIDP + (oy + 1 - dy) * HD_MIDP + (ox + 1 - dx) + 1;
          const float* wc = a.w0 + (size_t)(g * 8) * 9 + t;
#pragma unroll
          for (int c = 0; c < 8; ++c) acc = fmaf(q[c * HD_NMIDP], wc[9 * c], acc);
        }
      }
      a.dx0[(size_t)y * W + x] = acc;
    }
  }
}

int enc_tail(const float* din, const void* w1bpack, float w1binv, const float* act1, const float* w0, float* dx0, int H, int W, hipStream_t s) {
  if (!din || !w1bpack || !(w1binv > 0.f) || !act1 || !w0 || !dx0 || H < 1 || W < 1) return LEMO_ERR_ARG;
  TailArgs a{};
  a.din = din; a.w1b = reinterpret_cast<const uint4*>(w1bpack); a.w1binv = w1binv; a.act1 = act1; a.w0 = w0; a.dx0 = dx0;
  a.H = H; a.W = W;
  a.ntx = (W + HD_TW - 1) / HD_TW;
  a.ntiles = a.ntx * ((H + HD_TH - 1) / HD_TH);
  hipLaunchKernelGGL(enc_tail_kernel, dim3(a.ntiles), dim3(512), 0, s, a);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// enc_tail3 (conv variant 9): the tail with layer 2's backward-data (64 -> 32) in front -- d(pre-act 3) tile with a halo of 3 (16 x 20,
// 64 channels: 80 KB of LDS as two fp16 pieces), layer 2 backward x lrelu'(act[2]) on the 14 x 18 tile, then enc_tail's two stages
// (layer 1 backward x lrelu'(act[1]) on 12 x 16, layer 0 adjoint on 10 x 14).  d(pre-act 2) never leaves the CU; with enc_head3 no
// 1-D-tiled launch is left on the 32-channel side of the encoder.  Layer 2's output pixels are the 252 positions of the 14 x 18 tile
// in row-major order, 32 per N-tile (wave T = tile T, K = 36 steps): a tile's lanes cross rows of the pitch-20 input planes, which
// costs two 2-way bank conflicts per fragment read -- the B reads of this loop are 8 cycles of LDS next to 96 of MFMA.
constexpr int T3_IW = HD_TW + 6, T3_IH = HD_TH + 6, T3_NI = T3_IW * T3_IH;             // 20 x 16 = 320 input pixels
constexpr int T3_PL = T3_NI * 16, T3_GRP = 2 * T3_PL;                                  // input planes: 5120 B each, 8 groups = 81,920 B
constexpr int T3_MID_OFF = 8 * T3_GRP;                                                 // d(pre-act 2) planes (enc_tail's `inp`): 32,256 B
constexpr int T3_WMAX_OFF = T3_MID_OFF + 4 * HD_GRP_IN;
constexpr int T3_SMEM = T3_WMAX_OFF + 16 * 4;                                          // 114,240 B; d(pre-act 1) (27,648 B) reuses the input planes
static_assert(32 * HD_NMIDP * 4 <= T3_MID_OFF, "d(pre-act 1) fits in the dead input planes");

struct Tail3Args {
  TailArgs t;                      // din = d(pre-act 3): CG8P, 64 channels
  const uint4* w2b; float w2binv;  // layer 2 backward pack (pack_conv3x3_bwd_split_f16: cin 64, cout 32)
  const float* act2;               // saved activation of layer 1 (lrelu' operand), CG8P 32 channels
};

__global__ void __launch_bounds__(512)
enc_tail3_kernel(Tail3Args A3) {
  LEMO_DYN_SMEM(smem_f);
  unsigned char* smem = reinterpret_cast<unsigned char*>(smem_f);
  unsigned char* inp = smem + T3_MID_OFF;
  float* d1 = smem_f;
  float* wmax = reinterpret_cast<float*>(smem + T3_WMAX_OFF);       // [0..7] input maxima, [8..15] d(pre-act 2) maxima
  const TailArgs& a = A3.t;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W, Wp = W + 2, HWp = (H + 2) * Wp;
  int tile = (int)blockIdx.x;
  {
    const int q = a.ntiles >> 3, r = a.ntiles & 7, xcd = tile & 7, k = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int ty = tile / a.ntx, tx = tile - ty * a.ntx;
  const int y0 = ty * HD_TH, x0c = tx * HD_TW;
  // ---- staging: 16 x 20 input tile, 8 channel groups x 2 halves x 320 px = 5120 float4 chunks, 10 slots per thread (coordinates
  // outside the image clamp onto the zero border ring of the CG8P map: no gradient there)
  float sc, sci;
  {
    float4 st[10];
    int dst[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int c0 = tid + k * 512;
      const int gg = c0 / (2 * T3_NI), c = c0 - gg * (2 * T3_NI);
      const int px = c >> 1, half = c & 1;
      const int r = px / T3_IW, col = px - r * T3_IW;
      int gy = y0 - 3 + r, gx = x0c - 3 + col;
      gy = (gy < -1 ? -1 : (gy > H ? H : gy)) + 1;
      gx = (gx < -1 ? -1 : (gx > W ? W : gx)) + 1;
      st[k] = ld4(a.din + ((size_t)gg * HWp + gy * Wp + gx) * 8 + 4 * half);
      dst[k] = gg * T3_GRP + px * 16 + 8 * half;
    }
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < 10; ++k) m = absmax4(st[k], m);
    m = wave_max(m);
    if (lane == 0) wmax[wave] = m;
    __syncthreads();
    float mm = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) mm = fmaxf(mm, wmax[i]);
    f16_scale_for(mm, sc, sci);
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      uint2 s0, s1;
      split2x4(st[k], sc, s0, s1);
      *reinterpret_cast<uint2*>(smem + dst[k]) = s0;
      *reinterpret_cast<uint2*>(smem + dst[k] + T3_PL) = s1;
    }
  }
  __syncthreads();
  // ---- layer 2 backward-data (64 -> 32) on the 14 x 18 tile: wave T owns positions 32 T .. 32 T + 31 of its row-major order
  {
    const int o = 32 * wave + j, oc = o < HD_NX ? o : HD_NX - 1;
    const int r = oc / HD_XW, c = oc - r * HD_XW;
    const int li = (r + 1) * T3_IW + c + 1;
    const int y = y0 - 2 + r, x = x0c - 2 + c;
    const bool inimg = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y), xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
    const int poff = (yc + 1) * Wp + (xc + 1);
    float4 aux[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int c0 = q * 8 + 4 * h; aux[q] = ld4(A3.act2 + ((size_t)(c0 >> 3) * HWp + poff) * 8 + (c0 & 7)); }
    f32x16 acc[3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    hd_kloop<T3_GRP, T3_PL, 1, 36, T3_IW>(acc, A3.w2b, smem, li, lane);
    const float f = sci * A3.w2binv;
    float4 v2[4];
    float m2 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float ax[4] = {aux[q].x, aux[q].y, aux[q].z, aux[q].w};
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s = ((acc[0][4 * q + e] + acc[1][4 * q + e]) + acc[2][4 * q + e]) * f * lrelu_grad_from_out(ax[e]);
        v[e] = inimg ? s : 0.f;                             // outside the image there is no d(pre-act 2)
      }
      v2[q] = make_float4(v[0], v[1], v[2], v[3]);
      m2 = absmax4(v2[q], m2);
    }
    m2 = wave_max(m2);
    if (lane == 0) wmax[8 + wave] = m2;
    __syncthreads();
    float mm = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) mm = fmaxf(mm, wmax[8 + i]);
    f16_scale_for(mm, sc, sci);
    if (o < HD_NX) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint2 s0, s1;
        split2x4(v2[q], sc, s0, s1);
        unsigned char* d = inp + q * HD_GRP_IN + o * 16 + 8 * h;
        *reinterpret_cast<uint2*>(d) = s0;
        *reinterpret_cast<uint2*>(d + HD_PL_IN) = s1;
      }
    }
  }
  __syncthreads();
  // ---- layer 1 backward-data (32 -> 32): wave T < 6 owns mid N-tile T = mid rows 2T, 2T + 1 (enc_tail's stage)
  if (wave < 6) {
    const int my = 2 * wave + (j >> 4), mx = hd_lane_col(j);
    const int li = (my + 1) * HD_XW + mx + 1;
    f32x16 acc[3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    const int y = y0 - 1 + my, x = x0c - 1 + mx;
    const bool inimg = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y), xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
    const int poff = (yc + 1) * Wp + (xc + 1);
    float4 aux[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int c0 = q * 8 + 4 * h; aux[q] = ld4(a.act1 + ((size_t)(c0 >> 3) * HWp + poff) * 8 + (c0 & 7)); }
    hd_kloop<HD_GRP_IN, HD_PL_IN>(acc, a.w1b, inp, li, lane);
    const float f = sci * a.w1binv;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c0 = q * 8 + 4 * h;
      float v[4];
      const float ax[4] = {aux[q].x, aux[q].y, aux[q].z, aux[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s = ((acc[0][4 * q + e] + acc[1][4 * q + e]) + acc[2][4 * q + e]) * f * lrelu_grad_from_out(ax[e]);
        v[e] = inimg ? s : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) d1[(c0 + e) * HD_NMIDP + my * HD_MIDP + mx + 1] = v[e];
    }
  }
  __syncthreads();
  // ---- layer 0 adjoint on the 10 x 14 tile, in conv3x3_c1_bwd_kernel's order (enc_tail's stage)
  if (tid < HD_TH * HD_TW) {
    const int oy = tid / HD_TW, ox = tid - oy * HD_TW;
    const int y = y0 + oy, x = x0c + ox;
    if (y < H && x < W) {
      float acc = 0.f;
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int dy = t / 3 - 1, dx = t % 3 - 1;
          const float* q = d1 + (size_t)(g * 8) * HD_NMIDP + (oy + 1 - dy) * HD_MIDP + (ox + 1 - dx) + 1;
          const float* wc = a.w0 + (size_t)(g * 8) * 9 + t;
#pragma unroll
          for (int c = 0; c < 8; ++c) acc = fmaf(q[c * HD_NMIDP], wc[9 * c], acc);
        }
      }
      a.dx0[(size_t)y * W + x] = acc;
    }
  }
}

int enc_tail3(const float* din, const void* w2bpack, float w2binv, const float* act2, const void* w1bpack, float w1binv, const float* act1,
              const float* w0, float* dx0, int H, int W, hipStream_t s) {
  if (!din || !w2bpack || !(w2binv > 0.f) || !act2 || !w1bpack || !(w1binv > 0.f) || !act1 || !w0 || !dx0 || H < 1 || W < 1) return LEMO_ERR_ARG;
  // the opt-in to T3_SMEM bytes of dynamic LDS is a per-device attribute of the function: once per device of this process,
  // after checking that the device has that much (gfx950: 160 KiB; anything smaller fails loudly here, not at the launch)
  static int rcs[64];                                       // 0 = not asked yet, 1 = ok, else -(hipError_t) - 1
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LEMO_ERR_ARG;
  if (rcs[dev] == 0) {
    int lds = 0;
    hipError_t e = hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
    if (e == hipSuccess && lds < T3_SMEM) e = hipErrorInvalidValue;
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_tail3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, T3_SMEM);
    rcs[dev] = e == hipSuccess ? 1 : -(int)e - 1;
  }
  if (rcs[dev] != 1) return -(rcs[dev] + 1);
  Tail3Args A{};
  TailArgs& a = A.t;
  a.din = din; a.w1b = reinterpret_cast<const uint4*>(w1bpack); a.w1binv = w1binv; a.act1 = act1; a.w0 = w0; a.dx0 = dx0;
  a.H = H; a.W = W;
  a.ntx = (W + HD_TW - 1) / HD_TW;
  a.ntiles = a.ntx * ((H + HD_TH - 1) / HD_TH);
  A.w2b = reinterpret_cast<const uint4*>(w2bpack); A.w2binv = w2binv; A.act2 = act2;
  hipLaunchKernelGGL(enc_tail3_kernel, dim3(a.ntiles), dim3(512), T3_SMEM, s, A);
  return (int)hipGetLastError();
}

}  // namespace lemo
