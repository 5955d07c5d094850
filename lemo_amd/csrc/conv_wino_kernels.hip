// ONE 64 -> 64 3x3 layer of the smoothness encoder (models/AE_sep.py:11-30, 77-99), forward or backward-data, as a Winograd
// F(2 x 2, 3 x 3) convolution on the f16 matrix cores ("conv variant 10"; VERDICT r05 #1 candidate (c)).
//
// Why: every direct form of this layer issues 2 * 32830 * 64 * 576 / (32*32*16*2) * 3 products = 864 MFMAs per CU (the fused pair:
// 1.375 x that for its recomputed halo) and rounds 3-5 left the pair at 0.25 of the matrix roof with its K loops already at 72-82 %
// issue: the matrix work itself had to shrink.  F(2x2, 3x3) computes a 2 x 2 output tile from a 4 x 4 input patch with 16 instead of
// 36 multiplies per (cin, cout): Y = A^T [ (G g G^T) (.) (B^T d B) ] A, the (.) summed over cin = 16 independent
// [64 cout] x [64 cin] x [tiles] GEMMs, one per position of the 4 x 4 transformed patch: 384 MFMAs per CU and layer (2.25 x fewer).
//
// Arithmetic: the transforms are fp32 adds (B^T, A^T hold 0, +-1 only; G g G^T is evaluated in float64 on the host); the 16 GEMMs are
// split-f16 exactly as conv variant 4 (conv_f16.hpp): two error-compensated fp16 pieces per operand, three v_mfma_f32_32x32x16_f16
// per 16-deep k-chunk, fp32 accumulate, transformed weights pre-split on the host with one power-of-two scale, transformed
// activations scaled per workgroup.  Measured per-layer error vs float64 on the encoder's own weights and activations:
// 2.1e-7 .. 4.0e-7 of the layer maximum forward and backward-data, against 2.0e-7 .. 3.7e-7 for torch's fp32 convolution
// (tools/wino_numerics.py, profiles/r06_wino_numerics.txt): the 64-term accumulations are shorter than the direct form's 576.
//
// Decomposition: the even part of the image (H2 = H & ~1 rows) is cut into 2 x 2 tiles, ntx = ceil(W / 2) per tile row, numbered
// row-major; a workgroup (8 waves, one per CU) owns 32 consecutive tiles = ONE MFMA N-tile: 122 x 67 = 8174 tiles = 256 workgroups at
// 245 x 134.  (W odd: the last tile column's second output column is computed and dropped.)  The last image row of an odd H is a
// plain fp32 direct convolution spread over the workgroups (W * 64 outputs, 16 lanes each: wn_odd_row), from the tap-major fp32 pack.
//   phase 1  thread (tile n = tid & 31, channel quad cq = tid >> 5): 16 ld4 = its 4 x 4 patch of 4 channels straight from global
//            (CG8P: a pixel's 8 channels are 32 contiguous bytes), V = B^T d B in registers, workgroup maximum -> power-of-two scale,
//            two fp16 pieces into LDS as MFMA B fragments: V[pos 16][piece 2][group 8][tile 32][8 x f16] = 128 KB
//   phase 2  wave (mt = cout half, i = row of the 4 x 4): positions (i, j = 0..3), K = 64 in four 16-deep steps, products rotated
//            over the four accumulators (no back-to-back dependent MFMAs); A = transformed weights from L2 (32 KB per wave and layer
//            -- the stream that bounds this phase: 256 KB per workgroup through a 64 B/clk port), the first two steps requested before
//            phase 1, the others two steps ahead
//   phase 3  A^T . A: over j in registers, over i through LDS (the V planes are dead): wave (mt, a, b) finishes output pixel (a, b) of
//            every tile for its 32 couts: bias + LeakyReLU (forward) or x lrelu'(saved activation) (backward-data), dwordx4 stores.
#include "conv_common.hpp"
#include "conv_f16.hpp"

namespace lemo {

constexpr int WN_TILES = 32;                              // 2 x 2 output tiles per workgroup = one MFMA N-tile
constexpr int WN_PLANE = WN_TILES * 16;                   // bytes of one (position, piece, channel group) plane: 32 tiles x 8 f16
constexpr int WN_V_BYTES = 16 * 2 * 8 * WN_PLANE;         // 131,072
constexpr int WN_X_BYTES = 2 * 4 * 2 * 4 * 64 * 16;       // phase-3 exchange [mt][i][b][quad][lane] float4: 65,536 (aliases the V planes)
constexpr int WN_WMAX_OFF = WN_V_BYTES;
constexpr int WN_SMEM = WN_V_BYTES + 8 * 4;
static_assert(WN_X_BYTES <= WN_V_BYTES, "the exchange buffer reuses the dead V planes");

struct WinoArgs {
  const float* in;                 // CG8P, 64 channels
  const uint4* wU;                 // transformed weights, split-f16: [pos 16][kstep 4][mt 2][piece 2][lane 64][8 x f16]
  const float* wt;                 // fp32 tap-major pack wt[tap][8][64][8] of the same layer (odd last row only)
  const float* bias;               // EPI 0
  const float* aux;                // EPI 1: saved forward activation at the output positions
  float* out;
  float winv;                      // 2^-k of the host-side scale of wU
  int H, W, H2, ntx, T, nwg;
  unsigned long long* dbg;
};

// the last row of an image with odd H: out[H-1][x][co] as a direct fp32 convolution, W * 64 outputs, one per 16-lane row (lane = 4 input
// channels x 9 taps, DPP row sum): every workgroup takes 32 of them with operands requested before anything else in the kernel (their
// round trip overlaps the patch loads'), the outputs beyond 32 per workgroup go to the first few workgroups at the END of the kernel.
struct WnOdd { float4 v[9], w[9]; int x, co; bool ok; };
__device__ __forceinline__ void wn_odd_load(const WinoArgs& a, int tid, int oi, WnOdd& r) {
  const int H = a.H, W = a.W, Wp = W + 2, HWp = (H + 2) * Wp, total = W * 64;
  const int sub = tid & 15, g = sub >> 1, half = sub & 1;
  r.ok = oi < total;
  const int oc = r.ok ? oi : total - 1;
  r.x = oc >> 6; r.co = oc & 63;
  const float* ip = a.in + ((size_t)g * HWp + (size_t)(H - 1) * Wp + r.x) * 8 + 4 * half;      // padded (row H - 1, col x) = tap (-1, -1)
  const float* wp = a.wt + ((size_t)g * 64 + r.co) * 8 + 4 * half;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    r.v[t] = ld4(ip + ((t / 3) * Wp + (t % 3)) * 8);
    r.w[t] = ld4(wp + (size_t)t * 8 * 64 * 8);
  }
}
template <int EPI>
__device__ __forceinline__ void wn_odd_finish(const WinoArgs& a, int tid, const WnOdd& r) {
  const int H = a.H, Wp = a.W + 2, HWp = (H + 2) * Wp;
  float acc = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    acc = fmaf(r.v[t].x, r.w[t].x, acc); acc = fmaf(r.v[t].y, r.w[t].y, acc);
    acc = fmaf(r.v[t].z, r.w[t].z, acc); acc = fmaf(r.v[t].w, r.w[t].w, acc);
  }
  acc = row16_sum(acc);
  if (r.ok && (tid & 15) == 0) {
    const size_t o = ((size_t)(r.co >> 3) * HWp + (size_t)H * Wp + r.x + 1) * 8 + (r.co & 7);
    conv_store1<EPI>(a.out, a.bias, a.aux, o, r.co, acc);
  }
}

template <int EPI, bool DBG>
__global__ void __launch_bounds__(512)
conv3x3_wino_kernel(WinoArgs a) {
  unsigned long long t_start = 0, t_ld = 0, t_tr = 0, t_p1 = 0, t_mm = 0, t_x = 0;
  if (DBG) t_start = __builtin_amdgcn_s_memtime();
  LEMO_DYN_SMEM(smem_f);
  unsigned char* smem = reinterpret_cast<unsigned char*>(smem_f);
  float* wmax = reinterpret_cast<float*>(smem + WN_WMAX_OFF);
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, Wp = W + 2, HWp = (H + 2) * Wp;
  // XCD-aware order (workgroup b runs on XCD b % 8): XCD x owns a contiguous run of tile groups (vertical neighbours share input rows in one L2)
  int grp = (int)blockIdx.x;
  {
    const int q = a.nwg >> 3, r = a.nwg & 7, xcd = grp & 7, k = grp >> 3;
    grp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }

  const bool odd = a.H2 < H;
  WnOdd orow;
  // (unconditional: a branch here lets hipcc merge it with the one around wn_odd_finish and wait for these loads BEFORE the patch loads
  // are issued; an even H computes one row of throw-away sums from valid addresses)
  wn_odd_load(a, tid, (int)blockIdx.x * 32 + (tid >> 4), orow);              // the oldest loads of the kernel: back first
  orow.ok = orow.ok && odd;
  // ---- phase 2's first weight fragments: nothing depends on them either ---------------------------------------------------------
  const int mt = wave & 1, fi = wave >> 1;
  uint4 ra[2][4][2];                                       // [step parity][j][piece]
#define WN_LOAD_A(SET, KS)                                                                                      \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                              \
    _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_)                                                            \
      ra[SET][j_][p_] = a.wU[(unsigned)(((((4 * fi + j_) * 4 + (KS)) * 2 + mt) * 2 + p_) * 64) + lane];
  WN_LOAD_A(0, 0)
  WN_LOAD_A(1, 1)

  // ---- phase 1: patch of tile n, channels 4 cq .. 4 cq + 3 ---------------------------------------------------------------------
  float4 d[4][4];
  {
    const int n = tid & 31, cq = tid >> 5, g = cq >> 1, hf = cq & 1;
    int t = grp * WN_TILES + n;
    t = t < a.T ? t : a.T - 1;                               // surplus slots of the last group redo its last tile (never stored)
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    const float* ip = a.in + ((size_t)g * HWp + (size_t)(2 * ty) * Wp) * 8 + 4 * hf;      // padded row 2 ty = image row 2 ty - 1
    int col[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) col[c] = 2 * tx + c <= W + 1 ? 2 * tx + c : W + 1;        // W odd: the column past the border ring reads the ring (0)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) d[r][c] = ld4(ip + ((size_t)r * Wp + col[c]) * 8);
  }
  __builtin_amdgcn_sched_barrier(0);
  wn_odd_finish<EPI>(a, tid, orow);                        // waits for ITS loads only (the patch loads and weight fragments stay in flight)
  __builtin_amdgcn_sched_barrier(0);
  if (DBG) t_ld = __builtin_amdgcn_s_memtime();
  // V = B^T d B per channel: B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
  float4 V[4][4];
  float m = 0.f;
  {
#define WN_BT(o0, o1, o2, o3, i0, i1, i2, i3) { o0 = i0 - i2; o1 = i1 + i2; o2 = i2 - i1; o3 = i1 - i3; }
    float4 tt[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      WN_BT(tt[0][c].x, tt[1][c].x, tt[2][c].x, tt[3][c].x, d[0][c].x, d[1][c].x, d[2][c].x, d[3][c].x)
      WN_BT(tt[0][c].y, tt[1][c].y, tt[2][c].y, tt[3][c].y, d[0][c].y, d[1][c].y, d[2][c].y, d[3][c].y)
      WN_BT(tt[0][c].z, tt[1][c].z, tt[2][c].z, tt[3][c].z, d[0][c].z, d[1][c].z, d[2][c].z, d[3][c].z)
      WN_BT(tt[0][c].w, tt[1][c].w, tt[2][c].w, tt[3][c].w, d[0][c].w, d[1][c].w, d[2][c].w, d[3][c].w)
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      WN_BT(V[i][0].x, V[i][1].x, V[i][2].x, V[i][3].x, tt[i][0].x, tt[i][1].x, tt[i][2].x, tt[i][3].x)
      WN_BT(V[i][0].y, V[i][1].y, V[i][2].y, V[i][3].y, tt[i][0].y, tt[i][1].y, tt[i][2].y, tt[i][3].y)
      WN_BT(V[i][0].z, V[i][1].z, V[i][2].z, V[i][3].z, tt[i][0].z, tt[i][1].z, tt[i][2].z, tt[i][3].z)
      WN_BT(V[i][0].w, V[i][1].w, V[i][2].w, V[i][3].w, tt[i][0].w, tt[i][1].w, tt[i][2].w, tt[i][3].w)
#pragma unroll
      for (int j = 0; j < 4; ++j) m = absmax4(V[i][j], m);
    }
#undef WN_BT
  }
  m = wave_max(m);
  if (lane == 0) wmax[wave] = m;
  __syncthreads();
  float sv, svi;
  {
    float mm = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) mm = fmaxf(mm, wmax[i]);
    f16_scale_for(mm, sv, svi);
  }
  if (DBG) t_tr = __builtin_amdgcn_s_memtime();
  {
    const int n = tid & 31, cq = tid >> 5, g = cq >> 1, hf = cq & 1;
    unsigned char* vp = smem + g * WN_PLANE + n * 16 + 8 * hf;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint2 s0, s1;
        split2x4(V[i][j], sv, s0, s1);
        unsigned char* p = vp + ((4 * i + j) * 2) * (8 * WN_PLANE);
        *reinterpret_cast<uint2*>(p) = s0;
        *reinterpret_cast<uint2*>(p + 8 * WN_PLANE) = s1;
      }
  }
  __syncthreads();
  if (DBG) t_p1 = __builtin_amdgcn_s_memtime();

  // ---- phase 2: positions (fi, 0..3), couts 32 mt .. 32 mt + 31, all 32 tiles ----------------------------------------------------
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  {
    const int h = lane >> 5;
    const unsigned char* bp = smem + ((4 * fi) * 2) * (8 * WN_PLANE) + h * WN_PLANE + (lane & 31) * 16;
    uint4 rb[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          rb[j][p] = *reinterpret_cast<const uint4*>(bp + ((j * 2 + p) * 8 + 2 * ks) * WN_PLANE);
      __builtin_amdgcn_sched_barrier(0);
#define WN_MFMA(PA, PB)                                                                                       \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                            \
    acc[j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[ks & 1][j_][PA]),           \
                                                     __builtin_bit_cast(f16x8, rb[j_][PB]), acc[j_], 0, 0, 0);
      WN_MFMA(0, 1) WN_MFMA(1, 0) WN_MFMA(0, 0)             // smallest products first
#undef WN_MFMA
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 2 < 4) { WN_LOAD_A(ks & 1, ks + 2) }         // the slot these MFMAs just consumed
    }
  }
#undef WN_LOAD_A
  if (DBG) t_mm = __builtin_amdgcn_s_memtime();

  // ---- phase 3: Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]: over j here, over i through LDS ----------------------------------------
  const int mt2 = wave & 1, oa = (wave >> 1) & 1, ob = wave >> 2;      // this wave finishes output pixel (oa, ob) of every tile, couts of mt2
  const int h = lane >> 5;
  int po;                          // padded pixel offset of (tile of this lane, oa, ob)
  bool ok;
  {
    const int n = lane & 31;
    const int t = grp * WN_TILES + n;
    const int tc = t < a.T ? t : a.T - 1;
    const int ty = tc / a.ntx, tx = tc - ty * a.ntx;
    const int y = 2 * ty + oa, x = 2 * tx + ob;
    ok = t < a.T && x < W;
    po = (y + 1) * Wp + (x < W ? x : W - 1) + 1;
  }
  float4 eo[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {                             // epilogue operands: requested before the exchange barriers
    const int c0 = mt2 * 32 + q * 8 + 4 * h;
    eo[q] = EPI == 1 ? ld4(a.aux + ((size_t)(c0 >> 3) * HWp + po) * 8 + (c0 & 7)) : ld4(a.bias + c0);
  }
  {
    const float f = svi * a.winv;                           // back to the operands' own scale (exact: powers of two)
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][r] * f, m1 = acc[1][r] * f, m2 = acc[2][r] * f, m3 = acc[3][r] * f;
      s0[r] = (m0 + m1) + m2;
      s1[r] = (m1 - m2) - m3;
    }
    __syncthreads();                                        // every wave is done reading the V planes
    float* xw = smem_f + (size_t)(((mt * 4 + fi) * 2) * 4) * 256 + lane * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      st4(xw + q * 256, make_float4(s0[4 * q], s0[4 * q + 1], s0[4 * q + 2], s0[4 * q + 3]));
      st4(xw + (4 + q) * 256, make_float4(s1[4 * q], s1[4 * q + 1], s1[4 * q + 2], s1[4 * q + 3]));
    }
  }
  __syncthreads();
  if (DBG) t_x = __builtin_amdgcn_s_memtime();
  {
    // rows i = oa .. oa + 2 of column ob: oa = 0: r0 + r1 + r2 ; oa = 1: r1 - r2 - r3
    const float* xr = smem_f + (size_t)(((mt2 * 4 + oa) * 2 + ob) * 4) * 256 + lane * 4;
    constexpr int ISTR = 2 * 4 * 256;                       // floats between consecutive i
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 r0 = ld4(xr + q * 256), r1 = ld4(xr + ISTR + q * 256), r2 = ld4(xr + 2 * ISTR + q * 256);
      float4 v;
      if (oa == 0) v = make_float4((r0.x + r1.x) + r2.x, (r0.y + r1.y) + r2.y, (r0.z + r1.z) + r2.z, (r0.w + r1.w) + r2.w);
      else v = make_float4((r0.x - r1.x) - r2.x, (r0.y - r1.y) - r2.y, (r0.z - r1.z) - r2.z, (r0.w - r1.w) - r2.w);
      if (EPI == 1) {
        v.x *= lrelu_grad_from_out(eo[q].x); v.y *= lrelu_grad_from_out(eo[q].y);
        v.z *= lrelu_grad_from_out(eo[q].z); v.w *= lrelu_grad_from_out(eo[q].w);
      } else {
        v.x = lrelu(v.x + eo[q].x); v.y = lrelu(v.y + eo[q].y); v.z = lrelu(v.z + eo[q].z); v.w = lrelu(v.w + eo[q].w);
      }
      const int c0 = mt2 * 32 + q * 8 + 4 * h;
      if (ok) st4(a.out + ((size_t)(c0 >> 3) * HWp + po) * 8 + (c0 & 7), v);
    }
  }
  if (odd) {                                                // outputs 32 nwg .. W * 64 - 1 of the odd row, 32 per workgroup and round
    for (int o0 = (a.nwg + (int)blockIdx.x) * 32; o0 < W * 64; o0 += a.nwg * 32) {      // (245 x 134: one round on the first 12 workgroups)
      wn_odd_load(a, tid, o0 + (tid >> 4), orow);
      wn_odd_finish<EPI>(a, tid, orow);
    }
  }
  if (DBG && lane == 0) {
    unsigned long long* r = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
    r[0] = t_start; r[1] = t_ld; r[2] = t_tr; r[3] = t_p1; r[4] = t_mm; r[5] = t_x; r[6] = __builtin_amdgcn_s_memtime();
    r[7] = __builtin_amdgcn_s_getreg(63492);
  }
}

static int conv_wino_init() {
  static int rc = -1;
  if (rc >= 0) return rc;
  rc = 0;
#define OPTIN(EPI_, DBG_) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<EPI_, DBG_>), hipFuncAttributeMaxDynamicSharedMemorySize, WN_SMEM); if (e != hipSuccess) rc = (int)e; }
  OPTIN(0, false) OPTIN(1, false) OPTIN(0, true)
#undef OPTIN
  return rc;
}

bool conv3x3_wino_supported(int H, int W, int cin, int cout) {
  return cin == 64 && cout == 64 && H >= 2 && W >= 1 && (long)H * W <= (1l << 24);
}

// out = epilogue(conv3x3(in, w)): epi 0 lrelu(conv + bias) ; epi 1 conv * lrelu'(aux) (backward-data: wU / wt are the packs of the
// flipped, transposed weights).  wU: pack_conv3x3_wino_f16 (its inverse host scale: winv); wt: the fp32 tap-major pack of the same weights.
int conv3x3_wino_f16(const float* in, const void* wU, float winv, const float* wt, const float* bias, const float* aux, float* out,
                     int H, int W, int epi, hipStream_t s, unsigned long long* dbg) {
  if (!conv3x3_wino_supported(H, W, 64, 64) || (epi != 0 && epi != 1)) return LEMO_ERR_SHAPE;
  if (!in || !wU || !wt || !out || !(winv > 0.f) || (epi == 0 ? !bias : !aux) || (dbg && epi != 0)) return LEMO_ERR_ARG;
  if (int rc = conv_wino_init()) return rc;
  WinoArgs a{};
  a.in = in; a.wU = reinterpret_cast<const uint4*>(wU); a.wt = wt; a.bias = bias; a.aux = aux; a.out = out; a.winv = winv;
  a.H = H; a.W = W; a.H2 = H & ~1; a.ntx = (W + 1) / 2;
  a.T = (a.H2 / 2) * a.ntx;
  a.nwg = (a.T + WN_TILES - 1) / WN_TILES;
  a.dbg = dbg;
  if (dbg) hipLaunchKernelGGL((conv3x3_wino_kernel<0, true>), dim3(a.nwg), dim3(512), WN_SMEM, s, a);
  else if (epi == 0) hipLaunchKernelGGL((conv3x3_wino_kernel<0, false>), dim3(a.nwg), dim3(512), WN_SMEM, s, a);
  else hipLaunchKernelGGL((conv3x3_wino_kernel<1, false>), dim3(a.nwg), dim3(512), WN_SMEM, s, a);
  return (int)hipGetLastError();
}

}  // namespace lemo
