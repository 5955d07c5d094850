// 3x3 / stride 1 / pad 1 convolutions of the motion-smoothness encoder (models/AE_sep.py:11-30,
// 77-99, `Enc(downsample=False)`: 10 x [conv3x3 + bias + LeakyReLU(0.2)], no pooling) and their
// backward-data passes, as fp32 implicit GEMMs on the CDNA4 matrix cores.
//
// Why fp32 MFMA: the smoothness loss is weighted by 1e6 (opt_amass_temp.py:49) and the parity
// budget is 1e-5 relative on the loss scalar -> exact-f32 v_mfma_f32_32x32x2_f32 (a k-ordered fmaf
// chain), no bf16.
//
// Data layout in HBM ("CG8P"): an activation with C channels over an H x W image is stored as
//     act[C/8][(H+2)*(W+2)][8]          (channel-group major, zero border of 1 pixel, 8 channels
//                                        of one pixel contiguous = 32 B)
// so that, for one (tap, channel-group), the 32 pixels of an MFMA tile x 32 B are one contiguous
// 1 KiB run: every operand load below is a fully coalesced global_load_dwordx4 and the 3x3 halo
// needs no bounds checks (the border is never written).  Weights are pre-packed on the host as
//     wt[tap][Cin/8][Cout][8]
// GEMM roles: M = Cout (A operand = weights), N = pixels (B operand = activations), K = 9*Cin.
// With the 32x32x2 lane map (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]) one dwordx4 per lane feeds four
// consecutive MFMA k-steps (lane half h consumes channels 4h..4h+3 of the 8-group), and the
// accumulator map (col=l&31 -> pixel, row=(r&3)+8*(r>>2)+4*(l>>5) -> cout) makes the epilogue a
// dwordx4 store per (lane, 8-cout group): 1 KiB contiguous per wave again.
#include "conv_common.hpp"
#include "loss_device.hpp"

namespace lemo {

// epilogues EPI 0/1/2: conv_common.hpp
template <int MT, int EPI>
__global__ void __launch_bounds__(256)
conv3x3_mfma_kernel(const float* __restrict__ in, const float* __restrict__ wt,
                    const float* __restrict__ bias, const float* __restrict__ aux,
                    float* __restrict__ out, int H, int W, int cin_g, int cout) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const int tile = blockIdx.x * 4 + wave;
  const int m_base = blockIdx.y * (MT * 32);                 // first cout of this block
  const int p = tile * 32 + j;
  if (tile * 32 >= P) return;                                // whole wave out of range (uniform)
  const int pc = p < P ? p : P - 1;
  const int y = pc / W, x = pc - y * W;
  const int poff = (y + 1) * Wp + (x + 1);

  const float* in_l = in + (size_t)poff * 8 + 4 * h;
  const float* wt_l = wt + ((size_t)(m_base + j)) * 8 + 4 * h;
  const size_t in_gstride = (size_t)HWp * 8;
  const size_t wt_itstride = (size_t)cout * 8;

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  // split-K: grid.z slices the 9 * Cin/8 (tap, channel group) steps; EPI 3 writes raw partial sums of slice z to
  // out + z * (Cout/8 * HWp * 8) and conv_splitk_combine_kernel finishes (layers with few pixels and many channels
  // -- the infilling AE at 27x17 .. 14x9 -- have too few tiles to fill the chip otherwise: one wave would run
  // 1152 fp32 MFMAs = 70 us)
  const int nit_all = 9 * cin_g;
  const int it_lo = (int)((long)nit_all * blockIdx.z / gridDim.z), nit = (int)((long)nit_all * (blockIdx.z + 1) / gridDim.z);
  if (EPI == 3) out += (size_t)blockIdx.z * ((size_t)(cout >> 3) * HWp * 8);
  int tap = it_lo / cin_g, g = it_lo - tap * cin_g;
  float4 a_cur[MT], b_cur;
  {
    const int dy0 = tap / 3 - 1, dx0 = tap - (tap / 3) * 3 - 1;
    b_cur = ld4(in_l + (std::ptrdiff_t)(dy0 * Wp + dx0) * 8 + (size_t)g * in_gstride);
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[m] = ld4(wt_l + (size_t)it_lo * wt_itstride + (size_t)m * 256);
  }
  for (int it = it_lo; it < nit; ++it) {
    float4 a_nxt[MT], b_nxt;
    int g2 = g + 1, tap2 = tap;
    if (g2 == cin_g) { g2 = 0; tap2 = tap + 1; }
    if (it + 1 < nit) {
      const int dy = tap2 / 3 - 1, dx = tap2 - (tap2 / 3) * 3 - 1;
      b_nxt = ld4(in_l + (std::ptrdiff_t)(dy * Wp + dx) * 8 + (size_t)g2 * in_gstride);
#pragma unroll
      for (int m = 0; m < MT; ++m) a_nxt[m] = ld4(wt_l + (size_t)(it + 1) * wt_itstride + (size_t)m * 256);
    } else {
      b_nxt = b_cur;
#pragma unroll
      for (int m = 0; m < MT; ++m) a_nxt[m] = a_cur[m];
    }
    const float bs[4] = {b_cur.x, b_cur.y, b_cur.z, b_cur.w};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float as = s == 0 ? a_cur[m].x : s == 1 ? a_cur[m].y : s == 2 ? a_cur[m].z : a_cur[m].w;
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, bs[s], acc[m], 0, 0, 0);
      }
    }
    b_cur = b_nxt;
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
    g = g2; tap = tap2;
  }

  if (p < P) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = m_base + m * 32 + q * 8 + 4 * h;          // first of 4 consecutive couts
        const size_t o = ((size_t)(c0 >> 3) * HWp + poff) * 8 + (c0 & 7);
        float4 v = make_float4(acc[m][4 * q + 0], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]);
        if (EPI == 3) {
        } else if (EPI == 0 || EPI == 2) {
          const float4 bb = ld4(bias + c0);
          v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
          if (EPI == 0) { v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w); }
        } else {
          const float4 yy = ld4(aux + o);
          v.x *= lrelu_grad_from_out(yy.x); v.y *= lrelu_grad_from_out(yy.y);
          v.z *= lrelu_grad_from_out(yy.z); v.w *= lrelu_grad_from_out(yy.w);
        }
        st4(out + o, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Variant 1 ("balanced"): the launch geometry is chosen for the 256 CUs x 4 SIMDs of the MI355X.
//   * main blocks: 128 pixels x CPB couts; wave w = (32-pixel tile w&3, 32-cout tile w>>2), so a
//     64-cout layer runs 8 waves per block = 2 waves per SIMD (they cover each other's load
//     latency and share the activation lines through L1).  245x134 = 32830 px = 256 full blocks
//     (one per CU) + 62 px.
//   * tail blocks (the < 128 remaining pixels): 16 px x 16 cout units on v_mfma_f32_16x16x4_f32,
//     one per wave, so the remainder costs ~1/8 of a main wave instead of doubling the makespan
//     (a 257th main block would: all 256 CUs are already busy).
//   * operands are prefetched two k-iterations ahead through a 3-deep register ring.
// ------------------------------------------------------------------------------------------------
// One tail unit: 16 pixels x 16 couts over the full K = 9*Cin on v_mfma_f32_16x16x4_f32, operands
// straight from global memory (tap-major pack).  The unit runs next to the main blocks, so it must
// not be latency-bound: all loads of THREE taps (6*GP dwordx4 per lane) are issued before their
// MFMAs (3 exposed round trips instead of one per k-step).
template <int EPI, int GP>          // GP = Cin/16 channel-group pairs (2 or 4)
__device__ __forceinline__ void conv_tail_unit(const float* __restrict__ in, const float* __restrict__ wt,
                                               const float* __restrict__ bias, const float* __restrict__ aux,
                                               float* __restrict__ out, int H, int W, int cout, int m_base, int p0) {
  const int lane = threadIdx.x & 63;
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const size_t in_gstride = (size_t)HWp * 8, wt_itstride = (size_t)cout * 8;
  const int j = lane & 15, q4 = lane >> 4;                       // pixel / cout row ; k-quarter
  const int p = p0 + j;
  const int pc = p < P ? p : P - 1;
  const int y = pc / W, x = pc - y * W;
  const int poff = (y + 1) * Wp + (x + 1);
  // 16 k per step = two 8-channel groups: quarter q4 reads group (2*gp + (q4>>1)), floats 4*(q4&1)..
  const float* in_l = in + (size_t)poff * 8 + (size_t)(q4 >> 1) * in_gstride + 4 * (q4 & 1);
  const float* wt_l = wt + ((size_t)(q4 >> 1) * cout + (m_base + j)) * 8 + 4 * (q4 & 1);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int t3 = 0; t3 < 3; ++t3) {
    float4 a[3][GP], b[3][GP];
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
      const int tap = t3 * 3 + tt, dy = t3 - 1, dx = tt - 1;
#pragma unroll
      for (int gp = 0; gp < GP; ++gp) {
        b[tt][gp] = ld4(in_l + (std::ptrdiff_t)(dy * Wp + dx) * 8 + (size_t)(2 * gp) * in_gstride);
        a[tt][gp] = ld4(wt_l + (size_t)(tap * 2 * GP + 2 * gp) * wt_itstride);
      }
    }
    // pin the schedule: every load above is issued before the first MFMA below (hipcc otherwise
    // sinks the loads next to their uses to save registers and waits vmcnt(0) after each one)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tt = 0; tt < 3; ++tt)
#pragma unroll
      for (int gp = 0; gp < GP; ++gp) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt][gp].x, b[tt][gp].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt][gp].y, b[tt][gp].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt][gp].z, b[tt][gp].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt][gp].w, b[tt][gp].w, acc, 0, 0, 0);
      }
  }
  if (p < P) {
    const int c0 = m_base + 4 * q4;                              // D: col = pixel j, rows 4*q4 + r
    const size_t o = ((size_t)(c0 >> 3) * HWp + poff) * 8 + (c0 & 7);
    conv_store4<EPI>(out, bias, aux, o, c0, make_float4(acc[0], acc[1], acc[2], acc[3]));
  }
}

template <int EPI>
__global__ void __launch_bounds__(512)
conv3x3_mfma_v1_kernel(const float* __restrict__ in, const float* __restrict__ wt,
                       const float* __restrict__ bias, const float* __restrict__ aux,
                       float* __restrict__ out, int H, int W, int cin_g, int cout, int cpb, int full_blocks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const size_t in_gstride = (size_t)HWp * 8;
  const size_t wt_itstride = (size_t)cout * 8;
  const int nit = 9 * cin_g;
  const int cb = blockIdx.y * cpb;                              // first cout of this block column
  if ((int)blockIdx.x < full_blocks) {
    // ---------------- main path: 32 px x 32 cout per wave, 32x32x2 MFMA ----------------
    const int j = lane & 31, h = lane >> 5;
    const int p = (blockIdx.x * 4 + (wave & 3)) * 32 + j;       // always < P
    const int m_base = cb + (wave >> 2) * 32;
    const int y = p / W, x = p - y * W;
    const int poff = (y + 1) * Wp + (x + 1);
    const float* in_l = in + (size_t)poff * 8 + 4 * h;
    const float* wt_l = wt + ((size_t)(m_base + j)) * 8 + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#define LEMO_LOAD(IT, A_, B_)                                                                     \
    {                                                                                             \
      const int it_ = (IT) < nit ? (IT) : nit - 1;                                                \
      const int tap_ = it_ / cin_g, g_ = it_ - tap_ * cin_g;                                      \
      const int dy_ = (tap_ * 11 >> 5) - 1, dx_ = tap_ - (dy_ + 1) * 3 - 1;                       \
      B_ = ld4(in_l + (std::ptrdiff_t)(dy_ * Wp + dx_) * 8 + (size_t)g_ * in_gstride);            \
      A_ = ld4(wt_l + (size_t)it_ * wt_itstride);                                                 \
    }
#define LEMO_MFMA4(A_, B_)                                                                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.x, B_.x, acc, 0, 0, 0);                         \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.y, B_.y, acc, 0, 0, 0);                         \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.z, B_.z, acc, 0, 0, 0);                         \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.w, B_.w, acc, 0, 0, 0);
    float4 a0, b0, a1, b1, a2, b2;
    LEMO_LOAD(0, a0, b0)
    LEMO_LOAD(1, a1, b1)
    for (int it = 0; it < nit; it += 3) {                       // nit = 9*cin_g is a multiple of 3
      LEMO_LOAD(it + 2, a2, b2)
      LEMO_MFMA4(a0, b0)
      LEMO_LOAD(it + 3, a0, b0)
      LEMO_MFMA4(a1, b1)
      LEMO_LOAD(it + 4, a1, b1)
      LEMO_MFMA4(a2, b2)
    }
#undef LEMO_LOAD
#undef LEMO_MFMA4
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c0 = m_base + q * 8 + 4 * h;
      const size_t o = ((size_t)(c0 >> 3) * HWp + poff) * 8 + (c0 & 7);
      conv_store4<EPI>(out, bias, aux, o, c0, make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
    }
  } else {
    // ---------------- tail path: 16 px x 16 cout per wave, 16x16x4 MFMA ----------------
    const int nwaves = blockDim.x >> 6;
    const int unit = ((int)blockIdx.x - full_blocks) * nwaves + wave;
    const int ct = cpb >> 4;                                     // cout tiles per px tile
    const int pt = unit / ct, mt = unit - pt * ct;
    const int p0 = full_blocks * 128 + pt * 16;
    if (p0 >= P) return;                                         // uniform per wave
    const int j = lane & 15, q4 = lane >> 4;                     // pixel / cout row ; k-quarter
    const int p = p0 + j;
    const int pc = p < P ? p : P - 1;
    const int y = pc / W, x = pc - y * W;
    const int poff = (y + 1) * Wp + (x + 1);
    const int m_base = cb + mt * 16;
    // 16 k per step = two 8-channel groups: quarter q4 reads group (2*gp + (q4>>1)), floats 4*(q4&1)..
    const float* in_l = in + (size_t)poff * 8 + (size_t)(q4 >> 1) * in_gstride + 4 * (q4 & 1);
    const float* wt_l = wt + ((size_t)(q4 >> 1) * cout + (m_base + j)) * 8 + 4 * (q4 & 1);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int gpairs = cin_g >> 1;
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      for (int gp = 0; gp < gpairs; ++gp) {
        const float4 b = ld4(in_l + (std::ptrdiff_t)(dy * Wp + dx) * 8 + (size_t)(2 * gp) * in_gstride);
        const float4 a = ld4(wt_l + (size_t)(tap * cin_g + 2 * gp) * wt_itstride);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
      }
    }
    if (p < P) {
      const int c0 = m_base + 4 * q4;                            // D: col = pixel j, rows 4*q4 + r
      const size_t o = ((size_t)(c0 >> 3) * HWp + poff) * 8 + (c0 & 7);
      conv_store4<EPI>(out, bias, aux, o, c0, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Variant 2 ("LDS-tiled"): same CU-balanced geometry as variant 1, but the operands of the main
// blocks go through LDS.  Per 8-channel group the block stages ONCE
//     Bs: the 128-pixel tile plus its 3x3 halo (<= 404 padded pixels x 32 B = 12.9 KB)
//     As: the 9 taps x CPB couts of that group (18 KB for 64 couts; weights packed g-major: wt2)
// and all 9 taps x 8 waves read them with conflict-free ds_read_b128 (planes split by lane half h:
// Xs[h][idx][4]) instead of 16x redundant L1/L2 requests per k-step.  Staging is register-staged
// and double-buffered: global loads for group g+1 are issued before the 36 MFMAs of group g and
// written to the other LDS buffer after them (one barrier per group).
// ------------------------------------------------------------------------------------------------
#define CV2_NPX 416
template <int NT, int GPS> struct Cv2Cfg {
  static constexpr int CPB = NT / 8;                            // couts per block (64 or 32)
  static constexpr int NCH_A = 18 * CPB;                        // 16-B chunks of one group's weights
  static constexpr int B_GRP = 2 * CV2_NPX * 4;                 // floats per group: [2 planes][NPX][4]
  static constexpr int A_GRP = 9 * 2 * CPB * 4;                 // floats per group: [9][2 planes][CPB][4]
  static constexpr int B_BUF = GPS * B_GRP, A_BUF = GPS * A_GRP;
  static constexpr int A_OFF = 2 * B_BUF;
  static constexpr int SMEM_BYTES = (2 * B_BUF + 2 * A_BUF) * 4;
};

template <int EPI, int NT, int GPS, bool DBG = false>   // NT threads: 512 (cpb 64) / 256 (cpb 32); GPS groups per stage
__global__ void __launch_bounds__(NT)
conv3x3_mfma_v2_kernel(const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ wt2,
                       const float* __restrict__ bias, const float* __restrict__ aux,
                       float* __restrict__ out, int H, int W, int cin_g, int cout, int full_blocks, int tail_blocks,
                       unsigned long long* __restrict__ dbg) {
  // Tail blocks take the LOWEST block ids: they are dispatched first, so their waves are the oldest
  // on the SIMDs they share with a main block and win the (age-ordered) issue arbitration instead of
  // starving behind it (measured: dispatched last they ran 1.35x longer than a main wave).
  unsigned long long t_start = 0;
  if (DBG) t_start = __builtin_amdgcn_s_memtime();
  typedef Cv2Cfg<NT, GPS> Cfg;
  constexpr int CPB = Cfg::CPB, NCH_A = Cfg::NCH_A, B_GRP = Cfg::B_GRP, A_GRP = Cfg::A_GRP;
  constexpr int B_BUF = Cfg::B_BUF, A_BUF = Cfg::A_BUF, A_OFF = Cfg::A_OFF;
  // ONE dynamic LDS object (a second one makes hipcc drain vmcnt before every ds_read), 16-B aligned:
  // [2 buffers][GPS groups] activations, then [2 buffers][GPS groups] weights (127 KB at GPS = 2)
  LEMO_DYN_SMEM(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const size_t in_gstride = (size_t)HWp * 8;
  const int cb = blockIdx.y * CPB;
  if ((int)blockIdx.x >= tail_blocks) {
    const int j = lane & 31, h = lane >> 5;
    // XCD-aware tile order (guide T1): workgroup b runs on XCD b % 8, so give XCD x the contiguous
    // run of tiles [x*q, x*q+q): vertically adjacent tiles then share their halo rows in ONE L2
    // instead of each XCD fetching them from the fabric (PMC: 26.8 MB fetched for an 8.4 MB input).
    int tile = (int)blockIdx.x - tail_blocks;
    {
      const int q = full_blocks >> 3, r = full_blocks & 7, xcd = tile & 7, k = tile >> 3;
      tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;   // bijective for any count
    }
    const int pfirst = tile * 128, plast = pfirst + 127;
    const int yf = pfirst / W, yl = plast / W;
    const int q0 = (yf + 1) * Wp + (pfirst - yf * W + 1), q1 = (yl + 1) * Wp + (plast - yl * W + 1);
    const int qin = q0 - Wp - 1;                                 // first staged padded pixel
    const int npx = q1 - q0 + 2 * Wp + 3;                        // staged pixels (<= CV2_NPX, checked on host)
    const int p = pfirst + (wave & 3) * 32 + j;
    const int y = p / W, x = p - y * W;
    const int poff = (y + 1) * Wp + (x + 1);
    const int li = poff - qin;                                   // local index of the centre tap
    const int mt = wave >> 2;                                    // cout tile of this wave
    const int nchB = 2 * npx;
    const unsigned wt_gstride = 9u * cout * 8u;
    // Staging plan of this thread: NB activation chunks (16 B: pixel c>>1, half c&1) and NA weight
    // chunks per STAGE (= GPS channel groups).  Every slot issues an UNCONDITIONAL global load
    // (out-of-range slots re-read the last chunk) so that all loads of a stage are in flight together
    // and stay global_load (a pointer select would degrade them to flat_load, which also bumps
    // lgkmcnt and would stall the ds_read waits); only the LDS write is predicated.
    constexpr int NB = (GPS * 2 * CV2_NPX + NT - 1) / NT, NA = (GPS * NCH_A + NT - 1) / NT;
    unsigned offB[NB], offA[NA];
    int dstB[NB], dstA[NA];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      int c = threadIdx.x + k * NT;
      const bool ok = c < GPS * nchB;
      if (!ok) c = GPS * nchB - 1;
      const int gg = c / nchB;
      c -= gg * nchB;
      offB[k] = (unsigned)gg * (unsigned)in_gstride + (unsigned)qin * 8u + (unsigned)c * 4u;
      dstB[k] = ok ? gg * B_GRP + ((c & 1) * CV2_NPX + (c >> 1)) * 4 : -1;
    }
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      int ca = threadIdx.x + k * NT;
      const bool ok = ca < GPS * NCH_A;
      if (!ok) ca = GPS * NCH_A - 1;
      const int gg = ca / NCH_A;
      ca -= gg * NCH_A;
      const int tap = ca / (2 * CPB), r = ca - tap * 2 * CPB;
      offA[k] = (unsigned)gg * wt_gstride + ((unsigned)tap * cout + cb) * 8u + (unsigned)r * 4u;
      dstA[k] = ok ? A_OFF + gg * A_GRP + ((tap * 2 + (r & 1)) * CPB + (r >> 1)) * 4 : -1;
    }
    float4 stB[NB], stA[NA];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int k = 0; k < NB; ++k) stB[k] = ld4(in + offB[k]);
#pragma unroll
    for (int k = 0; k < NA; ++k) stA[k] = ld4(wt2 + offA[k]);
#pragma unroll
    for (int k = 0; k < NB; ++k) if (dstB[k] >= 0) st4(&smem[dstB[k]], stB[k]);
#pragma unroll
    for (int k = 0; k < NA; ++k) if (dstA[k] >= 0) st4(&smem[dstA[k]], stA[k]);
    __syncthreads();
    unsigned long long t_pro = 0, t_loop = 0;
    if (DBG) t_pro = __builtin_amdgcn_s_memtime();
    const float* a_rd = &smem[A_OFF + (h * CPB + mt * 32 + j) * 4];
    const float* b_rd = &smem[(h * CV2_NPX + li) * 4];
    const int nstage = cin_g / GPS;
    for (int sg = 0; sg < nstage; ++sg) {
      const int buf = sg & 1;
      if (sg + 1 < nstage) {
        const float* ing = in + (size_t)(sg + 1) * GPS * in_gstride;
        const float* wtg = wt2 + (size_t)(sg + 1) * GPS * wt_gstride;
#pragma unroll
        for (int k = 0; k < NB; ++k) stB[k] = ld4(ing + offB[k]);
#pragma unroll
        for (int k = 0; k < NA; ++k) stA[k] = ld4(wtg + offA[k]);
      }
      const float* ar = a_rd + buf * A_BUF;
      const float* br = b_rd + buf * B_BUF;
      // Operands are read from LDS one 3-tap chunk (one kernel row) AHEAD of the MFMAs that consume
      // them, through two register sets; the sched_barriers pin "reads of chunk c+1, then the 12 MFMAs
      // of chunk c".  Without this hipcc issues each tap's two ds_reads right before its 4 MFMAs and
      // both waves of a SIMD stall on the same LDS round trip after every tap (measured 24 % idle).
      float4 ra[2][3], rb[2][3];
#define CV2_READ3(SET, CH)                                                                        \
      _Pragma("unroll") for (int t_ = 0; t_ < 3; ++t_) {                                          \
        constexpr int gg_ = (CH) / 3, dy_ = (CH) % 3 - 1;                                         \
        ra[SET][t_] = ld4(ar + gg_ * A_GRP + (((CH) % 3) * 3 + t_) * (2 * CPB * 4));             \
        rb[SET][t_] = ld4(br + gg_ * B_GRP + (dy_ * Wp + (t_ - 1)) * 4);                          \
      }
#define CV2_MFMA3(SET)                                                                            \
      _Pragma("unroll") for (int t_ = 0; t_ < 3; ++t_) {                                          \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[SET][t_].x, rb[SET][t_].x, acc, 0, 0, 0);   \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[SET][t_].y, rb[SET][t_].y, acc, 0, 0, 0);   \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[SET][t_].z, rb[SET][t_].z, acc, 0, 0, 0);   \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[SET][t_].w, rb[SET][t_].w, acc, 0, 0, 0);   \
      }
      static_assert(GPS == 2, "chunk schedule below is written for 2 groups per stage");
      CV2_READ3(0, 0)
      __builtin_amdgcn_sched_barrier(0);
      CV2_READ3(1, 1)
      __builtin_amdgcn_sched_barrier(0);
      CV2_MFMA3(0)
      __builtin_amdgcn_sched_barrier(0);
      CV2_READ3(0, 2)
      __builtin_amdgcn_sched_barrier(0);
      CV2_MFMA3(1)
      __builtin_amdgcn_sched_barrier(0);
      CV2_READ3(1, 3)
      __builtin_amdgcn_sched_barrier(0);
      CV2_MFMA3(0)
      __builtin_amdgcn_sched_barrier(0);
      CV2_READ3(0, 4)
      __builtin_amdgcn_sched_barrier(0);
      CV2_MFMA3(1)
      __builtin_amdgcn_sched_barrier(0);
      CV2_READ3(1, 5)
      __builtin_amdgcn_sched_barrier(0);
      CV2_MFMA3(0)
      __builtin_amdgcn_sched_barrier(0);
      CV2_MFMA3(1)
#undef CV2_READ3
#undef CV2_MFMA3
      if (sg + 1 < nstage) {
#pragma unroll
        for (int k = 0; k < NB; ++k) if (dstB[k] >= 0) st4(&smem[dstB[k] + (buf ^ 1) * B_BUF], stB[k]);
#pragma unroll
        for (int k = 0; k < NA; ++k) if (dstA[k] >= 0) st4(&smem[dstA[k] + (buf ^ 1) * A_BUF], stA[k]);
      }
      __syncthreads();
    }
    if (DBG) t_loop = __builtin_amdgcn_s_memtime();
    const int m_base = cb + mt * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c0 = m_base + q * 8 + 4 * h;
      const size_t o = ((size_t)(c0 >> 3) * HWp + poff) * 8 + (c0 & 7);
      conv_store4<EPI>(out, bias, aux, o, c0, make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
    }
    if (DBG && lane == 0) {           // census: where and when did this wave run
      unsigned long long* r = dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (NT / 64) + wave) * 8;
      r[0] = __builtin_amdgcn_s_getreg(63492);        // HW_REG_HW_ID
      r[1] = __builtin_amdgcn_s_getreg(63508);        // HW_REG_XCC_ID
      r[2] = t_start; r[3] = __builtin_amdgcn_s_memtime(); r[4] = t_pro; r[5] = t_loop;
    }
  } else {
    // tail: 16 px x 16 cout units straight from global (tap-major pack `wt`)
    // ONE unit per tail block (wave 0 only; the other waves retire at once): 16 units land on 16
    // different CUs instead of loading 2 CUs with 8 extra waves each (measured: +40 % on those CUs)
    const int unit = (int)blockIdx.x;
    const int ct = CPB >> 4;
    const int pt = unit / ct, mtl = unit - pt * ct;
    const int p0 = full_blocks * 128 + pt * 16;
    if (wave == 0 && p0 < P) {                                   // uniform per wave
      if (cin_g == 8) conv_tail_unit<EPI, 4>(in, wt, bias, aux, out, H, W, cout, cb + mtl * 16, p0);
      else if (cin_g == 4) conv_tail_unit<EPI, 2>(in, wt, bias, aux, out, H, W, cout, cb + mtl * 16, p0);
    }
    if (DBG && lane == 0) {
      unsigned long long* r = dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (NT / 64) + wave) * 8;
      r[0] = __builtin_amdgcn_s_getreg(63492);
      r[1] = __builtin_amdgcn_s_getreg(63508);
      r[2] = t_start; r[3] = __builtin_amdgcn_s_memtime(); r[4] = 0; r[5] = 0;
    }
  }
}

// > 64 KB of dynamic LDS needs an explicit opt-in per kernel (once per process; never inside a capture
// because lemo_fit_create / the first eager call runs it first)
int conv_lds_init() {
  static int rc = -1;
  if (rc >= 0) return rc;
  rc = 0;
#define OPTIN(EPI_, NT_, DBG_) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_v2_kernel<EPI_, NT_, 2, DBG_>), hipFuncAttributeMaxDynamicSharedMemorySize, Cv2Cfg<NT_, 2>::SMEM_BYTES); if (e != hipSuccess) rc = (int)e; }
  OPTIN(0, 512, false) OPTIN(1, 512, false) OPTIN(2, 512, false)
  OPTIN(0, 256, false) OPTIN(1, 256, false) OPTIN(2, 256, false)
  OPTIN(0, 512, true)
#undef OPTIN
  return rc;
}

int conv3x3_mfma_lds(const float* in, const float* wt, const float* wt2, const float* bias, const float* aux, float* out,
                     int H, int W, int cin, int cout, int epi, hipStream_t s, unsigned long long* dbg) {
  if ((cin != 32 && cin != 64) || cout % 32 || H <= 0 || W <= 0 || epi < 0 || epi > 2) return LEMO_ERR_SHAPE;   // tail units are instantiated for Cin 32 / 64
  // staged pixels of a 128-pixel run: 127 + 2 per row end crossed + two halo rows + 3
  if (127 + 2 * (127 / W + 1) + 2 * (W + 2) + 3 > CV2_NPX) return LEMO_ERR_SHAPE;
  const int P = H * W;
  const int cpb = (cout % 64 == 0) ? 64 : 32;
  const int full = P / 128, rem = P - full * 128;
  const int units = ((rem + 15) / 16) * (cpb / 16);
  const int tailb = units;                                       // one 16x16 unit per tail block
  dim3 grid(full + tailb, cout / cpb);
  conv_lds_init();
  if (dbg) {                                 // census build of the forward 64-cout kernel (tools/conv_census.py)
    if (cpb != 64 || epi != 0) return LEMO_ERR_ARG;
    hipLaunchKernelGGL((conv3x3_mfma_v2_kernel<0, 512, 2, true>), grid, dim3(512), (Cv2Cfg<512, 2>::SMEM_BYTES), s, in, wt, wt2, bias, aux, out, H, W, cin / 8, cout, full, tailb, dbg);
    return (int)hipGetLastError();
  }
#define LAUNCH2(EPI_, NT_) hipLaunchKernelGGL((conv3x3_mfma_v2_kernel<EPI_, NT_, 2, false>), grid, dim3(NT_), (Cv2Cfg<NT_, 2>::SMEM_BYTES), s, in, wt, wt2, bias, aux, out, H, W, cin / 8, cout, full, tailb, (unsigned long long*)nullptr)
  if (cpb == 64) { if (epi == 0) LAUNCH2(0, 512); else if (epi == 1) LAUNCH2(1, 512); else LAUNCH2(2, 512); }
  else           { if (epi == 0) LAUNCH2(0, 256); else if (epi == 1) LAUNCH2(1, 256); else LAUNCH2(2, 256); }
#undef LAUNCH2
  return (int)hipGetLastError();
}

// out = epilogue(sum_z partial[z]) over the interior pixels (slices summed in a fixed order)
template <int EPI>
__global__ void __launch_bounds__(256)
conv_splitk_combine_kernel(const float* __restrict__ partial, int ks, const float* __restrict__ bias, const float* __restrict__ aux,
                           float* __restrict__ out, int H, int W, int cout) {
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * (cout >> 3) * 2) return;
  const int half = idx & 1, rest = idx >> 1;
  const int g = rest / P, p = rest - g * P;
  const int y = p / W, x = p - y * W;
  const size_t o = ((size_t)g * HWp + (y + 1) * Wp + (x + 1)) * 8 + 4 * half;
  const size_t slice = (size_t)(cout >> 3) * HWp * 8;
  float4 v = ld4(partial + o);
  for (int z = 1; z < ks; ++z) {
    const float4 t = ld4(partial + z * slice + o);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  conv_store4<EPI>(out, bias, aux, o, g * 8 + 4 * half, v);
}

int conv3x3_mfma_splitk(const float* in, const float* wt, const float* bias, const float* aux, float* out, float* partial, int ks,
                        int H, int W, int cin, int cout, int epi, hipStream_t s) {
  if (cin % 8 || cout % 32 || H <= 0 || W <= 0 || epi < 0 || epi > 2 || ks < 1 || ks > 9 * (cin / 8) || !partial) return LEMO_ERR_SHAPE;
  const int P = H * W;
  const int mt = (cout % 64 == 0) ? 2 : 1;
  dim3 grid((P + 127) / 128, cout / (mt * 32), ks);
  if (mt == 2) hipLaunchKernelGGL((conv3x3_mfma_kernel<2, 3>), grid, dim3(256), 0, s, in, wt, bias, aux, partial, H, W, cin / 8, cout);
  else hipLaunchKernelGGL((conv3x3_mfma_kernel<1, 3>), grid, dim3(256), 0, s, in, wt, bias, aux, partial, H, W, cin / 8, cout);
  int e = (int)hipGetLastError();
  if (e) return e;
  const int nthr = P * (cout / 8) * 2;
#define COMBINE(EPI_) hipLaunchKernelGGL((conv_splitk_combine_kernel<EPI_>), dim3((nthr + 255) / 256), dim3(256), 0, s, partial, ks, bias, aux, out, H, W, cout)
  if (epi == 0) COMBINE(0); else if (epi == 1) COMBINE(1); else COMBINE(2);
#undef COMBINE
  return (int)hipGetLastError();
}

int conv3x3_mfma(const float* in, const float* wt, const float* bias, const float* aux, float* out,
                 int H, int W, int cin, int cout, int epi, int variant, hipStream_t s) {
  if (cin % 8 || cout % 32 || H <= 0 || W <= 0 || epi < 0 || epi > 2) return LEMO_ERR_SHAPE;
  const int P = H * W;
  if (variant == 0) {
    const int mt = (cout % 64 == 0) ? 2 : 1;
    dim3 grid((P + 127) / 128, cout / (mt * 32));
#define LAUNCH(MT_, EPI_) hipLaunchKernelGGL((conv3x3_mfma_kernel<MT_, EPI_>), grid, dim3(256), 0, s, in, wt, bias, aux, out, H, W, cin / 8, cout)
    if (mt == 2) { if (epi == 0) LAUNCH(2, 0); else if (epi == 1) LAUNCH(2, 1); else LAUNCH(2, 2); }
    else         { if (epi == 0) LAUNCH(1, 0); else if (epi == 1) LAUNCH(1, 1); else LAUNCH(1, 2); }
#undef LAUNCH
    return (int)hipGetLastError();
  }
  if (variant != 1 || (cin % 16)) return LEMO_ERR_ARG;
  const int cpb = (cout % 64 == 0) ? 64 : 32;                    // couts per block: 8 or 4 waves
  const int nw = 4 * (cpb / 32);
  const int full = P / 128, rem = P - full * 128;
  const int units = ((rem + 15) / 16) * (cpb / 16);
  dim3 grid(full + (units + nw - 1) / nw, cout / cpb);
#define LAUNCH1(EPI_) hipLaunchKernelGGL((conv3x3_mfma_v1_kernel<EPI_>), grid, dim3(64 * nw), 0, s, in, wt, bias, aux, out, H, W, cin / 8, cout, cpb, full)
  if (epi == 0) LAUNCH1(0); else if (epi == 1) LAUNCH1(1); else LAUNCH1(2);
#undef LAUNCH1
  return (int)hipGetLastError();
}

// ---- first layer: 1 -> Cout channels (K = 9: VALU) --------------------------------------------
// x0: plain padded single-channel image [(H+2)*(W+2)] (zero border); w: [Cout][9]; out CG8P.
// One thread per (pixel, 8-cout group).
__global__ void __launch_bounds__(256)
conv3x3_c1_kernel(const float* __restrict__ x0, const float* __restrict__ w, const float* __restrict__ bias,
                  float* __restrict__ out, int H, int W, int cout) {
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int ng = cout >> 3;
  if (idx >= P * ng) return;
  const int g = idx / P, p = idx - g * P;
  const int y = p / W, x = p - y * W;
  const int poff = (y + 1) * Wp + (x + 1);
  float xin[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) xin[t] = x0[poff + (t / 3 - 1) * Wp + (t % 3 - 1)];
  float r[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float* wc = w + (size_t)(g * 8 + c) * 9;
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) a = fmaf(wc[t], xin[t], a);
    r[c] = lrelu(a + bias[g * 8 + c]);
  }
  float* o = out + ((size_t)g * HWp + poff) * 8;
  st4(o, make_float4(r[0], r[1], r[2], r[3]));
  st4(o + 4, make_float4(r[4], r[5], r[6], r[7]));
}

int conv3x3_c1(const float* x0, const float* w, const float* bias, float* out, int H, int W, int cout,
               hipStream_t s) {
  if (cout % 8) return LEMO_ERR_SHAPE;
  const int n = H * W * (cout / 8);
  hipLaunchKernelGGL(conv3x3_c1_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x0, w, bias, out, H, W, cout);
  return (int)hipGetLastError();
}

// ---- backward-data of the first layer: Cin(=32) channels of d(pre-activation) -> 1 channel ----
// dx0[y][x] = sum_co sum_tap dpre[co][y-dy][x-dx] * w[co][tap(dy,dx)]   (dx0: unpadded [H*W])
__global__ void __launch_bounds__(256)
conv3x3_c1_bwd_kernel(const float* __restrict__ dpre, const float* __restrict__ w, float* __restrict__ dx0,
                      int H, int W, int cout) {
  const int Wp = W + 2, HWp = (H + 2) * Wp, P = H * W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int y = p / W, x = p - y * W;
  const int poff = (y + 1) * Wp + (x + 1);
  float a = 0.f;
  for (int g = 0; g < (cout >> 3); ++g) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const float* q = dpre + ((size_t)g * HWp + poff - dy * Wp - dx) * 8;
      const float4 v0 = ld4(q), v1 = ld4(q + 4);
      const float* wc = w + (size_t)(g * 8) * 9 + t;
      a = fmaf(v0.x, wc[0], a); a = fmaf(v0.y, wc[9], a); a = fmaf(v0.z, wc[18], a); a = fmaf(v0.w, wc[27], a);
      a = fmaf(v1.x, wc[36], a); a = fmaf(v1.y, wc[45], a); a = fmaf(v1.z, wc[54], a); a = fmaf(v1.w, wc[63], a);
    }
  }
  dx0[p] = a;
}

int conv3x3_c1_bwd(const float* dpre, const float* w, float* dx0, int H, int W, int cout, hipStream_t s) {
  if (cout % 8) return LEMO_ERR_SHAPE;
  hipLaunchKernelGGL(conv3x3_c1_bwd_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, dpre, w, dx0, H, W, cout);
  return (int)hipGetLastError();
}

// (body: loss_device.hpp)
__global__ void __launch_bounds__(256)
smooth_loss_kernel(const float* __restrict__ z, float* __restrict__ dpre, float* __restrict__ partial,
                   int H, int W, int C, float coef2, double* __restrict__ acc) {
  smooth_loss_body((int)blockIdx.x, z, dpre, partial, H, W, C, coef2, acc);
}

int smooth_loss_blocks(int H, int W, int C) { return (H * W * (C / 8) * 2 + 256 * LEMO_SMOOTH_ITEMS - 1) / (256 * LEMO_SMOOTH_ITEMS); }

int smooth_loss(const float* z, float* dpre, float* partial, int H, int W, int C, float coef2, hipStream_t s, double* acc) {
  if (C % 8 || (!partial && !acc)) return LEMO_ERR_SHAPE;
  hipLaunchKernelGGL(smooth_loss_kernel, dim3(smooth_loss_blocks(H, W, C)), dim3(256), 0, s, z, dpre, partial, H, W, C, coef2, acc);
  return (int)hipGetLastError();
}

}  // namespace lemo
