// The smoothness encoder's MFMA layers (models/AE_sep.py:77-99, layers 1..9 of the 10; layer 0 is fused with the marker image) as
// launch sequences shared by the two engines (lemo_fit_desc / lemo_prox_desc carry the same enc_* fields).
#pragma once
#include "kernels.hpp"

namespace lemo {

#define ENC_CHK(e) do { int _e = (e); if (_e) return _e; } while (0)

// conv variant 7 = variant 5 + the fused head / tail (conv_head_kernels.hip): layers 0 and 1 ride with the marker image / the image gradient
template <class D> static inline bool enc_fused_head(const D& d) {
  return d.conv_variant >= 7 && d.conv_variant <= 10 && d.enc_ch[1] == 32 && d.enc_ch[2] == 32 && d.enc_w3[1] && d.enc_wbwd3[1];
}
// conv variant 8 (and 9) = variant 7 with layer 2 (32 -> 64) inside the head launch as well (enc_head3)
template <class D> static inline bool enc_fused_head3(const D& d) {
  return d.conv_variant >= 8 && enc_fused_head(d) && d.enc_ch[3] == 64 && d.enc_w3[2];
}
// conv variant 9 = variant 8 with layer 2's backward-data (64 -> 32) inside the tail launch (enc_tail3)
template <class D> static inline bool enc_fused_tail3(const D& d) {
  return d.conv_variant >= 9 && enc_fused_head(d) && d.enc_ch[3] == 64 && d.enc_wbwd3[2];
}
// the encoder's backward tail: d(pre-act l_last) in dact[cur] -> d(loss)/d(image); l_last as enc_chain_bwd was told
template <class D> static inline int enc_bwd_l_last(const D& d) { return enc_fused_tail3(d) ? 3 : (enc_fused_head(d) ? 2 : 1); }
template <class D> static inline int enc_bwd_tail(const D& d, int cur, int H, int W, hipStream_t s) {
  if (enc_fused_tail3(d))
    return enc_tail3(d.dact[cur], d.enc_wbwd3[2], d.enc_wbwd3_inv[2], d.act[2], d.enc_wbwd3[1], d.enc_wbwd3_inv[1], d.act[1], d.enc_w[0], d.dx0, H, W, s);
  if (enc_fused_head(d)) return enc_tail(d.dact[cur], d.enc_wbwd3[1], d.enc_wbwd3_inv[1], d.act[1], d.enc_w[0], d.dx0, H, W, s);
  return conv3x3_c1_bwd(d.dact[cur], d.enc_w[0], d.dx0, H, W, d.enc_ch[1], s);
}

// one layer (forward: act[l] -> act[l+1]; backward-data: d(pre-act l+1) -> d(pre-act l) with the saved activation act[l] as
// epilogue operand) on the kernel family conv_variant selects for its shape
template <class D>
static inline int enc_layer(const D& d, int l, bool bwd, const float* src, float* dst, int H, int W, hipStream_t s) {
  const int cin = bwd ? d.enc_ch[l + 1] : d.enc_ch[l], cout = bwd ? d.enc_ch[l] : d.enc_ch[l + 1];
  const float* wt = bwd ? d.enc_wbwd[l] : d.enc_w[l];
  const float* wt2 = bwd ? d.enc_wbwd2[l] : d.enc_w2[l];
  const void* w3 = bwd ? d.enc_wbwd3[l] : d.enc_w3[l];
  const float* bias = bwd ? nullptr : d.enc_b[l];
  const float* aux = bwd ? d.act[l] : nullptr;
  const int epi = bwd ? 1 : 0;
  // conv variant 10 = variant 9 with every 64 -> 64 layer as ONE Winograd F(2x2, 3x3) launch (conv_wino_kernels.hip) instead of the
  // fused pairs: enc_w3 / enc_wbwd3 of those layers hold the Winograd packs (pack_conv3x3_wino_f16), the 32-channel layers' the split-f16 packs
  if (d.conv_variant == 10 && w3 && conv3x3_wino_supported(H, W, cin, cout))
    return conv3x3_wino_f16(src, w3, bwd ? d.enc_wbwd3_inv[l] : d.enc_w3_inv[l], wt, bias, aux, dst, H, W, epi, s);
  if (d.conv_variant >= 3 && w3 && conv3x3_split_supported(H, W, cin, cout))
    return conv3x3_mfma_split(src, w3, wt, bias, aux, dst, H, W, cin, cout, epi, s, nullptr, d.conv_variant >= 4 ? 2 : 3,
                              bwd ? d.enc_wbwd3_inv[l] : d.enc_w3_inv[l]);
  if (d.conv_variant >= 2 && 127 + 2 * (127 / W + 1) + 2 * (W + 2) + 3 <= 416)
    return conv3x3_mfma_lds(src, wt, wt2, bias, aux, dst, H, W, cin, cout, epi, s);
  return conv3x3_mfma(src, wt, bias, aux, dst, H, W, cin, cout, epi, d.conv_variant >= 2 ? 1 : d.conv_variant, s);
}

// layers 1..9 forward: act[1] -> act[10].  conv_variant >= 5: consecutive layers whose three channel counts the pair kernel takes
// run as ONE launch (64 -> 64 -> 64: layers (3,4), (5,6), (7,8); 6 launches instead of 9), the intermediate activation is still
// written (the backward pass reads every act[l])
template <class D>
static inline int enc_chain_fwd(const D& d, int H, int W, hipStream_t s, int l_first = 1) {
  int l = l_first;                 // (2 when enc_head produced act[2] already: conv variant 7)
  while (l < 10) {
    if (d.conv_variant >= 5 && d.conv_variant != 10 && l + 1 < 10 && d.enc_w3[l] && d.enc_w3[l + 1] &&
        conv3x3_pair_supported(H, W, d.enc_ch[l], d.enc_ch[l + 1], d.enc_ch[l + 2])) {
      ENC_CHK(conv3x3_pair_f16(d.act[l], d.enc_w3[l], d.enc_w3_inv[l], d.enc_b[l], nullptr, d.act[l + 1],
                                                                           d.enc_w3[l + 1], d.enc_w3_inv[l + 1], d.enc_b[l + 1], nullptr, d.act[l + 2],
                                                                           H, W, 0, s, nullptr));
      l += 2;
    } else {
      ENC_CHK(enc_layer(d, l, false, d.act[l], d.act[l + 1], H, W, s));
      ++l;
    }
  }
  return 0;
}

// layers 9..1 backward-data: d(pre-act 10) in dact[0] -> d(pre-act 1) in dact[*cur_out] through the two ping-pong maps.
// conv_variant >= 5: pairs (9,8), (7,6), (5,4) in one launch each -- the intermediate gradient map never leaves the CU
template <class D>
static inline int enc_chain_bwd(const D& d, int H, int W, hipStream_t s, int* cur_out, int l_last = 1) {
  int cur = 0, l = 9;              // (l_last = 2: enc_tail takes d(pre-act 2) from here: conv variant 7)
  while (l >= l_last) {
    if (d.conv_variant >= 5 && d.conv_variant != 10 && l - 1 >= l_last && d.enc_wbwd3[l] && d.enc_wbwd3[l - 1] &&
        conv3x3_pair_supported(H, W, d.enc_ch[l + 1], d.enc_ch[l], d.enc_ch[l - 1])) {
      ENC_CHK(conv3x3_pair_f16(d.dact[cur], d.enc_wbwd3[l], d.enc_wbwd3_inv[l], nullptr, d.act[l], nullptr,
                                                                           d.enc_wbwd3[l - 1], d.enc_wbwd3_inv[l - 1], nullptr, d.act[l - 1],
                                                                           d.dact[1 - cur], H, W, 1, s, nullptr));
      l -= 2;
    } else {
      ENC_CHK(enc_layer(d, l, true, d.dact[cur], d.dact[1 - cur], H, W, s));
      --l;
    }
    cur = 1 - cur;
  }
  *cur_out = cur;
  return 0;
}

}  // namespace lemo
