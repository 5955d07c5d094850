"""Motion priors of LEMO on the HIP kernels: weight packing for the fp32-MFMA 3x3 convolutions and
``Enc`` / (later) ``AE`` modules with the reference's ``state_dict`` keys.

Reference: models/AE_sep.py:11-30,77-99 (``Enc(downsample=False, z_channel=64)``, the smoothness
prior -- SURVEY C1) and models/AE.py:78-108 (``AE``, the infilling prior).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import numpy as np
import torch
import torch.nn as nn

from . import _hip
from ._hip import ptr

ENC_CHANNELS = [1, 32, 32, 64, 64, 64, 64, 64, 64, 64, 64]
"""models/AE_sep.py:77-89 with z_channel=64: channels before/after each of the 10 conv layers."""


def enc_layer_keys() -> List[str]:
    return [f'enc_blc{b}.main.{i}' for b in range(1, 6) for i in (0, 2)]


def pack_conv3x3(w: np.ndarray) -> np.ndarray:
    """[Cout][Cin][3][3] -> wt[tap][Cin/8][Cout][8]  (forward: out[y,x] = sum w[ky,kx] in[y+ky-1,x+kx-1])."""
    co, ci = w.shape[:2]
    assert ci % 8 == 0
    t = w.reshape(co, ci // 8, 8, 9)                       # [co][g][pos][tap]
    return np.ascontiguousarray(t.transpose(3, 1, 0, 2), np.float32)


def pack_conv3x3_bwd(w: np.ndarray) -> np.ndarray:
    """Backward-data as a forward conv: roles of Cin/Cout swapped, taps flipped.
    [Cout][Cin][3][3] -> wt[tap'][Cout/8][Cin][8] with tap' = 8 - tap."""
    wf = w[:, :, ::-1, ::-1].transpose(1, 0, 2, 3)         # [ci][co][ky'][kx']
    return pack_conv3x3(np.ascontiguousarray(wf))


def pack_conv3x3_gmajor(w: np.ndarray) -> np.ndarray:
    """[Cout][Cin][3][3] -> wt2[Cin/8][tap][Cout][8] (LDS-tiled kernel: one channel group = 9 contiguous taps)."""
    return np.ascontiguousarray(pack_conv3x3(w).transpose(1, 0, 2, 3))


def pack_conv3x3_bwd_gmajor(w: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(pack_conv3x3_bwd(w).transpose(1, 0, 2, 3))


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> nearest-even bf16, returned as fp32 (low 16 bits zero)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def bf16_split3(x: np.ndarray):
    """x = hi + mid + lo exactly, each piece a bf16 value (held as fp32)."""
    x = np.ascontiguousarray(x, np.float32)
    hi = bf16_round(x)
    r1 = x - hi
    mid = bf16_round(r1)
    lo = bf16_round(r1 - mid)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    return hi, mid, lo


def pack_conv3x3_split(w: np.ndarray) -> np.ndarray:
    """[Cout][Cin][3][3] fp32 -> uint16 (bf16 bits) w3[Cin/16][tap][Cout/32][split 3][lane 64][8] for
    lemo_conv3x3_mfma_split: lane l = (cout & 31) + 32 * (channel group parity), 8 channels of the group."""
    co, ci = w.shape[:2]
    assert ci % 16 == 0 and co % 32 == 0
    pieces = np.stack(bf16_split3(w), 0)                    # [3][co][ci][3][3]
    bits = (pieces.view(np.uint32) >> 16).astype(np.uint16)
    t = bits.reshape(3, co // 32, 32, ci // 16, 2, 8, 9)    # [s][mt][i][kc][h][e][tap]
    t = t.transpose(3, 6, 1, 0, 4, 2, 5)                    # [kc][tap][mt][s][h][i][e]
    return np.ascontiguousarray(t).reshape(ci // 16, 9, co // 32, 3, 64, 8)


def pack_conv3x3_bwd_split(w: np.ndarray) -> np.ndarray:
    wf = w[:, :, ::-1, ::-1].transpose(1, 0, 2, 3)
    return pack_conv3x3_split(np.ascontiguousarray(wf))


def f16_split2(x: np.ndarray):
    """x * 2^k = hi + lo (two fp16 pieces, 2 x 11 significand bits) with 2^k chosen so that max|x| lands in [2^14, 2^15):
    returns (hi, lo, 2^-k).  The conv_variant-4 operand format (conv_split_kernels.hip header)."""
    x = np.ascontiguousarray(x, np.float32)
    m = float(np.abs(x).max())
    k = 14 - int(np.floor(np.log2(m))) if m > 0 else 0
    xs = x * np.float32(2.0 ** k)                              # exact (power of two)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    assert np.isfinite(hi).all()
    return hi, lo, float(2.0 ** -k)


def pack_conv3x3_split_f16(w: np.ndarray):
    """[Cout][Cin][3][3] fp32 -> (uint16 (f16 bits) w2[Cin/16][tap][Cout/32][piece 2][lane 64][8], winv) for
    lemo_conv3x3_mfma_split_f16: same fragment order as :func:`pack_conv3x3_split`, two pieces of weight * 2^k."""
    co, ci = w.shape[:2]
    assert ci % 16 == 0 and co % 32 == 0
    hi, lo, winv = f16_split2(w)
    bits = np.stack([hi, lo], 0).view(np.uint16)             # [2][co][ci][3][3]
    t = bits.reshape(2, co // 32, 32, ci // 16, 2, 8, 9)    # [s][mt][i][kc][h][e][tap]
    t = t.transpose(3, 6, 1, 0, 4, 2, 5)                    # [kc][tap][mt][s][h][i][e]
    return np.ascontiguousarray(t).reshape(ci // 16, 9, co // 32, 2, 64, 8), winv


def pack_conv3x3_bwd_split_f16(w: np.ndarray):
    wf = w[:, :, ::-1, ::-1].transpose(1, 0, 2, 3)
    return pack_conv3x3_split_f16(np.ascontiguousarray(wf))


WINO_VARIANT = 10
_WINO_G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], np.float64)


def pack_conv3x3_wino_f16(w: np.ndarray):
    """[64][64][3][3] fp32 -> (uint16 (f16 bits) wU[pos 16][kstep 4][mt 2][piece 2][lane 64][8], winv) for lemo_conv3x3_wino_f16
    (conv variant 10): U = G g G^T, the 4 x 4 Winograd F(2x2, 3x3) transform of every (cout, cin) filter, evaluated in float64 and
    split into two fp16 pieces of U * 2^k (one scale for the layer, max|U| -> [2^14, 2^15)); MFMA A-fragment order: lane l holds
    cout 32 mt + (l & 31), channels 16 kstep + 8 (l >> 5) .. + 7."""
    co, ci = w.shape[:2]
    assert co == 64 and ci == 64 and w.shape[2:] == (3, 3)
    U = np.einsum('ij,ocjk,lk->ocil', _WINO_G, np.asarray(w, np.float64), _WINO_G).reshape(co, ci, 16)
    m = float(np.abs(U).max())
    k = 14 - int(np.floor(np.log2(m))) if m > 0 else 0
    xs = U * 2.0 ** k
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float64)).astype(np.float16)
    assert np.isfinite(hi).all()
    bits = np.stack([hi, lo], 0).view(np.uint16)              # [s][co][ci][pos]
    t = bits.reshape(2, co // 32, 32, ci // 16, 2, 8, 16)     # [s][mt][i][ks][h][e][pos]
    t = t.transpose(6, 3, 1, 0, 4, 2, 5)                      # [pos][ks][mt][s][h][i][e]
    return np.ascontiguousarray(t).reshape(16, ci // 16, co // 32, 2, 64, 8), float(2.0 ** -k)


def pack_conv3x3_bwd_wino_f16(w: np.ndarray):
    wf = w[:, :, ::-1, ::-1].transpose(1, 0, 2, 3)
    return pack_conv3x3_wino_f16(np.ascontiguousarray(wf))


def cg8p_alloc(C_: int, H: int, W: int, device) -> torch.Tensor:
    """zeroed CG8P activation buffer [C/8][(H+2)*(W+2)][8] (border stays zero forever)."""
    return torch.zeros(max(C_ // 8, 1), (H + 2) * (W + 2), 8, dtype=torch.float32, device=device)


def to_cg8p(x: torch.Tensor) -> torch.Tensor:
    """[C,H,W] -> CG8P (test / plumbing helper)."""
    Cn, H, W = x.shape
    buf = torch.zeros(Cn // 8, H + 2, W + 2, 8, dtype=torch.float32, device=x.device)
    buf[:, 1:-1, 1:-1, :] = x.reshape(Cn // 8, 8, H, W).permute(0, 2, 3, 1)
    return buf.reshape(Cn // 8, (H + 2) * (W + 2), 8).contiguous()


def from_cg8p(buf: torch.Tensor, H: int, W: int) -> torch.Tensor:
    G = buf.shape[0]
    return buf.reshape(G, H + 2, W + 2, 8)[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).reshape(G * 8, H, W).contiguous()


class EncWeights:
    """Device-resident packed weights of the 10-layer smoothness encoder."""

    def split_pack(self, l: int, bwd: bool, variant: int):
        """(device pack, winv) of layer l for the split kernels: f16 x 2 pieces for variant >= 4, bf16 x 3 for 3"""
        if variant == WINO_VARIANT and self.w10[l] is not None:        # 64 -> 64 layers: the Winograd packs (pack_conv3x3_wino_f16)
            return (self.wbwd10[l], self.wbwd10_inv[l]) if bwd else (self.w10[l], self.w10_inv[l])
        if variant >= 4:
            return (self.wbwd4[l], self.wbwd4_inv[l]) if bwd else (self.w4[l], self.w4_inv[l])
        return ((self.wbwd3[l] if bwd else self.w3[l]), 1.0)

    def __init__(self, state: Dict[str, np.ndarray], device):
        self.keys = enc_layer_keys()
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device)
        self.w, self.b, self.wbwd, self.w2, self.wbwd2 = [], [], [], [], []
        self.w3, self.wbwd3 = [], []                       # split-bf16 packs (layer 0: None)
        self.w4, self.wbwd4, self.w4_inv, self.wbwd4_inv = [], [], [], []     # split-f16 packs + inverse host scales (variant 4)
        self.w10, self.wbwd10, self.w10_inv, self.wbwd10_inv = [], [], [], []  # Winograd packs of the 64 -> 64 layers (variant 10), else None
        t16 = lambda a: torch.from_numpy(a.view(np.int16)).to(device)
        for li, k in enumerate(self.keys):
            w = np.asarray(state[k + '.weight'], np.float32)
            b = np.asarray(state[k + '.bias'], np.float32)
            assert w.shape[0] == ENC_CHANNELS[li + 1] and w.shape[1] == ENC_CHANNELS[li], (k, w.shape)
            if li == 0:
                self.w.append(t(w.reshape(w.shape[0], 9)))
                self.wbwd.append(self.w[0])
                self.w2.append(self.w[0]); self.wbwd2.append(self.w[0])
                self.w3.append(None); self.wbwd3.append(None)
                self.w4.append(None); self.wbwd4.append(None); self.w4_inv.append(1.0); self.wbwd4_inv.append(1.0)
                self.w10.append(None); self.wbwd10.append(None); self.w10_inv.append(1.0); self.wbwd10_inv.append(1.0)
            else:
                self.w.append(t(pack_conv3x3(w)))
                self.wbwd.append(t(pack_conv3x3_bwd(w)))
                self.w2.append(t(pack_conv3x3_gmajor(w)))
                self.wbwd2.append(t(pack_conv3x3_bwd_gmajor(w)))
                self.w3.append(t16(pack_conv3x3_split(w)))
                self.wbwd3.append(t16(pack_conv3x3_bwd_split(w)))
                pf, fi = pack_conv3x3_split_f16(w)
                pb, bi = pack_conv3x3_bwd_split_f16(w)
                self.w4.append(t16(pf)); self.wbwd4.append(t16(pb)); self.w4_inv.append(fi); self.wbwd4_inv.append(bi)
                if w.shape[0] == 64 and w.shape[1] == 64:
                    uf, ufi = pack_conv3x3_wino_f16(w)
                    ub, ubi = pack_conv3x3_bwd_wino_f16(w)
                    self.w10.append(t16(uf)); self.wbwd10.append(t16(ub)); self.w10_inv.append(ufi); self.wbwd10_inv.append(ubi)
                else:
                    self.w10.append(None); self.wbwd10.append(None); self.w10_inv.append(1.0); self.wbwd10_inv.append(1.0)
            self.b.append(t(b))


# ----------------------------------------------------------------------------------------------
# Encoder as autograd ops over the C ABI
# ----------------------------------------------------------------------------------------------

DEFAULT_CONV_VARIANT = int(__import__('os').environ.get('LEMO_CONV_VARIANT', '9'))
"""Kernel family of the encoder's MFMA layers (``conv_variant`` of include/lemo_hip.h): 10 (round 6, selectable, measured slower: DESIGN 5) = 9 with every
64 -> 64 layer one Winograd F(2x2, 3x3) launch (csrc/conv_wino_kernels.hip) instead of the fused pairs; 9 (default since round 5) = 8 with layer 2's backward-data (64 -> 32) inside the
tail launch (+0.5 %); 8 = 7 with layer 2 (32 -> 64) inside the head launch as well (0 .. +0.9 % box to box); 7 = 5 plus the encoder's head and tail as one launch each (marker image + layers 0, 1 / their adjoints:
csrc/conv_head_kernels.hip, +2.1 % iterations/s); (6, the pairs on four-wave workgroups, was removed in round 6: csrc/attic); 5 = the engines run consecutive 64 -> 64
layers as fused PAIRS (one launch, intermediate in LDS; csrc/conv_pair_kernels.hip; the default since round 4) in the arithmetic
of 4 = split-f16 kernel (two error-compensated fp16 pieces per fp32 operand, 3 products; layer by layer -- what the module /
autograd path runs for 4 and 5 alike), 3 = split-bf16 kernel (three
exact bf16 pieces, 6 products), LDS-tiled fp32-MFMA kernel (2) for shapes they do not take; 2 / 1 = fp32 MFMA only.
The environment override exists for A/B runs of the parity suite."""


def check_conv_variant(v: int) -> int:
    """conv variants the library carries (include/lemo_hip.h, lemo_fit_desc.conv_variant); 6 -- the fused pairs on four-wave workgroups,
    measured 12 % slower in round 5 -- moved to csrc/attic in round 6"""
    v = int(v)
    if v == 6:
        raise ValueError('conv variant 6 (four-wave fused pairs) was removed in round 6: measured slower than variant 5 '
                         '(profiles/r05_pair4_check.txt); the source is in lemo_amd/csrc/attic')
    if v < 0 or v > WINO_VARIANT:
        raise ValueError(f'unknown conv variant {v}')
    return v


def _conv_layer(lib, enc: EncWeights, l: int, bwd: bool, x, out, aux, H, W, variant, s):
    """one MFMA layer (forward: epi 0 with bias; backward-data: epi 1 with the saved activation) on the best
    kernel family `variant` allows for its shape"""
    cin, cout = (ENC_CHANNELS[l + 1], ENC_CHANNELS[l]) if bwd else (ENC_CHANNELS[l], ENC_CHANNELS[l + 1])
    wt, wt2, w3 = (enc.wbwd[l], enc.wbwd2[l], enc.wbwd3[l]) if bwd else (enc.w[l], enc.w2[l], enc.w3[l])
    bias, epi = (None, 1) if bwd else (ptr(enc.b[l]), 0)
    auxp = ptr(aux) if bwd else None
    if variant == WINO_VARIANT and enc.w10[l] is not None and lib.conv3x3_wino_supported(H, W, cin, cout):
        wu, winv = enc.split_pack(l, bwd, variant)
        lib.check(lib.conv3x3_wino_f16(ptr(x), ptr(wu), winv, ptr(wt), bias, auxp, ptr(out), H, W, epi, None, s), 'conv3x3_wino_f16')
    elif variant >= 4 and w3 is not None and lib.conv3x3_split_supported(H, W, cin, cout):
        w4, winv = enc.split_pack(l, bwd, 4)
        lib.check(lib.conv3x3_mfma_split_f16(ptr(x), ptr(w4), winv, ptr(wt), bias, auxp, ptr(out), H, W, cin, cout, epi, s), 'conv3x3_mfma_split_f16')
    elif variant >= 3 and w3 is not None and lib.conv3x3_split_supported(H, W, cin, cout):
        lib.check(lib.conv3x3_mfma_split(ptr(x), ptr(w3), ptr(wt), bias, auxp, ptr(out), H, W, cin, cout, epi, s), 'conv3x3_mfma_split')
    elif variant >= 2 and 127 + 2 * (127 // W + 1) + 2 * (W + 2) + 3 <= 416:
        lib.check(lib.conv3x3_mfma_lds(ptr(x), ptr(wt), ptr(wt2), bias, auxp, ptr(out), H, W, cin, cout, epi, s), 'conv3x3_mfma_lds')
    else:
        lib.check(lib.conv3x3_mfma(ptr(x), ptr(wt), bias, auxp, ptr(out), H, W, cin, cout, epi, 1, s), 'conv3x3_mfma')


def _enc_forward(lib, enc: EncWeights, x: torch.Tensor, variant: int = None):
    """x [H,W] float32 on the device -> list of CG8P activations act[1..10] (act[0] = padded input)."""
    H, W = x.shape
    dev = x.device
    s = lib.stream(dev)
    variant = DEFAULT_CONV_VARIANT if variant is None else variant
    x0 = torch.zeros(H + 2, W + 2, dtype=torch.float32, device=dev)
    x0[1:-1, 1:-1] = x
    act = [x0] + [cg8p_alloc(ENC_CHANNELS[l], H, W, dev) for l in range(1, 11)]
    lib.check(lib.conv3x3_c1(ptr(x0), ptr(enc.w[0]), ptr(enc.b[0]), ptr(act[1]), H, W, ENC_CHANNELS[1], s), 'conv3x3_c1')
    for l in range(1, 10):
        _conv_layer(lib, enc, l, False, act[l], act[l + 1], None, H, W, variant, s)
    return act, variant


def _enc_backward(lib, enc: EncWeights, act, dpre10: torch.Tensor, variant: int) -> torch.Tensor:
    """d(pre-activation of layer 10) in CG8P -> d(input) [H,W]."""
    H, W = act[0].shape[0] - 2, act[0].shape[1] - 2
    dev = dpre10.device
    s = lib.stream(dev)
    cur, other = dpre10, cg8p_alloc(64, H, W, dev)
    for l in range(9, 0, -1):
        _conv_layer(lib, enc, l, True, cur, other, act[l], H, W, variant, s)
        cur, other = other, cur
    dx = torch.empty(H, W, dtype=torch.float32, device=dev)
    lib.check(lib.conv3x3_c1_bwd(ptr(cur), ptr(enc.w[0]), ptr(dx), H, W, ENC_CHANNELS[1], s), 'conv3x3_c1_bwd')
    return dx


class _SmoothPriorLoss(torch.autograd.Function):
    """x [H,W] -> mean((z[...,1:] - z[...,:-1])**2) with z = Enc(x): opt_amass_temp.py:389-391,
    temp_prox/fitting_temp_slide.py:1026-1031.  Loss and d(loss)/d(pre-act 10) come from one fused kernel."""

    @staticmethod
    def forward(ctx, x, enc: EncWeights, lib):
        x = x.contiguous().float()
        _hip.check_device(lib, x)
        H, W = x.shape
        act, use_lds = _enc_forward(lib, enc, x)
        cnt = 64 * H * (W - 1)
        nb = lib.smooth_loss_blocks(H, W, 64)
        part = torch.zeros(nb, dtype=torch.float32, device=x.device)
        dpre = cg8p_alloc(64, H, W, x.device)
        lib.check(lib.smooth_loss(ptr(act[10]), ptr(dpre), ptr(part), H, W, 64, 2.0 / cnt, lib.stream(x.device)), 'smooth_loss')
        ctx.enc, ctx.lib, ctx.act, ctx.dpre, ctx.use_lds = enc, lib, act, dpre, use_lds
        return (part.double().sum() / cnt).float()

    @staticmethod
    def backward(ctx, g):
        dx = _enc_backward(ctx.lib, ctx.enc, ctx.act, ctx.dpre, ctx.use_lds)
        return dx * g, None, None


class _EncFn(torch.autograd.Function):
    """x [H,W] -> z [64,H,W] (NCHW view of the last activation)."""

    @staticmethod
    def forward(ctx, x, enc: EncWeights, lib):
        x = x.contiguous().float()
        _hip.check_device(lib, x)
        act, use_lds = _enc_forward(lib, enc, x)
        ctx.enc, ctx.lib, ctx.act, ctx.use_lds = enc, lib, act, use_lds
        H, W = x.shape
        return from_cg8p(act[10], H, W)

    @staticmethod
    def backward(ctx, dz):
        H, W = dz.shape[1:]
        dz = dz.contiguous().float()
        z = from_cg8p(ctx.act[10], H, W)
        dpre = to_cg8p(dz * torch.where(z > 0, 1.0, 0.2))           # LeakyReLU'(pre-act 10) from the output sign
        return _enc_backward(ctx.lib, ctx.enc, ctx.act, dpre, ctx.use_lds), None, None


class _Conv3x3Param(nn.Module):
    """holds weight/bias under the reference's `main.{0,2}` key names"""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin, 3, 3))
        self.bias = nn.Parameter(torch.zeros(cout))


class _EncBlockParams(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.main = nn.ModuleDict({'0': _Conv3x3Param(cin, cout), '2': _Conv3x3Param(cout, cout)})


class Enc(nn.Module):
    """Drop-in for ``models/AE_sep.py::Enc(downsample=False, z_channel=64)`` (the smoothness prior):
    same ``state_dict`` keys (``enc_blcN.main.{0,2}.{weight,bias}``), ``forward(x[1,1,H,W]) ->
    (z, x.size(), s1, s2, s3, s4)`` like AE_sep.py:91-99.  Weights are treated as frozen (LEMO sets
    ``requires_grad=False``, opt_amass_temp.py:140-142); the backward is w.r.t. the input."""

    def __init__(self, downsample=False, z_channel=64, _lib=None):
        super().__init__()
        if downsample or z_channel != 64:
            raise NotImplementedError('LEMO instantiates the smoothness encoder as Enc(downsample=False, z_channel=64)')
        ch = ENC_CHANNELS
        for b in range(1, 6):
            setattr(self, f'enc_blc{b}', _EncBlockParams(ch[2 * b - 2], ch[2 * b]))
        self._lib_override, self._cache = _lib, {}

    def packed(self, device) -> EncWeights:
        ps = list(self.parameters())
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in ps)
        if self._cache.get('key') != key:
            sd = {k: v.detach().cpu().numpy() for k, v in self.state_dict().items()}
            self._cache = dict(key=key, enc=EncWeights(sd, device))
        return self._cache['enc']

    def forward(self, x):
        assert x.dim() == 4 and x.shape[0] == 1 and x.shape[1] == 1, 'the fitting path feeds [1,1,d,T] clips'
        lib = self._lib_override or _hip.get_lib()
        z = _EncFn.apply(x[0, 0], self.packed(x.device), lib)
        sz = lambda c: torch.Size([1, c, x.shape[2], x.shape[3]])
        return z.unsqueeze(0), x.size(), sz(32), sz(64), sz(64), sz(64)

    def smooth_loss(self, x):
        """mean((z[...,1:]-z[...,:-1])**2) for x [1,1,d,T] without materialising z in NCHW."""
        lib = self._lib_override or _hip.get_lib()
        return _SmoothPriorLoss.apply(x[0, 0], self.packed(x.device), lib)


def warn_if_wide_image(lib, H: int, W: int, conv_variant: int) -> bool:
    """The single-layer split kernels stage a 1-D tile of 128 pixels with its two halo rows and hold images up to W = 134 -- the
    T = 120 clip (and the 100-frame PROX window) they were built for.  Longer clips fit correctly and keep the fused 64 -> 64 pairs
    (any width), but the five encoder launches that are not pairs fall back to the fp32-input kernel (1563 it/s at
    238 frames where linear scaling from 119 would give ~1700).  Warn so that the slowdown is not silent; True when it applies."""
    import warnings
    H, W = int(H), int(W)
    too_wide = 127 + 2 * (127 // W + 1) + 2 * (W + 2) + 3 > 408          # conv_split_kernels.hip: CV3_NPX staged pixels
    if conv_variant >= 3 and H * W >= 128 and too_wide and not lib.conv3x3_split_supported(H, W, 64, 64):
        warnings.warn(f'lemo_amd: smoothness image {H} x {W} is wider than the split-f16 single-layer kernels take (W <= 134, i.e. clips of '
                      f'up to 120 frames); the 32-channel layers and the unpaired 64 -> 64 layers run on the slower fp32-input kernels',
                      RuntimeWarning, stacklevel=3)
        return True
    return False
