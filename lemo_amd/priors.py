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

    def __init__(self, state: Dict[str, np.ndarray], device):
        self.keys = enc_layer_keys()
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device)
        self.w, self.b, self.wbwd, self.w2, self.wbwd2 = [], [], [], [], []
        for li, k in enumerate(self.keys):
            w = np.asarray(state[k + '.weight'], np.float32)
            b = np.asarray(state[k + '.bias'], np.float32)
            assert w.shape[0] == ENC_CHANNELS[li + 1] and w.shape[1] == ENC_CHANNELS[li], (k, w.shape)
            if li == 0:
                self.w.append(t(w.reshape(w.shape[0], 9)))
                self.wbwd.append(self.w[0])
                self.w2.append(self.w[0]); self.wbwd2.append(self.w[0])
            else:
                self.w.append(t(pack_conv3x3(w)))
                self.wbwd.append(t(pack_conv3x3_bwd(w)))
                self.w2.append(t(pack_conv3x3_gmajor(w)))
                self.wbwd2.append(t(pack_conv3x3_bwd_gmajor(w)))
            self.b.append(t(b))
