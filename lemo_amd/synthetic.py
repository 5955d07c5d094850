"""SMPL-X-*shaped* synthetic body model and AMASS-*shaped* synthetic sequences.

The licensed SMPL-X ``.npz``, the VPoser checkpoint and AMASS are not available (SURVEY 7 "hard
parts"), so tests and ``bench.py`` run on seeded tensors with exactly the shapes, sparsity and
key names of the real files.  ``create()`` in :mod:`lemo_amd.body_model`
loads either a real ``SMPLX_*.npz`` or the dict returned here.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

# SMPL-X kinematic tree
SMPLX_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
     20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
     21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], dtype=np.int64)


def make_synthetic_smplx(seed: int = 0, V: int = 10475, F: int = 20908, n_shape: int = 20,
                         nnz_w: int = 4, nnz_j: int = 32) -> Dict[str, np.ndarray]:
    """Seeded SMPL-X-shaped model with the real file's key names.

    v_template ~ N(0, 0.3^2); shapedirs sigma 0.01; posedirs sigma 1e-3; J_regressor rows sum to 1
    with <= ``nnz_j`` non-zeros; skinning weights rows sum to 1 with <= ``nnz_w`` non-zeros.
    """
    rng = np.random.default_rng(seed)
    J = 55
    m = {}
    m['v_template'] = (rng.standard_normal((V, 3)) * 0.3).astype(np.float32)
    m['shapedirs'] = (rng.standard_normal((V, 3, n_shape)) * 0.01).astype(np.float32)
    m['posedirs'] = (rng.standard_normal((V, 3, (J - 1) * 9)) * 1e-3).astype(np.float32)
    jr = np.zeros((J, V), np.float32)
    for j in range(J):
        idx = rng.choice(V, size=min(nnz_j, V), replace=False)
        w = rng.random(idx.shape[0]).astype(np.float32) + 0.05
        jr[j, idx] = w / w.sum()
    m['J_regressor'] = jr
    wts = np.zeros((V, J), np.float32)
    cols = rng.integers(0, J, size=(V, nnz_w))
    vals = rng.random((V, nnz_w)).astype(np.float32) + 0.05
    for k in range(nnz_w):
        np.add.at(wts, (np.arange(V), cols[:, k]), vals[:, k])
    wts /= wts.sum(axis=1, keepdims=True)
    m['weights'] = wts.astype(np.float32)
    kt = np.zeros((2, J), np.int64)
    kt[0] = SMPLX_PARENTS
    kt[0, 0] = 2 ** 32 - 1                                  # real files store uint32(-1) at the root
    kt[1] = np.arange(J)
    m['kintree_table'] = kt
    m['f'] = rng.integers(0, V, size=(F, 3)).astype(np.int64)
    for s in ('l', 'r'):
        m['hands_components' + s] = (rng.standard_normal((45, 45)) * 0.3).astype(np.float32)
        m['hands_mean' + s] = (rng.standard_normal(45) * 0.1).astype(np.float32)
    m['lmk_faces_idx'] = rng.integers(0, F, size=(51,)).astype(np.int64)
    bary = rng.random((51, 3)).astype(np.float32) + 0.05
    m['lmk_bary_coords'] = (bary / bary.sum(axis=1, keepdims=True)).astype(np.float32)
    return m


def _smooth(rng, shape, sigma, win=9):
    """low-pass filtered N(0, sigma^2) along axis 0."""
    x = rng.standard_normal(shape) * sigma
    k = np.hanning(win + 2)[1:-1]
    k /= k.sum()
    pad = np.concatenate([np.repeat(x[:1], win // 2, 0), x, np.repeat(x[-1:], win // 2, 0)], 0)
    out = np.stack([np.convolve(pad[:, i], k, mode='valid') for i in range(x.shape[1])], 1)
    return out * (sigma / (out.std() + 1e-12))


def make_synthetic_sequence(seq_id: int = 0, B: int = 119) -> Dict[str, np.ndarray]:
    """AMASS-shaped per-frame parameters (seed 1000+seq_id).

    Returns ``init_params`` [B,72] (transl 3, global_orient aa 3, betas 10, vposer z 32,
    left/right hand PCA 12+12), a *perturbed* copy ``target_params`` whose model markers are the
    fitting target, and ``contact_lbl`` [B,4] in {0,1} with ~70 % ones in runs.
    """
    rng = np.random.default_rng(1000 + seq_id)
    p = np.zeros((B, 72), np.float64)
    p[:, 0:3] = np.cumsum(rng.standard_normal((B, 3)) * 0.01, 0) + np.array([0.0, 0.4, 1.0])
    p[:, 3:6] = np.array([0.0, 1.6, 3.14]) * 0.5 + _smooth(rng, (B, 3), 0.05)
    p[:, 6:16] = rng.standard_normal(10) * 0.5
    p[:, 16:48] = _smooth(rng, (B, 32), 0.7)
    p[:, 48:72] = rng.standard_normal((B, 24)) * 0.03
    tgt = p.copy()
    tgt[:, 0:6] += rng.standard_normal((B, 6)) * 0.02
    tgt[:, 16:] += rng.standard_normal((B, 56)) * 0.02
    lbl = np.zeros((B, 4), np.float32)
    for k in range(4):
        t = 0
        state = rng.random() < 0.7
        while t < B:
            run = int(rng.integers(8, 40))
            lbl[t:t + run, k] = 1.0 if state else 0.0
            t += run
            state = rng.random() < 0.7
    return dict(init_params=p.astype(np.float32), target_params=tgt.astype(np.float32), contact_lbl=lbl)


AE_CHANNELS = [(4, 32), (32, 64), (64, 128), (128, 256), (256, 256)]
"""models/AE.py:81-85 encoder (nin, nout) per block with in_channel=4."""


def make_ae_weights(seed: int = 7, in_channel: int = 4) -> Dict[str, np.ndarray]:
    """Seeded weights with the ``models/AE.py::AE`` state_dict keys/shapes (``runs/59547`` is absent
    from the reference checkout).  U(-1/sqrt(fan_in), 1/sqrt(fan_in))."""
    rng = np.random.default_rng(seed)
    w = {}

    def u(shape, fan_in):
        b = 1.0 / np.sqrt(fan_in)
        return ((rng.random(shape) * 2 - 1) * b).astype(np.float32)

    enc = [(in_channel, 32)] + AE_CHANNELS[1:]
    for i, (ci, co) in enumerate(enc, 1):
        w[f'enc_blc{i}.main.0.weight'] = u((co, ci, 3, 3), ci * 9)
        w[f'enc_blc{i}.main.0.bias'] = u((co,), ci * 9)
        w[f'enc_blc{i}.main.2.weight'] = u((co, co, 3, 3), co * 9)
        w[f'enc_blc{i}.main.2.bias'] = u((co,), co * 9)
    dec = [(256, 256), (256, 128), (128, 64), (64, 32), (32, 1)]
    for i, (ci, co) in enumerate(dec, 1):
        # ConvTranspose2d weight is [in, out, kh, kw]
        w[f'dec_blc{i}.deconv1.weight'] = u((ci, co, 3, 3), co * 9)
        w[f'dec_blc{i}.deconv1.bias'] = u((co,), co * 9)
        w[f'dec_blc{i}.deconv2.weight'] = u((co, co, 3, 3), co * 9)
        w[f'dec_blc{i}.deconv2.bias'] = u((co,), co * 9)
    return w
