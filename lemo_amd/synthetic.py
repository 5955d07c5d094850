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


# share of the 10475 vertices whose dominant skinning joint is joint j, shaped like the licensed model: head + face ~ 1/3 (FLAME),
# two eyeballs of 546 vertices, two hands of ~780 (15 finger joints each), the rest spread over trunk and limbs
_REGION_SHARE = np.array(
    [330, 300, 300, 330, 270, 270, 350, 170, 170, 420, 150, 150, 230, 190, 190, 2350, 260, 260, 230, 230, 130, 130, 850, 546, 546] +
    [52] * 30, dtype=np.float64)


def _coherent_skinning(rng, V: int, nnz_w: int, nnz_j: int):
    """Skinning weights / joint regressor / template with the LOCALITY of the licensed SMPL-X model, which Appendix B's shape and
    non-zero bounds do not capture: consecutive vertex indices belong to the same body part (runs of 20 .. 400 vertices whose
    dominant joint is the same, neighbouring runs on joints adjacent in the kinematic tree), a vertex is skinned to its part's
    joint and up to three tree neighbours (parent, children, grandparent), about a third of the vertices rigidly to one or two
    joints; a joint is regressed from vertices of its own part; the template is a skeleton with the parts' vertices scattered
    5 cm around their joints.  A 512-vertex chunk then touches 2 .. 12 joints instead of all 55 -- what the chunked LBS backward
    and the skinning gather see on the real model (the i.i.d.-joint model above is the worst case)."""
    J = 55
    par = SMPLX_PARENTS
    children = [[c for c in range(J) if par[c] == j] for j in range(J)]
    share = _REGION_SHARE / _REGION_SHARE.sum()
    cnt = np.maximum(np.floor(share * V).astype(np.int64), 1)
    while cnt.sum() > V:
        cnt[np.argmax(cnt)] -= 1
    cnt[15] += V - cnt.sum()
    # runs: every part is cut into pieces of 20 .. 400 vertices; the pieces are laid out along a depth-first walk of the tree
    # (children in random order), a part's pieces split between the way down and the way back up
    down, up = [[] for _ in range(J)], [[] for _ in range(J)]
    for j in range(J):
        left = int(cnt[j])
        pieces = []
        while left > 0:
            n = int(min(left, rng.integers(20, 401)))
            pieces.append(n); left -= n
        k = (len(pieces) + 1) // 2
        down[j], up[j] = pieces[:k], pieces[k:]
    order = []

    def walk(j):
        order.extend((j, n) for n in down[j])
        for c in rng.permutation(children[j]) if children[j] else []:
            walk(int(c))
        order.extend((j, n) for n in up[j])
    walk(0)
    prim = np.concatenate([np.full(n, j, np.int64) for j, n in order])
    assert prim.shape[0] == V
    # joint rest positions: a skeleton with 8 .. 25 cm bones
    jpos = np.zeros((J, 3))
    for j in range(1, J):
        d = rng.standard_normal(3)
        jpos[j] = jpos[par[j]] + d / np.linalg.norm(d) * (0.03 if j >= 25 else rng.uniform(0.08, 0.25))
    v_template = (jpos[prim] + rng.standard_normal((V, 3)) * 0.05).astype(np.float32)
    wts = np.zeros((V, J), np.float32)
    for v in range(V):
        j = int(prim[v])
        nb = ([int(par[j])] if par[j] >= 0 else []) + children[j] + ([int(par[par[j]])] if par[j] >= 0 and par[par[j]] >= 0 else [])
        r = rng.random()
        k = 1 if r < 0.15 else (2 if r < 0.35 else (3 if r < 0.6 else nnz_w))
        k = min(k, nnz_w, 1 + len(nb))
        others = list(rng.permutation(nb)[:k - 1]) if k > 1 else []
        w = np.concatenate([[1.0 + rng.random()], rng.random(len(others)) * 0.6 + 0.02])
        wts[v, [j] + [int(o) for o in others]] = (w / w.sum()).astype(np.float32)
    wts /= wts.sum(axis=1, keepdims=True)
    jr = np.zeros((J, V), np.float32)
    for j in range(J):
        own = np.nonzero(prim == j)[0]
        idx = rng.choice(own, size=min(nnz_j, own.shape[0]), replace=False)
        w = rng.random(idx.shape[0]).astype(np.float32) + 0.05
        jr[j, idx] = w / w.sum()
    return v_template, wts, jr


DEFAULT_COHERENT = __import__('os').environ.get('LEMO_SYNTH_MODEL', 'iid') == 'coherent'
"""what ``coherent=None`` means (``bench.py --model coherent`` sets it for every problem the process builds; tests and golden
fixtures use the i.i.d. model)"""


def make_synthetic_smplx(seed: int = 0, V: int = 10475, F: int = 20908, n_shape: int = 20,
                         nnz_w: int = 4, nnz_j: int = 32, coherent=None) -> Dict[str, np.ndarray]:
    """Seeded SMPL-X-shaped model with the real file's key names.

    v_template ~ N(0, 0.3^2); shapedirs sigma 0.01; posedirs sigma 1e-3; J_regressor rows sum to 1
    with <= ``nnz_j`` non-zeros; skinning weights rows sum to 1 with <= ``nnz_w`` non-zeros.
    ``coherent``: template, skinning weights and joint regressor with the index locality of the licensed model
    (:func:`_coherent_skinning`) instead of i.i.d. joints per vertex; everything else is drawn the same way.
    """
    rng = np.random.default_rng(seed)
    J = 55
    m = {}
    coherent = DEFAULT_COHERENT if coherent is None else bool(coherent)
    if coherent:
        crng = np.random.default_rng(seed + 7919)
        vt, cw, cjr = _coherent_skinning(crng, V, nnz_w, nnz_j)
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
    if coherent:
        m['v_template'], m['weights'], m['J_regressor'] = vt, cw, cjr
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
