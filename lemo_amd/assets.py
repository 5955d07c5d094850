"""Loaders for the data assets exported from the reference by ``tools/export_assets.py``.

These are DATA (vertex-id tables, normalisation statistics, the trained smoothness-encoder
weights ``runs/15217`` and one example clip) -- see that script for provenance (file:line).
"""
from __future__ import annotations

import os
from typing import Dict

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets')

ENC_KEYS = [f'enc_blc{b}.main.{i}.{p}' for b in range(1, 6) for i in (0, 2) for p in ('weight', 'bias')]


def asset_path(name: str) -> str:
    return os.path.join(_DIR, name)


def load_vertex_ids() -> Dict[str, np.ndarray]:
    """markers67 / markers81 (loader/SSM2*.json) and the four heel/toe id lists
    (opt_amass_temp.py:99-113 order)."""
    d = np.load(asset_path('vertex_ids.npz'))
    return {k: d[k].astype(np.int64) for k in d.files}


def load_smooth_stats():
    """preprocess_stats_smooth_withHand_global_markers.npz: Xmean (1,1,243) f32, Xstd (243,) f64."""
    d = np.load(asset_path('stats_smooth.npz'))
    return d['Xmean'], d['Xstd']


def load_smooth_encoder_weights() -> Dict[str, np.ndarray]:
    """``runs/15217/Enc_last_model.pkl`` as {state_dict key: ndarray}."""
    d = np.load(asset_path('smooth_enc_15217.npz'))
    return {k: d[k] for k in ENC_KEYS}


def load_example_clip():
    d = np.load(asset_path('example_clip0.npz'))
    return d['body_params'], d['contact_lbl']


def load_assets() -> dict:
    """Everything the AMASS temporal fit needs besides the body model and VPoser weights."""
    import torch
    Xmean, Xstd = load_smooth_stats()
    enc = load_smooth_encoder_weights()
    return dict(ids=load_vertex_ids(), Xmean=Xmean, Xstd=Xstd, enc_w=enc,
                enc_w_torch={k: torch.from_numpy(v) for k, v in enc.items()})
