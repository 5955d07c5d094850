"""PROX sliding-window schedule and the per-frame result ``.pkl`` wire format -- host logic.

Reference: ``temp_prox/data_parser_slide.py:199-212`` (windows of ``batch_size`` frames advanced by
``int(0.7 * batch_size)``; the frame lists of all windows are concatenated and consumed ``batch_size`` at a
time), ``:106-126`` (``read_prox_pkl``), ``:329-331`` (a window initialises every frame from the newest result
on disk: the current run's ``results/<frame>/000.pkl`` if it exists -- i.e. the overlap with the previous window
-- else the per-frame PROX fit), ``temp_prox/fit_temp_loadprox_slide.py:577-594`` (one protocol-2 pickle per
frame with ``camera_*``, the body-model parameters, ``pose_embedding`` and the decoded ``body_pose``, each with a
leading axis of 1), ``fitting_temp_slide.py:282-289`` (gradients of the first ``int(0.15 B)`` frames are erased in
every window but the first -- implemented in :class:`lemo_amd.prox.ProxTemporalFitter`).
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, Iterable, List, Mapping, Sequence, Tuple

import numpy as np

BODY_PARAM_KEYS = ('transl', 'global_orient', 'betas', 'body_pose', 'pose_embedding', 'left_hand_pose',
                   'right_hand_pose', 'jaw_pose', 'leye_pose', 'reye_pose', 'expression')
"""keys ``read_prox_pkl`` returns (data_parser_slide.py:106-126)."""


def slide_stride(batch_size: int) -> int:
    return int(batch_size * 0.7)


def sliding_windows(n_frames: int, batch_size: int) -> List[Tuple[int, int]]:
    """[start, end) frame ranges of the windows of one recording, in processing order.  The last window may be
    shorter than ``batch_size`` (the reference then feeds a short batch); recordings shorter than one window plus
    one stride yield the first window only... exactly as ``range(int(seq_n) + 1)`` does for negative ``seq_n``."""
    if n_frames < 1 or batch_size < 2:
        raise ValueError('need n_frames >= 1 and batch_size >= 2')
    stride = slide_stride(batch_size)
    wins = [(0, min(batch_size, n_frames))]
    n_more = n_frames - batch_size - stride + 1          # the reference loops i = 0 .. seq_n; most of those
    for i in range(max(n_more, 0)):                      # slices start past the end and are empty
        start = stride * (i + 1)
        if start >= n_frames:
            break
        wins.append((start, min(start + batch_size, n_frames)))
    return wins


def slide_frame_index(n_frames: int, batch_size: int) -> np.ndarray:
    """the concatenated frame list (``img_paths_slide``) as frame numbers"""
    return np.concatenate([np.arange(s, e) for s, e in sliding_windows(n_frames, batch_size)])


def frozen_prefix(batch_size: int, first_window: bool) -> int:
    """number of leading frames of a window whose gradients are erased (fitting_temp_slide.py:282-289)"""
    return 0 if first_window else int(0.15 * batch_size)


def result_path(root: str, frame_name: str) -> str:
    return os.path.join(root, 'results', frame_name, '000.pkl')


def write_result_pkl(path: str, camera_params: Mapping[str, np.ndarray], body_params: Mapping[str, np.ndarray],
                     pose_embedding: np.ndarray, body_pose: np.ndarray, i: int) -> Dict[str, np.ndarray]:
    """frame ``i`` of a fitted window -> ``path``; returns the dict that was written.  ``camera_params`` /
    ``body_params`` are the named parameters ([B, ...] arrays) of the camera and the body model."""
    row = lambda a: np.asarray(a)[i][None]
    result = {'camera_' + str(k): row(v) for k, v in camera_params.items()}
    result.update({k: row(v) for k, v in body_params.items()})
    result['pose_embedding'] = row(pose_embedding)
    result['body_pose'] = row(body_pose)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'wb') as f:
        pickle.dump(result, f, protocol=2)
    return result


def read_prox_pkl(path: str) -> Dict[str, np.ndarray]:
    with open(path, 'rb') as f:
        data = pickle.load(f)
    return {k: data[k][0] for k in BODY_PARAM_KEYS}


def init_params_for_window(frame_names: Sequence[str], current_dir: str, prox_dir: str) -> Dict[str, np.ndarray]:
    """stack the newest per-frame results: the current run's if present (overlap with the previous window), else the
    per-frame PROX fit (data_parser_slide.py:329-331)"""
    rows = []
    for fn in frame_names:
        p = result_path(current_dir, fn)
        rows.append(read_prox_pkl(p if os.path.exists(p) else result_path(prox_dir, fn)))
    return {k: np.stack([r[k] for r in rows]) for k in BODY_PARAM_KEYS}


def run_recording(frame_names: Sequence[str], batch_size: int, current_dir: str, prox_dir: str, fit_window) -> int:
    """drive ``fit_window(frame_names, init_params, first_window, n_frozen) -> (camera_params, body_params,
    pose_embedding, body_pose)`` over the windows of one recording and write every frame's result (later windows
    overwrite the overlap, like the reference).  Returns the number of windows."""
    wins = sliding_windows(len(frame_names), batch_size)
    for w, (s, e) in enumerate(wins):
        names = list(frame_names[s:e])
        init = init_params_for_window(names, current_dir, prox_dir)
        cam, body, emb, bp = fit_window(names, init, w == 0, frozen_prefix(batch_size, w == 0))
        for i, fn in enumerate(names):
            write_result_pkl(result_path(current_dir, fn), cam, body, emb, bp, i)
    return len(wins)
