"""PROX sliding-window schedule and the per-frame result ``.pkl`` wire format -- host logic.

Reference: ``temp_prox/data_parser_slide.py:199-212`` (windows of ``batch_size`` frames advanced by
``int(0.7 * batch_size)``; the frame lists of all windows are concatenated and consumed ``batch_size`` at a
time), ``:106-126`` (``read_prox_pkl``), ``:329-331`` (a window initialises every frame from the newest result
on disk: the current run's ``results/<frame>/000.pkl`` if it exists -- i.e. the overlap with the previous window
-- else the per-frame PROX fit), ``temp_prox/fit_temp_loadprox_slide.py:495-499`` (a window's shape = the mean of its frames' loaded betas),
``:577-594`` (one protocol-2 pickle per
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
    """[start, end) frame ranges of the windows of one recording, in processing order: the slices the reference's loop
    appends to ``img_paths_slide`` (data_parser_slide.py:199-212; checked against a literal restatement).  The last
    window(s) may be shorter than ``batch_size``.  NOTE the reference does not fit these windows one by one: it cuts
    their concatenation into DataLoader batches -- see :func:`reference_batches` for the batches it really forms
    (identical to these windows as long as every window is full)."""
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


def reference_batches(n_frames: int, batch_size: int, drop_last: bool = True) -> List[np.ndarray]:
    """The batches ``main_slide.py`` actually fits: the reference does NOT iterate over the windows -- it concatenates
    their frame lists (data_parser_slide.py:199-212) and lets a ``DataLoader(batch_size, shuffle=False,
    drop_last=True)`` (main_slide.py:146-149) cut that list into consecutive chunks.  Full windows are exactly one
    chunk each; the short tail windows are glued together, so a tail chunk can straddle two windows (e.g. n = 300,
    batch 100: windows (210,300) + (280,300) give the chunk 210..299 + 280..289), and an incomplete last chunk is
    dropped.  ``sliding_windows`` is the schedule as designed; this function is the schedule as executed."""
    idx = slide_frame_index(n_frames, batch_size)
    n_full = len(idx) // batch_size
    out = [idx[i * batch_size:(i + 1) * batch_size] for i in range(n_full)]
    if not drop_last and len(idx) % batch_size:
        out.append(idx[n_full * batch_size:])
    return out


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


def window_start_params(init: Mapping[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """what a window's fit starts from, given the stacked per-frame results (:func:`init_params_for_window`):
    ``fit_temp_loadprox_slide.py:495-499`` -- ``betas`` is replaced by its MEAN over the window's frames, repeated for every
    frame (the overlap carries the previous window's shape, the new frames the per-frame fits': one shape per window), and
    ``body_model.reset_params(**prox_params_dict)`` copies every named parameter; ``pose_embedding`` is taken as it is
    (:502-505).  float32 like the reference's numpy arrays (``np.mean`` of float32 rows in float32)."""
    out = {k: np.asarray(v) for k, v in init.items()}
    betas = np.asarray(init['betas'], np.float32)
    out['betas'] = np.repeat(np.expand_dims(np.mean(betas, axis=0), axis=0), betas.shape[0], axis=0)
    return out


def run_recording(frame_names: Sequence[str], batch_size: int, current_dir: str, prox_dir: str, fit_window,
                  reference_chunking: bool = False) -> int:
    """drive ``fit_window(frame_names, init_params, first_window, n_frozen) -> (camera_params, body_params,
    pose_embedding, body_pose)`` over one recording and write every frame's result (later windows overwrite the
    overlap, like the reference).  Default: one call per window of :func:`sliding_windows`, short tail windows included
    (every frame gets a result).  ``reference_chunking=True`` reproduces the batches the reference's DataLoader forms
    (:func:`reference_batches`: tail windows glued, incomplete last batch dropped) -- identical whenever every window is
    full.  Returns the number of calls."""
    if reference_chunking:
        batches = [list(b) for b in reference_batches(len(frame_names), batch_size)]
    else:
        batches = [list(range(s, e)) for s, e in sliding_windows(len(frame_names), batch_size)]
    for w, ids in enumerate(batches):
        names = [frame_names[i] for i in ids]
        init = window_start_params(init_params_for_window(names, current_dir, prox_dir))
        cam, body, emb, bp = fit_window(names, init, w == 0, frozen_prefix(batch_size, w == 0))
        for i, fn in enumerate(names):
            write_result_pkl(result_path(current_dir, fn), cam, body, emb, bp, i)
    return len(batches)
