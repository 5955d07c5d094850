"""The per-clip / per-window pipeline AROUND the fitting loop, on the device end to end.

AMASS (``opt_amass_temp.py``), one clip:
    :166-185   ``amass_mask_input``   hide the 22 upper-body markers and the contact rows, reflect-pad (8,8,1,1)
    :162-214   ``lemo_amd.infill.finetune_and_infill``  60 x [AE forward, L1 on the un-masked rows, backward, Adam]
    :273-325   ``decode_markers``     sigmoid -> contact labels, de-normalise, reorder, trajectory integration
                                      (ONE HIP launch: ``lemo_decode_clip``)
    :332-458   ``AmassTemporalFitter``  100 Adam iterations -> ``body_params_opt_t_72`` [T,72]
    :263-269   gender -> which SMPL-X model (``AmassClipPipeline(fitters={'male': ..., 'female': ...})``)
PROX (``temp_prox/fitting_temp_slide.py:776-941``), once per window at ``opt_step == 0``: ``prox_window_setup``.

Nothing here returns to the host between the stages: the clip image, the finetuned reconstruction, the decoded
markers and the fitted parameters are device tensors; ``rot_0_pivot`` is read by the decode kernel from device
memory.  The small index / mask algebra is torch glue on [208 x 119]-sized tensors; all arithmetic of substance
(AE, decode, encode, fit) runs in liblemo_hip.so.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import _hip
from ._hip import ptr
from .assets import asset_path
from .infill import AE, finetune_and_infill
from .markers import get_local_markers_4chan

MASK_MARKER_IDS = (14, 15, 18, 19, 29, 2, 20, 21, 30, 25, 16, 45, 46, 48, 49, 59, 32, 50, 51, 55, 60, 47)
"""opt_amass_temp.py:171-172 -- the upper-body markers hidden from the infilling network."""
P2D = (8, 8, 1, 1)
FOOT_MARKERS = (16, 47, 30, 60)
"""left heel, right heel, left toe, right toe among the 67 markers (fitting_temp_slide.py:795-799)."""


def load_infill_stats() -> Dict[str, np.ndarray]:
    """preprocess_stats/preprocess_stats_infill_local_markers_4chan.npz (exported by tools/export_assets.py)"""
    d = np.load(asset_path('stats_infill.npz'))
    return {k: d[k] for k in d.files}


_CONST_CACHE: Dict[tuple, torch.Tensor] = {}
"""small per-device constants (the statistics vector, the masked rows).  Uploading them per clip is not just a copy: a pageable
host-to-device copy is ordered behind everything the current stream holds, so the host thread stood still until the previous
clip's work had drained (62 of a clip's 66 ms were host time before round 3 cached them, tools/clip_pipeline_rate.py)."""


def _stats_vector(stats: Dict[str, np.ndarray], device) -> torch.Tensor:
    v = np.concatenate([np.asarray(stats['Xmean_local'], np.float64).reshape(-1), np.asarray(stats['Xstd_local'], np.float64).reshape(-1),
                        [float(stats['Xmean_global_xy']), float(stats['Xstd_global_xy']), float(stats['Xmean_global_r']),
                         float(stats['Xstd_global_r'])]])
    key = ('stats', v.tobytes(), str(device))                  # keyed by CONTENT: callers pass fresh dicts (load_infill_stats())
    if key not in _CONST_CACHE:
        _CONST_CACHE[key] = torch.from_numpy(v).to(device)
    return _CONST_CACHE[key]


def normalise_clip_image(img: torch.Tensor, stats: Dict[str, np.ndarray]) -> torch.Tensor:
    """[4, T, d] raw 4-channel image (``get_local_markers_4chan``) -> normalised [1, 4, d, T] as the loader hands it
    over (loader/optimize_loader_amass_new.py:351-355, 371-377; fitting_temp_slide.py:825-831)."""
    f = lambda k: torch.as_tensor(np.asarray(stats[k]), dtype=torch.float32, device=img.device)
    x = img.float().unsqueeze(0).clone()
    x[:, 0] = (x[:, 0] - f('Xmean_local')) / f('Xstd_local')
    x[:, 1:3] = (x[:, 1:3] - f('Xmean_global_xy')) / f('Xstd_global_xy')
    x[:, 3] = (x[:, 3] - f('Xmean_global_r')) / f('Xstd_global_r')
    return x.permute(0, 1, 3, 2).contiguous()


def amass_mask_rows() -> np.ndarray:
    """rows of the un-padded [d = 208] image zeroed at opt_amass_temp.py:173-183 (body_mode local_markers_4chan)"""
    r1 = np.asarray(MASK_MARKER_IDS) * 3 + 3
    return np.concatenate([r1, r1 + 1, r1 + 2])


def amass_mask_input(clip_img: torch.Tensor):
    """clip_img [1,4,d,T] -> (clip_img_input [1,4,d+2,T+16], train_mask bool [d+2,T+16]).
    ``train_mask`` is the reference's ``res_map[:, upper_body_row][:, 0:-5]`` selection (:199-204): every padded row that
    is not a masked marker row, minus the last five (4 contact rows + the pad row)."""
    # (no indexed assignment with a device index tensor here: `x[:, 0, rows, :] = 0` made the host wait for the device -- 48 ms per
    # clip behind the previous clip's fit, tools/clip_pipeline_rate.py -- the row selections are cached boolean masks instead)
    d, T = clip_img.shape[-2], clip_img.shape[-1]
    key = ('amass_masks', d, T, str(clip_img.device))
    if key not in _CONST_CACHE:
        rows = torch.as_tensor(amass_mask_rows())
        shown = torch.ones(d, dtype=torch.bool)
        shown[rows] = False
        shown[-4:] = False
        keep = torch.ones(d + 2, dtype=torch.bool)
        keep[rows + 1] = False
        keep[-5:] = False
        _CONST_CACHE[key] = (shown.to(clip_img.device), keep[:, None].expand(d + 2, T + 16).contiguous().to(clip_img.device))
    shown, keep = _CONST_CACHE[key]
    x = clip_img.clone()
    x[:, 0] = torch.where(shown[:, None], x[:, 0], torch.zeros((), dtype=x.dtype, device=x.device))
    x = F.pad(x, P2D, 'reflect')
    return x, keep.clone()          # a fresh tensor per call (device-to-device copy, no host sync): callers may edit it in place


def decode_markers(clip_img_rec: torch.Tensor, clip_img: torch.Tensor, rot_0_pivot, stats: Optional[Dict[str, np.ndarray]] = None,
                   post: Optional[torch.Tensor] = None, _lib=None):
    """opt_amass_temp.py:273-325.  clip_img_rec [d,T] (channel 0 of the network output, un-padded), clip_img [4,d,T]
    (rows 0 of channels 1-3 give the global trajectory), rot_0_pivot: device float64 tensor [1] (or a host scalar).
    Returns (contact_lbl_rec [T,4] in {0,1}, markers_rec [T,67,3] global) -- device tensors, one kernel launch."""
    lib = _lib or _hip.get_lib()
    dev = clip_img_rec.device
    _hip.check_device(lib, clip_img_rec)
    d, T = clip_img_rec.shape
    J = (d - 4) // 3
    assert d == 3 * J + 4 and clip_img.shape[-2:] == (d, T)
    st = _stats_vector(load_infill_stats() if stats is None else stats, dev)
    assert st.numel() == 2 * d + 4
    rec = clip_img_rec.detach().float().contiguous()
    traj = torch.stack([clip_img[1, 0], clip_img[2, 0], clip_img[3, 0]], 0).detach().float().contiguous()
    if isinstance(rot_0_pivot, torch.Tensor):
        piv = rot_0_pivot.detach().to(dev, torch.float64).reshape(-1)[:1].contiguous()
    else:
        piv = torch.tensor([float(np.asarray(rot_0_pivot).reshape(-1)[0])], dtype=torch.float64, device=dev)
    lbl = torch.empty(T, 4, dtype=torch.float32, device=dev)
    mk = torch.empty(T, J - 1, 3, dtype=torch.float32, device=dev)
    pp = None if post is None else post.detach().float().contiguous()
    lib.check(lib.decode_clip(ptr(rec), ptr(traj), ptr(st), ptr(piv), ptr(pp), T, J, ptr(lbl), ptr(mk), lib.stream(dev)), 'decode_clip')
    return lbl, mk


class AmassClipPipeline:
    """One clip of ``opt_amass_temp.py`` end to end: finetune the infilling AE on the masked clip, decode its output into
    target markers + contact labels, run the temporal fit, return the reference's ``body_params_opt_t_72``."""

    def __init__(self, fitters, ae: AE, ae_weights: Dict[str, torch.Tensor], stats: Optional[Dict[str, np.ndarray]] = None):
        self.fitters = fitters if isinstance(fitters, dict) else {'male': fitters, 'female': fitters}
        self.ae, self.ae_weights = ae, ae_weights
        self.stats = load_infill_stats() if stats is None else stats

    def fit_clip(self, clip_img: torch.Tensor, rot_0_pivot, init_params, gender=1, steps: int = 100, finetune_steps: int = 60,
                 use_graph: Optional[bool] = None) -> Dict[str, torch.Tensor]:
        """clip_img [1,4,208,T] normalised (device), init_params [T,72] (the per-frame result), gender 0 / 'female' or
        1 / 'male' (:263-269).  Returns dict(p72 [T,72] = body_params_opt_t_72 of the last iteration's forward (:457),
        contact_lbl_rec, markers_rec, clip_img_rec)."""
        g = gender if isinstance(gender, str) else ('female' if int(gender) == 0 else 'male')
        fit = self.fitters[g]
        x_in, mask = amass_mask_input(clip_img)
        rec, _ = finetune_and_infill(self.ae, self.ae_weights, x_in, mask, steps=finetune_steps, use_graph=use_graph)
        lbl, markers = decode_markers(rec[0, 0], clip_img[0], rot_0_pivot, self.stats, _lib=self.ae._lib_override)
        fit.load_sequence(init_params, markers, lbl)
        # the fitter's own persistent stream (its graphs are captured once); params72() orders itself after the run
        fit.step_async(steps, use_graph=True if use_graph is None else bool(use_graph))
        return dict(p72=fit.params72(), contact_lbl_rec=lbl, markers_rec=markers, clip_img_rec=rec, clip_img_input=x_in,
                    train_mask=mask)


    def fit_clips(self, clips, steps: int = 100, finetune_steps: int = 60, use_graph: Optional[bool] = None):
        """``fit_clip`` for a LIST of clips ``(clip_img, rot_0_pivot, init_params, gender)`` at dataset rate.  The clips go through
        the infilling AE ``infill.AE_CLIPS`` at a time (``finetune_and_infill_many``: every launch of a finetune step carries the
        whole group -- 18.3 instead of 29 ms per clip, on the caller's stream, no second hardware queue needed); each clip of the
        group is then decoded and fitted on its fitter's own stream, so the next group's finetune is enqueued while those fits
        replay.  Nothing in the loop makes the host wait for the device: ``init_params`` goes up on a separate upload stream (a
        pageable host-to-device copy on the caller's stream would wait for everything queued there), the fitted parameters are read
        on a separate result stream, and a fitter's next ``load_sequence`` waits for that read.  Whether the fits and the next
        finetune actually overlap is up to the runtime's hardware queues (round 3 steered that with an empirical rule about torch's
        stream pool; deleted): one after the other they cost 18.3 + 30 ms per clip, which is the rate this path promises.  Returns
        the list of ``fit_clip`` dicts, ordered on the caller's stream when the call returns.  Same kernels, same order per clip:
        results are identical to ``fit_clip`` one by one (tested)."""
        from . import infill
        from .infill import finetune_and_infill_many
        dev = next(iter(self.fitters.values())).device
        gpu = dev.type == 'cuda' and torch.cuda.is_available() and not next(iter(self.fitters.values())).lib.is_emu
        if gpu and getattr(self, '_up', None) is None:
            self._up, self._res = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            self._read_ev = {}
        cur = torch.cuda.current_stream(dev) if gpu else None
        outs, pinned = [], []
        clips = list(clips)
        for g0 in range(0, len(clips), infill.AE_CLIPS):
            group = clips[g0:g0 + infill.AE_CLIPS]
            masked = [amass_mask_input(c[0]) for c in group]
            recs = finetune_and_infill_many(self.ae, self.ae_weights, [m[0] for m in masked], [m[1] for m in masked], steps=finetune_steps,
                                            use_graph=use_graph)
            for (clip_img, rot_0_pivot, init_params, gender), (x_in, mask), (rec, _) in zip(group, masked, recs):
                g = gender if isinstance(gender, str) else ('female' if int(gender) == 0 else 'male')
                fit = self.fitters[g]
                lbl, markers = decode_markers(rec[0, 0], clip_img[0], rot_0_pivot, self.stats, _lib=self.ae._lib_override)
                if not gpu:
                    fit.load_sequence(init_params, markers, lbl)
                    fit.step_async(steps, use_graph=True if use_graph is None else bool(use_graph))
                    outs.append(dict(p72=fit.params72(), contact_lbl_rec=lbl, markers_rec=markers, clip_img_rec=rec, clip_img_input=x_in,
                                     train_mask=mask))
                    continue
                if isinstance(init_params, torch.Tensor) and init_params.is_cuda:
                    p_dev = init_params
                else:
                    hp = torch.from_numpy(np.ascontiguousarray(np.asarray(init_params, np.float32))).pin_memory()
                    with torch.cuda.stream(self._up):
                        p_dev = hp.to(dev, non_blocking=True)     # pinned + its own stream: the host does not wait for the device
                        ev = torch.cuda.Event(); ev.record(self._up)
                    pinned.append(hp)                             # alive until the copies have run (joined below)
                    cur.wait_event(ev)
                    p_dev.record_stream(cur)
                if id(fit) in self._read_ev:
                    cur.wait_event(self._read_ev[id(fit)])        # the previous clip's parameters have been read out
                fit.load_sequence(p_dev, markers, lbl)
                fit.step_async(steps, use_graph=True if use_graph is None else bool(use_graph))
                with torch.cuda.stream(self._res):
                    p72 = fit.params72()                          # waits for the fit on the RESULT stream only
                    ev = torch.cuda.Event(); ev.record(self._res)
                p72.record_stream(cur)
                self._read_ev[id(fit)] = ev
                outs.append(dict(p72=p72, contact_lbl_rec=lbl, markers_rec=markers, clip_img_rec=rec, clip_img_input=x_in, train_mask=mask))
        if gpu:
            cur.wait_stream(self._res)
            self._up.synchronize()                                # (the uploads finished long ago: releases `pinned` safely)
        return outs


# ----------------------------------------------------------------------------------------------------------------------
# PROX: the opt_step == 0 block of SMPLifyLoss.forward
# ----------------------------------------------------------------------------------------------------------------------
def canonical_frame(joints_frame0: torch.Tensor) -> torch.Tensor:
    """fitting_temp_slide.py:783-789: x = (j2 - j1) with z zeroed, normalised; y = z x x; R0 = [x y z] columns"""
    x = joints_frame0[2] - joints_frame0[1]
    x = torch.cat([x[:2], x.new_zeros(1)])
    x = x / torch.norm(x)
    z = x.new_tensor([0., 0., 1.])
    y = torch.linalg.cross(z, x)
    y = y / torch.norm(y)
    return torch.stack([x, y, z], dim=1)


def prox_contact_labels(markers_normed: torch.Tensor) -> torch.Tensor:
    """:793-816 -- contact = marker speed < 0.22 m/s (30 fps) AND height < (lowest marker + 0.10); last frame: height only"""
    foot = list(FOOT_MARKERS)
    m = markers_normed[:, foot]                                              # [T,4,3]
    vel = torch.norm((m[1:] - m[:-1]) * 30, dim=-1)                          # [T-1,4]
    thres = (m[:, :, 2] < markers_normed[:, :, 2].min() + 0.10).float()      # [T,4]
    lbl = torch.cat([(vel.abs() < 0.22).float(), thres.new_zeros(1, 4)], 0) * thres
    lbl[-1] = thres[-1]
    return lbl


def prox_window_setup(vertices_world: torch.Tensor, smplx_joints_world: torch.Tensor, marker_mask: torch.Tensor, ae: AE,
                      ae_weights: Dict[str, torch.Tensor], markers67_ids, stats: Optional[Dict[str, np.ndarray]] = None,
                      finetune_steps: int = 60, use_graph: Optional[bool] = None):
    """fitting_temp_slide.py:776-941: from the window's initial body (world-frame vertices [T,V,3] and unmapped joints
    [T,>=25,3]) and the occlusion mask [T,67] (1 = visible) produce the constants the infill terms use for the rest of
    the window: ``body_markers_rec`` [T-1,67,3] (PROX world frame) and ``contact_lbl_rec`` [T-1,4].  Returns None when
    nothing is occluded (:858).  No host round trip except that one early-out test."""
    stats = load_infill_stats() if stats is None else stats
    dev = vertices_world.device
    lib = ae._lib_override
    vw, jw = vertices_world.detach(), smplx_joints_world.detach()
    ids = torch.as_tensor(np.asarray(markers67_ids, np.int64), device=dev)
    markers = vw[:, ids]
    joints = jw[:, 0:25]
    j0 = joints[0]
    R0 = canonical_frame(j0)
    joints_n = torch.matmul(joints - j0[0], R0)
    markers_n = torch.matmul(markers - j0[0], R0)
    lbls = prox_contact_labels(markers_n)
    cur_body = torch.cat([joints_n[:, 0:1], markers_n], dim=1).contiguous()
    img, piv = get_local_markers_4chan(cur_body, lbls, _lib=lib)                 # [4,T-1,d], rot_0_pivot (device f64)
    clip_img = normalise_clip_image(img, stats)                                 # [1,4,d,T-1]
    mm = marker_mask.to(dev).float().repeat_interleave(3, dim=1).t()[None, None]  # [1,1,201,T]
    left = ((mm[:, :, 48:49] == 1) & (mm[:, :, 90:91] == 1)).float()
    right = ((mm[:, :, 141:142] == 1) & (mm[:, :, 180:181] == 1)).float()
    T = clip_img.shape[-1]
    mask = torch.cat([mm.new_ones(1, 1, 3, T), mm[..., :T], left[..., :T], right[..., :T], left[..., :T], right[..., :T]], dim=-2)
    x_in = clip_img.clone()
    x_in[:, 0:1] = x_in[:, 0:1] * mask
    if not bool(marker_mask.numel() > marker_mask.sum()):
        return None
    x_in = F.pad(x_in, P2D, 'reflect')
    mf = F.pad(mask, P2D, 'reflect')[0, 0].clone()
    mf[-5:, :] = 0
    rec, _ = finetune_and_infill(ae, ae_weights, x_in, mf == 1, steps=finetune_steps, use_graph=use_graph)
    post = torch.cat([markers_n[:, :, 2].min().reshape(1), torch.inverse(R0).reshape(-1), j0[0].reshape(-1)])
    lbl_rec, body_markers_rec = decode_markers(rec[0, 0], x_in[0, :, 1:-1, 8:-8], piv, stats, post=post, _lib=lib)
    return dict(body_markers_rec=body_markers_rec, contact_lbl_rec=lbl_rec, clip_img_input=x_in, train_mask=(mf == 1),
                clip_img_rec=rec, rot_0_pivot=piv)
