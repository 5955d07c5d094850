"""PROX sliding-window fitting iteration (the twin of the AMASS hot path) on the HIP kernels.

Reference: ``temp_prox/fitting_temp_slide.py`` -- closure ``fitting_func`` :239-311 and the parts of
``SMPLifyLoss.forward`` that are live under ``cfg_files/PROXD_temp_S2.yaml`` / ``S3.yaml``:
2-D keypoints :573-580, pose/shape/angle/hand/expression/jaw priors :586-615, cam->world :676-680, SDF
penetration :685-694, friction :699-739, infill L1 + contact velocity :944-992 (S3), smoothness prior
:997-1031, sum + ``loss_dict`` :1036-1061; camera ``temp_prox/camera.py:88-116``; ``JointMapper``
``misc_utils.py:44-57``; Adam lr 0.005 ``optimizers/optim_factory.py:43-46``.

Heavy arithmetic runs in liblemo_hip.so through the same modules the AMASS path uses
(:class:`lemo_amd.body_model.SMPLX`, :class:`lemo_amd.vposer.VPoser`, the smoothness-encoder loss, the SDF
sampler); the small loss algebra on [B,118] / [B,307] tensors is torch glue on the device.  Differences
from the reference that do not change values: SMPL-X is evaluated once (mapped joints are an index_select
of the unmapped ones), the SDF volume is not repeated B times, and every ``.item()`` branch
(:690,:719,:730,:736,:974-987) is a masked mean, so an iteration never syncs with the host.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _hip
from .assets import asset_path
from .body_model import SMPLX
from .priors import Enc
from .scene import sdf_sample
from .vposer import VPoser

S2_WEIGHTS = dict(data_weight=1.0, body_pose_weight=4.78e-5, shape_weight=0.0, hand_prior_weight=4.78e-5,
                  expr_prior_weight=0.03, jaw_prior_weight=0.03, sdf_penetration_weight=0.003,
                  motion_prior_smooth_weight=1e8, friction_normal_weight=10.0, friction_tangent_weight=20.0,
                  hand_weight=2.0, face_weight=2.0, motion_infill_rec_weight=0.0, motion_infill_contact_weight=0.0)
S3_WEIGHTS = dict(S2_WEIGHTS, friction_normal_weight=1.0, friction_tangent_weight=1.0,
                  motion_infill_rec_weight=2.0, motion_infill_contact_weight=0.1)
"""cfg_files/PROXD_temp_S{2,3}.yaml (single stage each)."""
PROX_CAMERA = dict(fx=1060.53, fy=1060.38, cx=951.30, cy=536.77)
"""cfg_files/PROXD_temp_S2.yaml:111-114."""
PARAM_NAMES = ('global_orient', 'transl', 'left_hand_pose', 'right_hand_pose', 'jaw_pose', 'leye_pose', 'reye_pose',
               'expression')
LOSS_KEYS = ('total_loss', 'joint_loss', 's2m_dist', 'm2s_dist', 'self_penetration_loss', 'sdf_penetration_loss',
             'contact_loss', 'smooth_acc_loss', 'smooth_vel_loss', 'motion_prior_smooth_loss', 'loss_fric_tangent',
             'loss_fric_normal', 'motion_infill_loss', 'motion_infill_contact_loss')


def load_prox_tables() -> Dict[str, np.ndarray]:
    """OpenPose(118) <- SMPL-X(127) joint map and the friction vertex set (tools/export_assets.py)."""
    d = np.load(asset_path('prox_tables.npz'))
    return dict(joint_map=d['joint_map'].astype(np.int64), contact_fric_verts_ids=d['contact_fric_verts_ids'].astype(np.int64))


class JointMapper(nn.Module):
    """temp_prox/misc_utils.py:44-57."""

    def __init__(self, joint_maps=None):
        super().__init__()
        if joint_maps is None:
            self.joint_maps = joint_maps
        else:
            self.register_buffer('joint_maps', torch.as_tensor(np.asarray(joint_maps), dtype=torch.long))

    def forward(self, joints, **kwargs):
        return joints if self.joint_maps is None else torch.index_select(joints, 1, self.joint_maps)


def joint_weights_for(B: int, w: dict, device) -> torch.Tensor:
    """data_parser_slide.py:238-250 (joints 1, 9, 12 ignored) + fit_temp_loadprox_slide.py:526-528."""
    jw = torch.ones(B, 118, device=device)
    jw[:, [1, 9, 12]] = 0.0
    jw[:, 25:76] = w['hand_weight']
    jw[:, 76:] = w['face_weight']
    return jw


def _masked_mean(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """``x[mask].mean()`` that is exactly 0 for an empty selection, without a host sync."""
    m = mask.to(x.dtype)
    return (x * m).sum() / m.sum().clamp(min=1.0)


class ProxTemporalFitter:
    """One sliding window (B frames) of the PROX temporal fit."""

    def __init__(self, body_model: SMPLX, vposer: VPoser, smooth_encoder: Enc, ids: Dict[str, np.ndarray],
                 Xmean, Xstd, weights: dict, R, t, sdf: torch.Tensor, grid_min, grid_max, params: Dict[str, np.ndarray],
                 gt_joints, joints_conf, joint_map=None, fric_ids=None, cam: Optional[dict] = None, marker_mask=None,
                 body_markers_rec=None, contact_lbl_rec=None, first_batch_flag: bool = False, lr: float = 0.005):
        dev = sdf.device
        self.device = dev
        f = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)
        li = lambda a: torch.as_tensor(np.asarray(a, np.int64), device=dev)
        tables = load_prox_tables()
        self.body_model, self.vposer, self.enc = body_model, vposer, smooth_encoder
        self.joint_map = li(tables['joint_map'] if joint_map is None else joint_map)
        self.fric_ids = li(tables['contact_fric_verts_ids'] if fric_ids is None else fric_ids)
        self.ids = {k: li(v) for k, v in ids.items()}
        self.Xmean, self.Xstd = f(np.asarray(Xmean)).view(1, 1, -1), f(np.asarray(Xstd)).view(-1)
        self.w = dict(weights)
        self.w['bending_prior_weight'] = 3.17 * self.w['body_pose_weight']
        self.cam = dict(PROX_CAMERA if cam is None else cam)
        self.R, self.t = f(R), f(t)
        self.sdf = sdf.contiguous().float()
        self.grid_min, self.grid_max = [float(v) for v in grid_min], [float(v) for v in grid_max]
        # parameters live in the smplx-compatible module, like body_model.reset_params(**prox_params) (:499)
        self.body_model.reset_params(**{k: params[k] for k in params if k != 'pose_embedding'})
        self.body_model.betas.requires_grad_(False)                       # fit_temp_loadprox_slide.py:511
        self.pose_embedding = f(params['pose_embedding']).clone().requires_grad_(True)
        self.gt_joints, self.joints_conf = f(gt_joints), f(joints_conf)
        B = self.pose_embedding.shape[0]
        self.B = B
        self.joint_weights = joint_weights_for(B, self.w, dev)
        self.marker_mask = None if marker_mask is None else f(marker_mask)
        self.body_markers_rec = None if body_markers_rec is None else f(body_markers_rec)
        self.contact_lbl_rec = None if contact_lbl_rec is None else f(contact_lbl_rec)
        self.first_batch_flag = first_batch_flag
        self.params = [p for n, p in self.body_model.named_parameters() if p.requires_grad] + [self.pose_embedding]
        # optim_factory.py:43-46.  capturable: the step count lives on the device so that a captured step can be replayed
        self.optimizer = torch.optim.Adam(self.params, lr=lr, capturable=self.pose_embedding.is_cuda)
        self._comp = None
        # small constants as device tensors, made once: a host -> device tensor construction inside an iteration is a
        # synchronous copy (and illegal while the iteration is being captured)
        self._cam_f = torch.tensor([self.cam['fx'], self.cam['fy']], device=dev).view(1, 1, 2)
        self._cam_c = torch.tensor([self.cam['cx'], self.cam['cy']], device=dev).view(1, 1, 2)
        self._angle_idx = torch.tensor([55, 58, 12, 15], device=dev) - 3
        self._angle_sgn = torch.tensor([1., -1., -1., -1.], device=dev)
        self._z_axis = torch.tensor([0., 0., 1.], device=dev)

    # temp_prox/camera.py:88-116 with the fixed identity camera pose of the PROX configs
    def camera(self, points: torch.Tensor) -> torch.Tensor:
        xy = points[:, :, :2] / points[:, :, 2:3]
        f, c = self._cam_f, self._cam_c
        return xy * f + c

    def loss_dict(self) -> Dict[str, torch.Tensor]:
        w, bm = self.w, self.body_model
        B = self.B
        body_pose = self.vposer.decode(self.pose_embedding, output_type='aa').view(B, -1)        # :243
        jm, bm.joint_mapper = bm.joint_mapper, None
        out = bm(return_verts=True, body_pose=body_pose, return_full_pose=True)                 # :248 / :253-258 in one pass
        bm.joint_mapper = jm
        verts, smplx_joints = out.vertices, out.joints
        joints118 = torch.index_select(smplx_joints, 1, self.joint_map)
        zero = torch.zeros((), device=self.device)
        # ---- 2-D keypoints
        wts = (self.joint_weights * self.joints_conf).unsqueeze(-1)
        joint_loss = torch.mean(wts ** 2 * torch.abs(self.gt_joints - self.camera(joints118))) * w['data_weight']
        # ---- priors
        pprior = self.pose_embedding.pow(2).sum() * w['body_pose_weight'] ** 2
        shape_loss = torch.sum(out.betas ** 2) * w['shape_weight'] ** 2
        idx, sgn = self._angle_idx, self._angle_sgn
        angle = torch.sum(torch.exp(out.full_pose[:, 3:66][:, idx] * sgn)) * w['bending_prior_weight'] ** 2
        lhand = torch.sum(out.left_hand_pose ** 2) * w['hand_prior_weight'] ** 2
        rhand = torch.sum(out.right_hand_pose ** 2) * w['hand_prior_weight'] ** 2
        expr = torch.sum(out.expression ** 2) * w['expr_prior_weight'] ** 2
        jaw = torch.sum((out.jaw_pose * w['jaw_prior_weight']) ** 2)
        # ---- to world
        vw = torch.matmul(verts, self.R.t()) + self.t
        jw = torch.matmul(smplx_joints, self.R.t()) + self.t
        # ---- SDF penetration + friction (one lookup of the volume for both)
        body_sdf = sdf_sample(vw, self.sdf, self.grid_min, self.grid_max, _lib=self.body_model._lib_override)
        neg = body_sdf < 0
        sdf_pen = w['sdf_penetration_weight'] * (body_sdf.abs() * neg.to(body_sdf.dtype)).sum() \
            if w['sdf_penetration_weight'] > 0 else zero
        vf = vw[:, self.fric_ids, :]
        vel = vf[1:] - vf[:-1]
        contact = body_sdf[0:-1][:, self.fric_ids] < 0.01
        vdn = vel[..., 2]                                                   # n = (0,0,1)
        goal_t = torch.norm(vel[..., :2], dim=-1)                          # |v - (v.n) n|
        fric_t = _masked_mean(goal_t, contact & (goal_t - 0.0001 > 0)) * w['friction_tangent_weight']
        fric_n = _masked_mean(vdn.abs(), contact & (vdn < 0)) * w['friction_normal_weight']
        # ---- infill terms (S3)
        infill, infill_contact = zero, zero
        if self.body_markers_rec is not None:
            markers = vw[:, self.ids['markers67'], :]
            mw = self.marker_mask.repeat_interleave(3).reshape([self.marker_mask.shape[0], -1, 3])
            T = self.body_markers_rec.shape[0]
            diff = (self.body_markers_rec - markers[0:T]).abs() * (1 - mw[0:T])
            occluded = (self.marker_mask.numel() > self.marker_mask.sum()).to(diff.dtype)   # device-side flag (:944)
            infill = w['motion_infill_rec_weight'] * _masked_mean(diff, diff > 0) * occluded
            vel30 = (vw[1:] - vw[:-1]) * 30
            tot = zero
            for k, name in enumerate(('left_heel', 'right_heel', 'left_toe', 'right_toe')):
                s = torch.norm(vel30[:, self.ids[name], :], dim=-1)
                sel = (self.contact_lbl_rec[:, k] == 1).unsqueeze(-1) & (s - 0.1 > 0)
                tot = tot + _masked_mean(s, sel)
            infill_contact = w['motion_infill_contact_weight'] * tot * occluded
        # ---- smoothness prior
        ms = vw[:, self.ids['markers81'], :]
        j0 = jw[0].detach()
        x_axis = j0[2] - j0[1]
        x_axis = torch.cat([x_axis[:2], torch.zeros(1, device=self.device)])
        x_axis = x_axis / torch.norm(x_axis)
        z_axis = self._z_axis
        y_axis = torch.linalg.cross(z_axis, x_axis)
        y_axis = y_axis / torch.norm(y_axis)
        R0 = torch.stack([x_axis, y_axis, z_axis], dim=1)
        ms = torch.matmul(ms - ms[0].detach()[0], R0)
        img = (ms.reshape(B, -1).unsqueeze(0) - self.Xmean) / self.Xstd
        img = img.permute(0, 2, 1).unsqueeze(1)
        img_v = torch.nn.functional.pad(img[:, :, :, 1:] - img[:, :, :, 0:-1], (8, 8, 1, 1), 'reflect')
        smooth = self.enc.smooth_loss(img_v) * w['motion_prior_smooth_weight']
        total = (joint_loss + pprior + shape_loss + angle + jaw + expr + lhand + rhand + sdf_pen + smooth + fric_t + fric_n +
                 infill + infill_contact)
        return dict(total_loss=total, joint_loss=joint_loss, s2m_dist=zero, m2s_dist=zero, self_penetration_loss=zero,
                    sdf_penetration_loss=sdf_pen, contact_loss=zero, smooth_acc_loss=zero, smooth_vel_loss=zero,
                    motion_prior_smooth_loss=smooth, loss_fric_tangent=fric_t, loss_fric_normal=fric_n,
                    motion_infill_loss=infill, motion_infill_contact_loss=infill_contact)

    def closure(self) -> Dict[str, torch.Tensor]:
        """``fitting_func`` (:239-311): zero_grad, loss, backward, erase grads of the first int(0.15 B) frames
        unless this is the first window."""
        self.optimizer.zero_grad()
        ld = self.loss_dict()
        ld['total_loss'].backward()
        if not self.first_batch_flag:
            erase_n = int(self.B * 0.15)
            for p in self.params:
                if p.grad is not None:
                    p.grad[0:erase_n, :] = 0
        return ld

    def step(self, n: int = 1, use_graph: Optional[bool] = None) -> Dict[str, torch.Tensor]:
        """n Adam iterations; returns the ``loss_dict`` of the last one (detached).

        ``use_graph`` (default: on a HIP device when n > 4): an iteration is ~60 HIP kernels plus ~250 small torch ops
        of loss algebra and is host-launch bound when issued one by one.  It is captured once (raw stream capture,
        ``lemo_capture_*``; three eager iterations first warm the allocator) and replayed n - 3 times on a private
        stream, inside this call.  The returned tensors are the captured iteration's outputs after the last replay."""
        lib = _hip.get_lib() if self.pose_embedding.is_cuda else None
        if use_graph is None:
            use_graph = lib is not None and n > 4
        use_graph = bool(use_graph) and n > 3              # the captured path runs 3 eager iterations first

        def one():
            ld = self.closure()
            self.optimizer.step()
            return {k: v.detach() for k, v in ld.items()}       # no autograd graph survives the iteration

        if not use_graph:
            ld = None
            for _ in range(n):
                ld = one()
            return ld
        import ctypes as C
        dev = self.pose_embedding.device
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                ld = one()
            del ld
            self.optimizer.zero_grad()
            # everything the captured iteration allocates comes from a PRIVATE pool that lives as long as the graph:
            # the replays write into those addresses, which the shared caching allocator could otherwise hand to
            # another allocation (another thread, an empty_cache()) between two replays
            pool = torch.cuda.MemPool()
            with torch.cuda.use_mem_pool(pool):
                lib.check(lib.capture_begin(side.cuda_stream), 'capture_begin')
                try:
                    ld = one()
                finally:
                    exe = C.c_void_p()
                    rc = lib.capture_end(side.cuda_stream, C.byref(exe))
            lib.check(rc, 'capture_end')
            try:
                for _ in range(n - 3):                           # capture records the iteration without running it
                    lib.check(lib.graph_launch(exe, side.cuda_stream), 'graph_launch')
                side.synchronize()
            finally:
                lib.check(lib.graph_destroy(exe), 'graph_destroy')
            ld = {k: v.clone() for k, v in ld.items()}
            self._capture_pool = pool        # the parameters' .grad tensors of the captured iteration live in it
        torch.cuda.current_stream(dev).wait_stream(side)
        return ld


# ----------------------------------------------------------------------------------------------------------------------
# Native engine: the same iteration as ``ProxTemporalFitter`` as ONE C call (lemo_prox_step) replaying hipGraphs
# ----------------------------------------------------------------------------------------------------------------------
WEIGHT_ORDER = ('data_weight', 'body_pose_weight', 'shape_weight', 'bending_prior_weight', 'hand_prior_weight',
                'expr_prior_weight', 'jaw_prior_weight', 'sdf_penetration_weight', 'motion_prior_smooth_weight',
                'friction_normal_weight', 'friction_tangent_weight', 'motion_infill_rec_weight', 'motion_infill_contact_weight')
"""order of ``lemo_prox_desc.weights`` (include/lemo_hip.h)."""
ENGINE_PARAMS = (('global_orient', 3), ('transl', 3), ('left_hand_pose', 12), ('right_hand_pose', 12), ('jaw_pose', 3),
                 ('leye_pose', 3), ('reye_pose', 3), ('expression', 10), ('pose_embedding', 32))
"""the optimised tensors in the engine's Adam order: ``body_model.parameters()`` with requires_grad + pose_embedding
(fit_temp_loadprox_slide.py:511-519)."""


class ProxWindowEngine(_hip.StreamOrdered):
    """One sliding window (B frames) of the PROX temporal fit on the native engine (``lemo_prox_*``): closure
    ``fitting_func`` (fitting_temp_slide.py:239-311), the S2 / S3-active ``SMPLifyLoss`` terms, backward, first-15 % erase
    and Adam (lr 0.005) are a fixed sequence of ~40 HIP kernels captured once and replayed -- no torch op, no host sync,
    no atomics-ordered accumulation (graph replay == eager launches bit for bit).  Same constructor as
    :class:`ProxTemporalFitter` (which stays as the module-level / autograd composition of the same kernels)."""

    def __init__(self, body_model: SMPLX, vposer: VPoser, smooth_encoder: Enc, ids: Dict[str, np.ndarray], Xmean, Xstd,
                 weights: dict, R, t, sdf: torch.Tensor, grid_min, grid_max, params: Dict[str, np.ndarray], gt_joints,
                 joints_conf, joint_map=None, fric_ids=None, cam: Optional[dict] = None, marker_mask=None,
                 body_markers_rec=None, contact_lbl_rec=None, first_batch_flag: bool = False, lr: float = 0.005,
                 conv_variant: Optional[int] = None):
        import ctypes as C
        from ._hip import ptr
        from .body_model import K_PAD, alloc_pose_ws
        from .priors import DEFAULT_CONV_VARIANT, ENC_CHANNELS, EncWeights, cg8p_alloc
        from .vposer import vposer_weight_struct
        self.lib = lib = body_model._lib_override or _hip.get_lib()
        dev = sdf.device
        self.device = dev
        _hip.check_device(lib, sdf)
        data = body_model.data
        assert data.ncomp == 12 and data.use_pca, 'the PROX configs fit 12 PCA coefficients per hand'
        self.data, self.dbody = data, body_model._device_body(dev)
        B = int(np.asarray(params['pose_embedding']).shape[0])
        self.B, V, nj = B, data.V, data.nj
        tables = load_prox_tables()
        ti = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.int32)).to(dev)
        tf = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        jm = np.asarray(tables['joint_map'] if joint_map is None else joint_map, np.int64)
        fric = np.asarray(tables['contact_fric_verts_ids'] if fric_ids is None else fric_ids, np.int64)
        assert len(set(fric.tolist())) == len(fric), 'duplicate friction vertex ids'
        m67, m81 = np.asarray(ids['markers67'], np.int64), np.asarray(ids['markers81'], np.int64)
        foot = [np.asarray(ids[k], np.int64) for k in ('left_heel', 'right_heel', 'left_toe', 'right_toe')]
        for f in foot:
            assert len(set(f.tolist())) == len(f), 'duplicate vertex id in a foot set'
        n_sj = data.n_joints_out
        nvj = n_sj - nj
        # inverse of the joint map (index_select backward as a gather)
        jl = [np.nonzero(jm == j)[0] for j in range(n_sj)]
        jm_start = np.cumsum([0] + [len(x) for x in jl])
        # special vertex set S
        extra, lmk_rows, lmk_bary = data.extra_ids.astype(np.int64), data.lmk_rows.astype(np.int64), data.lmk_bary
        S = np.unique(np.concatenate([fric, m67, m81] + foot + [extra, lmk_rows.reshape(-1)]))
        pos = {int(v): i for i, v in enumerate(S)}
        n_s = len(S)
        s_m67, s_m81, s_fric = (-np.ones(n_s, np.int32) for _ in range(3))
        s_mask = np.zeros(n_s, np.int32)
        for i, v in enumerate(m67): s_m67[pos[int(v)]] = i
        for i, v in enumerate(m81): s_m81[pos[int(v)]] = i
        for i, v in enumerate(fric): s_fric[pos[int(v)]] = i
        for k, f in enumerate(foot):
            for v in f: s_mask[pos[int(v)]] |= (1 << k)
        jlists = [[] for _ in range(n_s)]
        for i, v in enumerate(extra): jlists[pos[int(v)]].append((i, 1.0))
        for l in range(lmk_rows.shape[0]):
            for f in range(3): jlists[pos[int(lmk_rows[l, f])]].append((len(extra) + l, float(lmk_bary[l, f])))
        s_jstart = np.cumsum([0] + [len(x) for x in jlists])
        s_jidx = np.asarray([q[0] for x in jlists for q in x] + [0], np.int32)
        s_jw = np.asarray([q[1] for x in jlists for q in x] + [0.0], np.float32)
        T_ = self._t = dict(joint_map=ti(jm), jm_start=ti(jm_start), jm_list=ti(np.concatenate(jl + [np.zeros(1, np.int64)])),
                            s_vid=ti(S), s_m67=ti(s_m67), s_m81=ti(s_m81), s_mask=ti(s_mask), s_fric=ti(s_fric),
                            s_jstart=ti(s_jstart), s_jidx=ti(s_jidx), s_jw=tf(s_jw), fric=ti(fric), m67=ti(m67),
                            foot_start=ti(np.cumsum([0] + [len(f) for f in foot])), foot_vid=ti(np.concatenate(foot)),
                            row81=ti(m81), Xstd=tf(np.asarray(Xstd).reshape(-1)), Xmean=tf(np.asarray(Xmean).reshape(-1)))
        dt = self.dbody.t
        pc = _hip.ProxConst(len(jm), n_sj, ptr(T_['joint_map']), ptr(T_['jm_start']), ptr(T_['jm_list']), len(extra), lmk_rows.shape[0],
                            ptr(dt['extra_ids']), ptr(dt['lmk_rows']), ptr(dt['lmk_bary']), n_s, ptr(T_['s_vid']), ptr(T_['s_m67']),
                            ptr(T_['s_m81']), ptr(T_['s_mask']), ptr(T_['s_fric']), ptr(T_['s_jstart']), ptr(T_['s_jidx']), ptr(T_['s_jw']),
                            len(fric), ptr(T_['fric']), len(m67), ptr(T_['m67']), ptr(T_['foot_start']), ptr(T_['foot_vid']))
        Rn, tn = np.asarray(R, np.float32).reshape(3, 3), np.asarray(t, np.float32).reshape(3)
        c2w = np.concatenate([Rn.reshape(-1), tn]).astype(np.float32)
        T_['c2w'] = tf(c2w)
        n81 = len(m81)
        assert T_['Xstd'].numel() == 3 * n81
        fitc = _hip.FitConst(0, len(m67), n81, None, ptr(T_['row81']), None, None, None, None, None, None, ptr(T_['Xstd']),
                             ptr(T_['Xmean']), ptr(T_['c2w']))
        # weights
        w = dict(weights)
        w['bending_prior_weight'] = 3.17 * w['body_pose_weight']                    # fit_temp_loadprox_slide.py:524
        self.w = w
        wl = [float(w[k]) for k in WEIGHT_ORDER]
        T_['weights'] = tf(wl)
        # model weights
        vp_sd = {k: v.detach().cpu().numpy() for k, v in vposer.state_dict().items() if k.startswith('bodyprior_dec_')}
        self.vposer_struct, self._vp_t = vposer_weight_struct(vp_sd, dev)
        self.enc = EncWeights({k: v.detach().cpu().numpy() for k, v in smooth_encoder.state_dict().items()}, dev)
        # window data
        cam = dict(PROX_CAMERA if cam is None else cam)
        jw = joint_weights_for(B, w, dev)
        T_['w2'] = ((jw * tf(joints_conf)) ** 2).contiguous()
        T_['gt'] = tf(gt_joints)
        self.use_infill = body_markers_rec is not None and bool(np.asarray(marker_mask).size > np.asarray(marker_mask).sum())
        if body_markers_rec is not None:
            T_['mask'], T_['rec'], T_['clbl'] = tf(marker_mask), tf(body_markers_rec), tf(contact_lbl_rec)
            assert T_['rec'].shape[0] == B - 1 and T_['clbl'].shape == (B - 1, 4), 'body_markers_rec / contact_lbl_rec carry B - 1 frames'
        # parameters + Adam state
        self.P = {k: tf(np.asarray(params[k], np.float32).reshape(B, d)) for k, d in ENGINE_PARAMS}
        self.betas = tf(np.asarray(params['betas'], np.float32).reshape(B, 10))
        self.adam_m, self.adam_v = z(B, 81), z(B, 81)
        self.step_ctr = torch.zeros(1, dtype=torch.int32, device=dev)
        self.step_cur = torch.zeros(1, dtype=torch.int32, device=dev)
        self.nonfinite = torch.zeros(2, dtype=torch.int32, device=dev)
        # workspace
        self.H, self.W = 3 * n81 + 2, B - 1 + 16
        H, W = self.H, self.W
        pose_ws, self._pose_t, Bp = alloc_pose_ws(B, nj, dev, self.dbody.blend_f16)
        uset, self._uset_t = self.dbody.vertex_set('all', np.arange(V), frames=B)
        self.ws = dict(h1=z(B, 512), h2=z(B, 512), vo=z(B, 128), vp_scratch=z(B, 1152), verts=z(B, V, 3), v_posed=z(B, V, 3),
                       dverts=z(B, V, 3), x0=z((H + 2) * (W + 2)), canon=z(12), dx0=z(H * W), dJtr=z(B, nj, 3), dJv=z(B, nvj, 3),
                       dtr_j=z(B, 3), gp=z(B, 81), dfp_add=z(B, nj * 3), dvp=z(B, uset.NCs), dA=z(B, nj, 12), dtr_v=z(B, 3),
                       dX=z(B, K_PAD), g_go=z(B, 3), g_lh=z(B, 12), g_rh=z(B, 12), g_jaw=z(B, 3), g_leye=z(B, 3), g_reye=z(B, 3),
                       g_expr=z(B, 10), g_pe=z(B, 32), losses=z(16))
        self.loss_acc = torch.zeros(32 * 32 + 32 * 16, dtype=torch.float64, device=dev)
        self.act = [None] + [cg8p_alloc(ENC_CHANNELS[l], H, W, dev) for l in range(1, 11)]
        self.dact = [cg8p_alloc(64, H, W, dev), cg8p_alloc(64, H, W, dev)]
        self.sdf = sdf.contiguous().float()
        d = _hip.ProxDesc()
        d.B, d.Bp, d.V = B, Bp, V
        cv = DEFAULT_CONV_VARIANT if conv_variant is None else int(conv_variant)
        if cv in (2, 3, 4) and 127 + 2 * (127 // W + 1) + 2 * (W + 2) + 3 > 416:
            cv = 1                                 # (variant 5 stays for wide windows: the fused pairs take any width, fitting.py)
        from .priors import check_conv_variant
        self.conv_variant = d.conv_variant = check_conv_variant(cv)
        d.first_batch_flag, d.use_infill, d.T = int(bool(first_batch_flag)), int(self.use_infill), B - 1
        d.vposer, d.body, d.skin, d.uset, d.fit, d.pc = self.vposer_struct, self.dbody.body, self.dbody.skin, uset, fitc, pc
        for i, c in enumerate(ENC_CHANNELS): d.enc_ch[i] = c
        for l in range(10):
            d.enc_w[l], d.enc_b[l], d.enc_wbwd[l] = ptr(self.enc.w[l]), ptr(self.enc.b[l]), ptr(self.enc.wbwd[l])
            d.enc_w2[l], d.enc_wbwd2[l] = ptr(self.enc.w2[l]), ptr(self.enc.wbwd2[l])
            for bwd, dst, dinv in ((False, d.enc_w3, d.enc_w3_inv), (True, d.enc_wbwd3, d.enc_wbwd3_inv)):
                pack, winv = self.enc.split_pack(l, bwd, cv)                         # bf16 x 3 (variant 3) or f16 x 2 (variant 4)
                dst[l], dinv[l] = (ptr(pack) if pack is not None else None), float(winv)
        d.sdf = ptr(self.sdf)
        for i in range(3):
            d.sdf_dim[i], d.grid_min[i], d.grid_max[i] = int(self.sdf.shape[i]), float(grid_min[i]), float(grid_max[i])
        for i in range(12): d.cam2world[i] = float(c2w[i])
        for i, k in enumerate(('fx', 'fy', 'cx', 'cy')): d.cam[i] = float(cam[k])
        d.gt_joints, d.w2 = ptr(T_['gt']), ptr(T_['w2'])
        if body_markers_rec is not None:
            d.marker_mask, d.body_markers_rec, d.contact_lbl_rec = ptr(T_['mask']), ptr(T_['rec']), ptr(T_['clbl'])
        d.weights = ptr(T_['weights'])
        for i, v in enumerate(wl): d.weights_host[i] = v
        for k, _ in ENGINE_PARAMS: setattr(d, k, ptr(self.P[k]))
        d.betas, d.adam_m, d.adam_v = ptr(self.betas), ptr(self.adam_m), ptr(self.adam_v)
        d.step_ctr, d.step_cur, d.nonfinite, d.lr = ptr(self.step_ctr), ptr(self.step_cur), ptr(self.nonfinite), float(lr)
        for k in ('h1', 'h2', 'vo', 'vp_scratch', 'verts', 'v_posed', 'dverts', 'x0', 'canon', 'dx0', 'dJtr', 'dJv', 'dtr_j', 'gp',
                  'dfp_add', 'dvp', 'dA', 'dtr_v', 'dX', 'g_go', 'g_lh', 'g_rh', 'g_jaw', 'g_leye', 'g_reye', 'g_expr', 'g_pe', 'losses'):
            setattr(d, k, ptr(self.ws[k]))
        d.pose = pose_ws
        for l in range(1, 11): d.act[l] = ptr(self.act[l])
        d.dact[0], d.dact[1] = ptr(self.dact[0]), ptr(self.dact[1])
        d.loss_acc = ptr(self.loss_acc)
        self.desc = d
        self.handle = lib.prox_create(C.byref(d))
        if not self.handle:
            raise _hip.LemoHipError('lemo_prox_create rejected the descriptor')
        self.first_batch_flag = bool(first_batch_flag)
        self._init_order(self.device, lib)
        self._after_write()          # the parameter copies above were enqueued on the constructor's current stream

    def __del__(self):
        h, self.handle = getattr(self, 'handle', None), None
        if h:
            # The engine's buffers are torch tensors allocated on the default stream but written by graph replays on whatever
            # stream step() ran on.  When the last reference goes, the caching allocator may hand those blocks to the next
            # default-stream allocation at once -- while a replay is still in flight they would be written from two places.
            lib, rel = self.lib, getattr(_hip, 'release', None) if _hip is not None else None
            if rel is None:                  # interpreter shutdown: module globals are gone, the process is about to exit
                return
            rel(self.device, lib, lambda: lib.prox_destroy(h), getattr(self, '_run_ev', None))

    def _s(self):
        return None if self.lib.is_emu else torch.cuda.current_stream(self.device).cuda_stream

    def closure(self) -> Dict[str, float]:
        """forward + backward (no update, no erase): fills the loss record and the gradient buffers"""
        self._before_run()
        self.lib.check(self.lib.prox_closure(self.handle, self._s()), 'prox_closure')
        self._after_run()
        return self.loss_dict()

    def step(self, n: int = 1, use_graph: bool = True) -> None:
        """n x ``optimizer.step(closure)``; asynchronous.  Graph capture needs a non-default current stream."""
        self._before_run()
        self.lib.check(self.lib.prox_step(self.handle, int(n), int(bool(use_graph) and not self.lib.is_emu), self._s()), 'prox_step')
        self._after_run()

    # -- optimiser state in / out (C ABI lemo_prox_load_state / lemo_prox_save_state) ---------------------------
    def _state_struct(self, t: Dict[str, torch.Tensor]):
        from ._hip import ptr
        st = _hip.ProxState()
        for k, _ in ENGINE_PARAMS:
            setattr(st, k, ptr(t[k]))
        st.adam_m, st.adam_v, st.step = ptr(t['adam_m']), ptr(t['adam_v']), ptr(t['step'])
        return st

    @torch.no_grad()
    def save_state(self) -> Dict[str, torch.Tensor]:
        """the nine optimised tensors, Adam's ``exp_avg`` / ``exp_avg_sq`` as [B,81] blocks in ``ENGINE_PARAMS`` order and
        ``step`` = completed ``optimizer.step(closure)`` calls (device tensors, copies)"""
        import ctypes as C
        t = {k: torch.empty(self.B, d, dtype=torch.float32, device=self.device) for k, d in ENGINE_PARAMS}
        t['adam_m'], t['adam_v'] = (torch.empty(self.B, 81, dtype=torch.float32, device=self.device) for _ in range(2))
        t['step'] = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._before_run()
        self.lib.check(self.lib.prox_save_state(self.handle, C.byref(self._state_struct(t)), self._s()), 'prox_save_state')
        self._after_run()
        return t

    @torch.no_grad()
    def load_state(self, state: Dict) -> None:
        """continue from ``state`` (see :meth:`save_state`; numpy or tensors; ``adam_m`` / ``adam_v`` / ``step`` default to a
        fresh optimiser -- what a new window starts with, data_parser_slide.py:326-331 + fit_temp_loadprox_slide.py:511-519)"""
        import ctypes as C
        td = lambda a, w: (a.detach() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, np.float32))
                           ).to(self.device, torch.float32).reshape(self.B, w).contiguous()
        t = {k: td(state[k], d) for k, d in ENGINE_PARAMS}
        for k in ('adam_m', 'adam_v'):
            t[k] = td(state[k], 81) if state.get(k) is not None else torch.zeros(self.B, 81, dtype=torch.float32, device=self.device)
        sv = state.get('step', 0)
        t['step'] = (sv.detach().to(self.device, torch.int32).reshape(1) if isinstance(sv, torch.Tensor)
                     else torch.full((1,), int(sv), dtype=torch.int32, device=self.device))
        self._before_run()
        self.lib.check(self.lib.prox_load_state(self.handle, C.byref(self._state_struct(t)), self._s()), 'prox_load_state')
        self._after_run()
        self._keep = t

    def loss_dict(self) -> Dict[str, float]:
        self._before_read()
        v = self.ws['losses'].detach().cpu().numpy()
        return {k: float(v[i]) for i, k in enumerate(LOSS_KEYS)}

    def grads(self, erase: bool = True) -> Dict[str, torch.Tensor]:
        """d(total_loss)/d(parameter) of the last closure including the priors' own terms, with the first-15 % erase
        applied like the reference's closure does (:282-289)"""
        self._before_read()
        g = {'global_orient': self.ws['g_go'], 'transl': self.ws['dtr_v'] + self.ws['dtr_j'], 'left_hand_pose': self.ws['g_lh'],
             'right_hand_pose': self.ws['g_rh'], 'jaw_pose': self.ws['g_jaw'], 'leye_pose': self.ws['g_leye'],
             'reye_pose': self.ws['g_reye'], 'expression': self.ws['g_expr'], 'pose_embedding': self.ws['g_pe']}
        out, o = {}, 0
        for k, dim in ENGINE_PARAMS:
            v = g[k] + self.ws['gp'][:, o:o + dim]
            if erase and not self.first_batch_flag:
                v = v.clone()
                v[:int(self.B * 0.15)] = 0
            out[k] = v
            o += dim
        return out

    def nonfinite_step(self) -> int:
        """1-based index of the first iteration with a NaN / Inf total loss (0 = none): FittingMonitor.run_fitting's stop
        (fitting_temp_slide.py:198-204) inside the replayed graph -- later iterations skip their update"""
        self._before_read()
        return int(self.nonfinite[0].item())

    def write_back(self, body_model: SMPLX) -> torch.Tensor:
        """copy the fitted parameters into the smplx-compatible module (what fit_temp_loadprox_slide.py:577-594 pickles);
        returns pose_embedding"""
        self._before_read()
        with torch.no_grad():
            for k, _ in ENGINE_PARAMS[:-1]:
                getattr(body_model, k).copy_(self.P[k])
        return self.P['pose_embedding']
