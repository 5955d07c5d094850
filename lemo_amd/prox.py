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
        torch.cuda.current_stream(dev).wait_stream(side)
        return ld
