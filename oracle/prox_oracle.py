"""CPU oracle for the PROX twin of the hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Restates, in plain fp32 CPU PyTorch, one iteration of the PROX sliding-window fit with the loss terms
that are active under ``cfg_files/PROXD_temp_S2.yaml`` / ``S3.yaml``:
  * closure        temp_prox/fitting_temp_slide.py:239-311 (two SMPL-X forwards, backward, erase of the
                   first int(0.15*B) frames' gradients for non-first windows)
  * SMPLifyLoss    :573-616 (2-D joints + priors), :676-680 (cam->world), :685-694 (SDF penetration),
                   :699-739 (friction), :944-992 (infill L1 + contact velocity, S3), :997-1031 (smoothness
                   prior), :1036-1061 (sum + loss_dict)
  * camera         temp_prox/camera.py:88-116 ; priors temp_prox/prior.py:50-90 ; JointMapper misc_utils.py:44-57
  * optimiser      temp_prox/optimizers/optim_factory.py:43-46 (Adam, lr 0.005)
Pinned against the reference ITSELF: tests/golden/make_golden.py imports temp_prox.fitting_temp_slide / camera / prior /
misc_utils / optimizers.optim_factory (module stubs for the absent third-party packages, tests/golden/ref_harness.py)
and drives ``optimizer.step(closure)`` on the same seeded window: 14 loss_dict entries, three gradients and all
parameters after 3 Adam steps agree at 0.0 for S2 / S3 x first / later window (rows ``prox.*`` of
tests/golden/oracle_vs_reference.txt); tests/golden/prox_iter.npz is written from that reference run.
Per-window constants produced at opt_step == 0 by the infilling network (``body_markers_rec``,
``contact_lbl_rec``, :821-941) are inputs here.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import lemo_oracle as O

S2_WEIGHTS = dict(data_weight=1.0, body_pose_weight=4.78e-5, shape_weight=0.0, hand_prior_weight=4.78e-5,
                  expr_prior_weight=0.03, jaw_prior_weight=0.03, sdf_penetration_weight=0.003,
                  motion_prior_smooth_weight=1e8, friction_normal_weight=10.0, friction_tangent_weight=20.0,
                  hand_weight=2.0, face_weight=2.0, motion_infill_rec_weight=0.0, motion_infill_contact_weight=0.0)
S3_WEIGHTS = dict(S2_WEIGHTS, friction_normal_weight=1.0, friction_tangent_weight=1.0,
                  motion_infill_rec_weight=2.0, motion_infill_contact_weight=0.1)
"""cfg_files/PROXD_temp_S{2,3}.yaml; bending_prior_weight = 3.17 * body_pose_weight (fit_temp_loadprox_slide.py:524)."""
PARAM_NAMES = ('global_orient', 'transl', 'left_hand_pose', 'right_hand_pose', 'jaw_pose', 'leye_pose', 'reye_pose',
               'expression')
LOSS_KEYS = ('total_loss', 'joint_loss', 's2m_dist', 'm2s_dist', 'self_penetration_loss', 'sdf_penetration_loss',
             'contact_loss', 'smooth_acc_loss', 'smooth_vel_loss', 'motion_prior_smooth_loss', 'loss_fric_tangent',
             'loss_fric_normal', 'motion_infill_loss', 'motion_infill_contact_loss')


def joint_weights_for(B: int, w: dict) -> torch.Tensor:
    """data_parser_slide.py:238-250 + fit_temp_loadprox_slide.py:526-528."""
    jw = torch.ones(B, 118)
    jw[:, [1, 9, 12]] = 0.0
    jw[:, 25:76] = w['hand_weight']
    jw[:, 76:] = w['face_weight']
    return jw


THRESHOLDS = dict(fric_sdf=0.01, fric_vt=0.0001, fric_vn=0.0, infill_res=0.0, contact=0.1)
"""the selection constants of fitting_temp_slide.py:699-739 (friction: sdf < 0.01, |v_t| > 1e-4, v.n < 0) and :944-992 (infill residual > 0,
contact speed > 0.1).  ``ProxFitOracle.thresholds`` (None = these) lets tests/golden/make_teacher.py measure, in float64, how far each
loss_dict entry JUMPS when an element sits within fp32 rounding of its threshold (a thresholded mean changes by (x - mean) / n when one
element changes sides): the slack the teacher tests add to the 1e-5 loss gate for exactly those entries."""


class ProxFitOracle:
    def __init__(self, smplx: O.SmplxOracle, vposer_w, enc_w, joint_map, ids: Dict[str, np.ndarray], fric_ids,
                 Xmean, Xstd, weights: dict, cam: dict, R, t, sdf, grid_min, grid_max, params: Dict[str, np.ndarray],
                 gt_joints, joints_conf, marker_mask=None, body_markers_rec=None, contact_lbl_rec=None,
                 first_batch_flag: bool = False, lr: float = 0.005):
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        self.smplx, self.vposer_w, self.enc_w = smplx, vposer_w, enc_w
        self.joint_map = torch.from_numpy(np.asarray(joint_map, np.int64))
        self.ids = {k: torch.from_numpy(np.asarray(v, np.int64)) for k, v in ids.items()}
        self.fric_ids = torch.from_numpy(np.asarray(fric_ids, np.int64))
        self.Xmean, self.Xstd = f(np.asarray(Xmean)), f(np.asarray(Xstd))
        self.w, self.cam = dict(weights), dict(cam)
        self.w['bending_prior_weight'] = 3.17 * self.w['body_pose_weight']
        self.R, self.t = f(R), f(t)
        self.sdf, self.grid_min, self.grid_max = f(sdf), f(grid_min), f(grid_max)
        self.betas = f(params['betas'])                                   # fixed (:511)
        self.p = {k: f(params[k]).clone().requires_grad_(True) for k in PARAM_NAMES}
        self.pose_embedding = f(params['pose_embedding']).clone().requires_grad_(True)
        self.gt_joints, self.joints_conf = f(gt_joints), f(joints_conf)
        B = self.betas.shape[0]
        self.joint_weights = joint_weights_for(B, self.w)
        self.marker_mask = None if marker_mask is None else f(marker_mask)
        self.body_markers_rec = None if body_markers_rec is None else f(body_markers_rec)
        self.contact_lbl_rec = None if contact_lbl_rec is None else f(contact_lbl_rec)
        self.first_batch_flag = first_batch_flag
        self.opt = torch.optim.Adam(list(self.p.values()) + [self.pose_embedding], lr=lr)

    # camera.py:88-116 (rotation = I, translation = 0 parameters that are not optimised here)
    def camera(self, points):
        Rc = torch.eye(3).unsqueeze(0).repeat(points.shape[0], 1, 1)
        tc = torch.zeros(points.shape[0], 3)
        T = O.transform_mat(Rc, tc.unsqueeze(-1))
        ph = torch.cat([points, torch.ones(list(points.shape)[:-1] + [1])], dim=-1)
        proj = torch.einsum('bki,bji->bjk', [T, ph])
        img = proj[:, :, :2] / proj[:, :, 2].unsqueeze(-1)
        cm = torch.zeros(points.shape[0], 2, 2)
        cm[:, 0, 0], cm[:, 1, 1] = self.cam['fx'], self.cam['fy']
        return torch.einsum('bki,bji->bjk', [cm, img]) + torch.tensor([self.cam['cx'], self.cam['cy']]).view(1, 1, 2)

    def _body(self, mapped: bool):
        B = self.pose_embedding.shape[0]
        body_pose = O.vposer_decode(self.vposer_w, self.pose_embedding, 'aa').view(B, -1)
        p = self.p
        v, j, fp = self.smplx.forward(self.betas, p['global_orient'], body_pose, p['left_hand_pose'], p['right_hand_pose'],
                                      p['transl'], p['expression'], p['jaw_pose'], p['leye_pose'], p['reye_pose'],
                                      joint_mapper=self.joint_map if mapped else None)
        return v, j, fp

    def _canon(self, joints_world):
        j0 = joints_world[0].detach()
        x_axis = j0[2, :] - j0[1, :]
        x_axis = torch.cat([x_axis[:2], torch.zeros(1)])
        x_axis = x_axis / torch.norm(x_axis)
        z_axis = torch.tensor([0., 0., 1.])
        y_axis = torch.linalg.cross(z_axis, x_axis)
        y_axis = y_axis / torch.norm(y_axis)
        return torch.stack([x_axis, y_axis, z_axis], dim=1)

    @staticmethod
    def _masked_mean_or_zero(x, mask):
        if mask.sum().item() < 1:
            return torch.tensor(0.0)
        return x[mask].abs().mean()

    def loss_dict(self):
        w = self.w
        verts, joints118, full_pose = self._body(True)                   # :248
        _, smplx_joints, _ = self._body(False)                           # :253-258 (second forward)
        zero = torch.tensor(0.0)
        # ---- 2-D keypoints :573-580
        proj = self.camera(joints118)
        wts = (self.joint_weights * self.joints_conf).unsqueeze(-1)
        joint_loss = torch.mean(wts ** 2 * torch.abs(self.gt_joints - proj)) * w['data_weight']
        # ---- priors :586-615
        pprior = self.pose_embedding.pow(2).sum() * w['body_pose_weight'] ** 2
        shape_loss = torch.sum(self.betas ** 2) * w['shape_weight'] ** 2
        body_pose = full_pose[:, 3:66]
        idx = torch.tensor([55, 58, 12, 15]) - 3
        angle = torch.sum(torch.exp(body_pose[:, idx] * torch.tensor([1., -1., -1., -1.]))) * w['bending_prior_weight'] ** 2
        lh45 = torch.einsum('bi,ij->bj', [self.p['left_hand_pose'], self.smplx.lh_comp])
        rh45 = torch.einsum('bi,ij->bj', [self.p['right_hand_pose'], self.smplx.rh_comp])
        lhand = torch.sum(lh45 ** 2) * w['hand_prior_weight'] ** 2
        rhand = torch.sum(rh45 ** 2) * w['hand_prior_weight'] ** 2
        expr = torch.sum(self.p['expression'] ** 2) * w['expr_prior_weight'] ** 2
        jaw = torch.sum((self.p['jaw_pose'] * w['jaw_prior_weight']) ** 2)
        # ---- to world :676-680
        vw = torch.matmul(self.R, verts.permute(0, 2, 1)).permute(0, 2, 1) + self.t
        jw = torch.matmul(self.R, smplx_joints.permute(0, 2, 1)).permute(0, 2, 1) + self.t
        B, nv = vw.shape[0], vw.shape[1]
        # ---- SDF penetration :685-694  (the reference repeats the volume B times; same values)
        norm_v = (vw - self.grid_min) / (self.grid_max - self.grid_min) * 2 - 1
        body_sdf = F.grid_sample(self.sdf[None, None].expand(B, -1, -1, -1, -1), norm_v[:, :, [2, 1, 0]].view(-1, nv, 1, 1, 3),
                                 padding_mode='border', align_corners=False)
        sdf_pen = zero
        if w['sdf_penetration_weight'] > 0 and body_sdf.lt(0).sum().item() >= 1:
            s = body_sdf[body_sdf < 0].unsqueeze(-1).abs()
            sdf_pen = w['sdf_penetration_weight'] * s.pow(2).sum(dim=-1).sqrt().sum()
        # ---- friction :699-739
        fric_t, fric_n = zero, zero
        vf = vw[:, self.fric_ids, :]
        vel = vf[1:] - vf[:-1]
        sdf_f = body_sdf[0:-1, :, self.fric_ids, :, :].squeeze()
        thr = getattr(self, 'thresholds', None) or THRESHOLDS       # (test device: the reference's constants unless a test shifts them)
        sel = torch.where(sdf_f < thr['fric_sdf'])
        if len(sel[0]) > 0:
            n = torch.tensor([0.0, 0.0, 1.0]).repeat(len(sel[0]), 1)
            vc = vel[sel]
            vdn = torch.sum(vc * n, dim=-1)
            vt = vc - vdn.repeat(3, 1).permute(1, 0) * n
            goal_t = torch.norm(vt, dim=-1)
            if (goal_t - thr['fric_vt']).gt(0).sum().item() >= 1:
                fric_t = goal_t[goal_t > thr['fric_vt']].abs().mean() * w['friction_tangent_weight']
            if vdn.lt(thr['fric_vn']).sum().item() >= 1:
                fric_n = vdn[vdn < thr['fric_vn']].abs().mean() * w['friction_normal_weight']
        # ---- infill terms (S3) :944-992
        infill, infill_contact = zero, zero
        if self.body_markers_rec is not None and self.marker_mask.shape[0] * self.marker_mask.shape[1] > self.marker_mask.sum():
            markers = vw[:, self.ids['markers67'], :]
            mw = self.marker_mask.repeat_interleave(3).reshape([self.marker_mask.shape[0], -1, 3])
            T = self.body_markers_rec.shape[0]
            diff = (self.body_markers_rec - markers[0:T]).abs() * (1 - mw[0:T])
            diff = diff[diff > thr['infill_res']]
            infill = w['motion_infill_rec_weight'] * torch.mean(diff)
            vel30 = (vw[1:] - vw[:-1]) * 30
            tot = zero
            for k, name in enumerate(('left_heel', 'right_heel', 'left_toe', 'right_toe')):
                s = torch.norm(vel30[:, self.ids[name], :][self.contact_lbl_rec[:, k] == 1], dim=-1)
                if (s - thr['contact']).gt(0).sum().item() >= 1:
                    tot = tot + s[s > thr['contact']].abs().mean()
            infill_contact = w['motion_infill_contact_weight'] * tot
        # ---- smoothness prior :997-1031
        ms = vw[:, self.ids['markers81'], :]
        R0 = self._canon(jw[:, 0:75])
        ms = torch.matmul(ms - ms[0].detach()[0], R0)
        img = ms.reshape(ms.shape[0], -1).unsqueeze(0)
        img = (img - self.Xmean.view(1, 1, -1)) / self.Xstd
        img = img.permute(0, 2, 1).unsqueeze(1)
        img_v = F.pad(img[:, :, :, 1:] - img[:, :, :, 0:-1], (8, 8, 1, 1), 'reflect')
        z = O.enc_forward(self.enc_w, img_v)
        smooth = torch.mean((z[:, :, :, 1:] - z[:, :, :, 0:-1]) ** 2) * w['motion_prior_smooth_weight']
        total = (joint_loss + pprior + shape_loss + angle + zero + jaw + expr + lhand + rhand + zero + zero + sdf_pen + zero +
                 zero + zero + smooth + fric_t + fric_n + infill + infill_contact)
        return dict(total_loss=total, joint_loss=joint_loss, s2m_dist=zero, m2s_dist=zero, self_penetration_loss=zero,
                    sdf_penetration_loss=sdf_pen, contact_loss=zero, smooth_acc_loss=zero, smooth_vel_loss=zero,
                    motion_prior_smooth_loss=smooth, loss_fric_tangent=fric_t, loss_fric_normal=fric_n,
                    motion_infill_loss=infill, motion_infill_contact_loss=infill_contact)

    def closure(self):
        """fitting_func (:239-311): zero_grad, loss, backward, first-15 % erase."""
        self.opt.zero_grad()
        ld = self.loss_dict()
        ld['total_loss'].backward()
        B = self.pose_embedding.shape[0]
        erase_n = int(B * 0.15)
        if not self.first_batch_flag:
            for v in self.p.values():
                if v.grad is not None:
                    v.grad[0:erase_n, :] = 0
            self.pose_embedding.grad[0:erase_n, :] = 0
        return ld

    def step(self):
        ld = self.closure()
        self.opt.step()
        return {k: float(v) for k, v in ld.items()}


def slide_index_oracle(n_frames: int, batch_size: int):
    """temp_prox/data_parser_slide.py:199-212 restated literally on frame numbers (the reference runs it on image
    paths): returns the concatenated frame list ``img_paths_slide``."""
    img_paths = list(range(n_frames))
    slide_window_size = int(batch_size * 0.7)
    seq_n = (len(img_paths) - batch_size) - slide_window_size
    out = img_paths[0:batch_size]
    for i in range(int(seq_n) + 1):
        start = slide_window_size * (i + 1)
        end = min(start + batch_size, len(img_paths))
        out += img_paths[start:end]
    return out
