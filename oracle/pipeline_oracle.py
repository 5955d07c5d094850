"""CPU oracle for the per-clip / per-window pipeline AROUND the fitting loop  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Restates in plain CPU PyTorch / numpy (pinned against the reference's own text by tests/golden/make_golden.py,
rows ``amass_clip.*`` / ``prox_setup.*`` of tests/golden/oracle_vs_reference.txt):

  AMASS  opt_amass_temp.py:159-214  input masking (22 upper-body markers x 3 rows + the 4 contact rows), reflect pad,
                                    60 x [AE forward, L1 on the un-masked rows, backward, Adam 3e-6], eval forward
         opt_amass_temp.py:256-329  sigmoid -> contact labels, de-normalise, reorder, reconstruct_global_body
         opt_amass_temp.py:332-458  the 100-step fit (lemo_oracle.AmassFitOracle) and the saved [T,72] block
  PROX   fitting_temp_slide.py:776-941  the ``opt_step == 0`` block of SMPLifyLoss.forward: canonical frame, contact
                                    labels from marker velocity / height, 4-channel image, marker-mask input masking,
                                    finetune, decode back to PROX world coordinates
  per-frame  opt_amass_perframe.py:293-355  (BASELINE configs[0]) B independent single-frame fits
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import lemo_oracle as O
from . import markers_oracle as MO

MASK_MARKER_IDS = np.array([14, 15, 18, 19, 29, 2, 20, 21, 30, 25, 16, 45, 46, 48, 49, 59, 32, 50, 51, 55, 60, 47])
"""opt_amass_temp.py:171-172: the upper-body markers hidden from the infilling network."""
P2D = (8, 8, 1, 1)


def amass_mask_rows() -> np.ndarray:
    """rows of the un-padded [d = 208] image that are zeroed (:173-183, body_mode local_markers_4chan)"""
    r1 = MASK_MARKER_IDS * 3 + 3
    return np.concatenate([r1, r1 + 1, r1 + 2])


def amass_mask_input(clip_img: torch.Tensor):
    """clip_img [1,4,d,T] (normalised) -> (clip_img_input [1,4,d+2,T+16], train_mask bool [d+2,T+16]).
    train_mask selects ``res_map[:, upper_body_row][:, 0:-5]`` (:199-204)."""
    x = clip_img.clone()
    rows = amass_mask_rows()
    x[:, 0, rows, :] = 0.
    x[:, 0, -4:, :] = 0.
    x = F.pad(x, P2D, 'reflect')
    H, W = x.shape[-2], x.shape[-1]
    upper = sorted(set(range(H)) - set((rows + 1).tolist()))[0:-5]
    m = torch.zeros(H, W, dtype=torch.bool)
    m[upper, :] = True
    return x, m


def finetune(ae_w: Dict[str, torch.Tensor], clip_img_input: torch.Tensor, train_mask: torch.Tensor, steps: int = 60,
             lr: float = 3e-6):
    """:162-214 / fitting_temp_slide.py:861-893: Adam over all AE parameters on mean|rec - input| over train_mask.
    Returns (finetuned weights, clip_img_rec [1,1,d,T] un-padded)."""
    w = {k: v.detach().clone().requires_grad_(True) for k, v in ae_w.items()}
    opt = torch.optim.Adam(list(w.values()), lr=lr)
    for _ in range(steps):
        opt.zero_grad()
        rec, _ = O.ae_forward(w, clip_img_input)
        res = rec[:, 0] - clip_img_input[:, 0]
        loss = res[:, train_mask].abs().mean()
        loss.backward()
        opt.step()
    with torch.no_grad():
        rec, _ = O.ae_forward(w, clip_img_input)
    return {k: v.detach() for k, v in w.items()}, rec[:, :, 1:-1, 8:-8]


def decode_markers(clip_img_rec: torch.Tensor, clip_img: torch.Tensor, rot_0_pivot, stats: Dict[str, np.ndarray]):
    """opt_amass_temp.py:273-325 (twin: fitting_temp_slide.py:903-932).  clip_img_rec [d,T] (channel 0 of the
    network output), clip_img [4,d,T] (un-masked input; gives the global trajectory).
    Returns (contact_lbl_rec [T,4] f32, markers_rec [T,67,3] f64 global)."""
    T = clip_img_rec.shape[-1]
    lbl = torch.sigmoid(clip_img_rec[-4:, :].permute(1, 0)).clone()
    lbl[lbl > 0.5] = 1.0
    lbl[lbl <= 0.5] = 0.0
    body = clip_img_rec[0:-4, :]
    traj = torch.cat([clip_img[1, 0:1], clip_img[2, 0:1], clip_img[3, 0:1]], dim=0)            # [3,T]
    body = torch.cat([traj, body], dim=0).permute(1, 0).reshape(T, -1, 3).detach().numpy()      # f32 [T,1+1+67,3]
    body = np.reshape(body, (T, -1))
    body[:, 3:] = body[:, 3:] * stats['Xstd_local'][0:-4] + stats['Xmean_local'][0:-4]          # f64 math, f32 store
    body[:, 0:2] = body[:, 0:2] * stats['Xstd_global_xy'] + stats['Xmean_global_xy']
    body[:, 2] = body[:, 2] * stats['Xstd_global_r'] + stats['Xmean_global_r']
    body = np.reshape(body, (T, -1, 3))
    body = np.concatenate([np.zeros([T, 1, 3]), body[:, 1:], body[:, 0:1]], axis=1)             # f64 [T,1+68+1,3]
    glob = MO.reconstruct_global_body(body, rot_0_pivot)                                         # [T,68,3]
    return lbl, glob[:, 1:, :]


def amass_fit_clip(so: O.SmplxOracle, vposer_w, enc_w, ae_w, ids, Xmean, Xstd, stats, clip_img: torch.Tensor, rot_0_pivot,
                   init_params: np.ndarray, steps: int = 100, finetune_steps: int = 60, weights: Optional[dict] = None):
    """One clip of opt_amass_temp.py end to end (:159-458).  Returns a dict with every intermediate the tests pin:
    clip_img_rec, contact_lbl_rec, markers_rec, and body_params_opt_t_72 = p72 of the LAST forward (:457)."""
    x_in, m = amass_mask_input(clip_img)
    _, rec = finetune(ae_w, x_in, m, steps=finetune_steps)
    lbl, markers_rec = decode_markers(rec[0, 0], clip_img[0], rot_0_pivot, stats)
    fit = O.AmassFitOracle(so, vposer_w, enc_w, ids, Xmean, Xstd, init_params, markers_rec.astype(np.float32), lbl.numpy(),
                           faithful=False, weights=weights)
    hist = [fit.step() for _ in range(steps)]
    return dict(clip_img_input=x_in, train_mask=m, clip_img_rec=rec, contact_lbl_rec=lbl, markers_rec=markers_rec,
                p72=fit.last_p72.numpy(), hist=hist, fit=fit)


# ------------------------------------------------------------------------------------------------------------------
# PROX: the opt_step == 0 block
# ------------------------------------------------------------------------------------------------------------------
def canonical_frame(joints_frame0: torch.Tensor) -> torch.Tensor:
    """fitting_temp_slide.py:783-789 (same statements as opt_amass_temp.py:368-375)"""
    x_axis = joints_frame0[2, :] - joints_frame0[1, :]
    x_axis = torch.cat([x_axis[:2], torch.zeros(1)])
    x_axis = x_axis / torch.norm(x_axis)
    z_axis = torch.tensor([0., 0., 1.])
    y_axis = torch.linalg.cross(z_axis, x_axis)
    y_axis = y_axis / torch.norm(y_axis)
    return torch.stack([x_axis, y_axis, z_axis], dim=1)


def prox_contact_labels(markers_normed: torch.Tensor) -> torch.Tensor:
    """:793-816: velocity (< 0.22 m/s at 30 fps) AND height (< min z + 0.10) criteria on markers 16/47/30/60"""
    foot = [16, 47, 30, 60]
    vel = torch.cat([torch.norm((markers_normed[1:, i:i + 1] - markers_normed[0:-1, i:i + 1]) * 30, dim=-1) for i in foot], -1)
    lbl = torch.zeros(markers_normed.shape[0], 4)
    lbl[0:-1, :][torch.abs(vel) < 0.22] = 1.0
    z_thres = torch.min(markers_normed[:, :, -1]) + 0.10
    fm = torch.cat([markers_normed[:, i:i + 1] for i in foot], dim=-2)
    thres = (fm[:, :, 2] < z_thres).float()
    lbl = lbl * thres
    lbl[-1, :] = thres[-1, :]
    return lbl


def prox_window_setup(vertices_world: torch.Tensor, smplx_joints_world: torch.Tensor, marker_mask: torch.Tensor,
                      ae_w, stats: Dict[str, np.ndarray], markers67_ids, finetune_steps: int = 60):
    """fitting_temp_slide.py:776-941 on detached inputs.  Returns dict(body_markers_rec [T,67,3] f32 in PROX world
    coordinates, contact_lbl_rec [T,4], clip_img_input, train_mask, clip_img_rec) or None when nothing is occluded
    (:858, the block is skipped)."""
    vw, jw = vertices_world.detach(), smplx_joints_world.detach()
    markers = vw[:, torch.as_tensor(np.asarray(markers67_ids, np.int64)), :]
    joints_3d = jw[:, 0:25]
    j0 = joints_3d[0]
    R0 = canonical_frame(j0)
    joints_n = torch.matmul(joints_3d - j0[0], R0)
    markers_n = torch.matmul(markers - j0[0], R0)
    lbls = prox_contact_labels(markers_n).numpy()
    cur_body = torch.cat([joints_n[:, 0:1], markers_n], dim=1).numpy()
    img, rot_0_pivot = MO.get_local_markers_4chan(cur_body, lbls)
    clip_img = torch.from_numpy(img).float().unsqueeze(0)                                         # [1,4,T-1,d]
    f = lambda k: torch.from_numpy(np.asarray(stats[k])).float()
    clip_img[:, 0] = (clip_img[:, 0] - f('Xmean_local')) / f('Xstd_local')
    clip_img[:, 1:3] = (clip_img[:, 1:3] - f('Xmean_global_xy')) / f('Xstd_global_xy')
    clip_img[:, 3] = (clip_img[:, 3] - f('Xmean_global_r')) / f('Xstd_global_r')
    clip_img = clip_img.permute(0, 1, 3, 2)                                                        # [1,4,d,T-1]
    x_in = clip_img.clone()
    mm = marker_mask.repeat_interleave(3).reshape([marker_mask.shape[0], -1]).permute(1, 0).unsqueeze(0).unsqueeze(0)
    left = (mm[:, :, 48:49, :] == 1) * (mm[:, :, 90:91, :] == 1)
    right = (mm[:, :, 141:142, :] == 1) * (mm[:, :, 180:181, :] == 1)
    cmask = torch.cat([left, right, left, right], dim=-2).float()
    T = x_in.shape[-1]
    mask = torch.cat([torch.ones(1, 1, 3, T), mm[:, :, :, 0:T], cmask[:, :, :, 0:T]], dim=-2)      # [1,1,208,T]
    x_in[:, 0:1] = x_in[:, 0:1] * mask
    if not (marker_mask.shape[0] * marker_mask.shape[1] > marker_mask.sum()):
        return None
    x_in = F.pad(x_in, P2D, 'reflect')
    mf = F.pad(mask, P2D, 'reflect')[:, 0]
    mf[:, -5:, :] = 0
    train_mask = (mf[0] == 1)
    _, rec = finetune(ae_w, x_in, train_mask, steps=finetune_steps)
    x_un = x_in[:, :, 1:-1, 8:-8]
    lbl_rec, glob = decode_markers(rec[0, 0], x_un[0], rot_0_pivot, stats)
    out = torch.from_numpy(glob).float()
    out[:, :, 2] = out[:, :, 2] + markers_n[:, :, 2].min()
    out = torch.matmul(out, torch.inverse(R0)) + j0[0]
    return dict(body_markers_rec=out, contact_lbl_rec=lbl_rec, clip_img_input=x_in, train_mask=train_mask,
                clip_img_rec=rec, contact_lbls_in=lbls, rot_0_pivot=rot_0_pivot)


# ------------------------------------------------------------------------------------------------------------------
# per-frame fit (stage 1): opt_amass_perframe.py:291-363
# ------------------------------------------------------------------------------------------------------------------
def perframe_loss_terms(so: O.SmplxOracle, vposer_w, ids: torch.Tensor, w: dict, transl, rot6d, shape_t, other, tgt):
    """ONE evaluation of the per-frame objective (opt_amass_perframe.py:324-351): returns (total, parts, p72, verts).
    ``perframe_fit`` below is this in the reference's loop, and that loop is pinned to the reference's own text at 0.0
    (tests/golden/oracle_vs_reference.txt, row perframe.*), so the single iteration is pinned with it."""
    p75 = torch.cat([transl, rot6d, shape_t, other], dim=-1)
    p72 = O.convert_to_3D_rot(p75)
    body_pose = O.vposer_decode(vposer_w, p72[:, 16:48], 'aa').view(1, -1)
    verts, _, _ = so.forward(betas=p72[:, 6:16], global_orient=p72[:, 3:6], body_pose=body_pose,
                             left_hand_pose=p72[:, 48:60], right_hand_pose=p72[:, 60:], transl=p72[:, 0:3])
    parts = dict(marker=F.l1_loss(verts[:, ids, :], tgt), vposer=torch.mean(p72[:, 16:48] ** 2),
                 shape=torch.mean(p72[:, 6:16] ** 2), hand=torch.mean(p72[:, 48:] ** 2))
    loss = (w['rec_markers'] * parts['marker'] + w['vposer'] * parts['vposer'] +
            w['shape'] * parts['shape'] + w['hand'] * parts['hand'])
    return loss, parts, p72, verts


def perframe_iteration(so: O.SmplxOracle, vposer_w, markers67_ids, p72_aa: np.ndarray, target: np.ndarray,
                       weights: Optional[dict] = None):
    """losses and gradients of ONE per-frame iteration at the parameters ``p72_aa`` [72] (axis-angle orientation; converted
    to the 6-D parameterisation the loop optimises, :303-306) against ``target`` [67,3] -- the GPU gate of BASELINE
    configs[0].  Returns dict(total, marker, vposer, shape, hand, g_transl, g_rot6d, g_other, rot6d)."""
    w = dict(O.LOSS_WEIGHTS if weights is None else weights)
    ids = torch.as_tensor(np.asarray(markers67_ids, np.int64))
    p = torch.from_numpy(np.asarray(p72_aa, np.float32)).view(1, 72)
    transl = p[:, 0:3].clone().requires_grad_(True)
    rot6d = O.convert_to_6D_all(p[:, 3:6]).detach().clone().requires_grad_(True)
    other = p[:, 16:].clone().requires_grad_(True)
    tgt = torch.from_numpy(np.asarray(target, np.float32)).view(1, -1, 3)
    loss, parts, _, verts = perframe_loss_terms(so, vposer_w, ids, w, transl, rot6d, p[:, 6:16], other, tgt)
    loss.backward()
    out = {k: float(v.detach()) for k, v in parts.items()}
    out.update(total=float(loss.detach()), g_transl=transl.grad.numpy().copy(), g_rot6d=rot6d.grad.numpy().copy(),
               g_other=other.grad.numpy().copy(), rot6d=rot6d.detach().numpy().copy(), verts=verts.detach().numpy().copy())
    return out


def perframe_fit(so: O.SmplxOracle, vposer_w, markers67_ids, markers_rec: np.ndarray, betas: np.ndarray, steps: int = 100,
                 weights: Optional[dict] = None):
    """Returns ``body_params_opt_cur_clip`` [T,72] (p72 of each frame's LAST forward) and the per-frame final loss."""
    w = dict(O.LOSS_WEIGHTS if weights is None else weights)
    ids = torch.as_tensor(np.asarray(markers67_ids, np.int64))
    T = markers_rec.shape[0]
    shape_t = torch.from_numpy(np.asarray(betas, np.float32)).view(1, 10)
    out, last = [], []
    transl = rot6d = other = None
    for t in range(T):
        tgt = torch.from_numpy(np.asarray(markers_rec[t:t + 1], np.float32))
        if t == 0:                                                                           # :298-310
            transl = torch.tensor([[0.0, 0.4, 1.0]])
            rot6d = O.convert_to_6D_all(torch.tensor([[0.0, 1.6, 3.14]])).detach().clone()
            other = torch.zeros(1, 56)
            for p in (transl, rot6d, other):
                p.requires_grad = True
        opt = torch.optim.Adam([transl, rot6d, other], lr=0.1 if t == 0 else 0.01)          # :312-317
        for step in range(steps):
            if step > 60:
                for g in opt.param_groups:
                    g['lr'] = 0.01
            if step > 80:
                for g in opt.param_groups:
                    g['lr'] = 0.003
            opt.zero_grad()
            loss, _, p72, _ = perframe_loss_terms(so, vposer_w, ids, w, transl, rot6d, shape_t, other, tgt)
            loss.backward()
            opt.step()
        out.append(p72[0].detach().numpy().copy())
        last.append(float(loss.detach()))
    return np.asarray(out), np.asarray(last)
