"""CPU oracle for the LEMO temporal-fitting hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file restates, in plain fp32 CPU PyTorch, the arithmetic of the reference's per-iteration
temporal fitting loop (``/root/reference/opt_amass_temp.py:349-455``) and of every third-party
function that loop calls.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; the product (``lemo_amd``) never does and fails loudly when its
HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * ``lbs`` family        -- pinned: cross-checked in the build container against the reference's
                             vendored ``human_body_prior/body_model/lbs.py`` (tests/golden/make_golden.py).
  * ``Enc`` / ``AE``      -- pinned: cross-checked against ``models/AE_sep.py`` / ``models/AE.py``
                             imported from /root/reference, real ``runs/15217`` weights.
  * 6-D / axis-angle helpers -- pinned against ``utils/utils.py`` run through THIS file's
                             torchgeometry restatement; the torchgeometry==0.1.2 and smplx==0.1.26
                             wheels themselves are absent from /root/reference (requirements.txt:5,9)
                             => *parity unpinned* for those two packages; they are restated from
                             their published algorithm and held by invariants (scipy agreement,
                             round trips, orthonormality) in tests/test_oracle.py.
  * ``VPoser.decode``     -- pinned against ``human_body_prior/train/vposer_smpl.py`` imported with
                             stubs (seeded weights; no checkpoint ships with the reference).
  * AMASS loop body       -- restated line by line from ``opt_amass_temp.py:355-455`` (the script
                             itself cannot be imported: it needs smplx/open3d/AMASS at import time).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# torchgeometry==0.1.2 restatement (call sites: utils/utils.py:80,89; vposer_smpl.py:160,170)
# --------------------------------------------------------------------------------------------


def rotation_matrix_to_quaternion(rotation_matrix: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """tgm 0.1.2 ``rotation_matrix_to_quaternion`` -- input [N,3,4], output [N,4] (w first)."""
    rmat_t = torch.transpose(rotation_matrix, 1, 2)
    mask_d2 = rmat_t[:, 2, 2] < eps
    mask_d0_d1 = rmat_t[:, 0, 0] > rmat_t[:, 1, 1]
    mask_d0_nd1 = rmat_t[:, 0, 0] < -rmat_t[:, 1, 1]

    t0 = 1 + rmat_t[:, 0, 0] - rmat_t[:, 1, 1] - rmat_t[:, 2, 2]
    q0 = torch.stack([rmat_t[:, 1, 2] - rmat_t[:, 2, 1], t0,
                      rmat_t[:, 0, 1] + rmat_t[:, 1, 0], rmat_t[:, 2, 0] + rmat_t[:, 0, 2]], -1)
    t1 = 1 - rmat_t[:, 0, 0] + rmat_t[:, 1, 1] - rmat_t[:, 2, 2]
    q1 = torch.stack([rmat_t[:, 2, 0] - rmat_t[:, 0, 2], rmat_t[:, 0, 1] + rmat_t[:, 1, 0],
                      t1, rmat_t[:, 1, 2] + rmat_t[:, 2, 1]], -1)
    t2 = 1 - rmat_t[:, 0, 0] - rmat_t[:, 1, 1] + rmat_t[:, 2, 2]
    q2 = torch.stack([rmat_t[:, 0, 1] - rmat_t[:, 1, 0], rmat_t[:, 2, 0] + rmat_t[:, 0, 2],
                      rmat_t[:, 1, 2] + rmat_t[:, 2, 1], t2], -1)
    t3 = 1 + rmat_t[:, 0, 0] + rmat_t[:, 1, 1] + rmat_t[:, 2, 2]
    q3 = torch.stack([t3, rmat_t[:, 1, 2] - rmat_t[:, 2, 1],
                      rmat_t[:, 2, 0] - rmat_t[:, 0, 2], rmat_t[:, 0, 1] - rmat_t[:, 1, 0]], -1)

    # tgm writes `1 - mask` on bool tensors (errors on torch>=1.2); `~` is the meaning.
    c0 = (mask_d2 & mask_d0_d1).view(-1, 1).type_as(q0)
    c1 = (mask_d2 & ~mask_d0_d1).view(-1, 1).type_as(q0)
    c2 = (~mask_d2 & mask_d0_nd1).view(-1, 1).type_as(q0)
    c3 = (~mask_d2 & ~mask_d0_nd1).view(-1, 1).type_as(q0)
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    q = q / torch.sqrt(t0.view(-1, 1) * c0 + t1.view(-1, 1) * c1 + t2.view(-1, 1) * c2 + t3.view(-1, 1) * c3)
    return q * 0.5


def quaternion_to_angle_axis(quaternion: torch.Tensor) -> torch.Tensor:
    """tgm 0.1.2 ``quaternion_to_angle_axis`` -- [N,4] (w first) -> [N,3]."""
    q1, q2, q3 = quaternion[..., 1], quaternion[..., 2], quaternion[..., 3]
    sin_squared_theta = q1 * q1 + q2 * q2 + q3 * q3
    sin_theta = torch.sqrt(sin_squared_theta)
    cos_theta = quaternion[..., 0]
    two_theta = 2.0 * torch.where(cos_theta < 0.0,
                                  torch.atan2(-sin_theta, -cos_theta),
                                  torch.atan2(sin_theta, cos_theta))
    k_pos = two_theta / sin_theta
    k_neg = 2.0 * torch.ones_like(sin_theta)
    k = torch.where(sin_squared_theta > 0.0, k_pos, k_neg)
    return torch.stack([q1 * k, q2 * k, q3 * k], dim=-1)


def rotation_matrix_to_angle_axis(rotation_matrix: torch.Tensor) -> torch.Tensor:
    """tgm 0.1.2: [N,3,4] -> [N,3]."""
    return quaternion_to_angle_axis(rotation_matrix_to_quaternion(rotation_matrix))


def angle_axis_to_rotation_matrix(angle_axis: torch.Tensor) -> torch.Tensor:
    """tgm 0.1.2: [N,3] -> [N,4,4]; Rodrigues if theta^2 > 1e-6 else first-order Taylor."""
    eps = 1e-6
    _aa = angle_axis.unsqueeze(1)
    theta2 = torch.matmul(_aa, _aa.transpose(1, 2)).squeeze(1)           # [N,1]
    theta = torch.sqrt(theta2)
    wxyz = angle_axis / (theta + eps)
    wx, wy, wz = torch.chunk(wxyz, 3, dim=1)
    c, s = torch.cos(theta), torch.sin(theta)
    k_one = 1.0
    r00 = c + wx * wx * (k_one - c)
    r10 = wz * s + wx * wy * (k_one - c)
    r20 = -wy * s + wx * wz * (k_one - c)
    r01 = wx * wy * (k_one - c) - wz * s
    r11 = c + wy * wy * (k_one - c)
    r21 = wx * s + wy * wz * (k_one - c)
    r02 = wy * s + wx * wz * (k_one - c)
    r12 = -wx * s + wy * wz * (k_one - c)
    r22 = c + wz * wz * (k_one - c)
    rot_normal = torch.cat([r00, r01, r02, r10, r11, r12, r20, r21, r22], dim=1).view(-1, 3, 3)
    rx, ry, rz = torch.chunk(angle_axis, 3, dim=1)
    k1 = torch.ones_like(rx)
    rot_taylor = torch.cat([k1, -rz, ry, rz, k1, -rx, -ry, rx, k1], dim=1).view(-1, 3, 3)
    mask = (theta2 > eps).view(-1, 1, 1).type_as(theta2)
    rot3 = mask * rot_normal + (1 - mask) * rot_taylor
    out = torch.eye(4, dtype=angle_axis.dtype).view(1, 4, 4).repeat(angle_axis.shape[0], 1, 1)
    out[:, :3, :3] = rot3
    return out


# --------------------------------------------------------------------------------------------
# utils/utils.py:50-137  (6-D continuous rotation <-> axis angle)
# --------------------------------------------------------------------------------------------


def rot6d_to_matrix(x: torch.Tensor) -> torch.Tensor:
    """ContinousRotReprDecoder.decode -- utils/utils.py:63-70; vposer_smpl.py:53-62."""
    r = x.view(-1, 3, 2)
    b1 = F.normalize(r[:, :, 0], dim=1)
    dot = torch.sum(b1 * r[:, :, 1], dim=1, keepdim=True)
    b2 = F.normalize(r[:, :, 1] - dot * b1, dim=-1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack([b1, b2, b3], dim=-1)


def matrot2aa(m: torch.Tensor) -> torch.Tensor:
    """ContinousRotReprDecoder.matrot2aa -- utils/utils.py:74-81."""
    h = F.pad(m.reshape(-1, 3, 3), [0, 1])
    return rotation_matrix_to_angle_axis(h).view(-1, 3).contiguous()


def aa2matrot(aa: torch.Tensor) -> torch.Tensor:
    """ContinousRotReprDecoder.aa2matrot -- utils/utils.py:84-90."""
    return angle_axis_to_rotation_matrix(aa.reshape(-1, 3))[:, :3, :3].contiguous()


def convert_to_6D_all(aa: torch.Tensor) -> torch.Tensor:
    """utils/utils.py:127-130."""
    return aa2matrot(aa)[:, :, :-1].reshape([-1, 6])


def convert_to_3D_all(x6: torch.Tensor) -> torch.Tensor:
    """utils/utils.py:133-137."""
    return matrot2aa(rot6d_to_matrix(x6))


def convert_to_3D_rot(x: torch.Tensor) -> torch.Tensor:
    """utils/utils.py:111-123 -- [B, 3+6+rest] -> [B, 3+3+rest]."""
    return torch.cat([x[:, :3], matrot2aa(rot6d_to_matrix(x[:, 3:9])), x[:, 9:]], dim=-1)


# --------------------------------------------------------------------------------------------
# human_body_prior/body_model/lbs.py (vendored smplx.lbs) -- restated
# --------------------------------------------------------------------------------------------


def blend_shapes(betas, shape_disps):
    """lbs.py:142-163."""
    return torch.einsum('bl,mkl->bmk', [betas, shape_disps])


def vertices2joints(J_regressor, vertices):
    """lbs.py:122-139."""
    return torch.einsum('bik,ji->bjk', [vertices, J_regressor])


def batch_rodrigues(aa):
    """lbs.py:166-193 -- note ``angle = ||aa + 1e-8||`` (:178)."""
    n = aa.shape[0]
    angle = torch.norm(aa + 1e-8, dim=1, keepdim=True)
    rot_dir = aa / angle
    cos = torch.unsqueeze(torch.cos(angle), dim=1)
    sin = torch.unsqueeze(torch.sin(angle), dim=1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((n, 1), dtype=aa.dtype)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view((n, 3, 3))
    ident = torch.eye(3, dtype=aa.dtype).unsqueeze(dim=0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def transform_mat(R, t):
    """lbs.py:196-205."""
    return torch.cat([F.pad(R, [0, 0, 0, 1]), F.pad(t, [0, 0, 0, 1], value=1)], dim=2)


def batch_rigid_transform(rot_mats, joints, parents):
    """lbs.py:208-263."""
    B, N = rot_mats.shape[0], joints.shape[1]
    joints = torch.unsqueeze(joints, dim=-1)
    rel_joints = joints.clone()
    rel_joints[:, 1:] -= joints[:, parents[1:]]
    tm = transform_mat(rot_mats.reshape(-1, 3, 3), rel_joints.reshape(-1, 3, 1)).view(-1, N, 4, 4)
    chain = [tm[:, 0]]
    for i in range(1, parents.shape[0]):
        chain.append(torch.matmul(chain[int(parents[i])], tm[:, i]))
    transforms = torch.stack(chain, dim=1)
    posed_joints = transforms[:, :, :3, 3]
    jh = torch.cat([joints, torch.zeros([B, N, 1, 1], dtype=joints.dtype)], dim=2)
    init_bone = F.pad(torch.matmul(transforms, jh), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed_joints, transforms - init_bone


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights,
        return_intermediates: bool = False):
    """lbs.py:34-119.  ``pose`` [B,(J)*3] axis-angle; returns verts [B,V,3], joints [B,J,3]."""
    B = betas.shape[0]
    v_shaped = v_template + blend_shapes(betas, shapedirs)
    J = vertices2joints(J_regressor, v_shaped)
    rot_mats = batch_rodrigues(pose.reshape(-1, 3)).view([B, -1, 3, 3])
    ident = torch.eye(3, dtype=betas.dtype)
    pose_feature = (rot_mats[:, 1:, :, :] - ident).view([B, -1])
    v_posed = torch.matmul(pose_feature, posedirs).view(B, -1, 3) + v_shaped
    J_transformed, A = batch_rigid_transform(rot_mats, J, parents)
    W = lbs_weights.unsqueeze(dim=0).expand([B, -1, -1])
    nj = J_regressor.shape[0]
    T = torch.matmul(W, A.view(B, nj, 16)).view(B, -1, 4, 4)
    v_homo = torch.matmul(T, torch.cat([v_posed, torch.ones([B, v_posed.shape[1], 1], dtype=betas.dtype)],
                                       dim=2).unsqueeze(-1))
    verts = v_homo[:, :, :3, 0]
    if return_intermediates:
        return verts, J_transformed, dict(v_posed=v_posed, J=J, rot_mats=rot_mats, A=A)
    return verts, J_transformed


# --------------------------------------------------------------------------------------------
# smplx==0.1.26 ``SMPLX.forward`` restatement (call sites utils/utils.py:152,167)
# --------------------------------------------------------------------------------------------

# smplx.vertex_ids['smplx'] in VertexJointSelector order (face, feet, then l/r fingertips)
EXTRA_JOINT_VERTEX_IDS = [9120, 9929, 9448, 616, 6,                       # nose reye leye rear lear
                          5770, 5780, 8846, 8463, 8474, 8635,               # L big/small toe, heel; R ...
                          5361, 4933, 5058, 5169, 5286,                     # l thumb index middle ring pinky
                          8079, 7669, 7794, 7905, 8022]                     # r ...


class SmplxOracle:
    """Holds an SMPL-X-shaped model (dict of numpy arrays, real-file key names) and evaluates the
    smplx 0.1.26 forward on CPU.  ``model`` keys: v_template [V,3], shapedirs [V,3,>=20],
    posedirs [V,3,486], J_regressor [55,V], kintree_table [2,55], weights [V,55], f [F,3],
    hands_components{l,r} [45,45], hands_mean{l,r} [45], lmk_faces_idx [51], lmk_bary_coords [51,3].
    """

    def __init__(self, model: Dict[str, np.ndarray], num_pca_comps: int = 12, use_pca: bool = True,
                 flat_hand_mean: bool = False, num_betas: int = 10, extra_joint_ids=None):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        self.V = model['v_template'].shape[0]
        self.v_template = t(model['v_template'])
        sd = model['shapedirs']
        self.shapedirs = t(sd[:, :, :num_betas])
        self.expr_dirs = t(sd[:, :, 10:20] if sd.shape[-1] < 310 else sd[:, :, 300:310])
        self.posedirs = t(np.reshape(model['posedirs'], [-1, model['posedirs'].shape[-1]]).T)
        self.J_regressor = t(model['J_regressor'])
        parents = np.asarray(model['kintree_table'][0], dtype=np.int64).copy()
        parents[0] = -1
        self.parents = torch.from_numpy(parents)
        self.lbs_weights = t(model['weights'])
        self.faces = torch.from_numpy(np.asarray(model['f'], dtype=np.int64))
        self.use_pca = use_pca
        self.lh_comp = t(model['hands_componentsl'][:num_pca_comps])
        self.rh_comp = t(model['hands_componentsr'][:num_pca_comps])
        lh_mean = np.zeros(45, np.float32) if flat_hand_mean else model['hands_meanl']
        rh_mean = np.zeros(45, np.float32) if flat_hand_mean else model['hands_meanr']
        self.pose_mean = torch.cat([torch.zeros(3 + 63 + 9), t(lh_mean), t(rh_mean)])
        self.lmk_faces_idx = torch.from_numpy(np.asarray(model['lmk_faces_idx'], dtype=np.int64))
        self.lmk_bary = t(model['lmk_bary_coords'])
        ids = EXTRA_JOINT_VERTEX_IDS if extra_joint_ids is None else extra_joint_ids
        self.extra_ids = torch.tensor(ids, dtype=torch.int64)

    def full_pose(self, global_orient, body_pose, left_hand_pose, right_hand_pose,
                  jaw_pose=None, leye_pose=None, reye_pose=None):
        B = global_orient.shape[0]
        z3 = torch.zeros(B, 3)
        jaw_pose = z3 if jaw_pose is None else jaw_pose
        leye_pose = z3 if leye_pose is None else leye_pose
        reye_pose = z3 if reye_pose is None else reye_pose
        if self.use_pca:
            left_hand_pose = torch.einsum('bi,ij->bj', [left_hand_pose, self.lh_comp])
            right_hand_pose = torch.einsum('bi,ij->bj', [right_hand_pose, self.rh_comp])
        fp = torch.cat([global_orient, body_pose, jaw_pose, leye_pose, reye_pose,
                        left_hand_pose, right_hand_pose], dim=1)
        return fp + self.pose_mean

    def forward(self, betas, global_orient, body_pose, left_hand_pose, right_hand_pose, transl=None,
                expression=None, jaw_pose=None, leye_pose=None, reye_pose=None, joint_mapper=None):
        B = global_orient.shape[0]
        expression = torch.zeros(B, 10) if expression is None else expression
        fp = self.full_pose(global_orient, body_pose, left_hand_pose, right_hand_pose,
                            jaw_pose, leye_pose, reye_pose)
        shape_comp = torch.cat([betas, expression], dim=-1)
        shapedirs = torch.cat([self.shapedirs, self.expr_dirs], dim=-1)
        verts, joints = lbs(shape_comp, fp, self.v_template, shapedirs, self.posedirs,
                            self.J_regressor, self.parents, self.lbs_weights)
        # vertices2landmarks (static face landmarks)
        lmk_faces = self.faces[self.lmk_faces_idx]                      # [51,3]
        lmk_vertices = verts[:, lmk_faces]                               # [B,51,3,3]
        landmarks = torch.einsum('blfi,lf->bli', [lmk_vertices, self.lmk_bary])
        joints = torch.cat([joints, verts[:, self.extra_ids], landmarks], dim=1)   # 55+21+51
        if joint_mapper is not None:
            joints = joints[:, joint_mapper]
        if transl is not None:
            joints = joints + transl.unsqueeze(1)
            verts = verts + transl.unsqueeze(1)
        return verts, joints, fp


# --------------------------------------------------------------------------------------------
# VPoser.decode  --  human_body_prior/train/vposer_smpl.py:107-121
# --------------------------------------------------------------------------------------------


def vposer_decode(w: Dict[str, torch.Tensor], Z: torch.Tensor, output_type: str = 'aa', branches=None, hidden=None, perturb=None):
    """``w`` holds bodyprior_dec_{fc1,fc2,out}.{weight,bias}; dropout is identity in eval().

    Test devices (not reference behaviour): ``branches`` = two bool tensors [B,512], True where a hidden unit takes the slope-1 branch
    of its LeakyReLU whatever its sign (``AmassFitOracle.decisions``); ``hidden`` = a list that receives the two hidden activations;
    ``perturb(x6d, abs_sum)`` = replaces the out layer's result (abs_sum = sum_i |w_ji h_i| + |b_j|, the scale of an fp32 dot product's
    rounding error): how tests measure the CONDITIONING of the Gram-Schmidt decode that follows (tests/kink_attribution.py)."""
    assert output_type in ('matrot', 'aa')
    x = Z
    for i, name in enumerate(('bodyprior_dec_fc1', 'bodyprior_dec_fc2')):
        pre = F.linear(x, w[name + '.weight'], w[name + '.bias'])
        if branches is None:
            x = F.leaky_relu(pre, 0.2)
        else:
            x = pre * torch.where(branches[i], torch.ones((), dtype=pre.dtype), torch.full((), 0.2, dtype=pre.dtype))
        if hidden is not None:
            hidden.append(x.detach())
    h2 = x
    x = F.linear(x, w['bodyprior_dec_out.weight'], w['bodyprior_dec_out.bias'])
    if perturb is not None:
        x = perturb(x, F.linear(h2.detach().abs(), w['bodyprior_dec_out.weight'].abs(), w['bodyprior_dec_out.bias'].abs()))
    nj = x.shape[1] // 6
    x = rot6d_to_matrix(x).view([-1, 1, nj, 9])
    if output_type == 'aa':
        B = x.shape[0]
        h = F.pad(x.view(-1, 3, 3), [0, 1])
        return rotation_matrix_to_angle_axis(h).view(B, 1, -1, 3).contiguous()
    return x


def make_vposer_weights(seed: int = 2, latentD: int = 32, num_neurons: int = 512, nj: int = 21):
    """Seeded ``nn.Linear`` default init of the three decoder layers (no checkpoint ships)."""
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name, (fin, fout) in (('bodyprior_dec_fc1', (latentD, num_neurons)),
                              ('bodyprior_dec_fc2', (num_neurons, num_neurons)),
                              ('bodyprior_dec_out', (num_neurons, nj * 6))):
        bound = 1.0 / math.sqrt(fin)
        w[name + '.weight'] = (torch.rand(fout, fin, generator=g) * 2 - 1) * bound
        w[name + '.bias'] = (torch.rand(fout, generator=g) * 2 - 1) * bound
    return w


# --------------------------------------------------------------------------------------------
# models/AE_sep.py::Enc (downsample=False)  and  models/AE.py::AE (downsample=True)
# --------------------------------------------------------------------------------------------


def enc_forward(w: Dict[str, torch.Tensor], x: torch.Tensor, return_all: bool = False, branches=None):
    """models/AE_sep.py:91-99 with ``downsample=False`` (no pooling, :24-27).

    ``branches`` (test device, not reference behaviour): ten bool tensors, True where a unit is to take the slope-1 branch of its
    LeakyReLU whatever the sign of its pre-activation -- ``AmassFitOracle.decisions`` explains the use."""
    acts = []
    for blk in range(1, 6):
        for idx in (0, 2):
            pre = F.conv2d(x, w[f'enc_blc{blk}.main.{idx}.weight'], w[f'enc_blc{blk}.main.{idx}.bias'], stride=1, padding=1)
            if branches is None:
                x = F.leaky_relu(pre, 0.2)
            else:
                b = branches[len(acts)].reshape(pre.shape)
                x = pre * torch.where(b, torch.ones((), dtype=pre.dtype), torch.full((), 0.2, dtype=pre.dtype))
            acts.append(x)
    return (x, acts) if return_all else x


def ae_forward(w: Dict[str, torch.Tensor], x: torch.Tensor):
    """models/AE.py:93-108 with ``downsample=True, kernel=3``: 5x(conv,lrelu,conv,lrelu,maxpool 3/2/1)
    then 5x(deconv s2 with output_size, lrelu, deconv s1, [lrelu])."""
    sizes = [x.shape]
    for blk in range(1, 6):
        for idx in (0, 2):
            x = F.leaky_relu(F.conv2d(x, w[f'enc_blc{blk}.main.{idx}.weight'],
                                      w[f'enc_blc{blk}.main.{idx}.bias'], stride=1, padding=1), 0.2)
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        sizes.append(x.shape)
    z = x
    for blk in range(1, 6):
        tgt = sizes[5 - blk]
        # ConvTranspose2d(..., stride=2, padding=1)(x, output_size=tgt): output_padding chosen to hit tgt
        oph = tgt[2] - ((x.shape[2] - 1) * 2 - 2 + 3)
        opw = tgt[3] - ((x.shape[3] - 1) * 2 - 2 + 3)
        x = F.conv_transpose2d(x, w[f'dec_blc{blk}.deconv1.weight'], w[f'dec_blc{blk}.deconv1.bias'],
                               stride=2, padding=1, output_padding=(oph, opw))
        x = F.leaky_relu(x, 0.2)
        x = F.conv_transpose2d(x, w[f'dec_blc{blk}.deconv2.weight'], w[f'dec_blc{blk}.deconv2.bias'],
                               stride=1, padding=1)
        if blk < 5:
            x = F.leaky_relu(x, 0.2)
    return x, z


# --------------------------------------------------------------------------------------------
# AMASS temporal-fitting iteration  --  opt_amass_temp.py:349-455
# --------------------------------------------------------------------------------------------

LOSS_WEIGHTS = dict(rec_markers=1.0, contact_vel=0.03, smooth=1e6, vposer=0.02, shape=0.01, hand=0.01)
"""opt_amass_temp.py:47-52."""


class AmassFitOracle:
    """One sequence of the AMASS temporal fit, restated from ``opt_amass_temp.py:332-455``.

    ``faithful=True`` evaluates SMPL-X twice per iteration (gen_body_mesh_v1 + gen_body_joints_v1,
    utils/utils.py:141-169) exactly like the reference; ``False`` evaluates it once (same values).
    """

    def __init__(self, smplx: SmplxOracle, vposer_w, enc_w, ids: Dict[str, np.ndarray],
                 Xmean: np.ndarray, Xstd: np.ndarray, init_params: np.ndarray,
                 markers_rec: np.ndarray, contact_lbl: np.ndarray, weights: Optional[dict] = None,
                 faithful: bool = True):
        self.smplx, self.vposer_w, self.enc_w = smplx, vposer_w, enc_w
        self.ids = {k: torch.from_numpy(np.asarray(v, dtype=np.int64)) for k, v in ids.items()}
        self.Xmean = torch.from_numpy(np.asarray(Xmean)).float()          # [1,1,243]
        self.Xstd = torch.from_numpy(np.asarray(Xstd)).float()            # [243]
        self.w = dict(LOSS_WEIGHTS if weights is None else weights)
        self.faithful = faithful
        p = torch.from_numpy(np.asarray(init_params, dtype=np.float32))
        self.markers_rec = torch.from_numpy(np.asarray(markers_rec, dtype=np.float32))
        self.contact = torch.from_numpy(np.asarray(contact_lbl, dtype=np.float32))
        # :332-345
        self.transl = p[:, 0:3].clone().requires_grad_(True)
        self.rot6d = convert_to_6D_all(p[:, 3:6].clone()).detach().clone().requires_grad_(True)
        self.shape = p[:, 6:16].clone()
        self.other = p[:, 16:].clone().requires_grad_(True)
        self.opt = torch.optim.Adam([self.transl, self.rot6d, self.other], lr=0.01)
        self.step_idx = 0
        self.last_p72 = None
        # Test device (NOT reference behaviour; None = the reference's arithmetic): the objective is piecewise smooth -- 21 M LeakyReLU
        # units, 24 k L1 residuals, thresholded contact speeds -- and two correct implementations that round differently can sit on
        # different pieces.  ``decisions`` pins the piece: dict(lrelu=[10 bool tensors: unit on its slope-1 branch],
        # l1_sign=[B,67,3] in {-1,0,1}, contact=[4 bool tensors over the labelled (frame, vertex) speeds: inside the mean]).  With
        # another implementation's decisions loaded, this oracle in float64 returns the EXACT gradient of the piece that implementation
        # was on: what remains between the two is arithmetic, not luck (tests/kink_attribution.py).  Optional key vposer=[2 bool
        # tensors [B,512]]: the decoder's own 2 x 512 LeakyReLU units per frame (vposer_smpl.py:107-121) -- the fourth kink family.
        self.decisions = None

    # -- forward pieces -------------------------------------------------------------------
    def _body(self, p72):
        B = p72.shape[0]
        dec = getattr(self, 'decisions', None)
        self.last_hidden = []
        body_pose = vposer_decode(self.vposer_w, p72[:, 16:48], 'aa', branches=None if dec is None else dec.get('vposer'),
                                  hidden=self.last_hidden, perturb=getattr(self, 'perturb6d', None)).view(B, -1)
        return self.smplx.forward(betas=p72[:, 6:16], global_orient=p72[:, 3:6], body_pose=body_pose,
                                  left_hand_pose=p72[:, 48:60], right_hand_pose=p72[:, 60:],
                                  transl=p72[:, 0:3])

    def losses(self):
        """Returns (total, dict of the six scalars, p72, verts)  -- opt_amass_temp.py:355-453."""
        p75 = torch.cat([self.transl, self.rot6d, self.shape, self.other], dim=-1)
        p72 = convert_to_3D_rot(p75)
        verts, joints, _ = self._body(p72)
        if self.faithful:                                               # second SMPL-X forward (:364)
            _, joints, _ = self._body(p72)
        markers_opt = verts[:, self.ids['markers67'], :]
        markers_smooth = verts[:, self.ids['markers81'], :]
        j0 = joints[0].detach()
        x_axis = j0[2, :] - j0[1, :]
        x_axis = torch.cat([x_axis[:2], torch.zeros(1)])               # x_axis[-1] = 0
        x_axis = x_axis / torch.norm(x_axis)
        z_axis = torch.tensor([0., 0., 1.])
        y_axis = torch.linalg.cross(z_axis, x_axis)
        y_axis = y_axis / torch.norm(y_axis)
        R0 = torch.stack([x_axis, y_axis, z_axis], dim=1)
        m0 = markers_smooth[0].detach()
        g = torch.matmul(markers_smooth - m0[0], R0)                    # [B,81,3]
        img = g.reshape(g.shape[0], -1).unsqueeze(0)                    # [1,B,243]
        img = (img - self.Xmean) / self.Xstd
        img = img.permute(0, 2, 1).unsqueeze(1)                         # [1,1,243,B]
        img_v = img[:, :, :, 1:] - img[:, :, :, 0:-1]
        img_v = F.pad(img_v, (8, 8, 1, 1), 'reflect')                   # [1,1,245,B-1+16]
        dec = self.decisions
        z = enc_forward(self.enc_w, img_v) if dec is None else enc_forward(self.enc_w, img_v, branches=dec['lrelu'])
        z_v = z[:, :, :, 1:] - z[:, :, :, 0:-1]
        loss_smooth = torch.mean(z_v ** 2)

        if dec is None:
            loss_marker = F.l1_loss(markers_opt, self.markers_rec)
        else:
            loss_marker = torch.mean((markers_opt - self.markers_rec) * dec['l1_sign'].to(markers_opt.dtype))
        loss_vposer = torch.mean(p72[:, 16:48] ** 2)
        loss_shape = torch.mean(p72[:, 6:16] ** 2)
        loss_hand = torch.mean(p72[:, 48:] ** 2)

        loss_contact = torch.tensor(0.0)
        if self.w['contact_vel'] > 0:
            vel = (verts[1:] - verts[0:-1]) * 30
            for k, name in enumerate(('left_heel', 'right_heel', 'left_toe', 'right_toe')):
                lbl = self.contact[:, k]
                s = torch.norm(vel[:, self.ids[name], :][lbl[0:-1] == 1], dim=-1)
                part = torch.tensor(0.0)
                if dec is not None:
                    if bool(dec['contact'][k].any()):
                        part = s[dec['contact'][k]].abs().mean()
                elif (s - 0.1).gt(0).sum().item() >= 1:
                    part = s[s > 0.1].abs().mean()
                loss_contact = loss_contact + part
        total = (self.w['rec_markers'] * loss_marker + self.w['vposer'] * loss_vposer +
                 self.w['shape'] * loss_shape + self.w['hand'] * loss_hand +
                 self.w['contact_vel'] * loss_contact + self.w['smooth'] * loss_smooth)
        parts = dict(marker=loss_marker, vposer=loss_vposer, shape=loss_shape, hand=loss_hand,
                     contact=loss_contact, smooth=loss_smooth)
        return total, parts, p72, verts

    def step(self):
        """One Adam iteration (:349-455).  Returns dict of python floats."""
        if self.step_idx > 60:
            for g in self.opt.param_groups:
                g['lr'] = 0.005
        self.opt.zero_grad()
        total, parts, p72, _ = self.losses()
        total.backward()
        self.opt.step()
        self.step_idx += 1
        self.last_p72 = p72.detach()
        out = {k: float(v) for k, v in parts.items()}
        out['total'] = float(total)
        return out

    def params75(self):
        return torch.cat([self.transl, self.rot6d, self.shape, self.other], dim=-1).detach()


def mpjpe_mm(j_a: torch.Tensor, j_b: torch.Tensor, n_joints: int = 22) -> float:
    """mean over frames and the first 22 body joints of ||Ja - Jb||_2, in mm."""
    return float(torch.norm(j_a[:, :n_joints] - j_b[:, :n_joints], dim=-1).mean() * 1000.0)
