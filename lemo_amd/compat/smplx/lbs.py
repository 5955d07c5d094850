"""``smplx.lbs`` surface: ``lbs`` and ``transform_mat`` with the signatures of the vendored statement
(human_body_prior/body_model/lbs.py:34-35 and :196).

``lbs`` runs on the same HIP kernels as :class:`lemo_amd.body_model.SMPLX` (pose stage + blend-shape GEMM fused with
skinning, hand-written backward): the model tensors of a call are turned once into the kernels' layouts and cached, so
repeated calls with the same tensors cost one forward.  Scope: the 55-joint SMPL-X skeleton LEMO uses (any vertex
count), up to 20 shape coefficients, ``pose2rot=True`` (axis-angle pose) -- anything else raises instead of falling
back to a CPU path.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from ...body_model import SMPLX

_CACHE: Dict[Tuple, Tuple] = {}
_NJ = 55


def transform_mat(R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """[B,3,3], [B,3,1] -> homogeneous [B,4,4] (lbs.py:196-205; glue used by temp_prox/camera.py:88-116)."""
    return torch.cat([F.pad(R, [0, 0, 0, 1]), F.pad(t, [0, 0, 0, 1], value=1)], dim=2)


def _model_for(v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights, lib) -> SMPLX:
    tensors = (v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights)
    key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors) + (id(lib),)
    hit = _CACHE.get(key)
    # the entry keeps the caller's tensors alive (so their addresses cannot be recycled for another model) and a hit
    # must be the very same tensor objects
    if hit is not None and all(a is b for a, b in zip(hit[1], tensors)):
        return hit[0]
    v_template = v_template.reshape(-1, 3) if v_template.dim() == 3 else v_template       # smplx passes [1,V,3] too
    V = int(v_template.shape[0])
    nb = int(shapedirs.shape[-1])
    J = int(J_regressor.shape[0])
    if J != _NJ or int(parents.shape[0]) != _NJ or tuple(lbs_weights.shape) != (V, _NJ):
        raise NotImplementedError('lemo_amd lbs(): the 55-joint SMPL-X skeleton only (got %d joints)' % J)
    if nb > 20:
        raise NotImplementedError('lemo_amd lbs(): at most 20 shape coefficients (got %d)' % nb)
    P = (_NJ - 1) * 9
    if tuple(posedirs.shape) != (P, V * 3):
        raise ValueError('posedirs must be [%d, V*3] as in smplx (got %s)' % (P, tuple(posedirs.shape)))
    f32 = np.float32
    sd = np.zeros((V, 3, 20), f32)
    sd[:, :, :nb] = shapedirs.detach().cpu().numpy().reshape(V, 3, nb)
    par = parents.detach().cpu().numpy().astype(np.int64).copy()
    par[0] = -1
    model = dict(v_template=v_template.detach().cpu().numpy().astype(f32), shapedirs=sd,
                 posedirs=posedirs.detach().cpu().numpy().astype(f32).T.reshape(V, 3, P),
                 J_regressor=J_regressor.detach().cpu().numpy().astype(np.float64),
                 kintree_table=np.stack([par, np.arange(_NJ)]), weights=lbs_weights.detach().cpu().numpy().astype(f32),
                 hands_componentsl=np.eye(45, dtype=f32), hands_componentsr=np.eye(45, dtype=f32),
                 hands_meanl=np.zeros(45, f32), hands_meanr=np.zeros(45, f32),
                 f=np.array([[0, 1, 2]], np.int64), lmk_faces_idx=np.zeros(1, np.int64),
                 lmk_bary_coords=np.full((1, 3), 1.0 / 3.0, f32))
    m = SMPLX(model, use_pca=False, flat_hand_mean=True, num_betas=10, batch_size=1, extra_joint_ids=[0],
              create_transl=False, _lib=lib)
    for p in m.parameters():
        p.requires_grad_(False)
    if len(_CACHE) > 8:
        _CACHE.clear()
    _CACHE[key] = (m, tensors)
    return m


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights, pose2rot: bool = True,
        dtype=torch.float32, _lib=None):
    """Linear blend skinning: (vertices [B,V,3], posed joints [B,55,3]); differentiable w.r.t. ``betas`` and ``pose``.
    ``betas`` [B,nb], ``pose`` [B,165] axis-angle (lbs.py:34-119)."""
    if not pose2rot:
        raise NotImplementedError('lemo_amd lbs(): pose2rot=False (rotation-matrix pose) is not on LEMO\'s path')
    if dtype != torch.float32:
        raise NotImplementedError('lemo_amd lbs(): fp32 only')
    m = _model_for(v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights, _lib)
    B = int(max(betas.shape[0], pose.shape[0]))
    pose = pose.reshape(pose.shape[0], -1)
    if pose.shape[1] != 3 * _NJ:
        raise ValueError('pose must be [B,165] axis-angle')
    nb = int(betas.shape[1])
    b20 = F.pad(betas, [0, 20 - nb]) if nb < 20 else betas
    if b20.shape[0] != B:
        b20 = b20.expand(B, -1)
    if pose.shape[0] != B:
        pose = pose.expand(B, -1)
    out = m(betas=b20[:, :10], expression=b20[:, 10:20], global_orient=pose[:, 0:3], body_pose=pose[:, 3:66],
            jaw_pose=pose[:, 66:69], leye_pose=pose[:, 69:72], reye_pose=pose[:, 72:75],
            left_hand_pose=pose[:, 75:120], right_hand_pose=pose[:, 120:165], return_verts=True)
    return out.vertices, out.joints[:, :_NJ]
