"""Rotation helpers with the names LEMO's scripts import from ``utils/utils.py`` (:50-137).

``convert_to_3D_rot`` (6-D -> axis-angle, used inside the fitting loop, opt_amass_temp.py:356) runs
on the HIP kernels with an analytic backward.  ``convert_to_6D_all`` / ``convert_to_6D_rot`` are
per-clip setup (opt_amass_temp.py:335: once before the loop) and stay small torch programs
restating torchgeometry==0.1.2 ``angle_axis_to_rotation_matrix``.
"""
from __future__ import annotations

import torch

from . import _hip
from ._hip import ptr


def angle_axis_to_rotation_matrix(angle_axis: torch.Tensor) -> torch.Tensor:
    """tgm 0.1.2: [N,3] -> [N,4,4] (Rodrigues if theta^2 > 1e-6, else first-order Taylor)."""
    eps = 1e-6
    theta2 = (angle_axis * angle_axis).sum(1, keepdim=True)
    theta = torch.sqrt(theta2)
    w = angle_axis / (theta + eps)
    wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
    c, s = torch.cos(theta), torch.sin(theta)
    rn = torch.cat([c + wx * wx * (1 - c), wx * wy * (1 - c) - wz * s, wy * s + wx * wz * (1 - c),
                    wz * s + wx * wy * (1 - c), c + wy * wy * (1 - c), -wx * s + wy * wz * (1 - c),
                    -wy * s + wx * wz * (1 - c), wx * s + wy * wz * (1 - c), c + wz * wz * (1 - c)], 1).view(-1, 3, 3)
    rx, ry, rz = angle_axis[:, 0:1], angle_axis[:, 1:2], angle_axis[:, 2:3]
    one = torch.ones_like(rx)
    rt = torch.cat([one, -rz, ry, rz, one, -rx, -ry, rx, one], 1).view(-1, 3, 3)
    mask = (theta2 > eps).view(-1, 1, 1).to(angle_axis.dtype)
    out = torch.eye(4, dtype=angle_axis.dtype, device=angle_axis.device).repeat(angle_axis.shape[0], 1, 1)
    out[:, :3, :3] = mask * rn + (1 - mask) * rt
    return out


def convert_to_6D_all(x_batch: torch.Tensor) -> torch.Tensor:
    """utils/utils.py:127-130."""
    m = angle_axis_to_rotation_matrix(x_batch.reshape(-1, 3))[:, :3, :3]
    return m[:, :, :-1].reshape(-1, 6)


def convert_to_6D_rot(x_batch: torch.Tensor) -> torch.Tensor:
    """utils/utils.py:94-107."""
    return torch.cat([x_batch[:, :3], convert_to_6D_all(x_batch[:, 3:6]), x_batch[:, 6:]], dim=-1)


class _Rot6dToAA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x6, lib):
        x6 = x6.contiguous().float()
        _hip.check_device(lib, x6)
        aa = torch.empty(x6.shape[0], 3, dtype=torch.float32, device=x6.device)
        lib.check(lib.rot6d_to_aa_fwd(ptr(x6), 6, x6.shape[0], ptr(aa), lib.stream(x6.device)), 'rot6d_to_aa_fwd')
        ctx.save_for_backward(x6)
        ctx.lib = lib
        return aa

    @staticmethod
    def backward(ctx, g):
        (x6,) = ctx.saved_tensors
        g = g.contiguous().float()
        dx = torch.empty_like(x6)
        ctx.lib.check(ctx.lib.rot6d_to_aa_bwd(ptr(x6), 6, ptr(g), x6.shape[0], ptr(dx), ctx.lib.stream(x6.device)),
                      'rot6d_to_aa_bwd')
        return dx, None


def convert_to_3D_all(x6: torch.Tensor, _lib=None) -> torch.Tensor:
    """utils/utils.py:133-137 -- [N,6] -> [N,3]."""
    return _Rot6dToAA.apply(x6.reshape(-1, 6), _lib or _hip.get_lib())


def convert_to_3D_rot(x_batch: torch.Tensor, _lib=None) -> torch.Tensor:
    """utils/utils.py:111-123 -- [B, 3+6+rest] -> [B, 3+3+rest]."""
    return torch.cat([x_batch[:, :3], convert_to_3D_all(x_batch[:, 3:9], _lib), x_batch[:, 9:]], dim=-1)
