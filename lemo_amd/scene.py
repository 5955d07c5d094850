"""Scene (SDF) terms of the PROX fitting loss on the HIP kernels.

Reference: temp_prox/fitting_temp_slide.py:685-739 -- ``F.grid_sample(self.sdf, norm_vertices[:, :, [2,1,0]]...,
padding_mode='border')`` on a 256^3 signed-distance volume.  The reference repeats the volume B times
(fit_temp_loadprox_slide.py:299, 6.7 GB at B = 100); here one copy is sampled by all frames."""
from __future__ import annotations

import ctypes as C

import torch

from . import _hip
from ._hip import ptr


class _SdfSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, sdf, gmin, gmax, lib):
        shape = pts.shape[:-1]
        p = pts.reshape(-1, 3).contiguous().float()
        _hip.check_device(lib, p)
        N = p.shape[0]
        val = torch.empty(N, dtype=torch.float32, device=p.device)
        dval = torch.empty(N, 3, dtype=torch.float32, device=p.device)
        D, H, W = sdf.shape
        g0 = (C.c_float * 3)(*[float(v) for v in gmin])
        g1 = (C.c_float * 3)(*[float(v) for v in gmax])
        lib.check(lib.sdf_sample(ptr(sdf), D, H, W, ptr(p), N, g0, g1, ptr(val), ptr(dval), lib.stream(p.device)), 'sdf_sample')
        ctx.save_for_backward(dval)
        ctx.shape = pts.shape
        return val.view(shape)

    @staticmethod
    def backward(ctx, g):
        (dval,) = ctx.saved_tensors
        return (dval * g.reshape(-1, 1)).view(ctx.shape), None, None, None, None


def sdf_sample(points_world: torch.Tensor, sdf: torch.Tensor, grid_min, grid_max, _lib=None) -> torch.Tensor:
    """points_world [...,3] -> sdf value [...]; sdf [D,H,W] float32 contiguous on the same device."""
    assert sdf.dim() == 3 and sdf.is_contiguous() and sdf.dtype == torch.float32
    return _SdfSample.apply(points_world, sdf, grid_min, grid_max, _lib or _hip.get_lib())
