"""Marker-image encode / decode around the fitting loop on the GPU.

Drop-ins for the two numpy helpers of the reference, same names, argument order and return values:

    reconstruct_global_body(body_joints_input [T, 1+J+1, 3], rot_0_pivot) -> [T, J, 3]     utils/utils.py:184-203
    get_local_markers_4chan(cur_body [T, 1+67, 3], contact_lbls [T, 4]) -> ([4, T-1, d], rot_0_pivot)   :209-265

called at ``opt_amass_temp.py:273-325`` (decode of the infilling network's output into the target markers of the fit)
and by the loaders that build the clip images.  Inputs may be numpy arrays (results come back as float64 numpy, like
the reference) or torch tensors on the HIP device (results stay on the device, float32 / float64 pivot).  Unlike the
reference the inputs are not modified in place.  The arithmetic runs in ``liblemo_hip.so`` (marker_kernels.hip): the
sequential quaternion integration over T is a prefix sum of heading angles, one block per clip.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _hip
from ._hip import ptr


def _device_of(x, device):
    if isinstance(x, torch.Tensor):
        return x.device
    return torch.device(device if device is not None else 'cuda:0')


def _to_dev(x, dev):
    if isinstance(x, torch.Tensor):
        return x.detach().to(dev, torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)


def reconstruct_global_body(body_joints_input, rot_0_pivot, device=None, _lib=None):
    lib = _lib or _hip.get_lib()
    as_numpy = not isinstance(body_joints_input, torch.Tensor)
    dev = torch.device('cpu') if lib.is_emu else _device_of(body_joints_input, device)
    x = _to_dev(body_joints_input, dev)
    _hip.check_device(lib, x)
    assert x.dim() == 3 and x.shape[2] == 3 and x.shape[1] >= 3, 'expects [T, 1+J+1, 3]'
    T, J = x.shape[0], x.shape[1] - 2
    out = torch.empty(T, J, 3, dtype=torch.float32, device=dev)
    if isinstance(rot_0_pivot, torch.Tensor):          # a device tensor (e.g. from get_local_markers_4chan) is read in place
        piv = rot_0_pivot.detach().to(dev, torch.float64).reshape(-1)[:1].contiguous()
        lib.check(lib.reconstruct_global_body_dev(ptr(x), T, J, ptr(piv), ptr(out), lib.stream(dev)), 'reconstruct_global_body')
    else:
        rot0 = float(np.asarray(rot_0_pivot, np.float64).reshape(-1)[0])
        lib.check(lib.reconstruct_global_body(ptr(x), T, J, rot0, ptr(out), lib.stream(dev)), 'reconstruct_global_body')
    return out.cpu().numpy().astype(np.float64) if as_numpy else out


def get_local_markers_4chan(cur_body, contact_lbls, device=None, _lib=None):
    lib = _lib or _hip.get_lib()
    as_numpy = not isinstance(cur_body, torch.Tensor)
    dev = torch.device('cpu') if lib.is_emu else _device_of(cur_body, device)
    x, c = _to_dev(cur_body, dev), _to_dev(contact_lbls, dev)
    _hip.check_device(lib, x)
    assert x.dim() == 3 and x.shape[2] == 3 and c.shape == (x.shape[0], 4), 'expects [T, 1+67, 3] and [T, 4]'
    T, M1 = x.shape[0], x.shape[1]
    img = torch.empty(4, T - 1, 3 * M1 + 4, dtype=torch.float32, device=dev)
    piv = torch.zeros(1, dtype=torch.float64, device=dev)
    lib.check(lib.local_markers_4chan(ptr(x), ptr(c), T, M1, ptr(img), ptr(piv), lib.stream(dev)), 'local_markers_4chan')
    if as_numpy:
        return img.cpu().numpy().astype(np.float64), piv.cpu().numpy()
    return img, piv
