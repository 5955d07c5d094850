"""float64 instances of the oracle  --  TEST INFRASTRUCTURE.

Used to put a number on "fp32 rounding": the GPU path and the fp32 CPU oracle are both compared with the same
restatement evaluated in float64 (gradients, Adam trajectories), so that a parity tolerance can be stated as a
multiple of what fp32 arithmetic costs the REFERENCE's own CPU path instead of a guessed constant."""
from __future__ import annotations

import contextlib

import numpy as np
import torch

from . import lemo_oracle as O


@contextlib.contextmanager
def default_f64():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        yield
    finally:
        torch.set_default_dtype(old)


def _to_double(obj):
    for k, v in list(vars(obj).items()):
        if isinstance(v, torch.Tensor) and v.is_floating_point():
            setattr(obj, k, v.double())


def amass_fit_oracle_f64(model, vposer_w, enc_w, ids, Xmean, Xstd, init_params, markers_rec, contact_lbl, weights=None,
                         extra_joint_ids=None) -> O.AmassFitOracle:
    """AmassFitOracle whose every tensor is float64 (call its methods inside ``default_f64()``)."""
    with default_f64():
        so = O.SmplxOracle(model, extra_joint_ids=extra_joint_ids)
        _to_double(so)
        vw = {k: torch.as_tensor(np.asarray(v)).double() for k, v in vposer_w.items()}
        ew = {k: torch.as_tensor(np.asarray(v)).double() for k, v in enc_w.items()}
        fit = O.AmassFitOracle(so, vw, ew, ids, np.asarray(Xmean).reshape(1, 1, -1), Xstd, init_params, markers_rec, contact_lbl,
                               faithful=False, weights=weights)
        for k in ('Xmean', 'Xstd', 'markers_rec', 'contact', 'shape'):
            setattr(fit, k, getattr(fit, k).double())
        p = torch.from_numpy(np.asarray(init_params, np.float32))
        fit.transl = p[:, 0:3].double().clone().requires_grad_(True)
        fit.rot6d = O.convert_to_6D_all(p[:, 3:6].clone()).double().clone().requires_grad_(True)     # same f32 start as the fit
        fit.other = p[:, 16:].double().clone().requires_grad_(True)
        fit.opt = torch.optim.Adam([fit.transl, fit.rot6d, fit.other], lr=0.01)
    return fit


def perframe_iteration_f64(model, vposer_w, markers67_ids, p72_aa, target, weights=None, extra_joint_ids=None):
    """``pipeline_oracle.perframe_iteration`` (opt_amass_perframe.py:324-351, one evaluation + gradients) in float64, from
    the SAME float32 start point (the 6-D orientation is converted in float32 like the fit's)."""
    from . import pipeline_oracle as PO
    with default_f64():
        so = O.SmplxOracle(model, extra_joint_ids=extra_joint_ids)
        _to_double(so)
        vw = {k: torch.as_tensor(np.asarray(v)).double() for k, v in vposer_w.items()}
        w = dict(O.LOSS_WEIGHTS if weights is None else weights)
        ids = torch.as_tensor(np.asarray(markers67_ids, np.int64))
        p = torch.from_numpy(np.asarray(p72_aa, np.float32)).view(1, 72)
        transl = p[:, 0:3].double().clone().requires_grad_(True)
        rot6d = O.convert_to_6D_all(p[:, 3:6].clone()).double().clone().requires_grad_(True)
        other = p[:, 16:].double().clone().requires_grad_(True)
        tgt = torch.from_numpy(np.asarray(target, np.float32)).double().view(1, -1, 3)
        loss, parts, _, verts = PO.perframe_loss_terms(so, vw, ids, w, transl, rot6d, p[:, 6:16].double(), other, tgt)
        loss.backward()
        out = {k: float(v.detach()) for k, v in parts.items()}
        out.update(total=float(loss.detach()), g_transl=transl.grad.numpy().copy(), g_rot6d=rot6d.grad.numpy().copy(),
                   g_other=other.grad.numpy().copy(), verts=verts.detach().numpy().copy())
    return out


def perframe_fit_f64(model, vposer_w, markers67_ids, markers_rec, betas, steps=100, weights=None, extra_joint_ids=None):
    """``pipeline_oracle.perframe_fit`` (opt_amass_perframe.py:291-363) with every tensor in float64 from the same
    float32 start values: what the loop does when rounding is taken out of it -- the yardstick for how far two
    correct float32 implementations of this loop may drift apart (the marker term's sign() gradient and Adam's
    normalisation turn rounding-sized differences into O(lr) steps)."""
    from . import pipeline_oracle as PO
    import torch.nn.functional as F  # noqa: F401
    with default_f64():
        so = O.SmplxOracle(model, extra_joint_ids=extra_joint_ids)
        _to_double(so)
        vw = {k: torch.as_tensor(np.asarray(v)).double() for k, v in vposer_w.items()}
        w = dict(O.LOSS_WEIGHTS if weights is None else weights)
        ids = torch.as_tensor(np.asarray(markers67_ids, np.int64))
        T = markers_rec.shape[0]
        shape_t = torch.from_numpy(np.asarray(betas, np.float32)).double().view(1, 10)
        out, last = [], []
        transl = rot6d = other = None
        for t in range(T):
            tgt = torch.from_numpy(np.asarray(markers_rec[t:t + 1], np.float32)).double()
            if t == 0:
                transl = torch.tensor([[0.0, 0.4, 1.0]], dtype=torch.float32).double()
                rot6d = O.convert_to_6D_all(torch.tensor([[0.0, 1.6, 3.14]], dtype=torch.float32)).detach().double().clone()
                other = torch.zeros(1, 56, dtype=torch.float64)
                for p in (transl, rot6d, other):
                    p.requires_grad = True
            opt = torch.optim.Adam([transl, rot6d, other], lr=0.1 if t == 0 else 0.01)
            for step in range(steps):
                if step > 60:
                    for g in opt.param_groups:
                        g['lr'] = 0.01
                if step > 80:
                    for g in opt.param_groups:
                        g['lr'] = 0.003
                opt.zero_grad()
                loss, _, p72, _ = PO.perframe_loss_terms(so, vw, ids, w, transl, rot6d, shape_t, other, tgt)
                loss.backward()
                opt.step()
            out.append(p72[0].detach().numpy().copy())
            last.append(float(loss.detach()))
    return np.asarray(out), np.asarray(last)
