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


def prox_fit_oracle_f64(prob: dict, first_batch_flag: bool = False, cam: dict = None):
    """``prox_oracle.ProxFitOracle`` on a ``__graft_entry__.prox_small_problem``-shaped dict with every floating-point tensor in
    float64 (call its methods inside ``default_f64()``): the exact gradient / loss at a given fp32 state of a PROX window, the
    yardstick for "how far is a correct fp32 implementation of this closure from the reference's fp32" (the
    PROX gradient gates were constants).  ``grid_sample`` and the reductions run in float64 as well."""
    from . import prox_oracle as PO
    cam = cam or dict(fx=1060.53, fy=1060.38, cx=951.30, cy=536.77)
    with default_f64():
        so = O.SmplxOracle(prob['model'], extra_joint_ids=list(range(21)) if prob['V'] < 9930 else None)
        _to_double(so)
        vw = {k: torch.as_tensor(np.asarray(v)).double() for k, v in prob['vposer_w'].items()}
        ew = {k: torch.as_tensor(np.asarray(v)).double() for k, v in prob['enc_w'].items()}
        of = PO.ProxFitOracle(so, vw, ew, prob['joint_map'], prob['ids'], prob['fric_ids'], prob['Xmean'], prob['Xstd'], prob['weights'],
                              cam, prob['R'], prob['t'], prob['sdf'], prob['grid_min'], prob['grid_max'], prob['params'],
                              prob['gt_joints'], prob['joints_conf'], first_batch_flag=first_batch_flag, **prob['infill'])
        _to_double(of)
        of.joint_weights = of.joint_weights.double()
        of.p = {k: v.detach().double().clone().requires_grad_(True) for k, v in of.p.items()}
        of.pose_embedding = of.pose_embedding.detach().double().clone().requires_grad_(True)
        of.opt = torch.optim.Adam(list(of.p.values()) + [of.pose_embedding], lr=0.005)
    return of


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


class KinkProbe:
    """Which frames can a rounding-sized difference move by MORE than rounding?  (computed, not narrated: VERDICT r02 #5)

    The AMASS objective has three families of kinks: 21 M LeakyReLU units of the smoothness encoder (slope 0.2 | 1), the L1
    marker residuals (sign), and the contact term's ``speed > 0.1`` selection.  A unit / residual / speed that sits closer to
    its kink than the implementation's own error takes the other branch, and the gradient of the frames in its reach then
    changes by a finite amount however small the error was.  Wrapped around a float64 ``AmassFitOracle`` this probe records,
    for every evaluation of the objective, the kinks that lie within a stated tolerance of being crossed and the frames
    their flip can reach:

    * LeakyReLU unit of layer l at image column c with |activation| <= ``tol_act`` x max|activation of layer l|
      (default 3e-6 = ~8 x the measured error of an fp32-accurate convolution, 4e-7 of the layer maximum) -> reach:
      image columns c - 10 .. c + 10 (ten 3x3 layers), i.e. velocity columns j = c - 8 (reflect-padded by 8) and frames
      j, j + 1;
    * marker residual with |r| <= ``tol_res`` metres (default 3e-6: ~8 x an fp32 vertex at 1.6 m) -> its frame;
    * contact speed with |s - 0.1| <= ``tol_speed`` m/s (default 2e-4: speeds are 30 x vertex differences) -> frames t, t+1
      and every frame of that foot set's mean (a changed count rescales the whole term: ALL frames are marked).

    ``events`` lists (evaluation index, kind, frame range); ``frames(i)`` the exposed frames of evaluation i."""

    def __init__(self, fit: O.AmassFitOracle, tol_act: float = 3e-6, tol_res: float = 3e-6, tol_speed: float = 2e-4):
        self.fit, self.tol_act, self.tol_res, self.tol_speed = fit, tol_act, tol_res, tol_speed
        self.B = int(fit.markers_rec.shape[0])
        self.events, self.n_eval = [], 0
        self._orig_enc, self._orig_losses = None, None

    # -- patching -----------------------------------------------------------------------------------------------
    def __enter__(self):
        probe = self
        self._orig_enc = O.enc_forward

        def enc(w, x, return_all=False):
            z, acts = probe._orig_enc(w, x, True)
            probe._scan_acts(acts)
            return (z, acts) if return_all else z
        O.enc_forward = enc
        self._orig_losses = self.fit.losses

        def losses():
            out = probe._orig_losses()
            probe._scan_verts(out[3])
            probe.n_eval += 1
            return out
        self.fit.losses = losses
        return self

    def __exit__(self, *a):
        O.enc_forward = self._orig_enc
        self.fit.losses = self._orig_losses
        return False

    # -- scans ----------------------------------------------------------------------------------------------------
    def _col_frames(self, c: int, radius: int):
        """frames a kink at image column c can reach through `radius` 3x3 layers, the temporal difference and the reflect pad"""
        nd = self.B - 1
        fr = set()
        for cc in range(c - radius, c + radius + 1):
            if cc < 0 or cc > nd + 15:
                continue
            j = cc - 8
            if j < 0:
                j = -j                       # F.pad(..., 'reflect'): column 8 - k mirrors velocity column k
            elif j > nd - 1:
                j = 2 * (nd - 1) - j
            j = min(max(j, 0), nd - 1)
            fr.update((j, j + 1))
        return fr

    def _scan_acts(self, acts):
        for l, a in enumerate(acts, start=1):
            a = a.detach()
            m = float(a.abs().max())
            if m == 0.0:
                continue
            near = (a.abs() <= self.tol_act * m).nonzero()
            for c in sorted({int(v) for v in near[:, 3].tolist()}):
                self.events.append((self.n_eval, f'lrelu{l}', c, self._col_frames(c, 10)))

    def _scan_verts(self, verts):
        f = self.fit
        v = verts.detach()
        r = (v[:, f.ids['markers67'], :] - f.markers_rec).abs()
        for t in sorted({int(x) for x in (r <= self.tol_res).nonzero()[:, 0].tolist()}):
            self.events.append((self.n_eval, 'l1', t, {t}))
        if f.w['contact_vel'] > 0:
            vel = (v[1:] - v[:-1]) * 30
            for k, name in enumerate(('left_heel', 'right_heel', 'left_toe', 'right_toe')):
                lbl = f.contact[:, k]
                s = torch.norm(vel[:, f.ids[name], :], dim=-1)[lbl[0:-1] == 1]
                if s.numel() and bool(((s - 0.1).abs() <= self.tol_speed).any()):
                    self.events.append((self.n_eval, 'contact_' + name, -1, set(range(self.B))))

    # -- results ------------------------------------------------------------------------------------------------
    def frames(self, evaluation: int = 0):
        out = set()
        for e, _, _, fr in self.events:
            if e == evaluation:
                out |= fr
        return sorted(out)

    def count(self, upto: int = None):
        return sum(1 for e, *_ in self.events if upto is None or e <= upto)

    def summary(self, evaluation: int = 0):
        kinds = {}
        for e, kind, _, _ in self.events:
            if e == evaluation:
                k = 'lrelu' if kind.startswith('lrelu') else ('contact' if kind.startswith('contact') else kind)
                kinds[k] = kinds.get(k, 0) + 1
        return kinds


def flip_sensitivity_of(loss_fn, params, tol_act: float = 3e-6, subsets: int = 2, seed: int = 0):
    """COMPUTED kink exposure of a gradient that runs through the smoothness encoder.  ``KinkProbe`` shows that
    at BASELINE size hundreds of the encoder's 21 M LeakyReLU units sit within any realistic error band of their kink in EVERY
    evaluation -- proximity alone marks all frames.  What matters is how much the gradient can move when such units take the
    other branch.  This function evaluates, in float64, ``d loss_fn() / d params`` (a) as it is and (b) with the slope of
    EVERY unit whose pre-activation lies within ``tol_act`` x (layer maximum) of zero swapped (0.2 <-> 1; the value changes by
    < tol_act x max), plus ``subsets`` random halves of that set (guards against cancellation), and returns, per parameter
    tensor [B, d],

        S[frame] = max over the variants of  max_entries |G_variant - G| / max|G|

    -- a per-frame bound on what rounding-sized differences of an fp32-accurate forward can do to the gradient THROUGH THE
    ENCODER'S KINKS.  A frame's gradient error beyond ``rounding + 2 S`` is not explained by them.  (default tol_act 3e-6 = ~8 x
    the measured pre-activation error of the fp32-accurate convolutions, 4e-7 of the layer maximum.)  ``loss_fn`` must reach
    the encoder through ``lemo_oracle.enc_forward`` (both fit oracles do).  Uses ``torch.autograd.grad``: optimiser state and
    ``.grad`` fields are not touched."""
    import torch.nn.functional as F
    orig = O.enc_forward
    params = tuple(params)

    def make_enc(mode, gen):
        def enc(w, x, return_all=False):
            acts = []
            for blk in range(1, 6):
                for idx in (0, 2):
                    pre = F.conv2d(x, w[f'enc_blc{blk}.main.{idx}.weight'], w[f'enc_blc{blk}.main.{idx}.bias'], stride=1, padding=1)
                    y = F.leaky_relu(pre, 0.2)
                    if mode != 'none':
                        pa = pre.detach().abs()
                        m = pa <= tol_act * pa.max()
                        if mode == 'half':
                            m = m & (torch.rand(m.shape, generator=gen) < 0.5)
                        y = y + torch.where(m, torch.where(pre > 0, -0.8 * pre, 0.8 * pre), torch.zeros_like(pre))
                    x = y
                    acts.append(x)
            return (x, acts) if return_all else x
        return enc

    def grads(mode, gen=None):
        O.enc_forward = make_enc(mode, gen)
        try:
            with default_f64():
                return torch.autograd.grad(loss_fn(), params)
        finally:
            O.enc_forward = orig
    G0 = grads('none')
    variants = [grads('all')] + [grads('half', torch.Generator().manual_seed(seed + 1 + i)) for i in range(subsets)]
    out = []
    for i in range(len(params)):
        n = G0[i].abs().max()
        out.append(torch.stack([(g[i] - G0[i]).abs().max(1).values / n for g in variants]).max(0).values if float(n) > 0
                   else torch.zeros(G0[i].shape[0], dtype=G0[i].dtype))
    return out


def flip_sensitivity(fit: O.AmassFitOracle, tol_act: float = 3e-6, subsets: int = 2, seed: int = 0):
    """:func:`flip_sensitivity_of` for the AMASS objective at the oracle's current parameters:
    ``{'transl' | 'rot6d' | 'other': S[frame]}``"""
    S = flip_sensitivity_of(lambda: fit.losses()[0], (fit.transl, fit.rot6d, fit.other), tol_act, subsets, seed)
    return dict(zip(('transl', 'rot6d', 'other'), S))
