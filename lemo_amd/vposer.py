"""``VPoser`` decoder on the HIP kernels, with the reference's call surface.

Reference: human_body_prior/train/vposer_smpl.py:65-171 (``VPoser``; ``decode`` :107-121,
``matrot2aa`` :153-161) and human_body_prior/tools/model_loader.py:43-72 (``load_vposer``).
LEMO only ever calls ``vposer.decode(Z, output_type='aa')`` on the fitting path
(utils/utils.py:148; temp_prox/fitting_temp_slide.py:243); the encoder half is kept as plain torch
modules so that reference ``state_dict``s load with all keys matched.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _hip
from ._hip import ptr


class _DecodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, lib, Z, want_aa: bool):
        Z = Z.contiguous().float()
        _hip.check_device(lib, Z)
        B, dev = Z.shape[0], Z.device
        w = owner._packed(dev)
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        h1, h2, o = e(B, 512), e(B, 512), e(B, 128)
        matrot, aa = e(B, 21, 9), e(B, 63)
        lib.check(lib.vposer_decode_fwd(C.byref(w['struct']), ptr(Z), Z.shape[1], B, ptr(h1), ptr(h2), ptr(o),
                                        ptr(matrot), ptr(aa), lib.stream(dev)), 'vposer_decode_fwd')
        ctx.owner, ctx.lib, ctx.saved, ctx.want_aa = owner, lib, (h1, h2, o), want_aa
        return aa.view(B, 1, 21, 3) if want_aa else matrot.view(B, 1, 21, 9)

    @staticmethod
    def backward(ctx, g):
        h1, h2, o = ctx.saved
        B, dev = h1.shape[0], h1.device
        w = ctx.owner._packed(dev)
        g = g.contiguous().float()
        dz = torch.empty(B, 32, dtype=torch.float32, device=dev)
        d_aa, d_m = (ptr(g), None) if ctx.want_aa else (None, ptr(g))
        scratch = torch.empty(B, 1152, dtype=torch.float32, device=dev)
        ctx.lib.check(ctx.lib.vposer_decode_bwd(C.byref(w['struct']), ptr(h1), ptr(h2), ptr(o), d_aa, d_m, B, ptr(dz),
                                                32, ptr(scratch), ctx.lib.stream(dev)), 'vposer_decode_bwd')
        return None, None, dz, None


class VPoser(nn.Module):
    """state_dict-compatible with vposer_smpl.py:75-89; ``decode`` runs in liblemo_hip.so."""

    def __init__(self, num_neurons=512, latentD=32, data_shape=(1, 21, 3), use_cont_repr=True,
                 _lib: Optional[_hip.HipLib] = None):
        super().__init__()
        assert num_neurons == 512 and latentD == 32 and tuple(data_shape)[1] == 21 and use_cont_repr, \
            'kernels are specialised to VPoser v1.0 (512 neurons, 32-D latent, 21 joints)'
        self.latentD, self.num_joints, self.use_cont_repr = latentD, 21, True
        n_features = int(np.prod(data_shape))
        self.bodyprior_enc_bn1 = nn.BatchNorm1d(n_features)
        self.bodyprior_enc_fc1 = nn.Linear(n_features, num_neurons)
        self.bodyprior_enc_bn2 = nn.BatchNorm1d(num_neurons)
        self.bodyprior_enc_fc2 = nn.Linear(num_neurons, num_neurons)
        self.bodyprior_enc_mu = nn.Linear(num_neurons, latentD)
        self.bodyprior_enc_logvar = nn.Linear(num_neurons, latentD)
        self.bodyprior_dec_fc1 = nn.Linear(latentD, num_neurons)
        self.bodyprior_dec_fc2 = nn.Linear(num_neurons, num_neurons)
        self.bodyprior_dec_out = nn.Linear(num_neurons, self.num_joints * 6)
        self._lib_override = _lib
        self._pack_cache = {}

    def _packed(self, device):
        """decoder weights in the kernels' layouts (both orientations), cached per (device, version)."""
        ps = [self.bodyprior_dec_fc1.weight, self.bodyprior_dec_fc1.bias, self.bodyprior_dec_fc2.weight,
              self.bodyprior_dec_fc2.bias, self.bodyprior_dec_out.weight, self.bodyprior_dec_out.bias]
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in ps)
        if self._pack_cache.get('key') != key:
            t = [p.detach().to(device=device, dtype=torch.float32).contiguous() for p in ps]
            w3 = torch.zeros(128, 512, dtype=torch.float32, device=device); w3[:126] = t[4]
            b3 = torch.zeros(128, dtype=torch.float32, device=device); b3[:126] = t[5]
            tt = dict(w1=t[0], w1t=t[0].t().contiguous(), b1=t[1], w2=t[2], w2t=t[2].t().contiguous(), b2=t[3],
                      w3=w3, w3t=w3.t().contiguous(), b3=b3)
            st = _hip.VPoserW(*[ptr(tt[k]) for k in ('w1', 'w1t', 'b1', 'w2', 'w2t', 'b2', 'w3', 'w3t', 'b3')])
            self._pack_cache = dict(key=key, tensors=tt, struct=st)
        return self._pack_cache

    def decode(self, Zin, output_type='matrot'):
        assert output_type in ['matrot', 'aa']
        if self.training:
            raise RuntimeError('decode on the HIP path is the eval() forward (dropout = identity, '
                               'model_loader.py:70); call .eval() first')
        lib = self._lib_override or _hip.get_lib()
        return _DecodeFn.apply(self, lib, Zin, output_type == 'aa')


def vposer_weight_struct(weights: Dict[str, np.ndarray], device):
    """Pack ``bodyprior_dec_*`` arrays for the fitting engine -> (ctypes struct, tensors)."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device)
    w1, w2, w3 = (np.asarray(weights[f'bodyprior_dec_{n}.weight'], np.float32) for n in ('fc1', 'fc2', 'out'))
    w3p = np.zeros((128, 512), np.float32); w3p[:126] = w3            # out layer padded 126 -> 128 rows
    b3p = np.zeros(128, np.float32); b3p[:126] = np.asarray(weights['bodyprior_dec_out.bias'], np.float32)
    tt = dict(w1=t(w1), w1t=t(w1.T), b1=t(weights['bodyprior_dec_fc1.bias']),
              w2=t(w2), w2t=t(w2.T), b2=t(weights['bodyprior_dec_fc2.bias']),
              w3=t(w3p), w3t=t(w3p.T), b3=t(b3p))
    st = _hip.VPoserW(*[ptr(tt[k]) for k in ('w1', 'w1t', 'b1', 'w2', 'w2t', 'b2', 'w3', 'w3t', 'b3')])
    return st, tt


def make_vposer_weights(seed: int = 2) -> Dict[str, np.ndarray]:
    """Seeded ``nn.Linear``-style init of the decoder (no VPoser checkpoint ships with the
    reference -- SURVEY 8(d)); identical stream to ``oracle.lemo_oracle.make_vposer_weights``."""
    import math
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name, (fin, fout) in (('bodyprior_dec_fc1', (32, 512)), ('bodyprior_dec_fc2', (512, 512)),
                              ('bodyprior_dec_out', (512, 126))):
        bound = 1.0 / math.sqrt(fin)
        w[name + '.weight'] = ((torch.rand(fout, fin, generator=g) * 2 - 1) * bound).numpy()
        w[name + '.bias'] = ((torch.rand(fout, generator=g) * 2 - 1) * bound).numpy()
    return w


def _read_vposer_settings(expr_dir):
    """the experiment's ``*.ini`` (what ``configer.Configer`` reads at model_loader.py:36-39): returns a namespace with at
    least num_neurons / latentD / data_shape; the released vposer_v1_0 values when the file or a key is missing"""
    import ast
    import configparser
    vals = dict(num_neurons=512, latentD=32, data_shape=[1, 21, 3])
    inis = sorted(glob.glob(os.path.join(expr_dir, '*.ini')))
    if inis:
        cp = configparser.ConfigParser()
        cp.read(inis[0])
        for sec in cp.sections():
            for k, v in cp.items(sec):
                try:
                    val = ast.literal_eval(v)
                except Exception:
                    val = v
                for name in ('num_neurons', 'latentD', 'data_shape'):
                    if k.lower() == name.lower():
                        vals[name] = list(val) if name == 'data_shape' else int(val)
                    elif k not in vals:
                        vals.setdefault(k, val)
    ps = type('ps', (), vals)()
    ps.best_model_fname = None
    return ps


def load_vposer(expr_dir, vp_model='snapshot'):
    """``human_body_prior.tools.model_loader.load_vposer`` (model_loader.py:43-72): returns ``(vposer, ps)`` with the
    newest ``snapshots/*.pt`` of ``expr_dir`` loaded, in eval mode.  ``ps`` carries the settings of the experiment's
    ``*.ini`` (``num_neurons``, ``latentD``, ``data_shape``, ...), like the reference's ``Configer``.  ``vp_model``:
    ``'snapshot'`` (default) -- the reference would exec the directory's ``vposer_*.py`` to get the class that was trained;
    here the HIP-backed :class:`VPoser` is that class (same ``state_dict`` keys), after checking that the settings match
    what it implements -- or a class to instantiate instead (model_loader.py:66-67)."""
    if not os.path.exists(expr_dir):
        raise ValueError('Could not find the experiment directory: %s' % expr_dir)
    snaps = sorted(glob.glob(os.path.join(expr_dir, 'snapshots', '*.pt')), key=os.path.getmtime)
    if not snaps:
        raise FileNotFoundError(f'no VPoser snapshot under {expr_dir}/snapshots')
    ps = _read_vposer_settings(expr_dir)
    ps.best_model_fname = snaps[-1]
    cls = VPoser if isinstance(vp_model, str) else vp_model
    if cls is VPoser and (ps.num_neurons != 512 or ps.latentD != 32 or list(ps.data_shape)[-2:] != [21, 3]):
        raise NotImplementedError(f'lemo_amd.VPoser implements vposer_v1_0 (512 / 32 / [1,21,3]); {expr_dir} says '
                                  f'{ps.num_neurons} / {ps.latentD} / {ps.data_shape}')
    vp = cls(num_neurons=ps.num_neurons, latentD=ps.latentD, data_shape=tuple(ps.data_shape))
    vp.load_state_dict(torch.load(snaps[-1], map_location='cpu'))
    vp.eval()
    return vp, ps
