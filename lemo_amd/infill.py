"""Motion-infilling autoencoder ``AE`` (the infilling prior) on the HIP kernels, forward AND training step.

Reference: ``models/AE.py:11-108`` (``AE(downsample=True, in_channel=4, kernel=3)``: 5 x [conv, LeakyReLU,
conv, LeakyReLU, MaxPool2d(3,2,1)] then 5 x [ConvTranspose2d(k3,s2,p1)(x, output_size=...), LeakyReLU,
ConvTranspose2d(k3,s1,p1)(, LeakyReLU)]) and its per-clip self-supervised finetune
(``opt_amass_temp.py:154-214``, ``temp_prox/fitting_temp_slide.py:861-893``: 60 x [forward, L1 on the unmasked
rows, backward, Adam lr 3e-6] + one eval forward).

Same ``state_dict`` keys as the reference.  Every layer runs in liblemo_hip.so on CG8P activations:
stride-1 (transposed) convolutions on the fp32-MFMA ``conv3x3_mfma``; a stride-2 transposed convolution is
zero-stuffing to ``output_size`` followed by a stride-1 one; weight gradients on the MFMA ``conv3x3_wgrad``.
Channel counts that are not MFMA-tile multiples (4 inputs, 1 output) are zero-padded.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _hip
from ._hip import ptr
from .priors import cg8p_alloc, from_cg8p, to_cg8p

ENC_CH = [(4, 32), (32, 64), (64, 128), (128, 256), (256, 256)]
DEC_CH = [(256, 256), (256, 128), (128, 64), (64, 32), (32, 1)]
"""models/AE.py:81-91 with in_channel = 4."""


def _pad8(c):
    return (c + 7) // 8 * 8


def _pad32(c):
    return (c + 31) // 32 * 32


def _pack_fwd(w: torch.Tensor) -> torch.Tensor:
    """[Co][Ci][3][3] (channel counts already padded) -> wt[tap][Ci/8][Co][8] (priors.pack_conv3x3 on device)."""
    co, ci = w.shape[:2]
    return w.reshape(co, ci // 8, 8, 9).permute(3, 1, 0, 2).contiguous()


def _pack_bwd(w: torch.Tensor) -> torch.Tensor:
    """backward-data pack of the same conv (priors.pack_conv3x3_bwd on device)."""
    return _pack_fwd(w.flip(2, 3).transpose(0, 1).contiguous())


class _Layer:
    """one 3x3 stride-1 convolution in conv-equivalent form"""

    def __init__(self, name, cin, cout, deconv, cin_pad, cout_pad):
        self.name, self.cin, self.cout, self.deconv, self.cin_pad, self.cout_pad = name, cin, cout, deconv, cin_pad, cout_pad

    def conv_weight(self, w: torch.Tensor, fill: float = 0.0) -> torch.Tensor:
        """parameter -> padded conv weight [cout_pad][cin_pad][3][3]"""
        if self.deconv:                                            # ConvTranspose2d weight is [in][out][kh][kw]
            w = w.flip(2, 3).transpose(0, 1)
        out = torch.full((self.cout_pad, self.cin_pad, 3, 3), fill, dtype=torch.float32, device=w.device)
        out[:self.cout, :self.cin] = w
        return out

    def param_grad(self, dw: torch.Tensor) -> torch.Tensor:
        """conv-equivalent gradient [cout][cin][3][3] -> gradient in the parameter's own layout"""
        return dw.flip(2, 3).transpose(0, 1).contiguous() if self.deconv else dw


def _layers() -> List[_Layer]:
    L = []
    for b, (ci, co) in enumerate(ENC_CH, 1):
        L.append(_Layer(f'enc_blc{b}.main.0', ci, co, False, _pad8(ci), _pad32(co)))
        L.append(_Layer(f'enc_blc{b}.main.2', co, co, False, _pad32(co), _pad32(co)))
    for b, (ci, co) in enumerate(DEC_CH, 1):
        L.append(_Layer(f'dec_blc{b}.deconv1', ci, co, True, _pad32(ci), _pad32(co)))
        L.append(_Layer(f'dec_blc{b}.deconv2', co, co, True, _pad32(co), _pad32(co)))
    return L


def _conv(lib, x, wt, bias, aux, out, H, W, cin, cout, epi, s):
    # one wave owns a 32 px x (32|64) cout tile over the whole K = 9 Cin: layers with few pixels and many channels have
    # too few tiles for the chip (256 x 256 at 27 x 17: 64 waves running 1152 fp32 MFMAs each = 70 us) -> split K
    waves = ((H * W + 31) // 32) * (cout // (64 if cout % 64 == 0 else 32))
    ks = min(8, 9 * (cin // 8), max(1, 1024 // max(waves, 1)))
    if ks >= 2:
        part = torch.empty(ks * (cout // 8) * (H + 2) * (W + 2) * 8, dtype=torch.float32, device=out.device)
        lib.check(lib.conv3x3_mfma_splitk(ptr(x), ptr(wt), ptr(bias), ptr(aux), ptr(out), ptr(part), ks, H, W, cin, cout, epi, s),
                  'conv3x3_mfma_splitk')
        return
    lib.check(lib.conv3x3_mfma(ptr(x), ptr(wt), ptr(bias), ptr(aux), ptr(out), H, W, cin, cout, epi, 0, s), 'conv3x3_mfma')


class _FlatTables:
    """Gather indices between the FLAT parameter vector (weights and biases in `_layers()` order, each in the
    parameter's own layout, plus one trailing zero) and the packed operands of the kernels.  Built once per device by
    running the packing code on index tensors, so a training step needs three gathers instead of ~250 small
    zero / copy / flip / permute launches:
        packed forward weights  = flat[idx_fwd]      (all 20 layers back to back, views per layer)
        padded biases           = flat[idx_bias]
        packed backward weights = flat[idx_bwd]
        flat gradient           = dwdb[idx_grad]     (dwdb = conv-layout weight grads + bias grads of all layers)"""
    _cache: Dict[str, '_FlatTables'] = {}

    def __init__(self, device):
        L = _layers()
        self.L = L
        offs, o = [], 0
        for l in L:
            nw = l.cout * l.cin * 9
            offs.append((o, o + nw))                            # weight start, bias start
            o += nw + l.cout
        self.n_param = o
        Z = o                                                   # index of the trailing zero
        fwd, bwd, bias, self.sl_fwd, self.sl_bwd, self.sl_bias = [], [], [], [], [], []
        # position of every conv-layout gradient entry in dwdb, and the inverse map flat-param -> dwdb
        grad_idx = torch.empty(self.n_param, dtype=torch.int64)
        self.dw_off, self.db_off, g = [], [], 0
        for l, (ow, ob) in zip(L, offs):
            pshape = (l.cin, l.cout, 3, 3) if l.deconv else (l.cout, l.cin, 3, 3)
            idx = (ow + torch.arange(l.cout * l.cin * 9, dtype=torch.float32)).reshape(pshape)
            assert ob < 2 ** 24                                 # float32 carries the indices exactly through the packers
            cw = l.conv_weight(idx, fill=float(Z))
            for lst, sl, packer in ((fwd, self.sl_fwd, _pack_fwd), (bwd, self.sl_bwd, _pack_bwd)):
                pk = packer(cw).reshape(-1).long()
                sl.append((sum(t.numel() for t in lst), pk.numel(), pk.numel()))
                lst.append(pk)
            bi = torch.full((l.cout_pad,), Z, dtype=torch.int64)
            bi[:l.cout] = ob + torch.arange(l.cout)
            self.sl_bias.append((sum(t.numel() for t in bias), l.cout_pad))
            bias.append(bi)
            # gradients: the kernel writes dw in conv layout [cout][cin][3][3]; the parameter wants param_grad(dw)
            self.dw_off.append(g)
            didx = (g + torch.arange(l.cout * l.cin * 9, dtype=torch.float32)).reshape(l.cout, l.cin, 3, 3)
            grad_idx[ow:ob] = l.param_grad(didx).reshape(-1).long()
            g += l.cout * l.cin * 9
            self.db_off.append(g)
            grad_idx[ob:ob + l.cout] = g + torch.arange(l.cout)
            g += l.cout
        self.n_dwdb = g
        dev = torch.device(device)
        self.idx_fwd, self.idx_bwd = torch.cat(fwd).to(dev), torch.cat(bwd).to(dev)
        self.idx_bias, self.idx_grad = torch.cat(bias).to(dev), grad_idx.to(dev)

    @classmethod
    def get(cls, device) -> '_FlatTables':
        k = str(device)
        if k not in cls._cache:
            cls._cache[k] = cls(device)
        return cls._cache[k]


WGRAD_SECOND_STREAM = __import__('os').environ.get('LEMO_AE_SECOND_STREAM', '0') != '0'
"""Round 2 ran the weight gradients of the training step on a second stream beside the backward-data chain.  Round 3 measured what
that costs a GRAPH (tools/ae_concurrent.py, profiles/r03_ae_concurrent.txt): the 20 fork / join pairs make ``hipGraphLaunch`` of the
captured step ~0.9 ms of HOST time -- 55 of a clip's 88 ms were spent enqueueing, the finetune was host-bound and k clips side by side
did not overlap at all.  On ONE stream the 60 launches cost 4.7 ms of host time; a clip takes 81 ms (GPU-bound now) and two clips
side by side 53 ms each.  Default off since round 3 (``LEMO_AE_SECOND_STREAM=1`` or set this to True before the first clip restores
the second stream; read when a workspace is created).  Same kernels either way: identical results."""


class AEWorkspace:
    """Activation / gradient buffers of one training step, allocated and zeroed ONCE and handed out in call order.  A
    step asks for the same ~45 CG8P buffers in the same order every time and the kernels only ever write interiors (the
    zero border IS the convolution padding), so re-zeroing them per step -- 45 fill launches, 0.2 ms of a 2.0 ms step --
    is wasted work.  One workspace serves one step at a time: ``reset()`` (called by the forward) rewinds it, which is
    only legal once the previous step's backward has run (``finetune_and_infill`` guarantees that)."""

    def __init__(self, device):
        self.device, self.bufs, self.pos = torch.device(device), [], 0
        self.flats, self.fpos = [], 0
        # weight gradients may run on a second stream, next to the backward-data chain that does not depend on them (see
        # WGRAD_SECOND_STREAM below)
        self.side = torch.cuda.Stream(self.device) if (self.device.type == 'cuda' and WGRAD_SECOND_STREAM) else None

    def reset(self):
        self.pos = self.fpos = 0

    def flat(self, n: int) -> torch.Tensor:
        """uninitialised scratch of n floats (the slab partials of a weight gradient), same call-order protocol"""
        if self.fpos == len(self.flats):
            self.flats.append(torch.empty(n, dtype=torch.float32, device=self.device))
        b = self.flats[self.fpos]
        assert b.numel() == n
        self.fpos += 1
        return b

    def alloc(self, C_: int, H: int, W: int, device=None) -> torch.Tensor:
        if self.pos == len(self.bufs):
            self.bufs.append(cg8p_alloc(C_, H, W, self.device))
        b = self.bufs[self.pos]
        assert b.shape == (max(C_ // 8, 1), (H + 2) * (W + 2), 8), 'a workspace serves ONE input shape'
        self.pos += 1
        return b


def flatten_params(params) -> torch.Tensor:
    """weights and biases in `_layers()` order -> one flat vector (differentiable: autograd splits the gradient back)"""
    return torch.cat([p.reshape(-1).float() for p in params])


class _AEFn(torch.autograd.Function):
    """(x [4,H,W], flat parameter vector, see `flatten_params`) -> (out [H,W], z [256,h,w])"""

    @staticmethod
    def forward(ctx, lib, x, flat, ws=None, need_z=True):
        x = x.contiguous().float()
        if ws is not None:
            ws.reset()
        cg8p_alloc = ws.alloc if ws is not None else globals()['cg8p_alloc']
        _hip.check_device(lib, x)
        dev, s = x.device, lib.stream(x.device)
        T = _FlatTables.get(dev)
        L = T.L
        assert flat.numel() == T.n_param
        flatz = torch.cat([flat.detach().float(), torch.zeros(1, dtype=torch.float32, device=dev)])
        wf_all, b_all = flatz[T.idx_fwd], flatz[T.idx_bias]
        Wf = [wf_all[o:o + n] for o, n, _ in T.sl_fwd]
        B_ = [b_all[o:o + n] for o, n in T.sl_bias]
        H, Wd = x.shape[1:]
        xin = torch.zeros(8, H, Wd, dtype=torch.float32, device=dev)
        xin[:4] = x
        cur, curH, curW = to_cg8p(xin), H, Wd
        enc_rec, sizes = [], [(H, Wd)]
        for b in range(5):
            l0, l2 = L[2 * b], L[2 * b + 1]
            a0 = cg8p_alloc(l0.cout_pad, curH, curW, dev)
            _conv(lib, cur, Wf[2 * b], B_[2 * b], None, a0, curH, curW, l0.cin_pad, l0.cout_pad, 0, s)
            a2 = cg8p_alloc(l2.cout_pad, curH, curW, dev)
            _conv(lib, a0, Wf[2 * b + 1], B_[2 * b + 1], None, a2, curH, curW, l2.cin_pad, l2.cout_pad, 0, s)
            Ho, Wo = (curH - 1) // 2 + 1, (curW - 1) // 2 + 1
            P = cg8p_alloc(l2.cout_pad, Ho, Wo, dev)
            idx = torch.empty(l2.cout_pad // 8, Ho * Wo, 8, dtype=torch.uint8, device=dev)
            lib.check(lib.maxpool3s2_fwd(ptr(a2), curH, curW, ptr(P), ptr(idx), l2.cout_pad, s), 'maxpool3s2_fwd')
            enc_rec.append((cur, a0, a2, idx, curH, curW))
            cur, curH, curW = P, Ho, Wo
            sizes.append((Ho, Wo))
        z_buf, zH, zW = cur, curH, curW
        dec_rec = []
        for b in range(5):
            l1, l2 = L[10 + 2 * b], L[11 + 2 * b]
            tH, tW = sizes[4 - b]                                     # output_size = the matching encoder level
            S = cg8p_alloc(l1.cin_pad, tH, tW, dev)
            lib.check(lib.stuff2_fwd(ptr(cur), curH, curW, ptr(S), tH, tW, l1.cin_pad, s), 'stuff2_fwd')
            b1 = cg8p_alloc(l1.cout_pad, tH, tW, dev)
            _conv(lib, S, Wf[10 + 2 * b], B_[10 + 2 * b], None, b1, tH, tW, l1.cin_pad, l1.cout_pad, 0, s)
            b2 = cg8p_alloc(l2.cout_pad, tH, tW, dev)
            _conv(lib, b1, Wf[11 + 2 * b], B_[11 + 2 * b], None, b2, tH, tW, l2.cin_pad, l2.cout_pad, 0 if b < 4 else 2, s)
            dec_rec.append((cur, curH, curW, S, b1, b2, tH, tW))
            cur, curH, curW = b2, tH, tW
        out = from_cg8p(cur, H, Wd)[0]
        z = from_cg8p(z_buf, zH, zW)[:256] if need_z else out.new_empty(0)        # (the finetune loop never reads the latent)
        ctx.lib, ctx.T, ctx.flatz, ctx.enc_rec, ctx.dec_rec, ctx.shape, ctx.ws = lib, T, flatz, enc_rec, dec_rec, (H, Wd, zH, zW), ws
        ctx.need_z = need_z
        if not need_z:
            ctx.mark_non_differentiable(z)
        return out, z

    @staticmethod
    def backward(ctx, dout, dz):
        lib, T = ctx.lib, ctx.T
        L = T.L
        H, Wd, zH, zW = ctx.shape
        dev = ctx.flatz.device
        s = lib.stream(dev)
        cg8p_alloc = ctx.ws.alloc if ctx.ws is not None else globals()['cg8p_alloc']
        wb_all = ctx.flatz[T.idx_bwd]
        Wb = {i: wb_all[T.sl_bwd[i][0]:T.sl_bwd[i][0] + T.sl_bwd[i][1]] for i in range(1, 20)}   # layer 0 needs no backward-data
        dwdb = torch.empty(T.n_dwdb, dtype=torch.float32, device=dev)     # every layer's dw (conv layout) and db
        zeros_bias = torch.zeros(256, dtype=torch.float32, device=dev)
        nsl = lambda h, w: lib.conv3x3_wgrad_nslab(h, w)

        # With a workspace (the finetune loop) the 20 weight gradients (0.77 of the step's 2.0 ms when serialised; each
        # needs only d(pre-activation) of its layer and a saved activation) go to the workspace's second stream and overlap
        # the backward-data chain; the streams join before the gradients are gathered.  Same kernels: bit-identical.
        side = ctx.ws.side if (ctx.ws is not None and not lib.is_emu) else None
        main = torch.cuda.current_stream(dev) if side is not None else None
        keep = []          # operands of side-stream launches stay referenced until the join: a tensor released earlier goes
                           # back to the caching allocator, which may hand it to the next main-stream op while it is still read

        jobs = (_hip.WgradJob * len(L))()
        njobs = [0]

        def wgrad(i, dpre, xin, h, w):
            """slab partials of layer i now (second stream when there is one); all 20 reductions in ONE launch at the end"""
            l = L[i]
            npart = nsl(h, w) * 9 * l.cout_pad * l.cin_pad
            part = ctx.ws.flat(npart) if ctx.ws is not None else torch.empty(npart, dtype=torch.float32, device=dev)
            dw, db = dwdb[T.dw_off[i]:T.db_off[i]], dwdb[T.db_off[i]:T.db_off[i] + l.cout]
            sw = s
            if side is not None:
                side.wait_stream(main)                     # dpre was just produced on the main stream
                sw = side.cuda_stream
            keep.extend((dpre, xin, part))
            lib.check(lib.conv3x3_wgrad_partial(ptr(dpre), ptr(xin), h, w, l.cin_pad, l.cout_pad, ptr(part), sw), 'conv3x3_wgrad_partial')
            j = jobs[njobs[0]]
            j.partial, j.dy, j.dw, j.db = ptr(part), ptr(dpre), ptr(dw), ptr(db)
            j.nslab, j.cin, j.cout, j.cin_real, j.cout_real, j.H, j.W = nsl(h, w), l.cin_pad, l.cout_pad, l.cin, l.cout, h, w
            njobs[0] += 1

        # ---- decoder, last block first.  dpre = d(pre-activation of the block's deconv2 output)
        d = torch.zeros(32, H, Wd, dtype=torch.float32, device=dev)
        if dout is not None:
            d[0] = dout.float()
        dpre = to_cg8p(d)
        for b in range(4, -1, -1):
            zin, zh, zw, S, b1, b2, tH, tW = ctx.dec_rec[b]
            i1, i2 = 10 + 2 * b, 11 + 2 * b
            wgrad(i2, dpre, b1, tH, tW)
            dpre1 = cg8p_alloc(L[i1].cout_pad, tH, tW, dev)
            _conv(lib, dpre, Wb[i2], None, b1, dpre1, tH, tW, L[i2].cout_pad, L[i2].cin_pad, 1, s)     # * lrelu'(b1)
            wgrad(i1, dpre1, S, tH, tW)
            dS = cg8p_alloc(L[i1].cin_pad, tH, tW, dev)
            _conv(lib, dpre1, Wb[i1], zeros_bias, None, dS, tH, tW, L[i1].cout_pad, L[i1].cin_pad, 2, s)
            dzin = cg8p_alloc(L[i1].cin_pad, zh, zw, dev)
            # the block's input is the previous decoder block's LeakyReLU output (mask) or the latent z (no activation)
            lib.check(lib.stuff2_bwd(ptr(dS), tH, tW, ptr(zin) if b > 0 else None, ptr(dzin), zh, zw, L[i1].cin_pad, s), 'stuff2_bwd')
            dpre = dzin
        if dz is not None and ctx.need_z:
            dzp = torch.zeros(256, zH, zW, dtype=torch.float32, device=dev)
            dzp[:] = dz.float()
            dpre = dpre + to_cg8p(dzp)
        # ---- encoder, last block first.  dpre = d(pool output of the block)
        for b in range(4, -1, -1):
            xin, a0, a2, idx, h, w = ctx.enc_rec[b]
            i0, i2 = 2 * b, 2 * b + 1
            dpre2 = cg8p_alloc(L[i2].cout_pad, h, w, dev)
            lib.check(lib.maxpool3s2_bwd(ptr(dpre), ptr(idx), ptr(a2), ptr(dpre2), h, w, L[i2].cout_pad, s), 'maxpool3s2_bwd')
            wgrad(i2, dpre2, a0, h, w)
            dpre0 = cg8p_alloc(L[i0].cout_pad, h, w, dev)
            _conv(lib, dpre2, Wb[i2], None, a0, dpre0, h, w, L[i2].cout_pad, L[i2].cin_pad, 1, s)
            wgrad(i0, dpre0, xin, h, w)
            if b > 0:
                dprev = cg8p_alloc(L[i0].cin_pad, h, w, dev)
                _conv(lib, dpre0, Wb[i0], zeros_bias, None, dprev, h, w, L[i0].cout_pad, L[i0].cin_pad, 2, s)
                dpre = dprev
        if side is not None:
            main.wait_stream(side)
        lib.check(lib.conv3x3_wgrad_reduce_multi(jobs, njobs[0], s), 'conv3x3_wgrad_reduce_multi')
        return None, None, dwdb[T.idx_grad], None, None


class _Conv(nn.Module):
    def __init__(self, cin, cout, transposed=False):
        super().__init__()
        shape = (cin, cout, 3, 3) if transposed else (cout, cin, 3, 3)
        self.weight = nn.Parameter(torch.zeros(*shape))
        self.bias = nn.Parameter(torch.zeros(cout))


class _EncBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.main = nn.ModuleDict({'0': _Conv(cin, cout), '2': _Conv(cout, cout)})


class _DecBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.deconv1, self.deconv2 = _Conv(cin, cout, True), _Conv(cout, cout, True)


class AE(nn.Module):
    """Drop-in for ``models/AE.py::AE(downsample=True, in_channel=4, kernel=3)``: ``forward(x[1,4,d,T]) -> (out[1,1,d,T], z)``."""

    def __init__(self, downsample=True, in_channel=4, kernel=3, _lib: Optional[_hip.HipLib] = None):
        super().__init__()
        if not downsample or in_channel != 4 or kernel != 3:
            raise NotImplementedError('LEMO instantiates the infilling prior as AE(downsample=True, in_channel=4, kernel=3)')
        for b, (ci, co) in enumerate(ENC_CH, 1):
            setattr(self, f'enc_blc{b}', _EncBlock(ci, co))
        for b, (ci, co) in enumerate(DEC_CH, 1):
            setattr(self, f'dec_blc{b}', _DecBlock(ci, co))
        self._lib_override = _lib

    def ordered_parameters(self) -> List[nn.Parameter]:
        sd = dict(self.named_parameters())
        out = []
        for l in _layers():
            out += [sd[l.name + '.weight'], sd[l.name + '.bias']]
        return out

    def forward(self, x):
        assert x.dim() == 4 and x.shape[0] == 1 and x.shape[1] == 4, 'the infilling prior is fed [1,4,d,T] clips'
        lib = self._lib_override or _hip.get_lib()
        out, z = _AEFn.apply(lib, x[0], flatten_params(self.ordered_parameters()))
        return out[None, None], z[None]


class FlatAdam:
    """torch.optim.Adam-equivalent (defaults, no weight decay) running ONE HIP kernel over a flat copy of the
    parameters: the finetune optimiser of opt_amass_temp.py:162-164 (lr 3e-6)."""

    def __init__(self, params: List[nn.Parameter], lr: float, lib: Optional[_hip.HipLib] = None):
        self.params, self.lr, self.lib, self.t = [p for p in params if p.requires_grad], lr, lib or _hip.get_lib(), 0
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        # a single flat fp32 leaf is updated in place (no gather of gradients, no copy back)
        self.inplace = len(self.params) == 1 and self.params[0].dim() == 1 and self.params[0].dtype == torch.float32
        self.flat = self.params[0].data if self.inplace else torch.cat([p.detach().reshape(-1).float() for p in self.params])
        self.m, self.v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        self.step_ctr = torch.zeros(1, dtype=torch.int32, device=dev)      # completed steps (device copy: graph replay)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        self.t += 1
        if self.inplace:
            g = self.params[0].grad.contiguous()
            self.lib.check(self.lib.adam_flat_ctr(ptr(self.flat), ptr(g), ptr(self.m), ptr(self.v), self.flat.numel(), self.lr,
                                                  ptr(self.step_ctr), self.lib.stream(self.flat.device)), 'adam_flat')
            return
        g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in self.params])
        self.lib.check(self.lib.adam_flat_ctr(ptr(self.flat), ptr(g), ptr(self.m), ptr(self.v), self.flat.numel(), self.lr,
                                              ptr(self.step_ctr), self.lib.stream(self.flat.device)), 'adam_flat')
        o = 0
        for p in self.params:
            p.copy_(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()


class _FinetuneSession:
    """Everything one finetune loop needs, at FIXED device addresses, kept across clips of the same shape: the flat parameter
    leaf and its Adam state, static copies of the clip image and of mask / count, the step's workspace, the stream all of it
    runs on and -- after the first clip -- the captured graph of one training step.  A clip then costs a few small copies
    into the static buffers plus ``steps`` replays; capture + instantiation (~10 ms) is paid once per process and shape,
    not once per clip."""

    def __init__(self, lib, n_param, x_shape, lr, device):
        self.lib, self.lr, self.device = lib, lr, torch.device(device)
        self.gpu = self.device.type == 'cuda' and not lib.is_emu
        self.stream = torch.cuda.Stream(self.device) if self.gpu else None
        with self._on():
            # The step trains ONE flat leaf created on the stream the step runs on.  Autograd keeps one AccumulateGrad node
            # per leaf and pins it to the stream it was first used on: with the module's own parameters, a caller that still
            # holds an output of an earlier forward (made on another stream) forces a cross-stream event wait into every
            # backward -- fatal inside a stream capture (segfault in hipStreamEndCapture, tools/graph_repro.py).
            self.flat = torch.zeros(n_param, dtype=torch.float32, device=self.device).requires_grad_(True)
            self.opt = FlatAdam([self.flat], lr, lib)
            self.ws = AEWorkspace(self.device)           # the step's ~45 activation buffers: zeroed once, not per step
            self.x = torch.zeros(x_shape, dtype=torch.float32, device=self.device)
            self.moc = torch.zeros(x_shape[-2:], dtype=torch.float32, device=self.device)     # mask / count
        self.exe, self.pool = None, None

    def _on(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.gpu else contextlib.nullcontext()

    def __del__(self):
        exe, self.exe = getattr(self, 'exe', None), None
        if exe is not None:
            lib, rel = self.lib, getattr(_hip, 'release', None) if _hip is not None else None
            if rel is not None:              # (None: interpreter shutdown)
                rel(self.device, lib, lambda: lib.graph_destroy(exe))

    def train_step(self):
        # loss = (|rec - x| * m).sum() / cnt (opt_amass_temp.py:199-203); its gradient is closed-form, so the eleven small
        # launches of the loss and its autograd tape are three (same bits: +-1 / 0 times m / cnt)
        self.opt.zero_grad()
        rec, _ = _AEFn.apply(self.lib, self.x[0], self.flat, self.ws, False)
        rec.backward(torch.sign(rec.detach() - self.x[0, 0]) * self.moc)
        self.opt.step()

    def run(self, flat0, x, m_over_cnt, steps, use_graph, join=True):
        import ctypes as C
        lib = self.lib
        if self.gpu:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with self._on(), torch.no_grad():
            self.flat.copy_(flat0)                            # a fresh optimiser on the pretrained weights (:160-164)
            self.opt.m.zero_(); self.opt.v.zero_(); self.opt.step_ctr.zero_(); self.opt.t = 0
            self.x.copy_(x); self.moc.copy_(m_over_cnt)
        with self._on():
            left = steps
            if use_graph and self.exe is None and steps > 3:
                for _ in range(3):                           # eager: warms the allocator cache and every lazy init
                    self.train_step()
                left -= 3
                # After three identical steps every allocation of the step is served from the caching allocator and
                # everything runs on this stream (+ the workspace's second one, forked and joined inside the step), so the
                # recorded addresses stay valid.  Raw capture (lemo_capture_*).  The captured step allocates from a PRIVATE
                # pool that lives as long as the graph (the replays write into those addresses; the shared caching
                # allocator could hand them out again between two replays).
                self.opt.zero_grad()
                sh = self.stream.cuda_stream
                self.pool = torch.cuda.MemPool()
                with torch.cuda.use_mem_pool(self.pool):
                    lib.check(lib.capture_begin(sh), 'capture_begin')
                    try:
                        self.train_step()
                    finally:
                        exe = C.c_void_p()
                        rc = lib.capture_end(sh, C.byref(exe))
                lib.check(rc, 'capture_end')
                self.exe = exe
            if use_graph and self.exe is not None:
                for _ in range(left):                        # capture records the step without running it
                    lib.check(lib.graph_launch(self.exe, self.stream.cuda_stream), 'graph_launch')
            else:
                for _ in range(left):
                    self.train_step()
        if self.gpu and join:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return self.flat.detach()

    def join(self):
        """make the current stream wait for this session's work (``run(..., join=False)`` leaves that to the caller, so that
        several sessions can be enqueued before anything waits for any of them)"""
        if self.gpu:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)


class _EngineSession:
    """One clip shape's native finetune engine (``lemo_ae_*``, csrc/ae_engine.hip): the workspace it carves its parameters,
    Adam state, activations and gradients from, its stream, and the output buffers of the eval forward.  Kept across clips
    of the same shape like :class:`_FinetuneSession` (the captured graphs live inside the engine)."""

    def __init__(self, lib, x_shape, lr, device, clips: int = 1):
        # clips: K clips side by side in every launch (lemo_ae_desc.clips): K workspaces back to back, one engine, one stream
        self.lib, self.device, self.clips = lib, torch.device(device), int(clips)
        self.gpu = self.device.type == 'cuda' and not lib.is_emu
        H, W = int(x_shape[-2]), int(x_shape[-1])
        n = int(lib.ae_ws_floats(H, W))
        if n <= 0:
            raise _hip.LemoHipError(f'lemo_ae_ws_floats refuses a {H} x {W} clip image')
        self.stream = torch.cuda.Stream(self.device) if self.gpu else None       # used when the caller sits on the legacy default stream
        self.ws = torch.zeros(n * self.clips, dtype=torch.float32, device=self.device)       # zero borders = the convolutions' padding
        self.ws_bytes = 4 * n * self.clips
        h5, w5 = H, W
        for _ in range(5):
            h5, w5 = (h5 - 1) // 2 + 1, (w5 - 1) // 2 + 1
        self.rec = torch.empty(self.clips, H, W, dtype=torch.float32, device=self.device)
        self.z = torch.empty(self.clips, 256, h5, w5, dtype=torch.float32, device=self.device)
        self.flat = torch.empty(self.clips, int(lib.ae_n_param()), dtype=torch.float32, device=self.device)
        if self.gpu:
            torch.cuda.current_stream(self.device).synchronize()            # the zero fill, before another stream uses ws
        desc = _hip.AeDesc(H, W, float(lr), ptr(self.ws), n * self.clips, self.clips)
        self.h = lib.ae_create(C.byref(desc))
        if not self.h:
            raise _hip.LemoHipError('lemo_ae_create failed')
        self._ev = None

    def __del__(self):
        h, self.h = getattr(self, 'h', None), None
        if h:
            lib, rel = self.lib, getattr(_hip, 'release', None) if _hip is not None else None
            if rel is not None:              # (None: interpreter shutdown)
                def destroy(h=h, ws=self.ws):            # (the workspace outlives the engine's last launch)
                    lib.ae_destroy(h)
                rel(self.device, lib, destroy, self._ev)

    def _s(self):
        return self.stream.cuda_stream if self.gpu else None

    def _on(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.gpu else contextlib.nullcontext()

    def run(self, flat0, x, m_over_cnt, steps, use_graph, join=True, own_stream=False):
        """one clip on a one-clip session (:meth:`run_clips`); returns ``(flat, rec, z)`` views of the session's output buffers"""
        flat, rec, z = self.run_clips(flat0, [x], [m_over_cnt], steps, use_graph, join, own_stream)
        return flat[0], rec[0], z[0]

    def run_clips(self, flat0, xs, mocs, steps, use_graph, join=True, own_stream=False):
        """load the pretrained parameters and ``self.clips`` clips, `steps` training steps of all of them (each launch carries
        every clip), eval forward, parameters back out; returns views ``(flat [K, n], rec [K, H, W], z [K, 256, h, w])`` of the
        session's output buffers (valid until its next run).  Everything runs on the CALLER's current stream when that is not the
        legacy default stream (which cannot be captured) -- no hand-over between two hardware queues, which measured 29.8 instead
        of 33.4 ms per clip (``profiles/r03_hw_queues.txt``; the graphs are not bound to a stream); ``own_stream=True`` (or a caller
        on the default stream): on this session's stream, the caller's stream joined afterwards unless ``join=False``."""
        lib = self.lib
        assert len(xs) == len(mocs) == self.clips
        flat0 = flat0.contiguous().float()
        xs = [x.reshape(4, *x.shape[-2:]).contiguous().float() for x in xs]
        mocs = [m.contiguous().float() for m in mocs]
        assert flat0.numel() == self.flat.shape[1] and all(tuple(m.shape) == tuple(self.rec.shape[1:]) for m in mocs)
        run_on, sh = self.stream, self._s()
        if self.gpu:
            cur = torch.cuda.current_stream(self.device)
            if not own_stream and cur != torch.cuda.default_stream(self.device):
                run_on, sh = cur, cur.cuda_stream
            else:
                self.stream.wait_stream(cur)
            if self._ev is not None:
                # ALWAYS order the stream about to be used after the session's previous run: a run on caller
                # stream A followed by a lane / default-stream run used to wait for `cur` only, and ae_load / ae_step then
                # overwrote the workspace and rec / z / flat that the previous run (or its output clones on A) still used
                run_on.wait_event(self._ev)
            _hip.flush_deferred()
        for c, (x, moc) in enumerate(zip(xs, mocs)):
            lib.check(lib.ae_load_clip(self.h, c, ptr(flat0), ptr(x), ptr(moc), sh), 'ae_load_clip')
        lib.check(lib.ae_step(self.h, int(steps), 1 if (use_graph and self.gpu) else 0, sh), 'ae_step')
        for c in range(self.clips):                          # (clip 0 runs the eval forward of all clips)
            lib.check(lib.ae_forward_clip(self.h, c, ptr(self.rec[c]), ptr(self.z[c]), sh), 'ae_forward_clip')
            lib.check(lib.ae_params_clip(self.h, c, ptr(self.flat[c]), sh), 'ae_params_clip')
        if self.gpu:
            if self._ev is None:
                self._ev = torch.cuda.Event()
            self._ev.record(run_on)
            if run_on is self.stream:
                for t in [flat0] + xs + mocs:                # read by the launches above on this session's stream
                    t.record_stream(self.stream)
                if join:
                    self.join()
        return self.flat, self.rec, self.z

    def join(self):
        if self.gpu:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)


AE_CLIPS = max(1, min(16, int(__import__('os').environ.get('LEMO_AE_CLIPS', '8'))))
"""clips per engine in ``finetune_and_infill_many``: every launch of a training step carries this many clips (round 4; measured
on 210 x 135 clip images, ms per clip: 1 clip 29.0 | 2 23.8 | 3 22.6 | 4 20.6 | 8 18.3 -- ``profiles/r04_ae_clips.txt``; the kernels
of a step are throughput-bound from ~4 clips on, a clip's workspace is ~50 MB)"""

USE_ENGINE = __import__('os').environ.get('LEMO_AE_ENGINE', '1') != '0'
"""the finetune loop runs on the native step engine (round 3: 53 launches per step instead of ~150; csrc/ae_engine.hip).
``LEMO_AE_ENGINE=0`` (or ``engine=False``) keeps the round-2 path -- the autograd function + flat Adam under a captured graph
(:class:`_FinetuneSession`) -- which is also what ``AE.forward`` under autograd runs; tests compare the two."""

_SESSIONS: Dict[tuple, object] = {}
_MAX_SESSIONS = 8
"""sessions kept alive at once (LRU): each pins a workspace (or ~45 workspace buffers), a stream and instantiated graphs; PROX
tail windows and recordings of varying length would otherwise add one per clip shape without bound (ADVICE r02)."""
_MAX_SESSION_BYTES = 4 << 30
"""... and the bytes their workspaces may pin together (ADVICE r04: an engine session holds clips x ~160 MB at 210 x 135; counting
sessions alone let eight 8-clip engines of different shapes hold 10 GB).  The most recent session always stays."""


def _engine_groups(n: int):
    """split a run of n same-shape clips (n <= AE_CLIPS) into engine launches ``[(clips taken, engine size)]``: engine sizes are powers of
    two (1, 2, 4, 8, 16: at most five engine sizes and captured graph sets per clip shape instead of one per distinct tail length,
    ADVICE r04); a tail is padded up to the next power of two -- its last clip repeated in the spare slots, outputs dropped -- when that
    wastes at most one slot in eight (7 -> 8, 15 -> 16), and split otherwise (5 -> 4 + 1, 3 -> 2 + 1: a padded 3-on-4 engine measured
    25.1 ms per clip against 22.6 for 2 + 1)."""
    out = []
    while n > 0:
        k = 1
        while 2 * k <= n:
            k *= 2
        if k < n and 2 * k - n <= max(2 * k // 8, 0):
            out.append((n, 2 * k))
            n = 0
        else:
            out.append((k, k))
            n -= k
    return out


def _session(lib, n_param: int, shape, lr: float, device, slot: int = 0, engine: bool = False, clips: int = 1):
    key = (str(device), tuple(shape), float(lr), id(lib), int(slot), bool(engine), int(clips))
    ses = _SESSIONS.pop(key, None)
    if ses is None:
        if engine:
            ses = _EngineSession(lib, tuple(shape), lr, device, clips=clips)
        else:
            ses = _FinetuneSession(lib, n_param, tuple(shape), lr, device)
    _SESSIONS[key] = ses                                   # most recently used last
    pinned = lambda: sum(int(getattr(v, 'ws_bytes', 0)) for v in _SESSIONS.values())
    while len(_SESSIONS) > 1 and (len(_SESSIONS) > _MAX_SESSIONS or pinned() > _MAX_SESSION_BYTES):
        _SESSIONS.pop(next(iter(_SESSIONS)))               # its destructor waits for its last launch and releases the graphs
    return ses


def _store_params(model: 'AE', flat: torch.Tensor):
    with torch.no_grad():
        o = 0
        for p in model.ordered_parameters():
            p.copy_(flat[o:o + p.numel()].view_as(p))
            o += p.numel()


def finetune_and_infill(model: AE, weights: dict, clip_img_input: torch.Tensor, train_mask: torch.Tensor, steps: int = 60,
                        lr: float = 3e-6, use_graph: Optional[bool] = None, engine: Optional[bool] = None):
    """The per-clip block of opt_amass_temp.py:160-215: reload the pretrained weights, ``steps`` x [forward, L1 on
    ``train_mask`` (bool [d+2, T+16] over channel 0), backward, Adam], then one eval forward.  Returns
    ``(clip_img_rec [1,1,d,T] un-padded, z)``; the model holds the finetuned weights afterwards, as in the reference.

    ``engine`` (default :data:`USE_ENGINE`): the whole loop runs inside the native step engine (``lemo_ae_*``: 53 launches per
    step, captured into graphs on first use).  ``engine=False`` is the round-2 path: the training step -- ~110 HIP kernels plus
    ~40 small packing ops through the autograd function -- captured ONCE per process and clip shape into a graph (after 3 eager
    steps that also warm the allocator) and replayed (:class:`_FinetuneSession`); Adam's bias-correction step lives on the
    device (``lemo_adam_flat_ctr``) so the replays advance it.  ``use_graph`` (default: on a HIP device) applies to both; same
    kernels, same order: results are bit-identical to eager launches (tested)."""
    model.load_state_dict(weights)
    lib = model._lib_override or _hip.get_lib()
    m = train_mask.to(clip_img_input.dtype)
    cnt = m.sum()
    if use_graph is None:
        use_graph = clip_img_input.is_cuda
    use_graph = bool(use_graph) and clip_img_input.is_cuda and not lib.is_emu
    engine = USE_ENGINE if engine is None else bool(engine)
    flat0 = flatten_params([p.detach() for p in model.ordered_parameters()])
    ses = _session(lib, flat0.numel(), clip_img_input.shape, lr, clip_img_input.device, engine=engine)
    if engine:
        _hip.check_device(lib, clip_img_input)
        flat, rec, z = ses.run(flat0, clip_img_input, m * (1.0 / cnt), steps, use_graph)   # d(loss)/d(rec) = sign(rec - x) * m / cnt
        _store_params(model, flat)
        return rec[None, None, 1:-1, 8:-8].clone(), z[None].clone()
    flat = ses.run(flat0, clip_img_input, m * (1.0 / cnt), steps, use_graph)
    _store_params(model, flat)
    with torch.no_grad():
        rec, z = model(clip_img_input)
    return rec[:, :, 1:-1, 8:-8], z


def finetune_and_infill_many(model: AE, weights: dict, clips: List[torch.Tensor], train_masks: List[torch.Tensor], steps: int = 60,
                             lr: float = 3e-6, use_graph: Optional[bool] = None, engine: Optional[bool] = None):
    """:func:`finetune_and_infill` for several clips SIDE BY SIDE (the dataset-scale form, like
    ``lemo_amd.sharding.ConcurrentClips`` for the temporal fit and ``BatchedPerFrameFitter`` for stage 1).  Clips of one shape go
    ``AE_CLIPS`` at a time into ONE engine whose every launch carries all of them -- the clip is the last grid dimension of the
    convolution, pooling, weight-gradient and Adam launches; each clip has its own parameters, Adam state, step counter and
    workspace slice (``lemo_ae_desc.clips``; the reference finetunes a fresh copy of the pretrained model per clip).  One training
    step of one clip is 53 short launches whose time is mostly launch latency and tail, so K clips per launch cost far less than
    K steps: 29.0 ms for one clip alone -> 20.6 ms per clip with four -> 18.3 with eight (``profiles/r04_ae_clips.txt``; from there
    the step's kernels are throughput-bound: their summed time is 18.1 ms per clip).  (Round 3 overlapped two one-clip
    engines on two streams instead -- 24 ms per clip when the runtime happened to put the streams into different hardware queues,
    33 when not; gone.)  Everything is enqueued on the caller's stream (the session's own stream when the caller sits on the legacy
    default stream, which cannot be captured).  A group smaller than ``AE_CLIPS`` (the tail, or clips of a
    shape of their own) runs on power-of-two engines (:func:`_engine_groups`).  Each clip's result equals its solo
    ``finetune_and_infill`` to rounding (round 5: the convolutions' launch shapes are chosen for the clips in flight, 18.0 -> 16.4 ms
    per clip at 8; bit-identical for equal grouping; tested).  Returns the list of
    ``(clip_img_rec, z)`` in input order; the model is left with the LAST clip's finetuned weights.  (``engine=False``: the
    round-2 path, one session per clip, at most ``_MAX_SESSIONS`` clips.)"""
    assert 1 <= len(clips) == len(train_masks)
    lib = model._lib_override or _hip.get_lib()
    if use_graph is None:
        use_graph = clips[0].is_cuda
    use_graph = bool(use_graph) and clips[0].is_cuda and not lib.is_emu
    engine = USE_ENGINE if engine is None else bool(engine)
    model.load_state_dict(weights)
    flat0 = flatten_params([p.detach() for p in model.ordered_parameters()])
    mocs = [tm.to(x.dtype) * (1.0 / tm.to(x.dtype).sum()) for x, tm in zip(clips, train_masks)]     # (on the current stream, before any fork)
    if engine:
        out, last_flat = [None] * len(clips), None
        by_shape: Dict[tuple, List[int]] = {}
        for i, x in enumerate(clips):
            by_shape.setdefault((str(x.device), tuple(x.shape)), []).append(i)
        groups = []
        for idx in by_shape.values():
            for j in range(0, len(idx), AE_CLIPS):
                run_, o = idx[j:j + AE_CLIPS], 0
                for take, size in _engine_groups(len(run_)):
                    groups.append((run_[o:o + take], size))
                    o += take
        for grp, k in groups:
            x0 = clips[grp[0]]
            run = list(grp) + [grp[-1]] * (k - len(grp))          # spare slots repeat the last clip; their outputs are dropped
            ses = _session(lib, flat0.numel(), x0.shape, lr, x0.device, engine=True, clips=k)
            flat, rec, z = ses.run_clips(flat0, [clips[i] for i in run], [mocs[i] for i in run], steps, use_graph)
            # the session's next run overwrites its buffers: copies (ordered after the run: same stream, or joined)
            r, zz = rec[:, None, None, 1:-1, 8:-8].clone(), z[:, None].clone()
            if grp[-1] == len(clips) - 1:
                last_flat = flat[len(grp) - 1].clone()
            for c, i in enumerate(grp):
                out[i] = (r[c], zz[c])
        _store_params(model, last_flat)
        return out
    assert len(clips) <= _MAX_SESSIONS
    results, sessions = [], []
    for i, (x, moc) in enumerate(zip(clips, mocs)):                          # enqueue: clip i's loop on session i's stream
        ses = _session(lib, flat0.numel(), x.shape, lr, x.device, slot=i, engine=False)
        results.append(ses.run(flat0, x, moc, steps, use_graph, join=False))   # the current stream waits for nobody yet
        sessions.append(ses)
    for ses in sessions:
        ses.join()
    out = []
    for x, flat in zip(clips, results):                                      # eval forwards, one after the other
        _store_params(model, flat)
        with torch.no_grad():
            rec, z = model(x)
        out.append((rec[:, :, 1:-1, 8:-8].clone(), z.clone()))
    return out
