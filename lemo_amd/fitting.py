"""AMASS temporal fitting (stage 2 of LEMO) on the native MI355X engine.

Reference: ``opt_amass_temp.py::optimize`` -- per-clip setup :332-345, the 100-step Adam loop
:349-455 (the hot path), result :457-458.  One :class:`AmassTemporalFitter` owns every device
buffer of one sequence (parameters, Adam state, ~0.5 GB of workspace for B=119) and hands raw
pointers to ``liblemo_hip.so`` once (``lemo_fit_create``); an iteration is then a single C call
(``lemo_fit_step``) that replays captured hipGraphs (one node per kernel of the iteration) -- no host sync, no
``.item()`` (the reference has 4 per iteration, :431-443), SMPL-X evaluated once instead of twice (:357,:364).

:class:`PerFrameFitter` is the stage-1 twin (``opt_amass_perframe.py:293-355``, BASELINE configs[0]): B = 1, marker
L1 + the three L2 priors, frames fitted one after the other, each warm-started from the previous one.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _hip
from ._hip import ptr
from .body_model import BodyModelData, DeviceBody, alloc_pose_ws, K_PAD, load_model_dict
from .priors import DEFAULT_CONV_VARIANT, ENC_CHANNELS, EncWeights, cg8p_alloc
from .rotation import convert_to_6D_all
from .vposer import vposer_weight_struct

LOSS_WEIGHTS = dict(rec_markers=1.0, vposer=0.02, shape=0.01, hand=0.01, contact_vel=0.03, smooth=1e6)
"""opt_amass_temp.py:47-52 (order = the C ABI's ``weights[6]``)."""
FOOT_SETS = ('left_heel', 'right_heel', 'left_toe', 'right_toe')
"""columns of ``contact_lbl`` (opt_amass_temp.py:409-412)."""
LOSS_NAMES = ('marker', 'vposer', 'shape', 'hand', 'contact', 'smooth', 'total')


DEFAULT_SIDE_FULL_FORWARD = __import__('os').environ.get('LEMO_SIDE_FULL_FORWARD', '0') != '0'
"""all-vertex forward on the engine's side stream (AmassTemporalFitter.side_full_forward); the environment switch exists for A/B runs"""


class AmassTemporalFitter(_hip.StreamOrdered):
    def __init__(self, body, vposer_weights: Dict[str, np.ndarray], enc_state: Dict[str, np.ndarray],
                 ids: Dict[str, np.ndarray], Xmean: np.ndarray, Xstd: np.ndarray, B: int, device,
                 weights: Optional[dict] = None, full_vertices: bool = True, num_pca_comps: int = 12,
                 lr0: float = 0.01, lr1: float = 0.005, lr_switch: int = 60, conv_variant: Optional[int] = None,
                 lbs_blend_fp32: bool = False, per_frame: bool = False, lr2: float = 0.0, lr_switch2: int = 0,
                 lib: Optional[_hip.HipLib] = None, side_full_forward: Optional[bool] = None):
        """``side_full_forward`` (round 6, with ``full_vertices``): every iteration still regresses all V vertices (``vertices()`` returns them), but by a launch on
        the engine's side stream beside the per-frame launches that close the iteration instead of in front of the encoder; the loss path forwards the 253
        vertices it reads.  Same kernels, same bits (lemo_fit_desc.verts_side).  None = ``DEFAULT_SIDE_FULL_FORWARD``."""
        self.lib = lib or _hip.get_lib()
        self.device = torch.device(device)
        if not self.lib.is_emu and self.device.type != 'cuda':
            raise _hip.LemoHipError('AmassTemporalFitter needs a HIP device (no CPU fallback)')
        self.B, self.full = int(B), bool(full_vertices)
        if side_full_forward is None:
            side_full_forward = DEFAULT_SIDE_FULL_FORWARD
        self.side_full = bool(side_full_forward) and self.full and not per_frame
        self.full_output = self.full                         # what vertices() returns: every vertex of the model
        if self.side_full:
            self.full = False                                # the loss path: the set U only (index tables, verts / v_posed layouts, backward set)
        data = body if isinstance(body, BodyModelData) else BodyModelData(load_model_dict(body), num_pca_comps=num_pca_comps)
        assert data.ncomp == 12, 'the AMASS parameter vector carries 12 PCA coefficients per hand'
        self.data = data
        self.dev = DeviceBody(data, self.device)
        self.dev.skin.blend_fp32 = int(bool(lbs_blend_fp32))     # 0: split-bf16 blend GEMM (default); 1: fp32 MFMA
        dev = self.device
        ti = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.int32)).to(dev)
        tf = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)

        # ---- active vertex set U and the index tables of the loss kernels ---------------------
        m67, m81 = np.asarray(ids['markers67'], np.int64), np.asarray(ids['markers81'], np.int64)
        foot = [np.asarray(ids[k], np.int64) for k in FOOT_SETS]
        U = np.unique(np.concatenate([m67, m81] + foot))
        self.U, n = U, int(U.shape[0])
        slot = {int(v): i for i, v in enumerate(U)}
        row_of = (lambda v: int(v)) if self.full else (lambda v: slot[int(v)])
        self.nrows = data.V if self.full else n
        u_m67 = -np.ones(n, np.int32); u_m81 = -np.ones(n, np.int32); u_mask = np.zeros(n, np.int32)
        for i, v in enumerate(m67): u_m67[slot[int(v)]] = i
        for i, v in enumerate(m81): u_m81[slot[int(v)]] = i
        for k, f in enumerate(foot):
            for v in f: u_mask[slot[int(v)]] |= (1 << k)
        assert len(set(m67.tolist())) == len(m67) and len(set(m81.tolist())) == len(m81), 'duplicate marker ids'
        for k, f in zip(FOOT_SETS, foot):    # a repeated id would count twice in the loss but once in the gradient mask
            assert len(set(f.tolist())) == len(f), f'duplicate vertex id in foot set {k}'
        self.per_frame = bool(per_frame)
        if self.per_frame:
            u_m81[:] = -1                      # no smoothness term: the gradient kernels never touch the marker image
        foot_start = np.cumsum([0] + [len(f) for f in foot]).astype(np.int32)
        self._idx = dict(row67=ti([row_of(v) for v in m67]), row81=ti([row_of(v) for v in m81]),
                         foot_start=ti(foot_start), foot_row=ti([row_of(v) for f in foot for v in f]),
                         u_row=ti([row_of(v) for v in U]), u_m67=ti(u_m67), u_m81=ti(u_m81), u_mask=ti(u_mask),
                         Xstd=tf(np.asarray(Xstd).reshape(-1)), Xmean=tf(np.asarray(Xmean).reshape(-1)),
                         fwd_ids=ti(U))
        self.n, self.n67, self.n81 = n, len(m67), len(m81)
        assert self._idx['Xstd'].numel() == 3 * self.n81
        I = self._idx
        fit = _hip.FitConst(n, self.n67, self.n81, ptr(I['row67']), ptr(I['row81']), ptr(I['foot_start']),
                            ptr(I['foot_row']), ptr(I['u_row']), ptr(I['u_m67']), ptr(I['u_m81']), ptr(I['u_mask']),
                            ptr(I['Xstd']), ptr(I['Xmean']))
        vp_row = U if self.full else np.arange(n)
        uset, self._uset_t = self.dev.vertex_set(('fit', self.full), U, vp_row)

        # ---- model weights ----------------------------------------------------------------
        self.vposer_struct, self._vp_t = vposer_weight_struct(vposer_weights, dev)
        self.enc = EncWeights(enc_state, dev)
        w = dict(LOSS_WEIGHTS if weights is None else weights)
        self.weights = w
        wl = [w['rec_markers'], w['vposer'], w['shape'], w['hand'], w['contact_vel'], w['smooth']]
        self._w_dev = tf(wl)

        # ---- parameters, Adam state, workspace ---------------------------------------------
        B, nj = self.B, data.nj
        self.H, self.W = 3 * self.n81 + 2, B - 1 + 16
        H, W = self.H, self.W
        self.P = dict(transl=z(B, 3), rot6d=z(B, 6), other=z(B, 56), shape=z(B, 10))
        self.adam_m = [z(B, 3), z(B, 6), z(B, 56)]
        self.adam_v = [z(B, 3), z(B, 6), z(B, 56)]
        self.step_ctr = torch.zeros(1, dtype=torch.int32, device=dev)
        self.step_cur = torch.zeros(1, dtype=torch.int32, device=dev)
        self.loss_acc = torch.zeros(32 * 16, dtype=torch.float64, device=dev)
        self.target, self.contact = z(B, self.n67, 3), z(B, 4)
        pose_ws, self._pose_t, Bp = alloc_pose_ws(B, nj, dev, self.dev.blend_f16)
        self.Bp = Bp
        nsp = self.lib.smooth_loss_blocks(H, W, ENC_CHANNELS[10])
        self.ws = dict(go_aa=z(B, 3), body_aa=z(B, 63), h1=z(B, 512), h2=z(B, 512), vo=z(B, 128), vp_scratch=z(B, 1152),
                       verts=z(B, self.nrows, 3), v_posed=z(B, self.nrows, 3), x0=z((H + 2) * (W + 2)), canon=z(12),
                       dx0=z(H * W), spartial=z(nsp), vpartial=z(B, 9), losses=z(12), dverts=z(B, n, 3),
                       dvp=z(B, uset.NCs), dA=z(B, nj, 12), dX=z(B, K_PAD),
                       g_transl=z(B, 3), g_rot6d=z(B, 6), g_other=z(B, 56), g_go=z(B, 3), g_body=z(B, 63))
        self.act = [None] + [cg8p_alloc(ENC_CHANNELS[l], H, W, dev) for l in range(1, 11)]
        self.dact = [cg8p_alloc(64, H, W, dev), cg8p_alloc(64, H, W, dev)]

        d = _hip.FitDesc()
        d.B, d.Bp, d.V, d.nrows, d.full_vertices = B, Bp, data.V, self.nrows, int(self.full)
        if conv_variant is None:
            conv_variant = DEFAULT_CONV_VARIANT
        if int(conv_variant) in (2, 3, 4) and 127 + 2 * (127 // self.W + 1) + 2 * (self.W + 2) + 3 > 416:
            conv_variant = 1                      # LDS tile of variant 2 holds W <= 139 (B <= 124); the single-layer split kernels W <= 134
        # (variant 5 stays: the fused pairs take any width -- 12 of the 14 64 -> 64 layers; the engine sends the other launches to the
        # fp32-input kernel layer by layer, lemo_amd/csrc/enc_chain.hpp::enc_layer.  Round 4: all 18 launches used to fall back.)
        from .priors import check_conv_variant
        self.conv_variant = d.conv_variant = check_conv_variant(conv_variant)
        if not self.per_frame:
            from .priors import warn_if_wide_image
            warn_if_wide_image(self.lib, H, W, self.conv_variant)
        d.vposer, d.body, d.skin, d.uset, d.fit = self.vposer_struct, self.dev.body, self.dev.skin, uset, fit
        d.fwd_ids = ptr(I['fwd_ids'])
        for i, c in enumerate(ENC_CHANNELS): d.enc_ch[i] = c
        for l in range(10):
            d.enc_w[l], d.enc_b[l], d.enc_wbwd[l] = ptr(self.enc.w[l]), ptr(self.enc.b[l]), ptr(self.enc.wbwd[l])
            d.enc_w2[l], d.enc_wbwd2[l] = ptr(self.enc.w2[l]), ptr(self.enc.wbwd2[l])
            for bwd, dst, dinv in ((False, d.enc_w3, d.enc_w3_inv), (True, d.enc_wbwd3, d.enc_wbwd3_inv)):
                pack, winv = self.enc.split_pack(l, bwd, self.conv_variant)         # bf16 x 3 (variant 3) or f16 x 2 (variant 4)
                dst[l], dinv[l] = (ptr(pack) if pack is not None else None), float(winv)
        d.target, d.contact, d.weights = ptr(self.target), ptr(self.contact), ptr(self._w_dev)
        for i, v in enumerate(wl): d.weights_host[i] = v
        d.transl, d.rot6d, d.other, d.shape = (ptr(self.P[k]) for k in ('transl', 'rot6d', 'other', 'shape'))
        for i in range(3):
            d.adam_m[i], d.adam_v[i] = ptr(self.adam_m[i]), ptr(self.adam_v[i])
        d.step_ctr, d.lr0, d.lr1, d.lr_switch = ptr(self.step_ctr), lr0, lr1, lr_switch
        for k in ('go_aa', 'body_aa', 'h1', 'h2', 'vo', 'vp_scratch', 'verts', 'v_posed', 'x0', 'canon', 'dx0', 'spartial', 'vpartial',
                  'losses', 'dverts', 'dvp', 'dA', 'dX', 'g_transl', 'g_rot6d', 'g_other', 'g_go', 'g_body'):
            setattr(d, k, ptr(self.ws[k]))
        d.pose = pose_ws
        d.loss_acc, d.step_cur = ptr(self.loss_acc), ptr(self.step_cur)
        self.snap = z(B * 65)                                            # pre-update parameters of the last Adam step
        self.nonfinite = torch.zeros(2, dtype=torch.int32, device=dev)   # first iteration with a NaN / Inf total loss
        d.snap, d.nonfinite = ptr(self.snap), ptr(self.nonfinite)
        d.per_frame, d.lr2, d.lr_switch2 = int(self.per_frame), float(lr2), int(lr_switch2)
        self._stepped = False
        if self.side_full:
            self.ws['verts_side'], self.ws['transl_side'] = z(B, data.V, 3), z(B, 3)
            d.verts_side, d.transl_side = ptr(self.ws['verts_side']), ptr(self.ws['transl_side'])
        for l in range(1, 11): d.act[l] = ptr(self.act[l])
        d.dact[0], d.dact[1] = ptr(self.dact[0]), ptr(self.dact[1])
        self.desc = d
        self.handle = self.lib.fit_create(C.byref(d))
        if not self.handle:
            raise _hip.LemoHipError('lemo_fit_create rejected the descriptor')
        # Stream ordering is owned by the engine: state writes (load_sequence / reset_optimizer) and
        # launches (forward / backward / step) may be issued on DIFFERENT non-blocking streams; each side records an event
        # and the other side's stream waits for it, so a clip's first Adam update can never overtake the zeroing of its moments
        self._init_order(self.device, self.lib)
        self._own = None
        self._after_write()          # the constructor's fills / uploads sit on its current stream

    def own_stream(self):
        """the fitter's persistent side stream (None on the emulator): graph capture needs a non-default stream, and a
        per-clip ``torch.cuda.Stream()`` would walk through torch's stream pool"""
        if not self._gpu:
            return None
        if self._own is None:
            self._own = torch.cuda.Stream(self.device)
        return self._own

    def step_async(self, n: int, use_graph: bool = True) -> None:
        """``n`` iterations on :meth:`own_stream`, ordered after what the current stream holds so far; result readers
        (``params72`` / ``losses`` / ...) order themselves after it through the engine's events"""
        s = self.own_stream()
        if s is None:
            return self.step(n, use_graph=False)
        s.wait_stream(self._cur())
        with torch.cuda.stream(s):
            self.step(n, use_graph=use_graph)

    def __del__(self):
        h, self.handle = getattr(self, 'handle', None), None
        if h:
            # The engine's buffers are torch tensors allocated on the default stream but written by graph replays on whatever
            # stream step() ran on.  When the last reference goes, the caching allocator may hand those blocks to the next
            # default-stream allocation at once -- while a replay is still in flight they would be written from two places.
            lib, rel = self.lib, getattr(_hip, 'release', None) if _hip is not None else None
            if rel is None:                  # interpreter shutdown: module globals are gone, the process is about to exit
                return
            rel(self.device, lib, lambda: lib.fit_destroy(h), getattr(self, '_run_ev', None))

    # -- sequence setup (opt_amass_temp.py:332-345) -------------------------------------------
    @torch.no_grad()
    def load_sequence(self, init_params: np.ndarray, markers_rec: np.ndarray, contact_lbl: np.ndarray):
        """init_params [B,72] (per-frame fit result), markers_rec [B,67,3], contact_lbl [B,4] in {0,1}: numpy arrays or
        tensors (device tensors are taken as they are: no host round trip)."""
        td = lambda a: (a.detach().to(self.device, torch.float32) if isinstance(a, torch.Tensor)
                        else torch.as_tensor(np.asarray(a, np.float32), device=self.device))
        p = td(init_params)
        assert p.shape == (self.B, 72)
        self._before_write()
        self.P['transl'].copy_(p[:, 0:3])
        self.P['rot6d'].copy_(convert_to_6D_all(p[:, 3:6]))
        self.P['shape'].copy_(p[:, 6:16])
        self.P['other'].copy_(p[:, 16:])
        self.target.copy_(td(markers_rec))
        self.contact.copy_(td(contact_lbl))
        self.reset_optimizer()

    @torch.no_grad()
    def reset_optimizer(self):
        """a fresh ``optim.Adam`` (opt_amass_temp.py:343, opt_amass_perframe.py:312): moments, step count and the
        non-finite-loss latch"""
        self._before_write()
        for t in self.adam_m + self.adam_v:
            t.zero_()
        self.step_ctr.zero_()
        self.nonfinite.zero_()
        self._stepped = False
        self._after_write()

    # -- optimiser state in / out (C ABI lemo_fit_load_state / lemo_fit_save_state) ---------------------------
    STATE_KEYS = ('transl', 'rot6d', 'other', 'm_transl', 'm_rot6d', 'm_other', 'v_transl', 'v_rot6d', 'v_other')

    def _state_struct(self, t: Dict[str, torch.Tensor], step: torch.Tensor) -> '_hip.FitState':
        st = _hip.FitState()
        st.transl, st.rot6d, st.other, st.step = ptr(t['transl']), ptr(t['rot6d']), ptr(t['other']), ptr(step)
        for i, k in enumerate(('transl', 'rot6d', 'other')):
            st.adam_m[i], st.adam_v[i] = ptr(t['m_' + k]), ptr(t['v_' + k])
        return st

    @torch.no_grad()
    def save_state(self) -> Dict[str, torch.Tensor]:
        """what ``optim.Adam`` + the three parameter tensors of ``opt_amass_temp.py:332-345`` hold between two iterations:
        parameters, ``exp_avg`` (m_*), ``exp_avg_sq`` (v_*) and ``step`` = completed Adam steps (device tensors, copies)."""
        B = self.B
        t = {k: torch.empty(B, w, dtype=torch.float32, device=self.device)
             for k, w in zip(self.STATE_KEYS, (3, 6, 56) * 3)}
        step = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._before_run()
        self.lib.check(self.lib.fit_save_state(self.handle, C.byref(self._state_struct(t, step)), self._s()), 'fit_save_state')
        self._after_run()
        t['step'] = step
        return t

    @torch.no_grad()
    def load_state(self, state: Dict) -> None:
        """continue from ``state`` (the dict :meth:`save_state` returns; numpy arrays or tensors, ``step`` an int or a
        tensor): the next :meth:`step` is iteration ``step`` of the loop -- learning-rate level, bias corrections and all.
        Sequence data (``target``, ``contact``, ``shape``) are not part of the optimiser state: :meth:`load_sequence` first."""
        td = lambda a, w: (a.detach() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, np.float32))
                           ).to(self.device, torch.float32).reshape(self.B, w).contiguous()
        t = {k: td(state[k], w) for k, w in zip(self.STATE_KEYS, (3, 6, 56) * 3)}
        sv = state['step']
        step = (sv.detach().to(self.device, torch.int32).reshape(1) if isinstance(sv, torch.Tensor)
                else torch.full((1,), int(sv), dtype=torch.int32, device=self.device))
        self._before_run()
        self.lib.check(self.lib.fit_load_state(self.handle, C.byref(self._state_struct(t, step)), self._s()), 'fit_load_state')
        self._stepped = False
        self._after_run()
        self._keep = (t, step)              # the copies read caller-side buffers asynchronously: keep them until the next call

    # -- execution ---------------------------------------------------------------------------
    def _s(self):
        if self.lib.is_emu:
            return None
        return torch.cuda.current_stream(self.device).cuda_stream

    def forward(self) -> None:
        self._before_run()
        self.lib.check(self.lib.fit_forward(self.handle, self._s()), 'fit_forward')
        self._stepped = False
        self._after_run()

    def backward(self) -> None:
        self._before_run()
        self.lib.check(self.lib.fit_backward(self.handle, self._s()), 'fit_backward')
        self._after_run()

    def step(self, n: int = 1, use_graph: bool = True) -> None:
        """``n`` Adam iterations (forward, backward, update) -- asynchronous on the current stream.
        With ``use_graph`` the iteration is captured once and replayed; capture needs a non-default
        stream, so call inside ``with torch.cuda.stream(s):``.  Ordered after the last ``load_sequence`` / ``reset_optimizer``
        even when those were issued on another stream (the engine's own events)."""
        self._before_run()
        self.lib.check(self.lib.fit_step(self.handle, int(n), int(bool(use_graph) and not self.lib.is_emu), self._s()),
                       'fit_step')
        self._stepped = self._stepped or n > 0
        self._after_run()

    STAGES = ('vposer_pose_fwd', 'vertex_fwd', 'marker_encoder_fwd', 'losses', 'encoder_bwd', 'vertex_bwd', 'pose_vposer_bwd_tail')

    def stage_census(self, reps: int = 20) -> Dict[str, float]:
        """microseconds per iteration spent in each stage (``lemo_fit_census``: every stage replayed ``reps`` times back to back
        from its own graph) + ``'forward_backward'`` = the whole chain measured the same way.  Diagnostics: synchronises."""
        ms = (C.c_float * 8)()
        self._before_run()
        self.lib.check(self.lib.fit_census(self.handle, int(reps), ms, self._s()), 'fit_census')
        self._stepped = False
        self._after_run()
        out = {k: float(ms[i]) * 1e3 for i, k in enumerate(self.STAGES)}
        out['forward_backward'] = float(ms[7]) * 1e3
        return out

    def nonfinite_step(self) -> int:
        """1-based index of the first iteration (since the last ``load_sequence`` / ``reset_optimizer``) whose total loss
        was NaN or Inf, 0 if none (synchronises).  From the following iteration on the engine skipped every update --
        the ``break`` of ``FittingMonitor.run_fitting`` (fitting_temp_slide.py:198-204) inside a replayed graph."""
        self._before_read()
        return int(self.nonfinite[0].item())

    def prepare(self, n: int) -> None:
        """capture (without running) the hipGraphs an ``n``-iteration :meth:`step` on the current stream replays."""
        if not self.lib.is_emu:
            self.lib.check(self.lib.fit_prepare(self.handle, int(n), self._s()), 'fit_prepare')

    # -- results -------------------------------------------------------------------------------
    def losses(self) -> Dict[str, float]:
        self._before_read()
        v = self.ws['losses'].detach().cpu().numpy()
        return {k: float(v[i]) for i, k in enumerate(LOSS_NAMES)}

    def grads(self) -> Dict[str, torch.Tensor]:
        """raw gradients after ``backward()`` (the L2-prior terms on ``other`` are added inside the
        Adam kernel; ``grads_with_priors`` adds them here for comparison with autograd)."""
        self._before_read()
        return dict(transl=self.ws['g_transl'], rot6d=self.ws['g_rot6d'], other=self.ws['g_other'])

    def grads_with_priors(self) -> Dict[str, torch.Tensor]:
        g = {k: v.clone() for k, v in self.grads().items()}
        o, w = self.P['other'], self.weights
        Bn = 1 if self.per_frame else self.B                 # per_frame: every row is a fit of its own (FitTail.Bn)
        # the tail kernel's own expression, operation for operation: (w * 2) * p / (Bn * d) with a TRUE division -- torch divides
        # a device tensor by a Python scalar as a multiplication with the reciprocal (1 / 24 is not a power of two: last-bit
        # differences from the kernel in the hand columns, seen by the bit-exact Adam check of the teacher-forced tests)
        d32 = torch.full((), float(Bn) * 32.0, dtype=torch.float32, device=o.device)
        d24 = torch.full((), float(Bn) * 24.0, dtype=torch.float32, device=o.device)
        wv = torch.full((), float(np.float32(w['vposer']) * np.float32(2.0)), dtype=torch.float32, device=o.device)
        wh = torch.full((), float(np.float32(w['hand']) * np.float32(2.0)), dtype=torch.float32, device=o.device)
        g['other'][:, :32] += (wv * o[:, :32]) / d32
        g['other'][:, 32:] += (wh * o[:, 32:]) / d24
        return g

    def params75(self) -> torch.Tensor:
        return torch.cat([self.P['transl'], self.P['rot6d'], self.P['shape'], self.P['other']], dim=-1)

    def params72(self) -> torch.Tensor:
        """``body_params_opt_t_72`` of the LAST forward: [transl, global_orient aa, betas, z, hands] -- what the
        reference saves (opt_amass_temp.py:457-458: the parameters the last iteration's forward saw, i.e. after
        n - 1 updates).  After ``step(n)`` every column comes from that iteration: ``go_aa`` from its forward and the
        pre-update snapshot the Adam kernel took; after a bare ``forward()`` the live parameters are those values."""
        self._before_read()
        if self._stepped:
            B = self.B
            tr, ot = self.snap[:3 * B].view(B, 3), self.snap[9 * B:].view(B, 56)
        else:
            tr, ot = self.P['transl'], self.P['other']
        return torch.cat([tr, self.ws['go_aa'], self.P['shape'], ot], dim=-1)

    def vertices(self) -> torch.Tensor:
        """[B, V, 3] (``full_vertices``) or [B, n, 3] (the set U) of the last forward"""
        self._before_read()
        return self.ws['verts_side'] if self.side_full else self.ws['verts']

    def marker_vertices(self) -> torch.Tensor:
        """the 67 marker vertices of the last forward, [B,67,3] (whatever the vertex layout of the loss path is)"""
        self._before_read()
        return self.ws['verts'][:, self._idx['row67'].long()]

    def posed_joints(self) -> torch.Tensor:
        """the 55 posed skeleton joints + transl of the last forward, [B,55,3]."""
        self._before_read()
        return self._pose_t['Jtr'] + self.P['transl'][:, None, :]


class PerFrameFitter:
    """Stage 1 of LEMO on AMASS (``opt_amass_perframe.py:293-355``, BASELINE configs[0]): every frame of a clip is
    fitted on its own -- B = 1, 100 Adam steps on ``weight_loss_rec_markers * L1(markers) + vposer / shape / hand L2
    priors`` -- and frame t starts from the result of frame t - 1 (the parameter tensors live on across the loop,
    :297-310) with a fresh optimiser (lr 0.1 for the first frame, 0.01 after; 0.01 from step 61, 0.003 from step 81,
    :312-321).  Same engine as :class:`AmassTemporalFitter` (``lemo_fit_desc.per_frame``)."""

    INIT_TRANSL = (0.0, 0.4, 1.0)         # :300-301
    INIT_ORIENT = (0.0, 1.6, 3.14)        # :303-304

    def __init__(self, body, vposer_weights, enc_state, ids, Xmean, Xstd, device, weights: Optional[dict] = None,
                 lib: Optional[_hip.HipLib] = None, full_vertices: bool = False):
        # full_vertices = False: stage 1 reads 67 marker vertices of ONE frame and returns parameters only;
        # regressing all 10475 vertices per iteration is a 250-workgroup launch (40 of the iteration's 99 us at B = 1, and
        # the whole device) that nothing consumes.  Losses, gradients and results are identical (tested); True restores it.
        w = dict(LOSS_WEIGHTS if weights is None else weights, contact_vel=0.0, smooth=0.0)
        mk = lambda lr0: AmassTemporalFitter(body, vposer_weights, enc_state, ids, Xmean, Xstd, 1, device, weights=w,
                                             full_vertices=full_vertices,
                                             lr0=lr0, lr1=0.01, lr_switch=60, lr2=0.003, lr_switch2=80, per_frame=True, lib=lib)
        self.first, self.rest = mk(0.1), mk(0.01)
        self.device = self.first.device

    def _stream(self, use_graph: bool):
        """the persistent side stream the frames of this fitter run on (graph capture needs a non-default stream and graphs
        are per stream); None on the emulator / for eager launches"""
        if not (bool(use_graph) and not self.first.lib.is_emu):
            return None
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(self.device)
        return self._side

    def begin_clip(self, markers_rec: np.ndarray, betas: np.ndarray) -> dict:
        """state of one clip's frame loop (see :meth:`fit_frame`)"""
        mr = torch.as_tensor(np.asarray(markers_rec, np.float32), device=self.device)
        init = np.zeros((1, 72), np.float32)
        init[0, 0:3], init[0, 3:6], init[0, 6:16] = self.INIT_TRANSL, self.INIT_ORIENT, np.asarray(betas, np.float32)
        return dict(mr=mr, T=mr.shape[0], out=torch.empty(mr.shape[0], 72, device=self.device), init=init,
                    zero_lbl=np.zeros((1, 4), np.float32))

    @torch.no_grad()
    def fit_frame(self, st: dict, t: int, steps: int = 100, use_graph: bool = True) -> None:
        """frame t of the clip (frames must be fitted in order: t starts from the result of t - 1); asynchronous on the
        fitter's side stream when graphs are used"""
        import contextlib
        side = self._stream(use_graph)
        graph = side is not None
        if side is not None and t == 0:
            side.wait_stream(torch.cuda.current_stream(self.device))
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            fit = self.first if t == 0 else self.rest
            if t == 0:
                fit.load_sequence(st['init'], st['mr'][0:1], st['zero_lbl'])
            else:
                if t == 1:                                      # hand the running parameters to the lr-0.01 engine
                    for k in ('transl', 'rot6d', 'other', 'shape'):
                        fit.P[k].copy_(self.first.P[k])
                    fit.contact.zero_()
                fit.target.copy_(st['mr'][t:t + 1])
                fit.reset_optimizer()
            fit.step(steps, use_graph=graph)
            st['out'][t] = fit.params72()[0]

    def end_clip(self, st: dict) -> torch.Tensor:
        side = getattr(self, '_side', None)
        if side is not None:
            torch.cuda.current_stream(self.device).wait_stream(side)
        return st['out']

    @torch.no_grad()
    def fit_clip(self, markers_rec: np.ndarray, betas: np.ndarray, steps: int = 100, use_graph: bool = True) -> torch.Tensor:
        """markers_rec [T,67,3], betas [10] (fixed, ``beta_gt``) -> ``body_params_opt_cur_clip`` [T,72]"""
        st = self.begin_clip(markers_rec, betas)
        for t in range(st['T']):
            self.fit_frame(st, t, steps, use_graph)
        return self.end_clip(st)


class BatchedPerFrameFitter:
    """Stage 1 (``opt_amass_perframe.py:293-355``) for N clips AT ONCE through ONE engine: row i of the engine's batch is
    the current frame of clip i.  With ``lemo_fit_desc.per_frame`` every row is a fit of its own -- the marker mean and
    the prior means run over that row only, Adam state is per row, nothing in the per-frame objective couples rows -- so
    N frame fits cost the launches of one (an iteration is ~11 launches of N workgroups instead of one workgroup; the
    per-frame kernels have one workgroup per row).  One clip alone is bound by its 100 x 119 sequential iterations
    (~71 us each, ~140 frame fits/s); N clips in lockstep reach N times that until the device fills (VERDICT r02 #4ii).
    Each clip's result is bit-identical to its solo fit with :class:`PerFrameFitter` (tested: every kernel's arithmetic
    for a row is independent of the batch size).  Clips may differ in length: rows of finished clips keep fitting their
    last frame and are ignored.  (The non-finite-loss latch is per engine: a NaN in one clip freezes the batch.)"""

    def __init__(self, body, vposer_weights, enc_state, ids, Xmean, Xstd, device, batch: int, weights: Optional[dict] = None,
                 lib: Optional[_hip.HipLib] = None, full_vertices: bool = False):
        w = dict(LOSS_WEIGHTS if weights is None else weights, contact_vel=0.0, smooth=0.0)
        self.N = int(batch)
        mk = lambda lr0: AmassTemporalFitter(body, vposer_weights, enc_state, ids, Xmean, Xstd, self.N, device, weights=w,
                                             full_vertices=full_vertices, lr0=lr0, lr1=0.01, lr_switch=60, lr2=0.003, lr_switch2=80,
                                             per_frame=True, lib=lib)
        self.first, self.rest = mk(0.1), mk(0.01)
        self.device = self.first.device

    @torch.no_grad()
    def fit_clips(self, markers_list, betas_list, steps: int = 100, use_graph: bool = True):
        """markers_list[i] [T_i,67,3], betas_list[i] [10] -> list of ``body_params_opt_cur_clip`` [T_i,72] (device tensors)"""
        n = len(markers_list)
        assert 1 <= n <= self.N and len(betas_list) == n
        dev, N = self.device, self.N
        Ts = [int(np.asarray(m).shape[0]) for m in markers_list]
        Tmax = max(Ts)
        mr = torch.zeros(N, Tmax, np.asarray(markers_list[0]).shape[1], 3, device=dev)
        for i, m in enumerate(markers_list):
            t = torch.as_tensor(np.asarray(m, np.float32), device=dev)
            mr[i, :Ts[i]] = t
            mr[i, Ts[i]:] = t[-1]
        mr[n:] = mr[0]                                       # unused rows repeat clip 0 (ignored)
        init = np.zeros((N, 72), np.float32)
        init[:, 0:3], init[:, 3:6] = PerFrameFitter.INIT_TRANSL, PerFrameFitter.INIT_ORIENT
        for i in range(N):
            init[i, 6:16] = np.asarray(betas_list[min(i, n - 1)], np.float32)
        out = torch.empty(Tmax, N, 72, device=dev)
        zero_lbl = np.zeros((N, 4), np.float32)
        graph = bool(use_graph) and not self.first.lib.is_emu
        for t in range(Tmax):
            fit = self.first if t == 0 else self.rest
            if t == 0:
                fit.load_sequence(init, mr[:, 0], zero_lbl)
            else:
                fit._before_write()
                if t == 1:
                    for k in ('transl', 'rot6d', 'other', 'shape'):
                        fit.P[k].copy_(self.first.P[k])
                    fit.contact.zero_()
                fit.target.copy_(mr[:, t])
                fit.reset_optimizer()
            if graph:
                fit.step_async(steps, use_graph=True)
            else:
                fit.step(steps, use_graph=False)
            out[t] = fit.params72()
        return [out[:Ts[i], i] for i in range(n)]


def fit_clips_per_frame(fitters, markers_list, betas_list, steps: int = 100, use_graph: bool = True):
    """Stage 1 for several clips SIDE BY SIDE: ``fitters[i]`` (a :class:`PerFrameFitter` each, its own engines and stream) fits
    clip i; the frame loops advance in lockstep so that the device always holds one frame fit of every clip.  A B = 1 fit uses a
    sliver of the device (eleven launches of one workgroup or so per iteration, each dominated by its ~4.7 us launch boundary),
    and frame t of a clip needs frame t - 1 of the SAME clip only -- clips are the parallelism stage 1 has.  Results are
    bit-identical to ``fitters[i].fit_clip`` run one after the other (tested)."""
    assert len(fitters) >= len(markers_list) == len(betas_list)
    sts = [f.begin_clip(m, b) for f, m, b in zip(fitters, markers_list, betas_list)]
    for t in range(max(st['T'] for st in sts)):
        for f, st in zip(fitters, sts):
            if t < st['T']:
                f.fit_frame(st, t, steps, use_graph)
    return [f.end_clip(st) for f, st in zip(fitters, sts)]
