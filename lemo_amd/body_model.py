"""``smplx``-compatible SMPL-X body model on the MI355X HIP kernels.

Mirrors the call surface LEMO uses:
``smplx.create(model_path, model_type='smplx', gender=..., num_pca_comps=12, batch_size=B, ...)``
(opt_amass_temp.py:73-87, temp_prox/main_slide.py:160-179) returning an ``nn.Module`` whose
``forward(**params)`` yields an object with ``.vertices``, ``.joints``, ``.full_pose`` ...
(utils/utils.py:152,167; temp_prox/fitting_temp_slide.py:248-257), plus ``reset_params``,
``faces_tensor``, ``joint_mapper`` and ``get_num_verts``.

The arithmetic is smplx==0.1.26 ``SMPLX.forward`` -> ``lbs`` (vendored statement:
human_body_prior/body_model/lbs.py:34-119).  Compute runs in ``liblemo_hip.so``:
pose stage (hand PCA, pose mean, Rodrigues, joint regression, kinematic chain) ->
vertex stage (one fp32-MFMA GEMM for shape+pose blend shapes fused with ELL skinning).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _hip
from ._hip import ptr

# smplx.vertex_ids['smplx'] in VertexJointSelector order
EXTRA_JOINT_VERTEX_IDS = [9120, 9929, 9448, 616, 6, 5770, 5780, 8846, 8463, 8474, 8635,
                          5361, 4933, 5058, 5169, 5286, 8079, 7669, 7794, 7905, 8022]
K_PAD = 512          # blend-shape GEMM depth: 20 shape/expression + 486 pose features, padded
DENSE_CHUNK = 512    # LBS_DENSE_CHUNK of lbs_kernels.hip
GEMM_SLABS = int(os.environ.get('LEMO_GEMM_SLABS', '64'))      # K slabs of the all-vertex feature-gradient GEMM (gemm_nt16_splitk): x 4 M-blocks of 128 rows = 256 workgroups (round 3; 96 x 8 before)


def _roundup(x, m):
    return (x + m - 1) // m * m


def load_model_dict(model_path, gender: str = 'neutral', ext: str = 'npz') -> Dict[str, np.ndarray]:
    """Accept a dict (e.g. :func:`lemo_amd.synthetic.make_synthetic_smplx`), a file, or the smplx
    directory convention ``<model_path>/smplx/SMPLX_<GENDER>.<ext>``."""
    if isinstance(model_path, dict):
        return model_path
    p = model_path
    if os.path.isdir(p):
        for cand in (os.path.join(p, 'smplx', f'SMPLX_{gender.upper()}.{ext}'),
                     os.path.join(p, f'SMPLX_{gender.upper()}.{ext}')):
            if os.path.exists(cand):
                p = cand
                break
    if not os.path.isfile(p):
        raise FileNotFoundError(f'SMPL-X model file not found under {model_path!r}')
    if p.endswith('.npz'):
        d = np.load(p, allow_pickle=True)
        return {k: d[k] for k in d.files}
    import pickle
    with open(p, 'rb') as f:
        return dict(pickle.load(f, encoding='latin1'))


class BodyModelData:
    """Host-side preprocessing of one SMPL-X-shaped model into the layouts the kernels stream."""

    def __init__(self, model: Dict[str, np.ndarray], num_pca_comps: int = 12, use_pca: bool = True,
                 flat_hand_mean: bool = False, num_betas: int = 10, extra_joint_ids: Optional[Sequence[int]] = None):
        f32 = np.float32
        self.V = V = int(model['v_template'].shape[0])
        self.v_template = np.ascontiguousarray(model['v_template'], f32)
        sd = np.asarray(model['shapedirs'])
        expr = sd[:, :, 10:20] if sd.shape[-1] < 310 else sd[:, :, 300:310]
        shapedirs = np.concatenate([sd[:, :, :num_betas], expr], axis=-1).astype(f32)        # [V,3,20]
        self.nshape = int(shapedirs.shape[-1])
        posedirs = np.asarray(model['posedirs'], f32)                                       # [V,3,P]
        P = posedirs.shape[-1]
        self.nj = nj = P // 9 + 1
        assert self.nshape + P <= K_PAD
        # D[k][3v+c]: k < nshape shape/expression directions, then pose directions
        D = np.zeros((K_PAD, 3 * V), f32)
        D[:self.nshape] = shapedirs.reshape(3 * V, self.nshape).T
        D[self.nshape:self.nshape + P] = posedirs.reshape(3 * V, P).T
        self.D = D
        self.NC = _roundup(3 * V, 8)
        Dg = np.zeros((K_PAD // 8, self.NC, 8), f32)
        Dg[:, :3 * V, :] = D.reshape(K_PAD // 8, 8, 3 * V).transpose(0, 2, 1)
        self.Dg = Dg
        Jr = np.asarray(model['J_regressor'].todense() if hasattr(model['J_regressor'], 'todense')
                        else model['J_regressor'], np.float64)
        self.J_template = (Jr @ self.v_template.astype(np.float64)).astype(f32)              # [nj,3]
        self.J_dirs = np.einsum('jv,vck->jck', Jr, shapedirs.astype(np.float64)).astype(f32)  # [nj,3,nshape]
        parents = np.asarray(model['kintree_table'][0], np.int64).copy()
        parents[0] = -1
        self.parents = parents.astype(np.int32)
        depth = np.zeros(nj, np.int32)
        for j in range(1, nj):
            assert parents[j] < j, 'kinematic tree must be topologically ordered'
            depth[j] = depth[parents[j]] + 1
        order = np.argsort(depth, kind='stable').astype(np.int32)
        self.nlev = int(depth.max()) + 1
        self.level_joints = order
        self.level_start = np.searchsorted(depth[order], np.arange(self.nlev + 1)).astype(np.int32)
        cs, cl = [0], []
        for j in range(nj):
            cl += [c for c in range(1, nj) if parents[c] == j]
            cs.append(len(cl))
        self.child_start, self.child_list = np.asarray(cs, np.int32), np.asarray(cl + [0], np.int32)
        # skinning weights -> ELL
        W = np.asarray(model['weights'], f32)
        nnz = (W != 0).sum(1)
        self.KW = KW = max(int(nnz.max()), 1)
        order_w = np.argsort(-(W != 0).astype(np.int8), axis=1, kind='stable')[:, :KW]
        self.w_idx = np.ascontiguousarray(order_w, np.int32)
        self.w_val = np.ascontiguousarray(np.take_along_axis(W, order_w, 1), f32)
        self.w_idx[self.w_val == 0] = 0
        # hands / pose mean
        self.use_pca = bool(use_pca)
        self.ncomp = int(num_pca_comps) if use_pca else 0
        self.lh_comp = np.ascontiguousarray(model['hands_componentsl'][:max(self.ncomp, 1)], f32)
        self.rh_comp = np.ascontiguousarray(model['hands_componentsr'][:max(self.ncomp, 1)], f32)
        lhm = np.zeros(45, f32) if flat_hand_mean else np.asarray(model['hands_meanl'], f32)
        rhm = np.zeros(45, f32) if flat_hand_mean else np.asarray(model['hands_meanr'], f32)
        self.pose_mean = np.concatenate([np.zeros(3 * nj - 90, f32), lhm, rhm]).astype(f32)
        # joints from vertices
        self.faces = np.asarray(model['f'], np.int64)
        lmk_idx = np.asarray(model['lmk_faces_idx'], np.int64)
        self.lmk_rows = np.ascontiguousarray(self.faces[lmk_idx], np.int32)                  # [51,3] vertex ids
        self.lmk_bary = np.ascontiguousarray(model['lmk_bary_coords'], f32)
        ids = EXTRA_JOINT_VERTEX_IDS if extra_joint_ids is None else list(extra_joint_ids)
        # (the fit engines never gather these rows; the module route does: SMPLX.__init__ refuses ids beyond the model)
        self.extra_ids_in_range = not len(ids) or (min(ids) >= 0 and max(ids) < V)
        self.extra_ids = np.asarray(ids, np.int32)
        self.n_joints_out = nj + len(ids) + self.lmk_rows.shape[0]

    # -- compact backward structure for a vertex set ------------------------------------------
    def vertex_set(self, ids: np.ndarray, vp_row: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
        ids = np.asarray(ids, np.int64)
        n = ids.shape[0]
        NCs = _roundup(3 * n, 16)
        Dk = np.zeros((K_PAD, NCs), np.float32)
        cols = (ids[:, None] * 3 + np.arange(3)[None]).reshape(-1)
        Dk[:, :3 * n] = self.D[:, cols]
        ju, jw, js = [], [], [0]
        wi, wv = self.w_idx[ids], self.w_val[ids]
        for j in range(self.nj):
            u, k = np.nonzero((wi == j) & (wv != 0))
            ju.append(u.astype(np.int32)); jw.append(wv[u, k].astype(np.float32)); js.append(js[-1] + u.shape[0])
        extra = {}
        if n <= 4096:          # small sets also get the feature-contiguous copy used by the small-set FORWARD
            extra['DkT'] = np.ascontiguousarray(Dk.T)
        if n > 1024:           # sets beyond the staged per-frame kernel (LBS_BWD_STAGE): per-chunk offsets into the
            # (position-sorted) joint lists -> deterministic dense backward
            nchunk = (n + DENSE_CHUNK - 1) // DENSE_CHUNK
            tab = np.zeros((nchunk, self.nj + 1), np.int32)
            cu, cw, off = [], [], 0
            for c in range(nchunk):                  # chunk-major copy of the joint lists: one contiguous run per chunk
                for j in range(self.nj):
                    lo, hi = np.searchsorted(ju[j], [c * DENSE_CHUNK, (c + 1) * DENSE_CHUNK])
                    tab[c, j] = off
                    cu.append(ju[j][lo:hi]); cw.append(jw[j][lo:hi])
                    off += hi - lo
                tab[c, self.nj] = off
            extra['jcsr_chunk'] = tab
            extra['jc_u'] = np.concatenate(cu + [np.zeros(1, np.int32)]).astype(np.int32)
            extra['jc_w'] = np.concatenate(cw + [np.zeros(1, np.float32)]).astype(np.float32)
        return dict(n=n, NCs=NCs, ids=ids.astype(np.int32), **extra,
                    vp_row=(ids if vp_row is None else np.asarray(vp_row)).astype(np.int32), Dk=Dk,
                    jcsr_start=np.asarray(js, np.int32),
                    jcsr_u=np.concatenate(ju + [np.zeros(1, np.int32)]), jcsr_w=np.concatenate(jw + [np.zeros(1, np.float32)]))


class DeviceBody:
    """Device copies of :class:`BodyModelData` + the ctypes constant blocks of the C ABI."""

    def __init__(self, data: BodyModelData, device, blend_f16: Optional[bool] = None):
        self.data, self.device = data, torch.device(device)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.t = {k: t(getattr(data, k)) for k in (
            'Dg', 'v_template', 'J_template', 'J_dirs', 'parents', 'level_start', 'level_joints', 'child_start',
            'child_list', 'w_idx', 'w_val', 'lh_comp', 'rh_comp', 'pose_mean', 'lmk_rows', 'lmk_bary', 'extra_ids')}
        d, T = data, self.t
        self.body = _hip.BodyConst(d.nj, d.nshape, d.ncomp, d.nlev, ptr(T['parents']), ptr(T['level_start']),
                                   ptr(T['level_joints']), ptr(T['child_start']), ptr(T['child_list']),
                                   ptr(T['J_template']), ptr(T['J_dirs']), ptr(T['pose_mean']),
                                   ptr(T['lh_comp']), ptr(T['rh_comp']))
        # blend directions pre-split into two fp16 pieces (lemo_skin_const.DgH): the blend GEMM of lbs_verts_fwd then runs 3 fp16 MFMA
        # products per k-chunk on operands it only has to move (LEMO_BLEND_F16=0: the 3-piece bf16 split of the fp32 copy, converted
        # in the kernel -- round 2's form, kept for A/B)
        self.blend_f16 = BLEND_F16 if blend_f16 is None else bool(blend_f16)
        dgh, inv = (None, 0.0)
        if self.blend_f16:
            h, inv = split_f16_pairs(d.Dg)
            T['DgH'] = dgh = t(h)
        self.skin = _hip.SkinConst(d.V, d.NC, d.KW, 0, ptr(T['Dg']), ptr(T['v_template']), ptr(T['w_idx']), ptr(T['w_val']),
                                   ptr(dgh) if dgh is not None else None, inv)
        self._sets = {}

    def vertex_set(self, key, ids: np.ndarray, vp_row=None, frames: int = 0):
        """device copy of ``BodyModelData.vertex_set`` -> (ctypes struct, tensors).  The index tables and ``Dk`` / ``DkG`` are
        immutable and cached per model; with ``frames`` > 0 (large sets: the deterministic dense backward) the returned
        struct is a PRIVATE copy carrying freshly allocated scratch for that many frames -- the partial sums of
        ``lbs_bwd_chunk`` and the K-slab partials of the feature-gradient GEMM.  Scratch is never shared: an engine copies
        the struct by value into its descriptor and captured graphs, so a later, larger request must not free it, and two
        engines on two streams must not write the same partials."""
        if key not in self._sets:
            s = self.data.vertex_set(ids, vp_row)
            tt = {k: torch.from_numpy(v).to(self.device) for k, v in s.items() if isinstance(v, np.ndarray)}
            if 'jcsr_chunk' in tt:                  # Dk k-chunk major for the split-K GEMM: [NCs/16][512][16] (immutable)
                tt['DkG'] = tt['Dk'].view(K_PAD, s['NCs'] // 16, 16).permute(1, 0, 2).contiguous()
            st = _hip.VertexSetBwd(s['n'], s['NCs'], ptr(tt['ids']), ptr(tt['vp_row']), ptr(tt['Dk']),
                                   ptr(tt['DkT']) if 'DkT' in tt else None,
                                   ptr(tt['jcsr_start']), ptr(tt['jcsr_u']), ptr(tt['jcsr_w']),
                                   ptr(tt['jcsr_chunk']) if 'jcsr_chunk' in tt else None,
                                   ptr(tt['jc_u']) if 'jc_u' in tt else None, ptr(tt['jc_w']) if 'jc_w' in tt else None, None, 0)
            st.DkG = ptr(tt['DkG']) if 'DkG' in tt else None
            self._sets[key] = (st, tt)
        st, tt = self._sets[key]
        if 'jcsr_chunk' not in tt or frames <= 0:
            return st, tt
        mine = _hip.VertexSetBwd()
        C.memmove(C.byref(mine), C.byref(st), C.sizeof(st))
        own = dict(tt)                              # same immutable tensors + this caller's scratch (keeps both alive)
        nchunk = tt['jcsr_chunk'].shape[0]
        own['part'] = torch.zeros(frames, nchunk, self.data.nj * 12 + 4, dtype=torch.float32, device=self.device)
        # K-slab partials of the feature-gradient GEMM: rows n < frames are fully written by every launch and the reducer reads only
        # those -- no zero-fill needed (17 MB memset per autograd backward otherwise); `part` above IS read sparsely
        own['gemm_part'] = torch.empty(GEMM_SLABS * 128 * K_PAD, dtype=torch.float32, device=self.device)   # 32 x 128 x 512 floats
        mine.part, mine.part_frames = ptr(own['part']), frames
        mine.gemm_part, mine.gemm_slabs = ptr(own['gemm_part']), GEMM_SLABS
        return mine, own


BLEND_F16 = os.environ.get('LEMO_BLEND_F16', '1') != '0'


def split_f16_pairs(Dg: np.ndarray):
    """Dg [K/8][NC][8] fp32 -> ([K/8][NC][2 halves][hi 4 | lo 4] as uint16, 2^-k): x * 2^k = hi + lo with hi = f16(x 2^k),
    lo = f16(x 2^k - hi) (round to nearest even), k the largest power that keeps max|x| 2^k <= 2^15 (fp16 overflows at 65504;
    small entries go denormal in `lo` only below 2^-24 of the largest, i.e. below fp32's own resolution of the sums)."""
    m = float(np.abs(Dg).max())
    k = int(np.floor(np.log2(32768.0 / m))) if m > 0 else 0
    k = max(-14, min(k, 30))
    xs = Dg.astype(np.float32) * np.float32(2.0 ** k)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    G, NC, _ = Dg.shape
    out = np.empty((G, NC, 2, 2, 4), np.float16)                    # [group][column][half][piece][4]
    out[:, :, :, 0, :] = hi.reshape(G, NC, 2, 4)
    out[:, :, :, 1, :] = lo.reshape(G, NC, 2, 4)
    return out.view(np.uint16).reshape(G, NC, 16).view(np.int16), float(2.0 ** -k)


def alloc_pose_ws(B: int, nj: int, device, f16: bool):
    """Workspace of the pose stage (saved for backward).  Xg pad rows/cols must stay zero.  ``f16``: the form of the pre-split
    features XgS -- two fp16 pieces (goes with ``DeviceBody.blend_f16`` / ``lemo_skin_const.DgH``) or three bf16 pieces."""
    Bp = _roundup(B, 32)
    z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
    tt = dict(full_pose=z(B, nj * 3), R=z(B, nj, 9), J=z(B, nj, 3), T=z(B, nj, 12), A=z(B, nj, 12),
              Jtr=z(B, nj, 3), Xg=z(K_PAD // 8, Bp, 8),
              # Xg again as two fp16 (or three exact bf16) pieces in MFMA-fragment order (lemo_pose_ws.XgS): the blend GEMM's B operand
              XgS=torch.zeros(K_PAD // 16, 2 if f16 else 3, Bp, 2, 8, dtype=torch.int16, device=device))
    ws = _hip.PoseWs(ptr(tt['full_pose']), ptr(tt['R']), ptr(tt['J']), ptr(tt['T']), ptr(tt['A']), ptr(tt['Jtr']),
                     ptr(tt['Xg']), Bp, ptr(tt['XgS']), 1 if f16 else 0)
    return ws, tt, Bp


class _SmplxFn(torch.autograd.Function):
    """(betas, expr, go, body, jaw, leye, reye, lh, rh, transl) -> (verts, joints, full_pose)."""

    @staticmethod
    def forward(ctx, dev: DeviceBody, lib: _hip.HipLib, betas, expr, go, body, jaw, leye, reye, lh, rh, transl):
        d = dev.data
        B = go.shape[0]
        args = [a.contiguous().float() for a in (betas, expr, go, body, jaw, leye, reye, lh, rh)]
        betas, expr, go, body, jaw, leye, reye, lh, rh = args
        tr = None if transl is None else transl.contiguous().float()
        device = go.device
        _hip.check_device(lib, go)
        s = lib.stream(device)
        ws, tt, Bp = alloc_pose_ws(B, d.nj, device, dev.blend_f16)
        pin = _hip.PoseIn(ptr(go), ptr(body), ptr(jaw), ptr(leye), ptr(reye), ptr(lh), ptr(rh), lh.shape[1],
                          ptr(betas), betas.shape[1], ptr(expr))
        lib.check(lib.smplx_pose_fwd(C.byref(dev.body), C.byref(pin), C.byref(ws), B, s), 'smplx_pose_fwd')
        verts = torch.empty(B, d.V, 3, dtype=torch.float32, device=device)
        v_posed = torch.empty(B, d.V, 3, dtype=torch.float32, device=device)
        lib.check(lib.lbs_verts_fwd_xs(C.byref(dev.skin), ptr(tt['Xg']), ptr(tt['XgS']), Bp, ptr(tt['A']), d.nj, ptr(tr), None, d.V, B,
                                       ptr(verts), ptr(v_posed), s), 'lbs_verts_fwd')
        joints = torch.empty(B, d.n_joints_out, 3, dtype=torch.float32, device=device)
        lib.check(lib.joints_assemble(ptr(tt['Jtr']), d.nj, ptr(verts), d.V, ptr(dev.t['extra_ids']), len(d.extra_ids),
                                      ptr(dev.t['lmk_rows']), ptr(dev.t['lmk_bary']), d.lmk_rows.shape[0], ptr(tr), B,
                                      ptr(joints), s), 'joints_assemble')
        ctx.dev, ctx.lib, ctx.tt, ctx.ws, ctx.Bp, ctx.B = dev, lib, tt, ws, Bp, B
        ctx.v_posed, ctx.has_transl, ctx.hand_dim = v_posed, tr is not None, lh.shape[1]
        return verts, joints, tt['full_pose'].clone()

    @staticmethod
    def backward(ctx, dverts, djoints, dfp):
        dev, lib, tt, B, Bp = ctx.dev, ctx.lib, ctx.tt, ctx.B, ctx.Bp
        d = dev.data
        device = ctx.v_posed.device
        s = lib.stream(device)
        z = lambda *sh: torch.zeros(*sh, dtype=torch.float32, device=device)
        dverts = z(B, d.V, 3) if dverts is None else dverts.contiguous().float().clone()
        dJtr = None
        dtr_j = None
        if djoints is not None:
            djoints = djoints.contiguous().float()
            dJtr = djoints[:, :d.nj].contiguous()
            dtr_j = dJtr.sum(1)
            ne = len(d.extra_ids)
            dverts.index_add_(1, dev.t['extra_ids'].long(), djoints[:, d.nj:d.nj + ne])
            dl = djoints[:, d.nj + ne:]                                             # [B,51,3]
            contrib = dl.unsqueeze(2) * dev.t['lmk_bary'].view(1, -1, 3, 1)          # [B,51,3(f),3]
            dverts.index_add_(1, dev.t['lmk_rows'].long().view(-1), contrib.reshape(B, -1, 3))
        uset, _uset_keep = dev.vertex_set('all', np.arange(d.V), frames=B)     # private scratch for this call
        dvp, dA, dtransl, dX = z(B, uset.NCs), z(B, d.nj, 12), z(B, 3), z(B, K_PAD)
        lib.check(lib.lbs_verts_bwd(C.byref(dev.skin), C.byref(uset), ptr(tt['A']), d.nj, ptr(ctx.v_posed), d.V,
                                    ptr(dverts), B, Bp, ptr(dvp), ptr(dA), ptr(dtransl), ptr(dX), s), 'lbs_verts_bwd')
        g = dict(go=z(B, 3), body=z(B, 63), jaw=z(B, 3), leye=z(B, 3), reye=z(B, 3), lh=z(B, ctx.hand_dim),
                 rh=z(B, ctx.hand_dim), betas=z(B, d.nshape // 2), expr=z(B, d.nshape // 2))
        gi = _hip.PoseGradIn(ptr(dA), ptr(dJtr), ptr(dX))
        go = _hip.PoseGradOut(ptr(g['go']), ptr(g['body']), ptr(g['jaw']), ptr(g['leye']), ptr(g['reye']), ptr(g['lh']),
                              ptr(g['rh']), ctx.hand_dim, ptr(g['betas']), ptr(g['expr']))
        lib.check(lib.smplx_pose_bwd(C.byref(dev.body), C.byref(ctx.ws), C.byref(gi), C.byref(go), B, s), 'smplx_pose_bwd')
        if dfp is not None:                       # return_full_pose consumers (rare): plain torch plumbing
            dfp = dfp.float()
            g['go'] += dfp[:, 0:3]; g['body'] += dfp[:, 3:66]; g['jaw'] += dfp[:, 66:69]
            g['leye'] += dfp[:, 69:72]; g['reye'] += dfp[:, 72:75]
            if d.ncomp > 0:
                g['lh'] += dfp[:, 75:120] @ dev.t['lh_comp'].T
                g['rh'] += dfp[:, 120:165] @ dev.t['rh_comp'].T
            else:
                g['lh'] += dfp[:, 75:120]; g['rh'] += dfp[:, 120:165]
        gtr = None
        if ctx.has_transl:
            gtr = dtransl if dtr_j is None else dtransl + dtr_j
        return (None, None, g['betas'], g['expr'], g['go'], g['body'], g['jaw'], g['leye'], g['reye'], g['lh'],
                g['rh'], gtr)


class ModelOutput:
    """Attribute bag with the fields of smplx's ``ModelOutput`` (fitting_temp_slide.py:573-624)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def get(self, k, default=None):
        return self.__dict__.get(k, default)


class SMPLX(nn.Module):
    """Drop-in for ``smplx.SMPLX`` on the calls LEMO makes (see module docstring)."""

    NUM_BODY_JOINTS = 21
    NUM_JOINTS = 54

    def __init__(self, model_path, gender='neutral', ext='npz', num_pca_comps=12, use_pca=True, flat_hand_mean=False,
                 num_betas=10, batch_size=1, joint_mapper=None, create_global_orient=True, create_body_pose=True,
                 create_betas=True, create_left_hand_pose=True, create_right_hand_pose=True, create_expression=True,
                 create_jaw_pose=True, create_leye_pose=True, create_reye_pose=True, create_transl=True,
                 dtype=torch.float32, extra_joint_ids=None, _lib: Optional[_hip.HipLib] = None, **kwargs):
        super().__init__()
        assert dtype == torch.float32, 'the LEMO fitting path is fp32'
        self.data = BodyModelData(load_model_dict(model_path, gender, ext), num_pca_comps, use_pca, flat_hand_mean,
                                  num_betas, extra_joint_ids)
        if not self.data.extra_ids_in_range:
            raise ValueError(f'extra joint vertex ids reach {int(self.data.extra_ids.max())} but the model has {self.data.V} vertices '
                             '(the default ids are SMPL-X\'s: pass extra_joint_ids for a smaller model)')
        self.batch_size, self.joint_mapper, self.use_pca = batch_size, joint_mapper, use_pca
        self.num_pca_comps, self.gender, self.dtype = num_pca_comps, gender, dtype
        self._lib_override = _lib
        self._dev = {}
        hd = num_pca_comps if use_pca else 45
        B = batch_size
        for name, dim, create in (('betas', num_betas, create_betas), ('global_orient', 3, create_global_orient),
                                  ('body_pose', 63, create_body_pose), ('left_hand_pose', hd, create_left_hand_pose),
                                  ('right_hand_pose', hd, create_right_hand_pose), ('jaw_pose', 3, create_jaw_pose),
                                  ('leye_pose', 3, create_leye_pose), ('reye_pose', 3, create_reye_pose),
                                  ('expression', 10, create_expression), ('transl', 3, create_transl)):
            if create:
                self.register_parameter(name, nn.Parameter(torch.zeros(B, dim, dtype=dtype), requires_grad=True))
        self.register_buffer('faces_tensor', torch.from_numpy(self.data.faces), persistent=False)
        self.faces = self.data.faces

    def get_num_verts(self):
        return self.data.V

    @torch.no_grad()
    def reset_params(self, **params_dict):
        for name, p in self.named_parameters():
            if name in params_dict:
                p[:] = torch.as_tensor(params_dict[name], dtype=p.dtype, device=p.device).reshape(p.shape)
            else:
                p.fill_(0)

    def _device_body(self, device) -> DeviceBody:
        key = str(device)
        if key not in self._dev:
            self._dev[key] = DeviceBody(self.data, device)
        return self._dev[key]

    def forward(self, betas=None, global_orient=None, body_pose=None, left_hand_pose=None, right_hand_pose=None,
                transl=None, expression=None, jaw_pose=None, leye_pose=None, reye_pose=None, return_verts=True,
                return_full_pose=False, **kwargs):
        pick = lambda v, name: v if v is not None else getattr(self, name, None)
        go = pick(global_orient, 'global_orient')
        body = pick(body_pose, 'body_pose')
        if go is None or body is None:
            raise ValueError('global_orient and body_pose are required')
        B = max(go.shape[0], body.shape[0])
        dev_t = go.device
        zeros = lambda dim: torch.zeros(B, dim, dtype=torch.float32, device=dev_t)
        betas = pick(betas, 'betas')
        betas = zeros(10) if betas is None else betas
        if betas.shape[0] != B:
            betas = betas.expand(B, -1)
        expression = pick(expression, 'expression')
        expression = zeros(10) if expression is None else expression
        if expression.shape[0] != B:
            expression = expression.expand(B, -1)
        hd = self.num_pca_comps if self.use_pca else 45
        lh = pick(left_hand_pose, 'left_hand_pose'); lh = zeros(hd) if lh is None else lh
        rh = pick(right_hand_pose, 'right_hand_pose'); rh = zeros(hd) if rh is None else rh
        jaw = pick(jaw_pose, 'jaw_pose'); jaw = zeros(3) if jaw is None else jaw
        leye = pick(leye_pose, 'leye_pose'); leye = zeros(3) if leye is None else leye
        reye = pick(reye_pose, 'reye_pose'); reye = zeros(3) if reye is None else reye
        transl = pick(transl, 'transl')
        lib = self._lib_override or _hip.get_lib()
        verts, joints, fp = _SmplxFn.apply(self._device_body(dev_t), lib, betas, expression, go, body, jaw, leye, reye,
                                           lh, rh, transl)
        if self.joint_mapper is not None:
            joints = self.joint_mapper(joints)
        if self.use_pca:                     # smplx returns the 45-D expanded hands (without the mean)
            lh_out = lh @ self._device_body(dev_t).t['lh_comp']
            rh_out = rh @ self._device_body(dev_t).t['rh_comp']
        else:
            lh_out, rh_out = lh, rh
        return ModelOutput(vertices=verts if return_verts else None, joints=joints, betas=betas, expression=expression,
                           global_orient=go, body_pose=body, left_hand_pose=lh_out, right_hand_pose=rh_out,
                           jaw_pose=jaw, full_pose=fp if return_full_pose else None)


def create(model_path, model_type: str = 'smplx', **kwargs) -> SMPLX:
    """``smplx.create`` for ``model_type='smplx'`` (the only type LEMO instantiates)."""
    if model_type.lower() != 'smplx':
        raise ValueError('lemo_amd implements the SMPL-X model only (LEMO never creates another type)')
    return SMPLX(model_path, **kwargs)
