"""Build-container-only harness that RUNS the reference's own Python where it lies under /root/reference.

Nothing here is copied from the reference: its modules are imported (``temp_prox.*``, ``utils.utils``, ``models.*``,
``human_body_prior.*``) or, for the body of ``opt_amass_temp.py::optimize()`` -- which cannot be imported because
the module parses ``sys.argv`` and touches CUDA at import time -- a range of its source lines is read from the
file AT RUN TIME, dedented and ``exec``'d.  Only numbers leave this file (``oracle_vs_reference.txt`` rows and
``.npz`` fixtures written by ``make_golden.py``).

Third-party modules that are absent in this image are replaced by ``types.ModuleType`` stubs:
  torchvision, open3d, tensorboardX, configer, chamfer   -- imported but never called on the S2/S3/AMASS paths
  torchgeometry                                          -- the two functions the path calls, from the oracle's
                                                            restatement of tgm 0.1.2 (parity unpinned)
  smplx                                                  -- ``smplx.lbs`` is the reference's vendored
                                                            ``human_body_prior/body_model/lbs.py``; ``SMPLX.forward``
                                                            glue (smplx 0.1.26, absent) comes from ``RefSmplx`` below
"""
import importlib
import importlib.util
import os
import sys
import textwrap
import types
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
REF = '/root/reference'
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import lemo_oracle as O                      # noqa: E402

_state = {}


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_lbs():
    """the reference's vendored smplx.lbs (file-path import), with the torch>=2 ``.view`` fix of SURVEY 8(c)"""
    if 'lbs' not in _state:
        m = load_by_path('ref_lbs', f'{REF}/human_body_prior/body_model/lbs.py')
        _v2j = m.vertices2joints
        m.vertices2joints = lambda J, v: _v2j(J, v).contiguous()
        _state['lbs'] = m
    return _state['lbs']


def install_stubs():
    """Make ``import utils.utils``, ``models.*``, ``temp_prox.*`` and ``human_body_prior.train.vposer_smpl`` work."""
    if _state.get('stubs'):
        return
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name in ('torchvision', 'open3d', 'tensorboardX', 'chamfer'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['tensorboardX'].SummaryWriter = object
    tgm = types.ModuleType('torchgeometry')
    tgm.rotation_matrix_to_angle_axis = O.rotation_matrix_to_angle_axis
    tgm.angle_axis_to_rotation_matrix = O.angle_axis_to_rotation_matrix
    sys.modules['torchgeometry'] = tgm
    cfg = types.ModuleType('configer'); cfg.Configer = object
    sys.modules['configer'] = cfg
    lbs = ref_lbs()
    smplx_stub = types.ModuleType('smplx'); smplx_stub.lbs = lbs
    smplx_lbs = types.ModuleType('smplx.lbs')
    for k in ('lbs', 'transform_mat', 'batch_rodrigues', 'vertices2joints', 'blend_shapes', 'batch_rigid_transform'):
        setattr(smplx_lbs, k, getattr(lbs, k))
    sys.modules['smplx'] = smplx_stub; sys.modules['smplx.lbs'] = smplx_lbs
    import scipy.ndimage
    if 'scipy.ndimage.filters' not in sys.modules:
        try:
            import scipy.ndimage.filters                                  # noqa: F401
        except Exception:
            sys.modules['scipy.ndimage.filters'] = scipy.ndimage
    _state['stubs'] = True


def ref_utils():
    install_stubs()
    return importlib.import_module('utils.utils')


def ref_enc(real_weights=True):
    """the reference's ``Enc`` (models/AE_sep.py) with its own trained ``runs/15217`` weights"""
    install_stubs()
    from models.AE_sep import Enc
    enc = Enc(downsample=False, z_channel=64)
    if real_weights:
        enc.load_state_dict(torch.load(f'{REF}/runs/15217/Enc_last_model.pkl', map_location='cpu'))
    enc.eval()
    for p in enc.parameters():
        p.requires_grad = False
    return enc


def ref_ae(weights):
    install_stubs()
    from models.AE import AE
    ae = AE(downsample=True, in_channel=4, kernel=3)
    ae.load_state_dict(weights)
    return ae


def ref_vposer(w):
    """the reference's VPoser class (human_body_prior/train/vposer_smpl.py) with seeded decoder weights ``w``"""
    install_stubs()
    vp_mod = importlib.import_module('human_body_prior.train.vposer_smpl')
    vp = vp_mod.VPoser(num_neurons=512, latentD=32, data_shape=[1, 21, 3]).eval()
    vp.load_state_dict({**vp.state_dict(), **w})
    return vp


ModelOutput = namedtuple('ModelOutput', ['vertices', 'joints', 'full_pose', 'betas', 'global_orient', 'body_pose',
                                         'expression', 'left_hand_pose', 'right_hand_pose', 'jaw_pose'])


class RefSmplx(nn.Module):
    """Stand-in for pip ``smplx==0.1.26``'s ``SMPLX`` module (absent; *parity unpinned* for its glue): the
    forward glue follows ``oracle.SmplxOracle`` but the heavy part is the REFERENCE's vendored ``lbs()``; parameters,
    ``joint_mapper`` and the output field names are the ones the reference's callers use
    (utils/utils.py:152-169, fitting_temp_slide.py:236-258, fit_temp_loadprox_slide.py:499-515)."""

    PARAMS = (('betas', 10), ('global_orient', 3), ('transl', 3), ('left_hand_pose', 12), ('right_hand_pose', 12),
              ('jaw_pose', 3), ('leye_pose', 3), ('reye_pose', 3), ('expression', 10))

    def __init__(self, so: O.SmplxOracle, batch_size: int, joint_mapper=None):
        super().__init__()
        self.so = so
        self.joint_mapper = joint_mapper
        for n, d in self.PARAMS:
            self.register_parameter(n, nn.Parameter(torch.zeros(batch_size, d)))
        self.register_buffer('faces_tensor', so.faces.clone())

    @torch.no_grad()
    def reset_params(self, **kw):
        for n, p in self.named_parameters():
            if n in kw:
                p[:] = torch.as_tensor(kw[n], dtype=p.dtype)
            else:
                p.fill_(0)

    def forward(self, betas=None, global_orient=None, body_pose=None, left_hand_pose=None, right_hand_pose=None,
                transl=None, expression=None, jaw_pose=None, leye_pose=None, reye_pose=None, return_verts=True,
                return_full_pose=False, **kwargs):
        g = lambda v, n: getattr(self, n) if v is None else v
        betas, global_orient, transl = g(betas, 'betas'), g(global_orient, 'global_orient'), g(transl, 'transl')
        lh, rh = g(left_hand_pose, 'left_hand_pose'), g(right_hand_pose, 'right_hand_pose')
        expression, jaw = g(expression, 'expression'), g(jaw_pose, 'jaw_pose')
        leye, reye = g(leye_pose, 'leye_pose'), g(reye_pose, 'reye_pose')
        so = self.so
        fp = so.full_pose(global_orient, body_pose, lh, rh, jaw, leye, reye)
        shape_comp = torch.cat([betas, expression], dim=-1)
        shapedirs = torch.cat([so.shapedirs, so.expr_dirs], dim=-1)
        verts, joints = ref_lbs().lbs(shape_comp, fp, so.v_template, shapedirs, so.posedirs, so.J_regressor, so.parents,
                                      so.lbs_weights)
        lmk_vertices = verts[:, so.faces[so.lmk_faces_idx]]
        landmarks = torch.einsum('blfi,lf->bli', [lmk_vertices, so.lmk_bary])
        joints = torch.cat([joints, verts[:, so.extra_ids], landmarks], dim=1)
        if self.joint_mapper is not None:
            joints = self.joint_mapper(joints)
        joints = joints + transl.unsqueeze(1)
        verts = verts + transl.unsqueeze(1)
        lh45 = torch.einsum('bi,ij->bj', [lh, so.lh_comp]) if so.use_pca else lh
        rh45 = torch.einsum('bi,ij->bj', [rh, so.rh_comp]) if so.use_pca else rh
        return ModelOutput(vertices=verts, joints=joints, full_pose=fp, betas=betas, global_orient=global_orient,
                           body_pose=body_pose, expression=expression, left_hand_pose=lh45, right_hand_pose=rh45,
                           jaw_pose=jaw)


def exec_reference_lines(path, first, last, namespace):
    """exec lines [first, last] (1-based, inclusive) of a reference source file, read at run time, in ``namespace``"""
    with open(path) as f:
        lines = f.readlines()[first - 1:last]
    exec(compile(textwrap.dedent(''.join(lines)), f'{path}:{first}-{last}', 'exec'), namespace)
    return namespace


# ------------------------------------------------------------------------------------------------------------------
# AMASS: the loop body opt_amass_temp.py:355-453 run as text
# ------------------------------------------------------------------------------------------------------------------
def run_amass_loop_body(so: O.SmplxOracle, vposer_w, ids, Xmean, Xstd, init_params, markers_rec, contact_lbl,
                        weights=None, steps=0):
    """Set up the variables of opt_amass_temp.py:332-345 and exec the reference's loop-body text (355-453) followed
    by ``loss.backward(); optimizer.step()`` as at :454-455.  Returns losses + grads of iteration 0 and, when
    ``steps`` > 0, the parameters after each further Adam step."""
    import torch.nn.functional as F
    U = ref_utils()
    B = init_params.shape[0]
    w = dict(O.LOSS_WEIGHTS if weights is None else weights)
    args = types.SimpleNamespace(weight_loss_rec_markers=w['rec_markers'], weight_loss_contact_vel=w['contact_vel'],
                                 weight_loss_smooth=w['smooth'], weight_loss_vposer=w['vposer'],
                                 weight_loss_shape=w['shape'], weight_loss_hand=w['hand'])
    device = torch.device('cpu')
    ip = np.array(init_params, np.float32)              # a copy: torch.from_numpy shares memory and Adam updates in place
    ns = dict(torch=torch, F=F, np=np, args=args, device=device,
              convert_to_3D_rot=U.convert_to_3D_rot, gen_body_mesh_v1=U.gen_body_mesh_v1,
              gen_body_joints_v1=U.gen_body_joints_v1,
              smplx_model=RefSmplx(so, B), vposer_model=ref_vposer(vposer_w), smooth_encoder=ref_enc(),
              infill_marker_ids=[int(i) for i in ids['markers67']], smooth_marker_ids=[int(i) for i in ids['markers81']],
              left_heel_verts_id=np.asarray(ids['left_heel']), right_heel_verts_id=np.asarray(ids['right_heel']),
              left_toe_verts_id=np.asarray(ids['left_toe']), right_toe_verts_id=np.asarray(ids['right_toe']),
              Xmean_global_markers=torch.from_numpy(np.asarray(Xmean)).float(),
              Xstd_global_markers=torch.from_numpy(np.asarray(Xstd)).float(),
              markers_rec_t=torch.from_numpy(np.asarray(markers_rec, np.float32)),
              contact_lbl_rec=torch.from_numpy(np.asarray(contact_lbl, np.float32)))
    # :332-345 (same statements on the harness's inputs)
    ns['transl_opt_t'] = torch.from_numpy(ip[:, 0:3]).float()
    ns['rot_6d_opt_t'] = U.convert_to_6D_all(torch.from_numpy(ip[:, 3:6]).float()).detach().clone()
    ns['shape_t'] = torch.from_numpy(ip[:, 6:16]).float()
    ns['other_params_opt_t'] = torch.from_numpy(ip[:, 16:]).float()
    for k in ('transl_opt_t', 'rot_6d_opt_t', 'other_params_opt_t'):
        ns[k].requires_grad = True
    final = [ns['transl_opt_t'], ns['rot_6d_opt_t'], ns['other_params_opt_t']]
    opt = torch.optim.Adam(final, lr=0.01)
    out = {}
    for step in range(steps + 1):
        if step > 60:
            for gparam in opt.param_groups:
                gparam['lr'] = 0.005
        opt.zero_grad()
        exec_reference_lines(f'{REF}/opt_amass_temp.py', 355, 453, ns)
        ns['loss'].backward(retain_graph=True)
        if step == 0:
            out.update(total=float(ns['loss']), marker=float(ns['loss_marker']), vposer=float(ns['loss_vposer']),
                       shape=float(ns['loss_shape']), hand=float(ns['loss_hand']), contact=float(ns['loss_contact_vel']),
                       smooth=float(ns['loss_smooth']), g_transl=final[0].grad.numpy().copy(),
                       g_rot6d=final[1].grad.numpy().copy(), g_other=final[2].grad.numpy().copy(),
                       p72=ns['body_params_opt_t_72'].detach().numpy().copy(),
                       verts=ns['body_verts_opt_t'].detach().numpy().copy())
        if steps:
            opt.step()
            out.setdefault('p75_hist', []).append(torch.cat([final[0], final[1], ns['shape_t'], final[2]], -1).detach().numpy().copy())
            out.setdefault('total_hist', []).append(float(ns['loss']))
    return out


# ------------------------------------------------------------------------------------------------------------------
# Teacher-forcing fixtures: the reference's own LOOPS (optimizer construction, lr schedule, step) run as text with a
# passive recorder in place of ``optim.Adam`` -- the recorder IS torch.optim.Adam (a subclass that copies the optimiser
# state before and after the update of selected steps and changes nothing)
# ------------------------------------------------------------------------------------------------------------------
class AdamRecorder:
    """``recorder.Adam(params, lr=...)`` is what the reference's ``optim.Adam(final_params, lr=init_lr)`` line gets.
    For optimiser number ``o`` (in construction order) and its ``k``-th ``step()`` call (0-based) with ``(o, k)`` in
    ``record_at``: ``records[(o, k)]`` = dict(before=state, after=state, lr=the group's lr at the call, grads=[...],
    extra={name: float(ns[name])}); state = dict(params=[...], exp_avg=[...], exp_avg_sq=[...], step=completed steps)."""

    def __init__(self, record_at, ns=None, extra_names=()):
        self.record_at, self.ns, self.extra_names = set(record_at), ns, tuple(extra_names)
        self.records, self.n_opt = {}, 0

    @staticmethod
    def _snap(opt):
        ps = opt.param_groups[0]['params']
        st = [opt.state.get(p, {}) for p in ps]
        z = lambda p, s, k: (s[k].detach().numpy().copy() if k in s else np.zeros_like(p.detach().numpy()))
        step = int(st[0]['step']) if st and 'step' in st[0] else 0
        return dict(params=[p.detach().numpy().copy() for p in ps], exp_avg=[z(p, s, 'exp_avg') for p, s in zip(ps, st)],
                    exp_avg_sq=[z(p, s, 'exp_avg_sq') for p, s in zip(ps, st)], step=step)

    def Adam(self, params, **kw):
        rec, o = self, self.n_opt
        self.n_opt += 1

        class _Adam(torch.optim.Adam):
            _calls = 0

            def step(self, closure=None):
                k = self._calls
                self._calls += 1
                if (o, k) not in rec.record_at:
                    return super().step(closure)
                assert closure is None
                r = dict(before=rec._snap(self), lr=float(self.param_groups[0]['lr']),
                         grads=[p.grad.detach().numpy().copy() for p in self.param_groups[0]['params']],
                         extra={n: float(rec.ns[n]) for n in rec.extra_names} if rec.ns is not None else {})
                out = super().step()
                r['after'] = rec._snap(self)
                rec.records[(o, k)] = r
                return out
        return _Adam(params, **kw)


AMASS_LOSS_VARS = ('loss_marker', 'loss_vposer', 'loss_shape', 'loss_hand', 'loss_contact_vel', 'loss_smooth', 'loss')


def run_amass_loop_text(so: O.SmplxOracle, vposer_w, ids, Xmean, Xstd, init_params, markers_rec, contact_lbl, record_at,
                        weights=None):
    """exec opt_amass_temp.py:331-455 -- parameter init, ``optim.Adam(final_params, lr=init_lr)``, the WHOLE 100-step loop with
    its own lr switch (:350-352), ``loss.backward``, ``optimizer.step()`` -- with an :class:`AdamRecorder` as ``optim``.
    ``record_at``: iterable of step indices.  Returns (recorder.records keyed by step, final body_params_opt_t_72)."""
    import torch.nn.functional as F
    U = ref_utils()
    B = init_params.shape[0]
    w = dict(O.LOSS_WEIGHTS if weights is None else weights)
    args = types.SimpleNamespace(weight_loss_rec_markers=w['rec_markers'], weight_loss_contact_vel=w['contact_vel'],
                                 weight_loss_smooth=w['smooth'], weight_loss_vposer=w['vposer'],
                                 weight_loss_shape=w['shape'], weight_loss_hand=w['hand'])
    ns = dict(torch=torch, F=F, np=np, args=args, device=torch.device('cpu'),
              convert_to_3D_rot=U.convert_to_3D_rot, convert_to_6D_all=U.convert_to_6D_all, gen_body_mesh_v1=U.gen_body_mesh_v1,
              gen_body_joints_v1=U.gen_body_joints_v1,
              smplx_model=RefSmplx(so, B), vposer_model=ref_vposer(vposer_w), smooth_encoder=ref_enc(),
              infill_marker_ids=[int(i) for i in ids['markers67']], smooth_marker_ids=[int(i) for i in ids['markers81']],
              left_heel_verts_id=np.asarray(ids['left_heel']), right_heel_verts_id=np.asarray(ids['right_heel']),
              left_toe_verts_id=np.asarray(ids['left_toe']), right_toe_verts_id=np.asarray(ids['right_toe']),
              Xmean_global_markers=torch.from_numpy(np.asarray(Xmean)).float(),
              Xstd_global_markers=torch.from_numpy(np.asarray(Xstd)).float(),
              markers_rec_t=torch.from_numpy(np.asarray(markers_rec, np.float32)),
              contact_lbl_rec=torch.from_numpy(np.asarray(contact_lbl, np.float32)),
              init_params=np.array(init_params, np.float32))
    rec = AdamRecorder([(0, int(k)) for k in record_at], ns, AMASS_LOSS_VARS)
    ns['optim'] = types.SimpleNamespace(Adam=rec.Adam)
    exec_reference_lines(f'{REF}/opt_amass_temp.py', 331, 455, ns)
    assert rec.n_opt == 1
    return {k: r for (_, k), r in rec.records.items()}, ns['body_params_opt_t_72'].detach().numpy().copy()


# ------------------------------------------------------------------------------------------------------------------
# PROX: the reference's own SMPLifyLoss / FittingMonitor closure / camera / priors / JointMapper / optimizer factory
# ------------------------------------------------------------------------------------------------------------------
def ref_prox_modules():
    """import temp_prox.{fitting_temp_slide, camera, prior, misc_utils, optimizers.optim_factory} under the stubs"""
    install_stubs()
    if 'prox' not in _state:
        fitting = importlib.import_module('temp_prox.fitting_temp_slide')
        camera = importlib.import_module('temp_prox.camera')
        prior = importlib.import_module('temp_prox.prior')
        misc = importlib.import_module('temp_prox.misc_utils')
        optf = importlib.import_module('temp_prox.optimizers.optim_factory')
        _state['prox'] = types.SimpleNamespace(fitting=fitting, camera=camera, prior=prior, misc=misc, optf=optf)
    return _state['prox']


class RefProxWindow:
    """One PROX window driven through the reference's own objects, wired the way ``main_slide.py:120-238`` and
    ``fit_temp_loadprox_slide.py:253-545`` wire them (S2 / S3 YAML values arrive through ``prob['weights']``).

    ``prob`` is a ``__graft_entry__.prox_small_problem``-shaped dict.  The vertex-id tables SMPLifyLoss reads from
    the reference's json/npy files at construction are the real (V = 10475) ones; for a reduced synthetic model they
    are overwritten by the problem's ids after construction (plain attributes)."""

    def __init__(self, prob, first_batch_flag=False, ae_weights=None, product_lib=None):
        """``product_lib``: None -> the reference's own stack (oracle-backed ``RefSmplx`` around the vendored ``lbs()``, the
        reference's VPoser and Enc classes).  A ``lemo_amd._hip.HipLib`` (the host-emulated build of the UNMODIFIED kernel
        sources) -> the drop-in proof of the PROX side of the boundary: ``lemo_amd.compat`` smplx.create,
        ``lemo_amd.vposer.VPoser`` and ``lemo_amd.priors.Enc`` take their places and the reference's own
        ``FittingMonitor.create_fitting_closure`` / ``SMPLifyLoss.forward`` / ``optim_factory`` run on top of them,
        unmodified (``F.grid_sample`` etc. stay torch: they are the reference's own lines)."""
        M = ref_prox_modules()
        B, w = prob['B'], dict(prob['weights'])
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        so = O.SmplxOracle(prob['model'], extra_joint_ids=list(range(21)) if prob['V'] < 9930 else None)
        self.so = so
        joint_mapper = M.misc.JointMapper(np.asarray(prob['joint_map']))
        if product_lib is None:
            self.body_model = RefSmplx(so, B, joint_mapper=joint_mapper)
            self.vposer = ref_vposer({k: torch.from_numpy(v) for k, v in prob['vposer_w'].items()})
            smooth_model = ref_enc()
        else:
            from lemo_amd.compat import smplx as compat_smplx
            from lemo_amd.priors import Enc
            from lemo_amd.vposer import VPoser
            # main_slide.py:160-179: smplx.create(model_path, joint_mapper=..., create_body_pose=not use_vposer, **args)
            self.body_model = compat_smplx.create(prob['model'], model_type='smplx', gender='male', ext='npz', num_pca_comps=12,
                                                  joint_mapper=joint_mapper, create_global_orient=True, create_body_pose=False,
                                                  create_betas=True, create_left_hand_pose=True, create_right_hand_pose=True,
                                                  create_expression=True, create_jaw_pose=True, create_leye_pose=True,
                                                  create_reye_pose=True, create_transl=True, batch_size=B, dtype=torch.float32,
                                                  extra_joint_ids=list(range(21)) if prob['V'] < 9930 else None, _lib=product_lib)
            self.vposer = VPoser(_lib=product_lib).eval()
            self.vposer.load_state_dict({**self.vposer.state_dict(), **{k: torch.from_numpy(v) for k, v in prob['vposer_w'].items()}})
            smooth_model = Enc(_lib=product_lib)
            smooth_model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in prob['enc_w'].items()})
            smooth_model.eval()
            for p_ in smooth_model.parameters():
                p_.requires_grad = False
        cam = prob.get('cam') or dict(fx=1060.53, fy=1060.38, cx=951.30, cy=536.77)    # PROXD_temp_S2.yaml:111-114
        self.camera = M.camera.create_camera(focal_length_x=cam['fx'], focal_length_y=cam['fy'],
                                             center=torch.tensor([cam['cx'], cam['cy']]).view(-1, 2), batch_size=B,
                                             dtype=torch.float32)
        self.camera.rotation.requires_grad = False
        l2 = lambda: M.prior.create_prior(prior_type='l2', dtype=torch.float32)
        angle_prior = M.prior.create_prior(prior_type='angle', dtype=torch.float32)
        infill = prob.get('infill') or {}
        use_infill = w['motion_infill_rec_weight'] > 0 or w['motion_infill_contact_weight'] > 0
        motion_infill_model = ref_ae(ae_weights) if (use_infill and ae_weights is not None) else None
        sdf = f(prob['sdf'])
        D = sdf.shape[0]
        cwd = os.getcwd()
        os.chdir(f'{REF}/temp_prox')            # SMPLifyLoss.__init__ opens '../loader/...', '../preprocess_stats/...'
        try:
            self.loss = M.fitting.create_loss(
                loss_type='smplify', use_joints_conf=True, use_face=True, use_hands=True, body_pose_prior=l2(),
                shape_prior=l2(), angle_prior=angle_prior, expr_prior=l2(), left_hand_prior=l2(), right_hand_prior=l2(),
                jaw_prior=l2(), interpenetration=False, s2m=False, m2s=False, sdf_penetration=True,
                grid_min=f(prob['grid_min']).repeat(B, 1).unsqueeze(1), grid_max=f(prob['grid_max']).repeat(B, 1).unsqueeze(1),
                sdf=sdf.repeat(B, 1, 1, 1).unsqueeze(1), sdf_normals=None, voxel_size=None,
                R=f(prob['R']), t=f(prob['t']).reshape(1, 3), contact=False, dtype=torch.float32, smooth_acc=False,
                smooth_vel=False, use_motion_smooth_prior=True, motion_smooth_model=smooth_model, use_friction=True,
                contact_fric_verts_ids=np.asarray(prob['fric_ids']), use_motion_infill_prior=use_infill,
                motion_infill_model=motion_infill_model, infill_pretrain_weights=ae_weights, device=torch.device('cpu'))
        finally:
            os.chdir(cwd)
        L = self.loss
        if prob['V'] < 10475:                   # reduced model: the problem's own id tables
            L.smooth_marker_ids = [int(i) for i in prob['ids']['markers81']]
            L.infill_marker_ids = [int(i) for i in prob['ids']['markers67']]
            for k in ('left_heel', 'right_heel', 'left_toe', 'right_toe'):
                setattr(L, k + '_verts_id', np.asarray(prob['ids'][k]))
        L.Xmean_global_markers = f(np.asarray(prob['Xmean'])).view(1, 1, -1)
        L.Xstd_global_markers = f(np.asarray(prob['Xstd']))
        # fit_temp_loadprox_slide.py:499-528
        self.body_model.reset_params(**{k: v for k, v in prob['params'].items() if k != 'pose_embedding'})
        self.pose_embedding = f(prob['params']['pose_embedding']).clone()
        self.pose_embedding.requires_grad = True
        self.body_model.betas.requires_grad = False
        final_params = [p for p in self.body_model.parameters() if p.requires_grad] + [self.pose_embedding]
        self.final_params = final_params
        self.optimizer, create_graph = M.optf.create_optimizer(final_params, optim_type='adam', lr=0.005)
        self.optimizer.zero_grad()
        cw = {k: torch.tensor(v, dtype=torch.float32) for k, v in w.items()}
        cw['bending_prior_weight'] = 3.17 * cw['body_pose_weight']
        jw = torch.ones(1, 118)
        jw[:, [1, 9, 12]] = 0.0                 # data_parser_slide.py:238-250 (joints_to_ign)
        jw[:, 25:76] = cw['hand_weight']
        jw[:, 76:] = cw['face_weight']
        L.reset_loss_weights(cw)
        self.marker_mask = f(infill['marker_mask']) if 'marker_mask' in infill else torch.ones(B, 67)
        if 'body_markers_rec' in infill:        # per-window constants of opt_step == 0 handed in (:821-941)
            L.body_markers_rec = f(infill['body_markers_rec'])
            L.contact_lbl_rec = f(infill['contact_lbl_rec'])
        self.monitor = M.fitting.FittingMonitor(maxiters=1, model_type='smplx')
        self.monitor.steps = 1 if 'body_markers_rec' in infill or not use_infill else 0
        self.closure = self.monitor.create_fitting_closure(
            self.optimizer, self.body_model, camera=self.camera, gt_joints=f(prob['gt_joints']),
            joints_conf=f(prob['joints_conf']), marker_mask=self.marker_mask, joint_weights=jw, loss=self.loss,
            create_graph=create_graph, use_vposer=True, vposer=self.vposer, pose_embedding=self.pose_embedding,
            scan_tensor=None, scan_point_num=None, scene_v=None, return_verts=True, return_full_pose=True, writer=None,
            first_batch_flag=first_batch_flag)
        self.loss_dict = None
        _fwd = self.loss.forward

        def _capture(*a, **k):
            self.loss_dict = _fwd(*a, **k)
            return self.loss_dict
        self.loss.forward = _capture

    def iterate(self, n=1, record_at=None):
        """n x ``optimizer.step(closure)`` exactly like FittingMonitor.run_fitting (:196).  ``record_at``: call indices
        (counted from this object's construction) whose optimiser state before / after, gradients as the update saw them
        (after the closure's first-15 % erase) and loss_dict are kept in ``self.records``"""
        out = []
        for _ in range(n):
            k = self.n_calls = getattr(self, 'n_calls', -1) + 1
            keep = record_at is not None and k in record_at
            if keep:
                before = AdamRecorder._snap(self.optimizer)
            self.optimizer.step(self.closure)
            out.append({k_: float(v) for k_, v in self.loss_dict.items()})
            if keep:
                self.records = getattr(self, 'records', {})
                self.records[k] = dict(before=before, after=AdamRecorder._snap(self.optimizer), lr=float(self.optimizer.param_groups[0]['lr']),
                                       grads=[p.grad.detach().numpy().copy() for p in self.optimizer.param_groups[0]['params']],
                                       extra=dict(out[-1]))
        return out

    def param_names(self):
        """names of the optimised tensors in the optimiser's order (fit_temp_loadprox_slide.py:511-519)"""
        return [n for n, p in self.body_model.named_parameters() if p.requires_grad] + ['pose_embedding']

    def grads(self):
        bm = self.body_model
        return dict(pose_embedding=self.pose_embedding.grad.numpy().copy(), transl=bm.transl.grad.numpy().copy(),
                    global_orient=bm.global_orient.grad.numpy().copy())


# ------------------------------------------------------------------------------------------------------------------
# AMASS clip pipeline: opt_amass_temp.py:159-214 (mask + finetune + eval) and :256-329 (decode) run as text
# ------------------------------------------------------------------------------------------------------------------
def run_amass_finetune_text(ae_weights, clip_img, finetune_steps=60):
    """exec opt_amass_temp.py:159-214 for one ``data`` item.  Returns (clip_img_input [1,4,d+2,T+16],
    clip_img_rec [1,1,d,T], the row list ``upper_body_row`` the reference trained on, last finetune loss)."""
    import itertools
    import torch.nn.functional as F
    import torch.optim as optim
    install_stubs()
    weights = {k: v.clone() for k, v in ae_weights.items()}
    ns = dict(torch=torch, np=np, F=F, optim=optim, itertools=itertools, device=torch.device('cpu'),
              args=types.SimpleNamespace(body_mode='local_markers_4chan'), clip_img=clip_img.clone(),
              infill_model=ref_ae(weights), weights=weights, finetine_step_total=finetune_steps)
    exec_reference_lines(f'{REF}/opt_amass_temp.py', 159, 214, ns)
    return ns['clip_img_input'], ns['clip_img_rec'], list(ns.get('upper_body_row', [])), float(ns['loss']) if 'loss' in ns else None


def run_amass_decode_text(clip_img_rec, clip_img, rot_0_pivot):
    """exec opt_amass_temp.py:256-329 for clip i = 0.  Returns (contact_lbl_rec [T,4], markers_rec_t [T,67,3])."""
    import tempfile
    import torch.nn.functional as F
    U = ref_utils()
    stats = np.load(f'{REF}/preprocess_stats/preprocess_stats_infill_local_markers_4chan.npz')
    with tempfile.TemporaryDirectory() as tmp:
        ns = dict(torch=torch, np=np, F=F, device=torch.device('cpu'), i=0, save_folder=tmp, stats=stats,
                  args=types.SimpleNamespace(body_mode='local_markers_4chan'),
                  clip_img_rec_list=clip_img_rec.clone(), clip_img_list=clip_img.clone(),
                  rot_0_pivot_list=torch.as_tensor(np.asarray(rot_0_pivot, np.float64)).reshape(1, -1),
                  gender_list=torch.tensor([1]), smplx_model_male=None, smplx_model_female=None,
                  reconstruct_global_body=U.reconstruct_global_body)
        exec_reference_lines(f'{REF}/opt_amass_temp.py', 256, 329, ns)
    return ns['contact_lbl_rec'], ns['markers_rec_t']


# ------------------------------------------------------------------------------------------------------------------
# per-frame fit (BASELINE configs[0]): opt_amass_perframe.py:291-363 run as text
# ------------------------------------------------------------------------------------------------------------------
def run_perframe_text(so: O.SmplxOracle, vposer_w, markers67_ids, markers_rec, betas, steps=100, record_at=None):
    """exec the reference's per-frame loop.  ``total_steps = 100`` is a literal inside the text; ``steps`` < 100 is
    obtained by substituting that one literal (the lr switches at 60 / 80 are then simply never reached)."""
    import tempfile
    import torch.nn.functional as F
    import torch.optim as optim
    U = ref_utils()
    T = markers_rec.shape[0]
    path = f'{REF}/opt_amass_perframe.py'
    with open(path) as f:
        text = textwrap.dedent(''.join(f.readlines()[290:364]))
    assert text.count('total_steps = 100') == 1
    text = text.replace('total_steps = 100', f'total_steps = {int(steps)}')
    with tempfile.TemporaryDirectory() as tmp:
        if record_at is not None:           # [(frame, step), ...]: one optimiser per frame (:312), recorded passively
            rec = AdamRecorder(record_at, None, ('loss_marker', 'loss_vposer', 'loss_shape', 'loss_hand', 'loss'))
            optim = types.SimpleNamespace(Adam=rec.Adam)
        ns = dict(torch=torch, np=np, F=F, optim=optim, tqdm=lambda x: x, device=torch.device('cpu'), T=T, i=0, save_folder=tmp,
                  args=types.SimpleNamespace(weight_loss_rec_markers=1.0, weight_loss_vposer=0.02, weight_loss_shape=0.01,
                                             weight_loss_hand=0.01),
                  body_joints_rec=np.asarray(markers_rec, np.float64), beta_gt=torch.from_numpy(np.asarray(betas, np.float32)),
                  convert_to_6D_all=U.convert_to_6D_all, convert_to_3D_rot=U.convert_to_3D_rot, gen_body_mesh_v1=U.gen_body_mesh_v1,
                  smplx_model=RefSmplx(so, 1), vposer_model=ref_vposer(vposer_w), marker_ids=[int(v) for v in markers67_ids])
        if record_at is not None:
            rec.ns = ns
        exec(compile(text, f'{path}:291-364', 'exec'), ns)
    if record_at is not None:
        return ns['body_params_opt_cur_clip'], rec.records
    return ns['body_params_opt_cur_clip']


# ------------------------------------------------------------------------------------------------------------------
# Drop-in proof: the reference's loop-body TEXT against the PRODUCT modules (host-emulated kernel library)
# ------------------------------------------------------------------------------------------------------------------
def run_amass_loop_body_on_product(prob, markers_rec, emu_lib, steps=0):
    """exec opt_amass_temp.py:355-453 in a namespace whose smplx model, VPoser, smoothness encoder and
    ``convert_to_3D_rot`` are the lemo_amd drop-ins (``lemo_amd.compat`` smplx.create, ``lemo_amd.vposer.VPoser``,
    ``lemo_amd.priors.Enc``, ``lemo_amd.rotation``) running the UNMODIFIED kernel sources compiled for the host
    (liblemo_emu.so); ``gen_body_mesh_v1`` / ``gen_body_joints_v1`` stay the reference's own functions.  Backward is
    torch autograd through the HIP autograd Functions, the update ``optimizer.step()`` as at :454-455."""
    import functools
    import torch.nn.functional as F
    from lemo_amd import rotation
    from lemo_amd.compat import smplx as compat_smplx
    from lemo_amd.priors import Enc
    from lemo_amd.vposer import VPoser
    U = ref_utils()
    B = prob['B']
    w = dict(O.LOSS_WEIGHTS)
    args = types.SimpleNamespace(weight_loss_rec_markers=w['rec_markers'], weight_loss_contact_vel=w['contact_vel'],
                                 weight_loss_smooth=w['smooth'], weight_loss_vposer=w['vposer'],
                                 weight_loss_shape=w['shape'], weight_loss_hand=w['hand'])
    smplx_model = compat_smplx.create(prob['model'], model_type='smplx', gender='male', ext='npz', num_pca_comps=12,
                                      create_global_orient=True, create_body_pose=True, create_betas=True, create_left_hand_pose=True,
                                      create_right_hand_pose=True, create_expression=True, create_jaw_pose=True, create_leye_pose=True,
                                      create_reye_pose=True, create_transl=True, batch_size=B,
                                      extra_joint_ids=list(range(21)) if prob['V'] < 9930 else None, _lib=emu_lib)
    vposer_model = VPoser(_lib=emu_lib).eval()
    vposer_model.load_state_dict({**vposer_model.state_dict(), **{k: torch.from_numpy(v) for k, v in prob['vposer_w'].items()}})
    smooth_encoder = Enc(_lib=emu_lib)
    smooth_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in prob['enc_w'].items()})
    smooth_encoder.eval()
    for p in smooth_encoder.parameters():
        p.requires_grad = False
    ids, ip = prob['ids'], np.array(prob['seq']['init_params'], np.float32)     # a copy (Adam updates in place)
    ns = dict(torch=torch, F=F, np=np, args=args, device=torch.device('cpu'),
              convert_to_3D_rot=functools.partial(rotation.convert_to_3D_rot, _lib=emu_lib),
              gen_body_mesh_v1=U.gen_body_mesh_v1, gen_body_joints_v1=U.gen_body_joints_v1,
              smplx_model=smplx_model, vposer_model=vposer_model, smooth_encoder=smooth_encoder,
              infill_marker_ids=[int(i) for i in ids['markers67']], smooth_marker_ids=[int(i) for i in ids['markers81']],
              left_heel_verts_id=np.asarray(ids['left_heel']), right_heel_verts_id=np.asarray(ids['right_heel']),
              left_toe_verts_id=np.asarray(ids['left_toe']), right_toe_verts_id=np.asarray(ids['right_toe']),
              Xmean_global_markers=torch.from_numpy(np.asarray(prob['Xmean']).reshape(1, 1, -1)).float(),
              Xstd_global_markers=torch.from_numpy(np.asarray(prob['Xstd'])).float(),
              markers_rec_t=torch.from_numpy(np.asarray(markers_rec, np.float32)),
              contact_lbl_rec=torch.from_numpy(np.asarray(prob['seq']['contact_lbl'], np.float32)))
    ns['transl_opt_t'] = torch.from_numpy(ip[:, 0:3]).float()
    ns['rot_6d_opt_t'] = rotation.convert_to_6D_all(torch.from_numpy(ip[:, 3:6]).float()).detach().clone()
    ns['shape_t'] = torch.from_numpy(ip[:, 6:16]).float()
    ns['other_params_opt_t'] = torch.from_numpy(ip[:, 16:]).float()
    for k in ('transl_opt_t', 'rot_6d_opt_t', 'other_params_opt_t'):
        ns[k].requires_grad = True
    final = [ns['transl_opt_t'], ns['rot_6d_opt_t'], ns['other_params_opt_t']]
    opt = torch.optim.Adam(final, lr=0.01)
    out = {}
    for step in range(steps + 1):
        opt.zero_grad()
        exec_reference_lines(f'{REF}/opt_amass_temp.py', 355, 453, ns)
        ns['loss'].backward(retain_graph=True)
        if step == 0:
            out.update(total=float(ns['loss']), marker=float(ns['loss_marker']), vposer=float(ns['loss_vposer']),
                       shape=float(ns['loss_shape']), hand=float(ns['loss_hand']), contact=float(ns['loss_contact_vel']),
                       smooth=float(ns['loss_smooth']), g_transl=final[0].grad.numpy().copy(),
                       g_rot6d=final[1].grad.numpy().copy(), g_other=final[2].grad.numpy().copy())
        if steps:
            opt.step()
            out.setdefault('p75_hist', []).append(torch.cat([final[0], final[1], ns['shape_t'], final[2]], -1).detach().numpy().copy())
    return out


# ------------------------------------------------------------------------------------------------------------------
# PROX result pickles: the reference's writer (fit_temp_loadprox_slide.py:577-594) and reader (data_parser_slide.py:106-126)
# ------------------------------------------------------------------------------------------------------------------
def write_reference_result_pkls(rw: 'RefProxWindow', paths):
    """exec the reference's own result-writing lines on a fitted RefProxWindow; ``paths``: one file per frame"""
    import pickle
    ns = dict(batch_size=len(paths), camera=rw.camera, body_model=rw.body_model, use_vposer=True, pose_embedding=rw.pose_embedding,
              vposer=rw.vposer, results_list=[], result_fn_list=list(paths), pickle=pickle, torch=torch, np=np)
    exec_reference_lines(f'{REF}/temp_prox/fit_temp_loadprox_slide.py', 577, 594, ns)
    return ns['results_list']


def reference_read_prox_pkl(path):
    """the reference's reader, exec'd from its text (the module imports cv2 / open3d at the top)"""
    import pickle
    ns = dict(pickle=pickle)
    exec_reference_lines(f'{REF}/temp_prox/data_parser_slide.py', 106, 126, ns)
    return ns['read_prox_pkl'](path)
