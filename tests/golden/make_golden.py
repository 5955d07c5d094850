"""Pin the oracle against the importable pieces of the reference and emit golden vectors.

Run ONLY in the build container (needs /root/reference; it never travels to the GPU box):

    python tests/golden/make_golden.py

Step 1 cross-checks ``oracle/lemo_oracle.py`` against the reference's own Python, imported from
where it lies (nothing is copied):
    human_body_prior/body_model/lbs.py      (file-path import; vendored smplx.lbs)
    models/AE_sep.py::Enc, models/AE.py::AE (torchvision stubbed -- unused import)
    utils/utils.py 6-D / axis-angle helpers (torchgeometry stubbed by the oracle's restatement)
    human_body_prior/train/vposer_smpl.py::VPoser.decode (torchgeometry/configer/smplx stubbed)
Step 2 writes small ``.npz`` fixtures (inputs + expected outputs) next to this file.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import lemo_oracle as O                      # noqa: E402
from lemo_amd import synthetic                           # noqa: E402

torch.set_num_threads(8)


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def small_model(V=640, seed=3):
    m = synthetic.make_synthetic_smplx(seed=seed, V=V, F=1200)
    return m


def synthetic_marker_clip(seed=11, T=120):
    """pelvis + 67 markers of a body that walks a curved path and turns (SMPL-X axes: z up), + contact labels"""
    rng = np.random.default_rng(seed)
    t = np.arange(T) / 30.0
    heading = 0.6 * np.sin(0.7 * t) + 0.4 * t                        # turning while walking
    path = np.stack([np.cumsum(np.cos(heading)) * 0.03, np.cumsum(np.sin(heading)) * 0.03, np.zeros(T)], -1)
    rest = rng.normal(0, 0.25, (68, 3)); rest[:, 2] = np.abs(rest[:, 2]) * 2.0 + 0.05      # offsets in the body frame
    # give the four direction markers (26, 56 shoulders; 27, 57 hips -- utils.py:228) a left/right layout
    for idx, (side, h) in {26: (+1, 1.4), 56: (-1, 1.4), 27: (+1, 0.9), 57: (-1, 0.9)}.items():
        rest[idx + 1] = [0.0, 0.2 * side, h]
    c, s_ = np.cos(heading), np.sin(heading)
    R = np.stack([np.stack([c, -s_, np.zeros(T)], -1), np.stack([s_, c, np.zeros(T)], -1),
                  np.stack([np.zeros(T), np.zeros(T), np.ones(T)], -1)], 1)                 # [T,3,3] about z
    body = np.einsum('tij,mj->tmi', R, rest) + path[:, None] + rng.normal(0, 0.003, (T, 68, 3))
    body[:, 0] = path + np.array([0, 0, 0.95])
    contact = (rng.random((T, 4)) < 0.7).astype(np.float64)
    return body, contact


def decode_input_from_image(img):
    """what opt_amass_temp.py:300-323 feeds reconstruct_global_body: [T, 1+68+1, 3] = zero reference slot, the local
    pelvis + markers of channel 0, and the (dx, dz, dr) trajectory of channels 1-3"""
    T = img.shape[1]
    local = img[0, :, :-4].reshape(T, 68, 3)
    traj = np.stack([img[1, :, 0], img[2, :, 0], img[3, :, 0]], -1)[:, None]
    return np.concatenate([np.zeros((T, 1, 3)), local, traj], axis=1)


def check_against_reference():
    report = {}
    # ---- lbs.py --------------------------------------------------------------------------
    ref_lbs = _load_by_path('ref_lbs', f'{REF}/human_body_prior/body_model/lbs.py')
    _v2j = ref_lbs.vertices2joints
    ref_lbs.vertices2joints = lambda J, v: _v2j(J, v).contiguous()     # torch>=2 .view() fix
    m = small_model()
    so = O.SmplxOracle(m)
    g = torch.Generator().manual_seed(0)
    B = 4
    betas = torch.randn(B, 20, generator=g) * 0.5
    pose = torch.randn(B, 165, generator=g) * 0.3
    shapedirs = torch.cat([so.shapedirs, so.expr_dirs], -1)
    v_ref, j_ref = ref_lbs.lbs(betas, pose, so.v_template, shapedirs, so.posedirs, so.J_regressor,
                               so.parents, so.lbs_weights)
    v_o, j_o = O.lbs(betas, pose, so.v_template, shapedirs, so.posedirs, so.J_regressor, so.parents,
                     so.lbs_weights)
    report['lbs.verts'] = _rel(v_o, v_ref)
    report['lbs.joints'] = _rel(j_o, j_ref)
    report['lbs.rodrigues'] = _rel(O.batch_rodrigues(pose.view(-1, 3)), ref_lbs.batch_rodrigues(pose.view(-1, 3)))

    # ---- Enc / AE ------------------------------------------------------------------------
    sys.modules.setdefault('torchvision', types.ModuleType('torchvision'))
    sys.path.insert(0, REF)
    from models.AE_sep import Enc                                         # reference class
    from models.AE import AE
    enc = Enc(downsample=False, z_channel=64)
    sd = torch.load(f'{REF}/runs/15217/Enc_last_model.pkl', map_location='cpu')
    enc.load_state_dict(sd)
    enc.eval()
    x = torch.randn(1, 1, 245, 134, generator=g)
    with torch.no_grad():
        report['Enc.z'] = _rel(O.enc_forward(sd, x), enc(x)[0])
    ae = AE(downsample=True, in_channel=4, kernel=3).eval()
    ae.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_ae_weights(7).items()})
    xa = torch.randn(1, 4, 210, 135, generator=g)
    with torch.no_grad():
        o_ref, z_ref = ae(xa)
        o_o, z_o = O.ae_forward(ae.state_dict(), xa)
    report['AE.out'] = _rel(o_o, o_ref)
    report['AE.z'] = _rel(z_o, z_ref)
    ae_sd = {k: v.clone() for k, v in ae.state_dict().items()}

    # ---- utils/utils.py 6-D helpers through the tgm restatement ---------------------------
    tgm = types.ModuleType('torchgeometry')
    tgm.rotation_matrix_to_angle_axis = O.rotation_matrix_to_angle_axis
    tgm.angle_axis_to_rotation_matrix = O.angle_axis_to_rotation_matrix
    sys.modules['torchgeometry'] = tgm
    import scipy.ndimage
    if 'scipy.ndimage.filters' not in sys.modules:
        try:
            import scipy.ndimage.filters                                  # noqa: F401
        except Exception:
            sys.modules['scipy.ndimage.filters'] = scipy.ndimage
    ref_utils = importlib.import_module('utils.utils')
    x75 = torch.randn(16, 75, generator=g)
    report['utils.convert_to_3D_rot'] = _rel(O.convert_to_3D_rot(x75), ref_utils.convert_to_3D_rot(x75))
    aa = torch.randn(16, 3, generator=g)
    report['utils.convert_to_6D_all'] = _rel(O.convert_to_6D_all(aa), ref_utils.convert_to_6D_all(aa))

    # ---- utils/utils.py marker-image encode / decode around the loop --------------
    from oracle import markers_oracle as MO
    body, contact = synthetic_marker_clip(seed=11)
    ref_img, ref_piv = ref_utils.get_local_markers_4chan(body.copy(), contact.copy())
    o_img, o_piv = MO.get_local_markers_4chan(body, contact)
    report['utils.get_local_markers_4chan'] = _rel(torch.from_numpy(o_img), torch.from_numpy(ref_img))
    report['utils.get_local_markers_4chan.rot_0_pivot'] = float(np.abs(o_piv - ref_piv).max())
    dec_in = decode_input_from_image(ref_img)
    ref_glob = ref_utils.reconstruct_global_body(dec_in.copy(), ref_piv)
    o_glob = MO.reconstruct_global_body(dec_in, ref_piv)
    report['utils.reconstruct_global_body'] = _rel(torch.from_numpy(o_glob), torch.from_numpy(ref_glob))
    np.savez_compressed(os.path.join(HERE, 'markers_decode.npz'), body=body.astype(np.float32), contact=contact.astype(np.float32),
                        image=ref_img.astype(np.float32), rot_0_pivot=np.asarray(ref_piv, np.float64),
                        decode_in=dec_in.astype(np.float32), global_body=ref_glob.astype(np.float32))

    # ---- VPoser.decode -------------------------------------------------------------------
    cfg = types.ModuleType('configer'); cfg.Configer = object
    sys.modules['configer'] = cfg
    smplx_stub = types.ModuleType('smplx'); smplx_stub.lbs = ref_lbs
    smplx_lbs = types.ModuleType('smplx.lbs'); smplx_lbs.lbs = ref_lbs.lbs
    sys.modules['smplx'] = smplx_stub; sys.modules['smplx.lbs'] = smplx_lbs
    try:
        vp_mod = importlib.import_module('human_body_prior.train.vposer_smpl')
        vp = vp_mod.VPoser(num_neurons=512, latentD=32, data_shape=[1, 21, 3]).eval()
        w = O.make_vposer_weights(seed=2)
        vp.load_state_dict({**vp.state_dict(), **w})
        Z = torch.randn(8, 32, generator=g) * 0.7
        with torch.no_grad():
            report['VPoser.decode.aa'] = _rel(O.vposer_decode(w, Z, 'aa'), vp.decode(Z, 'aa'))
            report['VPoser.decode.matrot'] = _rel(O.vposer_decode(w, Z, 'matrot'), vp.decode(Z, 'matrot'))
    except Exception as e:                                                # pragma: no cover
        report['VPoser.import_error'] = repr(e)
    return report, ae_sd


def emit_golden(ae_sd):
    g = torch.Generator().manual_seed(11)
    out = {}
    # (5) 6-D -> aa, random + near-singular -------------------------------------------------
    x6 = torch.randn(64, 6, generator=g)
    aa_small = torch.randn(8, 3, generator=g) * 1e-4
    axis = torch.nn.functional.normalize(torch.randn(8, 3, generator=g), dim=1)
    aa_pi = axis * (np.pi - 1e-3)
    x6 = torch.cat([x6, O.convert_to_6D_all(aa_small), O.convert_to_6D_all(aa_pi)], 0)
    out['rot6d_in'] = x6.numpy()
    out['rot6d_aa'] = O.convert_to_3D_all(x6).numpy()
    np.savez(os.path.join(HERE, 'rot6d.npz'), **out)

    # (4) VPoser decode ---------------------------------------------------------------------
    w = O.make_vposer_weights(seed=2)
    Z = torch.randn(16, 32, generator=g) * 0.7
    np.savez(os.path.join(HERE, 'vposer_decode.npz'), Z=Z.numpy(),
             aa=O.vposer_decode(w, Z, 'aa').numpy(), matrot=O.vposer_decode(w, Z, 'matrot').numpy())

    # (1) LBS on a small synthetic model (full tensors) -------------------------------------
    m = small_model()
    so = O.SmplxOracle(m)
    B = 4
    p = dict(betas=torch.randn(B, 10, generator=g) * 0.5, global_orient=torch.randn(B, 3, generator=g),
             body_pose=torch.randn(B, 63, generator=g) * 0.3, lh=torch.randn(B, 12, generator=g) * 0.1,
             rh=torch.randn(B, 12, generator=g) * 0.1, transl=torch.randn(B, 3, generator=g))
    for k in ('global_orient', 'body_pose', 'transl', 'lh', 'rh', 'betas'):
        p[k].requires_grad_(True)
    ex_ids = [5, 17, 33, 100, 200, 300, 400, 500, 600, 610, 620, 7, 9, 11, 13, 15, 19, 21, 23, 25, 27]
    so.extra_ids = torch.tensor(ex_ids)
    verts, joints, fp = so.forward(p['betas'], p['global_orient'], p['body_pose'], p['lh'], p['rh'], p['transl'])
    wv = torch.randn(verts.shape, generator=g)
    wj = torch.randn(joints.shape, generator=g)
    ((verts * wv).sum() + (joints * wj).sum()).backward()
    np.savez(os.path.join(HERE, 'lbs_small.npz'), model_seed=3, model_V=640, extra_ids=np.asarray(ex_ids),
             **{k: v.detach().numpy() for k, v in p.items()},
             verts=verts.detach().numpy(), joints=joints.detach().numpy(), full_pose=fp.detach().numpy(),
             wv=wv.numpy(), wj=wj.numpy(), **{'g_' + k: v.grad.numpy() for k, v in p.items()})

    # (2) Enc with real weights -------------------------------------------------------------
    enc_w = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(ROOT, 'lemo_amd/assets/smooth_enc_15217.npz')).items()}
    x = (torch.randn(1, 1, 245, 134, generator=g) * 0.05).requires_grad_(True)
    z = O.enc_forward(enc_w, x)
    ls = torch.mean((z[..., 1:] - z[..., :-1]) ** 2)
    ls.backward()
    np.savez(os.path.join(HERE, 'enc_smooth.npz'), x=x.detach().numpy(), loss_smooth=float(ls),
             z_sum=float(z.double().sum()), z_abs_sum=float(z.double().abs().sum()),
             z_sub=z.detach()[0, ::8, ::16, ::16].numpy(), gx=x.grad.numpy())

    # (3) AE with numpy-seeded weights (lemo_amd.synthetic.make_ae_weights(7); runs/59547 is absent)
    xa = torch.randn(1, 4, 210, 135, generator=g).half().float()
    with torch.no_grad():
        oa, za = O.ae_forward(ae_sd, xa)
    np.savez_compressed(os.path.join(HERE, 'ae_infill.npz'), x=xa.numpy().astype(np.float16),
                        out_sum=float(oa.double().sum()), z_sum=float(za.double().sum()),
                        out_sub=oa[0, 0, ::7, ::5].numpy(), z=za.numpy())


def emit_amass_iteration():
    """(6) one full AMASS iteration on the full-size synthetic model + 10 Adam steps."""
    from lemo_amd.assets import load_assets
    A = load_assets()
    m = synthetic.make_synthetic_smplx(seed=0)
    so = O.SmplxOracle(m)
    vw = O.make_vposer_weights(seed=2)
    seq = synthetic.make_synthetic_sequence(0, B=119)
    with torch.no_grad():
        tp = torch.from_numpy(seq['target_params'])
        bp = O.vposer_decode(vw, tp[:, 16:48], 'aa').view(119, -1)
        tv, _, _ = so.forward(tp[:, 6:16], tp[:, 3:6], bp, tp[:, 48:60], tp[:, 60:], tp[:, 0:3])
        markers_rec = tv[:, torch.from_numpy(A['ids']['markers67']).long()].numpy()
    fit = O.AmassFitOracle(so, vw, A['enc_w_torch'], A['ids'], A['Xmean'], A['Xstd'],
                           seq['init_params'], markers_rec, seq['contact_lbl'], faithful=False)
    total, parts, p72, verts = fit.losses()
    total.backward()
    out = dict(markers_rec=markers_rec, p75_0=fit.params75().numpy(), p72_0=p72.detach().numpy(),
               total=float(total), **{'loss_' + k: float(v) for k, v in parts.items()},
               g_transl=fit.transl.grad.numpy().copy(), g_rot6d=fit.rot6d.grad.numpy().copy(),
               g_other=fit.other.grad.numpy().copy(),
               verts_sub=verts.detach()[:, ::97].numpy(), verts_sum=float(verts.double().sum()))
    fit.opt.zero_grad()
    hist = []
    for i in range(10):
        hist.append(fit.step())
        if i == 0:
            out['p75_after1'] = fit.params75().numpy()
    out['p75_after10'] = fit.params75().numpy()
    out['total_hist'] = np.asarray([h['total'] for h in hist])
    out['smooth_hist'] = np.asarray([h['smooth'] for h in hist])
    np.savez(os.path.join(HERE, 'amass_iter.npz'), **out)
    print('amass iteration:', {k: (float(v) if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items()})


def _flat_clip_image(seed=11):
    """normalised [1,4,208,119] clip image of the synthetic marker clip (what the AMASS loader hands over,
    loader/optimize_loader_amass_new.py:371-377) + its rot_0_pivot"""
    from oracle import markers_oracle as MO
    stats = dict(np.load(os.path.join(ROOT, 'lemo_amd/assets/stats_infill.npz')))
    body, contact = synthetic_marker_clip(seed=seed)
    img, piv = MO.get_local_markers_4chan(body, contact)
    clip = torch.from_numpy(img).float().unsqueeze(0)
    f = lambda k: torch.from_numpy(np.asarray(stats[k])).float()
    clip[:, 0] = (clip[:, 0] - f('Xmean_local')) / f('Xstd_local')
    clip[:, 1:3] = (clip[:, 1:3] - f('Xmean_global_xy')) / f('Xstd_global_xy')
    clip[:, 3] = (clip[:, 3] - f('Xmean_global_r')) / f('Xstd_global_r')
    return clip.permute(0, 1, 3, 2).contiguous(), piv, stats


def pin_amass_loop_and_clip(report):
    """Run the reference's own TEXT (opt_amass_temp.py:355-453 loop body, :159-214 mask + finetune, :256-329 decode)
    and compare with the oracle's restatements.  Emits tests/golden/amass_clip.npz (fixture of the per-clip setup)."""
    import ref_harness as RH
    from oracle import pipeline_oracle as PO
    from lemo_amd.assets import load_assets
    A = load_assets()
    # ---- loop body at the golden-(6) inputs: B = 119, V = 10475, real markers / encoder weights -------------
    m = synthetic.make_synthetic_smplx(seed=0)
    so = O.SmplxOracle(m)
    vw = O.make_vposer_weights(seed=2)
    seq = synthetic.make_synthetic_sequence(0, B=119)
    gold = np.load(os.path.join(HERE, 'amass_iter.npz'))
    r = RH.run_amass_loop_body(so, vw, A['ids'], A['Xmean'], A['Xstd'], seq['init_params'], gold['markers_rec'],
                               seq['contact_lbl'], steps=2)
    relf = lambda a, b: abs(a - b) / max(abs(b), 1e-30)
    for k in ('marker', 'vposer', 'shape', 'hand', 'contact', 'smooth'):
        report['amass_loop.' + k] = relf(float(gold['loss_' + k]), r[k])
    report['amass_loop.total'] = relf(float(gold['total']), r['total'])
    for k in ('g_transl', 'g_rot6d', 'g_other'):
        report['amass_loop.' + k] = _rel(torch.from_numpy(gold[k]), torch.from_numpy(r[k]))
    report['amass_loop.p75_after1'] = _rel(torch.from_numpy(gold['p75_after1']), torch.from_numpy(r['p75_hist'][0]))
    # ---- per-clip setup: mask + 60-step finetune + eval forward, then decode -------------------------------
    clip, piv, stats = _flat_clip_image()
    ae_w = {k: torch.from_numpy(v) for k, v in synthetic.make_ae_weights(7).items()}
    xin_r, rec_r, rows, loss_r = RH.run_amass_finetune_text(ae_w, clip, finetune_steps=60)
    xin_o, mask = PO.amass_mask_input(clip)
    _, rec_o = PO.finetune(ae_w, xin_o, mask, steps=60)
    report['amass_clip.masked_input'] = _rel(xin_o, xin_r)
    report['amass_clip.train_rows'] = 0.0 if sorted(rows[:-5]) == torch.nonzero(mask[:, 0]).flatten().tolist() else 1.0
    report['amass_clip.finetuned_rec'] = _rel(rec_o, rec_r)
    lbl_r, mk_r = RH.run_amass_decode_text(rec_r, clip, piv)
    lbl_o, mk_o = PO.decode_markers(rec_o[0, 0], clip[0], piv, stats)
    report['amass_clip.contact_lbl_rec'] = float((lbl_r - lbl_o).abs().max())
    report['amass_clip.markers_rec'] = _rel(torch.from_numpy(mk_o).float(), mk_r)
    # the decode statements on an image with mixed contact logits (a seeded-random AE emits one sign only)
    g = torch.Generator().manual_seed(5)
    rnd = rec_r + torch.randn(rec_r.shape, generator=g) * 0.5
    lbl_r2, mk_r2 = RH.run_amass_decode_text(rnd, clip, piv)
    lbl_o2, mk_o2 = PO.decode_markers(rnd[0, 0], clip[0], piv, stats)
    report['amass_clip.decode_mixed.contact_lbl_rec'] = float((lbl_r2 - lbl_o2).abs().max())
    report['amass_clip.decode_mixed.markers_rec'] = _rel(torch.from_numpy(mk_o2).float(), mk_r2)
    assert 0 < float(lbl_r2.sum()) < lbl_r2.numel()
    np.savez_compressed(os.path.join(HERE, 'amass_clip.npz'), clip_img=clip.numpy(), rot_0_pivot=np.asarray(piv, np.float64),
                        clip_img_input=xin_r.numpy(), train_mask=mask.numpy(), clip_img_rec=rec_r.numpy(),
                        finetune_last_loss=np.float64(loss_r), contact_lbl_rec=lbl_r.numpy(), markers_rec=mk_r.numpy(),
                        rec_mixed=rnd.numpy(), contact_lbl_mixed=lbl_r2.numpy(), markers_mixed=mk_r2.numpy())


def pin_dropin(report):
    """The drop-in claim: the reference's loop-body TEXT (opt_amass_temp.py:355-453 + backward + Adam step) is
    exec'd against the PRODUCT modules -- lemo_amd.compat smplx.create, lemo_amd.vposer.VPoser, lemo_amd.priors.Enc,
    lemo_amd.rotation.convert_to_3D_rot, running the unmodified kernel sources on the host emulator -- and against the
    oracle-backed objects; both runs start from the same seeded small problem.  Rows ``dropin.*`` = max rel error of the
    product run vs the reference run; the reference run's outputs are committed as dropin_amass_small.npz."""
    import __graft_entry__ as ge
    import ref_harness as RH
    from lemo_amd import _hip
    import subprocess
    subprocess.run(['make', '-C', os.path.join(ROOT, 'lemo_amd', 'csrc'), '-j8', 'emu'], check=True, capture_output=True)
    emu = _hip.HipLib(_hip.EMU_LIB_PATH, is_emu=True)
    prob = ge.small_problem()
    ofit, markers = ge.oracle_for(prob, faithful=True)
    ref = RH.run_amass_loop_body(ofit.smplx, ofit.vposer_w, prob['ids'], prob['Xmean'].reshape(1, 1, -1), prob['Xstd'],
                                 prob['seq']['init_params'], markers, prob['seq']['contact_lbl'], steps=2)
    got = RH.run_amass_loop_body_on_product(prob, markers, emu, steps=2)
    for k in ('marker', 'vposer', 'shape', 'hand', 'contact', 'smooth', 'total'):
        report['dropin.' + k] = abs(got[k] - ref[k]) / max(abs(ref[k]), 1e-30)
    for k in ('g_transl', 'g_rot6d', 'g_other'):
        report['dropin.' + k] = float(np.abs(got[k] - ref[k]).max() / np.abs(ref[k]).max())
    report['dropin.p75_after3'] = float(np.abs(got['p75_hist'][2] - ref['p75_hist'][2]).max())
    np.savez(os.path.join(HERE, 'dropin_amass_small.npz'), markers_rec=markers,
             **{k: np.asarray(ref[k]) for k in ('marker', 'vposer', 'shape', 'hand', 'contact', 'smooth', 'total', 'g_transl', 'g_rot6d', 'g_other')},
             p75_after1=ref['p75_hist'][0], p75_after3=ref['p75_hist'][2])


def pin_perframe(report):
    """BASELINE configs[0]: exec the reference's per-frame loop text (opt_amass_perframe.py:291-364) -- 3 frames, the
    full 100 steps each (so both lr switches fire) -- against oracle/pipeline_oracle.perframe_fit."""
    import __graft_entry__ as ge
    import ref_harness as RH
    from oracle import pipeline_oracle as PO
    prob = ge.small_problem()
    so = O.SmplxOracle(prob['model'], extra_joint_ids=list(range(21)))
    vw = {k: torch.from_numpy(v) for k, v in prob['vposer_w'].items()}
    _, markers = ge.oracle_for(prob)
    betas = prob['seq']['init_params'][0, 6:16]
    r = RH.run_perframe_text(so, vw, prob['ids']['markers67'], markers[:3], betas, steps=100)
    o, last = PO.perframe_fit(so, vw, prob['ids']['markers67'], markers[:3], betas, steps=100)
    report['perframe.body_params_opt_cur_clip'] = _rel(torch.from_numpy(o), torch.from_numpy(np.asarray(r)))
    np.savez(os.path.join(HERE, 'perframe_fit.npz'), markers_rec=markers[:3], betas=betas, p72=np.asarray(r), final_loss=last)


def pin_dropin_prox(report):
    """Drop-in proof of the PROX side of the boundary: the reference's own
    ``FittingMonitor.create_fitting_closure`` + ``SMPLifyLoss.forward`` + ``optim_factory.create_optimizer``
    (fitting_temp_slide.py:220-311, 564-1062, optimizers/optim_factory.py) run UNMODIFIED on top of the product's modules --
    ``lemo_amd.compat`` smplx.create, ``lemo_amd.vposer.VPoser``, ``lemo_amd.priors.Enc`` on the host-emulated kernel
    library -- for S2 / S3 x first / later window; compared with prox_iter.npz, which holds what the same reference code
    produced on its own stack (pin_prox_and_emit).  Rows ``dropin_prox.*``: 14 loss_dict entries (max rel), the three
    gradients (max-norm rel) and the parameters after three ``optimizer.step(closure)`` (max abs)."""
    import __graft_entry__ as ge
    import ref_harness as RH
    from lemo_amd import _hip
    from oracle.prox_oracle import LOSS_KEYS
    emu = _hip.HipLib(_hip.EMU_LIB_PATH, is_emu=True)
    g = np.load(os.path.join(HERE, 'prox_iter.npz'))
    for stage in ('S3', 'S2'):
        for first in (False, True):
            tag = f'{stage}_{"first" if first else "later"}'
            prob = ge.prox_small_problem(stage=stage, real_markers=True)
            rw = RH.RefProxWindow(prob, first_batch_flag=first, product_lib=emu)
            h = rw.iterate(1)[0]
            ref = g[tag + '_loss']
            report[f'dropin_prox.{tag}.loss_dict'] = max((abs(h[k] - float(r)) / max(abs(float(r)), 1e-30)) if h[k] != float(r) else 0.0
                                                         for k, r in zip(LOSS_KEYS, ref))
            gr = rw.grads()
            for k in ('pose_embedding', 'transl', 'global_orient'):
                r = g[f'{tag}_g_{k}']
                report[f'dropin_prox.{tag}.g_{k}'] = float(np.abs(gr[k] - r).max() / np.abs(r).max())
            rw.iterate(2)
            worst = 0.0
            for k, t in (('pose_embedding', rw.pose_embedding), ('transl', rw.body_model.transl), ('global_orient', rw.body_model.global_orient)):
                worst = max(worst, float(np.abs(t.detach().numpy() - g[f'{tag}_{k}_after3']).max()))
            report[f'dropin_prox.{tag}.params_after3'] = worst


def pin_prox_and_emit(report):
    """(7) Drive the reference's OWN SMPLifyLoss / FittingMonitor closure / PerspectiveCamera / L2Prior /
    SMPLifyAnglePrior / JointMapper / optim_factory (imported under module stubs, tests/golden/ref_harness.py) on the
    seeded window ``__graft_entry__.prox_small_problem(real_markers=True)`` for S2 / S3 x first / later window,
    compare the 14 loss_dict entries, three gradients and the parameters after 3 Adam steps with
    oracle/prox_oracle.py, and write tests/golden/prox_iter.npz FROM THE REFERENCE RUN.  Also pins the
    ``opt_step == 0`` block (:776-941) against oracle/pipeline_oracle.py and writes prox_setup.npz."""
    import __graft_entry__ as ge
    import ref_harness as RH
    from oracle import pipeline_oracle as PO
    from oracle.prox_oracle import LOSS_KEYS, PARAM_NAMES
    out = {}
    relf = lambda a, b: (abs(a - b) / max(abs(b), 1e-30)) if a != b else 0.0
    for stage in ('S3', 'S2'):
        for first in (False, True):
            prob = ge.prox_small_problem(stage=stage, real_markers=True)
            of = ge.prox_oracle_for(prob, first_batch_flag=first)
            rw = RH.RefProxWindow(prob, first_batch_flag=first)
            tag = f'{stage}_{"first" if first else "later"}'
            h = rw.iterate(1)[0]
            go = of.closure()
            out[tag + '_loss'] = np.asarray([h[k] for k in LOSS_KEYS])
            report[f'prox.{tag}.loss_dict'] = max(relf(float(go[k]), h[k]) for k in LOSS_KEYS)
            g = rw.grads()
            for k, og in (('pose_embedding', of.pose_embedding.grad), ('transl', of.p['transl'].grad),
                          ('global_orient', of.p['global_orient'].grad)):
                out[f'{tag}_g_{k}'] = g[k]
                report[f'prox.{tag}.g_{k}'] = _rel(og, torch.from_numpy(g[k]))
            of.opt.step()
            for _ in range(2):
                of.step()
            rw.iterate(2)
            worst = max(float((getattr(rw.body_model, k).detach() - of.p[k].detach()).abs().max()) for k in PARAM_NAMES)
            worst = max(worst, float((rw.pose_embedding.detach() - of.pose_embedding.detach()).abs().max()))
            report[f'prox.{tag}.params_after3'] = worst
            out[tag + '_pose_embedding_after3'] = rw.pose_embedding.detach().numpy().copy()
            out[tag + '_transl_after3'] = rw.body_model.transl.detach().numpy().copy()
            out[tag + '_global_orient_after3'] = rw.body_model.global_orient.detach().numpy().copy()
    np.savez(os.path.join(HERE, 'prox_iter.npz'), **out)
    # ---- opt_step == 0: per-window infill setup ------------------------------------------------------------
    stats = dict(np.load(os.path.join(ROOT, 'lemo_amd/assets/stats_infill.npz')))
    ae_w = {k: torch.from_numpy(v) for k, v in synthetic.make_ae_weights(7).items()}
    prob = ge.prox_small_problem(stage='S3', real_markers=True)
    mask = prob['infill']['marker_mask']
    prob['infill'] = dict(marker_mask=mask)
    rw = RH.RefProxWindow(prob, first_batch_flag=False, ae_weights=ae_w)
    rw.closure()
    of = ge.prox_oracle_for(dict(prob, infill={}), first_batch_flag=False)
    with torch.no_grad():
        verts, _, _ = of._body(True)
        _, sj, _ = of._body(False)
        vw = torch.matmul(of.R, verts.permute(0, 2, 1)).permute(0, 2, 1) + of.t
        jw = torch.matmul(of.R, sj.permute(0, 2, 1)).permute(0, 2, 1) + of.t
    s = PO.prox_window_setup(vw, jw, torch.from_numpy(mask), ae_w, stats, prob['ids']['markers67'])
    report['prox_setup.body_markers_rec'] = _rel(s['body_markers_rec'], rw.loss.body_markers_rec)
    report['prox_setup.contact_lbl_rec'] = float((s['contact_lbl_rec'] - rw.loss.contact_lbl_rec).abs().max())
    report['prox_setup.motion_infill_loss'] = 0.0 if np.isfinite(float(rw.loss_dict['motion_infill_loss'])) else 1.0
    np.savez_compressed(os.path.join(HERE, 'prox_setup.npz'), vertices_world=vw.numpy(), smplx_joints_world=jw.numpy(),
                        marker_mask=mask, body_markers_rec=rw.loss.body_markers_rec.numpy(),
                        contact_lbl_rec=rw.loss.contact_lbl_rec.numpy(), clip_img_input=s['clip_img_input'].numpy(),
                        train_mask=s['train_mask'].numpy(), clip_img_rec=s['clip_img_rec'].numpy(),
                        motion_infill_loss=np.float64(float(rw.loss_dict['motion_infill_loss'])))
    print('prox fixtures:', {k: v.shape for k, v in out.items()})
    # ---- result wire format: two per-frame pickles written by the reference's own lines :577-594, and the
    # product's writer read back by the reference's own reader (data_parser_slide.py:106-126)
    import pickle
    import tempfile
    from lemo_amd import prox_windows as PW
    prob = ge.prox_small_problem(stage='S2', real_markers=True)
    rw = RH.RefProxWindow(prob, first_batch_flag=True)
    rw.iterate(1)
    with tempfile.TemporaryDirectory() as tmp:
        paths = [os.path.join(tmp, f'{i:03d}.pkl') for i in range(prob['B'])]
        RH.write_reference_result_pkls(rw, paths)
        for i in (0, 5):
            with open(paths[i], 'rb') as f:
                raw = f.read()
            with open(os.path.join(HERE, f'prox_result_ref_frame{i}.pkl'), 'wb') as f:
                f.write(raw)
        ref5 = pickle.loads(raw)
        cam = {k[len('camera_'):]: np.repeat(v, prob['B'], 0) for k, v in ref5.items() if k.startswith('camera_')}
        body = {k: np.repeat(v, prob['B'], 0) for k, v in ref5.items() if not k.startswith('camera_') and k not in ('pose_embedding', 'body_pose')}
        mine = os.path.join(tmp, 'mine.pkl')
        PW.write_result_pkl(mine, cam, body, np.repeat(ref5['pose_embedding'], prob['B'], 0), np.repeat(ref5['body_pose'], prob['B'], 0), 3)
        back = RH.reference_read_prox_pkl(mine)
        theirs = RH.reference_read_prox_pkl(paths[5])
        report['prox_pkl.reference_reader_on_product_file'] = max(float(np.abs(back[k] - theirs[k]).max()) for k in theirs)
        report['prox_pkl.keys'] = 0.0 if set(pickle.load(open(mine, 'rb'))) == set(ref5) else 1.0


if __name__ == '__main__':
    rep, ae_sd = check_against_reference()
    print('oracle vs reference (max rel err):')
    bad = False
    for k, v in rep.items():
        print(f'  {k:28s} {v}')
        if isinstance(v, float) and v > 2e-6:
            bad = True
    with open(os.path.join(HERE, 'oracle_vs_reference.txt'), 'w') as f:
        for k, v in rep.items():
            f.write(f'{k}\t{v}\n')
    assert not bad, 'oracle disagrees with the reference'
    emit_golden(ae_sd)
    emit_amass_iteration()
    pins = {}
    pin_amass_loop_and_clip(pins)
    pin_perframe(pins)
    pin_prox_and_emit(pins)
    print('loop bodies and pipelines vs the text and classes of the reference itself (max rel err):')
    with open(os.path.join(HERE, 'oracle_vs_reference.txt'), 'a') as f:
        for k, v in pins.items():
            print(f'  {k:44s} {v}')
            f.write(f'{k}\t{v}\n')
            bad = bad or not (v <= 2e-6)
    assert not bad, 'oracle disagrees with the reference'
    drop = {}
    pin_dropin(drop)
    pin_dropin_prox(drop)
    print('reference loop text on the PRODUCT modules (emulator library) vs on the oracle-backed objects:')
    with open(os.path.join(HERE, 'dropin_vs_reference.txt'), 'w') as f:
        for k, v in drop.items():
            print(f'  {k:44s} {v}')
            f.write(f'{k}\t{v}\n')
    assert max(v for k, v in drop.items() if k.startswith('dropin.') and not k.startswith('dropin.g_') and k != 'dropin.p75_after3') <= 1e-5, 'loss scalars'
    assert max(v for k, v in drop.items() if k.startswith('dropin_prox.') and k.endswith('loss_dict')) <= 1e-5, 'PROX drop-in: loss_dict'
    assert max(v for k, v in drop.items() if k.startswith('dropin_prox.') and '.g_' in k) <= 2e-4, 'PROX drop-in: gradients'
    assert max(v for k, v in drop.items() if k.startswith('dropin_prox.') and k.endswith('params_after3')) <= 1e-4, 'PROX drop-in: parameters'
    assert max(v for k, v in drop.items() if k.startswith('dropin.g_')) <= 2e-4 and drop['dropin.p75_after3'] <= 1e-5
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
