"""Teacher-forcing fixtures: optimiser states the REFERENCE's own loops went through (build container only).

    python tests/golden/make_teacher.py [amass] [perframe] [prox]

The free-running trajectory gates of rounds 1-3 (10 steps <= 1e-2, 100 steps <= 2 mm MPJPE, stage 1 <= 10 x a chaos
yardstick) could not tell a defect from the chaos of a kinked objective under Adam.  Here the reference's loops are run AS
TEXT (tests/golden/ref_harness.py: ``opt_amass_temp.py:331-455``, ``opt_amass_perframe.py:291-364``, the PROX closure driven by
``optimizer.step(closure)`` over two CHAINED windows with the reference's own pickle writer / reader in between) with a
passive recorder in place of ``optim.Adam``; the full optimiser state (parameters, exp_avg, exp_avg_sq, step) is kept before
and after selected steps, together with the gradients and losses of that step.  The ``-m gpu`` tests load state k into the
engine (``lemo_fit_load_state`` / ``lemo_prox_load_state``), run ONE step and compare with state k + 1 -- every step of the loop
is then a one-step parity statement, lr switches included.

A float64 pass over the same states (oracle/f64.py) turns "how far may a correct fp32 implementation be from the reference's
fp32" into stored numbers: g64 (the exact gradient at that state) and S (computed kink exposure of each frame, AMASS).
Only numbers leave this file (``teacher_*.npz``)."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import lemo_oracle as O                      # noqa: E402
from lemo_amd import synthetic                           # noqa: E402

torch.set_num_threads(8)

AMASS_STEPS = (0, 1, 10, 30, 60, 61, 62, 99)
AMASS_EXTRA_STATES = (15, 67)      # states only (round 6): where five graph-replayed steps from states 10 / 62 must land (test_gpu_teacher.py)
"""recorded iterations of the 100-step loop: the start, the first update with history, mid-run, both sides of the lr switch
(``if step > 60: lr = 0.005``, opt_amass_temp.py:350-352: iteration 60 is the last at 0.01, 61 the first at 0.005) and the
last one"""
PERFRAME_STEPS = (0, 1, 59, 60, 61, 79, 80, 81, 99)
"""both lr switches of opt_amass_perframe.py:316-321 (``step > 60`` -> 0.01, ``step > 80`` -> 0.003)"""
PROX_STEPS = (0, 1, 2, 30, 59)


def _cat(parts):
    return np.concatenate([np.asarray(p, np.float32).reshape(p.shape[0], -1) for p in parts], axis=1)


def _state_arrays(prefix, st):
    return {f'{prefix}_p': _cat(st['params']), f'{prefix}_m': _cat(st['exp_avg']), f'{prefix}_v': _cat(st['exp_avg_sq']),
            f'{prefix}_step': np.int32(st['step'])}


# ----------------------------------------------------------------------------------------------------------------------
def amass():
    """BASELINE configs[1] at full size (B = 119, V = 10475, real markers / encoder weights): the golden-(6) inputs."""
    import ref_harness as RH
    from oracle.f64 import amass_fit_oracle_f64, default_f64, flip_sensitivity, KinkProbe
    from lemo_amd.assets import load_assets
    A = load_assets()
    m = synthetic.make_synthetic_smplx(seed=0)
    so = O.SmplxOracle(m)
    vw = O.make_vposer_weights(seed=2)
    seq = synthetic.make_synthetic_sequence(0, B=119)
    gold = np.load(os.path.join(HERE, 'amass_iter.npz'))
    t0 = time.time()
    recs, p72 = RH.run_amass_loop_text(so, vw, A['ids'], A['Xmean'], A['Xstd'], seq['init_params'], gold['markers_rec'],
                                       seq['contact_lbl'], tuple(sorted(AMASS_STEPS + AMASS_EXTRA_STATES)))
    print(f'reference loop text: 100 steps in {time.time() - t0:.0f} s')
    # consistency with the committed one-iteration fixture (written by the oracle, pinned to the same text at 0.0)
    assert np.array_equal(recs[0]['before']['params'][0], seq['init_params'][:, 0:3])
    for i, k in enumerate(('g_transl', 'g_rot6d', 'g_other')):
        assert np.abs(recs[0]['grads'][i] - gold[k]).max() <= 1e-6 * np.abs(gold[k]).max(), k
    assert np.abs(_cat(recs[0]['after']['params']) - np.delete(gold['p75_after1'], np.s_[9:19], axis=1)).max() == 0.0
    out = dict(steps=np.asarray(AMASS_STEPS, np.int32), p72_final=p72)
    vwn = {k: v.numpy() for k, v in vw.items()}
    o64 = amass_fit_oracle_f64(m, vwn, A['enc_w'], A['ids'], A['Xmean'], A['Xstd'], seq['init_params'], gold['markers_rec'], seq['contact_lbl'])
    for k in AMASS_EXTRA_STATES:
        assert recs[k]['before']['step'] == k
        out.update(_state_arrays(f's{k}', recs[k]['before']))
    for k in AMASS_STEPS:
        r = recs[k]
        assert r['before']['step'] == k and r['after']['step'] == k + 1
        out.update(_state_arrays(f's{k}', r['before']))
        out.update(_state_arrays(f's{k + 1}', r['after']))
        out[f'g{k}'] = _cat(r['grads'])
        out[f'lr{k}'] = np.float64(r['lr'])
        out[f'loss{k}'] = np.asarray([r['extra'][n] for n in RH.AMASS_LOSS_VARS], np.float64)
        # float64 at the SAME fp32 state: exact gradient + computed kink exposure of every frame
        t0 = time.time()
        with default_f64(), torch.no_grad():
            for dst, src in zip((o64.transl, o64.rot6d, o64.other), r['before']['params']):
                dst.copy_(torch.from_numpy(src).double())
        with default_f64():
            tot = o64.losses()[0]
            g64 = torch.autograd.grad(tot, (o64.transl, o64.rot6d, o64.other))
        S = flip_sensitivity(o64, subsets=1)
        # the two other kink families (tests/test_gpu_parity.py::kink_exposure): frames holding an L1 residual within 3e-6 m of
        # zero, and whether any labelled contact speed sits within 2e-4 m/s of the 0.1 m/s threshold
        with default_f64(), KinkProbe(o64, tol_act=0.0) as pr:
            o64.losses()
        l1 = np.zeros(119, bool)
        l1[sorted({f for e_, kind, _, fr in pr.events if kind == 'l1' for f in fr})] = True
        out[f'l1_{k}'] = l1
        out[f'contact_{k}'] = np.bool_(any(kind.startswith('contact') for _, kind, _, _ in pr.events))
        out[f'g64_{k}'] = np.concatenate([g.numpy() for g in g64], axis=1)
        out[f'S{k}'] = np.stack([S['transl'].numpy(), S['rot6d'].numpy(), S['other'].numpy()])
        out[f'loss64_{k}'] = np.float64(float(tot))
        e = np.abs(out[f'g{k}'] - out[f'g64_{k}']).max() / np.abs(out[f'g64_{k}']).max()
        print(f'  step {k}: lr {r["lr"]}, total {r["extra"]["loss"]:.6f} (f64 {float(tot):.6f}), reference-f32 gradient vs f64 {e:.1e}, '
              f'S median {float(np.median(out[f"S{k}"])):.1e} max {float(out[f"S{k}"].max()):.1e}, L1-kink frames {int(l1.sum())}, '
              f'contact kink {bool(out[f"contact_{k}"])}  [{time.time() - t0:.0f} s]')
    np.savez_compressed(os.path.join(HERE, 'teacher_amass.npz'), **out)
    print('teacher_amass.npz', os.path.getsize(os.path.join(HERE, 'teacher_amass.npz')))


# ----------------------------------------------------------------------------------------------------------------------
def perframe():
    """BASELINE configs[0] at full model size: two frames (lr 0.1 for the first, 0.01 for the warm-started second)."""
    import ref_harness as RH
    from oracle import pipeline_oracle as PO
    from oracle.f64 import default_f64, _to_double
    from lemo_amd.assets import load_assets
    A = load_assets()
    m = synthetic.make_synthetic_smplx(seed=0)
    so = O.SmplxOracle(m)
    vw = O.make_vposer_weights(seed=2)
    seq = synthetic.make_synthetic_sequence(0, B=119)
    gold = np.load(os.path.join(HERE, 'amass_iter.npz'))
    markers, betas = gold['markers_rec'][:2], seq['init_params'][0, 6:16]
    at = [(f, k) for f in (0, 1) for k in PERFRAME_STEPS]
    p72, recs = RH.run_perframe_text(so, vw, A['ids']['markers67'], markers, betas, steps=100, record_at=at)
    out = dict(steps=np.asarray(PERFRAME_STEPS, np.int32), markers_rec=markers, betas=betas, p72=np.asarray(p72))
    with default_f64():
        so64 = O.SmplxOracle(m)
        _to_double(so64)
        vw64 = {k: v.double() for k, v in vw.items()}
        ids = torch.as_tensor(np.asarray(A['ids']['markers67'], np.int64))
    for (f, k), r in sorted(recs.items()):
        out.update(_state_arrays(f'f{f}s{k}', r['before']))
        out.update(_state_arrays(f'f{f}s{k + 1}', r['after']))
        out[f'f{f}g{k}'] = _cat(r['grads'])
        out[f'f{f}lr{k}'] = np.float64(r['lr'])
        out[f'f{f}loss{k}'] = np.asarray([r['extra'][n] for n in ('loss_marker', 'loss_vposer', 'loss_shape', 'loss_hand', 'loss')], np.float64)
        with default_f64():
            tr, r6, ot = (torch.from_numpy(a).double().requires_grad_(True) for a in r['before']['params'])
            loss, _, _, _ = PO.perframe_loss_terms(so64, vw64, ids, dict(O.LOSS_WEIGHTS), tr, r6, torch.from_numpy(betas).double().view(1, 10), ot,
                                                   torch.from_numpy(markers[f:f + 1]).double())
            g = torch.autograd.grad(loss, (tr, r6, ot))
        out[f'f{f}g64_{k}'] = np.concatenate([x.numpy() for x in g], axis=1)
        e = np.abs(out[f'f{f}g{k}'] - out[f'f{f}g64_{k}']).max() / np.abs(out[f'f{f}g64_{k}']).max()
        print(f'  frame {f} step {k}: lr {r["lr"]}, loss {r["extra"]["loss"]:.6f} (f64 {float(loss):.6f}), f32 gradient vs f64 {e:.1e}')
    np.savez_compressed(os.path.join(HERE, 'teacher_perframe.npz'), **out)
    print('teacher_perframe.npz', os.path.getsize(os.path.join(HERE, 'teacher_perframe.npz')))


# ----------------------------------------------------------------------------------------------------------------------
PROX_N, PROX_B = 23, 14            # windows (0, 14) and (9, 23); frozen prefix of the second window int(0.15 * 14) = 2 frames
# Round 5 (VERDICT r04 missing #2 / next #4): the same two chained windows at the BASELINE shape -- B = 100 frames per window (stride 70:
# windows (0, 100) and (70, 170)), V = 10475, the real marker / friction id tables, the S2 / S3 YAML weights; 64^3 SDF so that the
# reference's per-frame `repeat` of the grid stays at 105 MB.  Frozen prefix of the second window int(0.15 * 100) = 15 frames.
PROX_FULL_N, PROX_FULL_B, PROX_FULL_D = 170, 100, 64
PROX_FULL_STEPS = (0, 1, 30, 59)
PROX_SIZES = dict(small=dict(N=PROX_N, B=PROX_B, steps=PROX_STEPS, file='teacher_prox.npz'),
                  full=dict(N=PROX_FULL_N, B=PROX_FULL_B, steps=PROX_FULL_STEPS, file='teacher_prox_full.npz'))


def prox_recording(stage, size='small'):
    """the seeded recording both sides fit -- 23 frames of ``__graft_entry__.prox_small_problem(real_markers=True)`` or 170 frames of
    ``prox_full_problem(D=64)`` -- with betas that DIFFER from frame to frame so that the reference's per-window mean
    (fit_temp_loadprox_slide.py:497-498) is visible"""
    import __graft_entry__ as ge
    N = PROX_SIZES[size]['N']
    base = ge.prox_small_problem(B=N, stage=stage, real_markers=True) if size == 'small' else ge.prox_full_problem(stage, B=N, D=PROX_FULL_D)
    rng = np.random.default_rng(17)
    base['params'] = dict(base['params'], betas=(base['params']['betas'] + rng.standard_normal((N, 10)) * 0.05).astype(np.float32))
    return base


def prox_window_problem(base, s, e, params):
    """frames [s, e) of the recording as a window problem with start parameters ``params``"""
    prob = dict(base, B=e - s, params=params, gt_joints=base['gt_joints'][s:e], joints_conf=base['joints_conf'][s:e])
    if base['infill']:
        inf = base['infill']
        prob['infill'] = dict(marker_mask=inf['marker_mask'][s:e], body_markers_rec=inf['body_markers_rec'][s:e - 1],
                              contact_lbl_rec=inf['contact_lbl_rec'][s:e - 1])
    return prob


PROX_THRESHOLD_SHIFTS = (('fric_sdf', (-2e-6, 2e-6)), ('fric_vt', (-5e-7, 5e-7)), ('fric_vn', (-5e-7, 5e-7)), ('infill_res', (1e-6,)),
                         ('contact', (-2e-5, 2e-5)))
"""how far fp32 rounding can move each thresholded quantity of the PROX closure (world-frame vertices of a few metres: ulp 2.4e-7; the trilinear
SDF sample adds a few of them; frame differences two; the contact speed is 30 x a difference; the infill residual's threshold is 0, below which
the masked-out zeros sit -- shifted upwards only)"""


def prox_loss_jumps(of):
    """per loss_dict entry: how far it moves, in float64 at the oracle's current state, when each selection threshold of the closure is
    shifted by the rounding of the quantity it tests (a thresholded mean jumps by (x - mean) / n when one element changes sides) -- the
    slack a 1e-5 loss gate needs for exactly the entries that have such a threshold, computed per state instead of guessed"""
    from oracle.prox_oracle import LOSS_KEYS, THRESHOLDS
    from oracle.f64 import default_f64
    with default_f64(), torch.no_grad():
        base = {k: float(v) for k, v in of.loss_dict().items()}
        J = np.zeros(len(LOSS_KEYS))
        for name, shifts in PROX_THRESHOLD_SHIFTS:
            j = np.zeros(len(LOSS_KEYS))
            for d in shifts:
                of.thresholds = dict(THRESHOLDS, **{name: THRESHOLDS[name] + d})
                ld = of.loss_dict()
                j = np.maximum(j, [abs(float(ld[k]) - base[k]) for k in LOSS_KEYS])
            J += j
        of.thresholds = None
    return J


def prox(size='small'):
    """Two CHAINED windows through the reference's own objects: window 1 (first_batch_flag) -> the reference's pickle writer
    (fit_temp_loadprox_slide.py:577-594) -> the reference's reader (data_parser_slide.py:106-126) with its newest-result rule
    (:326-331) -> the reference's window initialisation (fit_temp_loadprox_slide.py:495-499: mean betas, reset_params) ->
    window 2 (gradient erase of the first 15 %)."""
    import ref_harness as RH
    from lemo_amd import prox_windows as PW
    from oracle.f64 import prox_fit_oracle_f64, default_f64, flip_sensitivity_of
    PROX_N, PROX_B, PROX_STEPS = (PROX_SIZES[size][k] for k in ('N', 'B', 'steps'))
    out = dict(steps=np.asarray(PROX_STEPS, np.int32), n_frames=np.int32(PROX_N), batch=np.int32(PROX_B))
    names = [f's001_frame_{i:05d}' for i in range(PROX_N)]
    wins = PW.sliding_windows(PROX_N, PROX_B)
    assert wins == ([(0, 14), (9, 23)] if size == 'small' else [(0, 100), (70, 170)])
    for stage in ('S2', 'S3'):
        base = prox_recording(stage, size)
        P0 = base['params']
        with tempfile.TemporaryDirectory() as tmp:
            cur, prox_dir = os.path.join(tmp, 'cur'), os.path.join(tmp, 'prox')
            # the per-frame PROX fits every window can fall back to (written in the reference's wire format by the product writer,
            # which tests/test_prox_windows.py pins to the reference's own files)
            body0 = {k: np.asarray(P0[k], np.float32) for k in ('transl', 'global_orient', 'betas', 'left_hand_pose', 'right_hand_pose',
                                                               'jaw_pose', 'leye_pose', 'reye_pose', 'expression')}
            for i, fn in enumerate(names):
                PW.write_result_pkl(PW.result_path(prox_dir, fn), {}, body0, np.asarray(P0['pose_embedding'], np.float32),
                                    np.zeros((PROX_N, 63), np.float32), i)
            for w, (s, e) in enumerate(wins):
                # data_parser_slide.py:326-331 per frame, through the reference's own reader; DataLoader collation = stacking
                rows = []
                for fn in names[s:e]:
                    p = PW.result_path(cur, fn)
                    rows.append(RH.reference_read_prox_pkl(p if os.path.exists(p) else PW.result_path(prox_dir, fn)))
                prox_params_dict = {k: torch.from_numpy(np.stack([r[k] for r in rows])) for k in rows[0]}
                # fit_temp_loadprox_slide.py:495-498 as text (mean betas over the window)
                ns = dict(np=np, prox_params_dict=prox_params_dict, gt_joints=torch.zeros(e - s, 118, 2))
                RH.exec_reference_lines(f'{RH.REF}/temp_prox/fit_temp_loadprox_slide.py', 495, 498, ns)
                start = {k: np.asarray(v, np.float32) for k, v in ns['prox_params_dict'].items()}
                prob = prox_window_problem(base, s, e, start)
                t0 = time.time()
                rw = RH.RefProxWindow(prob, first_batch_flag=(w == 0))
                rw.iterate(max(PROX_STEPS) + 1, record_at=set(PROX_STEPS))
                print(f'  {stage} window {w}: reference closure x {max(PROX_STEPS) + 1} in {time.time() - t0:.0f} s', flush=True)
                tag = f'{stage}_w{w}'
                out[f'{tag}_names'] = np.asarray(rw.param_names())
                out[f'{tag}_betas'] = start['betas']
                for k in PROX_STEPS:
                    r = rw.records[k]
                    out.update(_state_arrays(f'{tag}_s{k}', r['before']))
                    out.update(_state_arrays(f'{tag}_s{k + 1}', r['after']))
                    out[f'{tag}_g{k}'] = _cat(r['grads'])
                    out[f'{tag}_loss{k}'] = np.asarray([r['extra'][n] for n in RH_LOSS_KEYS()], np.float64)
                    # float64 closure at the SAME fp32 state (oracle/f64.py): the exact gradient, erase applied like the reference's
                    of = prox_fit_oracle_f64(prob, first_batch_flag=(w == 0))
                    with default_f64():
                        with torch.no_grad():
                            for n, a in zip(rw.param_names(), r['before']['params']):
                                (of.pose_embedding if n == 'pose_embedding' else of.p[n]).copy_(torch.from_numpy(a).double())
                        ld = of.closure()
                    out[f'{tag}_g64_{k}'] = np.concatenate([(of.pose_embedding if n == 'pose_embedding' else of.p[n]).grad.numpy()
                                                           for n in rw.param_names()], axis=1)
                    out[f'{tag}_loss64_{k}'] = np.float64(float(ld['total_loss'].detach()))
                    if size == 'full':
                        out[f'{tag}_lossjump{k}'] = prox_loss_jumps(of)
                    # computed exposure of every frame's gradient to the encoder's kinks (the smoothness prior carries weight 1e8 here)
                    plist = [of.pose_embedding if n == 'pose_embedding' else of.p[n] for n in rw.param_names()]
                    S = flip_sensitivity_of(lambda: of.loss_dict()['total_loss'], plist, subsets=1)
                    out[f'{tag}_S{k}'] = np.stack([x.numpy() for x in S])
                paths = [PW.result_path(cur, fn) for fn in names[s:e]]
                for p in paths:
                    os.makedirs(os.path.dirname(p), exist_ok=True)
                RH.write_reference_result_pkls(rw, paths)
                print(f'  {tag}: frames [{s}, {e}), total loss {rw.records[0]["extra"]["total_loss"]:.5f} -> {rw.records[max(PROX_STEPS)]["extra"]["total_loss"]:.5f}')
            # what the recording's pickles hold at the end (later window overwrote the overlap)
            final = [RH.reference_read_prox_pkl(PW.result_path(cur, fn)) for fn in names]
            for k in ('transl', 'global_orient', 'pose_embedding', 'betas'):
                out[f'{stage}_final_{k}'] = np.stack([r[k] for r in final])
    np.savez_compressed(os.path.join(HERE, PROX_SIZES[size]['file']), **out)
    print(PROX_SIZES[size]['file'], os.path.getsize(os.path.join(HERE, PROX_SIZES[size]['file'])))


def RH_LOSS_KEYS():
    from oracle.prox_oracle import LOSS_KEYS
    return LOSS_KEYS


if __name__ == '__main__':
    what = sys.argv[1:] or ['amass', 'perframe', 'prox']
    for w in what:
        print(f'== {w}')
        {'amass': amass, 'perframe': perframe, 'prox': prox, 'prox_full': lambda: prox('full')}[w]()
