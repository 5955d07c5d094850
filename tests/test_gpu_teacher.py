"""Teacher-forced one-step parity on the MI355X (-m gpu): every recorded step of the REFERENCE's own loops as a one-step
statement (tests/golden/teacher_*.npz, written by tests/golden/make_teacher.py from runs of the reference's loop text;
tests/teacher_common.py has the three statements per step).  Replaces the free-running trajectory gates of rounds 1-3
("10 steps < 1e-2", "10 x a chaos yardstick + 10 mm") and gives N3 -- two CHAINED PROX windows -- an oracle-parity
test on the device."""
import os

import numpy as np
import pytest
import torch

import teacher_common as TC
from conftest import GOLDEN
from lemo_amd import synthetic
from lemo_amd.assets import load_assets

pytestmark = pytest.mark.gpu
GROUPS = (('transl', 0, 3), ('rot6d', 3, 9), ('other', 9, 65))
REPORT = []
MULTI_STEP_GATE = {0: 0.5, 60: 1.2e-2, 10: 4.5e-2, 62: 2.5e-2}     # (10 -> 15, 62 -> 67, five steps each, round 6: 3 x the measured 1.5e-2 / 8.2e-3 lr on either kernel family)     # x lr: 3 x the measured 1.6e-1 (0 -> 2: noise-level entries, Adam divides by sqrt(v) ~ 0) and 3.7e-3 (60 -> 63)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs the MI355X'
    from lemo_amd import _hip
    assert not _hip.get_lib().is_emu
    return torch.device('cuda:0')


@pytest.fixture(scope='module', autouse=True)
def _print_report():
    yield
    print('\n'.join(['', 'teacher-forced parity (per step):'] + REPORT))


def _split(a, prefix=''):
    return {prefix + k: a[:, i:j] for k, i, j in GROUPS}


def _state(T, tag, k):
    return dict(_split(T[f'{tag}s{k}_p']), **_split(T[f'{tag}s{k}_m'], 'm_'), **_split(T[f'{tag}s{k}_v'], 'v_'), step=int(T[f'{tag}s{k}_step']))


def _cat_state(st):
    c = lambda pre: np.concatenate([st[pre + k].cpu().numpy() for k, _, _ in GROUPS], axis=1)
    return dict(p=c(''), m=c('m_'), v=c('v_'))


@pytest.mark.timeout(900)
@pytest.mark.parametrize('conv_variant', [9, 2])
def test_amass_loop_teacher_forced_full_size(dev, conv_variant):
    """BASELINE configs[1] (B = 119, V = 10475, all vertices forwarded): iterations 0, 1, 10, 30, 60, 61, 62, 99 of the
    reference's 100-step loop (opt_amass_temp.py:344-455; 60 -> 61 is its lr switch) from the reference's own optimiser state."""
    from lemo_amd.fitting import AmassTemporalFitter
    from lemo_amd.vposer import make_vposer_weights
    A = load_assets()
    T = np.load(os.path.join(GOLDEN, 'teacher_amass.npz'))
    gold = np.load(os.path.join(GOLDEN, 'amass_iter.npz'))
    seq = synthetic.make_synthetic_sequence(0, B=119)
    fit = AmassTemporalFitter(synthetic.make_synthetic_smplx(seed=0), make_vposer_weights(2), A['enc_w'], A['ids'], A['Xmean'], A['Xstd'],
                              119, dev, full_vertices=True, conv_variant=conv_variant)
    fit.load_sequence(seq['init_params'], gold['markers_rec'], seq['contact_lbl'])
    s = torch.cuda.Stream(dev)
    names = ('marker', 'vposer', 'shape', 'hand', 'contact', 'smooth', 'total')
    worst_med = 0.0
    # Round 5 ("the derived bounds are loose where it matters"): at every recorded state the engine's gradient is ALSO
    # compared with the float64 gradient of the piece of the objective the engine itself is on (its own LeakyReLU / L1 / contact decisions
    # pinned, tests/kink_attribution.py) -- per frame inside 2e-5 + 4 R[frame] (R: computed conditioning of the 6-D decode), median frame
    # <= 1.5 x the fp32 CPU path's.  A 1e-3 gradient defect confined to a high-exposure frame late in the fit has nowhere to hide there.
    cond = None
    if conv_variant == 9:
        import kink_attribution as KA
        from oracle import lemo_oracle as O
        from oracle.f64 import amass_fit_oracle_f64, default_f64
        torch.set_num_threads(32)
        model, vw = synthetic.make_synthetic_smplx(seed=0), make_vposer_weights(2)
        o32 = O.AmassFitOracle(O.SmplxOracle(model), {k_: torch.from_numpy(v) for k_, v in vw.items()}, {k_: torch.from_numpy(v) for k_, v in A['enc_w'].items()},
                               A['ids'], np.asarray(A['Xmean']).reshape(1, 1, -1), A['Xstd'], seq['init_params'], gold['markers_rec'], seq['contact_lbl'], faithful=False)
        o64 = amass_fit_oracle_f64(model, vw, A['enc_w'], A['ids'], A['Xmean'], A['Xstd'], seq['init_params'], gold['markers_rec'], seq['contact_lbl'])
        cond = (KA, o32, o64, default_f64)
    for k in [int(x) for x in T['steps']]:
        lr = float(T[f'lr{k}'])
        assert lr == (0.01 if k <= 60 else 0.005)
        fit.load_state(_state(T, '', k))
        if cond is not None:
            KA, o32, o64, default_f64 = cond
            st_k = _split(T[f's{k}_p'])
            with torch.no_grad():
                for o_, dt in ((o32, torch.float32), (o64, torch.float64)):
                    for n_ in ('transl', 'rot6d', 'other'):
                        getattr(o_, n_).copy_(torch.from_numpy(st_k[n_]).to(dt))
            out = KA.analyse(fit, o32, o64, label=f'amass teacher state {k}', verbose=False)
            KA.check(out, f'amass teacher state {k}')
            REPORT.append(f'amass[v9] state {k}: gradient vs float64 on the engine\'s own piece: worst frame {float(out["cond_gpu"].max()):.1e} '
                          f'({float((out["cond_gpu"] / (KA.ROUND + KA.C_R * out["R"])).max()):.2f} of its computed bound; unconditioned {float(out["unc_gpu"].max()):.1e}), '
                          f'median {float(out["cond_gpu"].median()):.1e} | fp32 CPU path worst {float(out["cond_cpu"].max()):.1e} median {float(out["cond_cpu"].median()):.1e} '
                          f'| decisions differing from float64 {out["n_diff"]}')
        fit.forward(); fit.backward()
        torch.cuda.synchronize()
        L = fit.losses()
        for i, n in enumerate(names):
            r = float(T[f'loss{k}'][i])
            assert abs(L[n] - r) <= 1e-5 * abs(r), (k, n, L[n], r)
        g = fit.grads_with_priors()
        g_eng = np.concatenate([g[n].cpu().numpy() for n, _, _ in GROUPS], axis=1).astype(np.float64)
        g_ref, S = T[f'g{k}'].astype(np.float64), T[f'S{k}']
        # per frame and group: 2e-5 (rounding) + 4 S[frame] (computed exposure to the encoder's kinks, one factor 2 per side) of
        # the group's largest entry; 2e-3 for frames holding an L1 residual at its kink / any frame if a contact speed sits on the
        # threshold (tests/test_gpu_parity.py::test_fit_full_size_golden's bound, now at EVERY recorded step)
        E = np.zeros_like(g_ref)
        for gi, (n, a, b) in enumerate(GROUPS):
            scale = np.abs(g_ref[:, a:b]).max()
            bound = 2e-5 + 4.0 * S[gi]
            if bool(T[f'contact_{k}']):
                bound = np.full_like(bound, 2e-3)
            bound = np.where(T[f'l1_{k}'], np.maximum(bound, 2e-3), bound)
            e = np.abs(g_eng[:, a:b] - g_ref[:, a:b]).max(1) / scale
            bad = np.nonzero(e > bound)[0]
            assert bad.size == 0, (k, n, bad.tolist(), e[bad].tolist(), bound[bad].tolist())
            # the arithmetic where nothing flipped: the median over frames of the per-frame maximum may not exceed 3 x the same
            # statistic of the REFERENCE's fp32 gradient against float64 (g64 in the fixture) + 2e-5 -- late in the fit every frame is
            # within reach of some kink (S median 3e-3 at step 99) and the reference itself sits 2e-3 from float64
            e_ref = np.abs(g_ref[:, a:b] - T[f'g64_{k}'][:, a:b]).max(1) / scale
            worst_med = max(worst_med, float(np.median(e)))
            assert float(np.median(e)) <= 3.0 * float(np.median(e_ref)) + 2e-5, (k, n, float(np.median(e)), float(np.median(e_ref)))
            E[:, a:b] = (bound * scale)[:, None]
        with torch.cuda.stream(s):
            fit.step(1, use_graph=True)
        torch.cuda.synchronize()
        st = fit.save_state()
        assert int(st['step']) == k + 1
        got = _cat_state(st)
        before = dict(p=T[f's{k}_p'], m=T[f's{k}_m'], v=T[f's{k}_v'])
        TC.check_adam_arithmetic(f'amass step {k}', before, g_eng.astype(np.float32), got, k, lr)
        reg, _ = TC.check_next_state(f'amass[v{conv_variant}] step {k} (lr {lr:g})', got['p'], T[f's{k + 1}_p'], E, T[f's{k + 1}_v'], k, lr, REPORT)
        # besides the derived per-entry bound: a flat gate at 3 x the largest value measured over all recorded steps and both kernel
        # families (1.7e-3 lr, profiles/r04_teacher.txt) -- what replaces "10 steps: max < 1e-2, mean < 1e-4" 
        assert reg <= 5e-3, (k, reg)
    REPORT.append(f'amass[v{conv_variant}]: median-over-frames gradient error, worst step/group: {worst_med:.1e} of the group maximum')
    # Several graph-replayed steps in a row from a reference state (the teacher-forced checks are one-step statements; a state
    # carry-over defect between replays -- h1 reuse after load_state, a stale moment buffer -- would only show over consecutive steps):
    # reference states 0 -> 2 (two steps) and 60 -> 63 (three steps ACROSS the lr switch, opt_amass_temp.py:349-353), the engine's state
    # against the reference's own: Adam moments within what the per-step gradient parity allows, parameters within n x the one-step gate
    for k0, n in ((0, 2), (60, 3), (10, 5), (62, 5)):        # (10 -> 15, 62 -> 67: round 6, five steps from mid-fit states: make_teacher.AMASS_EXTRA_STATES)
        fit.load_state(_state(T, '', k0))
        with torch.cuda.stream(s):
            fit.step(n, use_graph=True)
        torch.cuda.synchronize()
        st = fit.save_state()
        assert int(st['step']) == k0 + n
        got = _cat_state(st)
        lr_n = 0.01 if k0 + n - 1 <= 60 else 0.005
        dp = np.abs(got['p'] - T[f's{k0 + n}_p']).max() / lr_n
        dm = np.abs(got['m'] - T[f's{k0 + n}_m']).max() / max(np.abs(T[f's{k0 + n}_m']).max(), 1e-30)
        REPORT.append(f'amass[v{conv_variant}] {n} replayed steps from reference state {k0}: max |dp|/lr {dp:.2e}, exp_avg max-rel {dm:.2e}')
        # measured 1.6e-1 lr (0 -> 2: 4843 of the 7735 entries are noise-level at step 0, see above) and 3.7e-3 lr (60 -> 63), exp_avg
        # 4e-4 / 3e-3; a lost update or a wrong lr level is >= 1 lr on every entry
        assert dp <= MULTI_STEP_GATE[k0] and dm <= (1e-2 if n <= 3 else 2e-2), (k0, n, dp, dm)      # (exp_avg after five steps: measured 5.8e-3)
    # body_params_opt_t_72 of the reference's LAST forward (opt_amass_temp.py:457) = state 99 through the 6-D -> aa conversion
    fit.load_state(_state(T, '', 99))
    fit.forward()
    torch.cuda.synchronize()
    assert np.abs(fit.params72().cpu().numpy() - T['p72_final']).max() < 2e-6


@pytest.mark.timeout(900)
@pytest.mark.parametrize('full_vertices', [False, True])
def test_perframe_loop_teacher_forced_full_size(dev, full_vertices):
    """BASELINE configs[0] (opt_amass_perframe.py:291-364): the lr-0.1 first frame and the warm-started second frame at steps
    0, 1, 59, 60, 61, 79, 80, 81, 99 -- both lr switches from both sides -- replacing the "10 x yardstick + 10 mm" gate"""
    from test_teacher_emu import perframe_teacher_check
    from lemo_amd.fitting import AmassTemporalFitter, LOSS_WEIGHTS
    from lemo_amd.vposer import make_vposer_weights
    T = np.load(os.path.join(GOLDEN, 'teacher_perframe.npz'))
    A = load_assets()
    model, vw = synthetic.make_synthetic_smplx(seed=0), make_vposer_weights(2)
    w = dict(LOSS_WEIGHTS, contact_vel=0.0, smooth=0.0)
    mk = lambda lr0: AmassTemporalFitter(model, vw, A['enc_w'], A['ids'], A['Xmean'], A['Xstd'], 1, dev, weights=w, full_vertices=full_vertices,
                                         lr0=lr0, lr1=0.01, lr_switch=60, lr2=0.003, lr_switch2=80, per_frame=True)
    s = torch.cuda.Stream(dev)

    def step_fn(fit):
        with torch.cuda.stream(s):
            fit.step(1, use_graph=True)
        torch.cuda.synchronize()
    perframe_teacher_check(T, mk, REPORT, step_fn=step_fn)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('stage', ['S2', 'S3'])
def test_prox_chained_windows_teacher_forced(dev, stage):
    """N3 (VERDICT r03 missing #1): two CHAINED windows -- window 2 initialised by the reference's reader from the pickles its
    writer produced after window 1, mean betas, first 15 % frozen -- at steps 0, 1, 2, 30, 59 of each window, on the device"""
    import __graft_entry__ as ge
    from test_teacher_emu import prox_teacher_check
    T = np.load(os.path.join(GOLDEN, 'teacher_prox.npz'))
    prox_teacher_check(T, stage, lambda prob, first: ge.prox_engine_for(prob, dev, first_batch_flag=first)[0], REPORT, to_np=lambda t: t.cpu().numpy())


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('stage', ['S2', 'S3'])
def test_prox_chained_windows_teacher_forced_baseline_size(dev, stage):
    """VERDICT r04 missing #2 / next #4: the same two chained windows at the BASELINE shape -- B = 100, V = 10475, real id tables, 64^3 SDF
    (tests/golden/teacher_prox_full.npz: states, gradients and loss_dict entries the REFERENCE's own closure / optimiser / pickle writer /
    reader produced at steps 0, 1, 30, 59 of each window, with the float64 gradient and the computed kink exposure of every state).  Per
    step: 14 losses <= 1e-5, the gradient frame by frame inside the computed bound (replaces the flat 3e-3 vs the build's own oracle of
    tests/test_gpu_r2.py::test_prox_engine_baseline_size), torch's Adam arithmetic bit for bit, the next state vs the reference's."""
    import __graft_entry__ as ge
    from test_teacher_emu import prox_teacher_check
    path = os.path.join(GOLDEN, 'teacher_prox_full.npz')
    assert os.path.exists(path), 'tests/golden/make_teacher.py prox_full writes it (build container, runs the reference)'
    T = np.load(path)
    prox_teacher_check(T, stage, lambda prob, first: ge.prox_engine_for(prob, dev, first_batch_flag=first)[0], REPORT, to_np=lambda t: t.cpu().numpy(),
                       size='full')
