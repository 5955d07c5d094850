"""How tight can GPU-vs-reference parity be?  (-m gpu)  The reference's own CPU path is fp32; here BOTH the GPU engine
and the fp32 CPU oracle are compared with the same restatement evaluated in float64 (oracle/f64.py), so every tolerance
is a multiple of what fp32 arithmetic costs the reference itself instead of a guessed constant.  Measured numbers of the
round: profiles/r02_gates.txt (tools/r02_gates.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from lemo_amd import synthetic
from lemo_amd.assets import load_assets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _run(dev, prob, markers, weights, steps):
    from lemo_amd.fitting import AmassTemporalFitter
    from oracle import lemo_oracle as O
    from oracle.f64 import amass_fit_oracle_f64, default_f64, flip_sensitivity
    ej = list(range(21)) if prob['V'] < 9930 else None
    so = O.SmplxOracle(prob['model'], extra_joint_ids=ej)
    vw = {k: torch.from_numpy(v) for k, v in prob['vposer_w'].items()}
    ew = {k: torch.from_numpy(v) for k, v in prob['enc_w'].items()}
    o32 = O.AmassFitOracle(so, vw, ew, prob['ids'], np.asarray(prob['Xmean']).reshape(1, 1, -1), prob['Xstd'], prob['seq']['init_params'],
                           markers, prob['seq']['contact_lbl'], faithful=False, weights=weights)
    o64 = amass_fit_oracle_f64(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'],
                               prob['seq']['init_params'], markers, prob['seq']['contact_lbl'], weights=weights, extra_joint_ids=ej)
    fit = AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'], prob['B'], dev,
                              weights=weights, full_vertices=True)
    fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    fit.forward(); fit.backward(); torch.cuda.synchronize()
    t32, p32, _, _ = o32.losses(); t32.backward()
    with default_f64():
        t64, p64, _, _ = o64.losses(); t64.backward()
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-300)
    S0 = flip_sensitivity(o64)              # computed: what the encoder's kinks can do to each frame's gradient (oracle/f64.py)
    L = fit.losses()
    out = dict(loss_gpu={k: rel(L[k], float(p64[k])) for k in p64 if float(p64[k]) != 0.0},
               loss_cpu={k: rel(float(p32[k]), float(p64[k])) for k in p64 if float(p64[k]) != 0.0}, grad_gpu={}, grad_cpu={}, traj=[])
    g = fit.grads_with_priors()
    for k, a32, a64 in (('transl', o32.transl, o64.transl), ('rot6d', o32.rot6d, o64.rot6d), ('other', o32.other, o64.other)):
        n = a64.grad.abs().max()
        eg, ec = (g[k].cpu().double() - a64.grad).abs() / n, (a32.grad.double() - a64.grad).abs() / n
        out['grad_gpu'][k], out['grad_cpu'][k] = float(eg.max()), float(ec.max())
        # per-frame maxima, median over frames: what the arithmetic does where no kink was crossed (see _check)
        out.setdefault('gradmed_gpu', {})[k] = float(eg.max(1).values.median())
        out.setdefault('gradmed_cpu', {})[k] = float(ec.max(1).values.median())
        out.setdefault('gradframe_gpu', {})[k] = eg.max(1).values
        out.setdefault('gradframe_cpu', {})[k] = ec.max(1).values
    out['S0'] = S0
    o32.opt.zero_grad(); o64.opt.zero_grad()
    fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    s = torch.cuda.Stream(dev)
    lr, Ssum = 0.01, 0.0
    for _ in range(steps):
        Sj = flip_sensitivity(o64, subsets=1)            # at the parameters this step's gradient is taken at
        Ssum += float(max(v.mean() for v in Sj.values()))
        with torch.cuda.stream(s):
            fit.step(1, use_graph=True)
        torch.cuda.synchronize()
        h32 = o32.step()
        with default_f64():
            h64 = o64.step()
        p = o64.params75()
        dg, dc = (fit.params75().cpu().double() - p).abs(), (o32.params75().double() - p).abs()
        out['traj'].append(dict(gpu_max=float(dg.max()), gpu_mean=float(dg.mean()), cpu_max=float(dc.max()), cpu_mean=float(dc.mean()),
                                tot_gpu=rel(fit.losses()['total'], h64['total']), tot_cpu=rel(h32['total'], h64['total']), kink_budget=lr * Ssum))
    return out


def _check(r, tag, flips=False, grad_flips=None):
    print(f'\n{tag}: gradient max-rel vs float64  gpu {r["grad_gpu"]}  cpu-f32 {r["grad_cpu"]}')
    for i, t in enumerate(r['traj']):
        print(f'   step {i}: params vs f64 gpu max {t["gpu_max"]:.1e} mean {t["gpu_mean"]:.1e} | cpu-f32 max {t["cpu_max"]:.1e} mean {t["cpu_mean"]:.1e}'
              f' | total rel gpu {t["tot_gpu"]:.1e} cpu {t["tot_cpu"]:.1e}')
    # every loss scalar of iteration 0: north_star's 1e-5, against float64
    assert max(r['loss_gpu'].values()) <= 1e-5, r['loss_gpu']
    # gradients: the GPU may not be further from float64 than 3x the fp32 CPU path is (worst group), + 1e-5 floor
    # (`grad_flips`: with the thresholded contact term on, a speed within an ulp of 0.1 m/s is inside the mean on one side
    # and outside on the other already in iteration 0 -- a few gradient entries move by ~1e-3 of the largest one; which
    # build trips one is luck, so with the term on only that bound is asserted and the strict comparison runs with it off)
    # Gradients, frame by frame, against a COMPUTED bound (replaces a blanket 2e-3 justified by narrative):
    #   err[frame] <= ROUND + 2 x S0[frame]
    # ROUND = 2e-5 of the group's largest entry (rounding of an fp32-accurate path; the fp32 CPU oracle's own worst frame
    # without kink effects is ~5e-6) and S0 = oracle.f64.flip_sensitivity: how far that frame's gradient moves when every
    # LeakyReLU unit of the encoder within 3e-6 x (layer maximum) of its kink takes the other branch -- evaluated in float64
    # at this very point.  A defect confined to a few frames has to hide below THEIR computed exposure (1e-4 .. 6e-4 at
    # BASELINE size, 0 where no unit is near a kink), not below a constant.  The two other kink families keep a constant:
    # frames with an L1 residual / contact speed near its kink (`grad_flips`) are bounded by 2e-3.
    grad_flips = flips if grad_flips is None else grad_flips
    ROUND = 2e-5
    n_exposed = 0
    for k in r['grad_gpu']:
        eg, S = r['gradframe_gpu'][k], r['S0'][k]
        bound = ROUND + 2.0 * S
        n_exposed = max(n_exposed, int((S > ROUND).sum()))
        bad = (eg > (torch.full_like(bound, 2e-3) if grad_flips else bound)).nonzero().flatten().tolist()
        assert not bad, (k, bad, [float(eg[i]) for i in bad], [float(bound[i]) for i in bad])
        # the reference's own fp32 path obeys the same computed bound (if it did not, the bound would be wrong, not the GPU)
        ec = r['gradframe_cpu'][k]
        assert not (ec > (torch.full_like(bound, 2e-3) if grad_flips else bound)).any(), (k, 'cpu-f32 breaks the computed bound')
    print(f'   computed kink exposure S0: median {max(float(v.median()) for v in r["S0"].values()):.1e} max {max(float(v.max()) for v in r["S0"].values()):.1e}'
          f' of the largest gradient entry; frames with S0 > {ROUND:g}: {n_exposed} of {len(r["S0"]["transl"])}')
    # The max norm measures which units happened to flip; the median over frames of the per-frame maximum measures the
    # arithmetic, and that must not be worse than the reference's own fp32 path.
    print(f'   per-frame gradient error, median over frames: gpu {r["gradmed_gpu"]}  cpu-f32 {r["gradmed_cpu"]}')
    worst_med = max(r['gradmed_cpu'].values())
    for k, v in r['gradmed_gpu'].items():
        assert v <= 3.0 * worst_med + 2e-6, (k, v, worst_med)
    # trajectory: mean parameter error and the loss of every iteration.  `flips`: the contact term averages the speeds
    # ABOVE 0.1 m/s (opt_amass_temp.py:429-443): a speed within an ulp of the threshold is in the mean on one side and out on
    # the other, which moves a few gradient entries by ~1e-3 and Adam (lr 1e-2) turns that into 1e-4-sized parameter
    # differences -- whether a given rounding pattern trips one is luck (the same small problem ran flip-free in
    # profiles/r02_gates.txt and tripped one in the next build), so with the term on only a bound is asserted
    assert r['traj'][0]['tot_gpu'] <= 1e-5                                               # the loss of iteration 0 is always tight
    if not grad_flips:
        assert r['traj'][0]['gpu_max'] <= 5e-6                                           # ... and so is the first update
    for i, t in enumerate(r['traj']):
        if flips:
            assert t['gpu_mean'] <= 2e-4 and t['tot_gpu'] <= 2e-3, (i, t)
        else:
            # mean parameter distance from float64: 3 x the fp32 CPU path's + what the encoder's kinks can add -- a gradient
            # that moves by a fraction S changes an Adam update by at most ~lr x S (kink_budget = lr x sum over the steps
            # so far of the mean computed exposure; 0 when no unit is near its kink)
            assert t['gpu_mean'] <= 3.0 * max(x['cpu_mean'] for x in r['traj'][:i + 1]) + 2e-6 + 3.0 * t['kink_budget'], (i, t)
            assert t['tot_gpu'] <= 1e-4 + 10.0 * t['kink_budget'], (i, t)      # (the 1e6-weighted smoothness term follows the parameters)


def test_small_problem_vs_float64(dev):
    import __graft_entry__ as ge
    from oracle import lemo_oracle as O
    small = ge.small_problem()
    _, mk = ge.oracle_for(small)
    r = _run(dev, small, mk, None, 10)
    _check(r, 'small problem, all terms', flips=True)
    # contact term off: the L1 marker term is the only non-smooth one left -- a residual within an ulp of zero has one sign in
    # fp32 and the other in float64, which moves one gradient entry by 2 w / N; rarer and smaller than a threshold flip
    r = _run(dev, small, mk, dict(O.LOSS_WEIGHTS, contact_vel=0.0), 10)
    _check(r, 'small problem, contact term off', flips=True, grad_flips=False)
    # no non-smooth term at all (contact and marker weights 0: priors + the 1e6-weighted smoothness term through the whole
    # encoder / LBS / VPoser chain): nothing amplifies fp32 noise -- after 10 replayed steps every parameter is within 2e-5 of
    # float64 (measured 1.2e-5, with gradients CLOSER to float64 than the fp32 CPU path's) and the loss of every iteration
    # within 1e-5
    r = _run(dev, small, mk, dict(O.LOSS_WEIGHTS, contact_vel=0.0, rec_markers=0.0), 10)
    _check(r, 'small problem, contact and marker terms off')
    assert max(t['tot_gpu'] - 10.0 * t['kink_budget'] for t in r['traj']) < 1e-5
    # single entries: within 2e-5 of float64 unless the computed exposure of the encoder's kinks explains more (an entry whose
    # gradient is moved across zero by a flipped unit takes an update of the other sign: 2 lr)
    assert max(t['gpu_max'] for t in r['traj']) < 2e-5 + (2 * 0.01 if r['traj'][-1]['kink_budget'] > 1e-7 else 0.0)


@pytest.mark.timeout(1200)
def test_baseline_size_vs_float64(dev):
    from lemo_amd.vposer import make_vposer_weights
    from oracle import lemo_oracle as O
    torch.set_num_threads(32)
    A = load_assets()
    g = np.load(os.path.join(GOLDEN, 'amass_iter.npz'))
    full = dict(model=synthetic.make_synthetic_smplx(seed=0), vposer_w=make_vposer_weights(2), enc_w=A['enc_w'], ids=A['ids'], Xmean=A['Xmean'],
                Xstd=A['Xstd'], seq=synthetic.make_synthetic_sequence(0, B=119), B=119, V=10475)
    _check(_run(dev, full, g['markers_rec'], None, 6), 'B=119 V=10475, all terms', flips=True)
    # the same with the contact term off: its threshold is not the only kink (see _check)
    _check(_run(dev, full, g['markers_rec'], dict(O.LOSS_WEIGHTS, contact_vel=0.0), 3), 'B=119 V=10475, contact term off', flips=True)


@pytest.mark.timeout(1500)
def test_gradient_error_statistics_over_seeds(dev):
    """one sequence said "the GPU's worst frame is 3-8 x further from float64 than the reference's
    fp32 CPU path" -- luck of which kink trips, or a property of the build?  Five sequences (seeds 0-4) x the encoder's kernel
    families (9 fused pairs + fused head / tail = default, 4 split-f16, 3 split-bf16, 2 fp32 MFMA): per seed the iteration-0 gradient of the GPU engine
    and of the fp32 CPU oracle against the float64 oracle -- worst frame and median over frames of the per-frame maximum, per
    parameter group.  Asserted: the median over seeds of (GPU worst / CPU worst) <= 2 for the shipped families (measured 1.15), and the GPU's median-frame
    error <= 3 x the CPU's + 2e-6 on every seed (the arithmetic where nothing flipped).  The table is the evidence
    (profiles/r05_gates.txt; round 4: r04_gates.txt)."""
    from lemo_amd.fitting import AmassTemporalFitter
    from lemo_amd.vposer import make_vposer_weights
    from oracle import lemo_oracle as O
    from oracle.f64 import amass_fit_oracle_f64, default_f64
    torch.set_num_threads(32)
    A = load_assets()
    model, vw = synthetic.make_synthetic_smplx(seed=0), make_vposer_weights(2)
    so = O.SmplxOracle(model)
    vwt = {k: torch.from_numpy(v) for k, v in vw.items()}
    ewt = {k: torch.from_numpy(v) for k, v in A['enc_w'].items()}
    variants = (9, 4, 3, 2)            # 9 = the default (fused pairs + fused head / tail, layers 0-2 each), 4 = layer by layer f16 x 2, 3 = bf16 x 3, 2 = fp32 MFMA
    fits = {v: AmassTemporalFitter(model, vw, A['enc_w'], A['ids'], A['Xmean'], A['Xstd'], 119, dev, full_vertices=True, conv_variant=v) for v in variants}
    rows, ratio, cond = [], {v: [] for v in variants}, []
    import kink_attribution as KA
    for seed in range(5):
        seq = synthetic.make_synthetic_sequence(seed, B=119)
        f0 = fits[variants[0]]
        f0.load_sequence(seq['target_params'], np.zeros((119, 67, 3), np.float32), seq['contact_lbl'])
        f0.forward(); torch.cuda.synchronize()
        markers = f0.marker_vertices().cpu().numpy().copy()
        o32 = O.AmassFitOracle(so, vwt, ewt, A['ids'], np.asarray(A['Xmean']).reshape(1, 1, -1), A['Xstd'], seq['init_params'], markers, seq['contact_lbl'], faithful=False)
        o64 = amass_fit_oracle_f64(model, vw, A['enc_w'], A['ids'], A['Xmean'], A['Xstd'], seq['init_params'], markers, seq['contact_lbl'])
        t32 = o32.losses()[0]; t32.backward()
        with default_f64():
            t64 = o64.losses()[0]; t64.backward()
        G64 = {k: getattr(o64, k).grad for k in ('transl', 'rot6d', 'other')}
        cpu = {k: ((getattr(o32, k).grad.double() - G64[k]).abs() / G64[k].abs().max()).max(1).values for k in G64}
        cpu_max, cpu_med = max(float(v.max()) for v in cpu.values()), max(float(v.median()) for v in cpu.values())
        for v in variants:
            fit = fits[v]
            fit.load_sequence(seq['init_params'], markers, seq['contact_lbl'])
            fit.forward(); fit.backward(); torch.cuda.synchronize()
            g = fit.grads_with_priors()
            gpu = {k: ((g[k].cpu().double() - G64[k]).abs() / G64[k].abs().max()).max(1).values for k in G64}
            gmax, gmed = max(float(x.max()) for x in gpu.values()), max(float(x.median()) for x in gpu.values())
            ltot = abs(fit.losses()['total'] - float(t64)) / float(t64)
            rows.append((seed, v, gmax, gmed, cpu_max, cpu_med, ltot))
            ratio[v].append(gmax / cpu_max)
            assert gmed <= 3.0 * cpu_med + 2e-6, (seed, v, gmed, cpu_med)
            assert ltot <= 1e-5, (seed, v, ltot)
        # Round 5: the same seed CONDITIONED on the engine's own kink decisions (tests/kink_attribution.py) -- the
        # float64 oracle pinned to the piece of the objective the engine was on, the conditioning of the 6-D decode computed per frame.
        # Every frame of every seed inside ROUND + C_R R[frame]; the engine's median frame <= 1.5 x the fp32 CPU path's; the worst
        # unconditioned frame attributed (which decisions differ from float64's, what each family explains).
        fit = fits[variants[0]]
        fit.load_sequence(seq['init_params'], markers, seq['contact_lbl'])
        out = KA.analyse(fit, o32, o64, label=f'seed {seed}')
        KA.check(out, f'seed {seed}')
        cond.append((seed, float(out['unc_gpu'].max()), float(out['cond_gpu'].max()), float((out['cond_gpu'] / (KA.ROUND + KA.C_R * out['R'])).max()),
                     float(out['cond_gpu'].median()), float(out['cond_cpu'].max()), float(out['cond_cpu'].median()), float(out['R'].max()), out['n_diff']))
    print('\nseed | GPU worst frame: unconditioned -> conditioned on its own decisions (x its computed bound) | conditioned median: GPU / CPU-fp32 | '
          'CPU-fp32 conditioned worst | max R | decisions differing from float64')
    for r in cond:
        print('  %d  | %.2e -> %.2e (%.2f) | %.2e / %.2e | %.2e | %.1e | %s' % (r[0], r[1], r[2], r[3], r[4], r[6], r[5], r[7], r[8]))
    print('\nseed variant | gradient vs float64: GPU worst frame / median frame | CPU-fp32 worst / median | GPU total loss rel')
    for r in rows:
        print('  %d    %d     | %.2e / %.2e | %.2e / %.2e | %.1e' % r)
    for v in variants:
        med = float(np.median(ratio[v]))
        print(f'  variant {v}: GPU worst / CPU worst over the 5 seeds: ' + ' '.join(f'{x:.2f}' for x in ratio[v]) + f'  -> median {med:.2f}')
        # the shipped families (9 = default, 4): median over the seeds <= 2; the others are reported with a looser gate -- five seeds
        # are few for a median of a heavy-tailed ratio (one flipped kink moves a seed's worst frame by 3 - 20 x on EITHER side:
        # seed 2 costs every GPU family the same 1.9e-2 frame, seeds 1 and 3 cost the CPU path more than the GPU)
        assert med <= (2.0 if v >= 4 else 5.0), (v, ratio[v])
