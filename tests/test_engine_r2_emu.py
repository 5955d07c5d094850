"""Round-2 engine features on the emulator library (same kernel sources, host-compiled): per-frame mode
(BASELINE configs[0]), self-consistent ``params72`` after ``step(n)``, the non-finite-loss latch."""
import numpy as np
import pytest
import torch

from conftest import rel_err


def _so_and_weights(prob):
    from oracle import lemo_oracle as O
    so = O.SmplxOracle(prob['model'], extra_joint_ids=list(range(21)))
    vw = {k: torch.from_numpy(v) for k, v in prob['vposer_w'].items()}
    return so, vw


@pytest.mark.timeout(900)
def test_perframe_fit_vs_oracle(emu_lib):
    """opt_amass_perframe.py:291-363 (B = 1, warm start from the previous frame, fresh Adam per frame, three-level lr)
    on the engine's per_frame mode vs oracle/pipeline_oracle.perframe_fit -- which make_golden.py pins to the
    reference's own loop text at 0.0 (row ``perframe.*`` of oracle_vs_reference.txt)."""
    import __graft_entry__ as ge
    from lemo_amd.fitting import PerFrameFitter
    from oracle import pipeline_oracle as PO
    prob = ge.small_problem()
    so, vw = _so_and_weights(prob)
    _, markers = ge.oracle_for(prob)
    betas = prob['seq']['init_params'][0, 6:16]
    mr = markers[:3]
    steps = 5
    ref, last = PO.perframe_fit(so, vw, prob['ids']['markers67'], mr, betas, steps=steps)
    pf = PerFrameFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'], 'cpu', lib=emu_lib)
    got = pf.fit_clip(mr, betas, steps=steps, use_graph=False).numpy()
    assert got.shape == (3, 72)
    # the per-frame loss of the last iteration (marker + priors only; contact and smoothness stay exactly 0)
    L = pf.rest.losses()
    assert L['contact'] == 0.0 and L['smooth'] == 0.0
    assert abs(L['total'] - last[-1]) <= 2e-3 * abs(last[-1])
    # Adam turns fp32 rounding of near-zero gradient entries into O(lr) differences (lr = 0.1 on the first frame), and
    # the L1 marker term has a sign() gradient: the trajectories stay together in the mean, single entries drift
    # (a fresh Adam's first update is lr * sign(g): an entry whose gradient is rounding noise at the warm start moves
    # by +-lr on either side)
    d = np.abs(got - ref)
    assert d.max() < 2e-2 and d.mean() < 5e-4, (d.max(), d.mean())
    ref2, _ = PO.perframe_fit(so, vw, prob['ids']['markers67'], mr[:2], betas, steps=2)
    got2 = pf.fit_clip(mr[:2], betas, steps=2, use_graph=False).numpy()
    assert np.abs(got2 - ref2).max() < 3e-5                              # one update per frame: tight
    # gradient of the per-frame objective at a fixed state (B = 1) vs autograd on the oracle: this is the tight check
    from oracle import lemo_oracle as O
    import torch.nn.functional as F
    p0 = prob['seq']['init_params'][3:4]
    eng = pf.rest
    eng.load_sequence(p0, markers[3:4], np.zeros((1, 4), np.float32))
    eng.forward(); eng.backward()
    pt = torch.from_numpy(p0)
    tr = pt[:, 0:3].clone().requires_grad_(True)
    r6 = O.convert_to_6D_all(pt[:, 3:6]).detach().clone().requires_grad_(True)
    ot = pt[:, 16:].clone().requires_grad_(True)
    p72 = O.convert_to_3D_rot(torch.cat([tr, r6, pt[:, 6:16], ot], -1))
    bp = O.vposer_decode(vw, p72[:, 16:48], 'aa').view(1, -1)
    verts, _, _ = so.forward(p72[:, 6:16], p72[:, 3:6], bp, p72[:, 48:60], p72[:, 60:], p72[:, 0:3])
    ids67 = torch.as_tensor(np.asarray(prob['ids']['markers67'], np.int64))
    loss = (F.l1_loss(verts[:, ids67], torch.from_numpy(markers[3:4])) + 0.02 * torch.mean(p72[:, 16:48] ** 2) +
            0.01 * torch.mean(p72[:, 6:16] ** 2) + 0.01 * torch.mean(p72[:, 48:] ** 2))
    loss.backward()
    g = eng.grads_with_priors()
    assert abs(eng.losses()['total'] - float(loss)) <= 1e-5 * float(loss)
    for k, r in (('transl', tr.grad), ('rot6d', r6.grad), ('other', ot.grad)):
        assert rel_err(g[k], r) < 2e-4, k
    # lr schedule: steps 0..60 lr0, 61..80 0.01, 81.. 0.003 -- checked on one scalar parameter with a constant gradient
    assert pf.first.desc.lr_switch == 60 and pf.first.desc.lr_switch2 == 80
    assert abs(pf.first.desc.lr0 - 0.1) < 1e-7 and abs(pf.rest.desc.lr0 - 0.01) < 1e-7 and abs(pf.first.desc.lr2 - 0.003) < 1e-7


@pytest.mark.timeout(900)
def test_params72_is_the_last_forward_and_nonfinite_latch(emu_lib):
    import __graft_entry__ as ge
    from lemo_amd.fitting import AmassTemporalFitter
    from oracle import lemo_oracle as O
    prob = ge.small_problem()
    ofit, markers = ge.oracle_for(prob)
    fit = AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'],
                              prob['B'], 'cpu', full_vertices=True, lib=emu_lib)
    fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    fit.step(2, use_graph=False)
    for _ in range(2):
        ofit.step()
    # the reference saves body_params_opt_t_72 of the LAST forward (opt_amass_temp.py:457): parameters after n - 1 updates
    p72 = fit.params72()
    assert float((p72 - ofit.last_p72).abs().max()) < 2e-5
    assert float((p72[:, 0:3] - fit.P['transl']).abs().max()) > 0          # ... not the live (post-update) parameters
    fit.forward()
    assert torch.equal(fit.params72()[:, 0:3], fit.P['transl'])             # after a bare forward they coincide
    assert fit.nonfinite_step() == 0
    # poison the target: the total loss becomes NaN in the next iteration; that iteration's update is still applied
    # (FittingMonitor.run_fitting checks AFTER optimizer.step, fitting_temp_slide.py:196-204), every later one is skipped
    fit.target[0, 0, 0] = float('nan')
    fit.step(1, use_graph=False)
    assert fit.nonfinite_step() == 3
    frozen = fit.params75().clone()
    fit.step(1, use_graph=False)
    assert torch.equal(torch.nan_to_num(fit.params75(), nan=7.0), torch.nan_to_num(frozen, nan=7.0))
    assert fit.nonfinite_step() == 3 and int(fit.step_ctr.item()) == 4
    fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    assert fit.nonfinite_step() == 0


def test_duplicate_foot_ids_rejected(emu_lib):
    import __graft_entry__ as ge
    from lemo_amd.fitting import AmassTemporalFitter
    prob = ge.small_problem()
    ids = dict(prob['ids'])
    ids['left_heel'] = np.concatenate([ids['left_heel'], ids['left_heel'][:1]])
    with pytest.raises(AssertionError):
        AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], ids, prob['Xmean'], prob['Xstd'], prob['B'], 'cpu',
                            lib=emu_lib)


@pytest.mark.timeout(900)
def test_concurrent_clips_helper_equals_separate_runs(emu_lib):
    """lemo_amd.sharding.ConcurrentClips on the emulator (no streams there: the clips run one after the other): every clip
    ends exactly where a run on its own ends"""
    import __graft_entry__ as ge
    from lemo_amd.fitting import AmassTemporalFitter
    from lemo_amd.sharding import ConcurrentClips
    prob = ge.small_problem()
    _, markers = ge.oracle_for(prob)
    mk = lambda: AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'],
                                     prob['B'], 'cpu', full_vertices=True, lib=emu_lib)
    fits = [mk(), mk()]
    inits = [prob['seq']['init_params'], prob['seq']['init_params'] * np.float32(0.9)]
    for f, ip in zip(fits, inits):
        f.load_sequence(ip, markers, prob['seq']['contact_lbl'])
    cc = ConcurrentClips(fits)
    cc.prepare(1)
    cc.step(1, use_graph=False)
    cc.synchronize()
    got = cc.params72()
    assert got.shape == (2, prob['B'], 72) and not torch.equal(got[0], got[1])
    live = [f.params75().clone() for f in fits]            # after the update (params72 is the state the last forward saw)
    solo = mk()
    for i, ip in enumerate(inits):
        solo.load_sequence(ip, markers, prob['seq']['contact_lbl'])
        before = solo.params75().clone()
        solo.step(1, use_graph=False)
        assert torch.equal(solo.params72(), got[i]) and torch.equal(solo.params75(), live[i])
        assert not torch.equal(solo.params75(), before)


@pytest.mark.timeout(900)
def test_run_of_iterations_equals_single_iteration_calls(emu_lib):
    """Inside a run of iterations (one graph, or one eager call) the first VPoser layer of iteration i + 1 comes out of
    iteration i's fused tail launch; a call of its own recomputes it with the stand-alone launch of the SAME kernel.  The two must
    agree bit for bit -- this is what makes replaying 20- / 5- / 1-iteration graphs equal to eager launches."""
    import __graft_entry__ as ge
    from lemo_amd.fitting import AmassTemporalFitter
    prob = ge.small_problem()
    _, markers = ge.oracle_for(prob)
    mk = lambda: AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'],
                                     prob['B'], 'cpu', full_vertices=False, lib=emu_lib)
    a, b = mk(), mk()
    for f in (a, b):
        f.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    a.step(3, use_graph=False)
    for _ in range(3):
        b.step(1, use_graph=False)
    assert torch.equal(a.params75(), b.params75()) and torch.equal(a.params72(), b.params72())
    assert a.losses() == b.losses() and int(a.step_ctr.item()) == int(b.step_ctr.item()) == 3


@pytest.mark.timeout(900)
def test_per_frame_clips_side_by_side_equal_sequential(emu_lib):
    """fit_clips_per_frame (stage 1 for several clips in lockstep, one PerFrameFitter each) == each clip's fit_clip, bit for bit;
    clips of different lengths"""
    import __graft_entry__ as ge
    from lemo_amd.fitting import PerFrameFitter, fit_clips_per_frame
    prob = ge.small_problem()
    _, markers = ge.oracle_for(prob)
    betas = [prob['seq']['init_params'][0, 6:16], prob['seq']['init_params'][0, 6:16] * np.float32(0.5)]
    clips = [markers[:3], markers[4:6]]
    mk = lambda: PerFrameFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'], 'cpu', lib=emu_lib)
    pfs = [mk(), mk()]
    got = fit_clips_per_frame(pfs, clips, betas, steps=3, use_graph=False)
    assert [tuple(g.shape) for g in got] == [(3, 72), (2, 72)]
    solo = mk()
    for i in range(2):
        assert torch.equal(solo.fit_clip(clips[i], betas[i], steps=3, use_graph=False), got[i]), i
    assert not torch.equal(got[0][:2], got[1])


@pytest.mark.timeout(900)
def test_per_frame_active_vertex_forward_equals_full_forward(emu_lib):
    """PerFrameFitter forwards only the loss-carrying vertices by default; full_vertices=True regresses all of them like
    the reference's smplx call does.  Same losses, same fit (the blend GEMM's summation order is the only difference)."""
    import __graft_entry__ as ge
    from lemo_amd.fitting import PerFrameFitter
    prob = ge.small_problem()
    _, markers = ge.oracle_for(prob)
    betas = prob['seq']['init_params'][0, 6:16]
    mk = lambda full: PerFrameFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'], 'cpu',
                                     lib=emu_lib, full_vertices=full)
    a, b = mk(False), mk(True)
    assert not a.first.full and b.first.full and b.first.vertices().shape[1] == prob['V']
    ra = a.fit_clip(markers[:2], betas, steps=2, use_graph=False)
    rb = b.fit_clip(markers[:2], betas, steps=2, use_graph=False)
    assert float((ra - rb).abs().max()) < 5e-5
    la, lb = a.rest.losses(), b.rest.losses()
    assert abs(la['total'] - lb['total']) <= 1e-5 * abs(lb['total'])


def test_batched_perframe_fit_is_bit_identical_to_solo(emu_lib):
    """stage 1 for several clips through ONE engine (row i = current frame of clip i, lemo_fit_desc.per_frame: every row a
    fit of its own) == each clip fitted alone with B = 1, bit for bit -- clips of different lengths, different betas"""
    import __graft_entry__ as ge
    from lemo_amd.fitting import PerFrameFitter, BatchedPerFrameFitter
    prob = ge.small_problem()
    _, markers = ge.oracle_for(prob)
    b0 = prob['seq']['init_params'][0, 6:16]
    clips = [markers[:3], markers[3:5] + 0.01, markers[6:10] * 1.02]
    betas = [b0, b0 * 0.5, -b0]
    args = (prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'], 'cpu')
    pf = PerFrameFitter(*args, lib=emu_lib)
    solo = [pf.fit_clip(m, b, steps=7, use_graph=False).clone() for m, b in zip(clips, betas)]
    bf = BatchedPerFrameFitter(*args, batch=4, lib=emu_lib)            # one spare row
    got = bf.fit_clips(clips, betas, steps=7, use_graph=False)
    for a, b in zip(solo, got):
        assert a.shape == b.shape and torch.equal(a, b)
