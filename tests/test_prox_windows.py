"""PROX sliding-window schedule and result-pkl wire format (host logic, CPU)."""
import os
import pickle

import numpy as np
import pytest

from lemo_amd import prox_windows as PW
from oracle.prox_oracle import slide_index_oracle


@pytest.mark.parametrize('n,b', [(100, 100), (169, 100), (170, 100), (171, 100), (240, 100), (1000, 100), (37, 10),
                                 (10, 10), (5, 10), (311, 64)])
def test_window_schedule_matches_reference_loop(n, b):
    assert PW.slide_frame_index(n, b).tolist() == slide_index_oracle(n, b)
    wins = PW.sliding_windows(n, b)
    assert wins[0][0] == 0 and all(e - s <= b for s, e in wins)
    assert all(wins[i + 1][0] - wins[i][0] == int(0.7 * b) for i in range(len(wins) - 1))


def test_reference_batches_are_the_dataloader_chunks():
    """data_parser_slide.py:199-212 + DataLoader(batch_size, drop_last=True) (main_slide.py:146-149): full windows are
    one batch each, tail windows are glued, the incomplete rest is dropped"""
    for n, b in [(100, 100), (240, 100), (300, 100), (1000, 100), (37, 10)]:
        flat = slide_index_oracle(n, b)
        ref = [flat[i:i + b] for i in range(0, len(flat) - b + 1, b)]
        got = PW.reference_batches(n, b)
        assert [g.tolist() for g in got] == ref
    b300 = PW.reference_batches(300, 100)
    assert b300[-1].tolist() == list(range(210, 300)) + list(range(280, 290))          # a batch that straddles two tail windows
    full = [w for w in PW.sliding_windows(1000, 100) if w[1] - w[0] == 100]
    assert [g.tolist() for g in PW.reference_batches(1000, 100)][:len(full)] == [list(range(s, e)) for s, e in full]


def test_frozen_prefix():
    assert PW.frozen_prefix(100, True) == 0 and PW.frozen_prefix(100, False) == 15 and PW.frozen_prefix(10, False) == 1


def test_result_pkl_wire_format_and_overlap_init(tmp_path):
    B = 10
    rng = np.random.default_rng(0)
    cam = {'rotation': rng.normal(size=(B, 3, 3)), 'translation': rng.normal(size=(B, 3))}
    body = {k: rng.normal(size=(B, d)).astype(np.float32) for k, d in
            dict(transl=3, global_orient=3, betas=10, left_hand_pose=12, right_hand_pose=12, jaw_pose=3, leye_pose=3,
                 reye_pose=3, expression=10).items()}
    emb, bp = rng.normal(size=(B, 32)).astype(np.float32), rng.normal(size=(B, 63)).astype(np.float32)
    cur, prox = str(tmp_path / 'cur'), str(tmp_path / 'prox')
    names = [f's001_frame_{i:05d}' for i in range(B)]
    for i, fn in enumerate(names):                                          # the "PROX fit" every window can fall back to
        PW.write_result_pkl(PW.result_path(prox, fn), cam, body, emb * 0, bp * 0, i)
    d = PW.write_result_pkl(PW.result_path(cur, names[3]), cam, body, emb, bp, 3)
    assert set(d) == {'camera_rotation', 'camera_translation', *body, 'pose_embedding', 'body_pose'}
    with open(PW.result_path(cur, names[3]), 'rb') as f:
        raw = f.read()
    assert raw[:2] == b'\x80\x02'                                            # pickle protocol 2
    back = pickle.loads(raw)
    assert all(v.shape[0] == 1 for v in back.values()) and np.array_equal(back['body_pose'][0], bp[3])
    init = PW.init_params_for_window(names, cur, prox)
    assert set(init) == set(PW.BODY_PARAM_KEYS) and init['pose_embedding'].shape == (B, 32)
    assert np.array_equal(init['pose_embedding'][3], emb[3]) and not init['pose_embedding'][4].any()   # newest result wins


def test_run_recording_overwrites_the_overlap(tmp_path):
    n, B = 24, 10
    names = [f'f{i:03d}' for i in range(n)]
    cur, prox = str(tmp_path / 'cur'), str(tmp_path / 'prox')
    z = lambda d: np.zeros((n, d), np.float32)
    body0 = {k: z(d) for k, d in dict(transl=3, global_orient=3, betas=10, left_hand_pose=12, right_hand_pose=12,
                                      jaw_pose=3, leye_pose=3, reye_pose=3, expression=10).items()}
    for i, fn in enumerate(names):
        PW.write_result_pkl(PW.result_path(prox, fn), {}, body0, z(32), z(63), i)
    calls = []

    def fit_window(fns, init, first, n_frozen):
        w = len(calls)
        calls.append((fns[0], fns[-1], first, n_frozen, float(init['transl'][:, 0].max())))
        body = {k: np.full((len(fns),) + v.shape[1:], w + 1, np.float32) for k, v in body0.items()}
        return {}, body, np.zeros((len(fns), 32), np.float32), np.zeros((len(fns), 63), np.float32)

    assert PW.run_recording(names, B, cur, prox, fit_window) == 4     # (0,10) (7,17) (14,24) and the short tail (21,24)
    assert calls[0][:4] == ('f000', 'f009', True, 0) and calls[1][:4] == ('f007', 'f016', False, 1)
    assert calls[1][4] == 1.0 and calls[2][4] == 2.0            # a window starts from the previous window's overlap
    assert PW.read_prox_pkl(PW.result_path(cur, 'f008'))['transl'][0] == 2.0    # later window overwrote the overlap
    assert PW.read_prox_pkl(PW.result_path(cur, 'f020'))['transl'][0] == 3.0
    assert PW.read_prox_pkl(PW.result_path(cur, 'f023'))['transl'][0] == 4.0


def test_reference_written_pickles_round_trip(tmp_path):
    """tests/golden/prox_result_ref_frame{0,5}.pkl were written by the REFERENCE's own lines
    (fit_temp_loadprox_slide.py:577-594, exec'd by make_golden.py on a fitted window): the product reader takes them, and
    the product writer emits the same keys / shapes / dtypes / values / pickle protocol for the same parameters"""
    from conftest import GOLDEN
    for i in (0, 5):
        path = os.path.join(GOLDEN, f'prox_result_ref_frame{i}.pkl')
        with open(path, 'rb') as f:
            raw = f.read()
        assert raw[:2] == b'\x80\x02'
        ref = pickle.loads(raw)
        got = PW.read_prox_pkl(path)
        assert set(got) == set(PW.BODY_PARAM_KEYS) and all(np.array_equal(got[k], ref[k][0]) for k in got)
        cam = {k[len('camera_'):]: np.repeat(v, 3, 0) for k, v in ref.items() if k.startswith('camera_')}
        body = {k: np.repeat(v, 3, 0) for k, v in ref.items() if not k.startswith('camera_') and k not in ('pose_embedding', 'body_pose')}
        mine = PW.write_result_pkl(str(tmp_path / f'{i}.pkl'), cam, body, np.repeat(ref['pose_embedding'], 3, 0), np.repeat(ref['body_pose'], 3, 0), 1)
        assert set(mine) == set(ref)
        for k in ref:
            assert mine[k].shape == ref[k].shape and mine[k].dtype == ref[k].dtype and np.array_equal(mine[k], ref[k]), k


@pytest.mark.timeout(1500)
def test_two_windows_chained_through_the_native_engine(emu_lib, tmp_path):
    """N3 end to end on the (emulated) device: a 17-frame recording, batch 10 -> windows (0,10) and (7,17); every window is
    fitted by the native PROX engine (lemo_prox_*) from the newest pickles; the second window starts from the first one's
    results on the 3-frame overlap and leaves its frozen first frame (int(0.15 * 10) = 1) untouched"""
    import torch
    import __graft_entry__ as ge
    from lemo_amd.prox import ENGINE_PARAMS
    n, B = 17, 10
    base = ge.prox_small_problem(B=n, stage='S2')
    names = [f's001_frame_{i:05d}' for i in range(n)]
    cur, prox = str(tmp_path / 'cur'), str(tmp_path / 'prox')
    P0 = base['params']
    body0 = {k: np.asarray(P0[k], np.float32) for k in ('transl', 'global_orient', 'betas', 'left_hand_pose', 'right_hand_pose', 'jaw_pose',
                                                       'leye_pose', 'reye_pose', 'expression')}
    for i, fn in enumerate(names):                                   # the per-frame PROX fits every window can fall back to
        PW.write_result_pkl(PW.result_path(prox, fn), {}, body0, np.asarray(P0['pose_embedding'], np.float32), np.zeros((n, 63), np.float32), i)
    seen = []

    def fit_window(fns, init, first, n_frozen):
        s = names.index(fns[0])
        prob = dict(base, B=len(fns), params=init, gt_joints=base['gt_joints'][s:s + len(fns)], joints_conf=base['joints_conf'][s:s + len(fns)])
        eng, bm = ge.prox_engine_for(prob, torch.device('cpu'), first_batch_flag=first, lib=emu_lib)
        before = {k: eng.P[k].clone() for k, _ in ENGINE_PARAMS}
        eng.step(2, use_graph=False)
        assert eng.nonfinite_step() == 0
        seen.append((s, first, n_frozen, before, {k: eng.P[k].clone() for k, _ in ENGINE_PARAMS}))
        body = {k: eng.P[k].numpy() for k, _ in ENGINE_PARAMS[:-1]}
        body['betas'] = np.asarray(init['betas'], np.float32)
        return {}, body, eng.P['pose_embedding'].numpy(), np.zeros((len(fns), 63), np.float32)

    assert PW.run_recording(names, B, cur, prox, fit_window) == 2
    (s0, f0, z0, b0, a0), (s1, f1, z1, b1, a1) = seen
    assert (s0, f0, z0) == (0, True, 0) and (s1, f1, z1) == (7, False, 1)
    for k, _ in ENGINE_PARAMS:
        assert torch.equal(b1[k][:3], a0[k][7:10]), k                  # window 2 starts from window 1's results on the overlap
        assert torch.equal(a1[k][:1], b1[k][:1]), k                    # ... and its frozen first frame does not move
    assert not torch.equal(a1['transl'][1:], b1['transl'][1:]) and not torch.equal(a0['transl'], b0['transl'])
    assert np.array_equal(PW.read_prox_pkl(PW.result_path(cur, names[8]))['transl'], a1['transl'][1].numpy())   # overlap overwritten
    assert np.array_equal(PW.read_prox_pkl(PW.result_path(cur, names[3]))['transl'], a0['transl'][3].numpy())


def test_window_two_starts_where_the_reference_started_it(tmp_path):
    """tests/golden/teacher_prox.npz holds what the REFERENCE's window 2 started from: its own reader
    (data_parser_slide.py:106-126, newest-result rule :326-331) over the pickles its own writer produced after window 1
    (fit_temp_loadprox_slide.py:577-594), then its mean-betas initialisation (:495-499).  The product's writer / reader /
    ``window_start_params`` on window 1's final state must give the same start, bit for bit."""
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    from make_teacher import prox_recording, PROX_N, PROX_B
    from lemo_amd.prox import ENGINE_PARAMS
    T = np.load(os.path.join(GOLDEN, 'teacher_prox.npz'))
    dims = dict(ENGINE_PARAMS)
    for stage in ('S2', 'S3'):
        base = prox_recording(stage)
        P0 = base['params']
        names = [f's001_frame_{i:05d}' for i in range(PROX_N)]
        cur, prox = str(tmp_path / stage / 'cur'), str(tmp_path / stage / 'prox')
        body0 = {k: np.asarray(P0[k], np.float32) for k in ('transl', 'global_orient', 'betas', 'left_hand_pose', 'right_hand_pose', 'jaw_pose',
                                                           'leye_pose', 'reye_pose', 'expression')}
        for i, fn in enumerate(names):
            PW.write_result_pkl(PW.result_path(prox, fn), {}, body0, np.asarray(P0['pose_embedding'], np.float32), np.zeros((PROX_N, 63), np.float32), i)
        (s0, e0), (s1, e1) = PW.sliding_windows(PROX_N, PROX_B)
        # window 1 started from the per-frame fits with ITS mean betas ...
        start0 = PW.window_start_params(PW.init_params_for_window(names[s0:e0], cur, prox))
        assert np.array_equal(start0['betas'], T[f'{stage}_w0_betas']) and not np.array_equal(start0['betas'], body0['betas'][s0:e0])
        # ... and ended in the fixture's last recorded state (after step 59): write it the way the product writes results
        order = [str(n) for n in T[f'{stage}_w0_names']]
        last = int(T['steps'][-1]) + 1
        pf, o, fin = T[f'{stage}_w0_s{last}_p'], 0, {}
        for n in order:
            fin[n] = pf[:, o:o + dims[n]]
            o += dims[n]
        body = {k: fin[k] for k in order if k != 'pose_embedding'}
        body['betas'] = T[f'{stage}_w0_betas']
        for i, fn in enumerate(names[s0:e0]):
            PW.write_result_pkl(PW.result_path(cur, fn), {}, body, fin['pose_embedding'], np.zeros((e0 - s0, 63), np.float32), i)
        start1 = PW.window_start_params(PW.init_params_for_window(names[s1:e1], cur, prox))
        assert np.array_equal(start1['betas'], T[f'{stage}_w1_betas'])
        p1, o = T[f'{stage}_w1_s0_p'], 0
        for n in [str(x) for x in T[f'{stage}_w1_names']]:
            assert np.array_equal(start1[n], p1[:, o:o + dims[n]]), (stage, n)
            o += dims[n]
        assert np.array_equal(start1['transl'][:e0 - s1], fin['transl'][s1 - s0:])       # the overlap IS window 1's result
