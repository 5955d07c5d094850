"""world_size-2 gloo test of the multi-GPU path (sequence sharding + ONE all-gather)."""
import os
import numpy as np
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lemo_amd.sharding import fit_sharded, fit_sharded_concurrent, gather_fitted_params, my_sequences, unshard_order


def test_partition_is_a_bijection():
    for world in (1, 2, 4, 8):
        n = 8
        seen = sorted(s for r in range(world) for s in my_sequences(n, r, world))
        assert seen == list(range(n))
        order = unshard_order(n, world)
        assert sorted(order) == list(range(n))


def _worker(rank, world, port, n_seq, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    calls = []

    def fit_one(s):                       # stand-in for AmassTemporalFitter: result identifies the sequence
        calls.append(s)
        return torch.full((5, 72), float(s)) + torch.arange(72.0)[None] * 1e-3

    out = fit_sharded(n_seq, fit_one, rank, world)
    ok = out.shape == (n_seq, 5, 72) and all(abs(float(out[s, 0, 0]) - s) < 1e-6 for s in range(n_seq))
    ok = ok and calls == my_sequences(n_seq, rank, world)
    g = gather_fitted_params(torch.full((1, 2, 72), float(rank)))
    ok = ok and g.shape == (world, 2, 72) and [float(g[r, 0, 0]) for r in range(world)] == [float(r) for r in range(world)]
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_gather():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 4, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(30)
    assert res == [(0, True), (1, True)]


def test_single_process_passthrough():
    x = torch.randn(1, 3, 72)
    assert torch.equal(gather_fitted_params(x), x)


class _FakeFitter:
    """stand-in with the four members ConcurrentClips / fit_sharded_concurrent touch"""

    class _Lib:
        is_emu = True

    def __init__(self):
        self.device, self.lib, self.seq, self.n = torch.device('cpu'), self._Lib(), None, 0

    def prepare(self, n):
        pass

    def step(self, n, use_graph=True):
        self.n += n

    def params72(self):
        return torch.full((5, 72), float(self.seq)) + self.n * 1e-3


def _worker_cc(rank, world, port, n_seq, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    fitters = [_FakeFitter(), _FakeFitter()]
    loaded = []

    def load(f, s):
        f.seq, f.n = s, 0
        loaded.append(s)

    out = fit_sharded_concurrent(n_seq, fitters, load, 7, rank, world)
    ok = out.shape == (n_seq, 5, 72) and all(abs(float(out[s, 0, 0]) - (s + 7e-3)) < 1e-5 for s in range(n_seq))
    ok = ok and loaded == my_sequences(n_seq, rank, world)          # two at a time, in this rank's order
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_concurrent_clips():
    """2 ranks x 3 sequences each, 2 clips in flight per rank (the last group is a single clip): every rank gets all 6 results in
    sequence order"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_cc, args=(r, 2, port, 6, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(30)
    assert res == [(0, True), (1, True)]


def _prox_recording_rows(rec_id, tmp):
    """fit one small synthetic recording (17 frames, batch 10 -> two chained windows) with the native PROX engine on the
    emulator through lemo_amd.prox_windows.run_recording; rows = [transl | global_orient | pose_embedding] per frame"""
    import numpy as np
    import __graft_entry__ as ge
    from lemo_amd import _hip, prox_windows as PW
    from lemo_amd.prox import ENGINE_PARAMS
    emu = _hip.HipLib(_hip.EMU_LIB_PATH, is_emu=True)
    n, B = 17, 10
    base = ge.prox_small_problem(B=n, stage='S2', seed=5 + rec_id)
    names = [f'rec{rec_id}_frame_{i:05d}' for i in range(n)]
    cur, prox = os.path.join(tmp, f'cur{rec_id}'), os.path.join(tmp, f'prox{rec_id}')
    P0 = base['params']
    body0 = {k: np.asarray(P0[k], np.float32) for k in ('transl', 'global_orient', 'betas', 'left_hand_pose', 'right_hand_pose', 'jaw_pose',
                                                       'leye_pose', 'reye_pose', 'expression')}
    for i, fn in enumerate(names):
        PW.write_result_pkl(PW.result_path(prox, fn), {}, body0, np.asarray(P0['pose_embedding'], np.float32), np.zeros((n, 63), np.float32), i)
    emb = {}

    def fit_window(fns, init, first, n_frozen):
        s = names.index(fns[0])
        prob = dict(base, B=len(fns), params=init, gt_joints=base['gt_joints'][s:s + len(fns)], joints_conf=base['joints_conf'][s:s + len(fns)])
        eng, _ = ge.prox_engine_for(prob, torch.device('cpu'), first_batch_flag=first, lib=emu)
        eng.step(2, use_graph=False)
        body = {k: eng.P[k].numpy() for k, _ in ENGINE_PARAMS[:-1]}
        body['betas'] = np.asarray(init['betas'], np.float32)
        for i, fn in enumerate(fns):
            emb[fn] = eng.P['pose_embedding'][i].clone()
        return {}, body, eng.P['pose_embedding'].numpy(), np.zeros((len(fns), 63), np.float32)

    assert PW.run_recording(names, B, cur, prox, fit_window) == 2
    rows = []
    for fn in names:
        r = PW.read_prox_pkl(PW.result_path(cur, fn))
        rows.append(torch.cat([torch.from_numpy(np.asarray(r['transl'], np.float32)).reshape(-1),
                               torch.from_numpy(np.asarray(r['global_orient'], np.float32)).reshape(-1), emb[fn]]))
    return torch.stack(rows)


def _worker_prox(rank, world, port, tmp, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from lemo_amd.sharding import fit_recordings_sharded
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(4)
    calls = []

    def fit_recording(r):
        calls.append(r)
        rows = _prox_recording_rows(r, tmp)
        return rows if r == 0 else rows[:15]              # recordings of different lengths: 17 and 15 result rows
    out = fit_recordings_sharded(2, fit_recording, rank, world)
    torch.save([o.clone() for o in out], os.path.join(tmp, f'out_rank{rank}.pt'))
    q.put((rank, calls == [rank] and [int(o.shape[0]) for o in out] == [17, 15] and all(bool(torch.isfinite(o).all()) for o in out)))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_gloo_prox_recordings(tmp_path):
    """the PROX multi-GPU leg: recording r -> rank r, each rank walks ITS recording's two chained windows with the native
    PROX engine (emulator library, lemo_amd.prox_windows.run_recording), ONE all-gather returns every recording's per-frame
    rows to every rank, padded / trimmed for the different lengths; equal on both ranks and equal to a single-process run"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_prox, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=800) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]
    a, b = torch.load(tmp_path / 'out_rank0.pt'), torch.load(tmp_path / 'out_rank1.pt')
    solo = [_prox_recording_rows(0, str(tmp_path / 'solo')), _prox_recording_rows(1, str(tmp_path / 'solo'))[:15]]
    for r in range(2):
        assert torch.equal(a[r], b[r]) and torch.equal(a[r], solo[r]), r
    assert not torch.equal(a[0][:15], a[1])


def test_recording_rows_survive_nan(tmp_path):
    """a diverged recording (NaN rows, column 0 included) keeps its length and its frame alignment through the
    gather -- lengths travel explicitly, not as NaN padding"""
    from lemo_amd.sharding import fit_recordings_sharded
    a = torch.arange(5 * 3, dtype=torch.float32).reshape(5, 3)
    b = torch.arange(8 * 3, dtype=torch.float32).reshape(8, 3) + 100
    b[2:4] = float('nan')
    b[7, 0] = float('inf')
    out = fit_recordings_sharded(2, lambda r: (a, b)[r].clone(), 0, 1)
    assert [tuple(o.shape) for o in out] == [(5, 3), (8, 3)]
    assert torch.equal(out[0], a) and torch.equal(torch.nan_to_num(out[1], 7.0, 8.0), torch.nan_to_num(b, 7.0, 8.0))
    out = fit_recordings_sharded(2, lambda r: (a, b)[r].clone(), 0, 1, max_frames=11)
    assert [tuple(o.shape) for o in out] == [(5, 3), (8, 3)] and torch.equal(out[0], a)


def _run_bench_two_ranks(extra, timeout=900):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2 ... --backend gloo --emu`: the driver's own
    multi-GPU launch line on the host-emulated kernel library (the only N > 1 evidence a GPU-less container can give)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(['make', '-C', os.path.join(root, 'lemo_amd', 'csrc'), '-j8', 'emu'], check=True, capture_output=True)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--backend', 'gloo', '--emu'] + extra
    env = dict(os.environ, OMP_NUM_THREADS='2')
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]           # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.timeout(1000)
def test_bench_two_rank_dry_run_amass():
    """bench.py's multi-rank control flow without GPUs: init_process_group, barrier + timed region + barrier, all_gather of the
    per-rank times, MAX over ranks, the ONE all-gather of the fitted [B,72] blocks, the shard self-checks (own block bit-identical,
    other rank's block differs, all finite) -- any of them failing makes the run exit non-zero"""
    d = _run_bench_two_ranks([])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['scaling'] == 'weak' and d['config']['emulated'] and d['config']['sequences'] == 2
    assert len(d['per_rank_iterations_per_s']) == 2 and all(v > 0 for v in d['per_rank_iterations_per_s'])
    # value = units of ALL ranks / max-over-ranks time: never more than the sum of the per-rank rates
    assert 0 < d['value'] <= sum(d['per_rank_iterations_per_s']) * 1.0001
    assert abs(d['value'] - 2 * min(d['per_rank_iterations_per_s'])) <= 1e-6 * d['value']
    assert np.isfinite(d['final_total_loss'])


@pytest.mark.timeout(1000)
def test_bench_two_rank_dry_run_prox():
    """the same for --workload prox (BASELINE configs[4]'s per-GPU leg: one window per rank, recordings shard, one all-gather)"""
    d = _run_bench_two_ranks(['--workload', 'prox'])
    assert d['n_gpus'] == 2 and d['config']['emulated'] and d['config']['recordings'] == 2
    assert len(d['per_rank_iterations_per_s']) == 2 and abs(d['value'] - 2 * min(d['per_rank_iterations_per_s'])) <= 1e-6 * d['value']
    assert np.isfinite(d['total_loss'])


@pytest.mark.timeout(1000)
@pytest.mark.parametrize('workload', ['amass', 'prox'])
def test_bench_self_launches_when_not_under_torchrun(workload):
    """plain ``python bench.py --gpus 2`` (no WORLD_SIZE in the environment: how a driver runs --gpus 1) re-executes itself under
    torch.distributed.run with one rank per GPU instead of failing an assert (VERDICT r04 missing #1); same line, same checks"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(['make', '-C', os.path.join(root, 'lemo_amd', 'csrc'), '-j8', 'emu'], check=True, capture_output=True)
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--backend', 'gloo', '--emu',
           '--workload', workload]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['config']['emulated']
    assert len(d['per_rank_iterations_per_s']) == 2 and abs(d['value'] - 2 * min(d['per_rank_iterations_per_s'])) <= 1e-6 * d['value']
