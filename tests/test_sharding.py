"""world_size-2 gloo test of the multi-GPU path (sequence sharding + ONE all-gather)."""
import os
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
