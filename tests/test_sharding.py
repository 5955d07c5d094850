"""world_size-2 gloo test of the multi-GPU path (sequence sharding + ONE all-gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lemo_amd.sharding import fit_sharded, gather_fitted_params, my_sequences, unshard_order


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
