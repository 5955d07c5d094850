"""k independent clips per GPU, one engine + one stream each, replaying their graphs side by side (diagnostic, GPU box only).

One clip's iteration is a chain of 33 dependent kernels: ~25 % of it is kernel-boundary latency and a third of its launches
(the per-frame kernels, 119 workgroups) leave more than half of the CUs idle.  A second, independent clip on another stream
fills those holes: this script measures the aggregate fitting-iterations/s for k = 1, 2, 3, 4 clips fitted concurrently
(each clip still runs the BASELINE configs[1] iteration: B = 119, V = 10475, all vertices forwarded).
Usage: python tools/concurrent_clips.py [steps=100] [kmax=4]
"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
fits, probs, streams = [], [], []
for k in range(kmax):
    f, p = bench.build_problem(k, 119, dev, full_vertices=True, conv_variant=bench.DEFAULT_CONV_VARIANT)
    fits.append(f); probs.append(p); streams.append(torch.cuda.Stream(dev))
for f, s in zip(fits, streams):
    with torch.cuda.stream(s):
        f.prepare(steps)
        f.prepare(20)
bench.clock_ramp(fits[0], streams[0], 250.0, True)
res = {}
load = lambda f, p: f.load_sequence(p['seq']['init_params'], p['markers'], p['seq']['contact_lbl'])
# every clip on its own first: the values the side-by-side runs must reproduce bit for bit
solo = []
for f, p, s in zip(fits, probs, streams):
    load(f, p)
    with torch.cuda.stream(s):
        f.step(10); f.step(steps)
    torch.cuda.synchronize(dev)
    solo.append((f.losses(), f.params75().clone()))
print('solo final total losses:', [repr(l['total']) for l, _ in solo], flush=True)
bad = 0
for k in range(1, kmax + 1):
    best = 0.0
    for rep in range(3):
        for f, p in zip(fits[:k], probs[:k]):
            load(f, p)                       # default stream; the engines order their launches behind it (own events)
        for f, s in zip(fits[:k], streams[:k]):
            with torch.cuda.stream(s):
                f.step(10)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for f, s in zip(fits[:k], streams[:k]):
            with torch.cuda.stream(s):
                f.step(steps)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        best = max(best, k * steps / dt)
        for i, f in enumerate(fits[:k]):
            same = f.losses() == solo[i][0] and torch.equal(f.params75(), solo[i][1])
            if not same:
                bad += 1
                print(f'MISMATCH k={k} rep={rep} clip={i}: {f.losses()["total"]!r} vs solo {solo[i][0]["total"]!r}', flush=True)
    losses = [f.losses()['total'] for f in fits[:k]]
    assert all(f.nonfinite_step() == 0 for f in fits[:k])
    res[k] = best
    print(f'{k} clip(s) side by side: {best:8.1f} fitting-iterations/s aggregate ({best / k:7.1f} per clip), '
          f'{1e3 * k / best:.3f} ms per iteration-of-any-clip; final total losses {[repr(l) for l in losses]} '
          f'(== solo, every repetition: {bad == 0})', flush=True)
print(json.dumps({'steps': steps, 'aggregate_iterations_per_s': res, 'bit_identical_to_solo': bad == 0}))
sys.exit(1 if bad else 0)
