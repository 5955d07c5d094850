"""Stage 1 (per-frame fit, BASELINE configs[0]) for N clips in lockstep through ONE engine (lemo_amd.fitting.BatchedPerFrameFitter):
frame fits per second for N = 1, 8, 32, 64, 119 and the bit-identity of every clip with its solo fit (diagnostic, GPU box only).
Usage: python tools/perframe_batched.py [frames=6] [steps=100]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import synthetic
from lemo_amd.assets import load_assets
from lemo_amd.fitting import PerFrameFitter, BatchedPerFrameFitter
from lemo_amd.vposer import make_vposer_weights

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device('cuda:0')
A = load_assets()
model = synthetic.make_synthetic_smplx(seed=0)
vw = make_vposer_weights(2)
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'amass_iter.npz'))
base = g['markers_rec']                                   # [119,67,3] markers of one synthetic clip
rng = np.random.default_rng(0)


def clip(i):
    """clip i: a window of the base clip, shifted and slightly scaled (distinct targets per clip)"""
    o = (7 * i) % (119 - frames)
    return (base[o:o + frames] * (1.0 + 0.002 * (i % 5)) + np.float32(0.01 * (i % 3))).astype(np.float32)


betas_of = lambda i: (synthetic.make_synthetic_sequence(i % 8, B=119)['init_params'][0, 6:16]).astype(np.float32)
args = (model, vw, A['enc_w'], A['ids'], A['Xmean'], A['Xstd'], dev)
pf = PerFrameFitter(*args)
solo = {}
for i in (0, 1, 7):
    solo[i] = pf.fit_clip(clip(i), betas_of(i), steps=steps).clone()
torch.cuda.synchronize()
t0 = time.perf_counter(); pf.fit_clip(clip(0), betas_of(0), steps=steps); torch.cuda.synchronize()
print(f'solo (B = 1 engine): {frames / (time.perf_counter() - t0):8.1f} frame fits/s', flush=True)
ok = True
for N in (1, 8, 32, 64, 119):
    bf = BatchedPerFrameFitter(*args, batch=N)
    clips, betas = [clip(i) for i in range(N)], [betas_of(i) for i in range(N)]
    got = bf.fit_clips(clips, betas, steps=steps)         # also captures the graphs
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(2):
        t0 = time.perf_counter()
        got = bf.fit_clips(clips, betas, steps=steps)
        torch.cuda.synchronize()
        best = max(best, N * frames / (time.perf_counter() - t0))
    same = all(torch.equal(got[i], solo[i]) for i in solo if i < N)
    ok = ok and same
    print(f'{N:4d} clips in lockstep: {best:9.1f} frame fits/s ({1e6 / best * N / steps:7.1f} us per iteration of the batch); '
          f'clips {[i for i in solo if i < N]} bit-identical to their solo fits: {same}', flush=True)
sys.exit(0 if ok else 1)
