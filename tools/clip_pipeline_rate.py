"""End-to-end per-clip rate of the AMASS stage-2 pipeline (opt_amass_temp.py:159-458: mask -> 60-step AE finetune -> decode -> 100-step
temporal fit) through lemo_amd.pipeline.AmassClipPipeline, N clips back to back on one GPU without a host synchronisation between
them -- fit_clip in a loop (host-bound before round 3 cached its per-clip uploads) and fit_clips (infill.AE_CLIPS clips per finetune
launch on the caller's stream, then their fits on the fitter's stream while the next group's finetune is enqueued).
Diagnostic, GPU box only.  Usage: python tools/clip_pipeline_rate.py [n_clips=8]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import pipeline as P, synthetic
from lemo_amd.assets import load_assets
from lemo_amd.fitting import AmassTemporalFitter
from lemo_amd.infill import AE
from lemo_amd.vposer import make_vposer_weights
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
A = load_assets()
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'amass_clip.npz'))
model = synthetic.make_synthetic_smplx(seed=0)
vw = make_vposer_weights(2)
ae_w = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_ae_weights(7).items()}
init = synthetic.make_synthetic_sequence(0, B=119)['init_params'].copy()
init[:, 0:3] = g['markers_rec'].mean(1) - np.array([0, 0, 0.2], np.float32)
clip = torch.from_numpy(g['clip_img']).to(dev)
piv = torch.from_numpy(g['rot_0_pivot']).to(dev)
fits = [AmassTemporalFitter(model, vw, A['enc_w'], A['ids'], A['Xmean'], A['Xstd'], 119, dev) for _ in range(2)]
pipes = [P.AmassClipPipeline(f, AE().to(dev), ae_w) for f in fits]
side = torch.cuda.Stream(dev)
torch.cuda.synchronize()
with torch.cuda.stream(side):
    for label, use in (('one fitter', pipes[:1]), ('two fitters in turn', pipes)):
        for p in use:                                             # graphs captured, sessions created
            p.fit_clip(clip, piv, init, gender=1, steps=100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [use[i % len(use)].fit_clip(clip, piv, init, gender=1, steps=100) for i in range(n)]
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'{label}: {n} clips in {dt * 1e3:.1f} ms = {dt / n * 1e3:.1f} ms per clip ({n / dt:.1f} clips/s); host time {th / n * 1e3:.1f} ms per clip', flush=True)
    for label, use in (('fit_clips, one fitter', pipes[0]),):
        items = [(clip, piv, init, 1)] * n
        use.fit_clips(items, steps=100); torch.cuda.synchronize()        # (engines of the group sizes created, graphs captured)
        t0 = time.perf_counter()
        outs = use.fit_clips(items, steps=100)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'{label}: {n} clips in {dt * 1e3:.1f} ms = {dt / n * 1e3:.1f} ms per clip ({n / dt:.1f} clips/s); host time {th / n * 1e3:.1f} ms per clip', flush=True)
    # the parts on their own, synchronised after each
    t0 = time.perf_counter(); o = pipes[0].fit_clip(clip, piv, init, gender=1, steps=100); torch.cuda.synchronize()
    print(f'one clip, synchronised: {(time.perf_counter() - t0) * 1e3:.1f} ms', flush=True)
