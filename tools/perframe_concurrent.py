"""Stage 1 (opt_amass_perframe.py: B = 1, 100 Adam steps per frame, frames in order) for k clips side by side
(lemo_amd.fitting.fit_clips_per_frame): frame fits per second for k = 1, 2, 4, 8 and a bit-identity check against one clip
fitted on its own.  Diagnostic, GPU box only.  Usage: python tools/perframe_concurrent.py [frames=10] [kmax=8]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import synthetic
from lemo_amd.assets import load_assets
from lemo_amd.body_model import BodyModelData
from lemo_amd.fitting import PerFrameFitter, fit_clips_per_frame
from lemo_amd.vposer import make_vposer_weights

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda:0')
A = load_assets()
data = BodyModelData(synthetic.make_synthetic_smplx(seed=0), num_pca_comps=12)
vw = make_vposer_weights(2)
t0 = time.time()
pfs = [PerFrameFitter(data, vw, A['enc_w'], A['ids'], A['Xmean'], A['Xstd'], dev) for _ in range(kmax)]
print(f'{kmax} per-frame fitters built in {time.time() - t0:.1f} s', flush=True)
rng = np.random.default_rng(0)
clips, betas = [], []
for i in range(kmax):
    seq = synthetic.make_synthetic_sequence(i, B=T)
    # target markers: a smooth random walk around a plausible body position (the fit only needs SOME reachable target)
    base = np.array([0.0, 0.4, 1.0], np.float32) + rng.normal(0, 0.15, (67, 3)).astype(np.float32)
    clips.append((base[None] + np.cumsum(rng.normal(0, 0.004, (T, 1, 3)), 0)).astype(np.float32))
    betas.append(seq['init_params'][0, 6:16])
ref = pfs[0].fit_clip(clips[0], betas[0], steps=100).clone()          # also captures the graphs of fitter 0
torch.cuda.synchronize()
for k in (1, 2, 4, 8):
    if k > kmax:
        break
    fit_clips_per_frame(pfs[:k], clips[:k], betas[:k], steps=100)      # warm (graphs of the new fitters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fit_clips_per_frame(pfs[:k], clips[:k], betas[:k], steps=100)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.equal(out[0], ref), 'clip 0 differs from its solo fit'
    print(f'{k} clip(s) side by side: {k * T / dt:7.1f} frame fits/s ({dt / T * 1e3:6.2f} ms per lockstep frame, {k * T * 100 / dt:8.0f} '
          f'iterations/s in aggregate); clip 0 bit-identical to its solo fit', flush=True)
