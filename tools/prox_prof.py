"""rocprof target: 12 eager PROX S3 iterations at B=100, V=10475, 256^3 SDF (diagnostic)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from lemo_amd import synthetic
from lemo_amd.assets import load_assets
from lemo_amd.prox import S3_WEIGHTS, load_prox_tables
dev = torch.device('cuda:0')
A = load_assets(); B = 100
small = ge.prox_small_problem(B=B, stage='S3')
rng = np.random.default_rng(3); D = 256
zz = np.linspace(-3, 6, D, dtype=np.float32)
prob = dict(small, model=synthetic.make_synthetic_smplx(seed=0), V=10475, ids=A['ids'], Xmean=A['Xmean'], Xstd=A['Xstd'],
            fric_ids=load_prox_tables()['contact_fric_verts_ids'],
            sdf=(np.broadcast_to(zz[None, None, :], (D, D, D)) - 1.40).astype(np.float32).copy(), weights=S3_WEIGHTS)
mask = np.ones((B, 67), np.float32); mask[40:60, :22] = 0
prob['infill'] = dict(marker_mask=mask, body_markers_rec=(rng.standard_normal((B - 1, 67, 3)) * 0.3).astype(np.float32),
                      contact_lbl_rec=(rng.random((B - 1, 4)) < 0.7).astype(np.float32))
fit, bm = ge.prox_fitter_for(prob, dev, first_batch_flag=False)
fit.step(12, use_graph=False)
torch.cuda.synchronize()
