"""rocprof target: stage 1 (per-frame fits, B = 1) of one clip -- 10 frames x 100 Adam steps on PerFrameFitter (diagnostic).
Which launches make up the 71 us of a B = 1 iteration?  (rocprofv3 --kernel-trace --stats --output-format csv, WITH a timeout)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import synthetic
from lemo_amd.assets import load_assets
from lemo_amd.body_model import BodyModelData
from lemo_amd.fitting import PerFrameFitter
from lemo_amd.vposer import make_vposer_weights
dev = torch.device('cuda:0')
A = load_assets()
pf = PerFrameFitter(BodyModelData(synthetic.make_synthetic_smplx(seed=0), num_pca_comps=12), make_vposer_weights(2), A['enc_w'], A['ids'],
                    A['Xmean'], A['Xstd'], dev)
rng = np.random.default_rng(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
base = np.array([0.0, 0.4, 1.0], np.float32) + rng.normal(0, 0.15, (67, 3)).astype(np.float32)
clip = (base[None] + np.cumsum(rng.normal(0, 0.004, (T, 1, 3)), 0)).astype(np.float32)
betas = synthetic.make_synthetic_sequence(0, B=T)['init_params'][0, 6:16]
use_graph = os.environ.get('PERFRAME_EAGER') is None
pf.fit_clip(clip, betas, steps=100, use_graph=use_graph)
torch.cuda.synchronize()
