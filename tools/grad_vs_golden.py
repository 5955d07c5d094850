"""iteration-0 gradients of the engine at B = 119 / V = 10475 against golden (6) (the fp32 CPU oracle): where do they differ?
(diagnostic; run with LEMO_HIP_LIB=... to compare library builds)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import synthetic
from lemo_amd.assets import load_assets
from lemo_amd.fitting import AmassTemporalFitter
from lemo_amd.vposer import make_vposer_weights
dev = torch.device('cuda:0')
A = load_assets()
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'amass_iter.npz'))
seq = synthetic.make_synthetic_sequence(0, B=119)
for cv in (3, 2):
    fit = AmassTemporalFitter(synthetic.make_synthetic_smplx(seed=0), make_vposer_weights(2), A['enc_w'], A['ids'], A['Xmean'], A['Xstd'], 119, dev,
                              full_vertices=True, conv_variant=cv)
    fit.load_sequence(seq['init_params'], g['markers_rec'], seq['contact_lbl'])
    fit.forward(); fit.backward(); torch.cuda.synchronize()
    gr = fit.grads_with_priors()
    for k in ('transl', 'rot6d', 'other'):
        a, b = gr[k].cpu().double(), torch.from_numpy(g['g_' + k]).double()
        d = (a - b).abs()
        fr = d.max(1).values
        top = torch.topk(fr, 5)
        print(f'conv_variant {cv} grad {k:7s}: max |diff| / max|ref| {float(d.max() / b.abs().max()):.2e}; worst frames {top.indices.tolist()} '
              f'({[f"{v:.1e}" for v in (top.values / b.abs().max()).tolist()]}); median frame error {float(fr.median() / b.abs().max()):.1e}')
