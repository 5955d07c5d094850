"""How far do 5 Adam steps of the small problem drift from the CPU oracle, per kernel family?  (diagnostic, GPU box)
Adam's update lr * m / (sqrt(v) + eps) turns rounding noise on entries with |g| ~ eps into O(lr) differences, so
this is a noise-amplification measurement, not an accuracy one: all families should show the same order."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
from lemo_amd import _hip
from lemo_amd.fitting import AmassTemporalFitter

dev = torch.device('cuda:0'); lib = _hip.get_lib()
prob = G.small_problem()
ofit, markers = G.oracle_for(prob)
res = {}
for conv, lbs in ((2, 0), (3, 0), (3, 1), (2, 1)):
    fit = AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'],
                              prob['B'], dev, full_vertices=True, conv_variant=conv, lbs_blend_fp32=not lbs)
    fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    fit.step(5, use_graph=False)
    torch.cuda.synchronize()
    res[(conv, lbs)] = fit.params75().cpu()
o, _ = G.oracle_for(prob)
for _ in range(5):
    o.step()
po = o.params75()
for k, v in res.items():
    d = (v - po).abs()
    print('conv variant %d, lbs variant %d vs oracle: max %.2e mean %.2e   (vs fp32 kernels: max %.2e mean %.2e)' % (
        k[0], k[1], float(d.max()), float(d.mean()), float((v - res[(2, 0)]).abs().max()), float((v - res[(2, 0)]).abs().mean())))
