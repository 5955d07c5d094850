"""Launch-shape sweep of the AE step engine's convolution (lemo_ae_conv) over the 39 convolutions of a training step at
[1,4,210,135]: us per launch (HIP events around 40 back-to-back launches) for every tile / pixel-tiles / K-slices combination,
next to the shape the engine's rule picks (mt = 0).  Diagnostic, GPU box only.  Usage: python tools/ae_conv_tune.py"""
import os, sys, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import _hip
from lemo_amd._hip import ptr
from lemo_amd.priors import cg8p_alloc
lib = _hip.get_lib(); dev = torch.device('cuda:0')
LV = [(210, 135)]
for _ in range(5):
    LV.append(((LV[-1][0] - 1) // 2 + 1, (LV[-1][1] - 1) // 2 + 1))
ENC = [(8, 32), (32, 64), (64, 128), (128, 256), (256, 256)]
DEC = [(256, 256), (256, 128), (128, 64), (64, 32), (32, 32)]
cases = []            # (name, level of the enumerated grid, cin, cout, epi, in_s)
for b, (ci, co) in enumerate(ENC):
    cases += [(f'fwd enc{b}.0', b, ci, co, 0, 1), (f'fwd enc{b}.2', b, co, co, 0, 1)]
for b, (ci, co) in enumerate(DEC):
    cases += [(f'fwd dec{b}.1', 4 - b, ci, co, 0, 1), (f'fwd dec{b}.2', 4 - b, co, co, 0, 1)]
for b, (ci, co) in enumerate(DEC):
    cases += [(f'bwd dec{b}.2', 4 - b, co, co, 1, 1), (f'bwd dec{b}.1 (even pixels)', 5 - b, co, ci, 1, 2)]
for b, (ci, co) in enumerate(ENC):
    cases += [(f'bwd enc{b}.2', b, co, co, 1, 1)] + ([(f'bwd enc{b}.0', b, co, ci, 2, 1)] if b else [])
s = torch.cuda.current_stream(dev).cuda_stream
seen = {}
tot_rule = tot_best = 0.0
for name, lv, cin, cout, epi, in_s in cases:
    H, W = LV[lv]
    fH, fW = LV[lv - 1] if in_s == 2 else (0, 0)
    key = (lv, cin, cout, epi, in_s)
    if key not in seen:
        x = cg8p_alloc(cin, fH or H, fW or W, dev); x.normal_()
        wt = torch.randn(9 * cin * cout, device=dev) * 0.05
        bias = torch.zeros(cout, device=dev)
        aux = cg8p_alloc(cout, fH or H, fW or W, dev); aux.normal_()
        out = cg8p_alloc(cout, H, W, dev)
        res = {}
        cands = [(0, 0, 0)] + [(mt, pt, ks) for mt in (1, 2, 3) for pt in (1, 2, 4) for ks in (1, 2, 4, 8, 16)]
        for mt, pt, ks in cands:
            def run():
                return lib.ae_conv(ptr(x), ptr(wt), ptr(bias), ptr(aux), ptr(out), H, W, fH, fW, in_s, 1, cin, cout, epi, mt, pt, ks, s)
            if run() != 0:
                continue
            for _ in range(5): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40): run()
            e1.record(); torch.cuda.synchronize()
            res[(mt, pt, ks)] = e0.elapsed_time(e1) * 1e3 / 40
        seen[key] = res
    res = seen[key]
    rule = res[(0, 0, 0)]
    order = sorted((v, k) for k, v in res.items() if k != (0, 0, 0))
    tot_rule += rule; tot_best += order[0][0]
    print('%-28s %3dx%-3d %3d->%-3d  rule %5.1f us | best ' % (name, H, W, cin, cout, rule) + '  '.join('%s %.1f' % (k, v) for v, k in order[:4]), flush=True)
print('sum over the 39 convolutions: rule %.0f us, best-of-sweep %.0f us' % (tot_rule, tot_best))
