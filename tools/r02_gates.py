"""How large is fp32 noise really?  GPU engine and fp32 CPU oracle, both against the float64 oracle: gradients of
iteration 0 and the Adam trajectory over 10 steps, with and without the contact term (threshold flips), small problem
and BASELINE size.  (diagnostic, GPU box; the numbers set the tolerances of tests/test_gpu_parity.py)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from lemo_amd import synthetic
from lemo_amd.assets import load_assets
from lemo_amd.fitting import AmassTemporalFitter
from lemo_amd.vposer import make_vposer_weights
from oracle import lemo_oracle as O
from oracle.f64 import amass_fit_oracle_f64, default_f64

dev = torch.device('cuda:0')
torch.set_num_threads(32)


def run(name, prob, markers, weights, steps=10):
    ej = list(range(21)) if prob['V'] < 9930 else None
    o32, _ = ge.oracle_for(prob, weights=weights) if prob['V'] < 9930 else (None, None)
    so = O.SmplxOracle(prob['model'], extra_joint_ids=ej)
    vw = {k: torch.from_numpy(v) for k, v in prob['vposer_w'].items()}
    ew = {k: torch.from_numpy(v) for k, v in prob['enc_w'].items()}
    o32 = O.AmassFitOracle(so, vw, ew, prob['ids'], np.asarray(prob['Xmean']).reshape(1, 1, -1), prob['Xstd'], prob['seq']['init_params'],
                           markers, prob['seq']['contact_lbl'], faithful=False, weights=weights)
    o64 = amass_fit_oracle_f64(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'],
                               prob['seq']['init_params'], markers, prob['seq']['contact_lbl'], weights=weights, extra_joint_ids=ej)
    fit = AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'], prob['B'], dev,
                              weights=weights, full_vertices=True)
    fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    fit.forward(); fit.backward(); torch.cuda.synchronize()
    t32, p32, _, _ = o32.losses(); t32.backward()
    with default_f64():
        t64, p64, _, _ = o64.losses(); t64.backward()
    L = fit.losses()
    rel = lambda a, b: abs(a - b) / abs(b) if b != 0 else float('nan')      # nan: the term is switched off
    print(f'== {name}: loss scalars rel err vs f64   gpu / cpu-f32')
    for k in ('marker', 'vposer', 'shape', 'hand', 'contact', 'smooth'):
        print(f'   {k:8s} {rel(L[k], float(p64[k])):.2e} / {rel(float(p32[k]), float(p64[k])):.2e}')
    g = fit.grads_with_priors()
    for k, a32, a64 in (('transl', o32.transl, o64.transl), ('rot6d', o32.rot6d, o64.rot6d), ('other', o32.other, o64.other)):
        n = a64.grad.abs().max()
        print(f'   grad {k:7s} max-rel vs f64: gpu {float((g[k].cpu().double() - a64.grad).abs().max() / n):.2e}  cpu-f32 {float((a32.grad.double() - a64.grad).abs().max() / n):.2e}')
    o32.opt.zero_grad(); o64.opt.zero_grad()
    fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    s = torch.cuda.Stream(dev)
    for i in range(steps):
        with torch.cuda.stream(s):
            fit.step(1, use_graph=True)
        torch.cuda.synchronize()
        h32 = o32.step()
        with default_f64():
            h64 = o64.step()
        p = o64.params75()
        dg, dc = (fit.params75().cpu().double() - p).abs(), (o32.params75().double() - p).abs()
        print(f'   step {i}: params vs f64  gpu max {float(dg.max()):.2e} mean {float(dg.mean()):.2e} | cpu-f32 max {float(dc.max()):.2e} mean {float(dc.mean()):.2e}'
              f' | total rel gpu {rel(fit.losses()["total"], h64["total"]):.2e} cpu {rel(h32["total"], h64["total"]):.2e}')


small = ge.small_problem()
_, mk = ge.oracle_for(small)
run('small, all terms', small, mk, None)
run('small, contact off', small, mk, dict(O.LOSS_WEIGHTS, contact_vel=0.0))
A = load_assets()
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'amass_iter.npz'))
full = dict(model=synthetic.make_synthetic_smplx(seed=0), vposer_w=make_vposer_weights(2), enc_w=A['enc_w'], ids=A['ids'], Xmean=A['Xmean'],
            Xstd=A['Xstd'], seq=synthetic.make_synthetic_sequence(0, B=119), B=119, V=10475)
run('B=119 V=10475, all terms', full, g['markers_rec'], None, steps=6)
run('B=119 V=10475, contact off', full, g['markers_rec'], dict(O.LOSS_WEIGHTS, contact_vel=0.0), steps=6)
