"""persistent 7-layer encoder chain vs 7 launches of the per-layer split-bf16 kernel: bit-exactness (forward epilogue and
backward-data epilogue with ping-pong buffers) and wall time.  Diagnostic, GPU box only."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import _hip
from lemo_amd._hip import ptr
from lemo_amd.assets import load_assets
from lemo_amd.priors import EncWeights, cg8p_alloc, to_cg8p

lib = _hip.get_lib(); dev = torch.device('cuda:0')
H, W = 245, 134
enc = EncWeights(load_assets()['enc_w'], dev)
s = torch.cuda.current_stream(dev).cuda_stream
print('chain supported on this device:', lib.conv3x3_split_chain_supported(H, W))
g = torch.Generator().manual_seed(1)
x = to_cg8p(torch.randn(64, H, W, generator=g).abs() * 0.3).to(dev)
nsync = lib.conv3x3_split_chain_sync_ints(H, W, 7)

def run(epi, chain, reps=1):
    """forward: act[3] -> ... -> act[10] (distinct buffers); backward: two ping-pong buffers, aux = saved activations"""
    layers = list(range(3, 10)) if epi == 0 else list(range(9, 2, -1))
    if epi == 0:
        bufs = [x.clone()] + [cg8p_alloc(64, H, W, dev) for _ in range(7)]
        io = [(bufs[i], bufs[i + 1]) for i in range(7)]
    else:
        pp = [x.clone(), cg8p_alloc(64, H, W, dev)]
        io = [(pp[i & 1], pp[1 - (i & 1)]) for i in range(7)]
        aux = [to_cg8p(torch.randn(64, H, W, generator=g)).to(dev) for _ in range(7)]
    c = _hip.ConvChain(); c.n = 7
    for i, l in enumerate(layers):
        c.inp[i], c.out[i] = ptr(io[i][0]), ptr(io[i][1])
        c.w3[i], c.wt[i] = (ptr(enc.w3[l]), ptr(enc.w[l])) if epi == 0 else (ptr(enc.wbwd3[l]), ptr(enc.wbwd[l]))
        if epi == 0: c.bias[i] = ptr(enc.b[l])
        else: c.aux[i] = ptr(aux[i])
    sync = torch.zeros(nsync, dtype=torch.int32, device=dev)
    def once():
        if chain:
            lib.check(lib.conv3x3_split_chain(C.byref(c), H, W, epi, ptr(sync), s), 'chain')
        else:
            for i in range(7):
                lib.check(lib.conv3x3_mfma_split(c.inp[i], c.w3[i], c.wt[i], c.bias[i], c.aux[i], c.out[i], H, W, 64, 64, epi, s))
    src = io[0][0].clone()
    once(); torch.cuda.synchronize()
    res = io[-1][1].clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record(); torch.cuda.synchronize()
    return res, e0.elapsed_time(e1) * 1e3 / max(reps, 1) / 7, int(sync[1].item()), (aux if epi else None)

torch.manual_seed(0)
for epi, name in ((0, 'forward (bias + LeakyReLU)'),):
    g = torch.Generator().manual_seed(1)
    ref, t_ref, _, _ = run(epi, False, reps=20)
    g = torch.Generator().manual_seed(1)
    got, t_ch, err, _ = run(epi, True, reps=20)
    print('%s: chain == 7 launches bit for bit: %s ; timeout flag %d ; per layer %.2f us (launches) vs %.2f us (chain)' % (
        name, bool(torch.equal(ref, got)), err, t_ref, t_ch))
    if not torch.equal(ref, got):
        d = (ref - got).abs(); print('   max diff', float(d.max()), 'mismatching', int((d > 0).sum()), 'of', d.numel())
# backward: the repeated launches of the timing loop overwrite their own input, so compare a single pass separately
for chain in (False, True):
    g = torch.Generator().manual_seed(2)
    res, t, err, _ = run(1, chain, reps=0)
    if not chain: refb = res
    else: print('backward-data (x lrelu-prime(aux), ping-pong buffers): chain == 7 launches bit for bit: %s ; timeout flag %d' % (bool(torch.equal(refb, res)), err))
