"""conv variants 3 (split-bf16) and 4 (split-f16) on the GPU: accuracy against float64 next to the fp32-MFMA kernel, wall time
of both (HIP events around 20 launches) and the per-wave census of the split kernel.  Diagnostic, GPU box only."""
import os, sys, collections
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import _hip
from lemo_amd._hip import ptr
from lemo_amd.assets import load_assets
from lemo_amd.priors import (cg8p_alloc, to_cg8p, from_cg8p, pack_conv3x3, pack_conv3x3_gmajor, pack_conv3x3_split,
                             pack_conv3x3_bwd, pack_conv3x3_bwd_split, pack_conv3x3_split_f16)

lib = _hip.get_lib(); dev = torch.device('cuda:0')
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (245, 134)
enc = load_assets()['enc_w']
w = np.asarray(enc['enc_blc3.main.0.weight'], np.float32); bnp = np.asarray(enc['enc_blc3.main.0.bias'], np.float32)
assert w.shape == (64, 64, 3, 3)
g = torch.Generator().manual_seed(0)
x = torch.randn(64, H, W, generator=g).abs() * 0.3
ref64 = F.leaky_relu(F.conv2d(x[None].double(), torch.from_numpy(w).double(), torch.from_numpy(bnp).double(), padding=1), 0.2)[0]
t = lambda a: torch.from_numpy(a).to(dev)
wt, wt2, w3 = t(pack_conv3x3(w)), t(pack_conv3x3_gmajor(w)), t(pack_conv3x3_split(w).view(np.int16))
_p4, w4inv = pack_conv3x3_split_f16(w); w4 = t(_p4.view(np.int16))
b = t(bnp); xin = to_cg8p(x).to(dev); s = torch.cuda.current_stream(dev).cuda_stream
res = {}
for name in ('fp32-mfma (variant 2)', 'split-bf16 (variant 3)', 'split-f16 (variant 4)'):
    out = cg8p_alloc(64, H, W, dev)
    def run():
        if name.startswith('fp32'):
            lib.check(lib.conv3x3_mfma_lds(ptr(xin), ptr(wt), ptr(wt2), ptr(b), None, ptr(out), H, W, 64, 64, 0, s))
        elif name.startswith('split-bf16'):
            lib.check(lib.conv3x3_mfma_split(ptr(xin), ptr(w3), ptr(wt), ptr(b), None, ptr(out), H, W, 64, 64, 0, s))
        else:
            lib.check(lib.conv3x3_mfma_split_f16(ptr(xin), ptr(w4), w4inv, ptr(wt), ptr(b), None, ptr(out), H, W, 64, 64, 0, s))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    got = from_cg8p(out.cpu(), H, W).double()
    err = (got - ref64).abs()
    res[name] = got
    print('%-24s %.2f us/launch   max err / max|ref| %.3e   rms err / rms ref %.3e   mean signed %.2e' % (
        name, e0.elapsed_time(e1) * 1e3 / 20, err.max() / ref64.abs().max(), (err.pow(2).mean().sqrt() / ref64.pow(2).mean().sqrt()),
        (got - ref64).mean() / ref64.abs().mean()))
    assert float(out.reshape(8, H + 2, W + 2, 8)[:, 0].abs().max()) == 0.0
d = (res['fp32-mfma (variant 2)'] - res['split-bf16 (variant 3)']).abs()
print('variant 2 vs variant 3: max |diff| / max|ref| %.3e' % (d.max() / ref64.abs().max()))
# backward-data epilogue
dy, aux = torch.randn(64, H, W, generator=g), torch.randn(64, H, W, generator=g)
wd = torch.from_numpy(w).double()
refdx = F.conv_transpose2d(dy[None].double(), wd, padding=1)[0] * torch.where(aux > 0, 1.0, 0.2).double()
dyb, auxb, dxb = to_cg8p(dy).to(dev), to_cg8p(aux).to(dev), cg8p_alloc(64, H, W, dev)
wb3, wtb = t(pack_conv3x3_bwd_split(w).view(np.int16)), t(pack_conv3x3_bwd(w))      # keep alive: ptr() does not
lib.check(lib.conv3x3_mfma_split(ptr(dyb), ptr(wb3), ptr(wtb), None, ptr(auxb), ptr(dxb), H, W, 64, 64, 1, s))
torch.cuda.synchronize()
e = (from_cg8p(dxb.cpu(), H, W).double() - refdx).abs().max() / refdx.abs().max()
print('backward-data (epi 1): max err / max|ref| %.3e' % e)

# census of both split variants
nblk = H * W // 128
for pieces, pack, winv in ((3, w3, 1.0), (2, w4, w4inv)):
    dbg = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device=dev)
    out = cg8p_alloc(64, H, W, dev)
    for it in range(3):
        dbg.zero_()
        lib.check(lib.conv3x3_mfma_split_census2(ptr(xin), ptr(pack), winv, pieces, ptr(wt), ptr(b), ptr(out), H, W, 64, 64, ptr(dbg), s))
        torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(nblk, 8, 8)
    hw, xcc, t0, t1, tp, tl = d[..., 0], d[..., 1] & 0xf, d[..., 2], d[..., 3], d[..., 4], d[..., 5]
    tm0, tm1 = d[..., 6], d[..., 7]
    print('pieces %d: per-wave cycles median %d max %d; breakdown (median ticks): prologue %d  loop %d [chunk0 %d, barrier %d, chunk1+barrier %d]  epilogue %d' % (
        pieces, np.median(t1 - t0), (t1 - t0).max(), np.median(tp - t0), np.median(tl - tp), np.median(tm0 - tp), np.median(tm1 - tm0), np.median(tl - tm1), np.median(t1 - tl)))
