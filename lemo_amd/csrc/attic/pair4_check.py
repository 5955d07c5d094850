"""conv variant 6 (four-wave fused pairs, two workgroups per CU: csrc/conv_pair4_kernels.hip) against variant 5 (one eight-wave workgroup
per CU) on the GPU -- VERDICT r04 next #1a "measured, not on paper": accuracy of both against float64 at 245 x 134, wall time of a forward
and a backward-data pair (20 launches back to back; a dependent chain of 6 like the iteration's), interleaved A/B/A/B, and the per-wave
census of the 4-wave kernel (staging | layer 1 | epilogue 1 + tile maximum | mid planes | layer 2 | exchange + epilogue 2 + stores) with
the overlap two co-resident workgroups achieve (launch span vs 2 x the median workgroup lifetime).  Diagnostic, GPU box only."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import _hip
from lemo_amd._hip import ptr
from lemo_amd.assets import load_assets
from lemo_amd.priors import EncWeights, cg8p_alloc, to_cg8p, from_cg8p

lib = _hip.get_lib(); dev = torch.device('cuda:0')
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (245, 134)
A = load_assets()
enc = EncWeights(A['enc_w'], dev)
g = torch.Generator().manual_seed(0)
x_cpu = torch.randn(64, H, W, generator=g) * 0.3
x = to_cg8p(x_cpu).to(dev)
bufs = [cg8p_alloc(64, H, W, dev) for _ in range(4)]
s = torch.cuda.current_stream(dev).cuda_stream
P = {(l, b): enc.split_pack(l, b, 5) for l in range(3, 10) for b in (False, True)}
K = {5: lib.conv3x3_pair_f16, 6: lib.conv3x3_pair4_f16}


def pair_fwd(v, src, mid, dst, l=3, dbg=None):
    (pa, ia), (pb, ib) = P[(l, False)], P[(l + 1, False)]
    lib.check(K[v](ptr(src), ptr(pa), ia, ptr(enc.b[l]), None, ptr(mid), ptr(pb), ib, ptr(enc.b[l + 1]), None, ptr(dst), H, W, 0, dbg, s))


def pair_bwd(v, src, a1, a0, dst, l=4):
    (pa, ia), (pb, ib) = P[(l, True)], P[(l - 1, True)]
    lib.check(K[v](ptr(src), ptr(pa), ia, None, ptr(a1), None, ptr(pb), ib, None, ptr(a0), ptr(dst), H, W, 1, None, s))


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


# ---- accuracy vs float64 (layers 3, 4 of the real encoder)
wname = lambda l: f'enc_blc{l // 2 + 1}.main.{(l % 2) * 2}'
w3, b3 = (torch.from_numpy(A['enc_w'][wname(3) + k]).double() for k in ('.weight', '.bias'))
w4, b4 = (torch.from_numpy(A['enc_w'][wname(4) + k]).double() for k in ('.weight', '.bias'))
a1 = F.leaky_relu(F.conv2d(x_cpu[None].double(), w3, b3, padding=1), 0.2)
a2 = F.leaky_relu(F.conv2d(a1, w4, b4, padding=1), 0.2)[0]
rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
for v in (5, 6):
    pair_fwd(v, x, bufs[0], bufs[1]); torch.cuda.synchronize()
    print('variant %d forward pair vs float64: mid %.2e  out %.2e' % (v, rel(from_cg8p(bufs[0], H, W).cpu(), a1[0]), rel(from_cg8p(bufs[1], H, W).cpu(), a2)))
d2 = to_cg8p(torch.randn(64, H, W, generator=g) * 1e-6).to(dev)
outs = {}
for v in (5, 6):
    pair_fwd(v, x, bufs[0], bufs[1]); pair_bwd(v, d2, bufs[0], x, bufs[2]); torch.cuda.synchronize()
    outs[v] = from_cg8p(bufs[2], H, W).cpu().clone()
print('backward-data pair, variant 6 vs variant 5: %.2e of max' % rel(outs[6], outs[5].double()))


def chain(v):
    pair_fwd(v, x, bufs[0], bufs[1], 3); pair_fwd(v, bufs[1], bufs[2], bufs[3], 5); pair_fwd(v, bufs[3], bufs[0], bufs[1], 7)
    pair_bwd(v, bufs[1], bufs[0], bufs[3], bufs[2], 9); pair_bwd(v, bufs[2], bufs[3], bufs[1], bufs[0], 7); pair_bwd(v, bufs[0], bufs[1], x, bufs[2], 5)


for rep in range(3):
    for v in (5, 6):
        print('rep %d variant %d: forward pair %.2f us | backward pair %.2f us | dependent chain of 6: %.2f us per launch' % (
            rep, v, timeit(lambda: pair_fwd(v, x, bufs[0], bufs[1])), timeit(lambda: pair_bwd(v, bufs[1], bufs[0], x, bufs[2])), timeit(lambda: chain(v), 10) / 6))

# ---- census of the 4-wave kernel
ntx, nty = (W + 13) // 14, (H + 4) // 5
nblk = ntx * nty
dbg = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device=dev)
for it in range(3):
    dbg.zero_()
    pair_fwd(6, x, bufs[0], bufs[1], 3, ptr(dbg)); torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nblk, 4, 8)
hw, tl2, t0, t1, tp, tl1, tmid, tmx = (d[..., i] for i in range(8))
med = lambda a: int(np.median(a))
print('%d workgroups of 4 waves; per-wave cycles median %d max %d: staging %d | layer 1 %d | epilogue 1 + max barrier %d | mid planes %d | layer 2 %d | exchange + epilogue 2 + stores %d' % (
    nblk, med(t1 - t0), (t1 - t0).max(), med(tp - t0), med(tl1 - tp), med(tmx - tl1), med(tmid - tmx), med(tl2 - tmid), med(t1 - tl2)))
slot = hw & 0xF
for sl in sorted(set(slot.flatten().tolist())):
    m = slot == sl
    print('   wave slot %d: %d waves; layer 1 median %d | layer 2 %d | lifetime %d' % (sl, int(m.sum()), med((tl1 - tp)[m]), med((tl2 - tmid)[m]), med((t1 - t0)[m])))
half = np.arange(nblk) >= (nblk + 1) // 2
for nm, m in (('first half of the grid', ~half), ('second half', half)):
    print('   %s: layer 1 median %d | layer 2 %d | lifetime %d' % (nm, med((tl1 - tp)[m]), med((tl2 - tmid)[m]), med((t1 - t0)[m])))
wg = t1.max(1) - t0.min(1)
print('per-workgroup lifetime median %d max %d cycles (s_memtime is per XCD: spans across workgroups are not comparable)' % (med(wg), wg.max()))
