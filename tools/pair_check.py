"""conv variant 5 (fused layer pairs, csrc/conv_pair_kernels.hip) on the GPU: wall time of a forward pair and a backward-data pair
(HIP events around 20 launches each, and around a dependent chain of 6 like the iteration's) next to two variant-4 launches, and the
per-wave census of the forward pair (shader-clock stamps: staging / layer 1 / mid planes / end).  Diagnostic, GPU box only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import _hip
from lemo_amd._hip import ptr
from lemo_amd.assets import load_assets
from lemo_amd.priors import EncWeights, cg8p_alloc, to_cg8p

lib = _hip.get_lib(); dev = torch.device('cuda:0')
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (245, 134)
enc = EncWeights(load_assets()['enc_w'], dev)
g = torch.Generator().manual_seed(0)
x = to_cg8p(torch.randn(64, H, W, generator=g) * 0.3).to(dev)
bufs = [cg8p_alloc(64, H, W, dev) for _ in range(4)]
s = torch.cuda.current_stream(dev).cuda_stream
P = {(l, b): enc.split_pack(l, b, 5) for l in range(3, 10) for b in (False, True)}


def pair_fwd(src, mid, dst, l=3):
    (pa, ia), (pb, ib) = P[(l, False)], P[(l + 1, False)]
    lib.check(lib.conv3x3_pair_f16(ptr(src), ptr(pa), ia, ptr(enc.b[l]), None, ptr(mid), ptr(pb), ib, ptr(enc.b[l + 1]), None, ptr(dst), H, W, 0, None, s))


def pair_bwd(src, a1, a0, dst, l=4):
    (pa, ia), (pb, ib) = P[(l, True)], P[(l - 1, True)]
    lib.check(lib.conv3x3_pair_f16(ptr(src), ptr(pa), ia, None, ptr(a1), None, ptr(pb), ib, None, ptr(a0), ptr(dst), H, W, 1, None, s))


def single(src, dst, l, bwd, aux=None):
    pk, iv = P[(l, bwd)]
    w = enc.wbwd[l] if bwd else enc.w[l]
    lib.check(lib.conv3x3_mfma_split_f16(ptr(src), ptr(pk), iv, ptr(w), None if bwd else ptr(enc.b[l]), ptr(aux) if bwd else None, ptr(dst), H, W, 64, 64,
                                         1 if bwd else 0, s))


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print('forward pair, same buffers back to back      %.2f us' % timeit(lambda: pair_fwd(x, bufs[0], bufs[1])))
print('two variant-4 launches (same two layers)     %.2f us' % timeit(lambda: (single(x, bufs[0], 3, False), single(bufs[0], bufs[1], 4, False))))
print('backward-data pair back to back              %.2f us' % timeit(lambda: pair_bwd(bufs[1], bufs[0], x, bufs[2])))
print('two variant-4 backward launches              %.2f us' % timeit(lambda: (single(bufs[1], bufs[3], 4, True, bufs[0]), single(bufs[3], bufs[2], 3, True, x))))


def chain():      # dependent chain like the iteration's: 3 forward pairs, 3 backward pairs, each reading what the previous wrote
    pair_fwd(x, bufs[0], bufs[1], 3); pair_fwd(bufs[1], bufs[2], bufs[3], 5); pair_fwd(bufs[3], bufs[0], bufs[1], 7)
    pair_bwd(bufs[1], bufs[0], bufs[3], bufs[2], 9); pair_bwd(bufs[2], bufs[3], bufs[1], bufs[0], 7); pair_bwd(bufs[0], bufs[1], x, bufs[2], 5)


print('dependent chain of 6 pairs                   %.2f us per launch' % (timeit(chain, 10) / 6))

ntx, nty = (W + 13) // 14, (H + 9) // 10
nblk = ntx * nty
dbg = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device=dev)
(pa, ia), (pb, ib) = P[(3, False)], P[(4, False)]
for it in range(3):
    dbg.zero_()
    lib.check(lib.conv3x3_pair_f16(ptr(x), ptr(pa), ia, ptr(enc.b[3]), None, ptr(bufs[0]), ptr(pb), ib, ptr(enc.b[4]), None, ptr(bufs[1]), H, W, 0, ptr(dbg), s))
    torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nblk, 8, 8)
t0, t1, tp, tl1, tmid, tex, tmx = d[..., 2], d[..., 3], d[..., 4], d[..., 5], d[..., 6], d[..., 1], d[..., 7]
med = lambda a: int(np.median(a))
print('%d workgroups; per-wave cycles median %d max %d; staging %d | layer 1 %d | exchange %d + epilogue 1 / max barrier %d + mid planes %d | layer 2 + epilogue 2 %d' % (
    nblk, med(t1 - t0), (t1 - t0).max(), med(tp - t0), med(tl1 - tp), med(tex - tl1), med(tmx - tex), med(tmid - tmx), med(t1 - tmid)))
for w in range(8):
    print('  wave %d: staging %d | L1 %d | exch %d | epi+max %d | planes %d | L2 %d' % (w, med(tp[:, w] - t0[:, w]), med(tl1[:, w] - tp[:, w]), med(tex[:, w] - tl1[:, w]),
                                                                                  med(tmx[:, w] - tex[:, w]), med(tmid[:, w] - tmx[:, w]), med(t1[:, w] - tmid[:, w])))
wg = (t1.max(1) - t0.min(1))
print('per-workgroup lifetime median %d max %d cycles; launch span %d cycles' % (med(wg), wg.max(), t1.max() - t0.min()))
