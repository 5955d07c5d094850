"""conv variant 10 (Winograd F(2x2, 3x3) 64 -> 64 layer, csrc/conv_wino_kernels.hip) on the GPU: accuracy of every 64 -> 64 layer of the
encoder (forward and backward-data, real weights) against float64 next to torch's fp32 convolution and the split-f16 direct kernel;
wall time of one launch (HIP events around 20 launches, and around a dependent chain of 14 like the iteration's) next to the fused
pair and two variant-4 launches; per-wave census (shader-clock stamps: loads issued / transformed / planes written / GEMMs / exchange /
end).  Diagnostic, GPU box only."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import _hip
from lemo_amd._hip import ptr
from lemo_amd.assets import load_assets
from lemo_amd.priors import EncWeights, cg8p_alloc, to_cg8p, from_cg8p, enc_layer_keys

lib = _hip.get_lib(); dev = torch.device('cuda:0')
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (245, 134)
A = load_assets()
enc = EncWeights(A['enc_w'], dev)
keys = enc_layer_keys()
g = torch.Generator().manual_seed(0)
s = torch.cuda.current_stream(dev).cuda_stream


def wino(src, dst, l, bwd, aux=None, dbg=None):
    pk, iv = enc.split_pack(l, bwd, 10)
    w = enc.wbwd[l] if bwd else enc.w[l]
    lib.check(lib.conv3x3_wino_f16(ptr(src), ptr(pk), iv, ptr(w), None if bwd else ptr(enc.b[l]), ptr(aux) if bwd else None, ptr(dst), H, W,
                                   1 if bwd else 0, ptr(dbg) if dbg is not None else None, s), 'wino')


def single(src, dst, l, bwd, aux=None):
    pk, iv = enc.split_pack(l, bwd, 4)
    w = enc.wbwd[l] if bwd else enc.w[l]
    lib.check(lib.conv3x3_mfma_split_f16(ptr(src), ptr(pk), iv, ptr(w), None if bwd else ptr(enc.b[l]), ptr(aux) if bwd else None, ptr(dst), H, W, 64, 64,
                                         1 if bwd else 0, s))


def pair_fwd(src, mid, dst, l=3):
    (pa, ia), (pb, ib) = enc.split_pack(l, False, 5), enc.split_pack(l + 1, False, 5)
    lib.check(lib.conv3x3_pair_f16(ptr(src), ptr(pa), ia, ptr(enc.b[l]), None, ptr(mid), ptr(pb), ib, ptr(enc.b[l + 1]), None, ptr(dst), H, W, 0, None, s))


# ---- accuracy, layer by layer, on a chain of realistic activations (float64 chain from a random image through the real weights) -------
x0 = torch.randn(1, 1, H, W, generator=g, dtype=torch.float64) * 0.5
acts = [x0]
for l in range(10):
    w, b = torch.from_numpy(A['enc_w'][keys[l] + '.weight']).double(), torch.from_numpy(A['enc_w'][keys[l] + '.bias']).double()
    acts.append(F.leaky_relu(F.conv2d(acts[-1], w, b, padding=1), 0.2))
gz = torch.randn(1, 64, H, W, generator=g, dtype=torch.float64) * 1e-5
print('layer | forward: torch-fp32  split-f16 direct  Winograd split-f16 | backward-data: torch-fp32  split-f16 direct  Winograd split-f16   (max err / layer max vs float64)')
worst = 0.0
for l in range(3, 10):
    w64, b64 = torch.from_numpy(A['enc_w'][keys[l] + '.weight']).double(), torch.from_numpy(A['enc_w'][keys[l] + '.bias']).double()
    ref = F.leaky_relu(F.conv2d(acts[l], w64, b64, padding=1), 0.2)[0]
    e = lambda y: float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    a32 = acts[l].float()
    src = to_cg8p(a32[0]).to(dev)
    o1, o2 = cg8p_alloc(64, H, W, dev), cg8p_alloc(64, H, W, dev)
    single(src, o1, l, False); wino(src, o2, l, False)
    torch.cuda.synchronize()
    row = [e(F.leaky_relu(F.conv2d(a32, w64.float(), b64.float(), padding=1), 0.2)[0]), e(from_cg8p(o1, H, W)), e(from_cg8p(o2, H, W))]
    # backward-data with the saved activation act[l]
    refb = F.conv_transpose2d(gz, w64, padding=1)[0] * torch.where(acts[l][0] > 0, 1.0, 0.2)
    eb = lambda y: float((y.double().cpu() - refb).abs().max() / refb.abs().max())
    gsrc, aux = to_cg8p(gz[0].float()).to(dev), to_cg8p(acts[l][0].float()).to(dev)
    single(gsrc, o1, l, True, aux); wino(gsrc, o2, l, True, aux)
    torch.cuda.synchronize()
    rowb = [eb(F.conv_transpose2d(gz.float(), w64.float(), padding=1)[0] * torch.where(acts[l][0] > 0, 1.0, 0.2).float()), eb(from_cg8p(o1, H, W)), eb(from_cg8p(o2, H, W))]
    worst = max(worst, row[2] / row[0], rowb[2] / rowb[0])
    print('%5d | %.2e  %.2e  %.2e | %.2e  %.2e  %.2e' % (l, *row, *rowb))
print('worst Winograd error / torch-fp32 error over the 14 layer applications: %.2f' % worst)

# ---- timing -------------------------------------------------------------------------------------------------------------------------
x = to_cg8p(torch.randn(64, H, W, generator=g) * 0.3).to(dev)
bufs = [cg8p_alloc(64, H, W, dev) for _ in range(4)]


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def chain14():          # the iteration's 14 launches: layers 3..9 forward, 9..3 backward, each reading what the previous one wrote
    cur = x
    for i, l in enumerate(range(3, 10)):
        wino(cur, bufs[i & 1], l, False); cur = bufs[i & 1]
    for i, l in enumerate(range(9, 2, -1)):
        wino(cur, bufs[2 + (i & 1)], l, True, bufs[i & 1]); cur = bufs[2 + (i & 1)]


for rep in range(3):
    print('rep %d: Winograd forward launch %.2f us | backward-data launch %.2f us | dependent chain of 14: %.2f us per launch || fused pair (two layers) %.2f us | '
          'variant-4 single layer %.2f us' % (rep, timeit(lambda: wino(x, bufs[0], 3, False)), timeit(lambda: wino(bufs[0], bufs[1], 3, True, x)), timeit(chain14, 10) / 14,
                                             timeit(lambda: pair_fwd(x, bufs[0], bufs[1])), timeit(lambda: single(x, bufs[0], 3, False))))

# ---- census -------------------------------------------------------------------------------------------------------------------------
T = (H // 2) * ((W + 1) // 2)
nwg = (T + 31) // 32
dbg = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
for it in range(3):
    dbg.zero_()
    wino(x, bufs[0], 3, False, dbg=dbg)
    torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nwg, 8, 8)
t0, tld, ttr, tp1, tmm, tx, t1 = (d[..., i] for i in range(7))
med = lambda a: int(np.median(a))
print('%d workgroups; per-wave cycles median %d max %d: patch loads issued (+ odd row) %d | loads landed + transform + max barrier %d | split + planes %d | 16 GEMMs %d | '
      'row transform + exchange %d | column transform + epilogue + stores %d' % (nwg, med(t1 - t0), (t1 - t0).max(), med(tld - t0), med(ttr - tld), med(tp1 - ttr),
                                                                               med(tmm - tp1), med(tx - tmm), med(t1 - tx)))
wg = t1.max(1) - t0.min(1)
print('per-workgroup lifetime median %d max %d cycles; first start to last end %d cycles' % (med(wg), wg.max(), t1.max() - t0.min()))
