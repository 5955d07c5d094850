"""Where/when do the workgroups of the 64->64 LDS-tiled conv run?  (diagnostic, GPU box only)
Prints blocks-per-CU histogram, per-wave duration in shader cycles and the launch span."""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import _hip
from lemo_amd._hip import ptr
from lemo_amd.priors import cg8p_alloc, pack_conv3x3, pack_conv3x3_gmajor

lib = _hip.get_lib(); dev = torch.device('cuda:0')
H, W = 245, 134
g = torch.Generator().manual_seed(0)
w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).numpy()
wt, wt2 = torch.from_numpy(pack_conv3x3(w)).to(dev), torch.from_numpy(pack_conv3x3_gmajor(w)).to(dev)
b = torch.zeros(64, device=dev)
x = cg8p_alloc(64, H, W, dev); x.normal_(); out = cg8p_alloc(64, H, W, dev)
nblk = H * W // 128 + 16
dbg = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device=dev)
s = torch.cuda.current_stream(dev).cuda_stream
for it in range(3):
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.check(lib.conv3x3_mfma_lds_census(ptr(x), ptr(wt), ptr(wt2), ptr(b), ptr(out), H, W, 64, 64, ptr(dbg), s))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
d = dbg.cpu().numpy().reshape(nblk, 8, 8)
tp, tl = d[..., 4], d[..., 5]
hw, xcc, t0, t1 = d[..., 0], d[..., 1] & 0xf, d[..., 2], d[..., 3]
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
valid = t1 > 0
cuid = (xcc * 8 + se) * 32 + sh * 16 + cu
blocks_cu = collections.Counter(cuid[:, 0][valid[:, 0]].tolist())
print('launch wall (events): %.1f us' % (ms * 1e3))
print('distinct CUs used:', len(blocks_cu), ' blocks-per-CU histogram:', sorted(collections.Counter(blocks_cu.values()).items()))
dur = (t1 - t0)[valid]
tmin = t0[valid].min()
print('per-wave cycles: min %d median %d max %d' % (dur.min(), np.median(dur), dur.max()))
print('span first-start -> last-end: %d cycles ; start skew (last start - first start): %d' % (t1[valid].max() - tmin, t0[valid].max() - tmin))
main = valid.copy(); main[:16] = False          # tail blocks have the lowest ids
print('main waves: median %d max %d ; tail waves: %s' % (np.median((t1 - t0)[main]), (t1 - t0)[main].max(), (t1 - t0)[:16][valid[:16]].tolist()))
wsimd = collections.Counter(zip(cuid[valid].tolist(), simd[valid].tolist()))
print('waves per (CU,SIMD) histogram:', sorted(collections.Counter(wsimd.values()).items()))
print('implied clock if 36.9k cycles == MFMA-bound: span cycles / wall = %.2f GHz' % ((t1[valid].max() - tmin) / (ms * 1e-3) / 1e9))
mm = main & (tp > 0)
print('main-wave breakdown (median cycles): prologue %d  loop %d  epilogue %d' % (np.median((tp - t0)[mm]), np.median((tl - tp)[mm]), np.median((t1 - tl)[mm])))
