import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch.nn.functional as F
from lemo_amd import _hip
from lemo_amd._hip import ptr
from lemo_amd.priors import to_cg8p, cg8p_alloc, from_cg8p, pack_conv3x3_bwd
lib = _hip.get_lib(); dev = torch.device('cuda:0'); s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
for (H, W, ci, co) in ((105, 68, 64, 64), (210, 135, 32, 32), (53, 34, 128, 128), (105, 68, 32, 64), (100, 64, 64, 64), (64,64,64,64)):
    x = torch.randn(ci, H, W, generator=g); dy = torch.randn(co, H, W, generator=g)
    w = torch.zeros(co, ci, 3, 3, requires_grad=True)
    F.conv2d(x[None], w, padding=1).backward(dy[None])
    xb, dyb = to_cg8p(x).to(dev), to_cg8p(dy).to(dev)
    nsl = lib.conv3x3_wgrad_nslab(H, W)
    part = torch.empty(nsl * 9 * co * ci, device=dev); dw = torch.empty(co, ci, 3, 3, device=dev); db = torch.empty(co, device=dev)
    lib.check(lib.conv3x3_wgrad(ptr(dyb), ptr(xb), H, W, ci, co, ci, co, ptr(part), ptr(dw), ptr(db), s))
    torch.cuda.synchronize()
    e = (dw.cpu() - w.grad).abs()
    print(H, W, ci, co, 'nslab', nsl, 'dw max-rel %.2e' % float(e.max() / w.grad.abs().max()), 'db %.2e' % float((db.cpu() - dy.sum((1, 2))).abs().max() / dy.sum((1,2)).abs().max()),
          'bad taps', [int((e[:, :, t // 3, t % 3] > 1e-3 * w.grad.abs().max()).sum()) for t in range(9)])
    # bwd-data via conv v0
    wt = torch.randn(co, ci, 3, 3, generator=g) * 0.1
    xr = x.clone().requires_grad_(True); F.conv2d(xr[None], wt, padding=1).backward(dy[None])
    wb = torch.from_numpy(pack_conv3x3_bwd(wt.numpy())).to(dev)
    dx = cg8p_alloc(ci, H, W, dev); zb = torch.zeros(256, device=dev)
    lib.check(lib.conv3x3_mfma(ptr(dyb), ptr(wb), ptr(zb), None, ptr(dx), H, W, co, ci, 2, 0, s))
    torch.cuda.synchronize()
    print('      bwd-data max-rel %.2e' % float((from_cg8p(dx, H, W).cpu() - xr.grad).abs().max() / xr.grad.abs().max()))
