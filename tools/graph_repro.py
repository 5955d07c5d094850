"""which part of the AE training step breaks torch.cuda.graph capture on this stack? (diagnostic, GPU box)"""
import os, sys, faulthandler
faulthandler.enable()
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import _hip
from lemo_amd._hip import ptr
which = sys.argv[1]
dev = torch.device('cuda:0'); lib = _hip.get_lib()
x = torch.randn(1 << 16, device=dev); y = torch.zeros_like(x)
m = torch.zeros_like(x); v = torch.zeros_like(x); ctr = torch.zeros(1, dtype=torch.int32, device=dev)

def step():
    if which == 'torch':
        y.copy_(x * 2 + 1)
    elif which == 'kernel':
        lib.check(lib.adam_flat_ctr(ptr(y), ptr(x), ptr(m), ptr(v), x.numel(), 1e-3, ptr(ctr), lib.stream(dev)))
    elif which == 'autograd':
        w = x.clone().requires_grad_(True)
        (w * w).sum().backward()
        y.copy_(w.grad)
    elif which == 'enc':
        from lemo_amd.priors import Enc
        global enc, xin
        xr = xin.clone().requires_grad_(True)
        z = enc(xr)[0]
        z.square().mean().backward()
        y[:xr.numel()].copy_(xr.grad.reshape(-1))

if which.startswith('ae'):
    from lemo_amd.infill import AE, FlatAdam
    from lemo_amd.synthetic import make_ae_weights
    ae = AE().to(dev); ae.load_state_dict({k: torch.from_numpy(v) if not isinstance(v, torch.Tensor) else v for k, v in make_ae_weights(0).items()})
    xa = torch.randn(1, 4, int(os.environ.get("AE_H", 66)), int(os.environ.get("AE_W", 40)), device=dev)
    opt = FlatAdam(list(ae.parameters()), 3e-6)
    def step():
        if which == 'ae_fwd':
            with torch.no_grad():
                rec, _ = ae(xa)
            y[:rec.numel()].copy_(rec.reshape(-1))
        elif which == 'ae_fwdbwd':
            opt.zero_grad()
            rec, _ = ae(xa)
            (rec - xa[:, :1]).abs().mean().backward()
        else:
            opt.zero_grad()
            rec, _ = ae(xa)
            (rec - xa[:, :1]).abs().mean().backward()
            opt.step()
if which == 'enc':
    from lemo_amd.priors import Enc
    from lemo_amd.assets import load_assets
    enc = Enc().to(dev); enc.load_state_dict({k: torch.from_numpy(v) for k, v in load_assets()['enc_w'].items()})
    xin = torch.randn(1, 1, 40, 30, device=dev)
if which == 'ae_fn':
    from lemo_amd.infill import finetune_and_infill
    w = {k: v.detach().clone() for k, v in ae.state_dict().items()}
    mask = torch.ones(xa.shape[2], xa.shape[3], device=dev) > 0
    if os.environ.get('AE_PRE'):
        (ae(xa)[0] * 1.0).sum().backward()          # eager autograd before, like the test
    rec, z = finetune_and_infill(ae, w, xa, mask, steps=12, lr=3e-6)
    torch.cuda.synchronize()
    print('ae_fn replayed ok', float(rec.abs().sum()))
    sys.exit(0)
side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream(dev).wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
print(which, 'captured'); sys.stdout.flush()
for _ in range(3): g.replay()
torch.cuda.synchronize()
print(which, 'replayed ok', float(y.abs().sum()), int(ctr.item()))
