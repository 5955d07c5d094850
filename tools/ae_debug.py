import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from lemo_amd import synthetic
from lemo_amd.infill import AE
from oracle import lemo_oracle as O
dev = torch.device('cuda:0')
w = {k: torch.from_numpy(v) for k, v in synthetic.make_ae_weights(7).items()}
for (H, W) in ((210, 135), (64, 40)):
    ae = AE().to(dev); ae.load_state_dict(w)
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, H, W, generator=gen)
    wo = torch.randn(H, W, generator=gen)
    out, z = ae(x.to(dev))
    (out[0, 0] * wo.to(dev)).sum().backward()
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    oo, zo = O.ae_forward(wr, x)
    (oo[0, 0] * wo).sum().backward()
    print('size', H, W, 'out rel', float((out.cpu()-oo).abs().max()/oo.abs().max()))
    for k, p in ae.named_parameters():
        a, b = p.grad.cpu().double(), wr[k].grad.double()
        print('   %-28s max-rel %.2e  l2-rel %.2e' % (k, float((a-b).abs().max()/b.abs().max()), float((a-b).norm()/b.norm())))
