"""k clips' infilling-AE finetunes side by side (lemo_amd.infill.finetune_and_infill_many): ms per clip for k = 1..4, graph replay and
eager launches, on the native step engine and on the round-2 autograd path; and how far the two paths' results are apart
(diagnostic, GPU box only).  Usage: python tools/ae_concurrent.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import synthetic, infill
from lemo_amd.infill import AE, finetune_and_infill, finetune_and_infill_many
dev = torch.device('cuda:0')
w = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_ae_weights(7).items()}
ae = AE().to(dev); ae.load_state_dict(w)
g = torch.Generator().manual_seed(0)
xs = [torch.randn(1, 4, 210, 135, generator=g).to(dev) for _ in range(4)]
mask = (torch.ones(210, 135) > 0).to(dev)
ra, za = finetune_and_infill(ae, w, xs[0], mask, steps=60, engine=True)
pa = [p.detach().clone() for p in ae.parameters()]
rb, zb = finetune_and_infill(ae, w, xs[0], mask, steps=60, engine=False)
pb = [p.detach().clone() for p in ae.parameters()]
print('engine vs autograd path after 60 steps: max |d rec| %.3e (max |rec| %.3f), max |d z| %.3e, max |d param| %.3e (lr 3e-6: 60 steps move a parameter by <= 1.8e-4)' % (
    float((ra - rb).abs().max()), float(rb.abs().max()), float((za - zb).abs().max()), max(float((a - b).abs().max()) for a, b in zip(pa, pb))), flush=True)
side = torch.cuda.Stream(dev)
torch.cuda.synchronize()
with torch.cuda.stream(side):
    for eng in (True, False):
        finetune_and_infill(ae, w, xs[0], mask, steps=60, engine=eng); torch.cuda.synchronize()
        t0 = time.perf_counter(); finetune_and_infill(ae, w, xs[0], mask, steps=60, engine=eng); torch.cuda.synchronize()
        print('one clip through finetune_and_infill, caller on its own stream, engine=%s: %.1f ms' % (eng, (time.perf_counter() - t0) * 1e3), flush=True)
for engine in (True, False):
    print('engine:', engine, ' (autograd path: second stream for the weight gradients: %s)' % infill.WGRAD_SECOND_STREAM, flush=True)
    for use_graph in (True, False):
        for k in (1, 2, 4):
            finetune_and_infill_many(ae, w, xs[:k], [mask] * k, steps=60, use_graph=use_graph, engine=engine); torch.cuda.synchronize()
            t0 = time.perf_counter()
            finetune_and_infill_many(ae, w, xs[:k], [mask] * k, steps=60, use_graph=use_graph, engine=engine); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            # host time to ENQUEUE (no sync): how long the launches alone take
            t0 = time.perf_counter()
            finetune_and_infill_many(ae, w, xs[:k], [mask] * k, steps=60, use_graph=use_graph, engine=engine)
            th = (time.perf_counter() - t0) * 1e3
            torch.cuda.synchronize()
            print(f'  graph={use_graph} k={k}: {dt:7.1f} ms total = {dt / k:6.1f} ms per clip; host enqueue time {th:7.1f} ms', flush=True)
