"""rocprof target: the native PROX engine at the BASELINE configs[3]/[4] shape (B = 100, V = 10475, 256^3 SDF), S3 weights.
    cd /tmp && rocprofv3 --kernel-trace --stats -d out -o p -- python $REPO/tools/prox_engine_prof.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as ge
_prox_full_problem = ge.prox_full_problem
dev = torch.device('cuda:0')
stage = sys.argv[1] if len(sys.argv) > 1 else 'S3'
eng, _ = ge.prox_engine_for(_prox_full_problem(stage), dev, first_batch_flag=False)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    eng.step(300, use_graph=True)
torch.cuda.synchronize()
for n in (100, 900):
    t0 = time.time()
    with torch.cuda.stream(s):
        eng.step(n, use_graph=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f'PROX engine {stage} B=100 V=10475: {n} iterations in {dt * 1e3:.1f} ms = {n / dt:.1f} it/s ({dt / n * 1e3:.3f} ms/iteration); total loss {eng.loss_dict()["total_loss"]:.3f}')
