"""Kernel census of the infilling-AE finetune with 8 clips per engine launch (the configuration VERDICT r04 #5 quotes): run under
rocprofv3 --kernel-trace --stats; three timed repetitions of n (argv[1], default 8) clips x 60 steps after one warm-up (diagnostic, GPU box only)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import synthetic, infill
from lemo_amd.infill import AE, finetune_and_infill_many
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
w = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_ae_weights(7).items()}
ae = AE().to(dev); ae.load_state_dict(w)
g = torch.Generator().manual_seed(0)
xs = [torch.randn(1, 4, 210, 135, generator=g).to(dev) for _ in range(n)]
masks = [(torch.rand(210, 135, generator=g) > 0.2).to(dev) for _ in range(n)]
infill.AE_CLIPS = n
side = torch.cuda.Stream(dev)
torch.cuda.synchronize()
with torch.cuda.stream(side):
    finetune_and_infill_many(ae, w, xs, masks, steps=60); torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter()
        finetune_and_infill_many(ae, w, xs, masks, steps=60); torch.cuda.synchronize()
        print(f'{n} clips per engine (LEMO_AE_SLAB_SCALE={os.environ.get("LEMO_AE_SLAB_SCALE", "default")}): {(time.perf_counter() - t0) * 1e3 / n:6.2f} ms per clip', flush=True)
