"""K clips per engine launch (lemo_ae_desc.clips): ms per clip of the 60-step infilling-AE finetune on 210 x 135 clip images
through lemo_amd.infill.finetune_and_infill_many, each configuration's results checked bit for bit against the solo runs
(diagnostic, GPU box only).  Usage: python tools/ae_clips.py [n_clips]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemo_amd import synthetic, infill
from lemo_amd.infill import AE, finetune_and_infill, finetune_and_infill_many
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
w = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_ae_weights(7).items()}
ae = AE().to(dev); ae.load_state_dict(w)
g = torch.Generator().manual_seed(0)
xs = [torch.randn(1, 4, 210, 135, generator=g).to(dev) for _ in range(n)]
masks = [(torch.rand(210, 135, generator=g) > 0.2).to(dev) for _ in range(n)]
side = torch.cuda.Stream(dev)
torch.cuda.synchronize()
with torch.cuda.stream(side):
    solo = [tuple(t.clone() for t in finetune_and_infill(ae, w, x, m, steps=60)) for x, m in zip(xs, masks)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for x, m in zip(xs, masks):
        finetune_and_infill(ae, w, x, m, steps=60)
    torch.cuda.synchronize()
    print(f'{n} clips one after the other through finetune_and_infill: {(time.perf_counter() - t0) * 1e3 / n:6.2f} ms per clip', flush=True)
    for clips in (1, 2, 3, 4, 6, 8, 12, 16):
        if clips > n:
            continue
        infill.AE_CLIPS = clips
        infill._SESSIONS.clear()
        many = finetune_and_infill_many(ae, w, xs, masks, steps=60); torch.cuda.synchronize()
        same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(solo, many))
        worst = max(float((a[0] - b[0]).abs().max()) for a, b in zip(solo, many))
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            finetune_and_infill_many(ae, w, xs, masks, steps=60); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        print(f'clips per engine {clips}: {best / n:6.2f} ms per clip ({n} clips, best of 3); bit-identical to solo: {same} (max |rec - solo| {worst:.1e})', flush=True)
