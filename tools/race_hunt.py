"""Which kernel's output changes when another clip runs next to it?  (diagnostic, GPU box only)

Round 2 claimed "every clip is bit-identical to a run on its own" while its own evidence showed 1.6489 vs 1.6486; round 3's
first visit showed the mismatch survives a correct stream ordering of load_sequence vs step.  This tool pins it down:
clip A's forward() + backward() (eager launches, no Adam: a pure function of the parameters) is run alone -> baseline
snapshot of EVERY intermediate buffer; then again while clip B replays its 100-step graphs on another stream; buffers are
compared in pipeline order and the first ones that differ are printed.
Usage: python tools/race_hunt.py [trials=40] [hammer=B|none]
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mode = sys.argv[2] if len(sys.argv) > 2 else 'B'         # B: another clip's graphs; torch: big torch matmuls + copies; none
hammer = mode != 'none'
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
A, pa = bench.build_problem(0, 119, dev, full_vertices=True, conv_variant=bench.DEFAULT_CONV_VARIANT)
B, pb = bench.build_problem(1, 119, dev, full_vertices=True, conv_variant=bench.DEFAULT_CONV_VARIANT)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
with torch.cuda.stream(sb):
    B.prepare(100)
with torch.cuda.stream(sa):
    A.step(7)                        # some non-trivial state
torch.cuda.synchronize(dev)


def buffers(f):
    out = []
    for k in ('h1', 'h2', 'vo'):
        out.append((k, f.ws[k]))
    for k in ('full_pose', 'R', 'J', 'T', 'A', 'Jtr', 'Xg', 'XgS'):
        out.append(('pose.' + k, f._pose_t[k]))
    for k in ('go_aa', 'verts', 'v_posed', 'x0', 'canon'):
        out.append((k, f.ws[k]))
    for l in range(1, 11):
        out.append((f'act[{l}]', f.act[l]))
    out += [('loss_acc', f.loss_acc), ('spartial', f.ws['spartial']), ('dact[0]', f.dact[0]), ('dact[1]', f.dact[1]), ('dx0', f.ws['dx0'])]
    for k in ('dvp', 'dA', 'dX', 'g_transl', 'g_rot6d', 'g_go', 'g_body', 'vp_scratch', 'g_other', 'losses'):
        out.append((k, f.ws[k]))
    return out


def run_a():
    with torch.cuda.stream(sa):
        A.forward()
        A.backward()


run_a(); torch.cuda.synchronize(dev)
base = [(k, t.clone()) for k, t in buffers(A)]
for _ in range(3):                                   # solo determinism
    run_a(); torch.cuda.synchronize(dev)
    bad = [k for (k, t), (_, b) in zip(buffers(A), base) if not torch.equal(t, b)]
    print('solo repeat differs in:', bad, flush=True)
X = torch.randn(4096, 4096, device=dev)
Z = torch.empty_like(X)


def where(name, t, b):
    """which (frame, ...) entries of a [B, ...] buffer differ"""
    d = (t != b)
    if d.dim() < 2 or d.shape[0] != 119:
        return f'{name}: {int(d.sum())} entries'
    fr = d.reshape(119, -1).any(1).nonzero().flatten().tolist()
    detail = ''
    if name in ('pose.R', 'pose.T', 'pose.A', 'pose.Jtr') and fr:
        f0 = fr[0]
        nj = 55
        jd = d[f0].reshape(nj, -1).any(1).nonzero().flatten().tolist()
        detail = f' frame {f0}: joints {jd}'
        if name == 'pose.R':
            detail += f' got {t[f0, jd[0]].flatten().tolist()} base {b[f0, jd[0]].flatten().tolist()}'
    return f'{name}: frames {fr}{detail}'


seen = {}
for trial in range(trials):
    if mode == 'B':
        with torch.cuda.stream(sb):
            B.step(100); B.step(100)
    elif mode == 'torch':
        with torch.cuda.stream(sb):
            for _ in range(60):
                Y = X @ X
                Z.copy_(Y)
    run_a()
    torch.cuda.synchronize(dev)
    bad = []
    for (k, t), (_, b) in zip(buffers(A), base):
        if not torch.equal(t, b):
            d = (t.double() - b.double()).abs()
            bad.append((k, int((d != 0).sum()), float(d.max()), float(b.double().abs().max())))
            if k in ('pose.full_pose', 'pose.R', 'pose.T', 'pose.A', 'pose.Xg', 'verts', 'v_posed', 'act[1]', 'act[2]', 'dA', 'dX', 'g_other'):
                print('   ', where(k, t, b), flush=True)
    if bad:
        first = bad[0][0]
        seen[first] = seen.get(first, 0) + 1
        print(f'trial {trial}: {len(bad)} buffers differ; in pipeline order: ' +
              '; '.join(f'{k} n={n} max|d|={m:.3g} (max|x|={x:.3g})' for k, n, m, x in bad[:8]), flush=True)
print('trials with a difference:', sum(seen.values()), 'of', trials, '; first differing buffer histogram:', seen)
