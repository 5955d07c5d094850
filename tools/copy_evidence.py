"""gpurun_out/<tag>/ (written by tools/gpu_round3.sh .. gpu_round6.sh) -> profiles/<round>_*: json / csv files replace their predecessors, text
files keep the previous visit's content below a separator (one generation).  Usage: python tools/copy_evidence.py r04final r04"""
import os, shutil, sys
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else 'r03'
src = os.path.join('gpurun_out', tag)
SEP = '# ---- previous visit of the round ----'
TEXT = ('.txt',)
for f in sorted(os.listdir(src)):
    p = os.path.join(src, f)
    if not os.path.isfile(p) or os.path.getsize(p) == 0 or f in ('bench.err', 'prof.err', 'pytest_tail.log', 'pytest_failures.txt', 'bench_prof.json'):
        continue
    dst = 'gpu_test_measurements.txt' if f == 'pytest_gpu_measurements.txt' else f
    q = os.path.join('profiles', rnd + '_' + dst)
    if f.endswith(TEXT) and f not in ('gpu.txt', 'smoke.txt', 'pmc_summary.txt'):
        new = ''.join(l for l in open(p) if 'amdgpu.ids' not in l)
        old = open(q).read() if os.path.exists(q) else ''
        old = old.split(SEP)[0].rstrip('\n')
        open(q, 'w').write(f'# visit {tag}\n' + new.rstrip('\n') + ('\n' + SEP + '\n' + old if old else '') + '\n')
    else:
        shutil.copy(p, q)
print('copied', tag, '->', rnd)
