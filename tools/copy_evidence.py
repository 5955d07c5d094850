"""gpurun_out/<tag>/ (written by tools/gpu_round3.sh) -> profiles/r03_*: json / csv files replace their predecessors, text files keep
the previous visit's content below a separator (one generation).  Usage: python tools/copy_evidence.py r03zz"""
import os, shutil, sys
tag = sys.argv[1]
src = os.path.join('gpurun_out', tag)
M = {'bench_driver_style.json': None, 'bench_driver_style_1.json': None, 'bench_driver_style_2.json': None, 'bench_100.json': None,
     'bench_100_again.json': None, 'bench_100_blend_bf16x3.json': None, 'bench_100_variant3_bf16x3.json': None, 'bench_active.json': None,
     'bench_prox.json': None, 'kernel_stats.csv': None, 'prox_kernel_stats.csv': None, 'pmc_summary.json': None, 'pmc_summary.txt': None,
     'ae_engine_kernel_stats.csv': None, 'gpu.txt': None, 'smoke.txt': None,
     'pytest_gpu_measurements.txt': 'gpu_test_measurements.txt'}
T = ['concurrent_clips.txt', 'perframe_batched.txt', 'race_hunt.txt', 'split_check.txt', 'prox_engine.txt', 'ae_concurrent.txt', 'ae_wgrad_probe.txt']
SEP = '# ---- previous visit of the round ----'
for f, dst in M.items():
    p = os.path.join(src, f)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join('profiles', 'r03_' + (dst or f)))
for f in T:
    p = os.path.join(src, f)
    if not os.path.exists(p) or os.path.getsize(p) == 0:
        continue
    new = ''.join(l for l in open(p) if 'amdgpu.ids' not in l)
    q = os.path.join('profiles', 'r03_' + f)
    old = open(q).read() if os.path.exists(q) else ''
    old = old.split(SEP)[0].rstrip('\n')
    open(q, 'w').write(f'# visit {tag} (tools/gpu_round3.sh)\n' + new.rstrip('\n') + ('\n' + SEP + '\n' + old if old else '') + '\n')
print('copied', tag)
