mkdir -p gpurun_out/r03w; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03w/prof -o ae -- python $R/tools/ae_prof.py engine graph > $R/gpurun_out/r03w/prof.log 2>&1; echo rc=$?
cd $R
f=$(find gpurun_out/r03w/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r03w/ae_engine_kernel_stats.csv
t=$(find gpurun_out/r03w/prof -name "*kernel_trace.csv" | head -1); cp "$t" gpurun_out/r03w/ae_engine_kernel_trace.csv
rm -rf gpurun_out/r03w/prof
cut -c1-150 gpurun_out/r03w/ae_engine_kernel_stats.csv | head -12
python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from lemo_amd import synthetic
from lemo_amd.infill import AE, finetune_and_infill, finetune_and_infill_many
dev = torch.device('cuda:0')
w = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_ae_weights(7).items()}
ae = AE().to(dev); ae.load_state_dict(w)
g = torch.Generator().manual_seed(0)
xs = [torch.randn(1, 4, 210, 135, generator=g).to(dev) for _ in range(4)]
mask = (torch.ones(210, 135) > 0).to(dev)
for k in (1, 2, 4):
    finetune_and_infill_many(ae, w, xs[:k], [mask] * k, steps=60); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); finetune_and_infill_many(ae, w, xs[:k], [mask] * k, steps=60); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print('k=%d: %.1f ms per clip' % (k, best / k), flush=True)
PY
timeout 600 python -m pytest tests/test_gpu_r2.py tests/test_gpu_parity.py -m gpu -x -q -s -k "finetune or infill or ae or engine" 2>&1 | grep -v Warning | tail -12
