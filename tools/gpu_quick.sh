#!/bin/bash
# quick GPU visit: parity tests, bench (default kernels), LBS census
TAG=${1:-quick}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_gpu.log
timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2>> $OUT/bench.err
timeout 200 python tools/lbs_census.py > $OUT/lbs_census.log 2>&1
python - <<PY
import json
d=json.load(open('$OUT/bench.json')); print('bench', round(d['value'],1),'it/s', round(d['ms_per_step'],4),'ms  conv', round(d['roofline']['kernel_ms']*1e3,2),'us', round(d['roofline']['frac'],3))
PY
grep -v amdgpu.ids $OUT/lbs_census.log | tail -8; tail -4 $OUT/pytest_gpu.log; tail -3 $OUT/bench.err
