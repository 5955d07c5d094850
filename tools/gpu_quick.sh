#!/bin/bash
# quick A/B of conv variants + parity tests
TAG=${1:-quick}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_gpu.log
for v in 2 1; do timeout 200 python bench.py --steps 100 --warmup 10 --conv-variant $v --no-cpu-baseline > $OUT/bench_v$v.json 2>> $OUT/bench.err; done
python - <<PY
import json
for v in (2,1):
    try:
        d=json.load(open('$OUT/bench_v%d.json'%v)); print('variant',v,round(d['value'],1),'it/s', round(d['ms_per_step'],4),'ms  conv', round(d['roofline']['kernel_ms']*1e3,2),'us', round(d['roofline']['frac'],3))
    except Exception as e: print('variant',v,'failed',e)
PY
tail -4 $OUT/pytest_gpu.log; tail -3 $OUT/bench.err
