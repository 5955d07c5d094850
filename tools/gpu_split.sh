#!/bin/bash
# split-bf16 conv: accuracy/timing check, parity tests and A/B bench of conv variants 3 vs 2
TAG=${1:-split}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/split_check.py > $OUT/split_check.log 2>&1
for v in 3 2; do timeout 200 python bench.py --steps 100 --warmup 10 --conv-variant $v --no-cpu-baseline > $OUT/bench_v$v.json 2>> $OUT/bench.err; done
LEMO_CONV_VARIANT=3 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_gpu.log
cat $OUT/split_check.log
python - <<PY
import json
for v in (3,2):
    try:
        d=json.load(open('$OUT/bench_v%d.json'%v)); print('variant',v,round(d['value'],1),'it/s', round(d['ms_per_step'],4),'ms  conv', round(d['roofline']['kernel_ms']*1e3,2),'us', round(d['roofline']['frac'],3))
    except Exception as e: print('variant',v,'failed',e)
PY
tail -4 $OUT/pytest_gpu.log; tail -3 $OUT/bench.err
