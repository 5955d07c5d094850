#!/bin/bash
# split-bf16 conv: accuracy/timing check + bench of conv variant 3 only
TAG=${1:-splitq}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/split_check.py > $OUT/split_check.log 2>&1
timeout 200 python bench.py --steps 100 --warmup 10 --conv-variant 3 --no-cpu-baseline > $OUT/bench_v3.json 2>> $OUT/bench.err
cat $OUT/split_check.log
python - <<PY
import json
d=json.load(open('$OUT/bench_v3.json')); print('variant 3',round(d['value'],1),'it/s', round(d['ms_per_step'],4),'ms  conv', round(d['roofline']['kernel_ms']*1e3,2),'us', round(d['roofline']['frac'],3))
PY
tail -3 $OUT/bench.err
