#!/bin/bash
TAG=${1:-r03g}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/split_check.py > $OUT/split_check.txt 2>&1; grep -v amdgpu $OUT/split_check.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_gates.py -q -s -k small > $OUT/pytest_gates.log 2>&1; grep -E "passed|failed|Error" $OUT/pytest_gates.log | tail -3
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench.err; python - <<PY
import json
d=json.load(open('$OUT/bench_full.json'))
print('value', d['value'], 'v100', d.get('value_100_steps'))
for k in ('prox_window','perframe','ae_finetune','concurrent_clips','cpu_baseline'):
    print(k, {kk: vv for kk, vv in d.get(k, {}).items() if kk in ('value','unit','error','one_clip_value','cores')})
PY
tail -3 $OUT/bench.err
timeout 600 python bench.py --workload prox --gpus 1 --steps 300 --warmup 100 > $OUT/bench_prox.json 2>> $OUT/bench.err; cut -c1-400 $OUT/bench_prox.json
