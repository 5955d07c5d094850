#!/bin/bash
# round-2 visit i: measured peaks of the box, new GPU tests (concurrent clips, chained PROX windows, AE workspace), default bench
export TMPDIR=/tmp
O=gpurun_out/${1:-r02i}; mkdir -p $O
./tools/ubench/peak_ubench > $O/peaks.txt 2>&1; cat $O/peaks.txt
timeout 900 python -m pytest tests/test_gpu_r2.py tests/test_gpu_parity.py -m gpu -q -s -k "concurrent or two_prox_windows or infill or finetune" 2>&1 | grep -E "passed|failed|Error|error|AE|finetune|s/clip|ms" | tail -20 | tee $O/pytest_new.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
