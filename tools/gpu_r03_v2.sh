#!/bin/bash
TAG=${1:-r03b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 500 python tools/race_hunt.py 40 > $OUT/race_hunt.txt 2>&1; tail -25 $OUT/race_hunt.txt | cut -c1-700
timeout 900 python -m pytest tests/test_gpu_r2.py -q -s -k "perframe" > $OUT/pytest_perframe.log 2>&1; grep -E "per-frame|marker residual|3 frames|Error|assert|passed|failed" $OUT/pytest_perframe.log | cut -c1-400
