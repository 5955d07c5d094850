#!/bin/bash
# Last visit of a round: the shipped library on a fresh box -- smoke, the PROX / LBS / fit-golden GPU tests, the two bench lines.
TAG=${1:-r06check}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1 | cut -c1-120
timeout 700 python -m pytest tests/test_gpu_teacher.py tests/test_gpu_r2.py tests/test_gpu_parity.py -m gpu -q -k "prox or lbs or vertex_backward or body_model or fit_full_size_golden or infill or ae_" > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | cut -c1-300
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --concurrent-clips 0 --no-extras > $OUT/bench_amass.json 2>> $OUT/err.log; python -c "import json; d=json.load(open('$OUT/bench_amass.json')); print('amass', d['value'], d['roofline']['frac'])"
timeout 300 python bench.py --workload prox --steps 300 --warmup 100 > $OUT/bench_prox.json 2>> $OUT/err.log; python -c "import json; d=json.load(open('$OUT/bench_prox.json')); print('prox', d['value'])"
