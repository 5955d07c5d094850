#!/bin/bash
# Round-6 visit A: the Winograd layer (conv variant 10) -- parity tests, wino_check.py (accuracy / timing / census), headline A/B
# against variant 9, three interleaved runs.  gpurun --timeout 1200 -- 'bash tools/gpu_r06a.sh r06a'
TAG=${1:-r06a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --concurrent-clips 0 --no-extras"
val() { python -c "import json,sys; print(json.load(open(sys.argv[1]))['value'])" $1 2>/dev/null; }
timeout 300 python tools/wino_check.py > $OUT/wino_check.txt 2>&1; cat $OUT/wino_check.txt | tail -n 22
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "wino or (fit_full_size_golden and 10) or (encoder_full_size_golden and 10)" > $OUT/pytest_v10.log 2>&1; tail -n 6 $OUT/pytest_v10.log
for i in 1 2 3; do
  for v in 9 10; do
    timeout 300 python bench.py --steps 100 --warmup 10 $B --conv-variant $v > $OUT/amass_v${v}_$i.json 2>> $OUT/err.log; echo "amass variant $v run $i: $(val $OUT/amass_v${v}_$i.json)"
  done
done
tail -n 5 $OUT/err.log
