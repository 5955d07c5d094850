#!/bin/bash
# round-2 visit b: new GPU tests, replay-schedule A/B, LBS census, fp32-noise gates
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_r2.py -m gpu -q -s 2>&1 | tail -120 > $OUT/pytest_r2.log
echo "pytest exit: $?" >> $OUT/pytest_r2.log
for i in 1 2; do
  for HD in 0 5; do
    LEMO_FIT_HEAD=$HD timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ramp-ms 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps20 head $HD: %.1f it/s  %.4f ms/step conv %.2f us lbs %.2f us' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']*1e3, d['roofline']['hbm']['kernel_ms']*1e3))" | tee -a $OUT/bench_head.txt
  done
done
for HD in 0 5; do
LEMO_FIT_HEAD=$HD timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --ramp-ms 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps100 head $HD: %.1f it/s  %.4f ms/step' % (d['value'], d['ms_per_step']))" | tee -a $OUT/bench_head.txt
done
timeout 120 python tools/lbs_census.py 2>&1 | grep -v amdgpu | tee $OUT/lbs_census.txt
timeout 900 python tools/r02_gates.py 2>&1 | grep -v "amdgpu\|Warn\|float(\|detach" | tee $OUT/gates.txt
tail -40 $OUT/pytest_r2.log
