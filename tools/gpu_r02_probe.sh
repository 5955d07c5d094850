#!/bin/bash
# round-2 probe: GPU tests, driver-style bench with / without the clock ramp, kernel stats, LBS census.
#   gpurun --timeout 1200 -- 'bash tools/gpu_r02_probe.sh r02a'
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit" | head -4 > $OUT/gpu.txt
nproc >> $OUT/gpu.txt; lscpu | grep -E "Model name|^CPU\(s\)|Core|Socket" >> $OUT/gpu.txt
timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -80 > $OUT/pytest_gpu.log
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
for i in 1 2; do
  for R in 0 300; do
    timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ramp-ms $R 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps20 ramp $R: %.1f it/s  %.4f ms/step conv %.2f us lbs %.2f us' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']*1e3, d['roofline']['hbm']['kernel_ms']*1e3))" | tee -a $OUT/bench_ramp.txt
  done
done
timeout 300 python bench.py --steps 100 --warmup 10 > $OUT/bench100.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench100.json')); print('steps100: %.1f it/s' % d['value'], d['cpu_baseline'])" | tee -a $OUT/bench_ramp.txt
timeout 120 python tools/lbs_census.py 2>&1 | grep -v amdgpu | tee $OUT/lbs_census.txt
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_prof.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats*" | head -3 | while read f; do cp "$f" $OUT/kernel_stats.csv; done
find $OUT/prof -name "*.db" -size +20M -delete 2>/dev/null
find $OUT/prof -name "*kernel_trace*" -size +20M -delete 2>/dev/null
head -30 $OUT/kernel_stats.csv 2>/dev/null
tail -5 $OUT/pytest_gpu.log
