#!/bin/bash
# One GPU-box visit: parity tests, bench, rocprof kernel stats.  Usage (from the repo root, via gpurun):
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r01a'
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit" | head -4 > $OUT/gpu.txt
nproc >> $OUT/gpu.txt; lscpu | grep -E "Model name|^CPU\(s\)|Core|Socket" >> $OUT/gpu.txt
timeout 1200 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > $OUT/pytest_gpu.log
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
timeout 400 python bench.py --steps 100 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit: $?" >> $OUT/bench.err
timeout 200 python bench.py --steps 100 --warmup 10 --active-vertices-only --no-cpu-baseline > $OUT/bench_active.json 2>> $OUT/bench.err
timeout 200 python bench.py --steps 100 --warmup 10 --conv-variant 2 --no-cpu-baseline > $OUT/bench_v2.json 2>> $OUT/bench.err
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_prof.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats*" | head -3 | while read f; do cp "$f" $OUT/kernel_stats.csv; done
ls -R $OUT/prof | head -20
find $OUT/prof -name "*.db" -size +20M -delete 2>/dev/null
find $OUT/prof -name "*kernel_trace*" -size +20M -delete 2>/dev/null
head -40 $OUT/kernel_stats.csv 2>/dev/null
cat $OUT/bench.json; tail -5 $OUT/pytest_gpu.log
