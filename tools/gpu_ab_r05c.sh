#!/bin/bash
# Round-5 A/B visit: PROX window with (a) the frame and dense loss roles in one launch, (b) the two closing reductions of the all-vertex
# backward in one launch, (c) prox_sparse's per-vertex chain ahead of the loss totals -- against the previous build (build_ab/old.so)
# and with each switch back on its own.  gpurun --timeout 900 -- 'bash tools/gpu_ab_r05c.sh r05abprox'
TAG=${1:-r05abprox}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
val() { python -c "import json,sys; print(json.load(open(sys.argv[1]))['value'])" $1 2>/dev/null; }
timeout 700 python -m pytest tests/test_gpu_teacher.py tests/test_gpu_r2.py tests/test_gpu_parity.py -m gpu -q -s -k "prox" > $OUT/pytest_prox.log 2>&1; grep -E "worst gradient|passed|failed" $OUT/pytest_prox.log | cut -c1-400
for i in 1 2 3; do
  LEMO_HIP_LIB=$R/lemo_amd/csrc/build_ab/old.so timeout 300 python bench.py --workload prox --steps 300 --warmup 100 > $OUT/prox_old_$i.json 2>> $OUT/err.log; echo "prox previous build run $i: $(val $OUT/prox_old_$i.json)"
  timeout 300 python bench.py --workload prox --steps 300 --warmup 100 > $OUT/prox_new_$i.json 2>> $OUT/err.log; echo "prox new run $i: $(val $OUT/prox_new_$i.json)"
  LEMO_PROX_TWO_LAUNCHES=1 timeout 300 python bench.py --workload prox --steps 300 --warmup 100 > $OUT/prox_new_two_launches_$i.json 2>> $OUT/err.log; echo "prox new, frame / dense as two launches run $i: $(val $OUT/prox_new_two_launches_$i.json)"
  LEMO_LBS_TWO_REDUCES=1 timeout 300 python bench.py --workload prox --steps 300 --warmup 100 > $OUT/prox_new_two_reduces_$i.json 2>> $OUT/err.log; echo "prox new, two reduction launches run $i: $(val $OUT/prox_new_two_reduces_$i.json)"
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/pp -o p -- python $R/tools/prox_engine_prof.py S3 > $R/$OUT/prox_engine.txt 2>&1
cd $R
find $OUT/pp -name "*kernel_stats*" | head -n 1 | while read f; do cp "$f" $OUT/prox_kernel_stats.csv; done
rm -rf $OUT/pp
head -n 26 $OUT/prox_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
tail -n 3 $OUT/err.log
