#!/bin/bash
TAG=${1:-r02f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $OUT/pytest_all.log
timeout 600 python -m pytest tests/test_gpu_gates.py -m gpu -q -s 2>&1 | grep -E "step|gradient|passed|failed" > $OUT/gates.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/proxprof -o p -- python $GRAFT_REPO_ROOT/tools/prox_engine_prof.py S3 > $GRAFT_REPO_ROOT/$OUT/prox_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/proxprof -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/prox_kernel_stats.csv; done
rm -rf $OUT/proxprof
timeout 120 python tools/prox_engine_prof.py S2 2>&1 | grep PROX >> $OUT/prox_prof.txt
cut -c1-150 $OUT/prox_kernel_stats.csv | head -30
cat $OUT/prox_prof.txt | grep PROX; cat $OUT/pytest_all.log | tail -8
