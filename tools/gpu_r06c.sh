#!/bin/bash
# Round-6 visit C: rocprofv3 kernel census of the infilling-AE finetune, 8 clips per engine, split-f16 vs fp32-input convolutions.
TAG=${1:-r06c}; R=$GRAFT_REPO_ROOT; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for a in f16 fp32; do
  cd /tmp && LEMO_AE_ARITH=$a timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$a -o p -- python $R/tools/ae_prof_k8.py 8 > $R/$OUT/ae_k8_$a.txt 2>&1
  cd $R; tail -n 3 $OUT/ae_k8_$a.txt
  f=$(find $OUT/prof_$a -name "*kernel_stats.csv" | head -n 1); cp $f $OUT/ae_k8_kernel_stats_$a.csv; head -n 14 $f | cut -c1-160
  rm -rf $OUT/prof_$a
done
