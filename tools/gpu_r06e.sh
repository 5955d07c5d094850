#!/bin/bash
# Round-6 visit E: infilling AE ms per clip, both arithmetics, after giving the split-f16 launches the fp32 rule's K slices
TAG=${1:-r06e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r2.py -m gpu -q -s -k "infill or finetune" > $OUT/pytest_ae.log 2>&1; tail -n 8 $OUT/pytest_ae.log
for a in f16 fp32 f16 fp32; do
  echo "== LEMO_AE_ARITH=$a" | tee -a $OUT/ae_clips.txt
  LEMO_AE_ARITH=$a timeout 600 python tools/ae_clips.py 16 2>&1 | tee -a $OUT/ae_clips.txt | grep -E "clips per engine (1|8|16):|one after"
done
