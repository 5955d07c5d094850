#!/bin/bash
# Round-6 visit B: infilling AE on split-f16 convolutions -- GPU parity tests of the AE, then ms per clip for both arithmetics
# (LEMO_AE_ARITH=fp32 = the fp32-input MFMA convolutions of rounds 3-5).  gpurun --timeout 1200 -- 'bash tools/gpu_r06b.sh r06b'
TAG=${1:-r06b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r2.py -m gpu -q -s -x -k "infill or finetune" > $OUT/pytest_ae.log 2>&1; tail -n 8 $OUT/pytest_ae.log
for a in f16 fp32 f16 fp32; do
  echo "== LEMO_AE_ARITH=$a" | tee -a $OUT/ae_clips.txt
  LEMO_AE_ARITH=$a timeout 600 python tools/ae_clips.py 16 2>&1 | tee -a $OUT/ae_clips.txt | grep -E "clips per engine (1|8|16):|one after"
done
