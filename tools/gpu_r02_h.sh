#!/bin/bash
# round-2 visit h: LBS blend-GEMM prefetch depth A/B (census + wall time), k clips side by side
export TMPDIR=/tmp
O=gpurun_out/${1:-r02h}; mkdir -p $O
python tools/lbs_census.py > $O/lbs_product.log 2>&1
for v in pfa3 pfa4; do
  LEMO_HIP_LIB=$PWD/lemo_amd/csrc/build_ab/$v.so python tools/lbs_census.py > $O/lbs_$v.log 2>&1
done
tail -2 $O/lbs_*.log
timeout 600 python tools/concurrent_clips.py 100 4 > $O/concurrent.log 2>&1
tail -6 $O/concurrent.log
