#!/bin/bash
# run tools/split_check.py on the product library and on every A/B build under lemo_amd/csrc/build_ab/
TAG=${1:-ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== product"; timeout 300 python tools/split_check.py 2>&1 | grep -v amdgpu.ids | tee $OUT/product.log
for so in lemo_amd/csrc/build_ab/*.so; do
  [ -f "$so" ] || continue
  echo "== $so"; LEMO_AB_LIB=$PWD/$so timeout 300 python tools/split_check.py 2>&1 | grep -v amdgpu.ids | tee $OUT/$(basename $so .so).log
done
