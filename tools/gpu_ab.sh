#!/bin/bash
# run tools/split_check.py on the product library and on every A/B build under lemo_amd/csrc/build_ab/
TAG=${1:-ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== product"; timeout 300 python tools/split_check.py 2>&1 | grep -v amdgpu.ids | tee $OUT/product.log
for so in lemo_amd/csrc/build_ab/*.so; do
  [ -f "$so" ] || continue
  echo "== $so"; LEMO_HIP_LIB=$PWD/$so timeout 300 python tools/split_check.py 2>&1 | grep -v amdgpu.ids | tee $OUT/$(basename $so .so).log
done
# engine-level A/B (same box): bench with each library
for so in "" lemo_amd/csrc/build_ab/*.so; do
  [ -z "$so" ] || [ -f "$so" ] || continue
  n=$([ -z "$so" ] && echo product || basename $so .so)
  LEMO_HIP_LIB=$([ -z "$so" ] || echo $PWD/$so) timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench %-10s %.1f it/s  conv %.2f us' % ('$n', d['value'], d['roofline']['kernel_ms']*1e3))"
done
