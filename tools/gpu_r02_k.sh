#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r02k}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gates.py -m gpu -q -x -s 2>&1 | grep -E "small problem|B=119|step|passed|failed|Error|assert" | tee $O/gates.log | tail -60
for rep in 1 2 3; do
  for so in "" lemo_amd/csrc/build_ab/prev.so; do
    n=$([ -z "$so" ] && echo product || basename $so .so)
    LEMO_HIP_LIB=$([ -z "$so" ] || echo $PWD/$so) timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --concurrent-clips 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %.1f it/s  %.1f us' % ('$n', d['value'], d['ms_per_step']*1e3))"
  done
done | tee $O/ab.log
