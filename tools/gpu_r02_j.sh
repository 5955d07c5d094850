#!/bin/bash
# round-2 visit j: fused tail launch -- full GPU suite, then same-box A/B against the previous library (build_ab/prev.so)
export TMPDIR=/tmp
O=gpurun_out/${1:-r02j}; mkdir -p $O
timeout 1300 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $O/pytest_all.log
for rep in 1 2 3; do
  for so in "" lemo_amd/csrc/build_ab/prev.so; do
    n=$([ -z "$so" ] && echo product || basename $so .so)
    LEMO_HIP_LIB=$([ -z "$so" ] || echo $PWD/$so) timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --concurrent-clips 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %.1f it/s  %.1f us' % ('$n', d['value'], d['ms_per_step']*1e3))"
  done
done | tee $O/ab.log
