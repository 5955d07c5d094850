#!/bin/bash
# same-box A/B of the whole fitting engine: product library vs every build under lemo_amd/csrc/build_ab/*.so, interleaved
export TMPDIR=/tmp
for rep in 1 2 3; do
  for so in "" lemo_amd/csrc/build_ab/*.so; do
    [ -z "$so" ] || [ -f "$so" ] || continue
    n=$([ -z "$so" ] && echo product || basename $so .so)
    LEMO_HIP_LIB=$([ -z "$so" ] || echo $PWD/$so) timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras --concurrent-clips 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-10s %.1f it/s  %.1f us' % ('$n', d['value'], d['ms_per_step']*1e3))"
  done
done
