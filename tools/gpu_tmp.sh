#!/bin/bash
# side-forward experiment: all-vertex forward on a parallel graph branch, forked after the pose stage (LEMO_SIDE_FORK=0)
# or before the per-frame backward kernels (=1), vs the default engine and vs the active-vertex engine, interleaved on one box
OUT=gpurun_out/r03l; mkdir -p $OUT
for i in 1 2; do
  timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --no-extras > $OUT/default_$i.json 2>> $OUT/bench.err
  LEMO_SIDE_FORK=0 timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --no-extras --side-forward > $OUT/side_early_$i.json 2>> $OUT/bench.err
  LEMO_SIDE_FORK=1 timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --no-extras --side-forward > $OUT/side_late_$i.json 2>> $OUT/bench.err
  timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --no-extras --active-vertices-only > $OUT/active_$i.json 2>> $OUT/bench.err
done
for f in $OUT/*.json; do python -c "
import json
d=json.load(open('$f'))
print('$f'.split('/')[-1], 'value %.1f it/s  (%.1f us/iteration)'%(d['value'], d['ms_per_step']*1e3), 'final loss', d['final_total_loss'])
"; done; tail -3 $OUT/bench.err
