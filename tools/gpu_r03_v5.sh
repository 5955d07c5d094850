#!/bin/bash
# round 3, visit 5: the library without auto-vectorised packed fp32 (Makefile) -- race hunt (1500 trials), concurrent clips
# bit-identity, batched stage 1, bench (variant 4), rocprof kernel stats, the whole GPU suite.
TAG=${1:-r03e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit" | head -4 > $OUT/gpu.txt
timeout 400 python tools/race_hunt.py 1500 B > $OUT/race_hunt_nopk.txt 2>&1; tail -3 $OUT/race_hunt_nopk.txt | cut -c1-400
timeout 500 python tools/concurrent_clips.py 110 4 > $OUT/concurrent_clips.txt 2>&1; echo "rc=$?" >> $OUT/concurrent_clips.txt; grep -E "MISMATCH|side by side|rc=|solo" $OUT/concurrent_clips.txt | cut -c1-330
timeout 600 python tools/perframe_batched.py 6 100 > $OUT/perframe_batched.txt 2>&1; cat $OUT/perframe_batched.txt | grep -v amdgpu | cut -c1-300
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --concurrent-clips 0 > $OUT/bench_driver_style_$i.json 2>> $OUT/bench.err; done
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o prof -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --ramp-ms 100 > $R/$OUT/bench_prof.json 2> $R/$OUT/prof.err
cd $R
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/kernel_stats.csv; done
rm -rf $OUT/prof
for f in $OUT/bench_driver_style_*.json $OUT/bench_prof.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']
print('$f', 'variant', d['config']['conv_variant'], 'value %.1f'%d['value'], 'v100 %.1f'%d.get('value_100_steps',0), 'conv us %.2f (b2b %.2f) frac %.3f'%(r['kernel_ms']*1e3, r['kernel_ms_back_to_back']*1e3, r['frac']), 'lbs us %.1f'%(r['hbm']['kernel_ms']*1e3), 'loss', d['final_total_loss'])
"; done
head -25 $OUT/kernel_stats.csv | cut -c1-160
timeout 1700 python -m pytest tests -m gpu -q -s -x > $OUT/pytest_full.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_full.log | tail -5; grep -E "passed|failed|MPJPE|it/s|iterations/s|max rel|vs float64|module-API|per-frame|3 frames|PROX|finetuned|clip pipeline|step [0-9]|gradient|s per clip|ms per clip|eager launches|vertices vs|max err" $OUT/pytest_full.log > $OUT/pytest_gpu_measurements.txt; tail -c 6000 $OUT/pytest_full.log > $OUT/pytest_tail.log
