#!/bin/bash
TAG=${1:-r03d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/race_hunt.py 400 B > $OUT/race_hunt_default.txt 2>&1; tail -4 $OUT/race_hunt_default.txt | cut -c1-600
LEMO_HIP_LIB=$PWD/lemo_amd/csrc/build_ab/liblemo_hip_noslp.so timeout 300 python tools/race_hunt.py 400 B > $OUT/race_hunt_noslp.txt 2>&1; tail -4 $OUT/race_hunt_noslp.txt | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r2.py -q -s -k "split_f16 or split_bf16 or encoder_full_size or fit_full_size_golden or perframe" > $OUT/pytest_sel.log 2>&1; grep -E "max err|per-frame|marker residual|3 frames|passed|failed|^FAILED|Error" $OUT/pytest_sel.log | cut -c1-400
for v in 4 3 4 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --concurrent-clips 0 --conv-variant $v > $OUT/bench_v${v}_$RANDOM.json 2>> $OUT/bench.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/'+'TAGX'.replace('TAGX','') + '*/bench_v*.json')): pass
PY
for f in $OUT/bench_v*.json; do python -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']
print('$f', 'variant', d['config']['conv_variant'], 'value %.1f'%d['value'], 'v100 %.1f'%d.get('value_100_steps',0), 'conv us %.2f (b2b %.2f) frac %.3f'%(r['kernel_ms']*1e3, r['kernel_ms_back_to_back']*1e3, r['frac']), 'lbs us %.1f'%(r['hbm']['kernel_ms']*1e3), 'loss', d['final_total_loss'])
"; done; tail -3 $OUT/bench.err
