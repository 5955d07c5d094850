#!/bin/bash
# visit 8: strip-tiled conv (this tree) vs the 1-D-tiled kernel (build_ab/liblemo_hip_slp.so = round state before the tiling), same box
TAG=${1:-r03h}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/split_check.py > $OUT/split_check.txt 2>&1; grep -v amdgpu $OUT/split_check.txt | cut -c1-300
for i in 1 2; do
  timeout 200 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --no-extras > $OUT/ab_strip_$i.json 2>> $OUT/bench.err
  LEMO_HIP_LIB=$PWD/lemo_amd/csrc/build_ab/liblemo_hip_slp.so timeout 200 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --no-extras > $OUT/ab_1d_$i.json 2>> $OUT/bench.err
done
for f in $OUT/ab_*.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']
print('$f', 'value %.1f'%d['value'], 'conv us %.2f (b2b %.2f)'%(r['kernel_ms']*1e3, r['kernel_ms_back_to_back']*1e3), 'lbs us %.1f'%(r['hbm']['kernel_ms']*1e3), 'loss', d['final_total_loss'])
"; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r2.py -q -s -k "split or encoder_full_size or fit_full_size_golden or 245x115 or prox_engine_baseline" > $OUT/pytest_sel.log 2>&1; grep -E "max err|passed|failed|^FAILED|Error|PROX" $OUT/pytest_sel.log | cut -c1-300
