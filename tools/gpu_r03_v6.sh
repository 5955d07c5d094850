#!/bin/bash
# visit 6: whole GPU suite (no -x) on the round's default build; same-box A/B of the auto-vectorised (packed fp32) build vs
# the default one; PMC passes for roofline.traffic
TAG=${1:-r03f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  timeout 200 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 > $OUT/ab_default_$i.json 2>> $OUT/bench.err
  LEMO_HIP_LIB=$PWD/lemo_amd/csrc/build_ab/liblemo_hip_slp.so timeout 200 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 > $OUT/ab_slp_$i.json 2>> $OUT/bench.err
done
for f in $OUT/ab_*.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']
print('$f', 'value %.1f'%d['value'], 'conv us %.2f (b2b %.2f)'%(r['kernel_ms']*1e3, r['kernel_ms_back_to_back']*1e3), 'lbs us %.1f'%(r['hbm']['kernel_ms']*1e3))
"; done
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc/$N -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --concurrent-clips 0 --no-graph --ramp-ms 0 > $R/$OUT/pmc_$N.log 2>&1
  find $R/$OUT/pmc/$N -name "*counter_collection.csv" | head -1 | while read f; do cp "$f" $R/$OUT/pmc/$N.csv; done
  rm -rf $R/$OUT/pmc/$N
done
cd $R
python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1
cp $OUT/pmc/pmc_summary.json $OUT/pmc_summary.json 2>/dev/null; rm -f $OUT/pmc/*.csv
grep -E "conv3x3_split_kernel<0, 64, 64|lbs_verts_fwd" $OUT/pmc_summary.txt | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -q -s > $OUT/pytest_full.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_full.log | tail -15 | cut -c1-300; grep -E "passed|failed|MPJPE|it/s|iterations/s|max rel|vs float64|module-API|per-frame|3 frames|PROX|finetuned|clip pipeline|step [0-9]|gradient|s per clip|ms per clip|eager launches|vertices vs|max err|kink|marker residual" $OUT/pytest_full.log > $OUT/pytest_gpu_measurements.txt; grep -v "^\"void" $OUT/pytest_full.log | tail -c 30000 > $OUT/pytest_tail.log; rm -f $OUT/pytest_full.log
