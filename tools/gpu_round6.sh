#!/bin/bash
# Round-6 evidence run: everything profiles/r06_* (final) is copied from.  One visit, one code state:
#   gpurun --timeout 3300 -- 'bash tools/gpu_round6.sh r06final'     then     python tools/copy_evidence.py r06final r06
TAG=${1:-r06final}; OUT=gpurun_out/$TAG; mkdir -p $OUT/pmc; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | tail -6 > $OUT/gpu.txt
nproc >> $OUT/gpu.txt; lscpu | grep -E "Model name|^CPU\(s\)|Core|Socket" >> $OUT/gpu.txt
git -C $R rev-parse HEAD >> $OUT/gpu.txt 2>/dev/null
# 1. PMC passes (separate runs, kernel-trace only) -> roofline.traffic of THIS code state
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc/$N -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --concurrent-clips 0 --no-extras --no-graph --ramp-ms 0 > $R/$OUT/pmc_$N.log 2>&1
  find $R/$OUT/pmc/$N -name "*counter_collection.csv" | head -1 | while read f; do cp "$f" $R/$OUT/pmc/$N.csv; done
  rm -rf $R/$OUT/pmc/$N
done
cd $R
python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1
cp $OUT/pmc/pmc_summary.json $OUT/pmc_summary.json 2>/dev/null
cp $OUT/pmc_summary.json profiles/r06_pmc_summary.json 2>/dev/null      # bench.py reads roofline.traffic from it (box-local copy)
rm -f $OUT/pmc/*.csv $OUT/pmc_*.log
# 2. the driver's own command (everything: stage census, variants, MPJPE, extras, concurrent clips, CPU baseline), then short ones
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2>> $OUT/bench.err
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --concurrent-clips 0 --no-extras > $OUT/bench_driver_style_$i.json 2>> $OUT/bench.err; done
for i in 1 2; do
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --no-extras > $OUT/bench_100_$i.json 2>> $OUT/bench.err
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --no-extras --conv-variant 10 > $OUT/bench_100_variant10_winograd_layers_$i.json 2>> $OUT/bench.err
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --no-extras --conv-variant 5 > $OUT/bench_100_variant5_pairs_without_fused_head_tail_$i.json 2>> $OUT/bench.err
done
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --no-extras --active-vertices-only > $OUT/bench_active.json 2>> $OUT/bench.err
timeout 400 python bench.py --workload prox --gpus 1 --steps 300 --warmup 100 > $OUT/bench_prox.json 2>> $OUT/bench.err
# 3. kernel stats of the same command (rocprofv3 --kernel-trace --stats), AMASS (default and Winograd variant) and PROX
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o prof -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --no-extras --ramp-ms 100 > $R/$OUT/bench_prof.json 2> $R/$OUT/prof.err
cd $R
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/kernel_stats.csv; done
rm -rf $OUT/prof
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o prof -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --no-extras --ramp-ms 100 --conv-variant 10 > $R/$OUT/bench_prof10.json 2> $R/$OUT/prof.err
cd $R
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/kernel_stats_variant10_winograd.csv; done
rm -rf $OUT/prof $OUT/bench_prof10.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/pp -o p -- python $R/tools/prox_engine_prof.py S3 > $R/$OUT/prox_engine.txt 2>&1
cd $R
find $OUT/pp -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/prox_kernel_stats.csv; done
rm -rf $OUT/pp
# 4. diagnostics: pair and Winograd census, per-layer numerics, concurrent clips (bit-identity asserted), race hunt (required gate)
timeout 300 python tools/pair_check.py > $OUT/pair_check.txt 2>&1
timeout 300 python tools/wino_check.py > $OUT/wino_check.txt 2>&1
timeout 600 python tools/enc_layer_numerics.py > $OUT/enc_layer_numerics.txt 2>&1
timeout 500 python tools/concurrent_clips.py 110 4 > $OUT/concurrent_clips.txt 2>&1; echo "rc=$?" >> $OUT/concurrent_clips.txt
timeout 400 python tools/race_hunt.py 1000 B > $OUT/race_hunt.txt 2>&1
timeout 300 python tools/ae_clips.py 16 > $OUT/ae_clips.txt 2>&1
timeout 300 python tools/clip_pipeline_rate.py 16 > $OUT/clip_pipeline_rate.txt 2>&1
# 5. the whole GPU suite with its printed measurements, smoke
timeout 3000 python -m pytest tests -m gpu -q -s > $OUT/pytest_full.log 2>&1; grep -E "^FAILED|^ERROR" $OUT/pytest_full.log > $OUT/pytest_failures.txt
grep -v "^\"void" $OUT/pytest_full.log | grep -E "passed|failed|MPJPE|it/s|iterations/s|max rel|vs float64|module-API|per-frame|3 frames|PROX|finetuned|clip pipeline|gradient|s per clip|ms per clip|eager launches|vertices vs|max err|kink|marker residual|split conv|fused pair|Winograd|free-running" > $OUT/pytest_gpu_measurements.txt
grep -v "^\"void" $OUT/pytest_full.log | grep -E "teacher-forced|next-state|amass\[|S[23]_w[01]|median-over|replayed steps" > $OUT/teacher.txt
grep -v "^\"void" $OUT/pytest_full.log | grep -E "^seed |^   |^      |^  [0-9] |^  variant" > $OUT/gates.txt
grep -v "^\"void" $OUT/pytest_full.log | tail -c 20000 > $OUT/pytest_tail.log; rm -f $OUT/pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2 > $OUT/smoke.txt
python - <<PY
import json
d=json.load(open('$OUT/bench_driver_style.json')); r=d['roofline']
print('value', d['value'], 'v100', d.get('value_100_steps'), 'pair us', r['kernel_ms']*1e3, 'frac', r['frac'], 'traffic', r['traffic'], 'mpjpe', d.get('mpjpe_mm'))
print('stage_us', d.get('stage_us')); print('variants', {k: v for k, v in d.get('variants', {}).items() if k != 'note'})
for k in ('prox_window','perframe','ae_finetune','concurrent_clips','cpu_baseline'):
    print(k, {kk: vv for kk, vv in d.get(k, {}).items() if kk in ('value','unit','error','one_clip_value','cores','bit_identical_to_solo','autograd_path_ms','side_by_side_ms_per_clip')})
for f in ('bench_100_1','bench_100_variant10_winograd_layers_1','bench_100_variant5_pairs_without_fused_head_tail_1','bench_100_2','bench_100_variant10_winograd_layers_2','bench_100_variant5_pairs_without_fused_head_tail_2','bench_active','bench_prox'):
    try:
        e=json.load(open('$OUT/'+f+'.json')); print(f, e['value'])
    except Exception as ex: print(f, 'ERR', ex)
PY
tail -3 $OUT/pytest_gpu_measurements.txt; cat $OUT/pytest_failures.txt; grep -E "rc=" $OUT/concurrent_clips.txt | cut -c1-200; tail -2 $OUT/race_hunt.txt; cat $OUT/smoke.txt | cut -c1-200
