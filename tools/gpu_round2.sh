#!/bin/bash
# Round-2 evidence run: everything profiles/r02_* is copied from.  One visit, one code state:
#   gpurun --timeout 2400 -- 'bash tools/gpu_round2.sh r02z'
TAG=${1:-r02z}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit" | head -4 > $OUT/gpu.txt
nproc >> $OUT/gpu.txt; lscpu | grep -E "Model name|^CPU\(s\)|Core|Socket" >> $OUT/gpu.txt
git -C $R rev-parse HEAD >> $OUT/gpu.txt 2>/dev/null
# 1. PMC passes (separate runs, kernel-trace only) on the same code
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc/$N -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --concurrent-clips 0 --no-graph --ramp-ms 0 > $R/$OUT/pmc_$N.log 2>&1
  find $R/$OUT/pmc/$N -name "*counter_collection.csv" | head -1 | while read f; do cp "$f" $R/$OUT/pmc/$N.csv; done
  rm -rf $R/$OUT/pmc/$N
done
cd $R
python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1
cp $OUT/pmc/pmc_summary.json $OUT/pmc_summary.json 2>/dev/null
# bench.py reads roofline.traffic from profiles/r02_pmc_summary.json: give it THIS visit's counters (box-local copy)
cp $OUT/pmc_summary.json profiles/r02_pmc_summary.json 2>/dev/null
rm -f $OUT/pmc/*.csv
# 2. the driver's own command, three times + the 100-step run with the CPU baseline
for i in 1 2 3; do timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --concurrent-clips 0 > $OUT/bench_driver_style_$i.json 2>> $OUT/bench.err; done
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2>> $OUT/bench.err
timeout 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 > $OUT/bench_100.json 2>> $OUT/bench.err
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --active-vertices-only > $OUT/bench_active.json 2>> $OUT/bench.err
# 3. kernel stats of the same command (rocprofv3 --kernel-trace --stats)
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o prof -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --ramp-ms 100 > $R/$OUT/bench_prof.json 2> $R/$OUT/prof.err
cd $R
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/kernel_stats.csv; done
find $OUT/prof -name "*domain_stats*" | head -1 | while read f; do cp "$f" $OUT/domain_stats.csv; done
rm -rf $OUT/prof
# 4. PROX engine: throughput + kernel stats
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/pp -o p -- python $R/tools/prox_engine_prof.py S3 > $R/$OUT/prox_engine.txt 2>&1
cd $R
find $OUT/pp -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/prox_kernel_stats.csv; done
rm -rf $OUT/pp
timeout 120 python tools/prox_engine_prof.py S2 2>&1 | grep PROX >> $OUT/prox_engine.txt
# 4b. what the box sustains (bf16 MFMA rate, HBM stream), k clips side by side, AE training-step kernel table
./tools/ubench/peak_ubench > $OUT/peaks.txt 2>&1
timeout 300 python tools/concurrent_clips.py 100 4 2>&1 | grep -E "clip|aggregate" > $OUT/concurrent_clips.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ae -o p -- python $R/tools/ae_prof.py > /dev/null 2>&1
cd $R
find $OUT/ae -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/ae_kernel_stats.csv; done
rm -rf $OUT/ae
# 5. the whole GPU suite with its printed measurements, the float64 gates, LBS census
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_full.log 2>&1; grep -E "^FAILED|^ERROR" $OUT/pytest_full.log > $OUT/pytest_failures.txt; grep -E "passed|failed|MPJPE|it/s|iterations/s|max rel|vs float64|module-API|per-frame|3 frames|PROX|finetuned|clip pipeline|step [0-9]|gradient|s per clip|ms per clip|eager launches|vertices vs" $OUT/pytest_full.log > $OUT/pytest_gpu_measurements.txt; tail -c 20000 $OUT/pytest_full.log > $OUT/pytest_tail.log; rm -f $OUT/pytest_full.log
timeout 120 python tools/lbs_census.py 2>&1 | grep blocks > $OUT/lbs_census.txt
timeout 600 python tools/r02_gates.py 2>&1 | grep -v "amdgpu\|Warn\|float(\|detach" > $OUT/gates.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2 > $OUT/smoke.txt
cat $OUT/bench_driver_style.json; tail -3 $OUT/pytest_gpu_measurements.txt; grep PROX $OUT/prox_engine.txt; cat $OUT/smoke.txt
