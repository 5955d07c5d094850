#!/bin/bash
# PMC passes of the headline iteration (separate runs, kernel-trace only, as MI355X_MICROARCH.md prescribes):
#   gpurun --timeout 1500 -- 'bash tools/gpu_pmc.sh r04pmc'
TAG=${1:-r04pmc}; OUT=gpurun_out/$TAG; mkdir -p $OUT/pmc; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc/$N -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --concurrent-clips 0 --no-extras --no-graph --ramp-ms 0 > $R/$OUT/pmc_$N.log 2>&1
  find $R/$OUT/pmc/$N -name "*counter_collection.csv" | head -1 | while read f; do cp "$f" $R/$OUT/pmc/$N.csv; done
  rm -rf $R/$OUT/pmc/$N
done
cd $R
python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1
cp $OUT/pmc/pmc_summary.json $OUT/pmc_summary.json 2>/dev/null
rm -f $OUT/pmc/*.csv
grep -A14 "conv3x3_pair_kernel<0" $OUT/pmc_summary.txt | head -40; grep -c . $OUT/pmc_summary.txt; tail -3 $OUT/pmc_SQ_WAIT*.log | cut -c1-200
