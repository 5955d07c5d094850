#!/bin/bash
# PMC passes for the dominant kernels (separate runs, kernel-trace only -- see MI355X_MICROARCH.md).
TAG=${1:-pmc}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$N -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $OUT/$N.log 2>&1
  find $OUT/$N -name "*counter_collection.csv" | head -1 | while read f; do cp "$f" $OUT/$N.csv; done
  rm -rf $OUT/$N
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT | tee $OUT/summary.txt
