#!/bin/bash
# Round-5 A/B visit: conv variant 9 (layers 2-0 backwards in the tail launch) against 8, and the PROX feature-gradient GEMM's operand ring
# depth / slab count.  gpurun --timeout 900 -- 'bash tools/gpu_ab_r05b.sh r05ab9'
TAG=${1:-r05ab9}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="--no-cpu-baseline --concurrent-clips 0 --no-extras"
val() { python -c "import json,sys; print(json.load(open(sys.argv[1]))['value'])" $1 2>/dev/null; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "encoder_full_size_golden and 9 or fit_full_size_golden and 9" > $OUT/pytest_v9.log 2>&1; tail -n 3 $OUT/pytest_v9.log
for i in 1 2 3; do
  for v in 8 9; do
    timeout 300 python bench.py --steps 100 --warmup 10 $B --conv-variant $v > $OUT/amass_v${v}_$i.json 2>> $OUT/err.log; echo "amass variant $v run $i: $(val $OUT/amass_v${v}_$i.json)"
  done
done
for i in 1 2; do
  for L in default sk5 sk7; do
    LIB=""; [ $L != default ] && LIB=$R/lemo_amd/csrc/build_ab/$L.so
    LEMO_HIP_LIB=$LIB timeout 300 python bench.py --workload prox --steps 300 --warmup 100 > $OUT/prox_${L}_$i.json 2>> $OUT/err.log; echo "prox $L run $i: $(val $OUT/prox_${L}_$i.json)"
  done
  LEMO_GEMM_SLABS=128 timeout 300 python bench.py --workload prox --steps 300 --warmup 100 > $OUT/prox_s128_$i.json 2>> $OUT/err.log; echo "prox slabs=128 run $i: $(val $OUT/prox_s128_$i.json)"
  LEMO_GEMM_SLABS=128 LEMO_HIP_LIB=$R/lemo_amd/csrc/build_ab/sk5.so timeout 300 python bench.py --workload prox --steps 300 --warmup 100 > $OUT/prox_s128_sk5_$i.json 2>> $OUT/err.log; echo "prox slabs=128 sk5 run $i: $(val $OUT/prox_s128_sk5_$i.json)"
  timeout 300 python bench.py --workload prox --steps 300 --warmup 100 --conv-variant 9 > $OUT/prox_v9_$i.json 2>> $OUT/err.log; echo "prox variant 9 run $i: $(val $OUT/prox_v9_$i.json)"
done
tail -n 5 $OUT/err.log
