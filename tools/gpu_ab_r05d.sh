#!/bin/bash
# Round-5 A/B visit: PROX window with the fused tail launch (last VPoser backward layer + Adam + next first VPoser layer) against the
# same build with LEMO_PROX_SEPARATE_ADAM=1 and against the round-4 launch structure (build_ab/old.so).
TAG=${1:-r05abtail}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
val() { python -c "import json,sys; print(json.load(open(sys.argv[1]))['value'])" $1 2>/dev/null; }
timeout 700 python -m pytest tests/test_gpu_teacher.py tests/test_gpu_r2.py tests/test_gpu_parity.py -m gpu -q -s -k "prox" > $OUT/pytest_prox.log 2>&1; grep -E "passed|failed" $OUT/pytest_prox.log | cut -c1-400
for i in 1 2 3; do
  LEMO_HIP_LIB=$R/lemo_amd/csrc/build_ab/old.so timeout 300 python bench.py --workload prox --steps 300 --warmup 100 > $OUT/prox_old_$i.json 2>> $OUT/err.log; echo "prox previous build run $i: $(val $OUT/prox_old_$i.json)"
  timeout 300 python bench.py --workload prox --steps 300 --warmup 100 > $OUT/prox_new_$i.json 2>> $OUT/err.log; echo "prox new run $i: $(val $OUT/prox_new_$i.json)"
  LEMO_PROX_SEPARATE_ADAM=1 timeout 300 python bench.py --workload prox --steps 300 --warmup 100 > $OUT/prox_new_separate_adam_$i.json 2>> $OUT/err.log; echo "prox new, Adam / first and last VPoser layers as launches of their own run $i: $(val $OUT/prox_new_separate_adam_$i.json)"
  timeout 300 python bench.py --workload prox --model coherent --steps 300 --warmup 100 > $OUT/prox_new_coherent_$i.json 2>> $OUT/err.log; echo "prox new, coherent model run $i: $(val $OUT/prox_new_coherent_$i.json)"
done
tail -n 3 $OUT/err.log
