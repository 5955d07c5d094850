#!/bin/bash
# Round-6 visit F: the all-vertex forward on the engine's side stream (lemo_fit_desc.verts_side): GPU parity test, then the headline A/B, three interleaved runs
TAG=${1:-r06f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --concurrent-clips 0 --no-extras"
val() { python -c "import json,sys; print(json.load(open(sys.argv[1]))['value'])" $1 2>/dev/null; }
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "side_full_forward" > $OUT/pytest_side.log 2>&1; tail -n 6 $OUT/pytest_side.log
for i in 1 2 3; do
  for v in 0 1; do
    LEMO_SIDE_FULL_FORWARD=$v timeout 300 python bench.py --steps 100 --warmup 10 $B > $OUT/amass_side${v}_$i.json 2>> $OUT/err.log; echo "side forward $v run $i: $(val $OUT/amass_side${v}_$i.json)"
  done
done
timeout 300 python bench.py --steps 100 --warmup 10 $B --active-vertices-only > $OUT/amass_active.json 2>> $OUT/err.log; echo "active vertices only: $(val $OUT/amass_active.json)"
tail -n 5 $OUT/err.log
