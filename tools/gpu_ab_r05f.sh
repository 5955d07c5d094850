#!/bin/bash
# Round-5 A/B visit: all-vertex LBS backward, joints with few entries per chunk summed by a thread per output (LBS_JOINT_SMALL = 64
# entries; 0 = a wave per joint as up to round 4; 32; 128), PROX window on the i.i.d.-joint model and on the one with SMPL-X's index locality
TAG=${1:-r05abjoint}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
val() { python -c "import json,sys; print(json.load(open(sys.argv[1]))['value'])" $1 2>/dev/null; }
timeout 700 python -m pytest tests/test_gpu_teacher.py tests/test_gpu_r2.py tests/test_gpu_parity.py -m gpu -q -k "prox or lbs or vertex_backward or body_model" > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | cut -c1-300
for i in 1 2; do
  for L in small0 default small32 small128; do
    LIB=""; [ $L != default ] && LIB=$R/lemo_amd/csrc/build_ab/$L.so
    LEMO_HIP_LIB=$LIB timeout 300 python bench.py --workload prox --steps 300 --warmup 100 > $OUT/prox_${L}_$i.json 2>> $OUT/err.log; echo "prox i.i.d. model, $L run $i: $(val $OUT/prox_${L}_$i.json)"
    LEMO_HIP_LIB=$LIB timeout 300 python bench.py --workload prox --model coherent --steps 300 --warmup 100 > $OUT/prox_coh_${L}_$i.json 2>> $OUT/err.log; echo "prox coherent model, $L run $i: $(val $OUT/prox_coh_${L}_$i.json)"
  done
done
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/pp -o p -- python $R/tools/prox_engine_prof.py S3 > $R/$OUT/prox_engine.txt 2>&1
cd $R
find $OUT/pp -name "*kernel_stats*" | head -n 1 | while read f; do cp "$f" $OUT/prox_kernel_stats.csv; done
rm -rf $OUT/pp
grep -E "lbs_bwd_chunk|gemm_nt16_splitk_bf16|prox_frame_dense|lbs_gemm_reduce" $OUT/prox_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,100-200
tail -n 2 $OUT/err.log
