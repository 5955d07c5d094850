#!/bin/bash
# round 3, visit 1: the stream-ordering fix on hardware (with and without the engine's events), the new configs[0] gates,
# and a driver-style bench line with the in-iteration roofline fields.
TAG=${1:-r03a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit" | head -4 > $OUT/gpu.txt
timeout 900 python -m pytest tests/test_gpu_r2.py -q -s -k "perframe or concurrent" > $OUT/pytest_new.log 2>&1; tail -5 $OUT/pytest_new.log
LEMO_UNORDERED=1 timeout 400 python tools/concurrent_clips.py 100 4 > $OUT/concurrent_unordered.txt 2>&1; echo "unordered rc=$?" >> $OUT/concurrent_unordered.txt
timeout 400 python tools/concurrent_clips.py 100 4 > $OUT/concurrent_ordered.txt 2>&1; echo "ordered rc=$?" >> $OUT/concurrent_ordered.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --concurrent-clips 0 > $OUT/bench_driver_style.json 2> $OUT/bench.err
grep -E "MISMATCH|side by side|rc=" $OUT/concurrent_unordered.txt $OUT/concurrent_ordered.txt | cut -c1-260
cat $OUT/bench_driver_style.json; tail -3 $OUT/bench.err
