#!/bin/bash
# visit r04b: fused pairs (conv variant 5) -- parity, A/B against variant 4, census, kernel stats
TAG=${1:-r04b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "fused_pair or fit_full_size_golden" > $OUT/pytest_pair.log 2>&1; echo "pytest pair rc=$?" > $OUT/rc.txt
timeout 300 python tools/pair_check.py > $OUT/pair_check.txt 2>&1
for i in 1 2; do
  for v in 5 4; do timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --no-extras --conv-variant $v > $OUT/bench_100_v${v}_$i.json 2>> $OUT/bench.err; done
done
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --concurrent-clips 0 --no-extras > $OUT/bench_driver_style_1.json 2>> $OUT/bench.err
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o prof -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --no-extras --ramp-ms 100 > $R/$OUT/bench_prof.json 2> $R/$OUT/prof.err
cd $R
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/kernel_stats.csv; done
rm -rf $OUT/prof
timeout 900 python -m pytest tests/test_gpu_teacher.py -m gpu -q -s > $OUT/pytest_teacher.log 2>&1; echo "pytest teacher rc=$?" >> $OUT/rc.txt
grep -v "^\"void" $OUT/pytest_teacher.log | grep -E "next-state|amass\[|perframe f|S[23]_w|median-over|passed|failed|^FAILED" > $OUT/teacher.txt
cat $OUT/rc.txt; grep -E "passed|failed|^FAILED|fused pair" $OUT/pytest_pair.log | tail -8; cat $OUT/pair_check.txt
python - <<PY
import json
for f in ('bench_100_v5_1','bench_100_v4_1','bench_100_v5_2','bench_100_v4_2','bench_driver_style_1'):
    try:
        e=json.load(open('$OUT/'+f+'.json')); r=e['roofline']; print(f, round(e['value'],1), 'kernel us', round(r['kernel_ms']*1e3,2), 'frac', round(r['frac'],3))
    except Exception as ex: print(f, 'ERR', ex)
PY
head -12 $OUT/kernel_stats.csv | cut -c1-160; tail -5 $OUT/teacher.txt
