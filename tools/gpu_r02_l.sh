#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r02l}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gates.py -m gpu -q -x -s 2>&1 | grep -E "small problem|B=119|passed|failed|Error|assert" | tee $O/gates.log | tail -12
for rep in 1 2; do
  for so in "" lemo_amd/csrc/build_ab/prev.so; do
    n=$([ -z "$so" ] && echo product || basename $so .so)
    LEMO_HIP_LIB=$([ -z "$so" ] || echo $PWD/$so) timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --concurrent-clips 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %.1f it/s  %.1f us' % ('$n', d['value'], d['ms_per_step']*1e3))"
  done
done | tee $O/ab.log
# (as first written -- no timeout, default rocpd output instead of csv, eager launches incl. the 250 ms ramp -- this line ran into the
# call's 1500 s limit and cost 25 GPU-minutes: ALWAYS `timeout`, `--output-format csv`, and a short `--ramp-ms` under the profiler)
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --concurrent-clips 0 --ramp-ms 100 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -24 $f | cut -c1-60,150-260 > $O/kernel_stats_head.txt; python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print('%-60s calls %5s avg %8.2f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
P
