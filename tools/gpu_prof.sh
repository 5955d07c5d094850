#!/bin/bash
# parity tests + bench + rocprof kernel stats (no CPU baseline): quick per-kernel picture
TAG=${1:-prof}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $OUT/pytest_gpu.log
timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2>> $OUT/bench.err
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/prof.err
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/kernel_stats.csv; done
rm -rf $OUT/prof
python - <<PY
import csv, json
d=json.load(open('$OUT/bench.json')); print('bench', round(d['value'],1),'it/s', round(d['ms_per_step']*1e3,1),'us  conv', round(d['roofline']['kernel_ms']*1e3,2),'us')
rows=list(csv.DictReader(open('$OUT/kernel_stats.csv')))
for r in rows[:24]:
    if 'at::native' in r['Name'] or 'rocclr' in r['Name']: continue
    n=int(r['Calls']); per=round(n/35.0)
    print('%-64s x%d  %7.2f us  -> %6.1f us/iter' % (r['Name'].replace('void ','').replace('lemo::','')[:64], per, float(r['AverageNs'])/1e3, float(r['AverageNs'])/1e3*per))
PY
tail -2 $OUT/pytest_gpu.log
