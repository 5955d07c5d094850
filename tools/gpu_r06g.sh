#!/bin/bash
# Round-6 visit G: side-stream all-vertex forward: workgroup-count / priority sweep + a kernel trace of the overlap
TAG=${1:-r06g}; R=$GRAFT_REPO_ROOT; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --concurrent-clips 0 --no-extras"
val() { python -c "import json,sys; print(json.load(open(sys.argv[1]))['value'])" $1 2>/dev/null; }
timeout 300 python bench.py --steps 100 --warmup 10 $B > $OUT/a.json 2>> $OUT/err.log; echo "in line: $(val $OUT/a.json)"
for blocks in 250 125 84 63 32; do
  LEMO_FIT_SIDE_BLOCKS=$blocks LEMO_SIDE_FULL_FORWARD=1 timeout 300 python bench.py --steps 100 --warmup 10 $B > $OUT/b.json 2>> $OUT/err.log; echo "side, $blocks workgroups, lowest priority: $(val $OUT/b.json)"
done
LEMO_FIT_SIDE_PRIO=0 LEMO_SIDE_FULL_FORWARD=1 timeout 300 python bench.py --steps 100 --warmup 10 $B > $OUT/b.json 2>> $OUT/err.log; echo "side, 125 workgroups, priority 0: $(val $OUT/b.json)"
timeout 300 python bench.py --steps 100 --warmup 10 $B > $OUT/a.json 2>> $OUT/err.log; echo "in line: $(val $OUT/a.json)"
cd /tmp && LEMO_SIDE_FULL_FORWARD=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/tr -o t -- python $R/bench.py --steps 40 --warmup 10 $B --ramp-ms 50 > /dev/null 2>> $R/$OUT/err.log
cd $R
f=$(find $OUT/tr -name "*kernel_trace.csv" | head -n 1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'smplx_pose_fwd' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b + 3]:
    n = r['Kernel_Name'].split('(')[0].replace('void lemo::', '').replace('lemo::', '')[:44]
    print('%-46s start %8.2f us  end %8.2f us  (%.2f)  queue %s' % (n, (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3,
                                                        (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?')))
PY
rm -rf $OUT/tr
