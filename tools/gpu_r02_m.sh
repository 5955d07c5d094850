#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/${1:-r02m}; mkdir -p $R/$O
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/ae -o p -- python $R/tools/ae_prof.py > $R/$O/ae_prof.log 2>&1
cd $R; f=$(find $O/ae -name "*kernel_stats.csv" | head -1); python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms over 21 passes: %.2f  (%.3f ms per pass)' % (tot/1e6, tot/1e6/21))
for r in rows[:26]:
    print('%-64s calls %5s avg %8.2f us tot %7.2f ms' % (r['Name'][:64], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
P
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r2.py -m gpu -q -s -k "infill or finetune or clip_pipeline" 2>&1 | grep -E "passed|failed|Error|error|finetune steps|eager launches|rror" | tail -8
