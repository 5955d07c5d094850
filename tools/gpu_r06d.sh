#!/bin/bash
# Round-6 visit D: PMC passes over the infilling-AE finetune (8 clips per engine), split-f16 vs fp32 convolutions: where do the waves' cycles go
TAG=${1:-r06d}; R=$GRAFT_REPO_ROOT; OUT=gpurun_out/$TAG; mkdir -p $OUT/pmc; export TMPDIR=/tmp
for a in f16 fp32; do
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    cd /tmp && LEMO_AE_ARITH=$a timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc/${a}_$i -o p -- python $R/tools/ae_prof_k8.py 8 > $R/$OUT/pmc_${a}_$i.log 2>&1
    cd $R
    find $OUT/pmc/${a}_$i -name "*counter_collection.csv" | head -1 | while read f; do cp "$f" $OUT/pmc/${a}_$i.csv; done
    rm -rf $OUT/pmc/${a}_$i
  done
done
python - <<'PY'
import csv, glob, collections, os
out = os.path.join('gpurun_out', os.environ.get('TAG', 'r06d'), 'pmc')
for a in ('f16', 'fp32'):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(out, a + '_*.csv'))):
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'].split('(')[0].replace('void ', '').replace('lemo::', '')
            res[k][row['Counter_Name']].append(float(row['Counter_Value']))
    print('==', a)
    for k, cs in sorted(res.items()):
        if 'conv' not in k and 'wgrad' not in k: continue
        print(k, 'dispatches', max(len(v) for v in cs.values()))
        print('    ' + '  '.join('%s %.3g' % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
rm -f $OUT/pmc/*.csv
