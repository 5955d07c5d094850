"""Summarise rocprofv3 counter_collection CSVs (one per --pmc pass) per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections, json
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, '*.csv'))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get('Kernel_Name', '')
            short = name.split('(')[0].replace('void ', '')
            if 'lemo::' not in short:
                continue
            res[short][row['Counter_Name']].append(float(row['Counter_Value']))
summary = {}
for k, cs in sorted(res.items()):
    summary[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    summary[k]['dispatches'] = max(len(v) for v in cs.values())
json.dump(summary, open(os.path.join(out, 'pmc_summary.json'), 'w'), indent=1)
for k, v in summary.items():
    if any(x in k for x in ('conv3x3_mfma_v2', 'conv3x3_split', 'conv3x3_pair', 'lbs_verts_fwd', 'gemm_nt16', 'smooth_loss', 'enc_head', 'enc_tail')):
        print(k)
        for c, val in v.items():
            print('    %-28s %.4g' % (c, val))
