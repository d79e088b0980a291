#!/bin/bash
# SQ counters for one conv shape (GPU box).  usage: tools/pmc_micro.sh <shape-substring> <what>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
d=$R/gpurun_out/pmc_micro; rm -rf $d
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $d -- python3 $R/tools/microbench_conv.py --only $1 --what $2 --iters 3 > $R/gpurun_out/pmc_micro.log 2>&1
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$d/*/*counter_collection.csv")[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'igemm' in r['Kernel_Name'] or 'wgrad_f32' in r['Kernel_Name']:
        agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[r['Kernel_Name'].split('(')[0]]['dur_us'].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k, v in agg.items():
    print(k)
    for c, xs in v.items():
        print('   %-28s %.4g (n=%d)' % (c, sum(xs) / len(xs), len(xs)))
PY
