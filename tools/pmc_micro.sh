#!/bin/bash
# SQ counters for one conv shape (GPU box), two passes of <= 8 counters.
# usage: tools/pmc_micro.sh <shape-substring> <what: fwd|dgrad|wgrad> [precision]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${3:-bf16x3}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD"; do
  d=$R/gpurun_out/pmc_micro$i; rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 $R/tools/microbench_conv.py --only $1 --what $2 --precision $P --iters 3 > $R/gpurun_out/pmc_micro$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("$R/gpurun_out/pmc_micro0", "$R/gpurun_out/pmc_micro1"):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'conv' in k or 'wgrad' in k:
                agg[k.split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
                agg[k.split('(')[0]]['dur_us'].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k, v in agg.items():
    print(k)
    for c, xs in sorted(v.items()):
        print('   %-28s %.4g (n=%d)' % (c, sum(xs) / len(xs), len(xs)))
PY
