#!/bin/bash
# SQ counters of the pre-split resblock kernels next to their fp32-operand twins (GPU box): two passes of 8 counters over
# tools/s16_check.py --time.  usage: tools/pmc_s16.sh <out.txt>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${1:-$R/gpurun_out/s16_counters.txt}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY"; do
  d=$R/gpurun_out/pmc_s16_$i; rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 $R/tools/s16_check.py --time --iters 3 > $R/gpurun_out/pmc_s16_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_s16_$i.log; exit 1; }
  i=$((i+1))
done
python3 - > $OUT <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("$R/gpurun_out/pmc_s16_0", "$R/gpurun_out/pmc_s16_1"):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            g = int(r['Grid_Size'])
            # the batch-32 launches only (4096 / 4225 tiles, 255 weight-gradient workgroups)
            if ('igemm_conv_x3' in k and g >= 4096 * 512) or ('wgrad_x3_krow' in k and g >= 255 * 512):
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
                agg[k]['dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print("# tools/pmc_s16.sh: SQ counters (sums over the chip, means over launches) of the resblock 3x3 128->128 kernels at batch 32,")
print("# 128x128 maps, under rocprofv3 --pmc (two passes of 8 counters): fp32-operand kernels (igemm_conv_x3_ws, wgrad_x3_krow)")
print("# and their pre-split twins (igemm_conv_x3_pre, wgrad_x3_krow_s16).  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = share of")
print("# conflicted LDS cycles; SQ_VALU_MFMA_BUSY_CYCLES / 1024 / (SQ_BUSY_CYCLES / 32) = matrix-pipe busy share per SIMD.")
for k, v in sorted(agg.items()):
    print(k)
    for c, xs in sorted(v.items()):
        print('   %-28s %.4g (n=%d)' % (c, sum(xs) / len(xs), len(xs)))
    if 'SQ_LDS_BANK_CONFLICT' in v and 'SQ_LDS_IDX_ACTIVE' in v:
        a, b = sum(v['SQ_LDS_BANK_CONFLICT']) / len(v['SQ_LDS_BANK_CONFLICT']), sum(v['SQ_LDS_IDX_ACTIVE']) / len(v['SQ_LDS_IDX_ACTIVE'])
        print('   -> conflicted LDS cycles   %.1f %%' % (100 * a / b))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'SQ_BUSY_CYCLES' in v:
        a, b = sum(v['SQ_VALU_MFMA_BUSY_CYCLES']) / len(v['SQ_VALU_MFMA_BUSY_CYCLES']), sum(v['SQ_BUSY_CYCLES']) / len(v['SQ_BUSY_CYCLES'])
        print('   -> matrix pipe busy        %.1f %%' % (100 * (a / 1024) / (b / 32)))
PY
cat $OUT
