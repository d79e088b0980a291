#!/bin/bash
# A/B of the generic tile's row-patch stages: output checksums must agree bit for bit, timings side by side
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=a2_3x3,a3_3x3s2,a6_convT,a7_3x3,DB_4x4s2_64
ACG_DEBUG_SWITCHES=1 ACG_NO_RP=1 timeout -k 10 200 python tools/microbench_conv.py --precision bf16x3 --only $L --what fwd,dgrad --digest > gpurun_out/rp_off.log 2>&1 || exit 1
timeout -k 10 200 python tools/microbench_conv.py --precision bf16x3 --only $L --what fwd,dgrad --digest > gpurun_out/rp_on.log 2>&1 || exit 1
grep -v digest gpurun_out/rp_off.log; grep -v digest gpurun_out/rp_on.log
diff <(grep digest gpurun_out/rp_off.log | sed 's/\[.*//') <(grep digest gpurun_out/rp_on.log | sed 's/\[.*//') && echo "DIGESTS IDENTICAL"
grep digest gpurun_out/rp_on.log | grep -c "RP=1"
if [ -f build/lib_nostore.so ]; then
  echo "no-store ablation:"
  ACGAN_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_nostore.so timeout -k 10 200 python tools/microbench_conv.py --precision bf16x3 --only $L --what fwd,dgrad 2>&1 | grep -v amdgpu.ids
fi
