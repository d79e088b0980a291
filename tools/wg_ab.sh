#!/bin/bash
# A/B of the three-taps-per-workgroup weight gradient (stride-2 64->128 / ConvTranspose layers)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=${1:-a3_3x3s2,a6_convT}
ACG_DEBUG_SWITCHES=1 ACG_NO_WGRAD_NT=1 timeout -k 10 200 python tools/microbench_conv.py --precision bf16x3 --only $L --what wgrad --digest 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python tools/microbench_conv.py --precision bf16x3 --only $L --what wgrad --digest 2>&1 | grep -v amdgpu.ids
