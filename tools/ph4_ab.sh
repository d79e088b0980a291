#!/bin/bash
# A/B of the four-phases-in-one-tile kernel (stride-2 64->128 data gradient, ConvTranspose forward)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ACG_DEBUG_SWITCHES=1 ACG_NO_PH4=1 timeout -k 10 200 python tools/microbench_conv.py --precision bf16x3 --only a3_3x3s2,a6_convT,DB_4x4s2_64 --what fwd,dgrad 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python tools/microbench_conv.py --precision bf16x3 --only a3_3x3s2,a6_convT,DB_4x4s2_64 --what fwd,dgrad 2>&1 | grep -v amdgpu.ids
