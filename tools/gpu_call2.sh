#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -q > gpurun_out/r02b_gputest.log 2>&1
rc=$?
tail -15 gpurun_out/r02b_gputest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
bash tools/profile_stats.sh bf16x3 r02b > gpurun_out/r02b_stats.log 2>&1 || exit 1
python3 tools/kernel_stats_md.py gpurun_out/r02b_kernel_stats.csv gpurun_out/r02b_kernel_stats.md 3 bf16x3
head -60 gpurun_out/r02b_kernel_stats.md
