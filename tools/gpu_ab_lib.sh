#!/bin/bash
# usage: tools/gpu_ab_lib.sh <tag> <other lib name> [pytest -k expr]: op/net tests on the in-tree library, then the bench step
# interleaved between it and build/lib_<name>.so (an alternative build of the same library, loaded through ACGAN_HIP_LIB)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=$1; OTHER=$2; K=$3
if [ -n "$K" ]; then
  timeout -k 10 900 python -m pytest tests/test_hip_ops.py tests/test_hip_nets.py tests/test_hip_step.py -m gpu -q -x -k "$K" > gpurun_out/${TAG}_test.log 2>&1 || { tail -20 gpurun_out/${TAG}_test.log; exit 1; }
  tail -2 gpurun_out/${TAG}_test.log
fi
for i in 1 2; do
ACGAN_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_$OTHER.so timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$OTHER', d['ms_per_step'])" || exit 1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in-tree', d['ms_per_step'])" || exit 1
done
