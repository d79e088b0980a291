#!/bin/bash
# one gpurun call: GPU parity suite, then (only if nothing was killed) bench lines and the per-layer traffic profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02a_gputest.log 2>&1
rc=$?
tail -5 gpurun_out/r02a_gputest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 600 python bench.py > gpurun_out/r02a_bench_cfg3.json 2> gpurun_out/r02a_bench_cfg3.err || exit 1
cat gpurun_out/r02a_bench_cfg3.json
timeout -k 10 300 python bench.py --config 2 --no-cpu-baseline > gpurun_out/r02a_bench_cfg2.json 2> gpurun_out/r02a_bench_cfg2.err || exit 1
cat gpurun_out/r02a_bench_cfg2.json
timeout -k 10 300 python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a_bench_cfg5.json 2> gpurun_out/r02a_bench_cfg5.err || exit 1
cat gpurun_out/r02a_bench_cfg5.json
timeout -k 10 900 bash tools/profile_layer_traffic.sh stem,a2,a3,a6,a7,a8,DB_4x4s2 r02a bf16x3 > gpurun_out/r02a_layer_traffic.log 2>&1 || exit 1
tail -30 gpurun_out/r02a_layer_traffic.log
