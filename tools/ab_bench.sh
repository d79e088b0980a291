#!/bin/bash
# usage: tools/ab_bench.sh <tag> <other lib name> [rounds] [bench args...]: the bench step interleaved between the in-tree library
# and build/lib_<name>.so (loaded through ACGAN_HIP_LIB), printing ms/step and the per-pass launch times of the resblock layer
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=$1; OTHER=$2; R=${3:-2}; shift $(( $# < 3 ? $# : 3 ))
SUM='import sys,json
d=json.loads(sys.stdin.read()); p=d["roofline"]["passes"]
print(sys.argv[1], d["ms_per_step"], " ".join("%s %.4f" % (k, v["avg_launch_ms"]) for k, v in p.items()), "s2fwd %.4f" % d["roofline_hbm"]["avg_launch_ms"], flush=True)'
for i in $(seq 1 $R); do
ACGAN_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_$OTHER.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>>gpurun_out/${TAG}_${OTHER}.err | python -c "$SUM" $OTHER | tee -a gpurun_out/${TAG}_ab.log || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>>gpurun_out/${TAG}_intree.err | python -c "$SUM" in-tree | tee -a gpurun_out/${TAG}_ab.log || exit 1
done
