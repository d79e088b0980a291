#!/bin/bash
# usage: tools/ab_rounds.sh <tag> <other tree (a checkout of another commit with its library built, e.g. build/r05_tree)> [rounds]
# the bench step of that tree and of this one, interleaved on ONE box (boxes of the pool differ by up to 3 %)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=$1; OTHER=$2; R=${3:-2}
SUM='import sys,json
d=json.loads(sys.stdin.read()); p=d["roofline"]["passes"]
print(sys.argv[1], d["ms_per_step"], " ".join("%s %.4f" % (k, v["avg_launch_ms"]) for k, v in p.items()), "s2fwd %.4f" % d["roofline_hbm"]["avg_launch_ms"], flush=True)'
for i in $(seq 1 $R); do
(cd $OTHER && timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>>$GRAFT_REPO_ROOT/gpurun_out/${TAG}_other.err) | python -c "$SUM" other | tee -a gpurun_out/${TAG}_rounds.log || exit 1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>>gpurun_out/${TAG}_this.err | python -c "$SUM" this | tee -a gpurun_out/${TAG}_rounds.log || exit 1
done
