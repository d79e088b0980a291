#!/bin/bash
# usage: tools/ab_env.sh <tag> <rounds> VAR=a VAR=b ...: the bench step under each setting of an experiment variable, interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=$1; R=$2; shift 2
SUM='import sys,json
d=json.loads(sys.stdin.read()); p=d["roofline"]["passes"]
print(sys.argv[1], d["ms_per_step"], " ".join("%s %.4f" % (k, v["avg_launch_ms"]) for k, v in p.items()), "s2fwd %.4f" % d["roofline_hbm"]["avg_launch_ms"], flush=True)'
for i in $(seq 1 $R); do
for kv in "$@"; do
env ACG_DEBUG_SWITCHES=1 ACGAN_DEBUG_SWITCHES=1 $kv timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$SUM" $kv | tee -a gpurun_out/${TAG}_ab.log || exit 1
done
done
