#!/bin/bash
# rocprofv3 per-kernel statistics of the bench step (GPU box).  usage: tools/profile_stats.sh <precision> <tag>
# leaves gpurun_out/<tag>_kernel_stats.csv (copy the ones to keep into profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}
p=${1:-bf16x3}; tag=${2:-stats_$p}
d=$R/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o $tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-step-graph --precision $p > $R/gpurun_out/${tag}_bench.log 2>&1 || exit 1
f=$(find $d -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv && head -16 $R/gpurun_out/${tag}_kernel_stats.csv | cut -c1-200
