#!/bin/bash
# usage: tools/profile_step.sh <tag>: full GPU parity suite is NOT run here; rocprofv3 kernel trace + stats of three bench
# steps, the per-kernel and per-shape tables, and one un-profiled bench line -> gpurun_out/<tag>_*
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r03}
cd $R && mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_line.json 2>gpurun_out/${tag}_bench.err || exit 1
bash tools/profile_stats.sh bf16x3 $tag > /dev/null || exit 1
python tools/kernel_stats_md.py gpurun_out/${tag}_kernel_stats.csv gpurun_out/${tag}_kernel_stats.md 3 bf16x3
tr=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
python tools/kernel_shapes_md.py $tr gpurun_out/${tag}_kernel_shapes.md 3
head -45 gpurun_out/${tag}_kernel_stats.md
python -c "import json; d=json.load(open('gpurun_out/${tag}_bench_line.json')); print(d['ms_per_step'], d['roofline']['three_pass_aggregate'])"
