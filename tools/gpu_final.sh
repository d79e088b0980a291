#!/bin/bash
# final measurement pass of a round (GPU box).  usage: tools/gpu_final.sh <tag> bench|profile
#   bench:   bench lines of configs 3 (default line, with cpu_baseline), 2, 5 and config 3 in exact fp32
#   profile: per-kernel statistics + per-shape table of the step, HBM traffic of the dominant kernel, layer table (all rows)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r04}
if [ "$2" = "bench" ]; then
  timeout -k 10 700 python bench.py > gpurun_out/${TAG}_bench_cfg3.json 2> gpurun_out/${TAG}_bench_cfg3.err || exit 1
  cut -c1-400 gpurun_out/${TAG}_bench_cfg3.json
  timeout -k 10 300 python bench.py --config 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg2.json 2> gpurun_out/${TAG}_bench_cfg2.err || exit 1
  timeout -k 10 300 python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg5.json 2> gpurun_out/${TAG}_bench_cfg5.err || exit 1
  timeout -k 10 300 python bench.py --precision f32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg3_f32.json 2> gpurun_out/${TAG}_bench_cfg3_f32.err || exit 1
  exit 0
fi
bash tools/profile_stats.sh bf16x3 $TAG > gpurun_out/${TAG}_stats.log 2>&1 || exit 1
python3 tools/kernel_stats_md.py gpurun_out/${TAG}_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.md 3 bf16x3
tr=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/kernel_shapes_md.py $tr gpurun_out/${TAG}_kernel_shapes.md 3
echo "stats done"
bash tools/profile_traffic.sh > gpurun_out/${TAG}_traffic.log 2>&1 || exit 1
python3 tools/summarize_traffic.py 'igemm_conv_x3_pre<true, true, false>' gpurun_out/${TAG}_resblock_conv_traffic.json
echo "traffic done"
bash tools/profile_layer_traffic.sh stem,a2,a3,a6,a7,a8,DB_4x4 $TAG > gpurun_out/${TAG}_layers.log 2>&1 || exit 1
head -12 gpurun_out/${TAG}_kernel_stats.md
