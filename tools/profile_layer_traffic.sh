#!/bin/bash
# Run on the GPU box (via gpurun): HBM traffic AND achieved GB/s of single convolution layers (fwd / dgrad / wgrad),
# from tools/microbench_conv.py --mark: one un-profiled run for the HIP-event times, then one rocprofv3 pass per counter
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; --pmc only together with --kernel-trace).
# usage: tools/profile_layer_traffic.sh <only: comma list of layer-name substrings> <tag> [precision]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ONLY=$1; TAG=$2; P=${3:-bf16x3}
python3 $R/tools/microbench_conv.py --only $ONLY --precision $P --iters 10 --mark --json $R/gpurun_out/layers_$TAG.json > $R/gpurun_out/layers_$TAG.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc_layers_${TAG}_$c
  rm -rf $d
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/tools/microbench_conv.py --only $ONLY --precision $P --iters 3 --mark > $R/gpurun_out/pmc_layers_${TAG}_$c.log 2>&1
done
python3 $R/tools/summarize_layer_traffic.py $TAG $R/gpurun_out/layer_traffic_$TAG.json
