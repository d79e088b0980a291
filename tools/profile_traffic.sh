#!/bin/bash
# Run on the GPU box (via gpurun): HBM traffic counters for the headline workload, collected in their own
# passes (FETCH_SIZE and WRITE_SIZE need separate --pmc runs on gfx950; never combined with trace domains
# other than --kernel-trace).  Output: gpurun_out/pmc_{fetch,write}/*.csv
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc_$(echo $c | tr A-Z a-z | sed 's/_size//')
  rm -rf $d
  timeout -k 10 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-step-graph > $R/gpurun_out/pmc_$c.log 2>&1
  echo "$c done: $(ls $d/*/ | tr '\n' ' ')"
done
