#!/bin/bash
# usage: tools/gpu_ab.sh <ENV_SWITCH> [tag] [notests]: GPU parity suite, then an interleaved A/B of one environment switch on the bench step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SW=$1; TAG=${2:-ab}
if [ "$3" != "notests" ]; then
timeout -k 10 1500 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_gputest.log 2>&1
rc=$?
tail -6 gpurun_out/${TAG}_gputest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
if [ $rc -ne 0 ]; then exit $rc; fi
fi
for i in 1 2; do
env ACG_DEBUG_SWITCHES=1 ACGAN_DEBUG_SWITCHES=1 $SW=1 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$SW=1', d['ms_per_step'])" || exit 1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'])" || exit 1
done
