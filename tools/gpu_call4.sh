#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02d_gputest.log 2>&1
rc=$?
tail -8 gpurun_out/r02d_gputest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
for i in 1 2; do
ACGAN_NO_NORM_MASK=1 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-mask', d['ms_per_step'])" || exit 1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mask', d['ms_per_step'])" || exit 1
done
