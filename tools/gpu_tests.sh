#!/bin/bash
# full GPU parity suite (no -x: report every failure)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-tests}
timeout -k 10 1700 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gputest.log 2>&1
rc=$?
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_gputest.log | tail -20
exit $rc
