#!/bin/bash
# usage: tools/gpu_exp.sh <tag> <lib names...>: resblock forward microbenchmark under alternative builds of the library (build/lib_<name>.so), interleaved twice
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=$1; shift
for r in 1 2; do
  for L in BASE "$@"; do
    if [ $L = BASE ]; then unset ACGAN_HIP_LIB; else export ACGAN_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_$L.so; fi
    echo "== $L" >> gpurun_out/${TAG}.log
    timeout -k 10 120 python tools/microbench_conv.py --only resblock --what fwd --iters 30 --precision bf16x3 >> gpurun_out/${TAG}.log 2>&1 || exit 1
  done
done
grep -E "^==|resblock" gpurun_out/${TAG}.log
