#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]: an alternative build of libacgan_hip.so as build/lib_<name>.so
# (objects in build/obj_<name>/), loaded through ACGAN_HIP_LIB for A/B timings and -DACG_STAMP diagnostics
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
SRC=domain-transfer-gan_amd/csrc
OBJ=build/obj_$NAME
mkdir -p $OBJ
SRCS=$(sed -n 's/^SRCS *:= *//p' $SRC/Makefile)
pids=()
for f in $SRCS; do
  if [ ! -f $OBJ/${f%.hip}.o ] || [ $SRC/$f -nt $OBJ/${f%.hip}.o ] || [ $SRC/conv_internal.h -nt $OBJ/${f%.hip}.o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $SRC/$f -o $OBJ/${f%.hip}.o &
    pids+=($!)
    if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -ldl -o build/lib_$NAME.so
ls -la build/lib_$NAME.so
