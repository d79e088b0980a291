#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]: an alternative build of libacgan_hip.so as build/lib_<name>.so
# (objects in build/obj_<name>/), loaded through ACGAN_HIP_LIB for A/B timings and -DACG_STAMP diagnostics.
# An object is rebuilt when its source or any header the Makefile lists is newer (or when the flags differ from the
# last build of this variant); a failed compile stops the script before the link.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
SRC=domain-transfer-gan_amd/csrc
OBJ=build/obj_$NAME
mkdir -p $OBJ
SRCS=$(sed -n 's/^SRCS *:= *//p' $SRC/Makefile)
DEPS="$SRC/common.h $SRC/conv_internal.h $SRC/conv_wgrad_tr_s16.inc include/acgan_hip.h"
if [ "$(cat $OBJ/.flags 2>/dev/null)" != "$*" ]; then rm -f $OBJ/*.o; echo "$*" > $OBJ/.flags; fi
pids=()
for f in $SRCS; do
  o=$OBJ/${f%.hip}.o
  stale=0
  [ -f $o ] || stale=1
  for d in $SRC/$f $DEPS; do [ $stale = 1 ] || [ ! $d -nt $o ] || stale=1; done
  if [ $stale = 1 ]; then
    rm -f $o
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $SRC/$f -o $o &
    pids+=($!)
    if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]} || exit 1; pids=("${pids[@]:1}"); fi
  fi
done
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -ldl -o build/lib_$NAME.so
ls -la build/lib_$NAME.so
