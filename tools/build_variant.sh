#!/bin/bash
# A/B builds: tools/build_variant.sh <name> <unit[,unit...]> [extra hipcc flags...]
#   -> hair-gs_amd/libhgs_<name>.so = the current objects with csrc/<unit>.hip (each listed unit) recompiled with the extra flags
# (load it with HGS_LIB=... python bench.py ...; boxes differ run to run, variants are only comparable inside one gpurun call)
set -e
HERE=$(cd "$(dirname "$0")" && pwd); CS=$HERE/../hair-gs_amd/csrc
name=$1; units=${2//,/ }; shift 2
make -s -C $CS
objs=$(ls $CS/build/*.o)
for unit in $units; do
  contract=-ffp-contract=fast-honor-pragmas
  case $unit in hgs_api|hgs_preprocess|hgs_binning|hgs_knn) contract=-ffp-contract=off;; esac
  noslp=
  case $unit in hgs_blend|hgs_losses|hgs_preprocess) noslp=-fno-slp-vectorize;; esac     # (as csrc/Makefile)
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-fast-math -munsafe-fp-atomics -Wall -Wno-unused-function $contract $noslp "$@" -c $CS/$unit.hip -o /tmp/variant_${name}_$unit.o
  objs=$(echo $objs | tr ' ' '\n' | grep -v "/$unit.o\$" | tr '\n' ' ')" /tmp/variant_${name}_$unit.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $HERE/../hair-gs_amd/libhgs_$name.so
echo built libhgs_$name.so
