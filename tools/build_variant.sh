#!/bin/bash
# Builds an experimental variant of libakz.so (extra -D flags on ONE translation unit) into gpurun_variants/<name>/libakz.so
# so that a single gpurun call can A/B several builds: `cp gpurun_variants/<name>/libakz.so cv_amd/lib/libakz.so` on the box.
# usage: tools/build_variant.sh <name> <source.hip> <flags...>
set -e
R=$(cd $(dirname $0)/.. && pwd)
N=$1; SRC=$2; shift 2
mkdir -p $R/gpurun_variants/$N
# (akz_scale_space.hip exists once per arithmetic combination, cv_amd/build.py: the experiment replaces the default copy)
TAG=""; [ "$SRC" == "akz_scale_space.hip" ] && TAG="_a0"
O=$R/gpurun_variants/$N/$(basename $SRC | tr . _)$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "$@" -x hip -c $R/cv_amd/csrc/$SRC -o $O
OBJS=""
for f in $R/cv_amd/lib/*.o; do
  if [ "$(basename $f)" == "$(basename $O)" ]; then OBJS="$OBJS $O"; else OBJS="$OBJS $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_variants/$N/libakz.so $OBJS -ldl
rm -f $O
ls -la $R/gpurun_variants/$N/libakz.so
