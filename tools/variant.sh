#!/bin/bash
# Build a variant of librt_amd.so for tools/ab.sh: tools/variant.sh NAME FILE.hip [-DMACRO=..]... — recompiles one device source
# with extra flags and links it with the objects of the last `python -m raytrace_amd.build` into raytrace_amd/librt_amd_NAME.so.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
C=raytrace_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I include "$@" -c $C/$src -o /tmp/variant_$name.o
objs=""
for f in rt_kernels rt_persist rt_paths rt_frame rt_post rt_api; do
  if [ "$f.hip" = "$src" ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs $C/$f.hip.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o raytrace_amd/librt_amd_$name.so $objs
echo raytrace_amd/librt_amd_$name.so
