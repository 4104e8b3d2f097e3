#!/bin/bash
# raytrace_amd/librt_amd_head.so with FILE.hip as committed (HEAD) and every other object from the working tree's last build —
# the "before" side of tools/abn.sh while FILE.hip is being edited.
set -e
cd "$(dirname "$0")/.."
src=${1:-rt_paths.hip}
C=raytrace_amd/csrc
git show HEAD:$C/$src > $C/_head_$src
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I include -c $C/_head_$src -o /tmp/variant_head.o
rm -f $C/_head_$src
objs=""
for f in rt_kernels rt_persist rt_paths rt_post rt_api; do
  if [ "$f.hip" = "$src" ]; then objs="$objs /tmp/variant_head.o"; else objs="$objs $C/$f.hip.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o raytrace_amd/librt_amd_head.so $objs
echo raytrace_amd/librt_amd_head.so
