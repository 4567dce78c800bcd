#!/bin/bash
# development: stage times of bench workloads under several builds of the library.  usage: LIBS="librt_amd librt_amd_x" tools/lib_ab.sh <workload>...
set -u
cd "$(dirname "$0")/.." || exit 1
for wl in "$@"; do
for lib in ${LIBS:-librt_amd}; do
echo "== $lib $wl"
RT_AMD_LIB=raytracer_amd/$lib.so timeout -k 10 200 python -u tools/ab_option.py --workload $wl --reps 3 2>&1 | grep -v "texture\|amdgpu.ids" || exit 1
done
done
