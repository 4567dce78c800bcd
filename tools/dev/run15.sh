#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run15; mkdir -p $O
for lib in librt_amd librt_amd_v1 librt_amd_v2; do
echo "== $lib"
RT_AMD_LIB=raytracer_amd/$lib.so timeout -k 10 200 python -u tools/ab_option.py --workload config2 --reps 20 --set grid_shadow=0 --set grid_shadow=1024 2>&1 | grep -v "texture\|amdgpu.ids" | tee -a $O/ab.log || exit 1
done
