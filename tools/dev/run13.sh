#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run13; mkdir -p $O
timeout -k 10 200 python -u tools/ab_option.py --workload config2 --reps 20 --set cull_umbra=0 --set cull_umbra=1 --set cull_cones=0 2>&1 | grep -v "texture\|amdgpu.ids" | tee $O/ab.log
