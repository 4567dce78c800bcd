#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run16; mkdir -p $O
timeout -k 10 200 python -u tools/ab_option.py --workload config2 --reps 20 --set shadow_queue=0 --set shadow_queue=1 --set shadow_queue=1,grid_shadow=2048 --set shadow_queue=1,grid_shadow=1024 --set shadow_queue=1,grid_shadow=2048,grid_shadow_drain=1024 --set shadow_queue=1,grid_shadow=2048,grid_shadow_drain=4096 2>&1 | grep -v "texture\|amdgpu.ids" | tee $O/ab.log
timeout -k 10 300 python -u tools/frame_latency.py --workload config2 --set shadow_queue=0 --set shadow_queue=0 --set shadow_queue=1 --set shadow_queue=1,grid_shadow=2048 --set shadow_queue=1,grid_shadow=1024 --set shadow_queue=1,grid_shadow=1024,grid_shadow_drain=1024 --set shadow_queue=1,grid_shadow=4096 2>&1 | grep -v "texture\|amdgpu.ids" | tee $O/lat.log
