#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run14; mkdir -p $O
timeout -k 10 300 python -u tools/frame_latency.py --workload config2 --set grid_shadow=0 --set grid_shadow=1024 --set grid_shadow=1280 --set grid_shadow=2048 --set grid_shadow=2560 --set grid_shadow=4096 --set grid_shadow=16384 --set grid_shadow=0 2>&1 | grep -v "texture\|amdgpu.ids" | tee $O/lat.log
