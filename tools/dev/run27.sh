#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
timeout -k 10 300 python -u tools/frame_latency.py --workload config2 --reps 400 --set shadow_queue_deep=1 --set shadow_queue_deep=0 --set shadow_queue_deep=1 --set shadow_queue_deep=0 --set grid_secondary=512 --set grid_shade=512 2>&1 | grep -v "texture\|amdgpu.ids"
