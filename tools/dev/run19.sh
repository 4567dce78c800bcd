#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run19; mkdir -p $O
timeout -k 10 300 python -u tools/frame_latency.py --workload config2 --set shadow_queue=0,cull_umbra=0 --set shadow_queue=0 --set shadow_queue=1 --set shadow_queue=1,shadow_streams=1 --set shadow_queue=1,overlap=0 --set shadow_queue=1,fast_samples=0 2>&1 | grep -v "texture\|amdgpu.ids" | tee $O/lat.log
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log
exit $rc
