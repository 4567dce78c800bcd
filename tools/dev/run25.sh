#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run25; mkdir -p $O
bash tools/dev/trace_frame.sh gpurun_out/run18 config2 shadow_queue=1 || exit 1
timeout -k 10 300 python -u tools/frame_latency.py --workload config2 --set shadow_queue=1 --set shadow_queue=1 2>&1 | grep -v "texture\|amdgpu.ids" | tee $O/lat.log
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log
exit $rc
