#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run10; mkdir -p $O
timeout -k 10 200 python -u tools/frame_latency.py --workload config2 --set overlap=0 --set shadow_streams=1 --set shadow_streams=2 --set shadow_streams=1 --set shadow_streams=2 --set graphs=0,shadow_streams=2 > $O/lat.log 2>&1 || { tail -20 $O/lat.log; exit 1; }
cat $O/lat.log
