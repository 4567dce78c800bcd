#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run12; mkdir -p $O
RT_AMD_LIB=raytracer_amd/librt_amd_hist.so timeout -k 10 120 python -u tools/dev/hist.py "-r -rd 4 -ss -sss 16" 2>&1 | grep -v texture
timeout -k 10 200 python -u tools/frame_latency.py --workload config2 --set cull_umbra=0 --set cull_umbra=1 --set cull_umbra=1,fast_samples=0 --set cull_umbra=0,fast_samples=0 > $O/lat.log 2>&1 || { tail -20 $O/lat.log; exit 1; }
cat $O/lat.log
timeout -k 10 200 python -u tools/frame_latency.py --workload config2 --flags="-r -rd 4 -ss -sss 16 -u" --set cull_umbra=0 --set cull_umbra=1 >> $O/lat.log 2>&1 || { tail -20 $O/lat.log; exit 1; }
tail -2 $O/lat.log
timeout -k 10 600 python -u -m pytest tests/test_gpu_soft_shadows.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log
exit $rc
