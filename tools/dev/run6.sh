#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run6; mkdir -p $O; rm -f $O/ab.log
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_soft_shadows.py -m gpu -x -q --timeout 300 --timeout-method thread -k "boxes_per_time or large_scene or sorted or crowded or all_effects" > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log
[ $rc -eq 0 ] || { echo "tests failed rc=$rc"; exit 1; }
for v in; do
  export RT_AMD_LIB=$PWD/raytracer_amd/librt_amd$v.so
  echo "== lib$v" >> $O/ab.log
  timeout -k 10 200 python -u tools/ab_option.py --workload config2 --set "" >> $O/ab.log 2>&1 || { echo ab2 failed; tail -5 $O/ab.log; exit 1; }
done
export RT_AMD_LIB=$PWD/raytracer_amd/librt_amd.so
timeout -k 10 200 python -u tools/ab_option.py --workload config3 --flags="-r -rd 4" --set "" >> $O/ab.log 2>&1 || { echo ab3r failed; exit 1; }
timeout -k 10 200 python -u tools/ab_option.py --workload config3 --flags="-ss -sss 16" --set "" >> $O/ab.log 2>&1 || { echo ab3s failed; exit 1; }
timeout -k 10 300 python -u tools/ab_option.py --workload config5s --reps 2 --set "" >> $O/ab.log 2>&1 || { echo ab5 failed; tail -5 $O/ab.log; exit 1; }
grep -v "amdgpu.ids" $O/ab.log
export RT_AMD_LIB=$PWD/raytracer_amd/librt_amd_ws.so
timeout -k 10 300 python -u tools/walk_stats.py --workload config5s > $O/ws_config5s.log 2>&1 || { echo ws5 failed; tail -5 $O/ws_config5s.log; exit 1; }
grep -v "amdgpu.ids" $O/ws_config5s.log
