#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run3; mkdir -p $O
rc=0
tail -5 $O/tests.log
[ $rc -eq 0 ] || { echo "tests failed rc=$rc"; exit 1; }
for v in ""; do
  export RT_AMD_LIB=$PWD/raytracer_amd/librt_amd$v.so
  echo "== lib$v" >> $O/ab.log
  timeout -k 10 200 python -u tools/ab_option.py --workload config3 --set "" >> $O/ab.log 2>&1 || { echo ab3 failed; tail -5 $O/ab.log; exit 1; }
  timeout -k 10 200 python -u tools/ab_option.py --workload config3u --set "" >> $O/ab.log 2>&1 || { echo ab3u failed; exit 1; }
  timeout -k 10 200 python -u tools/ab_option.py --workload config3 --flags="-r -rd 4" --set "" >> $O/ab.log 2>&1 || { echo ab3r failed; exit 1; }
  timeout -k 10 300 python -u tools/ab_option.py --workload config5s --reps 2 --set "" >> $O/ab.log 2>&1 || { echo ab5 failed; tail -5 $O/ab.log; exit 1; }
done
grep -v "amdgpu.ids" $O/ab.log
export RT_AMD_LIB=$PWD/raytracer_amd/librt_amd_ws.so
timeout -k 10 200 python -u tools/walk_stats.py --workload config3 > $O/ws_config3.log 2>&1 || { echo ws3 failed; tail -5 $O/ws_config3.log; exit 1; }
timeout -k 10 300 python -u tools/walk_stats.py --workload config5s > $O/ws_config5s.log 2>&1 || { echo ws5 failed; tail -5 $O/ws_config5s.log; exit 1; }
grep -v "amdgpu.ids" $O/ws_config3.log $O/ws_config5s.log
