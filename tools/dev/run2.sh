#!/bin/bash
# development run: large-scene parity tests on the new walk, then old (w3 build) vs new library on the large workloads
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run2; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "second_level or hierarchy or sorted or device_built or boxes_per_time or large_scene or degenerate or one_kernel or rand2k or rand300 or test2" > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log
[ $rc -eq 0 ] || { echo "tests failed rc=$rc"; exit 1; }
for v in _w3 ""; do
  export RT_AMD_LIB=$PWD/raytracer_amd/librt_amd$v.so
  echo "== lib$v" >> $O/ab.log
  timeout -k 10 200 python -u tools/ab_option.py --workload config3 --set "" >> $O/ab.log 2>&1 || { echo ab3 failed; tail -5 $O/ab.log; exit 1; }
  timeout -k 10 200 python -u tools/ab_option.py --workload config3u --set "" >> $O/ab.log 2>&1 || { echo ab3u failed; exit 1; }
  timeout -k 10 200 python -u tools/ab_option.py --workload config3 --flags="-r -rd 4" --set "" >> $O/ab.log 2>&1 || { echo ab3r failed; exit 1; }
  timeout -k 10 300 python -u tools/ab_option.py --workload config5s --reps 2 --set "" >> $O/ab.log 2>&1 || { echo ab5 failed; tail -5 $O/ab.log; exit 1; }
done
grep -v "amdgpu.ids" $O/ab.log
