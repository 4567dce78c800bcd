#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run7; mkdir -p $O; rm -f $O/ab.log
for v in "" _t8 _t32; do
  export RT_AMD_LIB=$PWD/raytracer_amd/librt_amd$v.so
  echo "== lib$v" >> $O/ab.log
  timeout -k 10 200 python -u tools/ab_option.py --workload config3 --flags="-r -rd 4" --set "" >> $O/ab.log 2>&1 || { echo ab3r failed; exit 1; }
  timeout -k 10 200 python -u tools/ab_option.py --workload config3 --flags="-ss -sss 16" --set "" >> $O/ab.log 2>&1 || { echo ab3s failed; exit 1; }
  timeout -k 10 300 python -u tools/ab_option.py --workload config5s --reps 2 --set "" >> $O/ab.log 2>&1 || { echo ab5 failed; tail -5 $O/ab.log; exit 1; }
done
grep -v "amdgpu.ids" $O/ab.log
