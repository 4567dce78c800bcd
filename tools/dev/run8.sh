#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run8; mkdir -p $O; rm -f $O/ab.log
timeout -k 10 600 python -u tools/ab_option.py --workload config5s --reps 2 --set "" --set "grid_shadow=768,grid_secondary=768,grid_primary=768" --set "grid_shadow=1536,grid_secondary=1536,grid_primary=1536" --set "grid_shadow=3072,grid_secondary=3072,grid_primary=3072" --set "grid_shadow=6144,grid_secondary=6144" --set "grid_shadow=16384,grid_secondary=16384,grid_primary=16384" >> $O/ab.log 2>&1 || { echo ab5 failed; tail -5 $O/ab.log; exit 1; }
timeout -k 10 300 python -u tools/ab_option.py --workload config3 --flags="-r -rd 4" --set "" --set "grid_shadow=768,grid_secondary=768,grid_primary=768" --set "grid_shadow=1536,grid_secondary=1536,grid_primary=1536" --set "grid_shadow=3072,grid_secondary=3072,grid_primary=3072"  >> $O/ab.log 2>&1 || { echo ab3 failed; tail -5 $O/ab.log; exit 1; }
timeout -k 10 300 python -u tools/ab_option.py --workload config3 --set "" --set "grid_primary=768" --set "grid_primary=1536" --set "grid_primary=3072" --set "grid_primary=6144" >> $O/ab.log 2>&1 || { echo ab3 failed; tail -5 $O/ab.log; exit 1; }
grep -v "amdgpu.ids" $O/ab.log
