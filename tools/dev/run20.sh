#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run20; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_soft_shadows.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 900 python -u tools/fuzz_cones.py --cases 1500 --seed 5 2>&1 | grep -v "texture\|amdgpu.ids" > $O/fuzz.log; rc=$?
tail -25 $O/fuzz.log
exit $rc
