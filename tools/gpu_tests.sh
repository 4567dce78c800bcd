#!/bin/bash
# the whole GPU suite in one process, every test under a time limit; the log stays under gpurun_out/
set -u
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/gpu_tests; mkdir -p $O
timeout -k 10 1100 python -u -m pytest tests -m gpu -q --timeout 400 --timeout-method thread > $O/tests.log 2>&1; rc=$?
tail -40 $O/tests.log
exit $rc
