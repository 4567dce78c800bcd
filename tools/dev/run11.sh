#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
export RT_AMD_LIB=raytracer_amd/librt_amd_hist.so
timeout -k 10 120 python -u tools/dev/hist.py "-r -rd 4 -ss -sss 16" 2>&1 | grep -v texture
