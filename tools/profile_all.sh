#!/bin/bash
# every profiled workload of a round: tools/profile.sh per workload (trace + counter passes), one after the other
set -u
cd "$(dirname "$0")/.." || exit 1
TAG=${1:-r03}
for wl in config2 config3 config1 config4s config5s; do
  short=${wl#config}
  bash tools/profile.sh ${TAG}c${short} $wl > gpurun_out/prof_${TAG}c${short}.log 2>&1 || { echo "$wl failed"; tail -5 gpurun_out/prof_${TAG}c${short}.log; exit 1; }
  echo "== $wl"; sed -n 3,10p gpurun_out/prof_${TAG}c${short}/summary.txt
done
