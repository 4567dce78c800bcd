#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run26; mkdir -p $O
run() { echo "== P=$1 OPTS=$2"; RT_BENCH_OPTS="$2" timeout -k 10 200 python bench.py --no-cpu-baseline --no-scaling-4k --no-variants --frames-in-flight $1 > $O/b.json 2> $O/b.err || { tail -3 $O/b.err; return 1; }; python3 -c "
import json; d=json.load(open('$O/b.json')); print(round(d['ms_per_step'],4), round(d['frame_ms_alone'],4), d['config']['frames_in_flight'])"; }
run 4 "" && run 3 "" && run 6 "" && run 8 "" && run 4 "grid_shadow=1280" && run 4 "grid_shadow=2048" && run 4 "grid_shadow=1024,grid_shadow_drain=1024" && run 4 "grid_shade=2048" && run 4 "overlap=1" && run 2 ""
