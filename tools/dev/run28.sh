#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run28; mkdir -p $O
run() { echo "== $1 share=$2"; RT_BENCH_SHARE=$2 timeout -k 10 300 python bench.py --workload $1 --no-cpu-baseline --no-scaling-4k --no-variants > $O/b.json 2> $O/b.err || { tail -3 $O/b.err; return 1; }; python3 -c "
import json; d=json.load(open('$O/b.json')); print(round(d['ms_per_step'],4), round(d['frame_ms_alone'],4), d['config']['frames_in_flight'], d['config']['image_crc32'])"; }
for wl in config3 config3u config2 bvh5 config1; do run $wl 0 && run $wl 1 || exit 1; done
