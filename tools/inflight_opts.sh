#!/bin/bash
# Development aid: the default bench workload with several values of one library option, frames in flight as bench.py does.
# usage: tools/inflight_opts.sh <workload> <option> <value> [<value> ...]
WL=$1; OPT=$2; shift 2
for v in "$@"; do
  RT_BENCH_OPTS=$OPT=$v timeout -k 10 300 python3 bench.py --workload "$WL" --no-cpu-baseline --no-scaling-4k --no-variants 2>/dev/null > /tmp/_b.json || exit 1
  python3 -c "
import json
d = json.load(open('/tmp/_b.json'))
print('$OPT=$v', 'in flight', d['config']['frames_in_flight'], 'ms_per_step %.4f' % d['ms_per_step'], 'alone %.4f' % d['frame_ms_alone'], 'Grays/s %.2f' % (d['value'] / 1e9))"
done
