#!/bin/bash
# usage: pmc_frame.sh <outdir> <workload> <kernel substring> <options> <counters...>   per-dispatch counters of the last frame's matching dispatches
set -u
cd "$(dirname "$0")/.." || exit 1
O=$1; WL=$2; K=$3; OPTS=$4; shift 4
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc -- python3 tools/frame_latency.py --workload $WL --reps 3 --set "overlap=0,graphs=0,$OPTS" > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
f=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" "$K" <<'PY'
import csv,sys
from collections import OrderedDict
rows=list(csv.DictReader(open(sys.argv[1])))
d=OrderedDict()
for r in rows:
    if sys.argv[2] not in r['Kernel_Name']: continue
    d.setdefault(int(r['Dispatch_Id']),{})[r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(d)
# the last frame's dispatches: the last 4 (or fewer)
for i in ids[-4:]:
    print(i, {k:f"{v:.4g}" for k,v in d[i].items()})
PY
