#!/bin/bash
# development: per-dispatch counters of one kernel while tools/ab_option.py renders a workload under one option set.
# usage: pmc_ab.sh <outdir> <kernel substring> "<counters>" <ab_option.py arguments...>    (prints the mean over the matching dispatches)
set -u
cd "$(dirname "$0")/.." || exit 1
O=$1; K=$2; CTR=$3; shift 3
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --pmc $CTR --output-format csv -d $O/pmc -- python3 tools/ab_option.py --reps 1 "$@" > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
f=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" "$K" <<'PY'
import csv,sys
from collections import OrderedDict
d=OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r['Kernel_Name']: continue
    d.setdefault(int(r['Dispatch_Id']),{})[r['Counter_Name']]=float(r['Counter_Value'])
n=len(d); tot={}
for v in d.values():
    for k,x in v.items(): tot[k]=tot.get(k,0)+x
print(n, "dispatches; sum:", {k:f"{v:.4g}" for k,v in sorted(tot.items())})
PY
rm -rf $O/pmc
