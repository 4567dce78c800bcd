#!/bin/bash
# usage: trace_frame.sh <outdir> <workload> <option set>...   per-dispatch kernel times of the last frame under each option set
set -u
cd "$(dirname "$0")/.." || exit 1
O=$1; WL=$2; shift 2
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
n=0
for v in "$@"; do
n=$((n+1))
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace$n -- python3 tools/frame_latency.py --workload $WL --reps 5 --set "overlap=0,graphs=0,$v" > $O/trace$n.log 2>&1 || { tail -5 $O/trace$n.log; exit 1; }
f=$(find $O/trace$n -name "*kernel_trace.csv" | head -1)
echo "== $v"
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_begin_frame' in r['Kernel_Name']]
s=idx[-2]; e=idx[-1]
t0=int(rows[s]['Start_Timestamp'])
out=[]
for r in rows[s:e]:
    n=r['Kernel_Name']
    n=n.replace('void (anonymous namespace)::','').replace('(anonymous namespace)::','')[:28]
    out.append(f"{n}:{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:.1f}")
print("  ".join(out)); print("total", (int(rows[e-1]['End_Timestamp'])-t0)/1e3)
PY
done
