#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run17; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/frame_latency.py --workload config2 --reps 5 --set overlap=0,graphs=0 > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_begin_frame' in r['Kernel_Name']]
s=idx[-2]; e=idx[-1]
t0=int(rows[s]['Start_Timestamp'])
for r in rows[s:e]:
    n=r['Kernel_Name'][:70]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  grid {r.get('Grid_Size_X','?'):>8} {n}")
PY
