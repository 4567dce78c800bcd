#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/run9; mkdir -p $O
export TMPDIR=/tmp RT_BENCH_OPTS=overlap=0,graphs=0
ARGS="--workload config5s --steps 2 --warmup 0 --no-cpu-baseline --no-scaling-4k --no-variants --frames-in-flight 1"
timeout -s KILL 200 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc_ta" -- python3 bench.py $ARGS > "$O/pmc_ta.log" 2>&1 || { echo "ta pass failed"; tail -5 "$O/pmc_ta.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d "$O/pmc_tcp" -- python3 bench.py $ARGS > "$O/pmc_tcp.log" 2>&1 || { echo "tcp pass failed"; tail -5 "$O/pmc_tcp.log"; }
python3 - <<'PY'
import csv, glob
from collections import defaultdict
for d in ("pmc_ta", "pmc_tcp"):
    for f in glob.glob(f"gpurun_out/run9/{d}/**/*counter_collection.csv", recursive=True):
        agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            k = "k_shadow" if "k_shadow" in k else "k_extend<false>" if "k_extendILb0" in k or "k_extend<false" in k else "k_extend<true>" if "k_extend" in k else None
            if not k: continue
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
        for k in agg:
            print(d, k, len(n[k]), {c: f"{v/len(n[k]):.4g}" for c, v in agg[k].items()})
PY
