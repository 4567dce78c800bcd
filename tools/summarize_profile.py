"""Condenses a tools/profile.sh output directory into one text summary (per-kernel time and counters)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(f"{root}/{pattern}", recursive=True))


def short(name):
    if "k_primary_shaded" in name:
        return "k_primary_shaded"
    for k in ("k_extend<true", "k_extend<false", "k_shadow", "k_shade", "k_finalize", "k_assemble"):
        if k in name:
            return k + (">" if k.startswith("k_extend") else "")
    return name.split("(")[0][-60:]


print("== kernel trace (per dispatch durations, ns) ==")
for f in find("trace/**/*kernel_trace.csv"):
    agg = defaultdict(list)
    for row in csv.DictReader(open(f)):
        agg[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    tot = sum(sum(v) for v in agg.values())
    print(f"{'kernel':28s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'share':>7s}")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:28s} {len(v):7d} {sum(v)/1e3:12.1f} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f} {sum(v)/tot:7.1%}")
for f in find("trace/**/*kernel_stats.csv"):
    print("\n== rocprofv3 --stats ==")
    print(open(f).read().strip())

traffic = {}
print("\n== counters (sum over dispatches of each kernel; per-dispatch average in brackets) ==")
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_l2"):
    for f in find(f"{d}/**/*counter_collection.csv"):
        agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[k].add(row["Dispatch_Id"])
        for k in sorted(agg):
            n = len(calls[k])
            print(f"{d:10s} {k:24s} calls {n:5d} " + "  ".join(f"{c}={v:.4g} [{v/n:.4g}]" for c, v in sorted(agg[k].items())))
            for c, v in agg[k].items():
                traffic.setdefault(k, {})[c] = v / n

# HBM traffic per launch: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
# wide coalesced reads (128-B requests tallied at 64 B), so the read side is doubled.
import json
out = {}
for k, c in traffic.items():
    rec = dict(c)
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        rd = 2.0 * 1024.0 * c.get("FETCH_SIZE", 0.0); wr = 1024.0 * c.get("WRITE_SIZE", 0.0)
        rec.update(read_bytes_per_launch=rd, write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr)
    out[k] = rec
# the form bench.py reads as profiles/counters_latest.json: {workload: {"source": ..., "kernels": {kernel: per-launch averages}}}
workload = sys.argv[2] if len(sys.argv) > 2 else "config2"
import datetime
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
wrapped = {workload: {"kernel_sources_sha": bench.kernel_sources_sha(), "collected": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"),
                      "source": "tools/profile.sh (rocprofv3 --pmc, one counter group per pass; FETCH_SIZE doubled for gfx950, FETCH_SIZE and "
                                "WRITE_SIZE x1024; one frame at a time, library options overlap=0,graphs=0); values are averages per launch",
                      "kernels": out}}
json.dump(wrapped, open(f"{root}/counters.json", "w"), indent=1)
print("\n== HBM bytes per launch (FETCH_SIZE x2 x1024 + WRITE_SIZE x1024), L2 hit rate, vector instructions ==")
for k, v in out.items():
    hit = v.get("TCC_HIT_sum"); miss = v.get("TCC_MISS_sum")
    l2 = f"{hit / (hit + miss):7.1%}" if hit is not None and miss is not None and hit + miss > 0 else "    n/a"
    print(f"{k:24s} read {v.get('read_bytes_per_launch', 0)/1e6:10.3f} MB  write {v.get('write_bytes_per_launch', 0)/1e6:10.3f} MB  L2 hit {l2}  "
          f"VALU insts {v.get('SQ_INSTS_VALU', 0):.4g}  wait_any/wave_cycles {v.get('SQ_WAIT_ANY', 0) / max(1.0, v.get('SQ_WAVE_CYCLES', 0)):.2f}")
