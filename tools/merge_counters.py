"""Merges the counters.json of tools/profile.sh output directories into profiles/counters_latest.json (what bench.py
reads), one workload per directory; each entry keeps the stamp of the kernel sources it was collected on.
usage: python tools/merge_counters.py gpurun_out/prof_<tag> [...]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
dst = ROOT / "profiles" / "counters_latest.json"
cur = json.loads(dst.read_text()) if dst.exists() else {}
for d in sys.argv[1:]:
    new = json.loads((Path(d) / "counters.json").read_text())
    for wl, rec in new.items():
        # only this library's kernels (the build's sorts and copies are not frame kernels)
        rec["kernels"] = {k: v for k, v in rec["kernels"].items() if k.startswith("k_")}
        cur[wl] = rec
        print(f"{wl}: {len(rec['kernels'])} kernels, sources {rec.get('kernel_sources_sha')}")
dst.write_text(json.dumps(cur, indent=1))
