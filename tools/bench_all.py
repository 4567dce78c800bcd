#!/usr/bin/env python3
"""Every bench.py workload on one GPU, one frame at a time and with the default frames in flight.

Writes gpurun_out/bench_all.json (per workload: the bench lines' timing fields) and prints a markdown table.
usage: python tools/bench_all.py [workload ...]
"""
import json
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ORDER = ["config1", "config2", "config3", "config3u", "bvh5", "bvh5u", "cube_mb", "config4s", "config4", "config5s"]


def run(workload: str, in_flight: int, steps: int, warmup: int) -> dict:
    """in_flight = 0: bench.py's default"""
    cmd = [sys.executable, str(ROOT / "bench.py"), "--workload", workload, "--no-cpu-baseline", "--no-scaling-4k", "--steps", str(steps), "--warmup", str(warmup)]
    if in_flight:
        cmd += ["--frames-in-flight", str(in_flight)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, timeout=900)     # progress notes go straight to stderr
    if out.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} failed")
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def main():
    todo = sys.argv[1:] or ORDER
    res = {}
    for wl in todo:
        heavy = wl in ("config4", "config4s", "config5s")
        steps, warmup = (2, 1) if heavy else ((20, 3) if wl == "cube_mb" else (100, 8))
        t0 = time.time()
        one = run(wl, 1, steps, warmup)
        entry = {"workload": one["config"]["workload"], "steps": steps, "rays_per_frame": one["config"]["rays_per_frame"],
                 "one_frame_at_a_time": {"ms_per_step": one["ms_per_step"], "rays_per_s": one["value"]},
                 "kernel_ms_of_one_profiled_frame": {k: v["ms"] for k, v in one["roofline"]["by_kernel"].items()}}
        if not heavy:
            many = run(wl, 0, steps, warmup)
            entry["frames_in_flight"] = {"n": many["config"]["frames_in_flight"], "ms_per_step": many["ms_per_step"],
                                         "rays_per_s": many["value"], "frame_ms_alone": many["config"]["frame_ms_alone"]}
        res[wl] = entry
        print(f"{wl}: {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
        (ROOT / "gpurun_out").mkdir(exist_ok=True)
        (ROOT / "gpurun_out" / "bench_all.json").write_text(json.dumps(res, indent=1))
    print("| Workload | ms per frame, one at a time | Grays/s | ms per frame, in flight | Grays/s |")
    print("|---|---|---|---|---|")
    for wl, e in res.items():
        a = e["one_frame_at_a_time"]; b = e.get("frames_in_flight")
        tail = f"{b['ms_per_step']:.3f} | {b['rays_per_s'] / 1e9:.1f}" if b else "(not used) | "
        print(f"| {e['workload']} | {a['ms_per_step']:.3f} | {a['rays_per_s'] / 1e9:.1f} | {tail} |")


if __name__ == "__main__":
    main()
