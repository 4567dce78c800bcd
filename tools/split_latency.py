"""Development aid: latency of ONE frame of a bench workload rendered as n interleaved-tile parts on the same GPU
(rt_multi with a repeated device id: one context and stream per part, blocks copied and re-ordered on the device),
against the same frame from one context.
usage: python tools/split_latency.py [--workload config2] [--parts 1 2 3 4]"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import bench
import cases
from raytracer_amd import api

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="config2")
ap.add_argument("--parts", type=int, nargs="*", default=[1, 2, 3, 4])
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--overlap", type=int, default=0)
args = ap.parse_args()
wl = bench.WORKLOADS[args.workload]
W, H = wl["res"]
cfg, _ = api.parse_args(wl["flags"])
text = cases.reference_scene(W, H) if wl["scene"] == "ref" else cases.blend_scene(wl["scene"][6:], W, H)
ls = api.LoadedScene(text, cfg, str(cases.TEXTURES))
want = None
for n in args.parts:
    m = api.MultiRenderer([0] * n, 4 << 30)
    m.set_option("overlap", args.overlap if n > 1 else 1)
    m.upload(ls, with_bvh=bool(cfg.bvh))
    img = m.render(cfg)
    if want is None:
        want = img
    for _ in range(5):
        m.render_device(cfg); m.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        m.render_device(cfg); m.synchronize()
    ms = (time.perf_counter() - t0) * 1000.0 / args.reps
    print(f"{args.workload}: {n} part(s): {ms:.3f} ms per frame, enqueue to image on the device; same image: {np.array_equal(img, want)}", flush=True)
    m.close()
