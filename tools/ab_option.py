"""Development aid: one bench workload rendered under several sets of library options (set before the upload), with the
per-stage device times of a plain frame and a check that the images agree.
usage: python tools/ab_option.py --workload config3 --set hierarchy_order=0 --set hierarchy_order=1 [--n 1000000]"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
import bench
import cases
from raytracer_amd import api, scenegen as sg

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="config3")
ap.add_argument("--set", action="append", default=[], help="comma-separated name=value options of one variant")
ap.add_argument("--n", type=int, default=0)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--fallback", type=int, default=0, help="rt_config.fallback: 2 = no retry of secondary rays after a BVH miss (not the reference's image)")
ap.add_argument("--flags", default=None, help="render flags instead of the workload's own, e.g. --flags='-r -rd 4'")
args = ap.parse_args()
wl = dict(bench.WORKLOADS[args.workload])
if args.flags is not None:
    wl["flags"] = args.flags.split()
W, H = wl["res"]
cfg, _ = api.parse_args(wl["flags"])
cfg.fallback = args.fallback
if wl["scene"] == "ref":
    ls = api.LoadedScene(cases.reference_scene(W, H), cfg, str(cases.TEXTURES))
elif wl["scene"].startswith("blend:"):
    ls = api.LoadedScene(cases.blend_scene(wl["scene"][6:], W, H), cfg, str(cases.TEXTURES))
else:
    moving = bool(cfg.motion_blur)
    ls = api.ArrayScene(sg.random_prims(args.n or wl["n"], wl.get("seed", 42), moving=moving), (W, H), moving=moving)
first = None
for variant in args.set or [""]:
    r = api.Renderer(0, 16 << 30)
    for kv in filter(None, variant.split(",")):
        r.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    r.set_option("overlap", 0)
    r.upload(ls, with_bvh=bool(cfg.bvh))
    out = torch.empty(r.out_bytes(), dtype=torch.uint8, device="cuda")
    st = r.render_device(cfg, out.data_ptr(), want_stats=True)
    img = out.cpu().numpy().copy()
    r.set_option("count_rays", 0)
    best = None
    for _ in range(args.reps):
        tm = r.render_device(cfg, out.data_ptr(), want_stats=True)
        if best is None or tm.ms_total < best.ms_total:
            best = tm
    same = "first" if first is None else ("same image" if np.array_equal(first, img) else f"{int((first != img).sum())} BYTES DIFFER")
    if first is None:
        first = img
    print(f"{variant or '(defaults)':40s} frame {best.ms_total:8.3f} ms  ext0 {best.ms_extend_primary:.3f} ext+ {best.ms_extend_secondary:.3f} "
          f"shadow {best.ms_shadow:.3f} shade {best.ms_shade:.3f}  rays {st.total_rays()}  {same}", flush=True)
    r.close()
