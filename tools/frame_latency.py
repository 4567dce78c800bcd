"""Development aid: the latency of one frame of a bench workload rendered alone (enqueue -> done, the recorded graph
replayed), under several sets of library options, with a check that the images agree.
usage: python tools/frame_latency.py --workload config2 --set shadow_streams=1 --set shadow_streams=2"""
import argparse
import sys
import time
import zlib
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import bench
import cases
from raytracer_amd import api

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="config2")
ap.add_argument("--set", action="append", default=[], help="comma-separated name=value options of one variant")
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--flags", default=None)
args = ap.parse_args()
wl = dict(bench.WORKLOADS[args.workload])
if args.flags is not None:
    wl["flags"] = args.flags.split()
W, H = wl["res"]
cfg, _ = api.parse_args(wl["flags"])
ls = api.LoadedScene(cases.reference_scene(W, H) if wl["scene"] == "ref" else cases.blend_scene(wl["scene"][6:], W, H), cfg, str(cases.TEXTURES))
first = None
for variant in args.set or [""]:
    r = api.Renderer(0, 16 << 30)
    for kv in filter(None, variant.split(",")):
        r.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    r.upload(ls, with_bvh=bool(cfg.bvh))
    out = torch.empty(r.out_bytes(), dtype=torch.uint8, device="cuda")
    for _ in range(8):
        r.render_device(cfg, out.data_ptr()); r.synchronize()
    crc = zlib.crc32(out.cpu().numpy().tobytes())
    each = []
    for _ in range(args.reps):
        t = time.perf_counter()
        r.render_device(cfg, out.data_ptr()); r.synchronize()
        each.append((time.perf_counter() - t) * 1e3)
    each.sort()
    same = "first" if first is None else ("same image" if crc == first else "IMAGE DIFFERS")
    first = crc if first is None else first
    print(f"{variant or '(defaults)':44s} alone: mean {sum(each) / len(each):.4f}  median {each[len(each) // 2]:.4f}  min {each[0]:.4f} ms   {same}", flush=True)
    r.close()
