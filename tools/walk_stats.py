"""Development aid: step counters of one frame of a bench workload, from a library whose kernels were built with
-DRT_WALK_STATS (make -C raytracer_amd/csrc B=build_ws OUT=../librt_amd_ws.so KDEFS=-DRT_WALK_STATS), loaded through
RT_AMD_LIB.  Prints, per counter, the lanes and the wavefronts that passed the site, per traced ray.
usage: RT_AMD_LIB=raytracer_amd/librt_amd_ws.so python tools/walk_stats.py --workload config3 [--n N] [--res WxH] [--flags='...']"""
import argparse
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
import bench
import cases
from raytracer_amd import api, scenegen as sg

NAMES = ["first-level box tests", "leaf searches", "second-level steps", "plane tests", "cube tests", "-", "sphere tests",
         "retries over the file-order set", "entry runs tested", "traces"]
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="config3")
ap.add_argument("--n", type=int, default=0)
ap.add_argument("--res", default=None)
ap.add_argument("--flags", default=None)
ap.add_argument("--set", default="", help="comma-separated name=value library options")
args = ap.parse_args()
wl = dict(bench.WORKLOADS[args.workload])
if args.flags is not None:
    wl["flags"] = args.flags.split()
W, H = map(int, args.res.split("x")) if args.res else wl["res"]
cfg, _ = api.parse_args(wl["flags"])
if wl["scene"] == "ref":
    ls = api.LoadedScene(cases.reference_scene(W, H), cfg, str(cases.TEXTURES))
elif wl["scene"].startswith("blend:"):
    ls = api.LoadedScene(cases.blend_scene(wl["scene"][6:], W, H), cfg, str(cases.TEXTURES))
else:
    moving = bool(cfg.motion_blur)
    ls = api.ArrayScene(sg.random_prims(args.n or wl["n"], wl.get("seed", 42), moving=moving), (W, H), moving=moving)
r = api.Renderer(0, 16 << 30)
for kv in filter(None, args.set.split(",")):
    r.set_option(kv.split("=")[0], int(kv.split("=")[1]))
r.set_option("walk_stats", 1); r.set_option("overlap", 0)
r.upload(ls, with_bvh=bool(cfg.bvh))
out = torch.empty(r.out_bytes(), dtype=torch.uint8, device="cuda")
st = r.render_device(cfg, out.data_ptr(), want_stats=True)
fn = api.lib().rt_debug_walk_stats
fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]; fn.restype = C.c_int
buf = (C.c_uint64 * 32)()
assert fn(r._ctx, buf) == 0
v = np.array(list(buf), dtype=np.float64)
rays = st.total_rays()
print(f"{bench.describe(args.workload)}  frame {st.ms_total:.2f} ms  rays {rays} (primary {st.primary_rays} shadow {st.shadow_rays} refl {st.reflection_rays} refr {st.refraction_rays})")
print(f"stage ms: ext0 {st.ms_extend_primary:.3f} ext+ {st.ms_extend_secondary:.3f} shadow {st.ms_shadow:.3f} shade {st.ms_shade:.3f}")
for k, name in enumerate(NAMES):
    if name == "-" or v[k] == 0:
        continue
    print(f"{name:34s} lanes {v[k]:14.0f} ({v[k] / max(v[9], 1):8.2f} per trace)   wavefront passes {v[16 + k]:12.0f}   lanes per pass {v[k] / max(v[16 + k], 1):5.1f}")
r.close(); ls.close()
