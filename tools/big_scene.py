"""Development aid: render the synthetic N-primitive scene (arrays, no JSON) in BVH and -u modes and time it."""
import sys, time, argparse
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from raytracer_amd import api, scenegen as sg
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100000)
ap.add_argument("--res", default="1920x1080")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--modes", default="bvh,u")
ap.add_argument("--opt", action="append", default=[], help="library option name=value")
ap.add_argument("flags", nargs="*")
args, extra = ap.parse_known_args()
W, H = map(int, args.res.split("x"))
t0 = time.time(); ls = api.ArrayScene(sg.random_prims(args.n, 42), (W, H)); t1 = time.time()
imgs = {}
for mode in args.modes.split(","):
    flags = list(args.flags) + extra + (["-u"] if mode == "u" else [])
    cfg, _ = api.parse_args(flags)
    r = api.Renderer(0, 16 << 30)
    for kv in args.opt:
        r.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    t2 = time.time(); r.upload(ls, with_bvh=bool(cfg.bvh)); t3 = time.time()
    out = torch.empty(r.out_bytes(), dtype=torch.uint8, device="cuda")
    for i in range(args.reps):
        r.set_option('count_rays', 1 if i == 0 else 0)      # later repetitions: stage times with the kernels plain frames run
        st = r.render_device(cfg, out.data_ptr(), want_stats=True)
        print(f"n={args.n} {mode:4s} upload {t3-t2:.2f}s frame {st.ms_total:10.3f} ms rays {st.total_rays()} {st.total_rays()/st.ms_total/1e6:8.4f} Grays/s "
              f"ext0 {st.ms_extend_primary:.2f} shadow {st.ms_shadow:.2f} shade {st.ms_shade:.2f}", flush=True)
    imgs[mode] = out.cpu().numpy().reshape(H, W, 3)
    r.close()
if len(imgs) == 2:
    d = np.abs(imgs["bvh"].astype(int) - imgs["u"].astype(int)).max(axis=2)
    print("bvh vs -u: pixels differing", int((d > 0).sum()), "of", W * H, "(the two modes are not expected to agree: last-leaf rule)")
