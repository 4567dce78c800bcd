"""Development aid: time a workload through rt_render_device and print rays/s and per-stage device times."""
import sys, time, argparse
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import cases
from raytracer_amd import api, scenegen as sg

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="1920x1080")
ap.add_argument("--scene", default="ref")
ap.add_argument("--n", type=int, default=1000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("flags", nargs="*")
args, extra = ap.parse_known_args()
flags = args.flags + extra
W, H = map(int, args.res.split("x"))
if args.scene == "ref":
    text = cases.reference_scene(W, H)
elif args.scene == "random":
    text = sg.random_scene(args.n, 42, (W, H))
elif args.scene == "cluster":
    text = sg.cluster_scene(args.n, 7, (W, H))
cfg, _ = api.parse_args(flags)
t0 = time.time(); ls = api.LoadedScene(text, cfg, str(cases.TEXTURES)); t1 = time.time()
r = api.Renderer(0, 16 << 30)
r.upload(ls, with_bvh=bool(cfg.bvh)); t2 = time.time()
out = torch.empty(r.out_bytes(), dtype=torch.uint8, device="cuda")
print(f"parse {t1-t0:.2f}s upload+bvh {t2-t1:.2f}s flags {flags}")
for i in range(args.reps):
    st = r.render_device(cfg, out.data_ptr(), want_stats=True)
    print(f"frame {st.ms_total:9.3f} ms  rays {st.total_rays():>13d}  {st.total_rays()/st.ms_total/1e6:8.3f} Grays/s  batches {st.batches} "
          f"ext0 {st.ms_extend_primary:.3f} ext+ {st.ms_extend_secondary:.3f} shadow {st.ms_shadow:.3f} shade {st.ms_shade:.3f}", flush=True)
