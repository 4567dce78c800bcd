"""Development aid: time the ingest stages (scene arrays, host BVH build, scene upload, BVH upload) for N primitives."""
import sys, time, argparse, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from raytracer_amd import api, scenegen as sg
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs="+", default=[1000000])
ap.add_argument("--no-gpu", action="store_true")
args = ap.parse_args()
for n in args.n:
    t0 = time.time(); ls = api.ArrayScene(sg.random_prims(n, 42), (1920, 1080)); t1 = time.time()
    b = ls.bvh(); t2 = time.time()
    line = f"n={n} arrays {t1-t0:.2f}s host_bvh {t2-t1:.2f}s"
    if not args.no_gpu:
        r = api.Renderer(0, 16 << 30)
        L = api.lib()
        t3 = time.time(); r._check(L.rt_upload_scene(r._ctx, C.byref(ls.scene))); t4 = time.time()
        r._check(L.rt_upload_bvh(r._ctx, b.nodes, b.n_nodes, b.prim_ids, b.n_prim_ids)); t5 = time.time()
        r.width, r.height = ls.width, ls.height
        cfg, _ = api.parse_args([])
        out = torch.empty(r.out_bytes(), dtype=torch.uint8, device="cuda")
        st = r.render_device(cfg, out.data_ptr(), want_stats=True)
        st = r.render_device(cfg, out.data_ptr(), want_stats=True)
        line += f" upload_scene {t4-t3:.2f}s upload_bvh {t5-t4:.2f}s frame {st.ms_total:.3f} ms sum {int(out.sum().item())}"
        r.close()
    print(line, flush=True)
