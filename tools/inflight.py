"""Development aid: throughput of P frames in flight (P contexts on P streams, round robin) for a shard of world W."""
import sys, argparse, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import cases
from raytracer_amd import api
ap = argparse.ArgumentParser(); ap.add_argument("--world", type=int, default=1); ap.add_argument("--p", type=int, nargs="+", default=[1, 2, 3, 4])
ap.add_argument("--ws", type=int, default=4); ap.add_argument("--setstream", action="store_true"); ap.add_argument("--stats", action="store_true"); ap.add_argument("--late", action="store_true"); ap.add_argument("--own", action="store_true"); ap.add_argument("--opt", default="")
a = ap.parse_args()
flags = ["-r", "-rd", "4", "-ss", "-sss", "16"]
cfg, _ = api.parse_args(flags)
ls = api.LoadedScene(cases.reference_scene(1920, 1080), cfg, str(cases.TEXTURES))
for P in a.p:
    rs, outs = [], []
    for k in range(P):
        r = api.Renderer(0, a.ws << 30)
        for kv in filter(None, a.opt.split(',')): r.set_option(kv.split('=')[0], int(kv.split('=')[1]))
        r.upload(ls); r.set_shard(0, a.world)
        if not a.own:
            s = torch.cuda.Stream(); r.set_stream(s.cuda_stream); r._s = s
        if a.setstream and k == 0 and not a.own: torch.cuda.set_stream(s)
        if a.late and k == 0:
            for _ in range(5): r.render_device(cfg, torch.empty(r.out_bytes(), dtype=torch.uint8, device='cuda').data_ptr())
            torch.cuda.synchronize()
        rs.append(r); outs.append(torch.empty(r.out_bytes(), dtype=torch.uint8, device="cuda"))
    if a.stats: rs[0].render_device(cfg, outs[0].data_ptr(), want_stats=True)
    for k in range(6 * P):
        rs[k % P].render_device(cfg, outs[k % P].data_ptr())
    torch.cuda.synchronize()
    n = 120
    t0 = time.perf_counter()
    for k in range(n):
        rs[k % P].render_device(cfg, outs[k % P].data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"world {a.world} frames in flight {P}: {dt*1e3:.4f} ms per frame", flush=True)
    for r in rs: r.close()
