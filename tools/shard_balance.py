"""Development aid: per-rank frame time and ray count of the interleaved tile shards of one frame, rendered one rank
after the other on one GPU (how even is the split the multi-GPU run relies on?)."""
import sys, argparse
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import cases
from raytracer_amd import api
ap = argparse.ArgumentParser(); ap.add_argument("--world", type=int, default=8); ap.add_argument("flags", nargs="*")
a, extra = ap.parse_known_args()
flags = a.flags + extra or ["-r", "-rd", "4", "-ss", "-sss", "16"]
cfg, _ = api.parse_args(flags)
ls = api.LoadedScene(cases.reference_scene(1920, 1080), cfg, str(cases.TEXTURES))
ms, rays = [], []
for rank in range(a.world):
    r = api.Renderer(0, 16 << 30); r.upload(ls); r.set_shard(rank, a.world)
    out = torch.empty(r.out_bytes(), dtype=torch.uint8, device="cuda")
    for _ in range(5):
        r.render_device(cfg, out.data_ptr())
    r.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = torch.cuda.Stream(); r.set_stream(s.cuda_stream)
    for _ in range(3):
        r.render_device(cfg, out.data_ptr())
    e0.record(s)
    for _ in range(20):
        r.render_device(cfg, out.data_ptr())
    e1.record(s); torch.cuda.synchronize()
    st = r.render_device(cfg, out.data_ptr(), want_stats=True)
    ms.append(e0.elapsed_time(e1) / 20); rays.append(st.total_rays()); r.close()
print("ms per rank  ", [round(x, 3) for x in ms], "max/mean", round(max(ms) / np.mean(ms), 3))
print("rays per rank", rays, "max/mean", round(max(rays) / np.mean(rays), 3))
