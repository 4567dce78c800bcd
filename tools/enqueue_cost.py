"""Development aid: host-side cost of enqueueing one frame (no synchronisation), vs frame size."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch, cases
from raytracer_amd import api
for res in ((240, 135), (1920, 1080)):
    for flags in ([], ["-r", "-rd", "4", "-ss", "-sss", "16"]):
        cfg, _ = api.parse_args(flags)
        ls = api.LoadedScene(cases.reference_scene(*res), cfg)
        r = api.Renderer(0, 16 << 30); r.upload(ls)
        out = torch.empty(r.out_bytes(), dtype=torch.uint8, device="cuda")
        for _ in range(20): r.render_device(cfg, out.data_ptr())
        r.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n): r.render_device(cfg, out.data_ptr())
        t1 = time.perf_counter(); r.synchronize(); t2 = time.perf_counter()
        print(f"{res} {flags}: enqueue {1e6*(t1-t0)/n:.1f} us/frame, total {1e6*(t2-t0)/n:.1f} us/frame", flush=True)
        r.close(); ls.close()
