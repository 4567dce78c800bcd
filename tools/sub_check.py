"""Development aid: the second-level hierarchy must not change a single pixel. Renders the same scenes with it
and (RT_NO_SUBBVH=1, in a child process) without it and compares."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from raytracer_amd import api, scenegen as sg

CASES = [(10000, "1920x1080", [], 0.0)] if os.environ.get("SUB_CHECK_ONE") else [(10000, "1920x1080", [], 0.0), (10000, "1920x1080", ["-u"], 0.0), (10000, "640x360", []), (10000, "640x360", ["-u"]), (10000, "640x360", ["-rd", "2", "-grs", "1"]),
         (100000, "480x270", []), (100000, "480x270", ["-u"])]

def render(n, res, flags, tf=0.1, plain=False):
    W, H = map(int, res.split("x"))
    ls = api.ArrayScene(sg.random_prims(n, 42, transparent_frac=tf), (W, H))
    cfg, _ = api.parse_args(flags)
    r = api.Renderer(0, 8 << 30)
    if plain:
        r.set_option("second_level_min_group", 1 << 40)
    r.upload(ls, with_bvh=bool(cfg.bvh))
    img, st = r.render(cfg); r.close()
    return img

for k, c in enumerate(CASES):
    c = tuple(c) + ((0.1,) if len(c) == 3 else ())
    n, res, flags = c[:3]
    img, ref = render(*c), render(*c, plain=True)
    d = np.abs(img.astype(int) - ref.astype(int)).max(axis=2)
    ys, xs = np.nonzero(d)
    print(n, res, flags, "differing pixels:", int((d > 0).sum()), list(zip(xs[:5].tolist(), ys[:5].tolist())), flush=True)
