"""Development aid: re-create case K of tools/fuzz_parity.py (same seed), render it with several library options and
print the pixels that differ from the oracle."""
import sys, argparse
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tools"))
import numpy as np
import cases, oracle, fuzz_parity
from raytracer_amd import api
ap = argparse.ArgumentParser(); ap.add_argument("--seed", type=int, default=11); ap.add_argument("--case", type=int, default=1142)
ap.add_argument("--dump", default=None); ap.add_argument("--no-gpu", action="store_true")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
for k in range(a.case + 1):
    text = fuzz_parity.random_scene(rng, (96, 54)); flags = fuzz_parity.random_flags(rng)
print("flags", flags)
if a.dump:
    Path(a.dump).write_text(text)
ref, _ = oracle.render(text, flags, str(cases.TEXTURES))
if a.no_gpu:
    np.save("/tmp/fuzz_case_ref.npy", ref); sys.exit(0)
cfg, _ = api.parse_args(flags)
for opts in ({}, {"lds_prims": 0}, {"lds_prims": 2}, {"overlap": 0, "graphs": 0}, {"second_level_min_group": 1 << 40}, {"device_build": 0}):
    ls = api.LoadedScene(text, cfg, str(cases.TEXTURES))
    r = api.Renderer(0)
    for n, v in opts.items(): r.set_option(n, v)
    r.upload(ls, with_bvh=bool(cfg.bvh)); img, _ = r.render(cfg); r.close(); ls.close()
    d = img.astype(np.int32) - ref
    ys, xs = np.nonzero(np.abs(d).max(axis=2))
    print(opts, [(int(x), int(y), img[y, x].tolist(), ref[y, x].tolist()) for x, y in zip(xs, ys)])
