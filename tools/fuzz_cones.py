"""Development aid: many random small scenes (at most eight primitives) with soft shadows, rendered with the cone rule of
k_shadow (pairs that cannot be hit, pairs in full shadow, the queue of pairs to trace) on and off; prints every case whose
frames differ.  usage: python tools/fuzz_cones.py [--cases 400] [--seed 1]"""
import argparse
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import cases
from raytracer_amd import api
from test_gpu_soft_shadows import _small_random_scene

ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=400); ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
bad = 0; on = off = 0
for k in range(args.cases):
    text = _small_random_scene(rng, (128, 72))
    flags = ["-ss", "-sss", str(int(rng.choice([4, 8, 16, 64])))]
    if rng.uniform() < 0.5:
        flags += ["-r", "-rd", str(int(rng.integers(1, 4)))] + (["-grs", "2"] if rng.uniform() < 0.5 else [])
    if rng.uniform() < 0.4:
        flags += ["-m", "-mbs", "2"]
    if rng.uniform() < 0.2:
        flags += ["-u"]
    cfg, _ = api.parse_args(flags); cfg.seed = k
    ls = api.LoadedScene(text, cfg, str(cases.TEXTURES))
    fast = int(rng.integers(0, 2))
    got = []
    for cull in (1, 0, 2):
        r = api.Renderer(0); r.set_option("cull_cones", 1 if cull else 0); r.set_option("fast_samples", fast)
        if cull == 2:
            r.set_option("cull_umbra", 0); r.set_option("shadow_queue", int(k % 2)); r.set_option("shadow_queue_rcap", 64)
        r.upload(ls, with_bvh=bool(cfg.bvh))
        got.append(r.render(cfg)); r.close()
    ls.close()
    on += got[0][1].shadow_rays; off += got[1][1].shadow_rays
    if not (np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[2][0], got[1][0])):
        bad += 1
        print(f"case {k} seed {args.seed} flags {flags} fast {fast}: {int((got[0][0] != got[1][0]).any(axis=2).sum())} / {int((got[2][0] != got[1][0]).any(axis=2).sum())} pixels differ", flush=True)
    if k % 100 == 99:
        print(f"... {k + 1} cases, {bad} bad", flush=True)
print(f"{args.cases} cases, {bad} with differences; shadow rays traced {on} with the rule, {off} without")
sys.exit(1 if bad else 0)
