"""Randomised parity sweep on scenes of tens to thousands of primitives (groups searched through second-level
hierarchies, records in hierarchy order): random populations x random deterministic flag sets on small films, GPU
(through the C ABI) against the CPU restatement - pixels and the four ray counts.  Development aid; exits non-zero
on any differing pixel."""
import argparse
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle, cases
from raytracer_amd import api, scenegen as sg

ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=100); ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
bad = 0
for k in range(args.cases):
    n = int(rng.choice([12, 40, 150, 600, 2500]))
    extent = float(rng.uniform(3.0, 9.0))
    rmax = float(rng.uniform(0.1, 0.6)) * (1.0 if n < 1000 else 0.4)
    res = [(96, 54), (128, 72), (61, 47)][int(rng.integers(0, 3))]
    text = sg.random_scene(n, int(rng.integers(1, 1 << 30)), res, with_ground=bool(rng.uniform() < 0.8), transparent_frac=float(rng.choice([0.0, 0.1, 0.4])),
                           extent=extent, rmin=rmax * 0.3, rmax=rmax)
    flags = []
    if rng.uniform() < 0.25 and n <= 600: flags.append("-u")
    if rng.uniform() < 0.6: flags += ["-rd", str(int(rng.integers(1, 4))), "-grs", str(int(rng.choice([1, 1, 3])))]
    ref, ost = oracle.render(text, flags, str(cases.TEXTURES))
    img, st = api.render_scene_text(text, flags, str(cases.TEXTURES))
    npx = int((np.abs(img.astype(np.int32) - ref).max(axis=2) > 0).sum())
    rays_ok = (st.primary_rays, st.shadow_rays_ref, st.reflection_rays_ref, st.refraction_rays_ref) == (ost.primary, ost.shadow, ost.reflection, ost.refraction)
    if npx or not rays_ok:
        bad += 1
        print(f"case {k} n {n} flags {flags}: {npx} pixels differ, rays equal: {rays_ok}", flush=True)
    if k % 25 == 24:
        print(f"... {k + 1} cases, {bad} with differences", flush=True)
print(f"{args.cases} cases, {bad} with differences")
sys.exit(1 if bad else 0)
