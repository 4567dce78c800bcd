"""Fixtures that need renders with several rand() seeds, which only the CPU restatement offers (the reference never
calls srand; the restatement with seed 1 reproduces the reference binary byte for byte, checked below where the
binary exists).  Output: seeded.npz
  floor_<case>_seed1 / _seed2 : uint8 images of cases.NOISE_FLOOR
  mirror_direct_var           : per-pixel variance (summed over channels) of the mirror scene without reflection
  mirror_mean, mirror_var     : per-pixel mean image and variance (summed over channels) with cases.MIRROR_FLAGS
  banded_<case>               : uint8 images of cases.STOCHASTIC_BANDED, rendered in cases.BANDS horizontal bands (band b
                                with rand() seed 1 + b), for flag sets the reference binary needs hours for
  alleffects_seed1 / _seed2   : cases.ALL_EFFECTS rendered in bands with two different sets of rand() seeds
  alleffects_high             : the same estimators with cases.ALL_EFFECTS_HIGH (256 primary samples per pixel)
  deep_mean_<scene>_<flags>   : float32 mean image over cases.DEEP_GLOSS_SEEDS rand() seeds of cases.DEEP_GLOSS
  seedmean_<scene>_<flags>    : float32 mean image over cases.SEED_MEAN_SEEDS rand() seeds (101, 102, ...) of the GPU-side flag
                                sets of cases.SEED_MEAN: the same estimators, sample count for sample count
Usage: make_seeded.py [floor] [mirror] [banded] [alleffects] [deep] [seedmean]   (default: all; parts not named are kept from the existing file)
"""
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import cases  # noqa: E402
import oracle  # noqa: E402
import ref_runner  # noqa: E402

TEX = str(cases.TEXTURES)


def render(job):
    scene, flags, seed = job
    return oracle.render(cases.SCENES[scene](), flags, TEX, rand_seed=seed)[0]


def mirror_stats(pool_map=map):
    seeds = range(1, cases.MIRROR_SEEDS + 1)
    a = np.stack(list(pool_map(render, [("mirror", cases.MIRROR_DIRECT_FLAGS, s) for s in seeds]))).astype(np.float64)
    b = np.stack(list(pool_map(render, [("mirror", cases.MIRROR_FLAGS, s) for s in seeds]))).astype(np.float64)
    return a.var(axis=0).sum(axis=2).astype(np.float32), b.mean(axis=0).astype(np.float32), b.var(axis=0).sum(axis=2).astype(np.float32)


def render_band(job):
    scene, flags, band = job[:3]
    seed0 = job[3] if len(job) > 3 else 1
    text = cases.SCENES[scene]()
    w, h = ref_runner.get_resolution(text)
    y0, y1 = (h * band) // cases.BANDS, (h * (band + 1)) // cases.BANDS
    img = oracle.render(text, flags, TEX, rand_seed=seed0 + band, window=(0, y0, w, y1))[0]
    return y0, y1, img[y0:y1]


def banded(ex, scene, flags, seed0=1):
    w, h = ref_runner.get_resolution(cases.SCENES[scene]())
    img = np.zeros((h, w, 3), dtype=np.uint8)
    for y0, y1, band in ex.map(render_band, [(scene, flags, b, seed0) for b in range(cases.BANDS)]):
        img[y0:y1] = band
    return img


def main():
    parts = set(sys.argv[1:]) or {"floor", "mirror", "banded", "alleffects", "deep", "seedmean"}
    path = HERE / "seeded.npz"
    out = dict(np.load(path)) if path.exists() else {}
    with ProcessPoolExecutor(max_workers=int(__import__("os").environ.get("SEEDED_WORKERS", "7"))) as ex:
        if "floor" in parts:
            jobs = [(s, f, seed) for s, f in cases.NOISE_FLOOR for seed in (1, 2)]
            for (s, f, seed), img in zip(jobs, ex.map(render, jobs)):
                out[f"floor_{cases.case_id(s, f)}_seed{seed}"] = img.astype(np.uint8)
                if seed == 1 and ref_runner.have_ref():
                    ref, _, _ = ref_runner.run_ref(cases.SCENES[s](), f, {p.name: p for p in cases.TEXTURES.glob("*.ppm")})
                    assert np.array_equal(ref, img), "seed 1 must be the reference binary's own image"
                    print(f"{cases.case_id(s, f)}: seed 1 equals the reference binary")
        if "mirror" in parts:
            out["mirror_direct_var"], out["mirror_mean"], out["mirror_var"] = mirror_stats(ex.map)
        if "banded" in parts:
            for s, f in cases.STOCHASTIC_BANDED:
                w, h = ref_runner.get_resolution(cases.SCENES[s]())
                img = np.zeros((h, w, 3), dtype=np.uint8)
                for y0, y1, band in ex.map(render_band, [(s, f, b) for b in range(cases.BANDS)]):
                    img[y0:y1] = band
                out[f"banded_{cases.case_id(s, f)}"] = img
        if "alleffects" in parts:
            scene, flags = cases.ALL_EFFECTS
            out["alleffects_seed1"] = banded(ex, scene, flags, 1)
            out["alleffects_seed2"] = banded(ex, scene, flags, 1001)
            out["alleffects_high"] = banded(ex, scene, cases.ALL_EFFECTS_HIGH, 2001)
        if "deep" in parts:
            for scene, flags in cases.DEEP_GLOSS:
                seeds = range(1, cases.DEEP_GLOSS_SEEDS + 1)
                imgs = np.stack(list(ex.map(render, [(scene, flags, s) for s in seeds]))).astype(np.float64)
                out[f"deep_mean_{cases.case_id(scene, flags)}"] = imgs.mean(axis=0).astype(np.float32)
                out[f"deep_var_{cases.case_id(scene, flags)}"] = imgs.var(axis=0).sum(axis=2).astype(np.float32)
        if "seedmean" in parts:
            for scene, flags, _ in cases.SEED_MEAN:
                seeds = range(101, 101 + cases.SEED_MEAN_SEEDS)
                imgs = np.stack(list(ex.map(render, [(scene, flags, s) for s in seeds]))).astype(np.float64)
                out[f"seedmean_{cases.case_id(scene, flags)}"] = imgs.mean(axis=0).astype(np.float32)
    np.savez_compressed(path, **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
