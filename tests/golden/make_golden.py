"""Regenerates the golden fixtures from the compiled, unmodified reference (oracle/_ref/raytracer_ref).

Runs only where /root/reference is mounted (`make -C oracle ref` first).  Outputs are data only:
  deterministic.npz : uint8 [H,W,3] image per deterministic case of tests/cases.py
  stochastic.npz    : uint8 [H,W,3] high-sample image per stochastic case, and per golden flag set of cases.SEED_MEAN
  config1_1080p.npz : the reference's default render of its own scene.json at 1920x1080
  kat_primitives.npz: known answers of Sphere/Cube/Plane::intersect, their getAABB and AABB::intersect on seeded random
                      inputs, printed by oracle/_ref/ref_kat (`make -C oracle kat`; `make_golden.py kat` regenerates)
"""
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import cases  # noqa: E402
import ref_runner  # noqa: E402

TEX = {p.name: p for p in cases.TEXTURES.glob("*.ppm")}


def run(job):
    scene, flags = job
    img, secs, _ = ref_runner.run_ref(cases.SCENES[scene](), flags, TEX)
    assert img.min() >= 0 and img.max() <= 255
    return cases.case_id(scene, flags), img.astype(np.uint8), secs


def make_kat(n=1500):
    """Records of oracle/ref_kat.cpp: float32 `kind, n_in, n_out, inputs, outputs`; stored as the raw stream."""
    import subprocess
    exe = HERE.parent.parent / "oracle" / "_ref" / "ref_kat"
    raw = np.frombuffer(subprocess.run([str(exe), str(n)], capture_output=True, check=True).stdout, dtype=np.float32)
    np.savez_compressed(HERE / "kat_primitives.npz", stream=raw)
    print(f"kat_primitives: {raw.size} floats")


def main():
    if sys.argv[1:] == ["kat"]:
        make_kat(); return
    sto_jobs = cases.STOCHASTIC + [(s, g) for s, _, g in cases.SEED_MEAN]
    if len(sys.argv) > 1:            # regenerate only the named cases, keep the rest
        for fname, catalogue in (("deterministic.npz", cases.DETERMINISTIC), ("stochastic.npz", sto_jobs)):
            have = dict(np.load(HERE / fname))
            want = {cases.case_id(s, f) for s, f in catalogue}
            named = [job for job in catalogue if cases.case_id(*job) in sys.argv[1:]]
            with ProcessPoolExecutor(max_workers=4) as ex:
                for k, v, secs in ex.map(run, named):
                    have[k] = v; print(f"{k:60s} {secs:8.2f} s", flush=True)
            np.savez_compressed(HERE / fname, **{k: v for k, v in have.items() if k in want})
        return
    with ProcessPoolExecutor(max_workers=6) as ex:
        det = list(ex.map(run, cases.DETERMINISTIC))
        sto = list(ex.map(run, sto_jobs))
    np.savez_compressed(HERE / "deterministic.npz", **{k: v for k, v, _ in det})
    np.savez_compressed(HERE / "stochastic.npz", **{k: v for k, v, _ in sto})
    for k, _, s in det + sto:
        print(f"{k:60s} {s:8.2f} s")
    img, secs, _ = ref_runner.run_ref(cases.reference_scene(1920, 1080), [], TEX)
    np.savez_compressed(HERE / "config1_1080p.npz", image=img.astype(np.uint8), loop_seconds=np.float64(secs))
    print(f"config1 1080p {secs:.2f} s")
    make_kat()


if __name__ == "__main__":
    main()
