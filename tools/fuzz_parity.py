"""Randomised parity sweep: random small scenes (all primitive classes, tilted n-gons, transparency, several lights,
textures, moving objects) x random deterministic flag sets, GPU (through the C ABI) against the CPU restatement.
Development aid; prints every case with a difference and exits non-zero if any pixel is off by more than 1 LSB."""
import sys, argparse
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle, cases
from raytracer_amd import api, scenegen as sg


def random_scene(rng, res):
    cam_loc = (rng.uniform(-3, 3), rng.uniform(-12, -6), rng.uniform(0.5, 5))
    gaze = (rng.uniform(-0.3, 0.3), 1.0, rng.uniform(-0.4, 0.1))
    cams = [sg.camera(cam_loc, gaze, res=res, focal_length=float(rng.choice([24.0, 35.0, 50.0])))]
    lights = [sg.light((rng.uniform(-6, 6), rng.uniform(-6, 4), rng.uniform(2, 9)), rng.uniform(100, 1500)) for _ in range(rng.integers(1, 4))]
    texs = [None, None, "checkerboard.ppm", "tex2.ppm"]

    def mat():
        return sg.material(rng.uniform(0, 1, 3), rng.uniform(0, 0.3, 3) * rng.integers(0, 2), float(rng.choice([0.0, 5.0, 64.0, 200.0])),
                           float(rng.choice([0.0, 0.0, 0.5, 0.9])), float(rng.choice([1.0, 1.3, 1.5, 0.0])), texs[rng.integers(0, 4)])
    spheres = [sg.sphere((rng.uniform(-4, 4), rng.uniform(-3, 8), rng.uniform(0, 3)), rng.uniform(0.2, 1.5), mat(),
                         end=(rng.uniform(-4, 4), rng.uniform(-3, 8), rng.uniform(0, 3)) if rng.uniform() < 0.3 else None)
               for _ in range(rng.integers(0, 8))]
    cubes = [sg.cube((rng.uniform(-4, 4), rng.uniform(-3, 8), rng.uniform(0, 3)), rng.uniform(0.2, 1.5), mat(),
                     rotation=rng.uniform(0, 6.28, 3) * rng.integers(0, 2, 3))
             for _ in range(rng.integers(0, 8))]
    planes = [sg.ground(float(rng.uniform(8, 30)), float(rng.uniform(-0.5, 0.2)), mat())] if rng.uniform() < 0.8 else []
    for _ in range(rng.integers(0, 3)):        # tilted convex n-gon: regular polygon in a random plane, vertices shuffled
        n = int(rng.integers(3, 7)); c = np.array([rng.uniform(-4, 4), rng.uniform(0, 8), rng.uniform(0.5, 3)])
        a = rng.normal(size=3); a /= np.linalg.norm(a); b = np.cross(a, rng.normal(size=3)); b /= np.linalg.norm(b)
        rad = rng.uniform(0.5, 3)
        verts = [c + rad * (np.cos(2 * np.pi * k / n) * a + np.sin(2 * np.pi * k / n) * b) for k in range(n)]
        rng.shuffle(verts)
        planes.append(sg.plane(verts, mat()))
    return sg.scene_text(cams, lights, spheres, cubes, planes)


def random_flags(rng):
    f = []
    if rng.uniform() < 0.3: f.append("-u")
    if rng.uniform() < 0.6: f += ["-rd", str(int(rng.integers(1, 4))), "-grs", str(int(rng.choice([1, 1, 2, 16])))]
    if rng.uniform() < 0.4: f.append("-t")
    return f


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=60); ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--blender-scenes", action="store_true", help="random flag sets and film sizes on the reference's own Blender scenes")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0; worst = 0
    names = sorted(p.stem for p in (cases.GOLDEN / "scenes").glob("*.json"))
    for k in range(args.cases):
        if args.blender_scenes:
            w = int(rng.integers(40, 140)); h = int(rng.integers(30, 90))
            text = cases.blend_scene(names[k % len(names)], w, h); flags = random_flags(rng)
        else:
            text = random_scene(rng, (96, 54)); flags = random_flags(rng)
        ref, ost = oracle.render(text, flags, str(cases.TEXTURES))
        img, st = api.render_scene_text(text, flags, str(cases.TEXTURES))
        d = np.abs(img.astype(np.int32) - ref)
        npx = int((d.max(axis=2) > 0).sum()); worst = max(worst, int(d.max()))
        rays_ok = (st.primary_rays, st.shadow_rays_ref, st.reflection_rays_ref, st.refraction_rays_ref) == (ost.primary, ost.shadow, ost.reflection, ost.refraction)
        if k % 200 == 199:
            print(f"... {k + 1} cases", flush=True)
        if npx or not rays_ok:
            bad += 1
            print(f"case {k} seed {args.seed} flags {flags}: {npx} pixels differ, max {int(d.max())} LSB, rays equal: {rays_ok}", flush=True)
    print(f"{args.cases} cases, {bad} with differences, worst {worst} LSB")
    sys.exit(1 if worst > 1 else 0)


if __name__ == "__main__":
    main()
