"""A sparse sample of the CPU restatement's frame of the largest configuration's scene at its own size: 10^7 random
primitives (scenegen.random_prims, seed 43) at 3840x2160, under deterministic flag sets (the configuration's own effects
draw random samples, which no two implementations share).  The scene text - 3.4 GB - is written row by row from the
table (scenegen.random_scene would build ten million Python dicts first) in json.dumps's own number format.
Output: windows_10m.npz  ->  grid_bvh, grid_refl : int32 [27, 48, 3], pixel (x, y) = (80 * column + 40, 80 * row + 40)
                             grid_u : int32 [9, 16, 3], pixel (240 * column + 120, 240 * row + 120)
`python make_windows_10m.py check` writes nothing: it renders the 10^6-primitive scene of make_windows.py from a text made
the same way and compares it with windows_1m.npz."""
import json
import sys
from multiprocessing import Pool
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import cases  # noqa: E402,F401  (puts the repository and oracle/ on sys.path)
import oracle  # noqa: E402
from raytracer_amd import scenegen as sg  # noqa: E402

N, SEED, RES = 10_000_000, 43, (3840, 2160)


def _rows(t):
    """The spheres and the cubes of a slice of the table, as the elements of two JSON arrays."""
    sph, cub = [], []
    f = float
    for row in t:
        x, y, z, size, r, g, b, tr = (f(v) for v in row[:8])
        loc = f'{{"x": {x!r}, "y": {y!r}, "z": {z!r}}}'
        mat = (f'{{"diffuse": {{"r": {r!r}, "g": {g!r}, "b": {b!r}}}, "specular": {{"r": 0.3, "g": 0.3, "b": 0.3}}, '
               f'"shininess": 64.0, "transparency": {tr!r}, "ior": 1.5, "texture": null}}')
        if row[12] == 0:
            sph.append(f'{{"start_location": {loc}, "end_location": {loc}, "radius": {size!r}, "material": {mat}}}')
        else:
            rx, ry, rz = f(row[9]), f(row[10]), f(row[11])
            cub.append(f'{{"start_location": {loc}, "end_location": {loc}, "rotation": {{"x": {rx!r}, "y": {ry!r}, "z": {rz!r}}}, '
                       f'"scale": {size!r}, "material": {mat}}}')
    return ", ".join(sph).encode(), ", ".join(cub).encode()


def scene_bytes(n, seed, res):
    t = sg.random_prims(n, seed)
    if n < 200_000:
        parts = [_rows(t)]
    else:
        with Pool(7) as pool:
            parts = pool.map(_rows, np.array_split(t, max(7, n // 100_000)))
    head = json.dumps({"cameras": [sg.camera(res=res)], "point_lights": [sg.light(sg.REF_LIGHT, 1000.0)]})[1:-1]
    ground = json.dumps([sg.ground(50.0, 0.0, sg.material(sg.GROUND_DIFFUSE))])
    txt = b"".join([b'{"properties": {', head.encode(), b', "spheres": [', b", ".join(p[0] for p in parts if p[0]),
                    b'], "cubes": [', b", ".join(p[1] for p in parts if p[1]), b'], "planes": ', ground.encode(), b"}}"])
    if b"e+" in txt or b"E" in txt:
        raise ValueError("number format the reference scanner cannot read")
    return txt


_TXT = None        # the scene text, inherited by the forked workers (never copied: the pages are only read)


def _band(job):
    res, flags, stride, row0, row1 = job
    first = stride // 2
    img, _ = oracle.render(_TXT, flags, window=(first, first + row0 * stride, res[0], first + (row1 - 1) * stride + 1), stride=(stride, stride), res=res)
    return img[first + row0 * stride:first + (row1 - 1) * stride + 1:stride, first::stride].astype(np.int32)


def grid(txt, res, flags, stride, workers=7):
    """Rows of the sampled grid in bands, one process per band (the restatement is single-threaded, and a ray that the
    tree does not stop costs it a pass over all primitives)."""
    global _TXT
    _TXT = txt
    rows = len(range(stride // 2, res[1], stride))
    cuts = np.linspace(0, rows, min(workers, rows) + 1).astype(int)
    with Pool(len(cuts) - 1) as pool:
        bands = pool.map(_band, [(res, flags, stride, int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:]) if b > a])
    return np.concatenate(bands, axis=0)


def main():
    if sys.argv[1:] == ["check"]:
        txt = scene_bytes(1_000_000, 42, (1920, 1080))
        got = grid(txt, (1920, 1080), [], 40)
        want = np.load(HERE / "windows_1m.npz")["grid_bvh"]
        print("same as windows_1m.npz:", np.array_equal(got, want))
        return
    path = HERE / "windows_10m.npz"
    out = dict(np.load(path)) if path.exists() else {}          # a run that was cut short carries on with the next grid
    txt = scene_bytes(N, SEED, RES)
    print(len(txt) >> 20, "MiB of scene text", flush=True)
    for flags, tag, stride in (([], "bvh", 80), (["-r", "-rd", "2", "-grs", "1"], "refl", 80), (["-u"], "u", 240)):
        if f"grid_{tag}" in out:
            continue
        out[f"grid_{tag}"] = g = grid(txt, RES, flags, stride)
        print(tag, g.shape, len(np.unique(g.reshape(-1, 3), axis=0)), "colours", flush=True)
        np.savez_compressed(path, **out)


if __name__ == "__main__":
    main()
