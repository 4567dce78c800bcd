"""A sparse sample of the CPU restatement's frame of the large-scene configuration at its own size: 10^6 random
primitives (scenegen.random_scene, seed 42), 1920x1080, with the BVH and with -u.  The whole frame is out of the
restatement's reach (every leaf of the reference's depth-8 tree holds thousands of primitives); every 40th column of
every 40th row - 1296 pixels spread over the frame, hence over as many different primitives - is not.
Output: windows_1m.npz  ->  grid_bvh, grid_u : int32 [27, 48, 3], pixel (x, y) = (40 * column + 20, 40 * row + 20)
Takes about fifteen minutes and 9 GB (the scene text alone is 1.1 GB)."""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import cases  # noqa: E402
import oracle  # noqa: E402
from raytracer_amd import scenegen as sg  # noqa: E402

N, SEED, RES = 1_000_000, 42, (1920, 1080)
STRIDE, FIRST = 40, 20


def main():
    text = sg.random_scene(N, SEED, RES)
    out = {}
    for flags, tag in (([], "bvh"), (["-u"], "u")):
        img, _ = oracle.render(text, flags, window=(FIRST, FIRST, RES[0], RES[1]), stride=(STRIDE, STRIDE))
        out[f"grid_{tag}"] = img[FIRST::STRIDE, FIRST::STRIDE].astype(np.int32)
        print(tag, out[f"grid_{tag}"].shape, len(np.unique(out[f"grid_{tag}"].reshape(-1, 3), axis=0)), "colours", flush=True)
    np.savez_compressed(HERE / "windows_1m.npz", **out)


if __name__ == "__main__":
    main()
