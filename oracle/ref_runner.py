"""Run the compiled, unmodified reference (oracle/_ref/raytracer_ref) on a scene and return its image.

TEST INFRASTRUCTURE ONLY. Nothing in the product imports this file.

The reference binary hard-codes its file layout (cwd = Code/):
  * scene:    the literal relative path  ..\\ASCII\\scene.json   (camera.cpp:18, sphere.cpp:21, ...) which on
              Linux is a file in cwd whose NAME contains backslashes;
  * template: ../Textures/blank_1920x1080.ppm                   (raytracer.cpp:63, image.cpp:15) whose
              width/height become the output size (image.cpp:28,90);
  * textures: ../Textures/<name>                                (image.cpp:15);
  * output:   ../Output/<-o name>                               (image.cpp:84).
This module stages that layout in a temporary directory, synthesises the (black) template at the scene's
film_resolution, runs the binary, and parses the P3 text it writes.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import tempfile
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF_BIN = HERE / "_ref" / "raytracer_ref"
# the reference's own front end (parsers, BVHNode::buildBVH, Image) bound to the HIP library in place of its pixel loop
# (oracle/ref_bind.cpp, `make -C oracle bind`); needs a GPU to run
BIND_BIN = HERE / "_ref" / "ref_bind"

_RES_RE = re.compile(r'("film_resolution"\s*:\s*\{\s*"width"\s*:\s*)([0-9.]+)(\s*,\s*"height"\s*:\s*)([0-9.]+)')


def have_ref() -> bool:
    return REF_BIN.exists() and os.access(REF_BIN, os.X_OK)


def have_bind() -> bool:
    return BIND_BIN.exists() and os.access(BIND_BIN, os.X_OK)


def set_resolution(scene_text: str, width: int, height: int) -> str:
    """Return scene_text with the first camera's film_resolution replaced (Q21: loop bounds = camera res)."""
    out, n = _RES_RE.subn(lambda m: f"{m.group(1)}{width}{m.group(3)}{height}", scene_text, count=1)
    if n != 1:
        raise ValueError("film_resolution block not found in scene text")
    return out


def get_resolution(scene_text: str) -> tuple[int, int]:
    m = _RES_RE.search(scene_text)
    if not m:
        raise ValueError("film_resolution block not found in scene text")
    return int(float(m.group(2))), int(float(m.group(4)))


def write_blank_template(path: Path, width: int, height: int) -> None:
    row = ("0 0 0 " * width).rstrip() + "\n"
    with open(path, "w") as f:
        f.write(f"P3\n{width} {height}\n255\n")
        for _ in range(height):
            f.write(row)


def parse_p3(path: Path) -> np.ndarray:
    toks = Path(path).read_bytes().split()
    if toks[0] != b"P3":
        raise ValueError("not a P3 file")
    w, h = int(toks[1]), int(toks[2])
    data = np.array(toks[4:4 + w * h * 3], dtype=np.int64)
    return data.reshape(h, w, 3).astype(np.int32)


def run_ref(scene_text: str, flags: list[str], textures: dict[str, Path] | None = None,
            timeout: float = 3600.0, keep_dir: str | None = None, exe: Path | None = None):
    """Render scene_text with the reference binary (or another executable that keeps its file layout, `exe`).
    Returns (image int32 [H,W,3], loop_seconds, stdout)."""
    exe = Path(exe) if exe else REF_BIN
    if not (exe.exists() and os.access(exe, os.X_OK)):
        raise FileNotFoundError(f"{exe} not built (make -C oracle ref / bind)")
    w, h = get_resolution(scene_text)
    root = Path(keep_dir) if keep_dir else Path(tempfile.mkdtemp(prefix="rtref_"))
    try:
        for d in ("Code", "Textures", "Output"):
            (root / d).mkdir(parents=True, exist_ok=True)
        (root / "Code" / "..\\ASCII\\scene.json").write_text(scene_text)
        write_blank_template(root / "Textures" / "blank_1920x1080.ppm", w, h)
        for name, src in (textures or {}).items():
            shutil.copyfile(src, root / "Textures" / name)
        t0 = time.time()
        p = subprocess.run([str(exe), *flags, "-o", "out.ppm"], cwd=root / "Code",
                           capture_output=True, text=True, timeout=timeout)
        wall = time.time() - t0
        if p.returncode != 0:
            raise RuntimeError(f"reference failed rc={p.returncode}: {p.stderr[-2000:]}")
        m = re.search(r"Time taken: ([0-9.eE+-]+)s", p.stdout)
        loop_s = float(m.group(1)) if m else wall
        img = parse_p3(root / "Output" / "out.ppm")
        return img, loop_s, p.stdout
    finally:
        if not keep_dir:
            shutil.rmtree(root, ignore_errors=True)
