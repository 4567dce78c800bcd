"""Scene / flag-set catalogue shared by the golden generator (tests/golden/make_golden.py) and the tests."""
from __future__ import annotations

import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

from raytracer_amd import scenegen as sg  # noqa: E402

GOLDEN = ROOT / "tests" / "golden"
TEXTURES = GOLDEN / "textures"

_RES_RE = re.compile(r'("film_resolution"\s*:\s*\{\s*"width"\s*:\s*)([0-9.]+)(\s*,\s*"height"\s*:\s*)([0-9.]+)')


def with_resolution(text: str, w: int, h: int) -> str:
    out, n = _RES_RE.subn(lambda m: f"{m.group(1)}{w}{m.group(3)}{h}", text, count=1)
    assert n == 1
    return out


def reference_scene(w: int = 1920, h: int = 1080) -> str:
    """The reference's own ASCII/scene.json (committed as a data fixture), optionally at another film size."""
    return with_resolution((GOLDEN / "scene.json").read_text(), w, h)


def blend_scene(name: str, w: int, h: int) -> str:
    """A scene of the reference that exists only as Blend/<name>.blend, exported to scene.json text by
    raytracer_amd.blend_export (tests/golden/make_scenes.py) and committed as a data fixture."""
    return with_resolution((GOLDEN / "scenes" / f"{name}.json").read_text(), w, h)


def mirror_scene(res=(200, 112)) -> str:
    """A fully lit, strongly reflecting wall behind a sphere that hovers over a matt floor: the wall's own shading is
    deterministic, the soft shadow on the floor is seen only through the reflection."""
    cams = [sg.camera(location=(0, -6, 2.5), gaze=(0, 1, -0.2), res=res)]
    lights = [sg.light((3, -2, 7), 400.0)]
    floor = sg.plane([(-8, -8, 0), (8, -8, 0), (-8, 8, 0), (8, 8, 0)], sg.material((0.6, 0.6, 0.6), (0, 0, 0), 10.0))
    wall = sg.plane([(-7, 6, 0), (7, 6, 0), (-7, 6, 7), (7, 6, 7)], sg.material((0.05, 0.05, 0.05), (1, 1, 1), 200.0))
    ball = sg.sphere((0.5, 1.5, 3.0), 0.9, sg.material((0.8, 0.3, 0.2), (0, 0, 0), 10.0))
    box = sg.cube((-3, 3, 0.5), 0.5, sg.material((0.2, 0.3, 0.8), (0, 0, 0), 10.0), (0, 0, 0.4))
    return sg.scene_text(cams, lights, [ball], [box], [floor, wall])


SCENES = {
    "mirror": mirror_scene,
    "ref": lambda: reference_scene(240, 135),
    # the reference's own Blender scenes
    "bvh5": lambda: blend_scene("bvh_vs_unaccel_5", 240, 135),
    "texmap1": lambda: blend_scene("texture_mapping1", 200, 112),
    "cube_mb": lambda: blend_scene("cube_mb", 160, 90),
    "blinn": lambda: blend_scene("blinn_phong_test", 200, 112),
    "test1": lambda: blend_scene("Test1", 200, 112),
    "test2": lambda: blend_scene("Test2", 160, 90),
    **{name: (lambda n=name: blend_scene(n, 160, 90)) for name in
       ("bvh_vs_unaccel_1", "bvh_vs_unaccel_2", "bvh_vs_unaccel_3", "bvh_vs_unaccel_4", "cube", "exported", "exported_rays", "scene",
        "Test3", "test_personal_1", "test_personal_2", "texture_mapping2")},
    "ref120": lambda: reference_scene(120, 68),
    "glass": lambda: sg.glass_scene((200, 112)),
    "glass_tex": lambda: sg.glass_scene((200, 112), True),
    "glass_tex120": lambda: sg.glass_scene((120, 68), True),
    "cluster16": lambda: sg.cluster_scene(16, 7, (200, 112)),
    "rand200": lambda: sg.random_scene(200, 42, (160, 90), rmin=0.2, rmax=0.6, transparent_frac=0.2, moving=True),
    # dense enough that the depth-8 tree's leaves hold tens of primitives (second-level hierarchy on the device)
    "rand300m": lambda: sg.random_scene(300, 5, (96, 54), rmin=0.15, rmax=0.45, moving=True, extent=6.0),
    "rand2k": lambda: sg.random_scene(2000, 11, (128, 72), rmin=0.15, rmax=0.5, transparent_frac=0.15, moving=True, extent=12.0),
    "rand2k64": lambda: sg.random_scene(2000, 11, (64, 36), rmin=0.15, rmax=0.5, transparent_frac=0.15, moving=True, extent=12.0),
}

# Deterministic flag sets: the image is a pure function of scene + flags; the bar is +-1 LSB per channel.
DETERMINISTIC = [
    ("ref", []),
    ("ref", ["-u"]),
    ("ref", ["-r"]),
    ("ref", ["-rd", "4", "-grs", "1"]),
    ("ref", ["-rd", "3", "-grs", "2"]),
    ("glass", []),
    ("glass", ["-u"]),
    ("glass", ["-rd", "3", "-grs", "1"]),
    ("glass", ["-u", "-rd", "3", "-grs", "1"]),
    ("glass", ["-rd", "2"]),
    ("glass_tex", ["-t"]),
    ("glass_tex", ["-t", "-u"]),
    ("glass_tex", ["-t", "-rd", "2", "-grs", "1"]),
    ("cluster16", []),
    ("cluster16", ["-u"]),
    ("cluster16", ["-rd", "3", "-grs", "1"]),
    ("cluster16", ["-u", "-rd", "3", "-grs", "1"]),
    ("rand200", []),
    ("rand200", ["-u"]),
    ("rand200", ["-rd", "2", "-grs", "1"]),
    ("rand2k", []),
    ("rand2k", ["-u"]),
    ("rand2k", ["-rd", "2", "-grs", "1"]),
    ("bvh5", []),
    ("bvh5", ["-u"]),
    ("bvh5", ["-rd", "3", "-grs", "1"]),
    ("texmap1", ["-t"]),
    ("texmap1", ["-t", "-u", "-r", "-grs", "1"]),
    ("cube_mb", []),
    ("blinn", ["-r", "-grs", "1"]),
    ("test1", ["-rd", "2", "-grs", "1"]),
    ("test2", []),
    ("test2", ["-u"]),
    # every other Blender scene of the reference (scene.blend has no light, exported*.blend two moving objects)
    *[(name, flags) for name in ("bvh_vs_unaccel_1", "bvh_vs_unaccel_2", "bvh_vs_unaccel_3", "bvh_vs_unaccel_4", "cube", "exported",
                                 "exported_rays", "scene", "Test3", "test_personal_1", "test_personal_2", "texture_mapping2")
      for flags in ([], ["-u", "-rd", "2", "-grs", "1", "-t"])],
]

# Stochastic flag sets: judged by PSNR >= 40 dB between two high-sample renders (the golden one is the reference's).
STOCHASTIC = [
    ("ref", ["-aa", "-aas", "256"]),
    ("ref", ["-ss", "-sss", "256"]),
    ("ref", ["-dof", "-dofs", "256"]),
    ("ref", ["-m", "-mbs", "256"]),
    ("ref", ["-r", "-gr", "-grs", "256"]),
    ("glass_tex", ["-t", "-aa", "-aas", "256", "-r", "-grs", "1"]),
    # moving primitives inside crowded leaves (second-level hierarchy over swept bounds)
    ("rand300m", ["-m", "-mbs", "128", "-r", "-grs", "1"]),
    # the flag set of the 4K multi-sample configuration (4096 samples per pixel) on a small film
    ("ref120", ["-aa", "-aas", "16", "-gr", "-grs", "16", "-dof", "-dofs", "16", "-m", "-mbs", "16"]),
    # the reference's motion-blur scene and its 18-object scene with soft shadows
    ("cube_mb", ["-m", "-mbs", "256"]),
    ("bvh5", ["-ss", "-sss", "256"]),
    # soft shadows below the primary hit: computeSoftShadows runs inside blinnPhong at every recursion level
    # (raytracer.cpp:330-332 under :366)
    ("ref", ["-rd", "2", "-grs", "1", "-ss", "-sss", "256"]),
    ("glass_tex", ["-t", "-rd", "2", "-grs", "1", "-ss", "-sss", "256"]),
    ("bvh5", ["-r", "-grs", "1", "-ss", "-sss", "128"]),
]

# Glossy reflection combined with soft shadows (every hit of a rndConeDirection ray runs computeSoftShadows,
# raytracer.cpp:400-413 under :330-332), and the headline configuration at its own depth.  GPU side: the mean of
# SEED_MEAN_SEEDS renders with `flags` and different seeds; golden side: ONE render of the reference binary with
# `golden_flags` (more samples per estimate; the expectation is the same up to the clamps' non-linearity).
SEED_MEAN = [
    # (scene, flags of each GPU render, flags of the reference's render)
    ("ref", ["-r", "-gr", "-grs", "16", "-ss", "-sss", "16"], ["-r", "-gr", "-grs", "64", "-ss", "-sss", "64"]),
    ("glass_tex120", ["-t", "-rd", "2", "-gr", "-grs", "16", "-ss", "-sss", "16"], ["-t", "-rd", "2", "-gr", "-grs", "32", "-ss", "-sss", "64"]),
    ("ref", ["-rd", "4", "-ss", "-sss", "16"], ["-rd", "4", "-grs", "1", "-ss", "-sss", "256"]),
]
SEED_MEAN_SEEDS = 16

# The all-effects flag set of the largest configuration (-aa -dof -m -t -ss -r -gr together) on a crowded scene whose
# primitives move inside second-level hierarchies.  The reference binary needs days for it; the CPU restatement renders
# it in bands (make_seeded.py).  Two fixtures: the flag set itself with two different rand() seeds (the noise floor of
# 64 samples per pixel), and the same estimators at 256 primary samples per pixel (own jitter, lens point and time each)
# against which the mean of ALL_EFFECTS_SEEDS GPU renders is held.
ALL_EFFECTS = ("rand2k64", ["-aa", "-aas", "4", "-dof", "-dofs", "4", "-m", "-mbs", "4", "-t", "-ss", "-sss", "16", "-r", "-gr", "-grs", "4"])
ALL_EFFECTS_HIGH = ["-aa", "-aas", "256", "-dof", "-dofs", "1", "-m", "-mbs", "1", "-t", "-ss", "-sss", "16", "-r", "-gr", "-grs", "4"]
ALL_EFFECTS_SEEDS = 16

# Soft shadows under identical reflection rays at depth >= 2: the device traces the FIRST reflection of a path gr_samples
# times and deeper ones once (rt_api.h, replicate_reflections); the mean image must still be the restatement's, whose
# every level re-traces gr_samples rays with their own shadow samples (raytracer.cpp:400-432).
DEEP_GLOSS = [("mirror", ["-rd", "3", "-grs", "4", "-ss", "-sss", "4"])]
DEEP_GLOSS_SEEDS = 32

# A crowded random scene with every effect that has a sample count switched on at once (moving primitives in
# second-level hierarchies, soft shadows, reflection and refraction).  The reference needs hours for this frame on one
# core, so the high-sample image comes from the CPU restatement rendered in horizontal bands on all cores (each band
# with its own rand() seed: a sample of the same distribution, not the byte-exact image of the reference binary).
STOCHASTIC_BANDED = [
    ("rand2k", ["-t", "-m", "-mbs", "64", "-ss", "-sss", "64", "-r", "-grs", "1"]),
]
BANDS = 24


def case_id(scene: str, flags: list[str]) -> str:
    return scene + ("_" + "".join(f.replace("-", "") for f in flags) if flags else "_default")


# Cases judged against the sampling noise itself: two renders of the CPU restatement with different rand() seeds
# (seed 1 = the reference binary's own output) give the noise floor a GPU render has to sit on.
NOISE_FLOOR = [
    ("glass_tex", ["-t", "-aa", "-aas", "16", "-ss", "-sss", "16", "-r", "-grs", "1"]),
    ("ref", ["-aa", "-aas", "4", "-dof", "-dofs", "8", "-m", "-mbs", "8", "-ss", "-sss", "8"]),
]
# Per-pixel mean and variance over MIRROR_SEEDS renders of the mirror scene: soft shadows seen through gr_samples
# identical reflection rays, each of which draws its own shadow samples in the reference
MIRROR_FLAGS = ["-r", "-ss", "-sss", "4"]
MIRROR_DIRECT_FLAGS = ["-ss", "-sss", "4"]
MIRROR_SEEDS = 48
