"""Synthetic scene.json writers (same schema as the Blender exporter, Blend/Export.py:279-287 of the reference).

Used by tests and bench.py to make inputs; produces TEXT, so every consumer (the reference binary, the CPU
restatement, the product loader) parses exactly the same bytes.
"""
from __future__ import annotations

import json
import math

import numpy as np


def _xyz(v):
    return {"x": float(v[0]), "y": float(v[1]), "z": float(v[2])}


def _rgb(v):
    return {"r": float(v[0]), "g": float(v[1]), "b": float(v[2])}


def material(diffuse=(0.8, 0.8, 0.8), specular=(0.3, 0.3, 0.3), shininess=64.0, transparency=0.0, ior=1.5,
             texture=None):
    return {"diffuse": _rgb(diffuse), "specular": _rgb(specular), "shininess": float(shininess),
            "transparency": float(transparency), "ior": float(ior), "texture": texture}


def camera(location=(0.0, -8.0, 1.5), gaze=(0.0, 1.0, 0.0), aperture=1.0, focal_distance=7.0,
           focal_length=50.0, sensor=(36.0, 24.0), res=(1920, 1080)):
    return {"location": _xyz(location), "gaze_vector": _xyz(gaze), "up_vector": _xyz((0.0, 0.0, 1.0)),
            "aperture": float(aperture), "focal_distance": float(focal_distance),
            "focal_length": float(focal_length), "sensor": {"width": float(sensor[0]), "height": float(sensor[1])},
            "film_resolution": {"width": int(res[0]), "height": int(res[1])}}


def light(location, intensity=1000.0):
    return {"location": _xyz(location), "radiant_intensity": float(intensity)}


def sphere(start, radius, mat, end=None):
    return {"start_location": _xyz(start), "end_location": _xyz(end if end is not None else start),
            "radius": float(radius), "material": mat}


def cube(start, scale, mat, rotation=(0.0, 0.0, 0.0), end=None):
    return {"start_location": _xyz(start), "end_location": _xyz(end if end is not None else start),
            "rotation": _xyz(rotation), "scale": float(scale), "material": mat}


def plane(corners, mat):
    return {"corners": [_xyz(c) for c in corners], "material": mat}


def _fmt(o, indent=0):
    """json.dumps, except that floats are printed without exponent '+' signs or 'E' (the reference's number
    scanner accepts only [0-9.-e], shape.cpp:32) and never in a form a light's scanner would cut short
    (pointlight.cpp:97 has no 'e')."""
    return json.dumps(o, indent=4)


def scene_text(cams, lights, spheres, cubes, planes) -> str:
    txt = _fmt({"properties": {"cameras": cams, "point_lights": lights, "spheres": spheres, "cubes": cubes,
                               "planes": planes}})
    if "e+" in txt or "E" in txt:
        raise ValueError("number format the reference scanner cannot read")
    return txt


def ground(size=50.0, z=0.0, mat=None):
    h = size / 2.0
    # Blender vertex order, not winding order (Export.py:109-112 of the reference)
    return plane([(-h, -h, z), (h, -h, z), (-h, h, z), (h, h, z)], mat or material((0.5, 0.5, 0.5)))


def glass_scene(res=(240, 135), textured=False) -> str:
    """Small hand-made scene exercising refraction, two lights, scaled/rotated cubes, a tilted polygon,
    and (optionally) textured sphere/cube/plane."""
    tex_s = "checkerboard.ppm" if textured else None
    tex_c = "tex2.ppm" if textured else None
    tex_p = "checkerboard.ppm" if textured else None
    cams = [camera(location=(0.5, -9.0, 2.5), gaze=(-0.05, 1.0, -0.15), aperture=2.0, focal_distance=8.0, res=res)]
    lights = [light((4.0, -2.0, 6.0), 800.0), light((-5.0, -4.0, 3.0), 300.0)]
    spheres = [
        sphere((-1.5, 0.0, 1.0), 1.0, material((0.9, 0.9, 0.9), (0.3, 0.3, 0.3), 100.0, 0.8, 1.5)),
        sphere((1.8, 1.5, 0.8), 0.8, material((0.1, 0.2, 0.9), (0.2, 0.25, 0.3), 30.0, 0.0, 1.5, tex_s),
               end=(1.8, 1.5, 2.0)),
        sphere((0.2, -2.5, 0.4), 0.4, material((0.9, 0.8, 0.1), (0.0, 0.0, 0.0), 10.0, 0.4, 1.2)),
    ]
    cubes = [
        cube((2.5, -1.5, 0.6), 0.6, material((0.8, 0.1, 0.1), (0.3, 0.3, 0.3), 64.0, 0.0, 1.45, tex_c),
             rotation=(0.2, 0.4, 0.9)),
        cube((-3.0, 2.0, 1.5), 1.5, material((0.2, 0.8, 0.3), (0.1, 0.1, 0.1), 5.0, 0.3, 1.3),
             rotation=(0.0, 0.0, 0.5), end=(-3.0, 2.0, 2.5)),
    ]
    planes = [
        plane([(-6, -6, 0), (6, -6, 0), (-6, 6, 0), (6, 6, 0)], material((0.4, 0.4, 0.4), (0.3, 0.3, 0.3), 64.0, 0.0, 1.5, tex_p)),
        plane([(-6, 6, 0), (6, 6, 0), (-6, 6.5, 5), (6, 6.5, 5)], material((0.7, 0.6, 0.5), (0.05, 0.05, 0.05), 20.0)),
        plane([(-1, -4, 0.5), (0, -5, 0.5), (1, -4, 0.5), (0.5, -3, 0.9), (-0.5, -3, 0.9)],
              material((0.3, 0.3, 0.9), (0.3, 0.3, 0.3), 90.0, 0.5, 1.1)),
    ]
    return scene_text(cams, lights, spheres, cubes, planes)


GROUND_DIFFUSE = (0.0, 0.12389342486858368, 0.002572941593825817)
REF_LIGHT = (4.076245307922363, 1.0054539442062378, 5.903861999511719)


def random_prims(n_prims: int, seed: int = 42, moving=False, transparent_frac=0.0, extent=20.0, rmin=0.02, rmax=0.1):
    """The random population behind random_scene / random_scene_arrays, vectorised: one table of 14 float64
    columns per primitive (centre xyz, size, diffuse rgb, transparency, end z offset, rotation xyz, is_cube, pad)."""
    rng = np.random.default_rng(seed)
    t = np.zeros((n_prims, 14), dtype=np.float64)
    t[:, 0] = rng.uniform(-extent, extent, n_prims)
    t[:, 1] = rng.uniform(-extent, extent, n_prims)
    t[:, 2] = rng.uniform(0.0, 10.0, n_prims)
    t[:, 3] = rng.uniform(rmin, rmax, n_prims)
    t[:, 4:7] = rng.uniform(0.0, 1.0, (n_prims, 3))
    t[:, 7] = np.where(rng.uniform(0.0, 1.0, n_prims) < transparent_frac, 0.6, 0.0)
    t[:, 8] = rng.uniform(0.0, 1.0, n_prims) if moving else 0.0
    t[:, 9:12] = rng.uniform(0.0, 2.0 * math.pi, (n_prims, 3))
    t[:, 12] = np.arange(n_prims) % 2
    return t


def random_scene(n_prims: int, seed: int = 42, res=(1920, 1080), with_ground=True, moving=False,
                 transparent_frac=0.0, extent=20.0, rmin=0.02, rmax=0.1) -> str:
    """Random spheres and cubes over a ground plane: 50 % spheres r in U[rmin,rmax], 50 % cubes scale in
    U[rmin,rmax] with rotation in U[0,2pi)^3, centres U[-extent,extent]^2 x [0,10], diffuse U[0,1]^3,
    specular 0.3, shininess 64; camera and light of the reference's scene.json."""
    t = random_prims(n_prims, seed, moving, transparent_frac, extent, rmin, rmax)
    cams = [camera(res=res)]
    lights = [light(REF_LIGHT, 1000.0)]
    spheres, cubes = [], []
    for row in t:
        c = (row[0], row[1], row[2])
        m = material(row[4:7], (0.3, 0.3, 0.3), 64.0, row[7], 1.5)
        end = (c[0], c[1], c[2] + row[8]) if moving else None
        if row[12] == 0:
            spheres.append(sphere(c, row[3], m, end=end))
        else:
            cubes.append(cube(c, row[3], m, rotation=row[9:12], end=end))
    planes = [ground(50.0, 0.0, material(GROUND_DIFFUSE))] if with_ground else []
    return scene_text(cams, lights, spheres, cubes, planes)


def cluster_scene(n_prims: int = 16, seed: int = 7, res=(480, 270)) -> str:
    """A stand-in for the reference's bvh_vs_unaccel_* scenes (present only as .blend files): a handful of
    spheres and cubes of ordinary size in front of the camera, above a ground plane."""
    rng = np.random.default_rng(seed)
    cams = [camera(res=res)]
    lights = [light((4.0, 1.0, 6.0), 1000.0)]
    spheres, cubes = [], []
    for i in range(n_prims):
        c = (rng.uniform(-4.0, 4.0), rng.uniform(-2.0, 10.0), rng.uniform(0.3, 3.0))
        r = rng.uniform(0.3, 0.8)
        m = material(rng.uniform(0.0, 1.0, 3), (0.3, 0.3, 0.3), float(rng.uniform(8.0, 120.0)),
                     0.7 if i % 5 == 4 else 0.0, 1.45)
        if i % 2 == 0:
            spheres.append(sphere(c, r, m))
        else:
            cubes.append(cube(c, r, m, rotation=rng.uniform(0.0, 2.0 * math.pi, 3)))
    planes = [ground(30.0, 0.0, material((0.3, 0.35, 0.3)))]
    return scene_text(cams, lights, spheres, cubes, planes)
