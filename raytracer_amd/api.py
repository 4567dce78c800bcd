"""ctypes binding of raytracer_amd/librt_amd.so (include/rt_api.h + include/rt_host.h).

Plumbing only: every call below lands in the C ABI; there is no Python or CPU rendering path.  If the library
(and therefore the HIP kernels) is missing, importing the binding's `lib()` raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
import os

LIB_PATH = Path(os.environ.get("RT_AMD_LIB", HERE / "librt_amd.so"))   # the override is a development aid
TILE = 32

RT_OK = 0


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


class Camera(C.Structure):
    _fields_ = [("location", C.c_float * 3), ("gaze", C.c_float * 3), ("aperture", C.c_float),
                ("focal_length", C.c_float), ("focal_distance", C.c_float), ("sensor_width", C.c_float),
                ("sensor_height", C.c_float), ("res_x", C.c_float), ("res_y", C.c_float)]


class Light(C.Structure):
    _fields_ = [("location", C.c_float * 3), ("rad_intensity", C.c_float)]


class Material(C.Structure):
    _fields_ = [("diffuse", C.c_float * 3), ("specular", C.c_float * 3), ("shininess", C.c_float),
                ("transparency", C.c_float), ("ior", C.c_float), ("tex_id", C.c_int32)]


class Sphere(C.Structure):
    _fields_ = [("start", C.c_float * 3), ("end", C.c_float * 3), ("radius", C.c_float), ("mat", Material)]


class Cube(C.Structure):
    _fields_ = [("start", C.c_float * 3), ("end", C.c_float * 3), ("rotation", C.c_float * 3),
                ("scale", C.c_float), ("mat", Material)]


class Plane(C.Structure):
    _fields_ = [("n_verts", C.c_int32), ("first_vert", C.c_int32), ("normal", C.c_float * 3), ("mat", Material)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.POINTER(C.c_int32))]


class Scene(C.Structure):
    _fields_ = [("camera", Camera),
                ("n_lights", C.c_int32), ("lights", C.POINTER(Light)),
                ("n_spheres", C.c_int32), ("spheres", C.POINTER(Sphere)),
                ("n_cubes", C.c_int32), ("cubes", C.POINTER(Cube)),
                ("n_planes", C.c_int32), ("planes", C.POINTER(Plane)),
                ("n_plane_verts", C.c_int32), ("plane_verts", C.POINTER(C.c_float)),
                ("n_textures", C.c_int32), ("textures", C.POINTER(Texture))]


class BvhNode(C.Structure):
    _fields_ = [("box_min", C.c_float * 3), ("box_max", C.c_float * 3), ("left", C.c_int32), ("right", C.c_int32),
                ("first_prim", C.c_int32), ("n_planes", C.c_int32), ("n_cubes", C.c_int32), ("n_spheres", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("soft_shadows", C.c_int32), ("ss_samples", C.c_int32), ("glossy_reflect", C.c_int32),
                ("gr_samples", C.c_int32), ("light_radius", C.c_float), ("anti_aliasing", C.c_int32),
                ("aa_samples", C.c_int32), ("textures", C.c_int32), ("bvh", C.c_int32),
                ("reflect_refract_depth", C.c_int32), ("dof", C.c_int32), ("dof_samples", C.c_int32),
                ("motion_blur", C.c_int32), ("mb_samples", C.c_int32), ("seed", C.c_uint32), ("fallback", C.c_int32)]

    def samples_per_pixel(self) -> int:
        return ((self.aa_samples if self.anti_aliasing else 1) * (self.dof_samples if self.dof else 1) *
                (self.mb_samples if self.motion_blur else 1))


class Stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("reflection_rays", C.c_uint64),
                ("refraction_rays", C.c_uint64), ("nodes_shaded", C.c_uint64), ("shadow_rays_ref", C.c_uint64),
                ("reflection_rays_ref", C.c_uint64), ("refraction_rays_ref", C.c_uint64), ("batches", C.c_uint64),
                ("launches_extend_primary", C.c_uint64), ("launches_extend_secondary", C.c_uint64),
                ("launches_shadow", C.c_uint64), ("launches_shade", C.c_uint64),
                ("ms_total", C.c_double), ("ms_extend_primary", C.c_double), ("ms_extend_secondary", C.c_double),
                ("ms_shadow", C.c_double), ("ms_shade", C.c_double), ("ms_other", C.c_double)]

    def total_rays(self) -> int:
        return self.primary_rays + self.shadow_rays + self.reflection_rays + self.refraction_rays

    def total_rays_ref(self) -> int:
        return self.primary_rays + self.shadow_rays_ref + self.reflection_rays_ref + self.refraction_rays_ref


class HostScene(C.Structure):
    _fields_ = [("scene", Scene), ("n_cameras", C.c_int32), ("storage", C.c_void_p)]


class HostBvh(C.Structure):
    _fields_ = [("nodes", C.POINTER(BvhNode)), ("n_nodes", C.c_int32), ("prim_ids", C.POINTER(C.c_int32)),
                ("n_prim_ids", C.c_int32)]


# every symbol the two headers declare
API_SYMBOLS = ["rt_default_config", "rt_create", "rt_destroy", "rt_set_stream", "rt_get_stream", "rt_set_option", "rt_upload_scene", "rt_upload_bvh",
               "rt_set_shard", "rt_local_tiles", "rt_max_tiles_per_rank", "rt_render", "rt_render_device",
               "rt_assemble_tiles", "rt_synchronize", "rt_last_error", "rt_render_device_noalloc", "rt_noalloc_capacity",
               "rt_alloc_image", "rt_free_image",
               "rt_scene_layout_bytes", "rt_export_scene_layout", "rt_adopt_scene_layout", "rt_scene_layout_resolution",
               "rt_scene_array_count", "rt_scene_array", "rt_adopt_scene_from", "rt_share_scene_from",
               "rt_multi_create", "rt_multi_destroy", "rt_multi_set_option", "rt_multi_upload", "rt_multi_render", "rt_multi_render_device",
               "rt_multi_synchronize", "rt_multi_image", "rt_multi_context", "rt_multi_uses_rccl", "rt_multi_last_error"]
HOST_SYMBOLS = ["rt_host_parse_args", "rt_host_parse_scene", "rt_host_load_scene", "rt_host_free_scene",
                "rt_host_build_bvh", "rt_host_free_bvh", "rt_host_read_ppm", "rt_host_write_ppm", "rt_host_write_ppm_binary", "rt_host_free",
                "rt_host_camera_ray"]

_lib = None


def image_buffer(n_bytes: int) -> np.ndarray:
    """A flat uint8 array in page-locked host memory (rt_alloc_image), freed with the array: rt_render copies a frame
    into it by one DMA transfer."""
    import weakref
    p = lib().rt_alloc_image(int(n_bytes))
    if not p:
        raise MemoryError(f"rt_alloc_image({n_bytes}) failed")
    buf = (C.c_uint8 * int(n_bytes)).from_address(p)
    arr = np.frombuffer(buf, dtype=np.uint8)
    weakref.finalize(buf, lib().rt_free_image, p)
    return arr


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FileNotFoundError(f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
                                "(there is no fallback path)")
    L = C.CDLL(str(LIB_PATH))
    L.rt_default_config.argtypes = [C.POINTER(Config)]
    L.rt_create.argtypes = [C.c_int, C.c_uint64]; L.rt_create.restype = C.c_void_p
    L.rt_destroy.argtypes = [C.c_void_p]; L.rt_destroy.restype = None
    L.rt_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.rt_get_stream.argtypes = [C.c_void_p]; L.rt_get_stream.restype = C.c_void_p
    L.rt_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.rt_upload_scene.argtypes = [C.c_void_p, C.POINTER(Scene)]
    L.rt_upload_bvh.argtypes = [C.c_void_p, C.POINTER(BvhNode), C.c_int32, C.POINTER(C.c_int32), C.c_int32]
    L.rt_set_shard.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.rt_local_tiles.argtypes = [C.c_void_p]; L.rt_local_tiles.restype = C.c_int32
    L.rt_max_tiles_per_rank.argtypes = [C.c_void_p]; L.rt_max_tiles_per_rank.restype = C.c_int32
    L.rt_render.argtypes = [C.c_void_p, C.POINTER(Config), C.c_void_p, C.POINTER(Stats)]
    L.rt_render_device.argtypes = [C.c_void_p, C.POINTER(Config), C.c_void_p, C.POINTER(Stats)]
    L.rt_assemble_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.rt_render_device_noalloc.argtypes = [C.c_void_p, C.POINTER(Config), C.c_void_p]
    L.rt_noalloc_capacity.argtypes = []; L.rt_noalloc_capacity.restype = C.c_uint64
    L.rt_scene_layout_bytes.argtypes = []; L.rt_scene_layout_bytes.restype = C.c_uint64
    L.rt_export_scene_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.rt_adopt_scene_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.rt_scene_array_count.argtypes = [C.c_void_p]; L.rt_scene_array_count.restype = C.c_int32
    L.rt_scene_array.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.rt_adopt_scene_from.argtypes = [C.c_void_p, C.c_void_p]
    L.rt_share_scene_from.argtypes = [C.c_void_p, C.c_void_p]
    L.rt_scene_layout_resolution.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rt_alloc_image.argtypes = [C.c_uint64]; L.rt_alloc_image.restype = C.c_void_p
    L.rt_free_image.argtypes = [C.c_void_p]; L.rt_free_image.restype = None
    L.rt_multi_create.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.c_uint64]; L.rt_multi_create.restype = C.c_void_p
    L.rt_multi_destroy.argtypes = [C.c_void_p]; L.rt_multi_destroy.restype = None
    L.rt_multi_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.rt_multi_upload.argtypes = [C.c_void_p, C.POINTER(Scene), C.POINTER(BvhNode), C.c_int32, C.POINTER(C.c_int32), C.c_int32]
    L.rt_multi_render.argtypes = [C.c_void_p, C.POINTER(Config), C.c_void_p]
    L.rt_multi_render_device.argtypes = [C.c_void_p, C.POINTER(Config)]
    L.rt_multi_synchronize.argtypes = [C.c_void_p]
    L.rt_multi_image.argtypes = [C.c_void_p]; L.rt_multi_image.restype = C.c_void_p
    L.rt_multi_context.argtypes = [C.c_void_p, C.c_int32]; L.rt_multi_context.restype = C.c_void_p
    L.rt_multi_uses_rccl.argtypes = [C.c_void_p]; L.rt_multi_uses_rccl.restype = C.c_int32
    L.rt_multi_last_error.argtypes = [C.c_void_p]; L.rt_multi_last_error.restype = C.c_char_p
    L.rt_synchronize.argtypes = [C.c_void_p]
    L.rt_last_error.argtypes = [C.c_void_p]; L.rt_last_error.restype = C.c_char_p
    L.rt_host_parse_args.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.POINTER(Config), C.c_char_p, C.c_int]
    L.rt_host_parse_scene.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Config), C.c_char_p, C.c_int]
    L.rt_host_parse_scene.restype = C.POINTER(HostScene)
    L.rt_host_load_scene.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Config), C.c_char_p, C.c_int]
    L.rt_host_load_scene.restype = C.POINTER(HostScene)
    L.rt_host_free_scene.argtypes = [C.POINTER(HostScene)]; L.rt_host_free_scene.restype = None
    L.rt_host_build_bvh.argtypes = [C.POINTER(Scene), C.c_int32, C.POINTER(HostBvh)]
    L.rt_host_free_bvh.argtypes = [C.POINTER(HostBvh)]; L.rt_host_free_bvh.restype = None
    L.rt_host_read_ppm.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.POINTER(C.c_int32))]
    L.rt_host_write_ppm.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    L.rt_host_write_ppm_binary.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_void_p]
    L.rt_host_free.argtypes = [C.c_void_p]; L.rt_host_free.restype = None
    L.rt_host_camera_ray.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, C.POINTER(C.c_float)]
    L.rt_host_camera_ray.restype = None
    _lib = L
    return L


# ---- host surface ----------------------------------------------------------------------------------------

def parse_args(flags: list[str], extensions: bool = True) -> tuple[Config, str]:
    """Config::parseArgs of the reference (config.cpp:7-91) on a flag list (program name not included)."""
    argv = [b"raytracer"] + [f.encode() for f in flags]
    arr = (C.c_char_p * len(argv))(*argv)
    cfg = Config()
    name = C.create_string_buffer(512)
    lib().rt_host_parse_args(len(argv), arr, 1 if extensions else 0, C.byref(cfg), name, 512)
    return cfg, name.value.decode()


class LoadedScene:
    """A parsed scene (rt_host_parse_scene) and, on demand, its host-built BVH (rt_host_build_bvh)."""

    def __init__(self, text: str, cfg: Config, textures_dir: str | None = None):
        err = C.create_string_buffer(512)
        self._hs = lib().rt_host_parse_scene(text.encode(), textures_dir.encode() if textures_dir else None,
                                             C.byref(cfg), err, 512)
        if not self._hs:
            raise RtError(-6, err.value.decode())
        self._bvh = None

    @property
    def scene(self) -> Scene:
        return self._hs.contents.scene

    @property
    def width(self) -> int:
        return int(self.scene.camera.res_x)

    @property
    def height(self) -> int:
        return int(self.scene.camera.res_y)

    def bvh(self, max_depth: int = 8) -> HostBvh:
        if self._bvh is None:
            b = HostBvh()
            rc = lib().rt_host_build_bvh(C.byref(self._hs.contents.scene), max_depth, C.byref(b))
            if rc != RT_OK:
                raise RtError(rc, "rt_host_build_bvh failed")
            self._bvh = b
        return self._bvh

    def bvh_nodes(self) -> np.ndarray:
        """Pre-order dump comparable with the oracle's orc_bvh_dump: min3, max3, has_left, has_right, np, nc, ns."""
        b = self.bvh()
        out = np.zeros((b.n_nodes, 11), dtype=np.float32)
        for i in range(b.n_nodes):
            n = b.nodes[i]
            out[i, 0:3] = list(n.box_min); out[i, 3:6] = list(n.box_max)
            out[i, 6] = 1.0 if n.left >= 0 else 0.0; out[i, 7] = 1.0 if n.right >= 0 else 0.0
            out[i, 8] = n.n_planes; out[i, 9] = n.n_cubes; out[i, 10] = n.n_spheres
        return out

    def close(self):
        if self._bvh is not None:
            lib().rt_host_free_bvh(C.byref(self._bvh)); self._bvh = None
        if self._hs:
            lib().rt_host_free_scene(self._hs); self._hs = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ArrayScene(LoadedScene):
    """The population of scenegen.random_prims turned straight into the C-ABI arrays (no scene text): what a
    caller with millions of primitives does instead of writing and re-parsing hundreds of MB of JSON.
    Same values as parsing scenegen.random_scene's text (float32 of the same float64 numbers)."""

    def __init__(self, table: np.ndarray, res=(1920, 1080), with_ground=True, moving=False):
        from . import scenegen as sg
        self._hs = None; self._bvh = None
        t = np.asarray(table, dtype=np.float64)
        sph, cub = t[t[:, 12] == 0], t[t[:, 12] == 1]

        def mats(rows, arr):
            for k in range(3):
                arr["mat"]["diffuse"][:, k] = rows[:, 4 + k]
                arr["mat"]["specular"][:, k] = 0.3
            arr["mat"]["shininess"] = 64.0; arr["mat"]["transparency"] = rows[:, 7]; arr["mat"]["ior"] = 1.5
            arr["mat"]["tex_id"] = -1

        mat_dt = np.dtype([("diffuse", "<f4", 3), ("specular", "<f4", 3), ("shininess", "<f4"), ("transparency", "<f4"),
                           ("ior", "<f4"), ("tex_id", "<i4")])
        sdt = np.dtype([("start", "<f4", 3), ("end", "<f4", 3), ("radius", "<f4"), ("mat", mat_dt)])
        cdt = np.dtype([("start", "<f4", 3), ("end", "<f4", 3), ("rotation", "<f4", 3), ("scale", "<f4"), ("mat", mat_dt)])
        assert sdt.itemsize == C.sizeof(Sphere) and cdt.itemsize == C.sizeof(Cube)
        self._s = np.zeros(len(sph), dtype=sdt); self._c = np.zeros(len(cub), dtype=cdt)
        for arr, rows in ((self._s, sph), (self._c, cub)):
            arr["start"] = rows[:, 0:3]; arr["end"] = rows[:, 0:3]
            if moving:
                arr["end"][:, 2] = (rows[:, 2] + rows[:, 8]).astype(np.float32)
            if moving == "xyz":                                  # not only upwards (tests)
                arr["end"][:, 0] = (rows[:, 0] + 0.7 * rows[:, 8]).astype(np.float32)
                arr["end"][:, 1] = (rows[:, 1] - 0.4 * rows[:, 8]).astype(np.float32)
            mats(rows, arr)
        self._s["radius"] = sph[:, 3]; self._c["scale"] = cub[:, 3]; self._c["rotation"] = cub[:, 9:12]
        self._lights = (Light * 1)(); self._lights[0].location[:] = sg.REF_LIGHT; self._lights[0].rad_intensity = 1000.0
        n_planes = 1 if with_ground else 0
        self._planes = (Plane * max(1, n_planes))()
        self._verts = np.array([[-25, -25, 0], [25, -25, 0], [-25, 25, 0], [25, 25, 0]], dtype=np.float32)
        if with_ground:
            p = self._planes[0]; p.n_verts = 4; p.first_vert = 0; p.normal[:] = (0.0, 0.0, 1.0)
            p.mat.diffuse[:] = sg.GROUND_DIFFUSE; p.mat.specular[:] = (0.3, 0.3, 0.3)
            p.mat.shininess = 64.0; p.mat.transparency = 0.0; p.mat.ior = 1.5; p.mat.tex_id = -1
        sc = Scene()
        cam = sc.camera
        cam.location[:] = (0.0, -8.0, 1.5); cam.gaze[:] = (0.0, 1.0, 0.0); cam.aperture = 0.0
        cam.focal_length = np.float32(50.0) / np.float32(1000.0); cam.focal_distance = 0.0
        cam.sensor_width = np.float32(36.0) / np.float32(1000.0); cam.sensor_height = np.float32(24.0) / np.float32(1000.0)
        cam.res_x, cam.res_y = float(res[0]), float(res[1])
        sc.n_lights = 1; sc.lights = self._lights
        sc.n_spheres = len(self._s); sc.spheres = self._s.ctypes.data_as(C.POINTER(Sphere))
        sc.n_cubes = len(self._c); sc.cubes = self._c.ctypes.data_as(C.POINTER(Cube))
        sc.n_planes = n_planes; sc.planes = self._planes
        sc.n_plane_verts = 4 if with_ground else 0; sc.plane_verts = self._verts.ctypes.data_as(C.POINTER(C.c_float))
        sc.n_textures = 0; sc.textures = None
        self._scene = sc

    @property
    def scene(self) -> Scene:
        return self._scene

    def bvh(self, max_depth: int = 8) -> HostBvh:
        if self._bvh is None:
            b = HostBvh()
            rc = lib().rt_host_build_bvh(C.byref(self._scene), max_depth, C.byref(b))
            if rc != RT_OK:
                raise RtError(rc, "rt_host_build_bvh failed")
            self._bvh = b
        return self._bvh

    def close(self):
        if self._bvh is not None:
            lib().rt_host_free_bvh(C.byref(self._bvh)); self._bvh = None


# ---- one flat byte blob per scene: what rank 0 broadcasts to the other ranks -----------------------------------

_BLOB_MAGIC = 0x52544231      # "RTB1"


def pack_scene(ls: "LoadedScene", with_bvh: bool = True) -> np.ndarray:
    """Serialises the C-ABI arrays of a loaded scene (and its host-built BVH) into one uint8 array:
    int64 header [magic, n_sections, (offset, nbytes) per section] followed by the raw arrays."""
    sc = ls.scene

    def raw(ptr, count, ctype):
        if not count:
            return np.zeros(0, dtype=np.uint8)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * C.sizeof(ctype),)).copy()

    tex_meta = np.array([[sc.textures[i].width, sc.textures[i].height] for i in range(sc.n_textures)], dtype=np.int32).reshape(-1)
    tex_data = [raw(sc.textures[i].rgb, sc.textures[i].width * sc.textures[i].height * 3, C.c_int32) for i in range(sc.n_textures)]
    sections = [np.frombuffer(bytes(sc.camera), dtype=np.uint8), raw(sc.lights, sc.n_lights, Light), raw(sc.spheres, sc.n_spheres, Sphere),
                raw(sc.cubes, sc.n_cubes, Cube), raw(sc.planes, sc.n_planes, Plane), raw(sc.plane_verts, sc.n_plane_verts * 3, C.c_float),
                tex_meta.view(np.uint8), np.concatenate(tex_data) if tex_data else np.zeros(0, dtype=np.uint8)]
    if with_bvh:
        b = ls.bvh()
        sections += [raw(b.nodes, b.n_nodes, BvhNode), raw(b.prim_ids, b.n_prim_ids, C.c_int32)]
    else:
        sections += [np.zeros(0, dtype=np.uint8), np.zeros(0, dtype=np.uint8)]
    head = np.zeros(2 + 2 * len(sections), dtype=np.int64)
    head[0], head[1] = _BLOB_MAGIC, len(sections)
    off = head.nbytes
    for i, sec in enumerate(sections):
        head[2 + 2 * i], head[3 + 2 * i] = off, sec.nbytes
        off += (sec.nbytes + 15) // 16 * 16
    blob = np.zeros(off, dtype=np.uint8)
    blob[:head.nbytes] = head.view(np.uint8)
    for i, sec in enumerate(sections):
        blob[head[2 + 2 * i]:head[2 + 2 * i] + sec.nbytes] = sec
    return blob


class PackedScene(LoadedScene):
    """A scene (and BVH) rebuilt from pack_scene's blob: the receiving side of the scene broadcast."""

    def __init__(self, blob: np.ndarray):
        self._hs = None; self._bvh = None
        self._blob = np.ascontiguousarray(blob, dtype=np.uint8)
        head = self._blob[:16].view(np.int64)
        if int(head[0]) != _BLOB_MAGIC:
            raise RtError(-1, "not a scene blob")
        n = int(head[1])
        table = self._blob[16:16 + 16 * n].view(np.int64).reshape(n, 2)
        sec = [self._blob[int(o):int(o) + int(b)] for o, b in table]
        sc = Scene()
        C.memmove(C.byref(sc.camera), sec[0].ctypes.data, C.sizeof(Camera))

        def ptr(a, ctype):
            return C.cast(a.ctypes.data, C.POINTER(ctype)) if a.nbytes else None

        sc.n_lights = sec[1].nbytes // C.sizeof(Light); sc.lights = ptr(sec[1], Light)
        sc.n_spheres = sec[2].nbytes // C.sizeof(Sphere); sc.spheres = ptr(sec[2], Sphere)
        sc.n_cubes = sec[3].nbytes // C.sizeof(Cube); sc.cubes = ptr(sec[3], Cube)
        sc.n_planes = sec[4].nbytes // C.sizeof(Plane); sc.planes = ptr(sec[4], Plane)
        sc.n_plane_verts = sec[5].nbytes // 12; sc.plane_verts = ptr(sec[5], C.c_float)
        meta = sec[6].view(np.int32).reshape(-1, 2)
        self._tex = (Texture * max(1, len(meta)))()
        off = 0
        for i, (w, h) in enumerate(meta):
            self._tex[i].width, self._tex[i].height = int(w), int(h)
            self._tex[i].rgb = C.cast(sec[7].ctypes.data + off, C.POINTER(C.c_int32))
            off += int(w) * int(h) * 12
        sc.n_textures = len(meta); sc.textures = self._tex if len(meta) else None
        self._scene = sc
        self._packed_bvh = None
        if sec[8].nbytes:
            b = HostBvh()
            b.nodes = ptr(sec[8], BvhNode); b.n_nodes = sec[8].nbytes // C.sizeof(BvhNode)
            b.prim_ids = ptr(sec[9], C.c_int32); b.n_prim_ids = sec[9].nbytes // 4
            self._packed_bvh = b

    @property
    def scene(self) -> Scene:
        return self._scene

    def bvh(self, max_depth: int = 8) -> HostBvh:
        if self._packed_bvh is None:
            raise RtError(-3, "the blob carries no BVH")
        return self._packed_bvh

    def close(self):
        pass


def camera_ray(cam: Camera, x: float, y: float) -> np.ndarray:
    out = (C.c_float * 6)()
    lib().rt_host_camera_ray(C.byref(cam), x, y, out)
    return np.array(list(out), dtype=np.float32)


def read_ppm(path: str) -> np.ndarray:
    w, h, m = C.c_int32(), C.c_int32(), C.c_int32()
    p = C.POINTER(C.c_int32)()
    rc = lib().rt_host_read_ppm(path.encode(), C.byref(w), C.byref(h), C.byref(m), C.byref(p))
    if rc != RT_OK:
        raise RtError(rc, f"cannot read {path}")
    arr = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    lib().rt_host_free(p)
    return arr


def write_ppm(path: str, img: np.ndarray, maxval: int = 255, binary: bool = False) -> None:
    """The reference's text form (image.cpp:83-99), or binary P6 (an extension) with binary=True."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if binary:
        rc = lib().rt_host_write_ppm_binary(path.encode(), img.shape[1], img.shape[0], img.ctypes.data_as(C.c_void_p))
    else:
        rc = lib().rt_host_write_ppm(path.encode(), img.shape[1], img.shape[0], maxval, img.ctypes.data_as(C.c_void_p))
    if rc != RT_OK:
        raise RtError(rc, f"cannot write {path}")


# ---- render boundary -------------------------------------------------------------------------------------

def struct_res(layout: bytes) -> tuple[int, int]:
    """Film size inside a scene layout blob (rt_export_scene_layout), read by the library (the blob's struct is private to it)."""
    w, h = C.c_int32(), C.c_int32()
    if lib().rt_scene_layout_resolution(layout, len(layout), C.byref(w), C.byref(h)) != RT_OK:
        raise RtError(-1, "not a scene layout of this library version")
    return int(w.value), int(h.value)


class Renderer:
    """One rt_ctx on one GPU."""

    def __init__(self, device: int = 0, workspace_bytes: int = 0):
        self._ctx = lib().rt_create(device, workspace_bytes)
        if not self._ctx:
            raise RtError(-2, lib().rt_last_error(None).decode())
        self.width = self.height = 0
        self.rank, self.world = 0, 1

    def _check(self, rc: int):
        if rc != RT_OK:
            raise RtError(rc, lib().rt_last_error(self._ctx).decode())

    def set_stream(self, stream_handle: int | None):
        self._check(lib().rt_set_stream(self._ctx, C.c_void_p(stream_handle) if stream_handle else None))

    def stream_handle(self) -> int:
        """The hipStream_t the context enqueues on (wrap it with torch.cuda.ExternalStream to order torch work against it)."""
        return int(lib().rt_get_stream(self._ctx) or 0)

    def set_option(self, name: str, value: int):
        self._check(lib().rt_set_option(self._ctx, name.encode(), value))

    def upload(self, ls: LoadedScene, with_bvh: bool = True):
        self._check(lib().rt_upload_scene(self._ctx, C.byref(ls.scene)))
        self.width, self.height = ls.width, ls.height
        if with_bvh:
            b = ls.bvh()
            self._check(lib().rt_upload_bvh(self._ctx, b.nodes, b.n_nodes, b.prim_ids, b.n_prim_ids))
        else:
            self._check(lib().rt_upload_bvh(self._ctx, None, 0, None, 0))

    def set_shard(self, rank: int, world: int):
        self._check(lib().rt_set_shard(self._ctx, rank, world))
        self.rank, self.world = rank, world

    # ---- the device-resident scene handed to other contexts (rt_export_scene_layout and friends) ----
    def export_layout(self) -> bytes:
        """After upload(): the small blob another context adopts before the device arrays are sent to it."""
        n = int(lib().rt_scene_layout_bytes())
        buf = C.create_string_buffer(n)
        self._check(lib().rt_export_scene_layout(self._ctx, buf, n))
        return buf.raw

    def adopt_layout(self, layout: bytes):
        """Allocates the scene's device arrays as `layout` describes them; their contents must be sent next."""
        self._check(lib().rt_adopt_scene_layout(self._ctx, layout, len(layout)))
        w, h = struct_res(layout)
        self.width, self.height = w, h

    def scene_arrays(self) -> list[tuple[int, int]]:
        """(device pointer, bytes) of every array of the exported / adopted scene, in the same order on both sides."""
        out = []
        for k in range(lib().rt_scene_array_count(self._ctx)):
            p, n = C.c_void_p(), C.c_uint64()
            self._check(lib().rt_scene_array(self._ctx, k, C.byref(p), C.byref(n)))
            out.append((int(p.value or 0), int(n.value)))
        return out

    def adopt_from(self, other: "Renderer"):
        """The scene of another context of this process, copied device to device (no host work, no build)."""
        self._check(lib().rt_adopt_scene_from(self._ctx, other._ctx))
        self.width, self.height = other.width, other.height

    def share_from(self, other: "Renderer"):
        """The scene of another context of the same GPU, used in place: `other` owns the arrays and must outlive this use."""
        self._check(lib().rt_share_scene_from(self._ctx, other._ctx))
        self.width, self.height = other.width, other.height
        self._shares = other          # keeps the owner alive

    @property
    def local_tiles(self) -> int:
        return lib().rt_local_tiles(self._ctx)

    @property
    def max_tiles_per_rank(self) -> int:
        return lib().rt_max_tiles_per_rank(self._ctx)

    def out_bytes(self) -> int:
        if self.world > 1:
            return self.max_tiles_per_rank * TILE * TILE * 3
        return self.width * self.height * 3

    def render(self, cfg: Config, out: np.ndarray | None = None, want_stats: bool = True) -> tuple[np.ndarray, Stats | None]:
        """rt_render: result copied to host. world == 1 -> [H,W,3] uint8; else the flat tile-major block.
        out: a flat uint8 array of out_bytes() to render into (e.g. from image_buffer()); default a new array."""
        if out is None:
            out = np.zeros(self.out_bytes(), dtype=np.uint8)
        elif out.dtype != np.uint8 or out.size != self.out_bytes() or not out.flags["C_CONTIGUOUS"]:
            raise ValueError(f"out must be a contiguous uint8 array of {self.out_bytes()} bytes")
        st = Stats() if want_stats else None
        self._check(lib().rt_render(self._ctx, C.byref(cfg), out.ctypes.data_as(C.c_void_p), C.byref(st) if st else None))
        if self.world == 1:
            out = out.reshape(self.height, self.width, 3)
        return out, st

    def render_device(self, cfg: Config, out_ptr: int, want_stats: bool = False) -> Stats | None:
        st = Stats() if want_stats else None
        self._check(lib().rt_render_device(self._ctx, C.byref(cfg), C.c_void_p(out_ptr), C.byref(st) if st else None))
        return st

    def render_device_noalloc(self, cfg: Config, out_ptr: int):
        """rt_render_device_noalloc: work buffers from the library's per-GPU scratch array, nothing allocated."""
        self._check(lib().rt_render_device_noalloc(self._ctx, C.byref(cfg), C.c_void_p(out_ptr)))

    def assemble(self, tiles_ptr: int, world: int, out_ptr: int):
        self._check(lib().rt_assemble_tiles(self._ctx, C.c_void_p(tiles_ptr), world, C.c_void_p(out_ptr)))

    def synchronize(self):
        self._check(lib().rt_synchronize(self._ctx))

    def close(self):
        if self._ctx:
            lib().rt_destroy(self._ctx); self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiRenderer:
    """rt_multi: several GPUs of one node driven by this process (a device id may repeat for rehearsals)."""

    def __init__(self, devices: list[int], workspace_bytes: int = 0):
        arr = (C.c_int32 * len(devices))(*devices)
        self._m = lib().rt_multi_create(len(devices), arr, workspace_bytes)
        if not self._m:
            raise RtError(-2, lib().rt_multi_last_error(None).decode())
        self.n = len(devices); self.width = self.height = 0

    def _check(self, rc: int):
        if rc != RT_OK:
            raise RtError(rc, lib().rt_multi_last_error(self._m).decode())

    @property
    def uses_rccl(self) -> bool:
        return bool(lib().rt_multi_uses_rccl(self._m))

    def set_option(self, name: str, value: int):
        self._check(lib().rt_multi_set_option(self._m, name.encode(), value))

    def upload(self, ls: LoadedScene, with_bvh: bool = True):
        if with_bvh:
            b = ls.bvh()
            self._check(lib().rt_multi_upload(self._m, C.byref(ls.scene), b.nodes, b.n_nodes, b.prim_ids, b.n_prim_ids))
        else:
            self._check(lib().rt_multi_upload(self._m, C.byref(ls.scene), None, 0, None, 0))
        self.width, self.height = ls.width, ls.height

    def render(self, cfg: Config) -> np.ndarray:
        out = np.zeros(self.width * self.height * 3, dtype=np.uint8)
        self._check(lib().rt_multi_render(self._m, C.byref(cfg), out.ctypes.data_as(C.c_void_p)))
        return out.reshape(self.height, self.width, 3)

    def render_device(self, cfg: Config):
        self._check(lib().rt_multi_render_device(self._m, C.byref(cfg)))

    def synchronize(self):
        self._check(lib().rt_multi_synchronize(self._m))

    def image_ptr(self) -> int:
        return int(lib().rt_multi_image(self._m) or 0)

    def close(self):
        if self._m:
            lib().rt_multi_destroy(self._m); self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_scene_text(text: str, flags: list[str], textures_dir: str | None = None, device: int = 0,
                      workspace_bytes: int = 0):
    """Convenience used by tests and smoke(): parse flags + scene, build the BVH on the host, render on the GPU."""
    cfg, _ = parse_args(flags)
    ls = LoadedScene(text, cfg, textures_dir)
    r = Renderer(device, workspace_bytes)
    try:
        r.upload(ls, with_bvh=bool(cfg.bvh))
        img, st = r.render(cfg)
    finally:
        r.close(); ls.close()
    return img, st


def tile_owner_slots(width: int, height: int, world: int):
    """For tests: (owner rank, index inside that rank's block) per tile, matching rt_set_shard's rule."""
    tx_n, ty_n = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    counts = [0] * world
    out = {}
    for ty in range(ty_n):
        for tx in range(tx_n):
            o = (tx + ty) % world
            out[(tx, ty)] = (o, counts[o]); counts[o] += 1
    return out, max(counts), tx_n, ty_n


def assemble_tiles_numpy(blocks: list[np.ndarray], width: int, height: int) -> np.ndarray:
    """Host-side equivalent of rt_assemble_tiles for tests: blocks[r] = rank r's flat tile-major bytes."""
    world = len(blocks)
    slots, max_tiles, tx_n, ty_n = tile_owner_slots(width, height, world)
    img = np.zeros((height, width, 3), dtype=np.uint8)
    for (tx, ty), (o, k) in slots.items():
        t = blocks[o][k * TILE * TILE * 3:(k + 1) * TILE * TILE * 3].reshape(TILE, TILE, 3)
        y0, x0 = ty * TILE, tx * TILE
        hh, ww = min(TILE, height - y0), min(TILE, width - x0)
        img[y0:y0 + hh, x0:x0 + ww] = t[:hh, :ww]
    return img
