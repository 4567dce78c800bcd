"""ctypes binding of oracle/librt_oracle.so (the CPU restatement). TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "librt_oracle.so"


class OrcConfig(C.Structure):
    _fields_ = [("softShadows", C.c_int), ("SSsamples", C.c_int), ("glossyReflect", C.c_int),
                ("GRsamples", C.c_int), ("lightRadius", C.c_float), ("antiAliasing", C.c_int),
                ("AAsamples", C.c_int), ("textures", C.c_int), ("bvh", C.c_int),
                ("reflectRefractDepth", C.c_int), ("dof", C.c_int), ("dofSamples", C.c_int),
                ("motionBlur", C.c_int), ("mbSamples", C.c_int)]


class OrcStats(C.Structure):
    _fields_ = [("primary", C.c_longlong), ("shadow", C.c_longlong), ("reflection", C.c_longlong),
                ("refraction", C.c_longlong), ("box_tests", C.c_longlong), ("prim_tests", C.c_longlong),
                ("seconds", C.c_double)]

    def total_rays(self) -> int:
        return self.primary + self.shadow + self.reflection + self.refraction


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise FileNotFoundError(f"{LIB_PATH} not built (make -C oracle port)")
        L = C.CDLL(str(LIB_PATH))
        L.orc_default_config.argtypes = [C.POINTER(OrcConfig)]
        L.orc_parse_args.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(OrcConfig), C.c_char_p, C.c_int]
        L.orc_render.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(OrcConfig), C.c_int, C.POINTER(C.c_int),
                                 C.c_longlong, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(OrcStats)]
        L.orc_camera_ray.argtypes = [C.c_char_p, C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.orc_bvh_dump.argtypes = [C.c_char_p, C.POINTER(OrcConfig), C.POINTER(C.c_float), C.c_int]
        L.orc_last_error.restype = C.c_char_p
        L.orc_set_window.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_set_window.restype = None
        L.orc_set_stride.argtypes = [C.c_int, C.c_int]
        L.orc_set_stride.restype = None
        _lib = L
    return _lib


def parse_args(flags: list[str]) -> OrcConfig:
    argv = [b"raytracer"] + [f.encode() for f in flags]
    arr = (C.c_char_p * len(argv))(*argv)
    cfg = OrcConfig()
    if lib().orc_parse_args(len(argv), arr, C.byref(cfg), None, 0) != 0:
        raise ValueError(f"the reference aborts on these flags (std::stoi throws): {flags}")
    return cfg


def render(scene_text, flags: list[str], textures_dir: str | None = None, rand_seed: int = 1, window=None, stride=(1, 1), res=None):
    """Returns (image int32 [H,W,3], OrcStats). rand_seed=1 is the state of a process that never called srand.
    scene_text: str, or - for scenes of gigabytes - the encoded bytes together with res = (width, height).
    window = (x0, y0, x1, y1): render only those pixels (exact for deterministic flag sets), the rest stays 0;
    stride = (sx, sy): of those, only every sx-th column and sy-th row counted from the window's corner."""
    from ref_runner import get_resolution  # same directory
    w, h = res if res is not None else get_resolution(scene_text)
    cfg = parse_args(flags)
    out = np.zeros(w * h * 3, dtype=np.int32)
    W, H = C.c_int(), C.c_int()
    st = OrcStats()
    lib().orc_set_stride(int(stride[0]), int(stride[1]))
    if window is not None:
        lib().orc_set_window(*[int(v) for v in window])
    else:
        lib().orc_set_window(0, 0, 1 << 30, 1 << 30)
    rc = lib().orc_render(scene_text if isinstance(scene_text, bytes) else scene_text.encode(), textures_dir.encode() if textures_dir else None, C.byref(cfg),
                          rand_seed, out.ctypes.data_as(C.POINTER(C.c_int)), out.size, C.byref(W), C.byref(H),
                          C.byref(st))
    if rc != 0:
        raise RuntimeError(f"orc_render rc={rc}: {lib().orc_last_error().decode()}")
    return out.reshape(H.value, W.value, 3), st


def camera_ray(scene_text: str, x: float, y: float) -> np.ndarray:
    out = (C.c_float * 6)()
    rc = lib().orc_camera_ray(scene_text.encode(), x, y, out)
    if rc != 0:
        raise RuntimeError(lib().orc_last_error().decode())
    return np.array(list(out), dtype=np.float32)


def kat(kind: int, inputs: np.ndarray) -> np.ndarray:
    """One known-answer record (layout of oracle/ref_kat.cpp) through the restatement."""
    a = np.ascontiguousarray(inputs, dtype=np.float32)
    out = np.zeros(19, dtype=np.float32)
    L = lib()
    L.orc_kat.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]; L.orc_kat.restype = C.c_int
    n = L.orc_kat(int(kind), a.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
    if n < 0:
        raise RuntimeError("orc_kat: unknown kind")
    return out[:n]


def bvh_dump(scene_text: str, flags: list[str] | None = None, cap: int = 4096) -> np.ndarray:
    cfg = parse_args(flags or [])
    out = np.zeros(cap * 12, dtype=np.float32)
    n = lib().orc_bvh_dump(scene_text.encode(), C.byref(cfg), out.ctypes.data_as(C.POINTER(C.c_float)), cap)
    if n < 0:
        raise RuntimeError(lib().orc_last_error().decode())
    return out.reshape(cap, 12)[:n].copy()
