"""Development aid: of the (hit, light) pairs of scene.json that the cone rules of k_shadow send to tracing, how many end with
every sample blocked, some, or none - the pairs a sharper rule could still decide without tracing.  Needs a library
built with -DRT_SHADOW_HIST (make -C raytracer_amd/csrc B=build_hist OUT=../librt_amd_hist.so KDEFS=-DRT_SHADOW_HIST).
usage: RT_AMD_LIB=raytracer_amd/librt_amd_hist.so python tools/shadow_pair_hist.py "-r -rd 4 -ss -sss 16" """
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch, cases
from raytracer_amd import api
flags = sys.argv[1].split()
cfg, _ = api.parse_args(flags)
ls = api.LoadedScene(cases.reference_scene(1920, 1080), cfg, str(cases.TEXTURES))
r = api.Renderer(0, 16 << 30)
r.set_option("walk_stats", 1); r.set_option("overlap", 0); r.set_option("shadow_queue", 0)
r.upload(ls, with_bvh=True)
out = torch.empty(r.out_bytes(), dtype=torch.uint8, device="cuda")
st = r.render_device(cfg, out.data_ptr(), want_stats=True)
fn = api.lib().rt_debug_walk_stats
fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]; fn.restype = C.c_int
buf = (C.c_uint64 * 32)()
assert fn(r._ctx, buf) == 0
v = list(buf)
print(flags, "shadow rays", st.shadow_rays, "hits shaded", st.nodes_shaded)
print("level 0 traced pairs: blocked", v[16], "partly", v[17], "unblocked", v[18])
print("deeper  traced pairs: blocked", v[20], "partly", v[21], "unblocked", v[22])
print("may mask (1 plane, 2 cube, 4 sphere) of traced pairs that end all unblocked:", v[0:8])
print("                                          ...       that end all blocked:  ", v[8:16])
print("                                          ...       that end partly:       ", v[24:32])
