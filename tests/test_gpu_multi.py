"""rt_multi_*: several devices driven by one process through the C ABI.  A one-GPU box can run the RCCL path only with a
one-device list (the library loads librccl, forms the communicator, and the group has no peers) and the sharded path
only with a repeated device id (RCCL cannot span duplicates: the blocks then move by device copies); the 8-GPU run
with real ncclSend / ncclRecv is the driver's."""
import copy
import subprocess

import numpy as np
import pytest
import torch

import cases
from raytracer_amd import api

pytestmark = pytest.mark.gpu


def single(ls, cfg):
    r = api.Renderer(0); r.upload(ls, with_bvh=bool(cfg.bvh))
    img, _ = r.render(cfg); r.close()
    return img


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], [0] * 8])
def test_frame_split_over_a_device_list_is_the_single_device_frame(devices, textures_dir):
    flags = ["-t", "-rd", "2", "-ss", "-sss", "4", "-aa", "-aas", "2"]
    cfg, _ = api.parse_args(flags); cfg.seed = 21
    ls = api.LoadedScene(cases.with_resolution(cases.SCENES["glass_tex"](), 333, 190), cfg, textures_dir)   # not a multiple of the tile size
    want = single(ls, cfg)
    m = api.MultiRenderer(devices)
    assert m.uses_rccl == (len(devices) == 1)
    m.upload(ls)
    for k in range(3):                                   # frames in a row re-use the blocks and the gather buffer
        c = copy.copy(cfg); c.seed = 21 + (k % 2)
        got = m.render(c)
        assert np.array_equal(got, want if c.seed == 21 else single(ls, c)), f"frame {k} on {len(devices)} shards differs"
    assert want.max() > 0
    # device-resident result
    m.render_device(cfg); m.synchronize()
    n = want.size
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    import ctypes as C
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(dev.data_ptr()), C.c_void_p(m.image_ptr()), C.c_size_t(n), C.c_int(3))
    assert np.array_equal(dev.cpu().numpy().reshape(want.shape), want)
    m.close(); ls.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0]])
def test_frames_enqueued_back_to_back_do_not_overwrite_blocks_still_being_gathered(devices, textures_dir):
    """rt_multi_render_device only enqueues.  Three frames with different seeds before one synchronisation: with the
    device-copy transport a block is read by a copy on the first device's stream, so the next frame of its own device
    has to wait for that copy; the image left behind must be the last frame's."""
    cfg, _ = api.parse_args(["-r", "-ss", "-sss", "4", "-aa", "-aas", "2"])
    ls = api.LoadedScene(cases.reference_scene(1920, 1080), cfg, textures_dir)
    m = api.MultiRenderer(devices); m.upload(ls)
    last = None
    for rep in range(3):
        for seed in (5, 6, 7):
            c = copy.copy(cfg); c.seed = seed + 10 * rep
            m.render_device(c); last = c
    m.synchronize()
    n = 1080 * 1920 * 3
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    import ctypes as C
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(dev.data_ptr()), C.c_void_p(m.image_ptr()), C.c_size_t(n), C.c_int(3))
    assert np.array_equal(dev.cpu().numpy().reshape(1080, 1920, 3), single(ls, last))
    m.close(); ls.close()


@pytest.mark.parametrize("n,flags", [(0, ["-t", "-rd", "2", "-ss", "-sss", "4"]), (30000, ["-r", "-rd", "2", "-grs", "1"]), (30000, ["-u"])])
def test_adopted_device_scene_renders_the_source_frames(n, flags, textures_dir):
    """The scene broadcast of the multi-GPU path, as far as one GPU goes: a second context adopts the layout of an
    uploaded scene, the device arrays are copied into it, and it renders the source's frames bit for bit - sharded too -
    without any host scene data.  Small scene (no second-level hierarchies) and 30 000 primitives (hierarchies of the
    leaves and of the file-order set, records in hierarchy order)."""
    from raytracer_amd import dist as rdist, scenegen as sg
    cfg, _ = api.parse_args(flags); cfg.seed = 5
    ls = api.ArrayScene(sg.random_prims(n, 3, transparent_frac=0.1), (640, 360)) if n else api.LoadedScene(cases.SCENES["glass_tex"](), cfg, textures_dir)
    src = api.Renderer(0); src.upload(ls, with_bvh=bool(cfg.bvh))
    want, st = src.render(cfg)
    assert want.max() > 0
    # (1) the library's own device copies
    a = api.Renderer(0); a.adopt_from(src)
    got, st_a = a.render(cfg)
    assert np.array_equal(got, want) and st_a.total_rays() == st.total_rays()
    # (2) layout blob + every array moved by the caller, here with torch copies between the two contexts' arrays
    b = api.Renderer(0)
    layout = src.export_layout()
    b.adopt_layout(layout)
    assert (b.width, b.height) == (src.width, src.height)
    pairs = list(zip(src.scene_arrays(), b.scene_arrays()))
    assert len(pairs) >= 5
    for (sp, sn), (dp, dn) in pairs:
        assert dn >= sn and sp and dp
        torch.as_tensor(rdist._DeviceBytes(dp, sn), device="cuda").copy_(torch.as_tensor(rdist._DeviceBytes(sp, sn), device="cuda"))
    torch.cuda.synchronize()
    got_b, _ = b.render(cfg)
    assert np.array_equal(got_b, want)
    # sharded frames of the adopted scene assemble to the same image
    blocks = []
    for rank in range(3):
        b.set_shard(rank, 3)
        blk, _ = b.render(cfg); blocks.append(blk.copy())
    assert np.array_equal(api.assemble_tiles_numpy(blocks, src.width, src.height), want)
    # (3) contexts of one GPU that render side by side use the source's arrays in place (frames in flight): same frames,
    # from two contexts at once on their own streams, and no device memory beyond the work buffers
    free0 = torch.cuda.mem_get_info()[0]
    s1 = api.Renderer(0); s1.share_from(src)
    s2 = api.Renderer(0); s2.share_from(src)
    assert sorted(src.scene_arrays()) == sorted(s1.scene_arrays())
    o1 = torch.empty(s1.out_bytes(), dtype=torch.uint8, device="cuda"); o2 = torch.empty_like(o1)
    for k in range(3):
        s1.render_device(cfg, o1.data_ptr()); s2.render_device(cfg, o2.data_ptr())
    s1.synchronize(); s2.synchronize()
    assert np.array_equal(o1.cpu().numpy().reshape(want.shape), want) and np.array_equal(o2.cpu().numpy().reshape(want.shape), want)
    scene_bytes = sum(n for _, n in src.scene_arrays())
    if scene_bytes > (64 << 20):
        assert free0 - torch.cuda.mem_get_info()[0] < scene_bytes          # two more contexts, less than one more copy of the scene
    s3 = api.Renderer(0); s3.share_from(a)                                 # ... also from a context that itself adopted the scene (a rank's first context)
    assert np.array_equal(s3.render(cfg)[0], want)
    s3.close()
    s1.upload(ls, with_bvh=bool(cfg.bvh))                                  # an upload ends the sharing
    assert np.array_equal(s1.render(cfg)[0], want)
    s1.close(); s2.close()
    # a tree cannot be uploaded into an adopted scene; a fresh upload replaces it
    with pytest.raises(api.RtError) as e:
        hb = ls.bvh()
        a._check(api.lib().rt_upload_bvh(a._ctx, hb.nodes, hb.n_nodes, hb.prim_ids, hb.n_prim_ids))
    assert e.value.code == -3
    a.upload(ls, with_bvh=bool(cfg.bvh))
    assert np.array_equal(a.render(cfg)[0], want)
    with pytest.raises(api.RtError):
        b.adopt_layout(layout[:100])
    src.close(); a.close(); b.close(); ls.close()


def test_device_broadcast_and_per_device_uploads_give_the_same_frames(textures_dir):
    """rt_multi_upload: one upload + build and a broadcast of the device arrays (default) against every device uploading
    and building for itself (option device_broadcast = 0)."""
    from raytracer_amd import scenegen as sg
    cfg, _ = api.parse_args(["-r", "-rd", "2", "-grs", "1"])
    ls = api.ArrayScene(sg.random_prims(20000, 8, transparent_frac=0.1), (480, 270))
    want = single(ls, cfg)
    for bc in (1, 0):
        m = api.MultiRenderer([0, 0, 0]); m.set_option("device_broadcast", bc); m.upload(ls)
        assert np.array_equal(m.render(cfg), want), f"device_broadcast {bc}"
        m.close()
    ls.close()


def test_multi_errors_and_no_bvh():
    with pytest.raises(api.RtError):
        api.MultiRenderer([0, 99])                      # no such device
    cfg, _ = api.parse_args(["-u"])
    ls = api.LoadedScene(cases.SCENES["ref"](), cfg)
    m = api.MultiRenderer([0, 0])
    with pytest.raises(api.RtError) as e:
        m.render(cfg)
    assert e.value.code == -3
    m.upload(ls, with_bvh=False)
    assert np.array_equal(m.render(cfg), single(ls, cfg))
    m.close(); ls.close()


def test_command_line_front_end_with_several_gpus(tmp_path, textures_dir):
    """raytracer --gpus 2 --gpu 0 would need two devices; the executable's multi-device branch is exercised through
    the library above.  Here: the single-device executable still writes the reference's P3 for the reference's scene."""
    exe = cases.ROOT / "raytracer_amd" / "raytracer"
    if not exe.exists():
        pytest.skip("CLI not built")
    scene = tmp_path / "scene.json"; scene.write_text(cases.SCENES["ref"]())
    p = subprocess.run([str(exe), "--scene", str(scene), "--textures", textures_dir, "--outdir", str(tmp_path), "-o", "o.ppm"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    import ref_runner
    img = ref_runner.parse_p3(tmp_path / "o.ppm")
    gold = np.load(cases.GOLDEN / "deterministic.npz")[cases.case_id("ref", [])].astype(np.int32)
    assert np.array_equal(img, gold)


def test_bench_frames_through_an_rccl_group_of_one_rank(textures_dir):
    """The N > 1 frame path of bench.py (pipelined gather on the communicator's stream, re-ordering, barriers) with the
    only RCCL group a one-GPU box can form: one rank.  The last timed frame must be the frame a plain context renders."""
    import json
    import sys
    import zlib
    p = subprocess.run([sys.executable, str(cases.ROOT / "bench.py"), "--rehearse-rccl", "--steps", "7", "--warmup", "2",
                        "--no-cpu-baseline", "--no-scaling-4k"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["frames_in_flight"] == 3
    import bench
    wl = bench.WORKLOADS["config2"]
    cfg, _ = api.parse_args(wl["flags"])
    ls = api.LoadedScene(cases.reference_scene(*wl["res"]), cfg, textures_dir)
    want = single(ls, cfg); ls.close()
    assert line["config"]["image_crc32"] == zlib.crc32(np.ascontiguousarray(want).tobytes())
