"""rt_render_device_noalloc (work buffers from the library's per-GPU scratch array, no allocation) against
rt_render_device, through the C ABI.  Bit-exact, two calls in a row, the second on a smaller input."""
import ctypes as C

import numpy as np
import pytest
import torch

import cases
from raytracer_amd import api

pytestmark = pytest.mark.gpu


def _both(r, cfg):
    n = r.out_bytes()
    want = torch.zeros(n, dtype=torch.uint8, device="cuda")
    got = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
    r.render_device(cfg, want.data_ptr()); r.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    r.render_device_noalloc(cfg, got.data_ptr()); r.synchronize()
    free_after = torch.cuda.mem_get_info()[0]
    assert free_after == free_before, "device memory in use changed across the call"
    return got.cpu().numpy(), want.cpu().numpy()


def test_noalloc_entry_matches_render_device_twice_second_call_smaller(textures_dir):
    assert api.lib().rt_noalloc_capacity() == 64 << 20
    # first call: full-HD film (33 MB of pixel sums + batches in what is left), recursion, soft shadows, textures
    flags = ["-rd", "2", "-ss", "-sss", "4", "-t"]
    cfg, _ = api.parse_args(flags); cfg.seed = 9
    r = api.Renderer(0)
    ls = api.LoadedScene(cases.reference_scene(1920, 1080), cfg, textures_dir)
    r.upload(ls)
    got, want = _both(r, cfg)
    assert want.max() > 0 and np.array_equal(got, want)
    ls.close()
    # second call, smaller input: a 200x112 film of another scene, several samples per pixel (accumulate + finalize)
    flags2 = ["-rd", "3", "-grs", "1", "-aa", "-aas", "3", "-u"]
    cfg2, _ = api.parse_args(flags2); cfg2.seed = 4
    ls2 = api.LoadedScene(cases.SCENES["glass"](), cfg2, textures_dir)
    r.upload(ls2, with_bvh=False)
    got2, want2 = _both(r, cfg2)
    assert want2.max() > 0 and np.array_equal(got2, want2)
    # and once more on the first configuration's flag set at the small size: stale contents of the array must not matter
    cfg3, _ = api.parse_args(["-rd", "2", "-grs", "1"])
    r.upload(ls2)
    got3, want3 = _both(r, cfg3)
    assert np.array_equal(got3, want3)
    ls2.close(); r.close()


def test_noalloc_entry_rejects_what_does_not_fit():
    cfg, _ = api.parse_args([])
    r = api.Renderer(0)
    ls = api.LoadedScene(cases.reference_scene(3840, 2160), cfg)        # 8.3 M pixel slots: 133 MB of pixel sums
    r.upload(ls)
    out = torch.zeros(r.out_bytes(), dtype=torch.uint8, device="cuda")
    with pytest.raises(api.RtError) as e:
        r.render_device_noalloc(cfg, out.data_ptr())
    assert e.value.code == -4 and "scratch" in str(e.value)
    r.render_device(cfg, out.data_ptr()); r.synchronize()               # the allocating entry point still works
    assert int(out.max()) > 0
    ls.close(); r.close()


def test_noalloc_entry_refuses_a_second_context_while_a_frame_may_be_in_flight(textures_dir):
    """One scratch array per GPU: the context that rendered into it owns it until its rt_synchronize; another context
    asking meanwhile gets RT_ERR_STATE instead of overwriting the frame, and the array is free again afterwards."""
    cfg, _ = api.parse_args(["-r", "-ss", "-sss", "4"]); cfg.seed = 3
    ls = api.LoadedScene(cases.reference_scene(640, 360), cfg, textures_dir)
    a, b = api.Renderer(0), api.Renderer(0)
    a.upload(ls); b.upload(ls)
    n = a.out_bytes()
    out_a = torch.zeros(n, dtype=torch.uint8, device="cuda"); out_b = torch.zeros(n, dtype=torch.uint8, device="cuda")
    want = torch.zeros(n, dtype=torch.uint8, device="cuda")
    a.render_device(cfg, want.data_ptr()); a.synchronize()
    a.render_device_noalloc(cfg, out_a.data_ptr())
    a.render_device_noalloc(cfg, out_a.data_ptr())                     # the same context again: ordered by its stream
    with pytest.raises(api.RtError) as e:
        b.render_device_noalloc(cfg, out_b.data_ptr())
    assert e.value.code == -3 and "another context" in str(e.value)
    a.synchronize()
    b.render_device_noalloc(cfg, out_b.data_ptr()); b.synchronize()     # free again
    assert np.array_equal(out_a.cpu().numpy(), want.cpu().numpy()) and np.array_equal(out_b.cpu().numpy(), want.cpu().numpy())
    b.render_device_noalloc(cfg, out_b.data_ptr())                      # b owns the array again and is destroyed with a frame enqueued
    b.close()                                                         # destroying the owner releases the array too
    a.render_device_noalloc(cfg, out_a.data_ptr()); a.synchronize()
    ls.close(); a.close()
