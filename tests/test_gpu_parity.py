"""Parity of the HIP render path, called through the C ABI (raytracer_amd/librt_amd.so), with the CPU
restatement (the checker) and with golden images rendered by the compiled reference.

Bar: deterministic flag sets within +-1 LSB per 8-bit channel (in practice bit-exact; the count of differing
pixels is asserted too); stochastic flag sets >= 40 dB PSNR between two high-sample renders."""
import numpy as np
import pytest
import torch

import cases
import oracle
from raytracer_amd import api, scenegen as sg

pytestmark = pytest.mark.gpu

DET = np.load(cases.GOLDEN / "deterministic.npz")
STO = np.load(cases.GOLDEN / "stochastic.npz")
TOL_LSB = 1                      # per 8-bit channel, deterministic renders
MAX_DIFFERING_FRACTION = 1e-3    # and at most 0.1 % of the pixels may differ at all
PSNR_DB = 40.0


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


def assert_close_images(img, ref, what):
    d = np.abs(img.astype(np.int32) - ref.astype(np.int32))
    differing = (d.max(axis=2) > 0).mean()
    assert d.max() <= TOL_LSB, f"{what}: max |diff| = {d.max()} LSB, {int((d.max(axis=2) > TOL_LSB).sum())} pixels beyond tolerance"
    assert differing <= MAX_DIFFERING_FRACTION, f"{what}: {differing:.2%} of the pixels differ"


def test_native_library_is_what_runs():
    assert torch.cuda.is_available()
    api.lib()
    assert any("librt_amd.so" in line for line in open("/proc/self/maps"))


@pytest.mark.parametrize("scene,flags", cases.DETERMINISTIC, ids=[cases.case_id(s, f) for s, f in cases.DETERMINISTIC])
def test_deterministic_matches_oracle_and_reference_golden(scene, flags, textures_dir):
    text = cases.SCENES[scene]()
    img, st = api.render_scene_text(text, flags, textures_dir)
    ref, ost = oracle.render(text, flags, textures_dir)
    assert_close_images(img, ref, "vs oracle")
    assert_close_images(img, DET[cases.case_id(scene, flags)], "vs reference golden")
    # ray accounting: the reference-equivalent counts (identical reflection rays and their subtrees counted
    # gr_samples times, as the reference traces them) equal what the restatement counts while it renders
    assert st.primary_rays == ost.primary
    assert (st.shadow_rays_ref, st.reflection_rays_ref, st.refraction_rays_ref) == (ost.shadow, ost.reflection, ost.refraction)
    if "-r" not in flags and "-rd" not in flags:
        # (shadow rays towards a light the surface faces away from are not traced: their factor multiplies N.L = 0)
        assert st.shadow_rays <= ost.shadow and st.reflection_rays == 0 and st.refraction_rays == 0


def test_config1_full_frame_matches_reference_golden():
    g = np.load(cases.GOLDEN / "config1_1080p.npz")["image"]
    img, st = api.render_scene_text(cases.reference_scene(1920, 1080), [])
    assert_close_images(img, g, "config 1 at 1920x1080")
    assert st.primary_rays == 1920 * 1080


@pytest.mark.parametrize("scene,flags", cases.STOCHASTIC, ids=[cases.case_id(s, f) for s, f in cases.STOCHASTIC])
def test_stochastic_psnr_vs_reference_high_sample_render(scene, flags, textures_dir):
    img, _ = api.render_scene_text(cases.SCENES[scene](), flags, textures_dir)
    gold = STO[cases.case_id(scene, flags)]
    p = psnr(img, gold)
    assert p >= PSNR_DB, f"PSNR {p:.1f} dB"


def test_stochastic_is_reproducible_and_seeded():
    text = cases.SCENES["ref"]()
    flags = ["-aa", "-aas", "4", "-ss", "-sss", "4", "-r", "-gr", "-grs", "2"]
    a, _ = api.render_scene_text(text, flags)
    b, _ = api.render_scene_text(text, flags)
    c, _ = api.render_scene_text(text, flags + ["--seed", "7"])
    assert np.array_equal(a, b)
    assert not np.array_equal(a, c)


def test_identical_reflection_rays_are_traced_once_but_summed_gr_times():
    """Without -gr the reference traces gr_samples identical reflection rays; tracing one and replaying the
    float sum must give the same image as the oracle's literal loop for any gr_samples."""
    text = cases.SCENES["glass"]()
    for grs in ("1", "3", "16"):
        img, st = api.render_scene_text(text, ["-rd", "2", "-grs", grs])
        ref, _ = oracle.render(text, ["-rd", "2", "-grs", grs])
        assert_close_images(img, ref, f"-grs {grs}")
        assert st.reflection_rays <= st.reflection_rays_ref


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_render_assembles_to_the_unsharded_image(world, textures_dir):
    """Full-size property: rendering interleaved tile shards (one context per rank, here all on GPU 0), gathering the
    blocks and assembling them on the device gives exactly the single-context image, stochastic effects included."""
    text = cases.reference_scene(1920, 1080)
    flags = ["-r", "-ss", "-sss", "4", "-aa", "-aas", "2"]
    cfg, _ = api.parse_args(flags)
    ls = api.LoadedScene(text, cfg)
    whole = api.Renderer(0); whole.upload(ls)
    full, st_full = whole.render(cfg)
    blocks, rays = [], 0
    for rank in range(world):
        r = api.Renderer(0); r.upload(ls); r.set_shard(rank, world)
        out = torch.empty(r.out_bytes(), dtype=torch.uint8, device="cuda")
        st = r.render_device(cfg, out.data_ptr(), want_stats=True)
        rays += st.total_rays()
        blocks.append(out)
        r.close()
    assert rays == st_full.total_rays()
    gathered = torch.cat(blocks)
    root = api.Renderer(0); root.upload(ls); root.set_shard(0, world)
    img = torch.empty(1080 * 1920 * 3, dtype=torch.uint8, device="cuda")
    root.assemble(gathered.data_ptr(), world, img.data_ptr()); root.synchronize()
    got = img.cpu().numpy().reshape(1080, 1920, 3)
    assert np.array_equal(got, full)
    assert np.array_equal(api.assemble_tiles_numpy([b.cpu().numpy() for b in blocks], 1920, 1080), full)
    root.close(); whole.close(); ls.close()


def test_small_workspace_renders_in_chunks_with_the_same_result():
    text = cases.reference_scene(640, 360)
    flags = ["-rd", "3", "-grs", "1", "-ss", "-sss", "4"]
    big, _ = api.render_scene_text(text, flags)
    small, st = api.render_scene_text(text, flags, workspace_bytes=8 << 20)
    assert np.array_equal(big, small)


def test_non_power_of_two_shadow_samples_and_many_lights():
    cams = [sg.camera(res=(160, 90))]
    lights = [sg.light((4, 1, 6), 500.0), sg.light((-4, -2, 5), 300.0), sg.light((0, 3, 8), 200.0)]
    m = sg.material((0.7, 0.6, 0.5), (0.2, 0.2, 0.2), 40.0)
    text = sg.scene_text(cams, lights, [sg.sphere((0, 1, 1), 1.0, m)], [sg.cube((2.5, 2, 0.7), 0.7, m, (0.1, 0.2, 0.3))], [sg.ground(20.0, 0.0, m)])
    img, st = api.render_scene_text(text, [])
    ref, ost = oracle.render(text, [])
    assert_close_images(img, ref, "3 lights, hard shadows")
    assert st.shadow_rays_ref == ost.shadow and st.shadow_rays <= ost.shadow
    a, sa = api.render_scene_text(text, ["-ss", "-sss", "48"])       # 48 is not a power of two: atomic accumulation path
    b, sb = api.render_scene_text(text, ["-ss", "-sss", "64"])
    assert sa.shadow_rays_ref == sa.nodes_shaded * 3 * 48 and sa.shadow_rays % 48 == 0 and 0 < sa.shadow_rays <= sa.shadow_rays_ref
    assert psnr(a, b) > 35.0


def test_empty_and_lightless_scenes():
    img, st = api.render_scene_text(sg.scene_text([sg.camera(res=(70, 40))], [], [], [], []), [])
    assert img.shape == (40, 70, 3) and img.max() == 0 and st.primary_rays == 70 * 40
    m = sg.material((0.8, 0.4, 0.2))
    text = sg.scene_text([sg.camera(res=(70, 40))], [], [sg.sphere((0, 0, 1.5), 1.0, m)], [], [])
    for flags in ([], ["-u"]):
        img, _ = api.render_scene_text(text, flags)
        ref, _ = oracle.render(text, flags)
        assert np.array_equal(img.astype(np.int32), ref) and img.max() > 0      # ambient term only


def test_error_paths():
    cfg, _ = api.parse_args([])
    r = api.Renderer(0)
    with pytest.raises(api.RtError) as e:
        r.render(cfg)                                   # nothing uploaded
    assert e.value.code == -3
    ls = api.LoadedScene(cases.SCENES["ref"](), cfg)
    r.upload(ls, with_bvh=False)
    with pytest.raises(api.RtError) as e:
        r.render(cfg)                                   # config wants the BVH, none uploaded
    assert e.value.code == -3
    cfg_u, _ = api.parse_args(["-u"])
    img, _ = r.render(cfg_u)
    assert img.max() > 0
    bad, _ = api.parse_args(["-aa", "-aas", "0"])
    with pytest.raises(api.RtError):
        r.render(bad)
    with pytest.raises(api.RtError):
        r.set_shard(3, 2)
    # two planes may not share vertices: the library re-orders and re-lays-out each plane's vertex range in place
    import ctypes as C
    sc = api.Scene(); C.memmove(C.byref(sc), C.byref(ls.scene), C.sizeof(api.Scene))
    planes = (api.Plane * 2)()
    for k in range(2):
        C.memmove(C.byref(planes[k]), C.byref(ls.scene.planes[0]), C.sizeof(api.Plane))
    sc.planes = planes; sc.n_planes = 2
    with pytest.raises(api.RtError) as e:
        r._check(api.lib().rt_upload_scene(r._ctx, C.byref(sc)))
    assert e.value.code == -1 and "overlaps" in str(e.value)
    with pytest.raises(api.RtError):
        r.render(cfg_u)                                 # the failed upload left no scene behind
    r.close(); ls.close()


@pytest.mark.parametrize("scene,flags", [("ref", ["-rd", "3", "-grs", "1", "-ss", "-sss", "4", "-t"]), ("glass_tex", ["-rd", "2", "-grs", "1", "-t", "-u"]),
                                         ("glass", ["-m", "-mbs", "1", "-rd", "2", "-grs", "1"])])
def test_lds_resident_primitive_records_change_nothing(scene, flags, textures_dir):
    """Tiny scenes keep their primitive records and materials in LDS (shadow kernel by default, the extend kernels as an
    option): identical frames and ray counts with the records in global memory."""
    cfg, _ = api.parse_args(flags)
    cfg.seed = 5
    ls = api.LoadedScene(cases.SCENES[scene](), cfg, textures_dir)
    got = []
    for mode in (0, 1, 2):
        r = api.Renderer(0); r.set_option("lds_prims", mode); r.upload(ls, with_bvh=bool(cfg.bvh))
        img, st = r.render(cfg)
        got.append((img, (st.shadow_rays, st.reflection_rays, st.refraction_rays, st.nodes_shaded)))
        r.close()
    ls.close()
    assert got[0][0].max() > 0
    for img, counts in got[1:]:
        assert np.array_equal(img, got[0][0]) and counts == got[0][1]


@pytest.mark.parametrize("scene,n,flags", [("ref", 0, ["-r", "-rd", "3", "-ss", "-sss", "16"]), ("ref", 0, ["-ss", "-sss", "5", "-u"]),
                                           ("dof", 0, ["-ss", "-sss", "8", "-rd", "2", "-grs", "1"]),
                                           ("random", 20000, ["-ss", "-sss", "4"]), ("random", 20000, ["-ss", "-sss", "4", "-u"])])
def test_reciprocal_arithmetic_on_shadow_samples_moves_almost_no_sample(scene, n, flags, textures_dir):
    """Soft-shadow sample rays are drawn at random, so their tests may run on the hardware's reciprocals (the default)
    instead of IEEE division: same rays, same thresholds, and a decision can only flip for a ray within rounding of a
    primitive's silhouette.  Against the exact arithmetic: same ray counts, and all but a handful of pixels identical."""
    cfg, _ = api.parse_args(flags)
    cfg.seed = 11
    if scene == "random":
        ls = api.ArrayScene(sg.random_prims(n, 3, transparent_frac=0.1), (640, 360))
    else:
        ls = api.LoadedScene(cases.reference_scene(480, 270) if scene == "ref" else cases.blend_scene(scene, 480, 270), cfg, textures_dir)
    got = []
    for fast in (1, 0):
        r = api.Renderer(0); r.set_option("fast_samples", fast); r.upload(ls, with_bvh=bool(cfg.bvh))
        img, st = r.render(cfg)
        got.append((img.astype(np.int32), (st.shadow_rays, st.reflection_rays, st.refraction_rays, st.nodes_shaded)))
        r.close()
    ls.close()
    assert got[0][0].max() > 0 and got[0][1] == got[1][1]
    diff = np.abs(got[0][0] - got[1][0]).max(axis=-1)
    assert (diff > 0).sum() <= max(3, 1e-3 * diff.size), f"{(diff > 0).sum()} pixels differ"
    # a flipped sample moves a pixel by up to 255 / ss_samples levels; grazing rays that start 0.001 along the light
    # direction from a surface (raytracer.cpp:549) are the rounding-sensitive ones, in the reference as well
    assert psnr(got[0][0], got[1][0]) > 50.0


@pytest.mark.parametrize("scene,flags", [("ref", []), ("ref", ["-r", "-rd", "3", "-ss", "-sss", "8"]), ("ref", ["-ss", "-sss", "5", "-u"]),
                                         ("glass_tex", ["-rd", "2", "-grs", "1", "-t"]), ("bvh5", []), ("cube_mb", ["-m", "-mbs", "2", "-aa", "-aas", "2"])])
def test_untraced_shadow_rays_of_unlit_sides_change_no_pixel(scene, flags, textures_dir):
    """raytracer.cpp:322-336 multiplies the shadow factor by max(0, N.L) only, so the shadow rays of a (hit, light) pair
    whose surface clearly faces away from the light are not traced (option skip_unlit, default on).  Against tracing
    them all: bit-identical frames (the generator is counter based, nothing shifts), the reference-equivalent counts
    unchanged, fewer rays traced; with the option off the traced count is the oracle's."""
    cfg, _ = api.parse_args(flags)
    cfg.seed = 3
    text = cases.SCENES[scene]()
    ls = api.LoadedScene(text, cfg, textures_dir)
    got = []
    for skip in (1, 0):
        r = api.Renderer(0); r.set_option("skip_unlit", skip); r.upload(ls, with_bvh=bool(cfg.bvh))
        img, st = r.render(cfg)
        got.append((img, st))
        r.close()
    ls.close()
    assert got[0][0].max() > 0 and np.array_equal(got[0][0], got[1][0])
    on, off = got[0][1], got[1][1]
    assert (on.shadow_rays_ref, on.reflection_rays, on.refraction_rays, on.nodes_shaded) == (off.shadow_rays_ref, off.reflection_rays, off.refraction_rays, off.nodes_shaded)
    assert 0 < on.shadow_rays < off.shadow_rays
    if not any(x in flags for x in ("-ss", "-aa", "-m", "-dof", "-gr")):
        _, ost = oracle.render(text, flags, textures_dir)
        assert off.shadow_rays_ref == ost.shadow
        if "-r" not in flags and "-rd" not in flags:
            assert off.shadow_rays == ost.shadow


@pytest.mark.parametrize("scene,flags", [("ref", []), ("ref", ["-u"]), ("glass_tex", ["-t"]), ("glass_tex", ["-t", "-u"]), ("bvh5", []), ("rand300m", []),
                                         ("ref", ["-aa", "-aas", "2", "-dof", "-dofs", "2", "-m", "-mbs", "2"]), ("glass", ["-m", "-mbs", "3", "-gr", "-grs", "4"])])
def test_one_kernel_frames_equal_the_three_stage_pipeline(scene, flags, textures_dir):
    """Frames without recursion and with hard shadows are rendered by one kernel (generate, trace, shadow rays, shade).
    Its images, ray counts and shaded-hit counts must equal those of the extend / shadow / shade pipeline with the
    option off - unsharded, as one of three shards, and in several batches."""
    cfg, _ = api.parse_args(flags + ["-rd", "0"]); cfg.seed = 5
    ls = api.LoadedScene(cases.SCENES[scene](), cfg, textures_dir)
    for world, workspace in ((1, 0), (3, 0), (1, 6 << 20)):
        got = []
        for fuse in (1, 0):
            r = api.Renderer(0, workspace); r.set_option("fuse_simple", fuse)
            r.upload(ls, with_bvh=bool(cfg.bvh)); r.set_shard(world - 1, world)
            img, st = r.render(cfg)
            got.append((img, (st.primary_rays, st.shadow_rays, st.shadow_rays_ref, st.nodes_shaded, st.reflection_rays), st.launches_shadow)); r.close()
        assert got[0][0].max() > 0 and np.array_equal(got[0][0], got[1][0]), f"world {world}: {(got[0][0] != got[1][0]).any(axis=-1).sum()} pixels differ"
        assert got[0][1] == got[1][1] and got[0][2] == 0 and got[1][2] > 0
    ls.close()


# ---- large scenes: the second-level hierarchy ----------------------------------------------------------------

@pytest.mark.parametrize("n,res,flags", [(20000, (960, 540), []), (20000, (960, 540), ["-u"]),
                                         (20000, (480, 270), ["-rd", "2", "-grs", "1"]), (200000, (320, 180), [])])
def test_second_level_hierarchy_changes_no_pixel(n, res, flags):
    """Searching a crowded group through its hierarchy must give exactly the image of testing its primitives one
    by one (the reference's leaf loop / brute-force loop), on scenes far larger than the oracle can render whole."""
    table = sg.random_prims(n, 42, transparent_frac=0.1)
    cfg, _ = api.parse_args(flags)
    imgs = []
    for min_group in (8, 1 << 40):
        ls = api.ArrayScene(table, res)
        r = api.Renderer(0)
        r.set_option("second_level_min_group", min_group)
        r.upload(ls, with_bvh=bool(cfg.bvh))
        img, st = r.render(cfg)
        imgs.append(img); r.close(); ls.close()
    assert imgs[0].max() > 0
    assert np.array_equal(imgs[0], imgs[1])


@pytest.mark.parametrize("flags", [[], ["-u"], ["-rd", "2", "-grs", "1"]])
@pytest.mark.parametrize("device_build", [1, 0])
def test_records_in_hierarchy_order_keep_the_reference_tie_rule(flags, device_build):
    """Groups with a hierarchy store their records in hierarchy order, and what decides an exact tie between two
    reported distances - the primitive tested first by the reference wins (strict '<', bvh.cpp:70-95,
    raytracer.cpp:202-247) - is kept per entry slot.  Every primitive of this scene exists twice, with identical
    geometry and different colours, the second copies in shuffled order: every hit is an exact tie.  The frame must
    equal the one of the plain search (no hierarchy), with the records re-ordered and with the option off."""
    rng = np.random.default_rng(5)
    t = sg.random_prims(3000, 21, transparent_frac=0.1, extent=8.0, rmin=0.15, rmax=0.5)
    twin = t[rng.permutation(len(t))].copy()
    twin[:, 4:7] = 1.0 - twin[:, 4:7]
    both = np.concatenate([t[:1500], twin[:1700], t[1500:], twin[1700:]])
    cfg, _ = api.parse_args(flags)
    got = []
    for opts in ({"second_level_min_group": 1 << 40}, {"hierarchy_order": 1}, {"hierarchy_order": 0}):
        ls = api.ArrayScene(both, (480, 270))
        r = api.Renderer(0)
        r.set_option("device_build", device_build)
        for k, v in opts.items():
            r.set_option(k, v)
        r.upload(ls, with_bvh=bool(cfg.bvh))
        img, st = r.render(cfg)
        got.append((img, (st.shadow_rays, st.reflection_rays, st.refraction_rays)))
        r.close(); ls.close()
    assert got[0][0].max() > 0
    # the twins' colours differ, so a wrong winner shows
    assert len(np.unique(got[0][0].reshape(-1, 3), axis=0)) > 100
    for other in got[1:]:
        assert np.array_equal(got[0][0], other[0]) and got[0][1] == other[1]


@pytest.mark.parametrize("flags,moving", [(["-rd", "2", "-grs", "1"], False), (["-r", "-gr", "-grs", "3", "-rd", "2"], False),
                                          (["-m", "-mbs", "2", "-r", "-gr", "-grs", "2", "-ss", "-sss", "4"], "xyz"), (["-u", "-rd", "3", "-grs", "1"], False)])
def test_sorted_secondary_rays_change_no_pixel(flags, moving):
    """Scenes with second-level hierarchies sort the secondary rays of every level by origin cell and direction before
    tracing them.  Only the order of tracing changes: frames and ray counts equal those of the unsorted queues, on one
    batch and on several (a small workspace)."""
    table = sg.random_prims(20000, 13, transparent_frac=0.2, extent=10.0, rmin=0.05, rmax=0.3, moving=bool(moving))
    cfg, _ = api.parse_args(flags)
    got = []
    for sort_rays, workspace in ((1, 0), (0, 0), (1, 16 << 20)):
        ls = api.ArrayScene(table, (480, 270), moving=moving)
        r = api.Renderer(0, workspace)
        r.set_option("sort_rays", sort_rays)
        r.set_option("sort_rays_min_prims", 0)                 # (by default only scenes of millions of primitives sort)
        r.upload(ls, with_bvh=bool(cfg.bvh))
        img, st = r.render(cfg)
        got.append((img, (st.shadow_rays, st.reflection_rays, st.refraction_rays, st.nodes_shaded), st.batches))
        r.close(); ls.close()
    assert got[0][0].max() > 0 and got[0][1][1] > 10000 and got[2][2] > got[0][2]
    for other in got[1:]:
        assert np.array_equal(got[0][0], other[0]) and got[0][1] == other[1]


@pytest.mark.parametrize("n,flags", [(30000, []), (30000, ["-u"]), (20000, ["-m", "-mbs", "1", "-rd", "2", "-grs", "1"]), (300, ["-rd", "2", "-grs", "1"]), (9, [])])
def test_device_built_hierarchies_equal_host_built(n, flags):
    """The hierarchies and leaf-order records made on the GPU (Morton sort, perfect trees) and the ones made on the host
    (median splits) steer the same search: identical frames, identical ray counts."""
    table = sg.random_prims(n, 7, transparent_frac=0.1)
    cfg, _ = api.parse_args(flags)
    got = []
    for device_build in (1, 0):
        ls = api.ArrayScene(table, (640, 360))
        r = api.Renderer(0)
        r.set_option("device_build", device_build)
        r.upload(ls, with_bvh=bool(cfg.bvh))
        img, st = r.render(cfg)
        got.append((img, (st.shadow_rays, st.reflection_rays, st.refraction_rays)))
        r.close(); ls.close()
    assert got[0][0].max() > 0
    assert np.array_equal(got[0][0], got[1][0]) and got[0][1] == got[1][1]


@pytest.mark.parametrize("n,flags,moving", [(20000, ["-m", "-mbs", "4"], True), (20000, ["-m", "-mbs", "2", "-r", "-rd", "2", "-ss", "-sss", "4"], "xyz"),
                                            (20000, ["-m", "-mbs", "2", "-u"], "xyz"), (20000, [], True), (20000, ["-m", "-mbs", "3"], "xyz"),
                                            (30000, ["-m", "-mbs", "2", "-r", "-gr", "-grs", "2", "-rd", "2"], True)])
def test_boxes_per_time_change_no_pixel(n, flags, moving):
    """Scenes in which primitives move keep, in the second-level hierarchies, boxes at time 0 and their change until
    time 1 instead of boxes over the whole motion.  Either kind of box only decides what a ray may skip: frames and ray
    counts are identical with the option off, and identical to testing every primitive of a group one by one."""
    table = sg.random_prims(n, 11, transparent_frac=0.1, moving=True)
    cfg, _ = api.parse_args(flags)
    got = []
    for motion_boxes, min_group in ((1, 8), (0, 8), (1, 1 << 40)):
        if min_group > 8 and n > 20000:
            continue
        ls = api.ArrayScene(table, (480, 270), moving=moving)
        r = api.Renderer(0)
        r.set_option("motion_boxes", motion_boxes)
        r.set_option("second_level_min_group", min_group)
        r.upload(ls, with_bvh=bool(cfg.bvh))
        img, st = r.render(cfg)
        got.append((img, (st.shadow_rays, st.reflection_rays, st.refraction_rays)))
        r.close(); ls.close()
    assert got[0][0].max() > 0
    for other in got[1:]:
        assert np.array_equal(got[0][0], other[0]) and got[0][1] == other[1]


def test_render_into_page_locked_memory():
    """rt_render into memory from rt_alloc_image gives the image it gives into an ordinary array, with and without
    statistics, and the buffer outlives the call that filled it."""
    ls = api.LoadedScene(cases.reference_scene(320, 180), api.parse_args([])[0], str(cases.TEXTURES))
    cfg, _ = api.parse_args(["-r", "-rd", "2"])
    r = api.Renderer(0)
    r.upload(ls, with_bvh=True)
    want, st = r.render(cfg)
    buf = api.image_buffer(r.out_bytes())
    buf[:] = 7
    got, st2 = r.render(cfg, out=buf)
    assert np.array_equal(got, want) and st2.shadow_rays == st.shadow_rays
    buf[:] = 7
    got, none = r.render(cfg, out=buf, want_stats=False)
    assert none is None and np.array_equal(got, want)
    with pytest.raises(ValueError):
        r.render(cfg, out=np.zeros(5, dtype=np.uint8))
    r.close(); ls.close()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("moving", [False, True])
def test_large_scene_windows_match_oracle(moving):
    """50 000 primitives at 1920x1080: the whole frame is out of the oracle's reach, pixel windows are not.
    moving: every primitive has a second location (hierarchies with boxes per time); without -m the reference renders
    them at their first one, with recursion the secondary rays and their shadow rays walk those hierarchies too."""
    n, res = 50000, (1920, 1080)
    text = sg.random_scene(n, 42, res, moving=moving)
    ls = api.ArrayScene(sg.random_prims(n, 42, moving=moving), res, moving=moving)
    for flags in ([], ["-u"]) + ((["-rd", "2", "-grs", "1"],) if moving else ()):
        cfg, _ = api.parse_args(flags)
        r = api.Renderer(0); r.upload(ls, with_bvh=bool(cfg.bvh))
        img, _ = r.render(cfg); r.close()
        for (x0, y0) in [(950, 530), (1660, 190), (300, 900)]:
            win = (x0, y0, x0 + 20, y0 + 12)
            ref, _ = oracle.render(text, flags, window=win)
            a, b = img[y0:y0 + 12, x0:x0 + 20].astype(np.int32), ref[y0:y0 + 12, x0:x0 + 20]
            assert np.array_equal(a, b), f"{flags} window {win}: {(a != b).any(axis=2).sum()} pixels differ"
    ls.close()


def test_million_primitive_frame_matches_the_oracle_on_a_pixel_grid():
    """The large-scene configuration at its own size: 10^6 random primitives (seed 42) at 1920x1080, BVH and -u.  Every
    40th pixel of every 40th row of the CPU restatement's frame - 1296 pixels over as many different primitives - is
    committed as a fixture (tests/golden/make_windows.py: a quarter of an hour and 9 GB on the host, which is why it is
    not rendered here); the frame goes through the second-level hierarchies (thousands of primitives per leaf of the
    reference's depth-8 tree) and the one-kernel frame path."""
    grid = np.load(cases.GOLDEN / "windows_1m.npz")
    n, res = 1_000_000, (1920, 1080)
    ls = api.ArrayScene(sg.random_prims(n, 42), res)
    for flags, tag in (([], "bvh"), (["-u"], "u")):
        cfg, _ = api.parse_args(flags)
        r = api.Renderer(0); r.upload(ls, with_bvh=bool(cfg.bvh))
        img, st = r.render(cfg); r.close()
        assert st.primary_rays == res[0] * res[1]
        ref = grid[f"grid_{tag}"]
        got = img[20::40, 20::40].astype(np.int32)
        assert got.shape == ref.shape and len(np.unique(ref.reshape(-1, 3), axis=0)) > 100
        assert np.array_equal(got, ref), f"{flags}: {(got != ref).any(axis=2).sum()} of {ref.shape[0] * ref.shape[1]} sampled pixels differ"
    ls.close()


def test_ten_million_primitive_frame_matches_the_oracle_on_a_pixel_grid():
    """The largest configuration's scene at its own size: 10^7 random primitives (seed 43) at 3840x2160, under the
    deterministic flag sets the CPU restatement can be compared on bit for bit (the configuration's own effects draw random
    samples): plain, with two levels of mirror reflection, and -u.  Every 80th pixel of every 80th row of the restatement's
    frames (every 240th for -u, where each ray meets all 10^7 primitives) is committed as a fixture
    (tests/golden/make_windows_10m.py).  A leaf of the reference's depth-8 tree holds about 39 000 primitives here
    (bvh.cpp:65-95 tests them one by one)."""
    grid = np.load(cases.GOLDEN / "windows_10m.npz")
    n, res = 10_000_000, (3840, 2160)
    ls = api.ArrayScene(sg.random_prims(n, 43), res)
    for flags, tag, stride in (([], "bvh", 80), (["-r", "-rd", "2", "-grs", "1"], "refl", 80), (["-u"], "u", 240)):
        cfg, _ = api.parse_args(flags)
        r = api.Renderer(0); r.upload(ls, with_bvh=bool(cfg.bvh))
        img, st = r.render(cfg); r.close()
        assert st.primary_rays == res[0] * res[1]
        ref = grid[f"grid_{tag}"]
        got = img[stride // 2::stride, stride // 2::stride].astype(np.int32)
        assert got.shape == ref.shape and len(np.unique(ref.reshape(-1, 3), axis=0)) > (100 if stride == 80 else 2)
        assert np.array_equal(got, ref), f"{flags}: {(got != ref).any(axis=2).sum()} of {ref.shape[0] * ref.shape[1]} sampled pixels differ"
        if tag == "refl":
            assert st.reflection_rays > 0
    ls.close()


@pytest.mark.parametrize("flags", [[], ["-u"], ["-rd", "2", "-grs", "1"]])
def test_second_level_hierarchy_changes_no_pixel_at_a_million_primitives(flags):
    """second_level_min_group on / off at 10^6 primitives on a 160x90 film: the plain search tests the thousands of
    primitives of every pierced leaf (all 10^6 with -u) one by one, as bvh.cpp:65-95 and raytracer.cpp:202-247 do."""
    table = sg.random_prims(1_000_000, 42, transparent_frac=0.1)
    cfg, _ = api.parse_args(flags)
    got = []
    for min_group in (8, 1 << 40):
        ls = api.ArrayScene(table, (160, 90))
        r = api.Renderer(0)
        r.set_option("second_level_min_group", min_group)
        r.upload(ls, with_bvh=bool(cfg.bvh))
        img, st = r.render(cfg)
        got.append((img, (st.shadow_rays, st.reflection_rays, st.refraction_rays))); r.close(); ls.close()
    assert got[0][0].max() > 0
    assert np.array_equal(got[0][0], got[1][0]) and got[0][1] == got[1][1]


def test_randomised_scenes_and_flags_match_oracle(textures_dir):
    """80 random small scenes (all primitive classes, tilted shuffled n-gons, transparency, several lights, textures,
    moving objects) x random deterministic flag sets: images bit-exact, reference-equivalent ray counts equal."""
    import sys
    sys.path.insert(0, str(cases.ROOT / "tools"))
    import fuzz_parity
    rng = np.random.default_rng(2025)
    for k in range(80):
        text = fuzz_parity.random_scene(rng, (96, 54)); flags = fuzz_parity.random_flags(rng)
        ref, ost = oracle.render(text, flags, textures_dir)
        img, st = api.render_scene_text(text, flags, textures_dir)
        assert np.array_equal(img.astype(np.int32), ref), f"case {k} flags {flags}: {(img != ref).any(axis=2).sum()} pixels differ"
        assert (st.primary_rays, st.shadow_rays_ref, st.reflection_rays_ref, st.refraction_rays_ref) == \
               (ost.primary, ost.shadow, ost.reflection, ost.refraction), f"case {k} flags {flags}: ray counts"


def test_scene_blob_round_trip_renders_identically(textures_dir):
    """The arrays rank 0 broadcasts (api.pack_scene) rebuild a scene that renders to the same image."""
    flags = ["-t", "-rd", "2", "-grs", "1"]
    cfg, _ = api.parse_args(flags)
    ls = api.LoadedScene(cases.SCENES["glass_tex"](), cfg, textures_dir)
    ps = api.PackedScene(api.pack_scene(ls, True))
    imgs = []
    for s in (ls, ps):
        r = api.Renderer(0); r.upload(s); imgs.append(r.render(cfg)[0]); r.close()
    assert np.array_equal(imgs[0], imgs[1]) and imgs[0].max() > 0


def test_graph_replay_equals_plain_launches():
    """rt_render_device records a frame's launch sequence on its second call and replays it from the third on;
    the replayed frames must be the frame rt_render produces, also after the scene changes."""
    flags = ["-rd", "3", "-grs", "1", "-ss", "-sss", "4"]
    cfg, _ = api.parse_args(flags)
    r = api.Renderer(0)
    for scene in ("ref", "glass"):
        ls = api.LoadedScene(cases.SCENES[scene](), cfg)
        r.upload(ls)
        want, _ = r.render(cfg)
        out = torch.zeros(r.out_bytes(), dtype=torch.uint8, device="cuda")
        for k in range(5):
            out.zero_()
            r.render_device(cfg, out.data_ptr()); r.synchronize()
            got = out.cpu().numpy().reshape(want.shape)
            assert np.array_equal(got, want), f"{scene}: call {k} differs"
        ls.close()
    r.set_option("graphs", 0)
    out.zero_(); r.render_device(cfg, out.data_ptr()); r.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(want.shape), want)
    r.close()


def test_recorded_frames_survive_reallocation_of_the_work_buffers():
    """A deeper or wider frame re-allocates the work buffers; frame graphs recorded before hold the old addresses and
    must be dropped: configuration A three times (recorded and replayed), then B (grows the buffers), then A again."""
    cfg_a, _ = api.parse_args(["-grs", "1"])
    cfg_b, _ = api.parse_args(["-rd", "4", "-gr", "-grs", "3", "-ss", "-sss", "4"])
    ls = api.LoadedScene(cases.SCENES["glass"](), cfg_a)
    r = api.Renderer(0); r.upload(ls)
    want_a, _ = r.render(cfg_a)
    r.close()
    r = api.Renderer(0); r.upload(ls)                     # fresh context: its first frames size the buffers for A only
    out = torch.zeros(r.out_bytes(), dtype=torch.uint8, device="cuda")
    out_b = torch.zeros_like(out)
    for _ in range(3):
        out.zero_(); r.render_device(cfg_a, out.data_ptr()); r.synchronize()
        assert np.array_equal(out.cpu().numpy().reshape(want_a.shape), want_a)
    r.render_device(cfg_b, out_b.data_ptr()); r.synchronize()
    assert int(out_b.max()) > 0
    filler = [torch.full((1 << 22,), 255, dtype=torch.uint8, device="cuda") for _ in range(8)]   # may land where the old buffers were
    for _ in range(3):
        out.zero_(); r.render_device(cfg_a, out.data_ptr()); r.synchronize()
        assert np.array_equal(out.cpu().numpy().reshape(want_a.shape), want_a)
    assert all(int(t.min()) == 255 for t in filler)
    r.close(); ls.close()


def test_rank_without_tiles_renders_an_empty_block():
    """More ranks than tiles (a 64x32 film has two): the ranks that own nothing return a zeroed block, and the assembled
    image is the unsharded one."""
    text = cases.reference_scene(64, 32)
    cfg, _ = api.parse_args(["-r", "-grs", "1"])
    ls = api.LoadedScene(text, cfg)
    whole = api.Renderer(0); whole.upload(ls)
    full, _ = whole.render(cfg); whole.close()
    blocks = []
    for rank in range(8):
        r = api.Renderer(0); r.upload(ls); r.set_shard(rank, 8)
        out = torch.full((r.out_bytes(),), 9, dtype=torch.uint8, device="cuda")
        st = r.render_device(cfg, out.data_ptr(), want_stats=True)
        if r.local_tiles == 0:
            assert st.total_rays() == 0 and int(out.max()) == 0
            blk, _ = r.render(cfg)
            assert blk.max() == 0
        blocks.append(out.cpu().numpy()); r.close()
    assert full.max() > 0 and np.array_equal(api.assemble_tiles_numpy(blocks, 64, 32), full)
    ls.close()


def test_shadow_overlap_equals_serial_launches():
    """Without statistics the shadow rays of the primary hits run on a second stream beside the deeper levels'
    tracing; with many batches per frame (small workspace, several samples) the forked frames must equal the
    single-stream ones bit for bit, with and without graph replay."""
    flags = ["-rd", "3", "-grs", "1", "-ss", "-sss", "4", "-aa", "-aas", "3", "-dof", "-dofs", "2"]
    cfg, _ = api.parse_args(flags)
    ls = api.LoadedScene(cases.SCENES["glass"](), cfg)
    imgs = {}
    for overlap in (0, 1):
        for graphs in (0, 1):
            r = api.Renderer(0, workspace_bytes=24 << 20)
            r.set_option("overlap", overlap); r.set_option("graphs", graphs)
            r.upload(ls)
            out = torch.zeros(r.out_bytes(), dtype=torch.uint8, device="cuda")
            for _ in range(3):
                out.zero_(); r.render_device(cfg, out.data_ptr()); r.synchronize()
            imgs[(overlap, graphs)] = out.cpu().numpy().copy()
            r.close()
    ls.close()
    base = imgs[(0, 0)]
    assert base.max() > 0
    for key, img in imgs.items():
        assert np.array_equal(img, base), f"overlap, graphs = {key} differs from the serial frame"


@pytest.mark.parametrize("inline", [False, True])
def test_pipelined_frames_two_shards_in_one_process(inline):
    """The multi-rank frame pipeline (render k+1 beside gather k and the re-ordering of k, double-buffered) with two
    shard contexts on one GPU and device copies standing in for the RCCL gather.  Every frame uses another seed, so a
    block or gather buffer that is reused too early shows up as a wrong final image."""
    from raytracer_amd import dist as rdist
    import copy
    text = cases.reference_scene(640, 360)
    flags = ["-rd", "2", "-grs", "1", "-ss", "-sss", "4"]
    cfg, _ = api.parse_args(flags)
    ls = api.LoadedScene(text, cfg)
    dev = torch.device("cuda", 0)
    world = 2
    rs = []
    for k in range(world):
        r = api.Renderer(0); r.upload(ls); r.set_shard(k, world); rs.append(r)
    n = rs[0].out_bytes()
    comm = torch.cuda.Stream(dev)
    mailbox, handles = {}, {}

    class Handle:
        ev = None
        def wait(self):
            if self.ev is not None:
                torch.cuda.current_stream().wait_event(self.ev)

    def make_gather(rank):
        count = [0]
        def start(block, out):
            k = count[0]; count[0] += 1
            h = handles.setdefault(k, Handle())
            ready = torch.cuda.Event(); ready.record(torch.cuda.current_stream())
            mailbox.setdefault(k, {})[rank] = (block, ready)
            if rank == 0:                                   # rank 1 has stepped first in every frame
                with torch.cuda.stream(comm):
                    for q in range(world):
                        blk, ev = mailbox[k][q]
                        comm.wait_event(ev)
                        out[q * n:(q + 1) * n].copy_(blk, non_blocking=True)
                    h.ev = torch.cuda.Event(); h.ev.record(comm)
            return h
        return start

    pipes = [rdist.FramePipeline(rs[k], copy.copy(cfg), k, world, dev, torch.cuda.ExternalStream(rs[k].stream_handle(), device=dev), make_gather(k),
                                 assemble_inline=inline)
             for k in range(world)]
    frames = 7
    for frame in range(frames):
        for k in (1, 0):
            pipes[k].cfg.seed = 100 + frame
            pipes[k].step()
    torch.cuda.synchronize()
    got = pipes[0].image.cpu().numpy().reshape(360, 640, 3)
    single = api.Renderer(0); single.upload(ls)
    last = copy.copy(cfg); last.seed = 100 + frames - 1
    want, _ = single.render(last)
    prev = copy.copy(cfg); prev.seed = 100 + frames - 2
    other, _ = single.render(prev)
    assert not np.array_equal(want, other)
    assert np.array_equal(got, want)
    for r in rs + [single]:
        r.close()
    ls.close()


@pytest.mark.parametrize("kind", ["coincident", "collinear", "nonfinite", "nine", "huge"])
def test_degenerate_populations_build_and_render_like_the_plain_search(kind):
    """Hierarchy construction on populations that break assumptions of a spatial sort: every centre identical, all on
    a line, non-finite coordinates, a group just above the threshold, coordinates spanning 10^12.  The frame must equal
    the one rendered with the primitives searched one by one, with device-built and host-built hierarchies."""
    rng = np.random.default_rng(3)
    n = 9 if kind == "nine" else 600
    t = sg.random_prims(n, 9)
    if kind == "coincident":
        t[:, 0:3] = (0.5, 4.0, 1.5); t[:, 3] = np.linspace(0.2, 0.9, n)
    elif kind == "collinear":
        t[:, 0] = np.linspace(-6, 6, n); t[:, 1] = 5.0; t[:, 2] = 1.0
    elif kind == "nonfinite":
        t[::7, 0] = np.nan; t[3::11, 1] = np.inf; t[5::13, 3] = np.nan
    elif kind == "huge":
        t[::5, 0] *= 1e12; t[1::5, 2] = -1e12; t[2::5, 3] = 1e6
    cfg, _ = api.parse_args(["-rd", "1", "-grs", "1"])
    got = []
    for opts in ({"second_level_min_group": 1 << 40}, {"device_build": 1}, {"device_build": 0}):
        ls = api.ArrayScene(t, (160, 90))
        r = api.Renderer(0)
        for k, v in opts.items():
            r.set_option(k, v)
        r.upload(ls)
        img, st = r.render(cfg)
        got.append((img, st.shadow_rays_ref, st.reflection_rays_ref)); r.close(); ls.close()
    for img, a, b in got[1:]:
        assert np.array_equal(img, got[0][0]) and (a, b) == got[0][1:], kind
