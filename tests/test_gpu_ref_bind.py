"""The reference's own front end bound to the HIP library: oracle/_ref/ref_bind is oracle/ref_bind.cpp linked against
the reference's unmodified objects - its flag parser, its five scene parsers, its PPM reader, BVHNode::buildBVH - and
librt_amd.so.  It flattens the reference's pointer tree into rt_bvh_node[], calls rt_upload_scene / rt_upload_bvh /
rt_render in place of the pixel loop and writes the frame with Image::modifyPixel / writeImage.  The PPM it writes
must be the one the unmodified reference wrote for the golden fixtures."""
import numpy as np
import pytest

import cases
import ref_runner

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_runner.have_bind(), reason="oracle/_ref/ref_bind not built (needs the reference's sources: make -C oracle bind)")]

DET = np.load(cases.GOLDEN / "deterministic.npz")
TEX = {p.name: p for p in cases.TEXTURES.glob("*.ppm")}


@pytest.mark.parametrize("scene,flags", [("ref", []), ("ref", ["-u"]), ("ref", ["-rd", "4", "-grs", "1"]), ("glass_tex", ["-t"]),
                                         ("glass_tex", ["-t", "-rd", "2", "-grs", "1"]), ("bvh5", []), ("bvh5", ["-rd", "3", "-grs", "1"]),
                                         ("test2", []), ("rand2k", ["-rd", "2", "-grs", "1"])],
                         ids=lambda v: "".join(v) if isinstance(v, list) else v)
def test_reference_front_end_bound_to_the_library_writes_the_reference_image(scene, flags):
    img, _, out = ref_runner.run_ref(cases.SCENES[scene](), flags, TEX, exe=ref_runner.BIND_BIN)
    gold = DET[cases.case_id(scene, flags)].astype(np.int32)
    assert img.shape == gold.shape
    assert np.array_equal(img, gold), f"{(img != gold).any(axis=2).sum()} pixels differ"
    if "-u" not in flags:
        assert "bound the reference's BVH" in out
