"""The device computes sphere texture coordinates with restatements of glibc's atan2f / asinf (csrc/ref_libm.h):
compiled for the host, they must agree with the C library bit for bit."""
import shutil
import subprocess

import pytest

import cases


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_ref_libm_equals_the_c_library(tmp_path):
    exe = tmp_path / "check_ref_libm"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-builtin", f"-I{cases.ROOT / 'raytracer_amd' / 'csrc'}",
                    "-o", str(exe), str(cases.ROOT / "tests" / "check_ref_libm.cpp")], check=True)
    p = subprocess.run([str(exe), "97", "3000000"], capture_output=True, text=True)      # ~70 M inputs, a few seconds
    assert p.returncode == 0, p.stdout[-2000:]
    assert "0 mismatches" in p.stdout
