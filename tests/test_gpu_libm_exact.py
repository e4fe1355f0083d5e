"""The DEVICE code of csrc/ort_libm.h (glibc 2.35's sin / cos / sincos / log / atan2 / acos restated; straight and
predicated forms) against the host's libm, bit for bit, on the GPU: tests/csrc/check_libm_gpu.hip, built by the csrc
Makefile (`libm-check`) into build/check_libm_gpu."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "check_libm_gpu")


@pytest.mark.gpu
def test_device_libm_is_bit_identical_to_glibc():
    if not os.path.exists(EXE):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "opticalraytrace_amd", "csrc"), "libm-check"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([EXE, "24000000"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if "calls" in ln]
    assert len(lines) == 12 and all("mismatches 0" in ln for ln in lines), out.stdout
