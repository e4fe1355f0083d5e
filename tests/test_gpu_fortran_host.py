"""The reference-side binding of INTEGRATION.md §A, for real: a FORTRAN program (compiled here with
flang from the `module ort_c` block of INTEGRATION.md) fills `type(ort_system)`, calls ort_create /
ort_trace / ort_read / ort_destroy on the GPU and writes `image(-200:200,-200:200,2)` in the
reference's own storage order; the result must equal the Python host's bit for bit."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import make_system
from parity import SEED

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLANG = "/opt/rocm/lib/llvm/bin/flang"

PROGRAM = """
program fortran_host
    use iso_c_binding
    use ort_c
    implicit none
    type(ort_system), target :: sys
    type(c_ptr) :: ctx
    integer(c_int32_t) :: image(-200:200, -200:200, 2)      ! src/main.f90:35
    integer(c_int64_t) :: counters(8), nphotons
    type(ort_system) :: systems(2)
    type(c_ptr) :: d_img(2), d_cnt(2)
    integer(c_int32_t) :: image2(-200:200, -200:200, 2)
    integer(c_int64_t) :: counters2(8)
    integer :: u
    open(newunit=u, file="system.bin", access="stream", form="unformatted", status="old")
    read(u) sys
    close(u)
    nphotons = 300000_c_int64_t
    ctx = c_null_ptr
    if (ort_create(sys, 0_c_int, c_null_ptr, ctx) /= 0) error stop "ort_create"
    if (ort_trace(ctx, 1_c_int, 0_c_int64_t, nphotons, 123456789_c_int64_t) /= 0) error stop "ring"
    if (ort_trace(ctx, 2_c_int, 0_c_int64_t, nphotons, 123456789_c_int64_t) /= 0) error stop "point"
    if (ort_read(ctx, image, counters) /= 0) error stop "ort_read"
    open(newunit=u, file="image.bin", access="stream", form="unformatted", status="replace")
    write(u) image
    close(u)
    print '(A,8(1X,I0))', 'counters', counters
    print '(A,1X,I0,1X,I0)', 'layer_sums', sum(int(image(:, :, 1), c_int64_t)), sum(int(image(:, :, 2), c_int64_t))
    print '(A,1X,I0)', 'centre_bin_point', image(0, 0, 2)
    ! a BATCH from the Fortran host (ort_trace_batch, runner.py's loops): two settings files — here twice the same system —
    ! traced in one launch per loop, both into the context's own accumulators: every bin and counter must come out doubled
    systems(1) = sys; systems(2) = sys
    if (ort_reset(ctx) /= 0) error stop "ort_reset"
    if (ort_device_image(ctx, d_img(1)) /= 0) error stop "ort_device_image"
    if (ort_device_counters(ctx, d_cnt(1)) /= 0) error stop "ort_device_counters"
    d_img(2) = d_img(1); d_cnt(2) = d_cnt(1)
    if (ort_trace_batch(ctx, 2_c_int, systems, 1_c_int, 0_c_int64_t, nphotons, 123456789_c_int64_t, d_img, d_cnt) /= 0) error stop "batch ring"
    if (ort_trace_batch(ctx, 2_c_int, systems, 2_c_int, 0_c_int64_t, nphotons, 123456789_c_int64_t, d_img, d_cnt) /= 0) error stop "batch point"
    if (ort_read(ctx, image2, counters2) /= 0) error stop "ort_read"
    print '(A,1X,L1,1X,L1)', 'batch_doubles', all(image2 == 2 * image), all(counters2 == 2 * counters)
    if (ort_destroy(ctx) /= 0) error stop "ort_destroy"
end program
"""


@pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not installed")
def test_fortran_program_drives_the_gpu_path(tmp_path, hip_library):
    from opticalraytrace_amd.capi import Context, pack_system
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```fortran\n(module ort_c.*?end module)\n```", text, flags=re.S)
    assert m, "INTEGRATION.md §A lost its `module ort_c` block"
    (tmp_path / "host.f90").write_text(m.group(1) + "\n" + PROGRAM)
    libdir = os.path.dirname(hip_library)
    r = subprocess.run([FLANG, "-o", "host", "host.f90", "-L" + libdir, "-lort_hip", "-Wl,-rpath," + libdir],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    _, osys = make_system("large")
    (tmp_path / "system.bin").write_bytes(bytes(pack_system(osys)))
    out = subprocess.run([str(tmp_path / "host")], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-2000:]
    lines = {ln.split()[0]: ln.split()[1:] for ln in out.stdout.splitlines() if ln.strip()}
    n = 300000
    with Context(osys, device=0) as ctx:
        ctx.trace(1, 0, n, SEED)
        ctx.trace(2, 0, n, SEED)
        img, cnt = ctx.read()
    got = np.fromfile(tmp_path / "image.bin", dtype=np.int32).reshape(2, 401, 401)   # xp fastest: the same bytes
    assert np.array_equal(got, img)
    assert [int(v) for v in lines["counters"]] == [int(v) for v in cnt]
    assert [int(v) for v in lines["layer_sums"]] == [int(img[0].sum()), int(img[1].sum())]
    assert int(lines["centre_bin_point"][0]) == int(img[1, 200, 200])
    assert int(img[1].sum()) > 100000
    assert lines["batch_doubles"] == ["T", "T"], lines["batch_doubles"]      # ort_trace_batch from Fortran: two simulations, one launch per loop
