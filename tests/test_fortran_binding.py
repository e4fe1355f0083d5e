"""The Fortran `bind(C)` mirror of include/ort.h that INTEGRATION.md §A hands to a maintainer of
the reference is COMPILED here (flang, the compiler the oracle build uses) and linked against
libort_hip.so: struct sizes, ABI version and an error return are checked from Fortran, so the
documented mirror cannot drift from the header unnoticed.  No GPU needed."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLANG = "/opt/rocm/lib/llvm/bin/flang"

PROGRAM = """
program check_binding
    use iso_c_binding
    use ort_c
    implicit none
    type(ort_surface) :: sf
    type(ort_system), target :: sys
    type(c_ptr) :: ctx
    integer(c_int) :: rc
    print '(A,I0)', 'sizeof_surface ', c_sizeof(sf)
    print '(A,I0)', 'sizeof_system ', c_sizeof(sys)
    print '(A,I0)', 'abi ', ort_abi_version()
    sys%abi_version = 0                      ! wrong on purpose: ort_create must refuse before any device work
    sys%n_surfaces = 0
    ctx = c_null_ptr
    rc = ort_create(sys, 0_c_int, c_null_ptr, ctx)
    print '(A,I0)', 'create_rc ', rc
    print '(A,L1)', 'ctx_null ', .not. c_associated(ctx)
    print '(A,I0)', 'destroy_null_rc ', ort_destroy(c_null_ptr)
end program
"""


@pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not installed")
def test_fortran_mirror_of_the_header_compiles_and_agrees(tmp_path, hip_library):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```fortran\n(module ort_c.*?end module)\n```", text, flags=re.S)
    assert m, "INTEGRATION.md §A lost its `module ort_c` block"
    src = tmp_path / "ort_c_check.f90"
    src.write_text(m.group(1) + "\n" + PROGRAM)
    exe = tmp_path / "ort_c_check"
    libdir = os.path.dirname(hip_library)
    r = subprocess.run([FLANG, "-o", str(exe), str(src), "-L" + libdir, "-lort_hip", "-Wl,-rpath," + libdir],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    got = dict(ln.split() for ln in out.stdout.splitlines() if ln.strip())
    assert got["sizeof_surface"] == "112" and got["sizeof_system"] == "2984"      # include/ort.h
    assert got["abi"] == "2"
    assert got["create_rc"] == "-1" and got["ctx_null"] == "T"                     # ORT_E_INVALID, *out left null
    assert got["destroy_null_rc"] == "0"
