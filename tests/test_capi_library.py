"""The C-ABI library loads and exports every symbol include/ort.h declares (no compute, no GPU)."""
import ctypes as C
import os
import re

from opticalraytrace_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "ort.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ort_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(hip_library):
    lib = C.CDLL(hip_library)
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ort.h but not exported"
    assert sorted(capi.EXPORTED_SYMBOLS) == names


def test_struct_layout_matches_header(hip_library):
    assert C.sizeof(capi.OrtSurface) == 112
    assert C.sizeof(capi.OrtSystem) == 32 + 2 * 12 * 112 + 33 * 8
    lib = capi.load_library()
    assert lib.ort_abi_version() == capi.ABI_VERSION


def test_library_is_built_from_the_sources_in_this_tree(hip_library):
    """ort_build_id() = hash of the kernel sources (csrc/Makefile): a stale binary is detected."""
    assert capi.build_id() == capi.source_build_id()
    assert len(capi.build_id()) == 16 and capi.build_id() != "unstamped"


def test_invalid_arguments_return_codes_without_a_device(hip_library):
    lib = capi.load_library()
    h = C.c_void_p()
    assert lib.ort_create(None, 0, None, C.byref(h)) == -1          # ORT_E_INVALID before any device work
    assert lib.ort_trace(None, 2, 0, 10, 1) == -1
    assert lib.ort_read(None, None, None) == -1
    assert lib.ort_allreduce(None, 1) == -1 and lib.ort_allreduce(None, 0) == -1
    assert b"NULL" in lib.ort_last_error()


def test_product_never_imports_the_oracle():
    """The oracle is a checker: nothing under opticalraytrace_amd/ may reference it."""
    pkg = os.path.join(ROOT, "opticalraytrace_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(d, f)).read()
                assert "oracle" not in src, (d, f)


def test_emitter_and_surface_codes_match_the_header():
    """capi.py restates the ORT_EMIT_* / ORT_SURF_* / flag codes of include/ort.h: they must be the header's."""
    text = open(os.path.join(ROOT, "include", "ort.h")).read()
    defs = {k: int(v.rstrip("u"), 0) for k, v in re.findall(r"#define\s+(ORT_[A-Z_0-9]+)\s+(\d+u?)\b", text)}
    for name in ("RING", "POINT", "SPOT", "CRS", "IMAGE", "ISORS", "ISORS_NORING"):
        assert defs["ORT_EMIT_" + name] == getattr(capi, "EMIT_" + name), name
    assert defs["ORT_EMIT_ISORS_NORING"] == 6
