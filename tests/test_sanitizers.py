"""AddressSanitizer + UndefinedBehaviorSanitizer over the checkers' C on the CPU (sanitizers belong here: the GPU pool
offers none): oracle/ort_oracle.c on every light source and on the scattering walk, oracle/ref/ref_rng.c (the one
runtime entry the compiled reference is given), and csrc/ort_libm.h compiled for the host (check_libm_host).
A finding ends the child process with a sanitizer report and a non-zero status."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g"]


def _asan_runtime():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not p or not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip("gcc's libasan.so is not installed")
    return os.path.realpath(p)


def _san_env():
    env = dict(os.environ)
    env["LD_PRELOAD"] = _asan_runtime()
    env["ASAN_OPTIONS"] = "detect_leaks=0:exitcode=97"          # (the interpreter's own allocations are not the subject)
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1:exitcode=98"
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_oracle_and_rng_harness_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    script = textwrap.dedent("""
        import ctypes as C, os, sys
        import numpy as np
        sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
        from conftest import make_system
        from oracle.binding import Oracle
        seed = 123456789
        for name in ("large", "ellipse", "small_iris_after", "small_f60_nobottle", "small_spot", "large_crs", "small_isors",
                     "large_image", "small_scatter_c", "small_scatter_bc"):
            settings, osys = make_system(name)
            orc = Oracle(osys)
            n = min(settings.nphotons, 3000)
            for phase in (1, 2):
                out = orc.trace_rays(phase, n, seed=seed)                      # keyed draws, in-oracle emission
                m = min(n, 500)
                u = np.random.default_rng(phase).random((8, m))                # a SHORT table: rays that want more draws run off its end
                orc.trace_rays(phase, m, u=u)
                em = out["emitted"][:, :m].copy()
                orc.trace_rays(phase, m, pos_dir_in=em, u=u, draw_base=2)
            img, cnt = orc.trace(1, 0, n, seed)
            orc.trace(2, 7, n, seed, img, cnt)
            assert int(img.sum()) == int(cnt[4] + cnt[5]), name
        orc.set_wide_draws(True); orc.trace_rays(2, 1000, seed=seed); orc.set_wide_draws(False)
        # the reference harness's uniform source (oracle/ref/ref_rng.c): table and keyed mode, the runtime entry itself
        rng = C.CDLL(os.path.join(ROOT, "oracle", "_san", "libref_rng_san.so"))
        rng.ortref_draw.restype = C.c_double
        rng.ortref_rng_table.argtypes = [C.POINTER(C.c_double), C.c_int64, C.c_int32, C.c_int32]
        rng.ortref_rng_key.argtypes = [C.c_uint64, C.c_int32, C.c_uint64, C.c_int32]
        t = np.arange(12, dtype=np.float64) / 16.0
        rng.ortref_rng_table(t.ctypes.data_as(C.POINTER(C.c_double)), 3, 4, 0)
        got = [rng.ortref_draw() for _ in range(6)]                             # two draws beyond the table
        assert got[:4] == [0.0, 3 / 16, 6 / 16, 9 / 16] and got[4:] == [0.5, 0.5], got
        rng.ortref_rng_key(seed, 2, (1 << 40) - 1, 0)
        x = C.c_double(-1.0); px = C.pointer(x); desc = C.pointer(px)           # a descriptor begins with the base address
        rng._FortranARandomNumber(desc, b"f.f90", 44)
        assert 0.0 <= x.value < 1.0 and rng.ortref_rng_draws() == 1
        sz = C.c_int(0); psz = C.pointer(sz); rng._FortranARandomSeedSize(C.pointer(psz), b"f", 1); assert sz.value == 1
        rng._FortranARandomSeedPut(None, b"f", 1)
        print("sanitized run complete")
        """).replace("ROOT", repr(ROOT))
    env = _san_env()
    env["ORT_ORACLE_SO"] = os.path.join(ROOT, "oracle", "_san", "libort_oracle_san.so")
    p = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0 and "sanitized run complete" in p.stdout, p.stdout[-2000:] + p.stderr[-6000:]


def test_libm_restatement_under_asan_ubsan():
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    exe = os.path.join(ROOT, "build", "check_libm_host_san")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-mfma", "-ffp-contract=off", "-fno-fast-math"] + SAN_FLAGS +
                       ["-o", exe, os.path.join(ROOT, "tests", "csrc", "check_libm_host.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:exitcode=97", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=98")
    out = subprocess.run([exe, "400000"], capture_output=True, text=True, timeout=900, env=env)
    from oracle.binding import host_libm_is_pinned
    if host_libm_is_pinned():
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    else:                                                   # another libm: mismatches are expected (status 1), a report is not
        assert out.returncode in (0, 1) and "Sanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-4000:]
