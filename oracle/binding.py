"""TEST INFRASTRUCTURE — ctypes bindings of the checkers.

  * Oracle      — oracle/libort_oracle.so, the plain-C restatement (ort_oracle.c)
  * Reference   — oracle/_ref/libort_ref.so, the reference's own Fortran path
                  sources compiled with flang (build container only, or prebuilt)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
Nothing in opticalraytrace_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libort_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libort_ref.so")
REF_PROG = os.path.join(HERE, "_ref", "raytrace")

_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int32)


class OrcVec(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


class OrcPlano(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("thickness", "curve_radius", "radius", "fb", "f", "n1", "n2")] + \
               [("centre", OrcVec), ("flatNormal", OrcVec)]


class OrcDoublet(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("R1", "R2", "R3", "radius", "fb", "f", "n1", "n2", "n3", "thickness")] + \
               [("centre1", OrcVec), ("centre2", OrcVec), ("centre3", OrcVec)]


class OrcBottle(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("nbottle", "ncontents", "thickness", "radiusa", "radiusb")] + \
               [("centre", OrcVec), ("ellipse", C.c_int32), ("pad", C.c_int32)] + \
               [(k, C.c_double) for k in ("mua_b", "mus_b", "mua_c", "mus_c")]


class OrcSystem(C.Structure):
    _fields_ = [("L2", OrcPlano * 2), ("L3", OrcDoublet * 2), ("bottle", OrcBottle)] + \
               [(k, C.c_double) for k in ("cosThetaMax", "r1", "r2", "img_plane", "fibre_offset",
                                          "image_diameter", "iris_radius")] + \
               [(k, C.c_int32) for k in ("iris_before", "iris_after", "use_bottle", "source",
                                         "nphotons", "pad2")] + \
               [(k, C.c_double) for k in ("isors_offset", "ring_width", "spot_size")] + \
               [("img_counts", C.POINTER(C.c_int32))]


SOURCE_CODES = {"point": 0, "spot": 1, "crs": 2, "isors": 3, "image": 4}
ISORS_NO_RING = 5      # test-only source code: iSORS(ring = .false.) in phase 1 (pins bottle_backward_sub)


PINNED_SO = os.path.join(HERE, "libort_oracle_pinned.so")
LIBM_GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden", "libm_glibc235.npz")
_HOST_LIBM_PINNED: Optional[bool] = None


def host_libm_mismatches() -> list:
    """Which of the host libm's sin / cos / sincos / log / atan2 / acos do NOT return glibc 2.35's (x86-64, FMA variants)
    committed known answers (tests/golden/libm_glibc235.npz, a sample of each) — empty on the machine the device code
    was pinned on.  The device reproduces glibc 2.35 wherever it runs; the oracle and oracle/_ref call the HOST's libm."""
    g = np.load(LIBM_GOLD)
    libm = C.CDLL("libm.so.6")
    for f in ("sin", "cos", "log", "acos"):
        getattr(libm, f).restype = C.c_double
        getattr(libm, f).argtypes = [C.c_double]
    libm.atan2.restype = C.c_double
    libm.atan2.argtypes = [C.c_double, C.c_double]
    libm.sincos.restype = None
    libm.sincos.argtypes = [C.c_double, _DP, _DP]

    def same(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return bool(((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))).all())
    bad = []
    idx = np.arange(0, 12000, 7)
    for f, x, w in (("sin", g["ang"], g["sin"]), ("cos", g["ang"], g["cos"]), ("log", g["log_x"], g["log"]), ("acos", g["acos_x"], g["acos"])):
        i = idx[idx < len(x)]
        if not same([getattr(libm, f)(float(v)) for v in x[i]], w[i]):
            bad.append(f)
    i = idx[idx < len(g["atan2"])]
    if not same([libm.atan2(float(y), float(x)) for y, x in zip(g["atan2_y"][i], g["atan2_x"][i])], g["atan2"][i]):
        bad.append("atan2")
    i = idx[idx < len(g["ang"])]
    sv, cv = C.c_double(), C.c_double()
    got = []
    for v in g["ang"][i]:
        libm.sincos(float(v), C.byref(sv), C.byref(cv))
        got.append((sv.value, cv.value))
    got = np.array(got)
    if not (same(got[:, 0], g["sincos_s"][i]) and same(got[:, 1], g["sincos_c"][i])):
        bad.append("sincos")
    return bad


def host_libm_is_pinned() -> bool:
    global _HOST_LIBM_PINNED
    if _HOST_LIBM_PINNED is None:
        _HOST_LIBM_PINNED = not host_libm_mismatches()
    return _HOST_LIBM_PINNED


def oracle_libm() -> str:
    """'host' or 'pinned': which libm the oracle this process loads is built on.  ORT_ORACLE_LIBM=host|pinned decides;
    unset (auto): the host's libm where it IS glibc 2.35's (then oracle == oracle/_ref == the reference built here, bit
    for bit), else the pinned build (oracle/pinned_libm.cpp: csrc/ort_libm.h compiled for the host) — so that the GPU
    parity suite does not go red for the host's reason on a box with another libm."""
    want = os.environ.get("ORT_ORACLE_LIBM", "auto")
    if want in ("host", "pinned"):
        return want
    return "host" if host_libm_is_pinned() else "pinned"


def build_oracle(force: bool = False, libm: Optional[str] = None) -> str:
    """Path of the checker to load: ORT_ORACLE_SO (an explicit build, e.g. the sanitizer one), else the host-libm or the
    pinned-libm build (oracle_libm(); `libm` overrides), (re)built by oracle/Makefile when older than its sources."""
    explicit = os.environ.get("ORT_ORACLE_SO")
    if explicit and libm is None:
        return explicit
    pinned = (libm or oracle_libm()) == "pinned"
    so = PINNED_SO if pinned else ORACLE_SO
    deps = ["ort_oracle.c", "ort_oracle.h"] + (["pinned_libm.cpp", os.path.join("..", "opticalraytrace_amd", "csrc", "ort_libm.h")] if pinned else [])
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(os.path.join(HERE, f)) for f in deps):
        subprocess.run(["make", "-C", HERE, "pinned" if pinned else "oracle"], check=True, capture_output=True)
    return so


def fill_system(osys) -> OrcSystem:
    """opticalraytrace_amd.system.OpticalSystem -> the oracle's mirror of the Fortran types."""
    S = OrcSystem()
    for i in range(2):
        p, d = osys.L2[i], osys.L3[i]
        P = S.L2[i]
        P.thickness, P.curve_radius, P.radius, P.fb, P.f, P.n1, P.n2 = \
            p.thickness, p.curve_radius, p.radius, p.fb, p.f, p.n1, p.n2
        P.centre = OrcVec(0.0, 0.0, p.centre_z)
        P.flatNormal = OrcVec(0.0, 0.0, -1.0)
        D = S.L3[i]
        D.R1, D.R2, D.R3, D.radius, D.fb, D.f = d.R1, d.R2, d.R3, d.radius, d.fb, d.f
        D.n1, D.n2, D.n3, D.thickness = d.n1, d.n2, d.n3, d.thickness
        D.centre1 = OrcVec(0.0, 0.0, d.centre1_z)
        D.centre2 = OrcVec(0.0, 0.0, d.centre2_z)
        D.centre3 = OrcVec(0.0, 0.0, d.centre3_z)
    b = osys.bottle
    B = S.bottle
    B.nbottle, B.ncontents, B.thickness, B.radiusa, B.radiusb = \
        b.nbottle, b.ncontents, b.thickness, b.radiusa, b.radiusb
    B.centre = OrcVec(*b.centre)
    B.ellipse = 1 if b.ellipse else 0
    B.mua_b, B.mus_b, B.mua_c, B.mus_c = b.mua_b, b.mus_b, b.mua_c, b.mus_c
    s = osys.settings
    S.cosThetaMax, S.r1, S.r2, S.img_plane = osys.cos_theta_max, osys.r1, osys.r2, osys.img_plane
    S.fibre_offset, S.image_diameter, S.iris_radius = s.fibre_offset, s.image_diameter, s.iris_size
    S.iris_before, S.iris_after = int(s.iris == "before"), int(s.iris == "after")
    S.use_bottle = int(s.use_bottle)
    S.source = SOURCE_CODES[s.light_source]
    S.nphotons = s.nphotons
    S.isors_offset, S.ring_width, S.spot_size = s.isors_offset, s.ring_width, osys.crs_spot_size
    return S


def _dp(a):
    return a.ctypes.data_as(_DP) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(_IP) if a is not None else None


class Oracle:
    def __init__(self, osys=None, source_override=None, libm: Optional[str] = None):
        """`libm`: None = the session's choice (oracle_libm()), "host" / "pinned" = that build (tests that compare the
        oracle with oracle/_ref — the reference compiled here, on the host's libm — ask for "host")."""
        self.lib = C.CDLL(build_oracle(libm=libm))
        self.lib.orc_pinned_libm.restype = C.c_int
        self.pinned_libm = bool(self.lib.orc_pinned_libm())
        L = self.lib
        L.orc_sellmeier.restype = C.c_double
        L.orc_sellmeier.argtypes = [C.c_double] * 7
        L.orc_cauchy.restype = L.orc_dispersion.restype = C.c_double
        L.orc_cauchy.argtypes = L.orc_dispersion.argtypes = [C.c_double] * 4
        L.orc_uniform.restype = C.c_double
        L.orc_uniform.argtypes = [C.c_uint64, C.c_int32, C.c_uint64, C.c_int32]
        L.orc_set_wide_draws.restype = None
        L.orc_set_wide_draws.argtypes = [C.c_int32]
        L.orc_trace_rays.restype = C.c_int
        L.orc_trace_rays.argtypes = [C.POINTER(OrcSystem), C.c_int, C.c_int64, _DP, C.c_int, _DP,
                                     C.c_int, C.c_uint64, C.c_uint64, _DP, _DP, _IP, _IP, _IP, _IP]
        L.orc_trace.restype = C.c_int
        L.orc_trace.argtypes = [C.POINTER(OrcSystem), C.c_int, C.c_uint64, C.c_uint64, C.c_uint64,
                                _IP, C.POINTER(C.c_uint64), C.c_int]
        L.orc_init_emit_image.argtypes = [_DP, C.c_int32, C.c_uint64, _IP]
        self.sys = fill_system(osys) if osys is not None else None
        if source_override is not None:
            self.sys.source = source_override
        self._counts = None
        if osys is not None and osys.settings.light_source == "image":
            # the oracle builds its own histogram from the image file (its own init_emit_image)
            img = np.fromfile(osys.image_source_path, np.float64)
            assert img.size == 512 * 512
            self._counts = np.zeros(512 * 512, np.int32)
            L.orc_init_emit_image(_dp(img), osys.settings.nphotons, osys.image_seed, _ip(self._counts))
            self.sys.img_counts = self._counts.ctypes.data_as(C.POINTER(C.c_int32))

    def uniform(self, seed, phase, ray, draw) -> float:
        return self.lib.orc_uniform(seed, phase, ray, draw)

    def set_wide_draws(self, on: bool) -> None:
        """Keyed draws: ORT-RNG-v2w (53 bits, one hash per draw) instead of v2.  Process-wide: switch it back."""
        self.lib.orc_set_wide_draws(1 if on else 0)

    def trace_rays(self, phase, n, pos_dir_in=None, u=None, draw_base=0, seed=0, first_ray=0):
        if pos_dir_in is not None:
            pos_dir_in = np.ascontiguousarray(pos_dir_in, dtype=np.float64)
        nu = 0
        if u is not None:
            u = np.ascontiguousarray(u, dtype=np.float64)
            nu = u.shape[0]
        out = dict(pos_dir=np.zeros((6, n)), emitted=np.zeros((6, n)),
                   status=np.zeros(n, np.int32), bin_xy=np.zeros((2, n), np.int32),
                   n_isect=np.zeros(n, np.int32), n_draws=np.zeros(n, np.int32))
        rc = self.lib.orc_trace_rays(C.byref(self.sys), phase, n, _dp(pos_dir_in), nu, _dp(u),
                                     draw_base, seed, first_ray, _dp(out["pos_dir"]),
                                     _dp(out["emitted"]), _ip(out["status"]), _ip(out["bin_xy"]),
                                     _ip(out["n_isect"]), _ip(out["n_draws"]))
        assert rc == 0
        return out

    def trace(self, phase, first, n, seed, image=None, counters=None, nthreads=0):
        if image is None:
            image = np.zeros((2, 401, 401), np.int32)
        if counters is None:
            counters = np.zeros(8, np.uint64)
        rc = self.lib.orc_trace(C.byref(self.sys), phase, first, n, seed, _ip(image),
                                counters.ctypes.data_as(C.POINTER(C.c_uint64)), nthreads)
        assert rc == 0
        return image, counters


def reference_available() -> bool:
    return os.path.exists(REF_SO)


class Reference:
    """The reference's own Fortran path (oracle/_ref/libort_ref.so)."""

    def __init__(self, settings, res_dir: str, image_seed: int = 123456789, source_override=None):
        self.lib = C.CDLL(REF_SO)
        L = self.lib
        L.ortref_init.restype = C.c_int
        L.ortref_init.argtypes = [C.c_char_p] * 3 + [C.c_double] * 6 + \
            [C.c_int, C.c_double, C.c_int]
        L.ortref_constants.argtypes = [_DP]
        L.ortref_trace_rays.argtypes = [C.c_int, C.c_int64, C.c_int, _DP, C.c_int, _DP, C.c_int,
                                        _DP, _DP, _IP, _IP, _IP]
        L.ortref_trace.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, _IP,
                                   C.POINTER(C.c_int64)]
        s = settings
        iris_mode = {"none": 0, "before": 1, "after": 2}[s.iris]
        rc = L.ortref_init(os.path.join(res_dir, s.bottle_file).encode(),
                           os.path.join(res_dir, s.L2_file).encode(),
                           os.path.join(res_dir, s.L3_file).encode(),
                           s.wavelength, s.alpha, s.n_axicon, s.ring_width, s.image_diameter,
                           s.fibre_offset, iris_mode, s.iris_size, int(s.use_bottle))
        assert rc == 0
        L.ortref_set_source.argtypes = [C.c_int, C.c_int] + [C.c_double] * 5
        L.ortref_set_source(source_override if source_override is not None else SOURCE_CODES[s.light_source],
                            s.nphotons, s.isors_offset, s.crs_spot_size, s.alpha, s.n_axicon, s.ring_width)
        self.counts = None
        if s.light_source == "image":
            L.ortref_image_source.argtypes = [C.c_char_p, C.c_int, C.c_int64, _IP]
            self.counts = np.zeros((512, 512), np.int32)      # Fortran imgin(512,512): [second][first] here
            L.ortref_image_source(os.path.join(res_dir, s.image_source).encode(), s.nphotons,
                                  image_seed, _ip(self.counts))

    def constants(self) -> np.ndarray:
        out = np.zeros(64)
        self.lib.ortref_constants(_dp(out))
        return out

    def trace_rays(self, phase, n, pos_dir_in=None, u=None, draw_base=0):
        """u must be given ([nu][n]); the Fortran side has column-major (n, nu) = same memory."""
        u = np.ascontiguousarray(u, dtype=np.float64)
        nu = u.shape[0]
        have_in = pos_dir_in is not None
        pin = np.ascontiguousarray(pos_dir_in, dtype=np.float64) if have_in else np.zeros((6, n))
        out = dict(pos_dir=np.zeros((6, n)), emitted=np.zeros((6, n)),
                   status=np.zeros(n, np.int32), bin_xy=np.zeros((2, n), np.int32),
                   n_draws=np.zeros(n, np.int32))
        self.lib.ortref_trace_rays(phase, n, int(have_in), _dp(pin), nu, _dp(u), draw_base,
                                   _dp(out["pos_dir"]), _dp(out["emitted"]), _ip(out["status"]),
                                   _ip(out["bin_xy"]), _ip(out["n_draws"]))
        return out

    def trace(self, phase, first, n, seed, image=None):
        if image is None:
            image = np.zeros((2, 401, 401), np.int32)
        lost = C.c_int64(0)
        self.lib.ortref_trace(phase, first, n, seed, _ip(image), C.byref(lost))
        return image, lost.value
