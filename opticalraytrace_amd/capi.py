"""ctypes binding of libort_hip.so (include/ort.h) — the thin host/device boundary.

There is no CPU fallback: if the HIP library is missing or no device is present
every call raises.  The library is built in-tree by `__graft_entry__.build()` or
`make -C opticalraytrace_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import numpy as np

from .system import MAX_SURFACES, OpticalSystem, TWOPI

ABI_VERSION = 2
IMAGE_N = 401
IMAGE_BINS = 2 * IMAGE_N * IMAGE_N
NUM_COUNTERS = 8
MAX_PATH = 6                 # ORT_MAX_PATH
MAX_RAYS_PER_LAUNCH = 1 << 27    # ORT_MAX_RAYS_PER_LAUNCH

ST_BINNED, ST_NA_REJECT, ST_OFF_GRID, ST_LOST_BOTTLE, ST_LOST_TELESCOPE, ST_HELP3, ST_NO_INTERSECTION = range(7)
(C_LOST_RING, C_LOST_POINT, C_ISECT_RING, C_ISECT_POINT,
 C_BINNED_RING, C_BINNED_POINT, C_HELP3_RING, C_HELP3_POINT) = range(8)

EMIT_RING, EMIT_POINT, EMIT_SPOT, EMIT_CRS, EMIT_IMAGE, EMIT_ISORS, EMIT_ISORS_NORING = range(7)

_ERRORS = {-1: "ORT_E_INVALID", -2: "ORT_E_NODEVICE", -3: "ORT_E_HIP", -4: "ORT_E_NOMEM", -5: "ORT_E_NOCOMM"}


class OrtError(RuntimeError):
    pass


class OrtSurface(C.Structure):
    _fields_ = [("cx", C.c_double), ("cy", C.c_double), ("cz", C.c_double),
                ("radius", C.c_double), ("radius_b", C.c_double),
                ("n1", C.c_double), ("n2", C.c_double), ("eta", C.c_double),
                ("aperture", C.c_double), ("mua", C.c_double), ("mus", C.c_double),
                ("hgg", C.c_double), ("scat_radius", C.c_double),
                ("kind", C.c_int32), ("flags", C.c_uint32)]


class OrtSystem(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("n_surfaces", C.c_int32 * 2),
                ("ring_ellipse", C.c_int32), ("split", C.c_int32 * 2), ("emitter", C.c_int32 * 2),
                ("surfaces", (OrtSurface * MAX_SURFACES) * 2),
                ("cos_theta_max", C.c_double),
                ("ring_r1", C.c_double), ("ring_r2", C.c_double),
                ("ring_lens_r2", C.c_double), ("ring_lens_z", C.c_double),
                ("ring_bottle_ra", C.c_double), ("ring_bottle_rb", C.c_double),
                ("ring_bottle_z", C.c_double),
                ("bin_width", C.c_double), ("inv_bin_width", C.c_double), ("na_angle", C.c_double), ("na_cos_min", C.c_double),
                ("twopi", C.c_double), ("spot_dphi", C.c_double), ("spot_dtheta", C.c_double),
                ("crs_sigma", C.c_double), ("crs_radius", C.c_double), ("crs_cy", C.c_double),
                ("crs_cz", C.c_double), ("img_lens_r2", C.c_double), ("img_lens_z", C.c_double),
                ("point_offset", C.c_double),
                ("isors_sigma", C.c_double), ("isors_k", C.c_double), ("isors_height", C.c_double),
                ("isors_base_pos", C.c_double), ("isors_z", C.c_double), ("isors_rad1", C.c_double),
                ("isors_rad2", C.c_double), ("isors_cy", C.c_double), ("isors_cz", C.c_double),
                ("isors_lens_r2", C.c_double), ("isors_lens_z", C.c_double)]


def pack_system(osys: OpticalSystem) -> OrtSystem:
    """OpticalSystem -> the POD the kernels stage into LDS (include/ort.h `ort_system`)."""
    cs = OrtSystem()
    cs.abi_version = ABI_VERSION
    for ph in (1, 2):
        surfs = osys.surfaces(ph)
        cs.n_surfaces[ph - 1] = len(surfs)
        cs.split[ph - 1] = osys.queue_split_of(surfs, ph)
        for k, s in enumerate(surfs):
            d = cs.surfaces[ph - 1][k]
            d.cx, d.cy, d.cz = s.cx, s.cy, s.cz
            d.radius, d.radius_b = s.radius, s.radius_b
            d.n1, d.n2 = s.n1, s.n2
            d.eta = s.n1 / s.n2               # surfaces.f90:279,352: n1/n2 (IEEE division)
            d.aperture = s.aperture
            d.mua, d.mus, d.hgg, d.scat_radius = s.mua, s.mus, s.hgg, s.scat_radius
            d.kind, d.flags = s.kind, s.flags
    l2 = osys.L2[0]
    b = osys.bottle
    cs.ring_ellipse = 1 if b.ellipse else 0
    cs.cos_theta_max = osys.cos_theta_max
    cs.ring_r1, cs.ring_r2 = osys.r1, osys.r2
    rl = l2.radius + 10e-3                   # sourceMod.f90:285
    cs.ring_lens_r2 = rl * rl
    cs.ring_lens_z = l2.fb
    cs.ring_bottle_ra, cs.ring_bottle_rb, cs.ring_bottle_z = b.radiusa, b.radiusb, b.centre[2]
    cs.bin_width = osys.bin_width
    cs.inv_bin_width = 1.0 / osys.bin_width
    cs.na_angle = osys.na_angle
    cs.na_cos_min = osys.na_cos_min
    cs.twopi = TWOPI
    # emitters per phase (src/main.f90:95-101, :132-142)
    src = osys.settings.light_source
    cs.emitter[0] = {"crs": EMIT_CRS, "isors": EMIT_ISORS}.get(src, EMIT_RING)
    cs.emitter[1] = {"spot": EMIT_SPOT, "image": EMIT_IMAGE}.get(src, EMIT_POINT)
    # an emitter no settings file selects (EMIT_ISORS_NORING: the reference's call of it is commented out, src/main.f90:141)
    for p, e in enumerate(getattr(osys, "emitter_override", None) or (None, None)):
        if e is not None:
            cs.emitter[p] = int(e)
    l2p = osys.L2[1]                          # main.f90:133: emit_image(imgin, pos, dir, L2) after the 843 nm rebuild
    cs.img_lens_r2 = l2p.radius * l2p.radius
    cs.img_lens_z = l2p.fb
    nrays_sqrt = math.sqrt(float(osys.settings.nphotons))          # sourceMod.f90:135-141
    if nrays_sqrt > 0:
        cs.spot_dphi = TWOPI / nrays_sqrt
        cs.spot_dtheta = math.acos(osys.cos_theta_max) / nrays_sqrt
    cs.crs_sigma = osys.crs_spot_size
    cs.crs_radius = b.radiusa + b.thickness
    cs.crs_cy, cs.crs_cz = b.centre[1], b.centre[2]
    # isors source (src/sourceMod.f90:162-247, src/main.f90:97,140)
    cs.point_offset = b.centre[2] if src == "isors" else 0.0
    s = osys.settings
    axicon_n, radius, height = 1.4, 12.7e-3, 1.1e-3                 # sourceMod.f90:180-182
    alpha = math.atan(height / radius)
    cs.isors_sigma = s.ring_width
    cs.isors_k = (radius / height) * (radius / height)
    cs.isors_height = height
    cs.isors_base_pos = (s.isors_offset + s.ring_width) / math.tan(alpha * (axicon_n - 1.))
    cs.isors_z = b.radiusa + b.centre[2] + 2.0 ** -52               # epsilon(1.) of a real*8
    cs.isors_rad1 = b.radiusa - b.thickness
    cs.isors_rad2 = (b.radiusb - b.thickness) if b.ellipse else (b.radiusa - b.thickness)
    cs.isors_cy, cs.isors_cz = b.centre[1], b.centre[2]
    cs.isors_lens_r2 = l2.radius * l2.radius
    cs.isors_lens_z = l2.fb
    return cs


_LIB: Optional[C.CDLL] = None
_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int32)


def library_path() -> str:
    if os.environ.get("ORT_HIP_LIB"):          # development: A/B another build of the same ABI
        return os.environ["ORT_HIP_LIB"]
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libort_hip.so")


# csrc/Makefile SOURCES, in its order: the translation units, then the headers
KERNEL_UNITS = ("ort_hip", "ort_k_prog64", "ort_k_strict", "ort_k_wide", "ort_k_prog32", "ort_k_fast", "ort_k_generic", "ort_k_scatter",
                "ort_k_batch", "ort_k_exp")
KERNEL_SOURCES = tuple(u + ".hip" for u in KERNEL_UNITS) + (
    "ort_device.h", "ort_fastd.h", "ort_libm.h", "ort_libm_tables.h", os.path.join("..", "..", "include", "ort.h"),
    "ort_trace.h", "ort_scatter.h", "ort_launch.h", "ort_k_program.h")


def pack_systems(systems):
    """A contiguous array of ort_system records (ort_trace_batch) from OpticalSystem objects."""
    arr = (OrtSystem * len(systems))()
    for i, osys in enumerate(systems):
        arr[i] = pack_system(osys)
    return arr


def source_build_id() -> str:
    """What ort_build_id() of a library built from the sources in this tree returns
    (csrc/Makefile BUILD_ID: SHA-256 over the kernel sources, first 16 hex digits)."""
    import hashlib
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_library(path: Optional[str] = None, no_torch: Optional[bool] = None, older_build: bool = False) -> C.CDLL:
    """Load libort_hip.so and declare every symbol of include/ort.h.  Raises if absent.

    PyTorch-ROCm bundles its own HIP/HSA runtime.  It must be the first one loaded into the process: libort_hip.so then
    binds to that same runtime (same SONAME), and torch tensors, streams and RCCL share one device context with the
    kernels; loading /opt/rocm's runtime first leaves a later `import torch` with "No HIP GPUs are available".  So torch
    is imported here first — unless the caller says the process will never use it (`no_torch=True`: the single-GPU
    process entry, whose start-up would otherwise be mostly that import; the library then binds to /opt/rocm's runtime).
    ORT_NO_TORCH=1 in the environment is read as that same statement, as an override only: nothing here or in the process
    entry writes it (a child process, or a later RayTracer in the same process, must not inherit the decision).  A process
    that loaded the library without torch and imports torch afterwards is told so (`torch_safe()`)."""
    global _LIB, _LOADED_WITHOUT_TORCH
    if _LIB is not None and path is None:
        return _LIB
    p = path or library_path()
    if no_torch is None:
        no_torch = os.environ.get("ORT_NO_TORCH") == "1"
    import sys as _sys
    if not no_torch or "torch" in _sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    elif path is None:
        _LOADED_WITHOUT_TORCH = True
    if not os.path.exists(p):
        raise OrtError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; "
                       "g.build()'` — there is no CPU fallback for the trace path")
    lib = C.CDLL(p)
    vp, u64, i32, i64 = C.c_void_p, C.c_uint64, C.c_int, C.c_int64
    sig = {
        "ort_abi_version": (C.c_int, []),
        "ort_build_id": (C.c_char_p, []),
        "ort_allreduce": (C.c_int, [C.POINTER(vp), i32]),
        "ort_allreduce_ranks": (C.c_int, [C.POINTER(C.c_int)]),
        "ort_comm_destroy": (C.c_int, []),
        "ort_last_kernel_name": (C.c_int, [vp, C.c_char_p, i32]),
        "ort_last_error": (C.c_char_p, []),
        "ort_device_count": (C.c_int, [C.POINTER(C.c_int)]),
        "ort_create": (C.c_int, [C.POINTER(OrtSystem), i32, vp, C.POINTER(vp)]),
        "ort_destroy": (C.c_int, [vp]),
        "ort_set_system": (C.c_int, [vp, C.POINTER(OrtSystem)]),
        "ort_reset": (C.c_int, [vp]),
        "ort_flush": (C.c_int, [vp]),
        "ort_set_image_source": (C.c_int, [vp, C.POINTER(C.c_int64)]),
        "ort_trace": (C.c_int, [vp, i32, u64, u64, u64]),
        "ort_trace_batch": (C.c_int, [vp, i32, C.POINTER(OrtSystem), i32, u64, u64, u64, C.POINTER(vp), C.POINTER(vp)]),
        "ort_emit": (C.c_int, [vp, i32, u64, u64, u64, vp]),
        "ort_trace_resident": (C.c_int, [vp, i32, u64, u64, u64, i32, vp]),
        "ort_trace_rays": (C.c_int, [vp, i32, i64, _DP, i32, _DP, i32, u64, u64,
                                     _DP, _DP, _IP, _IP, _IP, _IP]),
        "ort_trace_paths": (C.c_int, [vp, i32, i64, u64, u64, _DP, _IP, _IP]),
        "ort_read": (C.c_int, [vp, _IP, C.POINTER(C.c_uint64)]),
        "ort_attach_buffers": (C.c_int, [vp, vp, vp]),
        "ort_device_image": (C.c_int, [vp, C.POINTER(vp)]),
        "ort_device_counters": (C.c_int, [vp, C.POINTER(vp)]),
        "ort_synchronize": (C.c_int, [vp]),
        "ort_work_counters": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "ort_reserve": (C.c_int, [vp, u64]),
        "ort_last_kernel_ms": (C.c_int, [vp, i32, C.POINTER(C.c_float)]),
        "ort_set_timing": (C.c_int, [vp, i32]),
        "ort_kernel_times": (C.c_int, [vp, C.POINTER(C.c_float), i32, C.POINTER(C.c_int)]),
        "ort_set_kernel_variant": (C.c_int, [vp, i32]),
        "ort_set_precision": (C.c_int, [vp, i32]),
    }
    for name, (res, args) in sig.items():
        if older_build and not hasattr(lib, name):     # development (tools/abbench.py): an A/B against a build of an earlier round
            continue
        fn = getattr(lib, name)          # AttributeError if the library lacks a symbol
        fn.restype, fn.argtypes = res, args
    if lib.ort_abi_version() != ABI_VERSION:
        raise OrtError("libort_hip.so ABI version mismatch")
    if path is None:
        _LIB = lib
    return lib


_LOADED_WITHOUT_TORCH = False


def torch_safe() -> bool:
    """False when this process bound libort_hip.so to /opt/rocm's HIP runtime (load_library(no_torch=True)): importing
    torch afterwards would bring a second runtime into the process."""
    return not _LOADED_WITHOUT_TORCH


EXPORTED_SYMBOLS = ["ort_abi_version", "ort_build_id", "ort_allreduce", "ort_allreduce_ranks", "ort_comm_destroy", "ort_last_kernel_name", "ort_last_error", "ort_device_count", "ort_create",
                    "ort_destroy", "ort_set_system", "ort_set_image_source", "ort_reset", "ort_flush", "ort_trace", "ort_trace_batch", "ort_emit",
                    "ort_trace_resident", "ort_trace_rays", "ort_trace_paths", "ort_read", "ort_attach_buffers",
                    "ort_device_image",
                    "ort_device_counters", "ort_synchronize", "ort_work_counters", "ort_reserve", "ort_last_kernel_ms",
                    "ort_set_timing", "ort_kernel_times", "ort_set_kernel_variant", "ort_set_precision"]


def _check(lib, rc: int, what: str) -> None:
    if rc != 0:
        msg = lib.ort_last_error()
        raise OrtError(f"{what} failed: {_ERRORS.get(rc, rc)}: "
                       f"{msg.decode() if msg else ''}")


def _dptr(a: Optional[np.ndarray]):
    return a.ctypes.data_as(_DP) if a is not None else None


def _iptr(a: Optional[np.ndarray]):
    return a.ctypes.data_as(_IP) if a is not None else None


class Context:
    """One device's tracer: image accumulator + counters + staged surface table."""

    def __init__(self, osys: OpticalSystem, device: int = 0, stream: int = 0):
        self.lib = load_library()
        self.system = osys
        self._csys = pack_system(osys)
        self._h = C.c_void_p()
        _check(self.lib, self.lib.ort_create(C.byref(self._csys), device,
                                             C.c_void_p(stream) if stream else None,
                                             C.byref(self._h)), "ort_create")
        self.device = device
        self._stage_image_source(osys)

    def close(self) -> None:
        if self._h:
            self.lib.ort_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ------------------------------------------------------------------
    def set_system(self, osys: OpticalSystem) -> None:
        self.system = osys
        self._csys = pack_system(osys)
        _check(self.lib, self.lib.ort_set_system(self._h, C.byref(self._csys)), "ort_set_system")
        self._stage_image_source(osys)

    def _stage_image_source(self, osys: OpticalSystem) -> None:
        if osys.settings.light_source != "image":
            return
        from .image_source import cdf, histogram, load_image
        counts = histogram(load_image(osys.image_source_path), osys.settings.nphotons, osys.image_seed)
        table = np.ascontiguousarray(cdf(counts), dtype=np.int64)
        _check(self.lib, self.lib.ort_set_image_source(self._h, table.ctypes.data_as(C.POINTER(C.c_int64))),
               "ort_set_image_source")

    def reset(self) -> None:
        _check(self.lib, self.lib.ort_reset(self._h), "ort_reset")

    def flush(self) -> None:
        """Fold the hits still held in the per-XCD replicas into the image (needed before the
        attached device buffers are read by anything but this library)."""
        _check(self.lib, self.lib.ort_flush(self._h), "ort_flush")

    def trace(self, phase: int, first_ray: int, n_rays: int, seed: int) -> None:
        _check(self.lib, self.lib.ort_trace(self._h, phase, first_ray, n_rays, seed), "ort_trace")

    def trace_batch(self, packed, phase: int, first_ray: int, n_rays: int, seed: int, image_ptrs, counter_ptrs) -> None:
        """ort_trace_batch: loop `phase` of every system of `packed` (pack_systems) over the same ray range, simulation i into
        the device accumulators image_ptrs[i] (0 / None: no image wanted) and counter_ptrs[i] — the simulations whose lists are
        surface programs share multi-system launches (include/ort.h).  Asynchronous."""
        n = len(packed)
        vp = C.c_void_p
        imgs = (vp * n)(*[vp(int(p)) if p else vp(None) for p in image_ptrs])
        cnts = (vp * n)(*[vp(int(p)) for p in counter_ptrs])
        _check(self.lib, self.lib.ort_trace_batch(self._h, n, packed, phase, first_ray, n_rays, seed, imgs, cnts), "ort_trace_batch")

    def emit(self, phase: int, first_ray: int, n_rays: int, seed: int, d_pos_dir: int) -> None:
        _check(self.lib, self.lib.ort_emit(self._h, phase, first_ray, n_rays, seed,
                                           C.c_void_p(d_pos_dir)), "ort_emit")

    def trace_resident(self, phase: int, first_ray: int, n_rays: int, seed: int,
                       draw_base: int, d_pos_dir: int) -> None:
        _check(self.lib, self.lib.ort_trace_resident(self._h, phase, first_ray, n_rays, seed,
                                                     draw_base, C.c_void_p(d_pos_dir)),
               "ort_trace_resident")

    def synchronize(self) -> None:
        _check(self.lib, self.lib.ort_synchronize(self._h), "ort_synchronize")

    def work_counters(self):
        """(ring rays culled by segment 0 without being emitted, rays deferred to the literal re-run)
        since the last reset: executed-work bookkeeping, not part of the result."""
        w = np.zeros(2, dtype=np.uint64)
        _check(self.lib, self.lib.ort_work_counters(self._h, w.ctypes.data_as(C.POINTER(C.c_uint64))), "ort_work_counters")
        return int(w[0]), int(w[1])

    def reserve(self, n_rays: int) -> None:
        """Allocate the per-launch scratch of traces of up to n_rays rays now (optional)."""
        _check(self.lib, self.lib.ort_reserve(self._h, n_rays), "ort_reserve")

    def read(self):
        image = np.zeros((2, IMAGE_N, IMAGE_N), dtype=np.int32)
        counters = np.zeros(NUM_COUNTERS, dtype=np.uint64)
        _check(self.lib, self.lib.ort_read(self._h, _iptr(image),
                                           counters.ctypes.data_as(C.POINTER(C.c_uint64))), "ort_read")
        return image, counters

    def attach_buffers(self, d_image: int, d_counters: int) -> None:
        _check(self.lib, self.lib.ort_attach_buffers(
            self._h, C.c_void_p(d_image) if d_image else None,
            C.c_void_p(d_counters) if d_counters else None), "ort_attach_buffers")

    def device_image_ptr(self) -> int:
        p = C.c_void_p()
        _check(self.lib, self.lib.ort_device_image(self._h, C.byref(p)), "ort_device_image")
        return p.value

    def device_counters_ptr(self) -> int:
        p = C.c_void_p()
        _check(self.lib, self.lib.ort_device_counters(self._h, C.byref(p)), "ort_device_counters")
        return p.value

    def set_timing(self, enable: bool) -> None:
        _check(self.lib, self.lib.ort_set_timing(self._h, int(enable)), "ort_set_timing")

    def set_kernel_variant(self, variant: int) -> None:
        _check(self.lib, self.lib.ort_set_kernel_variant(self._h, variant), "ort_set_kernel_variant")

    def kernel_times(self, capacity: int = 64):
        """ms of the most recent ort_trace launches (oldest first); synchronises on their events."""
        buf = (C.c_float * capacity)()
        n = C.c_int(0)
        _check(self.lib, self.lib.ort_kernel_times(self._h, buf, capacity, C.byref(n)), "ort_kernel_times")
        return [buf[i] for i in range(n.value)]

    def last_kernel_name(self) -> str:
        """The kernel instantiation the last trace launch of this context ran, as the source spells it (e.g.
        `(trace_queue_kernel<MODE_FUSED, true, false, T, P, false, RNG>)` is reported with its arguments filled in by the
        launcher that chose it): tests and bench.py state which kernel a figure belongs to."""
        buf = C.create_string_buffer(256)
        _check(self.lib, self.lib.ort_last_kernel_name(self._h, buf, 256), "ort_last_kernel_name")
        return buf.value.decode()

    def set_precision(self, precision: int) -> None:
        """0 = fp64 (reference arithmetic, default), 1 = fp32 study path."""
        _check(self.lib, self.lib.ort_set_precision(self._h, precision), "ort_set_precision")

    def last_kernel_ms(self, kind: int = 0) -> float:
        ms = C.c_float()
        _check(self.lib, self.lib.ort_last_kernel_ms(self._h, kind, C.byref(ms)), "ort_last_kernel_ms")
        return ms.value

    def trace_paths(self, phase: int, n: int, seed: int = 0, first_ray: int = 0):
        """Tracker entry: (path [n][MAX_PATH][3], npath [n], status [n]) for keyed rays."""
        path = np.zeros((n, MAX_PATH, 3))
        npath = np.zeros(n, np.int32)
        status = np.zeros(n, np.int32)
        _check(self.lib, self.lib.ort_trace_paths(self._h, phase, n, seed, first_ray, _dptr(path),
                                                  _iptr(npath), _iptr(status)), "ort_trace_paths")
        return path, npath, status

    def trace_rays(self, phase: int, n: int, pos_dir_in: Optional[np.ndarray] = None,
                   u: Optional[np.ndarray] = None, draw_base: int = 0, seed: int = 0,
                   first_ray: int = 0):
        """Parity entry; returns dict(pos_dir, emitted, status, bin_xy, n_isect, n_draws)."""
        if pos_dir_in is not None:
            pos_dir_in = np.ascontiguousarray(pos_dir_in, dtype=np.float64)
            assert pos_dir_in.shape == (6, n)
        nu = 0
        if u is not None:
            u = np.ascontiguousarray(u, dtype=np.float64)
            assert u.ndim == 2 and u.shape[1] == n
            nu = u.shape[0]
        out = dict(pos_dir=np.zeros((6, n)), emitted=np.zeros((6, n)),
                   status=np.zeros(n, np.int32), bin_xy=np.zeros((2, n), np.int32),
                   n_isect=np.zeros(n, np.int32), n_draws=np.zeros(n, np.int32))
        _check(self.lib, self.lib.ort_trace_rays(
            self._h, phase, n, _dptr(pos_dir_in), nu, _dptr(u), draw_base, seed, first_ray,
            _dptr(out["pos_dir"]), _dptr(out["emitted"]), _iptr(out["status"]),
            _iptr(out["bin_xy"]), _iptr(out["n_isect"]), _iptr(out["n_draws"])), "ort_trace_rays")
        return out


def build_id() -> str:
    return load_library().ort_build_id().decode()


def allreduce(contexts) -> None:
    """ort_allreduce: sum image + counters over the contexts of this process (one per device)."""
    lib = load_library()
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    _check(lib, lib.ort_allreduce(arr, len(contexts)), "ort_allreduce")


def comm_destroy() -> None:
    """ort_comm_destroy: give back the RCCL communicators ort_allreduce keeps for this process."""
    lib = load_library()
    _check(lib, lib.ort_comm_destroy(), "ort_comm_destroy")


def allreduce_ranks() -> int:
    """Ranks of the communicator the last ort_allreduce used, as RCCL reports it (0 before the first)."""
    lib = load_library()
    n = C.c_int(0)
    _check(lib, lib.ort_allreduce_ranks(C.byref(n)), "ort_allreduce_ranks")
    return n.value


def device_count() -> int:
    lib = load_library()
    n = C.c_int(0)
    _check(lib, lib.ort_device_count(C.byref(n)), "ort_device_count")
    return n.value
