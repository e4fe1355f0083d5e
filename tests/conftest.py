import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (/dev/kfd absent)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hip_library():
    """Path of the in-tree libort_hip.so.  `make` (a no-op when the library is newer than every
    source the Makefile lists) runs wherever hipcc exists; a library whose ort_build_id() is not the
    hash of the sources next to it fails the session: a stale binary cannot pass silently."""
    from opticalraytrace_amd.capi import build_id, library_path, source_build_id
    p = library_path()
    if os.path.exists("/opt/rocm/bin/hipcc") and not os.environ.get("ORT_HIP_LIB"):
        subprocess.run(["make", "-C", os.path.dirname(p), "libort_hip.so"], check=True, capture_output=True)
    if not os.path.exists(p):
        pytest.fail(f"{p} missing and cannot be built: the product has no fallback")
    if not os.environ.get("ORT_HIP_LIB") and build_id() != source_build_id():
        pytest.fail(f"{p} was built from other sources (ort_build_id {build_id()} != {source_build_id()})")
    return p


CONFIGS = {
    # name: Settings overrides.  cfg1/cfg2 are BASELINE.json configs[0]/[1].
    "large": dict(bottle_file="clearBottle-large.params"),
    "small": dict(bottle_file="clearBottle-small.params"),
    "large_iris_before": dict(bottle_file="clearBottle-large.params", iris="before", iris_size=0.8),
    "small_iris_after": dict(bottle_file="clearBottle-small.params", iris="after", iris_size=0.8),
    "ellipse": dict(bottle_file="clearBottle-ellipse.params"),
    "small_f60_nobottle": dict(bottle_file="clearBottle-small.params", use_bottle=False,
                               L3_file="achromaticDoublet-f60.0mm.params",
                               L2_file="planoConvex-f49.8mm.params", fibre_offset=1e-3),
    # SURVEY §8 f2 emitters: spot (create_spot, runner.py -s uses 100 rays) and crs (point_on_bottle)
    "small_spot": dict(bottle_file="clearBottle-small.params", light_source="spot", nphotons=100),
    "large_crs": dict(bottle_file="clearBottle-large.params", light_source="crs", crs_spot_size=1e-3),
    # isors (iSORS: Gaussian beam through an axicon onto the bottle, src/sourceMod.f90:162-247)
    "small_isors": dict(bottle_file="clearBottle-small.params", light_source="isors", isors_offset=0.5e-3),
    # image source (emit_image): a synthetic integer-valued 512 x 512 source image, see source_image()
    "large_image": dict(bottle_file="clearBottle-large.params", light_source="image",
                        image_source="synthetic-source.dat", nphotons=60000),
    # in-bottle scattering (no shipped bottle scatters; coefficients chosen so that a
    # ray scatters a few times: contents mua 20 / mus 150, wall mua 30 / mus 400 [1/m])
    "small_scatter_c": dict(bottle_file="scatterBottle-contents.params"),
    "small_scatter_bc": dict(bottle_file="scatterBottle-both.params"),
}

SCATTER_BOTTLES = {"scatterBottle-contents.params": [0.0, 0.0, 20.0, 150.0],
                   "scatterBottle-both.params": [10.0, 300.0, 5.0, 80.0]}


def isors_safe_uniforms(u, rng):
    """Make a uniform table safe for the UNMODIFIED iSORS of the reference: a ray that reflects at
    the axicon runs into `error stop "no intersection with bottle!"` (src/sourceMod.f90:216-218),
    which would end the test process.  rang's first pair is put inside the unit disc (so the axicon
    draw is draw 2) and draw 2 is kept above 0.5 (the axicon's reflectance is ~0.03): every ray
    refracts.  The reflecting rays are exercised oracle-vs-HIP only (the reference has no
    behaviour to offer for them)."""
    import numpy as np
    n = u.shape[1]
    r, th = np.sqrt(rng.random(n)) * 0.999, rng.random(n) * 2 * np.pi
    u[0], u[1] = (r * np.cos(th) + 1) / 2, (r * np.sin(th) + 1) / 2
    u[2] = 0.5 + 0.5 * rng.random(n)
    return u


def source_image():
    """Deterministic stand-in for bpm.py's Bessel image (the reference ships none): an
    integer-valued ring + core, so that sum(image) is exact in any summation order."""
    import numpy as np
    y, x = np.mgrid[0:512, 0:512]
    r = np.hypot(x - 255.5, (y - 255.5) * 1.1)
    img = np.floor(40.0 * np.exp(-((r - 90.0) / 14.0) ** 2) + 25.0 * np.exp(-(r / 9.0) ** 2)
                   + ((x * 7 + y * 13) % 5 == 0))
    return img.astype(np.float64)


def res_dir_with_image(src_res: str) -> str:
    """A res/ directory = the .params files of `src_res` + synthetic-source.dat (the image
    source file lives next to the settings' other files, setupMod.f90:120-121)."""
    import shutil
    import tempfile
    d = os.path.join(tempfile.gettempdir(), "ort_res_" + str(abs(hash(src_res)) % 100000) + "_" + str(os.getpid()))
    if not os.path.exists(os.path.join(d, "synthetic-source.dat")):
        os.makedirs(d, exist_ok=True)
        for f in os.listdir(src_res):
            if f.endswith(".params"):
                shutil.copy(os.path.join(src_res, f), os.path.join(d, f))
        base = open(os.path.join(d, "clearBottle-small.params")).read().splitlines()[:12]
        for name, mu in SCATTER_BOTTLES.items():            # 16-value bottle files (lens.f90:195-208)
            with open(os.path.join(d, name), "w") as f:
                f.write("\n".join(base + [repr(m) for m in mu]) + "\n")
        # ... and an elliptical bottle that scatters (no golden fixture: HIP kernel variants and the oracle only)
        base = open(os.path.join(d, "clearBottle-ellipse.params")).read().splitlines()[:12]
        with open(os.path.join(d, "scatterBottle-ellipse.params"), "w") as f:
            f.write("\n".join(base + [repr(m) for m in SCATTER_BOTTLES["scatterBottle-both.params"]]) + "\n")
        source_image().tofile(os.path.join(d, "synthetic-source.dat"))
    return d


def needs_extended_res(settings) -> bool:
    return settings.light_source == "image" or settings.bottle_file in SCATTER_BOTTLES or settings.bottle_file == "scatterBottle-ellipse.params"


def make_system(name: str, **over):
    """The system of CONFIGS[name], `over` = further Settings overrides (e.g. an iris in front of another light source)."""
    from opticalraytrace_amd.params import Settings, resource_dir
    from opticalraytrace_amd.system import OpticalSystem
    s = Settings(**{**dict(nphotons=100000, make_images=True), **CONFIGS[name], **over})
    res = res_dir_with_image(resource_dir()) if needs_extended_res(s) else None
    return s, OpticalSystem.from_settings(s, res)


@pytest.fixture(scope="session", params=list(CONFIGS))
def config_name(request):
    return request.param


def pytest_report_header(config):
    """Which libm the session's oracle is built on (oracle/binding.py: the host's where it is glibc 2.35's — the library
    the device code reproduces — else the pinned build on csrc/ort_libm.h compiled for the host)."""
    try:
        from oracle.binding import host_libm_mismatches, oracle_libm
        bad = host_libm_mismatches()
        return (f"oracle libm: {oracle_libm()}" + (f" (host libm differs from glibc 2.35 in {bad})" if bad else
                                                   " (host libm == glibc 2.35 x86-64 FMA variants: oracle == oracle/_ref == device)"))
    except Exception as e:                       # the header never fails a session
        return f"oracle libm: unknown ({e!r})"
