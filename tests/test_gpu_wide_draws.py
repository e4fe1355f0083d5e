"""Kernel variant bit 5: the 53-bit draw stream ORT-RNG-v2w (one hash per draw, u = (h >> 11) * 2^-53) — what the
reference's ran2() holds in its real(8) (src/random_mod.f90:39-46: random_number fills the 53 bits of a double), where
the default stream hands out 32-bit draws.  The stream has three statements (csrc/ort_device.h, oracle/ort_oracle.c,
opticalraytrace_amd/rng.py); they must agree draw for draw, the HIP path on it must equal the CPU checker on it bit for
bit, and its statistics must be those of the UNMODIFIED reference program (tests/golden/refprog_*.npz)."""
import numpy as np
import pytest

from conftest import make_system
from parity import SEED, load_golden, merge_status, rel_err
from stats_util import assert_same_distribution

WIDE = 1 | 32                 # default kernels + 53-bit draws: the WIDE instantiations of the queued kernels (csrc/ort_k_wide.hip)
WIDE_STRICT = 1 | 32 | 64     # ... + the emitters through glibc's own sin / cos: bit for bit against the checker


@pytest.fixture()
def wide_oracle():
    """The checker with its keyed mode on the 53-bit stream; the switch is process-wide, so it is put back."""
    from oracle.binding import Oracle
    made = []

    def get(osys):
        o = Oracle(osys)
        o.set_wide_draws(True)
        made.append(o)
        return o

    yield get
    for o in made:
        o.set_wide_draws(False)


def test_the_two_host_statements_of_the_stream_agree(wide_oracle):
    from opticalraytrace_amd.rng import uniforms
    _, osys = make_system("small")
    orc = wide_oracle(osys)
    k = np.arange(64)
    for seed, phase, ray in ((SEED, 1, 0), (SEED, 2, 12345), (7, 2, (1 << 40) - 1)):
        want = np.array([orc.uniform(seed, phase, ray, int(i)) for i in k])
        got = uniforms(seed, phase, ray, k, wide=True)
        assert np.array_equal(got, want)
        assert np.all((got >= 0) & (got < 1))
        m = got * 2.0 ** 53
        assert np.array_equal(m, np.floor(m))                      # multiples of 2^-53 ...
        assert np.any(np.mod(m, 2.0 ** 21) != 0)                   # ... that are not multiples of 2^-32
    orc.set_wide_draws(False)
    v2 = np.array([orc.uniform(SEED, 1, 0, int(i)) for i in k])
    assert np.array_equal(v2, uniforms(SEED, 1, 0, k))             # the default stream is untouched
    assert np.array_equal(v2 * 2.0 ** 32, np.floor(v2 * 2.0 ** 32))


@pytest.mark.parametrize("name", ["large", "small"])
def test_checker_statistics_on_the_wide_stream_match_reference_program(wide_oracle, name):
    g = load_golden("refprog_" + name)
    n_ref = int(g["nphotons"])
    ring = np.zeros(401 * 401, np.int64); ring[g["ring_idx"]] = g["ring_cnt"]
    point = np.zeros(401 * 401, np.int64); point[g["point_idx"]] = g["point_cnt"]
    _, o = make_system(name)
    orc = wide_oracle(o)
    n = 1_000_000
    img = np.zeros((2, 401, 401), np.int32); cnt = np.zeros(8, np.uint64)
    orc.trace(1, 0, n, SEED, img, cnt); orc.trace(2, 0, n, SEED, img, cnt)
    assert_same_distribution(img[0], ring.reshape(401, 401), n, n_ref, f"{name} ring")
    assert_same_distribution(img[1], point.reshape(401, 401), n, n_ref, f"{name} point")


_SYSTEMS = ["large", "small", "large_iris_before", "ellipse", "large_crs", "small_isors", "large_image", "small_spot",
            "small_scatter_bc"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", _SYSTEMS)
def test_keyed_rays_on_the_wide_stream_equal_the_checker_bit_for_bit(hip_library, wide_oracle, name):
    """The parity entry with u == NULL: every ray's emitted state, final state, status, bin and DRAW COUNT."""
    from opticalraytrace_amd.capi import Context
    _, osys = make_system(name)
    orc = wide_oracle(osys)
    n = 20_000
    with Context(osys) as ctx:
        ctx.set_kernel_variant(WIDE_STRICT)
        for phase in (1, 2):
            got = ctx.trace_rays(phase, n, seed=SEED, first_ray=1000)
            want = orc.trace_rays(phase, n, seed=SEED, first_ray=1000)
            assert np.array_equal(merge_status(got["status"]), merge_status(want["status"])), (name, phase)
            assert np.array_equal(got["n_draws"], want["n_draws"]), (name, phase)
            assert np.array_equal(got["emitted"], want["emitted"]), (name, phase, rel_err(got["emitted"], want["emitted"]))
            reach = merge_status(want["status"]) <= 1
            assert np.array_equal(got["pos_dir"][:, reach], want["pos_dir"][:, reach]), (name, phase)
            b = merge_status(want["status"]) == 0
            assert np.array_equal(got["bin_xy"][:, b], want["bin_xy"][:, b])
            first = got if phase == 1 else first
        # and it IS another stream: the default one gives other rays
        # (phase 1: every source draws there; the spot source's phase 2 is sequential)
        ctx.set_kernel_variant(1 | 64)
        other = ctx.trace_rays(1, n, seed=SEED, first_ray=1000)
        assert not np.array_equal(other["emitted"], first["emitted"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", _SYSTEMS)
def test_image_on_the_wide_stream_equals_the_checkers(hip_library, wide_oracle, name):
    from opticalraytrace_amd.capi import Context
    _, osys = make_system(name)
    orc = wide_oracle(osys)
    n = 200_003 if osys.settings.light_source != "image" else 50_000
    with Context(osys) as ctx:
        names = {}
        for phase in (1, 2):
            ctx.trace(phase, 5, n, SEED)
            names[phase] = ctx.last_kernel_name()
        ctx.reset()
        ctx.set_kernel_variant(WIDE_STRICT)
        for phase in (1, 2):
            ctx.trace(phase, 5, n, SEED)
            # the stream is a template flag of the production kernels: a surface program stays a surface program, the
            # scattering pipeline stays the pipeline (round 4 sent every launch to the lockstep kernel)
            kname = ctx.last_kernel_name()
            if "strict=0, wide=0" in names[phase]:
                assert kname == names[phase].replace("strict=0, wide=0", "strict=1, wide=1"), (names[phase], kname)
            else:
                assert ("scatter_front_kernel" in kname) == ("scatter_front_kernel" in names[phase]) and "trace_kernel<" not in kname, (names[phase], kname)
        img, cnt = ctx.read()
        wimg = np.zeros((2, 401, 401), np.int32); wc = np.zeros(8, np.uint64)
        for phase in (1, 2):
            orc.trace(phase, 5, n, SEED, wimg, wc)
        assert np.array_equal(img, wimg), int(np.abs(img.astype(np.int64) - wimg).sum())
        assert np.array_equal(cnt, wc), (cnt, wc)
        # the default emitters on the same stream: the usual budget of a ray or two on a bin edge
        ctx.set_kernel_variant(WIDE)
        ctx.reset()
        for phase in (1, 2):
            ctx.trace(phase, 5, n, SEED)
        img2, cnt2 = ctx.read()
        assert int(np.abs(img2.astype(np.int64) - wimg).sum()) <= 4
        assert np.array_equal(cnt2[:2], wc[:2])


@pytest.mark.gpu
def test_resident_bundle_on_the_wide_stream(hip_library):
    """ort_emit + ort_trace_resident continue the ray's own draws: same image as the fused trace."""
    import torch
    from opticalraytrace_amd.capi import Context
    _, osys = make_system("large")
    n = 100_000
    with Context(osys) as ctx:
        ctx.set_kernel_variant(WIDE)
        for phase in (1, 2):
            ctx.trace(phase, 0, n, SEED)
        img, cnt = ctx.read()
        ctx.reset()
        buf = torch.empty((6, n), dtype=torch.float64, device="cuda")
        for phase, base in ((1, 4), (2, 2)):
            ctx.emit(phase, 0, n, SEED, buf.data_ptr())
            ctx.trace_resident(phase, 0, n, SEED, base, buf.data_ptr())
        img2, cnt2 = ctx.read()
    assert np.array_equal(img, img2)
    assert np.array_equal(cnt, cnt2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["large", "small"])
def test_wide_stream_statistics_match_the_unmodified_program(hip_library, name):
    """4e6 rays per layer on the 53-bit stream against the unmodified reference program's own images."""
    from opticalraytrace_amd.capi import Context
    g = load_golden("refprog_" + name)
    n_ref = int(g["nphotons"])
    ring = np.zeros(401 * 401, np.int64); ring[g["ring_idx"]] = g["ring_cnt"]
    point = np.zeros(401 * 401, np.int64); point[g["point_idx"]] = g["point_cnt"]
    _, o = make_system(name)
    n = 4_000_000
    with Context(o) as ctx:
        ctx.set_kernel_variant(WIDE)
        ctx.trace(1, 0, n, SEED + 2)
        ctx.trace(2, 0, n, SEED + 2)
        img, cnt = ctx.read()
        # the 32-bit stream over the same rays: the two are statistically the same experiment
        ctx.set_kernel_variant(1)
        ctx.reset()
        ctx.trace(1, 0, n, SEED + 2)
        ctx.trace(2, 0, n, SEED + 2)
        img32, cnt32 = ctx.read()
    assert_same_distribution(img[0], ring.reshape(401, 401), n, n_ref, f"{name} ring")
    assert_same_distribution(img[1], point.reshape(401, 401), n, n_ref, f"{name} point")
    out = str(g["stdout"]).split()
    ring_t, point_t = float(out[out.index("Ring") + 2].rstrip("%")), float(out[out.index("Point") + 2].rstrip("%"))
    assert abs(100 * (1 - int(cnt[0]) / n) - ring_t) < 0.1
    assert abs(100 * (1 - int(cnt[1]) / n) - point_t) < 0.3
    assert_same_distribution(img[0], img32[0], n, n, f"{name} ring, v2w vs v2")
    assert_same_distribution(img[1], img32[1], n, n, f"{name} point, v2w vs v2")
    assert not np.array_equal(img, img32)


@pytest.mark.gpu
def test_fp32_path_on_the_wide_stream(hip_library):
    """fp32 takes the top 24 bits of the same hashes: the images of the two precisions stay as close as on v2."""
    from opticalraytrace_amd.capi import Context
    _, o = make_system("large")
    n = 1_000_000
    with Context(o) as ctx:
        ctx.set_kernel_variant(WIDE)
        ctx.trace(2, 0, n, SEED)
        img64, cnt64 = ctx.read()
        ctx.set_precision(1)
        ctx.reset()
        ctx.trace(2, 0, n, SEED)
        img32, cnt32 = ctx.read()
    assert abs(int(cnt32[5]) - int(cnt64[5])) < 2e-3 * n
    assert int(np.abs(img32.astype(np.int64) - img64).sum()) < 2e-2 * int(cnt64[5])
