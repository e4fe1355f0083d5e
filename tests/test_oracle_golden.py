"""The C oracle against the committed golden vectors (generated from the reference
itself by tests/golden/make_golden.py).  CPU only; bit-exact."""
import numpy as np
import pytest

from conftest import CONFIGS, make_system
from parity import emit_draws, SEED, assert_rays_equal, load_golden, merge_status, sparse_image
from oracle.binding import Oracle


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("phase", [1, 2])
def test_oracle_matches_golden_rays(name, phase):
    g = load_golden(name)
    settings, osys = make_system(name)
    orc = Oracle(osys)
    u = g[f"p{phase}_u"]
    n = u.shape[1]
    got = orc.trace_rays(phase, n, u=u)
    want = dict(status=g[f"p{phase}_status"], bin_xy=g[f"p{phase}_bin"],
                n_draws=g[f"p{phase}_ndraws"], pos_dir=g[f"p{phase}_pos_dir"])
    assert np.array_equal(got["emitted"], g[f"p{phase}_emitted"]), "emitted rays not bit-exact"
    assert_rays_equal(got, want, exact=True, what=f"{name} phase {phase}")
    # explicit-input variant (no emitter on the path)
    base = emit_draws(settings, phase)
    gotx = orc.trace_rays(phase, n, pos_dir_in=g[f"p{phase}_emitted"], u=u, draw_base=base)
    wantx = dict(status=g[f"p{phase}x_status"], bin_xy=g[f"p{phase}x_bin"],
                 n_draws=g[f"p{phase}x_ndraws"], pos_dir=g[f"p{phase}x_pos_dir"])
    assert_rays_equal(gotx, wantx, exact=True, what=f"{name} phase {phase} explicit")
    # both Fresnel branches were forced in the fixture
    if name not in ("ellipse",) and settings.light_source == "point":
        st = merge_status(g[f"p{phase}_status"])
        assert (st[:8] != 0).all(), "u=0 rays must reflect somewhere and be lost or rejected"


@pytest.mark.parametrize("name", list(CONFIGS))
def test_oracle_matches_golden_images(name):
    g = load_golden(name)
    settings, osys = make_system(name)
    orc = Oracle(osys)
    img = np.zeros((2, 401, 401), np.int32)
    cnt = np.zeros(8, np.uint64)
    for phase in (1, 2):
        if settings.light_source == "image" and phase == 2:
            # sequential source: the fixture holds the image of table-mode rays (make_golden.py)
            n = settings.nphotons
            ub = np.random.default_rng(int(g["img2_u_seed"])).random((10, n))
            rb = orc.trace_rays(2, n, u=ub)
            ok = rb["status"] == 0
            np.add.at(img[1], (rb["bin_xy"][1][ok] + 200, rb["bin_xy"][0][ok] + 200), 1)
            cnt[1] += np.uint64((rb["status"] >= 3).sum())
            cnt[5] += np.uint64(ok.sum())
            continue
        if settings.light_source == "isors" and phase == 1:
            # the fixture holds table-mode rays that all refract at the axicon (make_golden.py)
            from conftest import isors_safe_uniforms
            n = settings.nphotons
            rs = np.random.default_rng(int(g["img1_u_seed"]))
            ub = isors_safe_uniforms(rs.random((48, n)), rs)
            rb = orc.trace_rays(1, n, u=ub)
            ok = rb["status"] == 0
            np.add.at(img[0], (rb["bin_xy"][1][ok] + 200, rb["bin_xy"][0][ok] + 200), 1)
            cnt[0] += np.uint64((rb["status"] >= 3).sum())
            cnt[4] += np.uint64(ok.sum())
            continue
        orc.trace(phase, 0, settings.nphotons, SEED, img, cnt)
    want = sparse_image(g["img1_idx"], g["img1_cnt"]) + sparse_image(g["img2_idx"], g["img2_cnt"])
    assert np.array_equal(img, want)
    assert int(cnt[0]) == int(g["img1_lost"]) and int(cnt[1]) == int(g["img2_lost"])
    assert int(cnt[4]) == int(img[0].sum()) and int(cnt[5]) == int(img[1].sum())


def test_rng_known_answers():
    """ORT-RNG-v2 pinned numerically (SplitMix64 finaliser, one hash per pair of draws; independent
    Python restatement), and the host's own numpy restatement (opticalraytrace_amd/rng.py)."""
    M = (1 << 64) - 1
    G = 0x9E3779B97F4A7C15

    def mix(z):
        z ^= z >> 30; z = (z * 0xBF58476D1CE4E5B9) & M
        z ^= z >> 27; z = (z * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    def uni(seed, phase, ray, k):
        base = mix(seed ^ ((G * phase) & M))
        c = ((ray << 24) + k) & M
        h = mix((base + G * ((c >> 1) + 1)) & M)
        return ((h & 0xFFFFFFFF) if (c & 1) else (h >> 32)) * 2.0 ** -32

    orc = Oracle()
    for seed, phase, ray, k in [(SEED, 1, 0, 0), (SEED, 2, 12345678901, 8), (0, 2, 2 ** 31 - 1, 3),
                                (2 ** 63 + 5, 1, 999, 15)]:
        assert orc.uniform(seed, phase, ray, k) == uni(seed, phase, ray, k)
        from opticalraytrace_amd.rng import uniforms
        assert float(uniforms(seed, phase, ray, [k])[0]) == uni(seed, phase, ray, k)
    us = np.array([orc.uniform(SEED, 2, i, i % 9) for i in range(20000)])
    assert 0.0 <= us.min() and us.max() < 1.0
    assert abs(us.mean() - 0.5) < 0.01 and abs(us.var() - 1 / 12) < 0.005


def test_oracle_partition_invariance():
    """Images add over any split of the global ray index range."""
    _, osys = make_system("small")
    orc = Oracle(osys)
    a, ca = orc.trace(2, 0, 30000, SEED)
    b = np.zeros_like(a); cb = np.zeros(8, np.uint64)
    for lo, n in [(0, 1), (1, 9999), (10000, 20000)]:
        orc.trace(2, lo, n, SEED, b, cb)
    assert np.array_equal(a, b) and np.array_equal(ca, cb)


def test_image_source_histogram_matches_reference():
    """init_emit_image: the host's histogram (product code), the oracle's and the compiled
    reference's imgin agree cell for cell (same keyed rounding draws)."""
    from opticalraytrace_amd.image_source import cdf, histogram, load_image
    g = load_golden("large_image")
    settings, osys = make_system("large_image")
    counts = histogram(load_image(osys.image_source_path), settings.nphotons, osys.image_seed)
    orc = Oracle(osys)
    assert np.array_equal(counts, orc._counts)
    # the fixture holds the Fortran imgin(512,512) as the C-order view [second][first] = scan order
    assert np.array_equal(counts.reshape(512, 512), g["imgin_counts"])
    c = cdf(counts)
    assert c[0] == 0 and c[-1] == counts.sum() and abs(int(c[-1]) - settings.nphotons) < 600
