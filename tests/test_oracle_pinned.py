"""The pinned-libm build of the oracle (oracle/libort_oracle_pinned.so: ort_oracle.c on oracle/pinned_libm.cpp =
csrc/ort_libm.h compiled for the host) — the checker oracle/binding.py selects on a machine whose own libm is not glibc
2.35.  CPU only.  (1) it reproduces the committed golden vectors of the reference on ANY machine: the fixtures came out
of the reference on glibc 2.35, which is what the pinned build restates; (2) where the host's libm IS the pinned one it
equals the default build bit for bit on every light source and on the scattering walk."""
import numpy as np
import pytest

from conftest import CONFIGS, make_system
from parity import SEED, assert_rays_equal, load_golden
from oracle.binding import Oracle, host_libm_is_pinned


@pytest.mark.parametrize("name", list(CONFIGS))
def test_pinned_oracle_matches_the_golden_rays(name):
    g = load_golden(name)
    _, osys = make_system(name)
    orc = Oracle(osys, libm="pinned")
    assert orc.pinned_libm
    for phase in (1, 2):
        u = g[f"p{phase}_u"]
        got = orc.trace_rays(phase, u.shape[1], u=u)
        want = dict(status=g[f"p{phase}_status"], bin_xy=g[f"p{phase}_bin"], n_draws=g[f"p{phase}_ndraws"], pos_dir=g[f"p{phase}_pos_dir"])
        assert np.array_equal(got["emitted"], g[f"p{phase}_emitted"]), f"{name} phase {phase}: emitted rays not bit-exact"
        assert_rays_equal(got, want, exact=True, what=f"{name} phase {phase} (pinned libm)")


@pytest.mark.parametrize("name", list(CONFIGS))
def test_pinned_oracle_equals_the_host_libm_oracle(name):
    if not host_libm_is_pinned():
        pytest.skip("the host's libm is not glibc 2.35's: the two builds are MEANT to differ here")
    settings, osys = make_system(name)
    a, b = Oracle(osys, libm="host"), Oracle(osys, libm="pinned")
    assert not a.pinned_libm and b.pinned_libm
    n = min(settings.nphotons, 20000)
    for phase in (1, 2):
        if settings.light_source == "image" and phase == 2:
            n = min(n, 4000)
        ra, rb = a.trace_rays(phase, n, seed=SEED), b.trace_rays(phase, n, seed=SEED)       # keyed draws, in-oracle emission
        for k in ra:
            assert np.array_equal(ra[k], rb[k]), f"{name} phase {phase}: {k}"
    ia, ca = a.trace(2, 0, n, SEED)
    ib, cb = b.trace(2, 0, n, SEED)
    assert np.array_equal(ia, ib) and np.array_equal(ca, cb)
