"""The C oracle against the REFERENCE ITSELF (oracle/_ref/libort_ref.so: the reference's own
Fortran path sources compiled with flang, see oracle/Makefile) on fresh random rays — beyond
the committed fixtures.  Skipped where the library has not been built (it needs
/root/reference at build time; the prebuilt .so travels with the repository snapshot).
Nothing here reads /root/reference at run time: the packaged .params files are used."""
import numpy as np
import pytest

from conftest import CONFIGS, make_system, needs_extended_res, res_dir_with_image
from parity import emit_draws, merge_status
from oracle.binding import Oracle, Reference, reference_available
from opticalraytrace_amd.params import resource_dir

pytestmark = pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libort_ref.so not built")


@pytest.mark.parametrize("name", list(CONFIGS))
def test_oracle_equals_reference_bit_for_bit(name):
    settings, osys = make_system(name)
    res = res_dir_with_image(resource_dir()) if needs_extended_res(settings) else resource_dir()
    ref = Reference(settings, res)
    orc = Oracle(osys)
    n = min(settings.nphotons, 30000)
    u = np.random.default_rng(sum(map(ord, name))).random((160, n))
    for phase in (1, 2):
        a = orc.trace_rays(phase, n, u=u)
        b = ref.trace_rays(phase, n, u=u)
        assert np.array_equal(merge_status(a["status"]), b["status"]), (name, phase)
        assert np.array_equal(a["emitted"], b["emitted"]), (name, phase)
        assert np.array_equal(a["n_draws"], b["n_draws"]), (name, phase)
        reach = b["status"] <= 1
        assert np.array_equal(a["pos_dir"][:, reach], b["pos_dir"][:, reach]), (name, phase)
        binned = b["status"] == 0
        assert np.array_equal(a["bin_xy"][:, binned], b["bin_xy"][:, binned]), (name, phase)
        # explicit-input variant: the emitted rays fed back, draws from a fixed slot
        base = emit_draws(settings, phase)
        ax = orc.trace_rays(phase, n, pos_dir_in=b["emitted"], u=u, draw_base=base)
        bx = ref.trace_rays(phase, n, pos_dir_in=b["emitted"], u=u, draw_base=base)
        assert np.array_equal(merge_status(ax["status"]), bx["status"])
        r2 = bx["status"] <= 1
        assert np.array_equal(ax["pos_dir"][:, r2], bx["pos_dir"][:, r2])


def test_reference_constants_match_host_model():
    settings, osys = make_system("large")
    c = Reference(settings, resource_dir()).constants()
    assert osys.cos_theta_max == c[33] and osys.r1 == c[34] and osys.r2 == c[35]
    assert osys.img_plane == c[36] and osys.na_angle == c[39] and osys.bin_width == c[40]
    assert osys.L3[1].n3 == c[22] and osys.bottle.ncontents == c[1]
