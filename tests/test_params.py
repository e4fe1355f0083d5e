"""Host logic: the reference's text formats, dispersion formulas and derived run constants."""
import math
import os

import numpy as np
import pytest

from conftest import CONFIGS, make_system
from parity import load_golden
from opticalraytrace_amd.params import (AchromaticDoublet, GlassBottle, ParamsError, PlanoConvex,
                                        Settings, first_tokens, parse_logical, parse_real, resource_dir,
                                        sellmeier)
from opticalraytrace_amd.system import OpticalSystem, acos_threshold

RES = resource_dir()


def test_fortran_reals_and_logicals():
    assert parse_real("785d-9") == 785e-9
    assert parse_real("1.d-2") == 1e-2
    assert parse_real("5") == 5.0 and parse_real("0.") == 0.0 and parse_real("1.0D0") == 1.0
    assert parse_logical("true") and parse_logical(".TRUE.") and parse_logical("T")
    assert not parse_logical("false") and not parse_logical(".f.")
    with pytest.raises(ParamsError):
        parse_real("bessel.dat")


def test_known_answer_constants_from_the_survey():
    """SURVEY §8(a) pins (values produced by the flang-built reference)."""
    wl1, wl2 = 785e-9, 843e-9
    p1 = PlanoConvex.from_file(os.path.join(RES, "planoConvex-f39.9mm.params"), wl1)
    p2 = PlanoConvex.from_file(os.path.join(RES, "planoConvex-f39.9mm.params"), wl2)
    assert p1.n2 == 1.51107956490822759 and p2.n2 == 1.50996498599669438
    assert p1.centre_z == 2.15000000000000052e-02 and p1.flat_z == 3.57000000000000026e-02
    assert p1.radius == 1.26999999999999995e-02
    d1 = AchromaticDoublet.from_file(os.path.join(RES, "achromaticDoublet-f50.0mm.params"), wl1,
                                     2.0 * p1.fb + p1.thickness)
    d2 = AchromaticDoublet.from_file(os.path.join(RES, "achromaticDoublet-f50.0mm.params"), wl2,
                                     2.0 * p1.fb + p1.thickness)
    assert d1.n2 == 1.64311133601352610 and d2.n2 == 1.64162212436063437
    assert d1.n3 == 1.78533573103620524 and d2.n3 == 1.78202673555623159
    assert (d1.centre1_z, d1.centre2_z, d1.centre3_z) == (1.56350000000000017e-01, 1.03249999999999995e-01,
                                                          6.50000000000000577e-03)
    b = GlassBottle.from_file(os.path.join(RES, "clearBottle-large.params"), wl1)
    assert b.nbottle == 1.51747665273036514 and b.ncontents == 1.35766838724121075
    s = Settings(bottle_file="clearBottle-large.params")
    o = OpticalSystem.from_settings(s)
    assert o.cos_theta_max == 9.42159141664439037e-01
    assert o.img_plane == 1.77099999999999980e-01
    assert o.bessel_diameter == 3.53379844792511008e-03
    assert (o.r1, o.r2) == (9.20393302263280711e-06, 3.12193286763947941e-06)
    assert o.bin_width == 2.49376558603491289e-05
    assert o.na_angle == 2.21814470496794425e-01
    o = OpticalSystem.from_settings(Settings(bottle_file="clearBottle-small.params"))
    assert o.bessel_diameter == 1.87398402541483124e-03
    assert (o.r1, o.r2) == (1.88783210209514354e-06, 8.77954031877493742e-07)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_derived_constants_match_reference_fixture(name):
    """Every constant the reference's constructors / set-up lines produced (golden `constants`)."""
    c = load_golden(name)["constants"]
    _, o = make_system(name)
    b, l2a, l2b, l3a, l3b = o.bottle, o.L2[0], o.L2[1], o.L3[0], o.L3[1]
    got = [b.nbottle, b.ncontents, b.thickness, b.radiusa, b.radiusb, b.centre[0], b.centre[1],
           b.centre[2], float(b.ellipse), l2a.n1, l2a.n2, l2b.n2, l2a.centre_z, l2a.curve_radius,
           l2a.thickness, l2a.radius, l2a.fb, l2a.f, l3a.n1, l3a.n2, l3a.n3, l3b.n2, l3b.n3,
           l3a.centre1_z, l3a.centre2_z, l3a.centre3_z, l3a.R1, l3a.R2, l3a.R3, l3a.radius, l3a.fb,
           l3a.f, l3a.thickness, o.cos_theta_max, o.r1, o.r2, o.img_plane, o.bessel_diameter,
           o.distance, o.na_angle, o.bin_width, math.pi, l3b.centre1_z, l3b.centre2_z, l3b.centre3_z,
           l2b.centre_z]
    assert np.array_equal(np.array(got), c[:46]), np.nonzero(np.array(got) != c[:46])
    if len(c) > 46:
        assert o.crs_spot_size == c[46]                      # setupMod.f90:136


def test_bottle_line_count_rule(tmp_path):
    """12 values or >= 16; 13-15 abort in the reference (src/lens.f90:195-208)."""
    base = open(os.path.join(RES, "clearBottle-small.params")).read().splitlines()
    p = tmp_path / "b14.params"
    p.write_text("\n".join(base + ["0.   mua", "0.0   mus"]) + "\n")
    with pytest.raises(ParamsError):
        GlassBottle.from_file(str(p), 785e-9)
    p16 = tmp_path / "b16.params"
    p16.write_text("\n".join(base + ["0.", "0.", "0.", "0."]) + "\n")
    assert not GlassBottle.from_file(str(p16), 785e-9).scatters
    ps = tmp_path / "bs.params"
    ps.write_text("\n".join(base + ["1.", "10.", "0.", "0."]) + "\n")
    assert GlassBottle.from_file(str(ps), 785e-9).scatters
    short = tmp_path / "short.params"
    short.write_text("\n".join(base[:11]) + "\n")
    with pytest.raises(ParamsError):
        GlassBottle.from_file(str(short), 785e-9)


def test_settings_round_trip_in_runner_layout(tmp_path):
    s = Settings(nphotons=12345, make_images=True, iris="before", iris_size=0.75,
                 bottle_file="clearBottle-small.params", fibre_offset=-1e-3)
    f = tmp_path / "test_0.params"
    s.write(str(f))
    lines = f.read_text().splitlines()
    assert len(lines) == 20
    assert lines[1].startswith("7.85d-07") and lines[1][35:] == "# wavelength"   # runner.py:99-108
    assert all(ln[35:37] == "# " for ln in lines)
    assert Settings.from_file(str(f)) == s
    # the shipped style: comments without '#', logicals as words, d exponents
    g = tmp_path / "hand.params"
    g.write_text("0.5d-3   ring\n785d-9 wl\n100 n\n5 alpha\n1.45 ax\ntrue b\nfalse t\ntrue img\n1.d-2 d\n"
                 "0.0 fo\npoint src\nnone iris\n1.0 sz\nclearBottle-small.params b\n"
                 "planoConvex-f39.9mm.params l2\nachromaticDoublet-f40.0mm.params l3\nbessel-normal.dat\n"
                 "settings-test folder\n1.5d-3 iso\n1.d-3 crs")
    h = Settings.from_file(str(g))
    assert h.nphotons == 100 and h.wavelength == 785e-9 and h.L3_file == "achromaticDoublet-f40.0mm.params"
    bad = Settings(light_source="laser")
    with pytest.raises(ParamsError):
        bad.validate()
    with pytest.raises(ParamsError):
        Settings(iris="middle").validate()
    with pytest.raises(ParamsError):
        Settings(nphotons=20000, use_tracker=True).validate()


def test_every_source_of_the_reference_builds_and_unknown_ones_fail_loudly():
    """setupMod.f90:85-99: image, spot, point, isors, crs — anything else is `error stop "No such
    source type!"`, here a ParamsError.  isors changes `distance` (main.f90:60-64)."""
    with pytest.raises(ParamsError):
        OpticalSystem.from_settings(Settings(light_source="laser"))
    for src in ("point", "spot", "crs", "image", "isors"):
        OpticalSystem.from_settings(Settings(light_source=src, nphotons=100))
    a = OpticalSystem.from_settings(Settings(light_source="isors", isors_offset=0.5e-3))
    b = OpticalSystem.from_settings(Settings(light_source="point", isors_offset=0.5e-3))
    assert a.distance == a.bottle.radiusa + 0.5e-3 and b.distance == b.bottle.radiusa + b.bottle.centre[2]
    assert a.r1 != b.r1


def test_bottle_clamp():
    """src/main.f90:54-58: a bottle that would touch the lens is moved back."""
    o = OpticalSystem.from_settings(Settings(bottle_file="clearBottle-small_17.5mm.params",
                                             L2_file="planoConvex-f29.9mm.params"))
    l2 = o.L2[0]
    if o.bottle_moved:
        assert o.bottle.centre[2] == l2.fb - o.bottle.radiusa - 2e-3
    assert l2.fb > o.bottle.radiusa + o.bottle.centre[2]


def test_surface_lists():
    _, o = make_system("large")
    assert [s.name for s in o.surfaces(1)] == ["L2 flat", "L2 curved", "L3 face 1", "L3 face 2",
                                               "L3 face 3", "image plane"]
    assert len(o.surfaces(2)) == 8                       # max 8 per point ray, 6 per ring ray
    _, o = make_system("large_iris_before")
    assert [s.name for s in o.surfaces(2)][4] == "iris before" and len(o.surfaces(2)) == 9
    _, o = make_system("small_f60_nobottle")
    assert len(o.surfaces(2)) == 6


def test_na_threshold_is_equivalent_to_acos_test():
    na = math.asin(0.22)
    thr = acos_threshold(na)
    assert math.acos(thr) <= na < math.acos(np.nextafter(thr, 0.0))
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(0.9, 1.0, 20000), thr + np.arange(-50, 50) * 2.0 ** -53])
    for x in xs:
        assert (math.acos(x) > na) == (x < thr)


def test_packaged_params_equal_reference_values():
    """opticalraytrace_amd/res/*.params carry the reference's numbers (checked when it is present)."""
    ref = "/root/reference/res"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present on this machine")
    for f in sorted(os.listdir(RES)):
        a = [parse_real(t) for t in first_tokens(os.path.join(RES, f))]
        b = [parse_real(t) for t in first_tokens(os.path.join(ref, f))[:len(a)]]
        assert a == b, f
