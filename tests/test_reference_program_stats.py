"""Against the UNMODIFIED reference program run end to end (tests/golden/refprog_*.npz,
made by tests/golden/make_golden.py::refprog): output file name, file format, and the
statistics of images/transmission (its RNG stream is the Fortran runtime's, so only
statistics are comparable — SURVEY §7)."""
import numpy as np
import pytest

from conftest import make_system
from parity import SEED, load_golden, sparse_image
from stats_util import assert_same_distribution
from opticalraytrace_amd.tracer import output_basename


def _ref_layers(g):
    ring = np.zeros(401 * 401, np.int64); ring[g["ring_idx"]] = g["ring_cnt"]
    point = np.zeros(401 * 401, np.int64); point[g["point_idx"]] = g["point_cnt"]
    tot = np.zeros(401 * 401, np.int64); tot[g["total_idx"]] = g["total_cnt"]
    assert np.array_equal(ring + point, tot)          # imageMod.f90:110-112
    return ring.reshape(401, 401), point.reshape(401, 401)


@pytest.mark.parametrize("name", ["large", "small"])
def test_file_name_equals_the_reference_programs(name):
    g = load_golden("refprog_" + name)
    _, o = make_system(name)
    assert output_basename(o) == str(g["stem"])


@pytest.mark.parametrize("name", ["large", "small"])
def test_oracle_statistics_match_reference_program(name):
    from oracle.binding import Oracle
    g = load_golden("refprog_" + name)
    n_ref = int(g["nphotons"])
    ring, point = _ref_layers(g)
    _, o = make_system(name)
    orc = Oracle(o)
    n = 1_000_000
    img = np.zeros((2, 401, 401), np.int32); cnt = np.zeros(8, np.uint64)
    orc.trace(1, 0, n, SEED, img, cnt); orc.trace(2, 0, n, SEED, img, cnt)
    assert_same_distribution(img[0], ring, n, n_ref, f"{name} ring")
    assert_same_distribution(img[1], point, n, n_ref, f"{name} point")
    # printed transmissions of the reference run (main.f90:180-181), e.g. "49.25%"
    out = str(g["stdout"]).split()
    ring_t, point_t = float(out[out.index("Ring") + 2].rstrip("%")), float(out[out.index("Point") + 2].rstrip("%"))
    assert abs(100 * (1 - int(cnt[0]) / n) - ring_t) < 0.1
    assert abs(100 * (1 - int(cnt[1]) / n) - point_t) < 0.3


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["large", "small"])
def test_gpu_statistics_match_reference_program(hip_library, name):
    from opticalraytrace_amd.capi import Context
    g = load_golden("refprog_" + name)
    n_ref = int(g["nphotons"])
    ring, point = _ref_layers(g)
    _, o = make_system(name)
    n = 4_000_000
    with Context(o) as ctx:
        ctx.trace(1, 0, n, SEED + 1)          # a different seed: must not matter statistically
        ctx.trace(2, 0, n, SEED + 1)
        img, cnt = ctx.read()
    assert_same_distribution(img[0], ring, n, n_ref, f"{name} ring")
    assert_same_distribution(img[1], point, n, n_ref, f"{name} point")


@pytest.mark.gpu
def test_end_to_end_settings_file_run(hip_library, tmp_path):
    """`bin/raytrace <settings>` replaced: same settings text in, same files out."""
    import os
    from opticalraytrace_amd.params import Settings, resource_dir
    from opticalraytrace_amd.tracer import run_settings_file
    g = load_golden("refprog_small")
    s = Settings(nphotons=1_000_000, make_images=True, bottle_file="clearBottle-small.params",
                 data_folder="run")
    cfg = tmp_path / "cfg.params"
    s.write(str(cfg))
    res = run_settings_file(str(cfg), res_dir=resource_dir(), data_dir=str(tmp_path / "data"), verbose=False)
    folder = tmp_path / "data" / "run"
    stem = str(g["stem"])
    for layer in ("ring", "point", "total"):
        f = folder / f"{stem}_image-{layer}.dat"
        assert f.exists() and f.stat().st_size == 401 * 401 * 8
    tot = np.fromfile(folder / f"{stem}_image-total.dat", np.float64)
    assert tot.sum() == res.image.sum() and tot.sum() > 0
    rows = open(folder / "trans-stats.dat").read().splitlines()
    assert len(rows) == 4 and rows[0].lstrip().startswith("r/%")              # header and record: two 79-column lines each
    # the record's layout is the reference program's (flang list-directed): same line breaks, same fields 3..12
    ref_rows = str(g["stats"]).splitlines()
    assert rows[:2] == ref_rows[:2] and rows[3] == ref_rows[3]
    assert rows[2].split(" , ")[2:] == ref_rows[2].split(" , ")[2:]
    # the reference program printed 2.46 % / 60.07 % for this set-up at 1e6 rays
    assert abs(res.ring_transmitted - 2.46) < 0.1 and abs(res.point_transmitted - 60.07) < 0.3
    ring, point = _ref_layers(g)
    assert_same_distribution(res.image[0], ring, 1_000_000, int(g["nphotons"]), "ring")
    assert_same_distribution(res.image[1], point, 1_000_000, int(g["nphotons"]), "point")
