"""runner.py's experiment loops on the GPU path: every simulation of the lens /
iris / offset / point sweeps runs on one reused context, and a sample of them is checked
against the oracle image for the same keyed rays."""
import os

import numpy as np
import pytest

from parity import SEED

pytestmark = pytest.mark.gpu


def test_runner_sweeps_on_one_context(hip_library, tmp_path):
    from opticalraytrace_amd.sweeps import Sweep
    from opticalraytrace_amd.system import OpticalSystem
    from opticalraytrace_amd.params import resource_dir
    from oracle.binding import Oracle
    n = 20000
    sw = Sweep(nphotons=n, data_dir=str(tmp_path), settings_dir=str(tmp_path / "settings"))
    try:
        sw.lens_experiment()
        assert len(sw.results) == 5 * 5 * 3                       # runner.py:254-261
        sw.iris_experiment()
        assert len(sw.results) == 75 + 4 * 11                     # 2 x 5 sizes + 1 per bottle
        sw.offset_experiment()
        sw.point_images()
        assert len(sw.results) == 75 + 44 + 6 + 4                 # -16 mm is not shipped by the reference
    finally:
        sw.close()
    # the stats files hold one row per simulation (+ header)
    rows = open(tmp_path / "images-lens" / "trans-stats.dat").read().splitlines()
    assert len(rows) == 76
    assert len(open(tmp_path / "iris" / "trans-stats.dat").read().splitlines()) == 45
    assert len([f for f in os.listdir(tmp_path / "images-offset") if f.endswith("-total.dat")]) == 6
    assert len(os.listdir(tmp_path / "settings")) >= 75
    # sample: every 9th simulation against the oracle, same keyed rays
    for name, s, res in sw.results[::9]:
        osys = OpticalSystem.from_settings(s, resource_dir())
        orc = Oracle(osys)
        img = np.zeros((2, 401, 401), np.int32); cnt = np.zeros(8, np.uint64)
        orc.trace(1, 0, n, SEED, img, cnt); orc.trace(2, 0, n, SEED, img, cnt)
        assert np.abs(res.image.astype(np.int64) - img).sum() <= 4, name
        assert np.abs(res.counters.astype(np.int64) - cnt.astype(np.int64)).max() <= 2, name
    # a smaller iris never transmits more (same rays, same draws)
    by = {(s.bottle_file, s.use_bottle, s.iris, s.iris_size): r for _, s, r in sw.results if s.data_folder == "iris"}
    for iris in ("before", "after"):
        t = [by[("clearBottle-small.params", True, iris, z)].point_transmitted for z in (1.0, 0.8, 0.6, 0.4, 0.2)]
        assert all(a >= b for a, b in zip(t, t[1:])), (iris, t)


def test_repeated_runs_are_ordered_with_torch_resets(hip_library):
    """reset() (torch zero_ on the current stream) -> trace -> read must be stream-ordered: the
    context runs on the stream torch hands over (the null stream when that is torch's current
    stream), so 40 back-to-back runs give 40 identical results."""
    from conftest import make_system
    from opticalraytrace_amd.tracer import RayTracer
    _, osys = make_system("small")
    t = RayTracer(osys)
    try:
        first = t.run(20000)
        for _ in range(40):
            r = t.run(20000)
            assert np.array_equal(r.counters, first.counters)
            assert np.array_equal(r.image, first.image)
    finally:
        t.close()
