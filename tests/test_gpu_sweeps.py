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
    def records(path):                                  # a record of the stats file is two lines (79-column records)
        return open(path).read().count("point,")
    assert records(tmp_path / "images-lens" / "trans-stats.dat") == 75
    assert open(tmp_path / "images-lens" / "trans-stats.dat").read().startswith(" r/%, p/%, l2%f")
    assert records(tmp_path / "iris" / "trans-stats.dat") == 44
    assert len([f for f in os.listdir(tmp_path / "images-offset") if f.endswith("-total.dat")]) == 6
    assert len(os.listdir(tmp_path / "settings")) >= 75
    # sample: every 9th simulation against the oracle, same keyed rays
    for name, s, res in sw.results[::9]:
        osys = OpticalSystem.from_settings(s, resource_dir())
        orc = Oracle(osys)
        img = np.zeros((2, 401, 401), np.int32); cnt = np.zeros(8, np.uint64)
        orc.trace(1, 0, n, SEED, img, cnt); orc.trace(2, 0, n, SEED, img, cnt)
        if s.make_images:
            assert np.abs(res.image.astype(np.int64) - img).sum() <= 4, name
        else:
            assert res.image is None, name        # make_images false (src/main.f90:183): the counters are the result
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


def test_isors_and_bessel_sweeps(hip_library, tmp_path):
    """runner.py's iSORS_vs_Bessel loop (:267-320: isors source vs the point source with the bottle
    moved to the same ring offset, 7 offsets each) and `-b` (:209-228, image source) on the GPU path;
    a sample against the oracle, the reflecting isors rays counted where the reference aborts."""
    import shutil
    from conftest import res_dir_with_image
    from opticalraytrace_amd.params import ParamsError, resource_dir
    from opticalraytrace_amd.sweeps import Sweep
    from opticalraytrace_amd.system import OpticalSystem
    from oracle.binding import Oracle
    n = 30000
    sw = Sweep(nphotons=n, data_dir=str(tmp_path), settings_dir=str(tmp_path / "settings"))
    try:
        sw.isors_vs_bessel()
        assert len(sw.results) == 14
        with pytest.raises(ParamsError):                     # the packaged res/ holds no Bessel image
            sw.bessel_images()
    finally:
        sw.close()
    assert [s.light_source for _, s, _ in sw.results] == ["isors"] * 7 + ["point"] * 7
    t = open(tmp_path / "iSORS_vs_Bessel" / "trans-stats.dat").read()
    assert t.count("isors,") == 7 and t.count("point,") == 7
    for name, s, res in sw.results[::3]:
        osys = OpticalSystem.from_settings(s, resource_dir())
        orc = Oracle(osys)
        img = np.zeros((2, 401, 401), np.int32); cnt = np.zeros(8, np.uint64)
        orc.trace(1, 0, n, SEED, img, cnt); orc.trace(2, 0, n, SEED, img, cnt)
        assert np.abs(res.image.astype(np.int64) - img).sum() <= 6, name
        assert np.abs(res.counters.astype(np.int64) - cnt.astype(np.int64)).max() <= 3, name
    # the isors ring moves outwards with the offset: its transmission through the fixed optics changes
    t = [r.ring_transmitted for _, s, r in sw.results[:7]]
    assert len(set(t)) > 3
    # -b with an image source file present (the tests' synthetic stand-in for bpm.py's output)
    res = res_dir_with_image(resource_dir())
    shutil.copy(os.path.join(res, "synthetic-source.dat"), os.path.join(res, "bessel-smear.dat"))
    sw = Sweep(nphotons=n, res_dir=res, data_dir=str(tmp_path / "b"))
    try:
        sw.bessel_images()
        assert len(sw.results) == 4 and all(s.light_source == "image" for _, s, _ in sw.results)
        # (the elliptical bottle transmits nothing from this source plane: all rays end in its wall, as in the oracle)
        assert [r.image[1].sum() > 0 for _, _, r in sw.results] == [True, True, False, True]
    finally:
        sw.close()


def test_isors_keyed_rays_vs_oracle(hip_library):
    """Keyed isors rays, the ~3 % that reflect at the axicon included: per-ray outcome, draw count
    and state against the oracle (ORT_ST_NO_INTERSECTION where the reference would abort), and
    the images of both loops."""
    from conftest import make_system
    from opticalraytrace_amd.capi import Context
    from oracle.binding import Oracle
    _, osys = make_system("small_isors")
    orc = Oracle(osys)
    n = 40000
    with Context(osys) as ctx:
        for phase in (1, 2):
            want = orc.trace_rays(phase, n, seed=SEED, first_ray=7)
            got = ctx.trace_rays(phase, n, seed=SEED, first_ray=7)
            assert np.array_equal(got["n_draws"], want["n_draws"])
            assert (got["status"] != want["status"]).sum() <= 2
            same = got["status"] == want["status"]
            for k in (0, 3):
                w = want["emitted"][k:k + 3]
                scale = np.maximum(np.sqrt((w * w).sum(0)), 1e-300)
                assert (np.abs(got["emitted"][k:k + 3] - w) / scale)[:, same].max() <= 1e-12
            if phase == 1:
                gone = want["status"] == 6
                assert 0.02 < gone.mean() < 0.04 and np.array_equal(got["status"] == 6, gone)
        ctx.reset()
        ctx.trace(1, 0, n, SEED)
        ctx.trace(2, 0, n, SEED)
        img, cnt = ctx.read()
    wimg = np.zeros((2, 401, 401), np.int32); wc = np.zeros(8, np.uint64)
    orc.trace(1, 0, n, SEED, wimg, wc); orc.trace(2, 0, n, SEED, wimg, wc)
    assert np.abs(img.astype(np.int64) - wimg).sum() <= 6
    assert np.abs(cnt.astype(np.int64) - wc.astype(np.int64)).max() <= 3


def test_batched_sweep_equals_one_by_one(hip_library, tmp_path):
    """SURVEY f1's design point: the lens experiment's 75 systems queued as ONE batch (ort_set_system staged
    asynchronously through the library's ring of 16 system slots — the batch is longer than the ring —, every
    simulation binning into its own slice of one [75][2][401][401] device array, one wait, one copy back)
    against the same 75 run one at a time: images, counters and the files they leave, identical."""
    from opticalraytrace_amd.sweeps import Sweep
    n = 60_000
    out = {}
    # batched = through ort_trace_batch (multi-system launches: one launch per surface program and loop over the whole batch),
    # queued = the same batch one simulation after the other on the context, single = one run_settings at a time
    for mode, batched, multi in (("batched", True, True), ("queued", True, False), ("single", False, False)):
        sw = Sweep(nphotons=n, data_dir=str(tmp_path / mode), batched=batched, multi_system=multi)
        try:
            sw.lens_experiment()
            sw.iris_experiment(bottles=[("clearBottle-small.params", True)])     # images written (make_images)
        finally:
            sw.close()
        out[mode] = sw.results
    assert len(out["batched"]) == len(out["queued"]) == len(out["single"]) == 75 + 11
    for mode in ("batched", "queued"):
        for (na, sa, ra), (nb, sb, rb) in zip(out[mode], out["single"]):
            assert na == nb and sa == sb
            if sa.make_images:
                assert np.array_equal(ra.image, rb.image), (mode, na)
            else:
                assert ra.image is None, (mode, na)            # make_images false (src/main.f90:183): nothing of the image comes back
            assert np.array_equal(ra.counters, rb.counters), (mode, na)
    assert len({r.counters.tobytes() for _, _, r in out["batched"]}) > 60       # the systems really differ
    assert len({r.image.tobytes() for _, s, r in out["batched"] if s.make_images}) >= 6
    for mode in ("batched", "queued"):
        for folder in ("images-lens", "iris"):
            a, b = tmp_path / mode / folder, tmp_path / "single" / folder
            assert sorted(os.listdir(a)) == sorted(os.listdir(b))
            for f in os.listdir(a):
                assert open(a / f, "rb").read() == open(b / f, "rb").read(), (mode, f)
