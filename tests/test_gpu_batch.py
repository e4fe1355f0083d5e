"""Multi-system launches (ort_trace_batch; SURVEY §8 f1, runner.py:113-261): a batch of simulations traced in one launch per
surface program must accumulate, simulation by simulation, exactly what ort_set_system / ort_attach_buffers / ort_trace one
at a time accumulate — images, all 8 counters, deferred rays included — and leave the context as it found it."""
import numpy as np
import pytest

from conftest import make_system
from parity import SEED

pytestmark = pytest.mark.gpu

# every list the batch unit holds a program for (ring / point, iris before / behind, no bottle, elliptical bottle, the crs and
# isors sources) + what it does NOT hold and traces one by one inside the call (the spot source: generic walk; scattering media)
MIX = ["large", "small", "ellipse", "large_iris_before", "small_iris_after", "small_f60_nobottle", "large_crs", "small_isors",
       "small_spot", "small_scatter_bc", "large", "small_scatter_c"]


def _one_by_one(ctx, systems, n, first):
    out = []
    for osys in systems:
        ctx.set_system(osys)
        ctx.reset()
        for phase in (1, 2):
            ctx.trace(phase, first, n, SEED)
        out.append(ctx.read())
    return out


def test_trace_batch_equals_one_by_one(hip_library):
    import torch
    from opticalraytrace_amd.capi import Context, pack_systems
    systems = [make_system(name, nphotons=100)[1] for name in MIX]          # (the spot source's angular steps come from nphotons)
    n, first = 150_011, 12_345
    base = systems[0]
    with Context(base) as ctx:
        want = _one_by_one(ctx, systems, n, first)
        ctx.set_system(base)
        ctx.reset()
        ctx.trace(2, 0, 50_000, SEED)                                         # hits pending in the context's own accumulators
        images = torch.zeros((len(systems), 2, 401, 401), dtype=torch.int32, device="cuda")
        counters = torch.zeros((len(systems), 8), dtype=torch.int64, device="cuda")
        no_image = {1, 8, 9}                                                  # a program kernel, the generic walk, scattering
        torch.cuda.synchronize()
        packed = pack_systems(systems)
        names = []
        for phase in (1, 2):
            ctx.trace_batch(packed, phase, first, n, SEED,
                            [0 if i in no_image else images[i].data_ptr() for i in range(len(systems))],
                            [counters[i].data_ptr() for i in range(len(systems))])
            names.append(ctx.last_kernel_name())
        ctx.synchronize()
        own_img, own_cnt = ctx.read()                                         # the context's own run is untouched by the batch
        got_img, got_cnt = images.cpu().numpy(), counters.cpu().numpy().astype(np.uint64)
        ctx.reset()
        ctx.trace(2, 0, 50_000, SEED)
        ref_img, ref_cnt = ctx.read()
    assert np.array_equal(own_img, ref_img) and np.array_equal(own_cnt, ref_cnt)
    for i, (name, (wimg, wcnt)) in enumerate(zip(MIX, want)):
        assert np.array_equal(got_cnt[i], wcnt), (name, got_cnt[i], wcnt)
        if i in no_image:
            assert not got_img[i].any(), name                                 # nothing was binned anywhere it could be seen
        else:
            assert np.array_equal(got_img[i], wimg), (name, int(np.abs(got_img[i].astype(np.int64) - wimg).sum()))
    assert int(sum(c[5] for _, c in want)) > 100_000                          # rays were binned
    assert all("trace_batch_kernel<" in k or "trace_queue_kernel" in k or "trace_kernel" in k for k in names), names


def test_trace_batch_multi_system_launches_are_used_and_defer(hip_library):
    """The lens experiment's 75 systems at 1e6 photons (runner.py:232-261) through RayTracer.run_many: multi-system launches
    == one simulation after the other, counters bit for bit — with rays deferred to the batched literal re-run on the way."""
    from opticalraytrace_amd.params import Settings
    from opticalraytrace_amd.sweeps import L2_FOCALS, L3_FOCALS, LENS_BOTTLES
    from opticalraytrace_amd.system import OpticalSystem
    from opticalraytrace_amd.tracer import RayTracer
    n = 1_000_000
    systems = [OpticalSystem.from_settings(Settings(nphotons=n, light_source="point", make_images=(k == 0), bottle_file=b, use_bottle=u,
                                                    L3_file=f"achromaticDoublet-f{f3}mm.params", L2_file=f"planoConvex-f{f2}mm.params"))
               for k, f3 in enumerate(L3_FOCALS) for f2 in L2_FOCALS for b, u in LENS_BOTTLES]
    t = RayTracer(systems[0])
    try:
        flags = [s.settings.make_images for s in systems]
        w0 = t.ctx.work_counters()
        multi = t.run_many(systems, want_images=flags)
        name = t.ctx.last_kernel_name()
        w1 = t.ctx.work_counters()
        t.multi_system_launches = False
        single = t.run_many(systems, want_images=flags)
        w2 = t.ctx.work_counters()
    finally:
        t.close()
    assert "trace_batch_kernel<" in name, name
    for i, (a, b) in enumerate(zip(multi, single)):
        assert np.array_equal(a.counters, b.counters), (i, a.counters, b.counters)
        assert (a.image is None) == (b.image is None) == (not flags[i])
        if a.image is not None:
            assert np.array_equal(a.image, b.image), i
    assert w1[1] - w0[1] == w2[1] - w1[1] > 0, (w0, w1, w2)          # the same rays were deferred (and re-run) either way
    assert w1[0] - w0[0] == w2[0] - w1[0] > 10_000_000                # ... and culled in the ring loops


def test_trace_batch_arguments(hip_library):
    import torch
    from opticalraytrace_amd.capi import Context, OrtError, pack_systems
    _, osys = make_system("small")
    _, img_sys = make_system("large_image")
    cnt = torch.zeros((2, 8), dtype=torch.int64, device="cuda")
    with Context(osys) as ctx:
        ctx.trace_batch(pack_systems([]), 2, 0, 1000, SEED, [], [])                     # an empty batch is nothing
        ctx.trace_batch(pack_systems([osys]), 2, 0, 0, SEED, [0], [cnt[0].data_ptr()])  # ... and so is one without rays
        with pytest.raises(OrtError, match="image source"):
            ctx.trace_batch(pack_systems([osys, img_sys]), 2, 0, 1000, SEED, [0, 0], [cnt[0].data_ptr(), cnt[1].data_ptr()])
        with pytest.raises(OrtError, match="counters"):
            ctx.trace_batch(pack_systems([osys]), 2, 0, 1000, SEED, [0], [0])
        with pytest.raises(OrtError, match="phase"):
            ctx.trace_batch(pack_systems([osys]), 3, 0, 1000, SEED, [0], [cnt[0].data_ptr()])
        bad = pack_systems([osys])
        bad[0].n_surfaces[1] = 99
        with pytest.raises(OrtError, match="system 0 of the batch"):
            ctx.trace_batch(bad, 2, 0, 1000, SEED, [0], [cnt[0].data_ptr()])
        ctx.synchronize()
        assert not cnt.cpu().numpy().any()
        # other arithmetics and kernel variants: the batch is traced one by one inside the call, same meaning
        for prec, variant in ((1, 1), (2, 1), (0, 3), (0, 0), (0, 1 | 32), (0, 1 | 64)):
            ctx.set_kernel_variant(1); ctx.set_precision(prec); ctx.set_kernel_variant(variant)
            ctx.reset()
            ctx.trace(2, 5, 30_000, SEED)
            _, want = ctx.read()
            cnt.zero_()
            torch.cuda.synchronize()
            ctx.trace_batch(pack_systems([osys, osys]), 2, 5, 30_000, SEED, [0, 0], [cnt[0].data_ptr(), cnt[1].data_ptr()])
            ctx.synchronize()
            got = cnt.cpu().numpy().astype(np.uint64)
            assert np.array_equal(got[0], want) and np.array_equal(got[1], want), (prec, variant, got, want)
            assert "batch" not in ctx.last_kernel_name()
        ctx.set_kernel_variant(1); ctx.set_precision(0)


def test_trace_batch_ragged_sizes(hip_library):
    """Partial batches, a single ray, a range that starts behind 2^33: multi-system launches == one by one, images and counters."""
    import torch
    from opticalraytrace_amd.capi import Context, pack_systems
    systems = [make_system(name)[1] for name in ("large", "ellipse", "small_iris_after")]
    packed = pack_systems(systems)
    images = torch.zeros((3, 2, 401, 401), dtype=torch.int32, device="cuda")
    counters = torch.zeros((3, 8), dtype=torch.int64, device="cuda")
    with Context(systems[0]) as ctx:
        for n, first in ((1, 0), (63, 5), (64, 0), (65, 1 << 33), (129, 7), (4097, 3), (65472 * 4 + 1, 11)):
            want = _one_by_one(ctx, systems, n, first)
            images.zero_(); counters.zero_()
            torch.cuda.synchronize()
            for phase in (1, 2):
                ctx.trace_batch(packed, phase, first, n, SEED, [images[i].data_ptr() for i in range(3)], [counters[i].data_ptr() for i in range(3)])
                assert "trace_batch_kernel<" in ctx.last_kernel_name()
            ctx.synchronize()
            gi, gc = images.cpu().numpy(), counters.cpu().numpy().astype(np.uint64)
            for i, (wimg, wcnt) in enumerate(want):
                assert np.array_equal(gc[i], wcnt) and np.array_equal(gi[i], wimg), (n, first, i, gc[i], wcnt)


_RANGE_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from conftest import make_system
from opticalraytrace_amd.capi import Context
_, osys = make_system("large")
n = int(sys.argv[3])
with Context(osys) as c:
    c.trace(1, 3, n, 123456789); c.trace(2, 3, n, 123456789)
    img, cnt = c.read()
np.savez(sys.argv[2], img=img, cnt=cnt)
"""


def test_longest_wave_ranges_keep_their_16_bit_queue_indices(hip_library, tmp_path):
    """The queued program kernels keep a queued ray as 16 bits relative to its wave's range; plan_ranges holds a range below
    65 472 rays.  With the grid squeezed to two workgroups (development knob) every wave gets a range of exactly that length:
    the images and counters of both loops equal the default plan's."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = []
    for knobs in ({}, {"ORT_DEV_MAX_BLOCKS": "2"}):
        out = str(tmp_path / f"r{len(got)}.npz")
        r = subprocess.run([sys.executable, "-c", _RANGE_CHILD, root, out, str(1_200_003)], env={**os.environ, **knobs}, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        got.append(np.load(out))
    assert np.array_equal(got[0]["img"], got[1]["img"]) and np.array_equal(got[0]["cnt"], got[1]["cnt"])
    assert int(got[0]["cnt"][5]) > 400_000


@pytest.mark.parametrize("precision", [0, 1])
def test_launch_sizes_around_the_range_plan_thresholds(hip_library, precision):
    """plan_ranges changes shape with the launch size — equal ranges on one round of workgroups below 2.5e6 rays, the two-level
    plan above, longer short ranges in the ring programs as the launch grows (one more batch per 1e6 rays; fp32: per 4e5), 12
    batches in the point loop above 1.6e7 —: at every border the queued kernels' images and counters equal the lockstep
    kernel's, which knows no plan."""
    from opticalraytrace_amd.capi import Context
    _, osys = make_system("large")
    with Context(osys) as c:
        c.set_precision(precision)
        for n in (2_499_937, 2_500_000, 2_500_065, 6_000_001, 16_000_001, 33_000_003):
            out = []
            for variant in (1, 0):
                c.set_kernel_variant(variant)
                c.reset()
                c.trace(1, 7, n, SEED)
                if n <= 16_000_001:
                    c.trace(2, 7, n, SEED)
                out.append(c.read())
            c.set_kernel_variant(1)
            assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]), (precision, n, out[0][1], out[1][1])
