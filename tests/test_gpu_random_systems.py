"""HIP path against the oracle on RANDOM optical systems (tests/random_systems.py: the shipped lens
and bottle files perturbed, random wavelength / iris / fibre offset / image diameter / bottle shape).
The filtered predicates carry margins and the surface programs are matched field by field: neither
may depend on the shipped files.  Same bars as test_gpu_parity.py: explicit-input rays bit-exact,
every kernel variant the same image, resident bundles == the oracle's image exactly."""
import numpy as np
import pytest

from parity import SEED, emit_draws, rel_err
from random_systems import SEEDS, random_system

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=SEEDS)
def system(request, hip_library):
    from opticalraytrace_amd.capi import Context
    from opticalraytrace_amd.system import OpticalSystem
    from oracle.binding import Oracle
    settings, res = random_system(request.param)
    osys = OpticalSystem.from_settings(settings, res)
    ctx = Context(osys, device=0)
    yield request.param, settings, osys, ctx, Oracle(osys)
    ctx.close()


@pytest.mark.parametrize("phase", [1, 2])
def test_explicit_rays_bit_exact(system, phase):
    seed, settings, osys, ctx, orc = system
    n = 30000 + 11
    u = np.random.default_rng(50 + seed).random((9, n))
    ref = orc.trace_rays(phase, n, u=u)
    base = emit_draws(settings, phase)
    want = orc.trace_rays(phase, n, pos_dir_in=ref["emitted"], u=u, draw_base=base)
    got = ctx.trace_rays(phase, n, pos_dir_in=ref["emitted"], u=u, draw_base=base)
    for key in ("status", "bin_xy", "n_isect", "n_draws"):
        assert np.array_equal(got[key], want[key]), (seed, phase, key)
    assert np.array_equal(got["pos_dir"], want["pos_dir"]), (seed, phase, rel_err(got["pos_dir"], want["pos_dir"]))


def test_kernel_variants_agree_and_resident_bundles_equal_the_oracle(system):
    import torch
    seed, settings, osys, ctx, orc = system
    n = 200_003
    out = []
    for variant in (0, 1, 3):             # lockstep literal, queued filtered (+ literal re-run), queued literal
        ctx.set_kernel_variant(variant)
        ctx.reset()
        ctx.trace(1, 0, n, SEED)
        ctx.trace(2, 0, n, SEED)
        out.append(ctx.read())
    ctx.set_kernel_variant(1)
    for v in (1, 2):
        assert np.array_equal(out[0][0], out[v][0]), (seed, v)
        assert np.array_equal(out[0][1], out[v][1]), (seed, v, out[0][1], out[v][1])
    # oracle-emitted rays as a resident bundle through the bulk kernel: the image is the oracle's, exactly
    m = 100_000
    want = np.zeros((2, 401, 401), np.int32)
    ctx.reset()
    for phase in (1, 2):
        base = emit_draws(settings, phase)
        em = orc.trace_rays(phase, m, seed=SEED)["emitted"]
        res = orc.trace_rays(phase, m, pos_dir_in=em, seed=SEED, draw_base=base)
        ok = res["status"] == 0
        np.add.at(want[phase - 1], (res["bin_xy"][1][ok] + 200, res["bin_xy"][0][ok] + 200), 1)
        bundle = torch.from_numpy(np.ascontiguousarray(em)).to("cuda:0")
        ctx.trace_resident(phase, 0, m, SEED, base, bundle.data_ptr())
        ctx.synchronize()
    img, cnt = ctx.read()
    assert np.array_equal(img, want), (seed, int(np.abs(img.astype(np.int64) - want).sum()))
    assert int(cnt[4]) == int(want[0].sum()) and int(cnt[5]) == int(want[1].sum())
