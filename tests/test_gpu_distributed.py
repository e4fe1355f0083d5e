"""The multi-GPU path as far as ONE MI355X can exercise it, through the product code:
RayTracer + ort_attach_buffers + RCCL (`nccl` backend) all-reduce on the tracer's stream, shards of
the global ray range traced by separate contexts, the C-ABI ort_allreduce, and bench.py's own
rank launcher.  The world-size 2 / 3 shard + reduce logic runs under gloo in
test_distributed_gloo.py; the 8-GPU run is the driver's."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import make_system
from parity import SEED

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 300_007        # odd: uneven shards


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def plain(hip_library):
    """The single-context run everything below must reproduce bit for bit."""
    from opticalraytrace_amd.tracer import RayTracer
    _, osys = make_system("large")
    t = RayTracer(osys, device=0)
    try:
        res = t.run(N, seed=SEED)
    finally:
        t.close()
    assert res.image[0].sum() > 0 and res.image[1].sum() > 0.3 * N
    return osys, res


def test_rccl_reduce_on_one_rank_equals_plain_run(plain):
    """init_process_group("nccl", world_size=1): RayTracer.run() with the all-reduce forced — the
    collective really runs on the attached torch tensors, stream-ordered after the trace."""
    import torch
    import torch.distributed as dist
    from opticalraytrace_amd.tracer import RayTracer
    osys, want = plain
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        t = RayTracer(osys, device=0, rank=0, world=1)
        try:
            t.reset()
            for phase in (1, 2):
                t.trace_phase(phase, N, SEED)
            t.reduce(force=True)                    # RCCL all-reduce of image + counters
            got = t.result(N)
            # twice: the reduce of a second run must see the second run's trace, not race it
            t.reset()
            for phase in (1, 2):
                t.trace_phase(phase, N, SEED)
            t.reduce(force=True)
            again = t.result(N)
        finally:
            t.close()
    finally:
        dist.destroy_process_group()
    for r in (got, again):
        assert np.array_equal(r.image, want.image)
        assert np.array_equal(r.counters, want.counters)


@pytest.mark.parametrize("world", [2, 5])
def test_shards_on_separate_contexts_sum_to_the_single_run(plain, world):
    """Rank r of `world` traces shard_range(N, r, world) on its own context (all on this one GPU);
    the integer sum of the shard images / counters is the single run's, bit for bit."""
    from opticalraytrace_amd.tracer import RayTracer, shard_range
    osys, want = plain
    img = np.zeros_like(want.image, dtype=np.int64)
    cnt = np.zeros(8, dtype=np.uint64)
    covered = 0
    tracers = [RayTracer(osys, device=0, rank=r, world=world) for r in range(world)]
    try:
        for t in tracers:
            t.reset()
            for phase in (1, 2):
                t.trace_phase(phase, N, SEED)
        for r, t in enumerate(tracers):
            res = t.result(N)
            covered += shard_range(N, r, world)[1]
            assert 0 < res.image.sum() < want.image.sum()
            img += res.image
            cnt += res.counters
    finally:
        for t in tracers:
            t.close()
    assert covered == N
    assert np.array_equal(img, want.image.astype(np.int64))
    assert np.array_equal(cnt, want.counters)


def test_c_abi_allreduce(plain):
    """ort_allreduce over the contexts of one process (here: one — a box with one GPU): RCCL is
    resolved at run time, the group call runs on the context's stream and leaves the sums of a
    one-rank communicator, i.e. the run itself."""
    import ctypes as C
    from opticalraytrace_amd import capi
    osys, want = plain
    lib = capi.load_library()
    assert lib.ort_allreduce(None, 1) == -1
    with capi.Context(osys, device=0) as a, capi.Context(osys, device=0) as b:
        arr = (C.c_void_p * 2)(a._h, b._h)
        assert lib.ort_allreduce(arr, 2) == -1 and b"one device" in lib.ort_last_error()
        assert lib.ort_allreduce(arr, 0) == -1
        a.reset()
        for phase in (1, 2):
            a.trace(phase, 0, N, SEED)
        capi.allreduce([a])
        capi.allreduce([a])                         # the communicator is kept between calls
        img, cnt = a.read()
    assert np.array_equal(img, want.image)
    assert np.array_equal(cnt, want.counters)


def test_c_abi_allreduce_ends_its_group_on_a_failed_collective(plain):
    """A collective that fails INSIDE ncclGroupStart ... ncclGroupEnd (here: an invalid datatype planted through the test
    hook ort_debug_fault_allreduce, armed for one call) is reported — and the group is ended all the same, the cached
    communicators are dropped: the next ort_allreduce of the process initialises new ones, works and leaves the right sums.
    (Round 3's ort_allreduce returned from inside the open group.)"""
    import ctypes as C
    from opticalraytrace_amd import capi
    osys, want = plain
    lib = capi.load_library()
    with capi.Context(osys, device=0) as a:
        a.reset()
        for phase in (1, 2):
            a.trace(phase, 0, N, SEED)
        arr = (C.c_void_p * 1)(a._h)
        capi.allreduce([a])                         # communicators exist
        assert capi.allreduce_ranks() == 1
        lib.ort_debug_fault_allreduce(1)
        rc = lib.ort_allreduce(arr, 1)
        assert rc != 0 and b"ncclAllReduce" in lib.ort_last_error(), (rc, lib.ort_last_error())
        assert capi.allreduce_ranks() == 0          # ... and were dropped by the failure
        capi.allreduce([a])                         # the group was ended: this one goes through, on new communicators
        assert capi.allreduce_ranks() == 1
        capi.comm_destroy()                         # ort_comm_destroy: given back; the next reduce would make new ones
        assert capi.allreduce_ranks() == 0
        capi.comm_destroy()                         # (idempotent)
        img, cnt = a.read()
    assert np.array_equal(img, want.image)
    assert np.array_equal(cnt, want.counters)


def _bench(*flags, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_starts_its_own_ranks_or_fails_cleanly():
    """`python bench.py --gpus 2` as the driver issues it (no launcher): on a box with fewer GPUs a
    message and a non-zero exit, no traceback; with enough GPUs the line says n_gpus 2."""
    import torch
    p = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-fast", "--no-fp32")
    if torch.cuda.device_count() < 2:
        assert p.returncode != 0 and "GPU(s)" in p.stderr and "Traceback" not in p.stderr
        assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    else:
        assert p.returncode == 0, p.stderr[-2000:]
        line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["value"] > 0
        assert line["reduce_verified"] is True and line["ranks_seen"] == 2


def test_bench_one_rank_through_rccl():
    """The N > 1 code path of bench.py on one rank (--force-dist: process group, barrier, RCCL
    all-reduce inside the timed region, max over ranks)."""
    p = _bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--force-dist", "--no-fast", "--no-fp32",
               "--rays", "1000000")
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["value"] > 1e9
    assert abs(line["config"]["intersections_per_step"] / 1e6 - 6.31) < 0.05      # 6.31 per point ray
    assert line["roofline"]["bound"] == "valu_fp64" and 0 < line["roofline"]["frac"] < 1
    # the run proves its own reduce: one extra sharded + all-reduced step == the same rays traced by rank 0 alone
    assert line["reduce_verified"] is True and line["ranks_seen"] == 1 and line["reduce_ms"] > 0
    assert 0 < line["kernel_ms_per_step_over_ranks"]["min"] <= line["kernel_ms_per_step_over_ranks"]["max"]


def test_bench_single_process_layout():
    """`bench.py --single-process`: one process, one context per GPU, ort_allreduce.  One GPU: the line of a
    one-context run with the reduce forced (a one-rank RCCL communicator inside the library); asking for
    more GPUs than the box has fails cleanly, as the torchrun layout does."""
    p = _bench("--gpus", "1", "--single-process", "--force-dist", "--steps", "3", "--warmup", "1", "--rays", "1000000")
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 1e9 and "single-process" in line["config"]["host"]
    assert abs(line["config"]["intersections_per_step"] / 1e6 - 6.31) < 0.05
    assert line["reduce_ms"] > 0 and 0 < line["roofline"]["frac"] < 1
    assert line["reduce_verified"] is True and line["ranks_seen"] == 1            # ncclCommCount of the library's communicator
    assert 0 < line["kernel_ms_per_step_over_ranks"]["min"] <= line["kernel_ms_per_step_over_ranks"]["max"]
    import torch
    n = torch.cuda.device_count() + 1
    p = _bench("--gpus", str(n), "--single-process", "--steps", "3", "--warmup", "1")
    assert p.returncode != 0 and "GPU(s)" in p.stderr and "Traceback" not in p.stderr


def test_c_abi_allreduce_over_two_devices(plain):
    """ort_allreduce over TWO contexts on two devices (skipped on a one-GPU box; the driver's 8-GPU node
    runs it): shard 0 on device 0, shard 1 on device 1, one ort_allreduce — both contexts then hold the
    single run's image and counters, bit for bit.  Two collectives (image, counters) per communicator in
    one RCCL group."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: UNVERIFIED on hardware until an 8-GPU lease runs it")
    from opticalraytrace_amd import capi
    from opticalraytrace_amd.tracer import shard_range
    osys, want = plain
    with capi.Context(osys, device=0) as a, capi.Context(osys, device=1) as b:
        for r, c in enumerate((a, b)):
            lo, cnt = shard_range(N, r, 2)
            c.reset()
            for phase in (1, 2):
                c.trace(phase, lo, cnt, SEED)
        capi.allreduce([a, b])
        for c in (a, b):
            img, cnt = c.read()
            assert np.array_equal(img, want.image) and np.array_equal(cnt, want.counters)


def _two_rank_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import make_system as mk
    from opticalraytrace_amd.tracer import RayTracer
    _, osys = mk("large")
    t = RayTracer(osys, device=0, rank=rank, world=world)
    try:
        res = t.run(N, seed=SEED)
        many = t.run_many([osys, osys])                      # a batch: one all-reduce of the stacked images
    finally:
        t.close()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), image=res.image, counters=res.counters,
             many_image=np.stack([m.image for m in many]), many_counters=np.stack([m.counters for m in many]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_one_gpu_over_gloo(plain, tmp_path):
    """The N > 1 path of the PRODUCT code with N = 2 on this one GPU: two processes, each a RayTracer (HIP context,
    attached torch tensors, its shard of the global ray range), the sum over ranks by torch.distributed — over gloo,
    because RCCL needs one device per rank (that leg is the 8-GPU node's).  Both ranks end with the single run's image
    and counters, bit for bit; so does a batch (run_many)."""
    import torch.multiprocessing as mp
    osys, want = plain
    mp.spawn(_two_rank_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        g = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(g["image"], want.image) and np.array_equal(g["counters"], want.counters), r
        n2 = osys.settings.nphotons                          # run_many traces the systems' own photon counts
        assert g["many_image"].shape[0] == 2 and np.array_equal(g["many_image"][0], g["many_image"][1])
        assert int(g["many_counters"][0][3]) > 6 * n2 * 0.9


def test_bench_two_rank_flow_rehearsed_on_one_gpu():
    """`bench.py --gpus 2 --rehearse`: the driver's N > 1 invocation end to end on one GPU — bench.py starts its own
    two ranks (torch.distributed.run), both trace their shard on device 0, barrier + max over ranks, the sum of image and
    counters inside the timed region (gloo here, RCCL on a real node), rank 0 prints ONE line."""
    p = _bench("--gpus", "2", "--rehearse", "--steps", "3", "--warmup", "1", "--no-fast", "--no-fp32", "--rays", "1000000")
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and "rehearsal" in line["config"]
    assert line["config"]["rays_per_layer_per_step"] == 2_000_000 and line["config"]["rays_per_gpu_per_launch"] == 1_000_000
    assert abs(line["config"]["intersections_per_step"] / 2e6 - 6.31) < 0.05          # both shards counted
    assert line["value"] > 1e9 and line["reduce_ms"] > 0
    # two REAL shards summed (gloo): the extra step of both ranks, reduced, equals rank 0's trace of the whole range
    assert line["reduce_verified"] is True and line["ranks_seen"] == 2
    assert 0 < line["kernel_ms_per_step_over_ranks"]["min"] <= line["kernel_ms_per_step_over_ranks"]["max"]
