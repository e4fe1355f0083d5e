"""N > 1 path on CPU: two gloo ranks run RayTracer.run() (shard -> trace -> all-reduce) with
the oracle injected as the per-rank tracer; the reduced image must equal the single-rank one
bit for bit (draws are keyed on the GLOBAL ray index, integer adds commute — SURVEY §8e)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_RAYS = 30011          # odd on purpose: uneven shards


def _oracle_run(osys, rank=0, world=1):
    """Test-only ShardedRun whose shards are traced by the CPU oracle into host tensors, so the
    package's shard/reduce code runs under gloo on a machine without a GPU."""
    from opticalraytrace_amd.tracer import ShardedRun
    from oracle.binding import Oracle

    class OracleRun(ShardedRun):
        def __init__(self):
            super().__init__(torch.zeros((2, 401, 401), dtype=torch.int32),
                             torch.zeros(8, dtype=torch.int64), rank, world)
            self.orc = Oracle(osys)

        def _trace_shard(self, phase, lo, cnt, seed):
            self.orc.trace(phase, lo, cnt, seed, self.image.numpy(),
                           self.counters.numpy().view(np.uint64), nthreads=2)

        # hooks of run_many (a batch of simulations): host arrays, a fresh oracle per system
        def _new_accumulators(self, n_images, n):
            return torch.zeros((n_images, 2, 401, 401), dtype=torch.int32), torch.zeros((n, 8), dtype=torch.int64)

        def _begin_simulation(self, system, image, counters):
            self.orc = Oracle(system)
            self.image, self.counters = image, counters
    return OracleRun()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import make_system
    _, osys = make_system("small")
    res = _oracle_run(osys, rank, world).run(N_RAYS, seed=123456789)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), image=res.image, counters=res.counters)
    dist.barrier()
    dist.destroy_process_group()


def _batch_systems():
    from conftest import make_system
    return [make_system(name)[1] for name in ("small", "large_iris_before", "small_f60_nobottle")]


def _batch_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    systems = _batch_systems()
    for s in systems:
        s.settings.nphotons = 9001
    res = _oracle_run(systems[0], rank, world).run_many(systems, seed=123456789)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), image=np.stack([r.image for r in res]),
             counters=np.stack([r.counters for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_run_equals_single_rank(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import make_system
    _, osys = make_system("small")
    single = _oracle_run(osys).run(N_RAYS, seed=123456789)
    for r in range(world):
        g = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(g["image"], single.image)           # every rank holds the global image
        assert np.array_equal(g["counters"], single.counters)
    assert single.image.sum() > 0 and int(single.counters[3]) > 6 * N_RAYS * 0.9


def test_batched_sweep_over_two_ranks_equals_single_rank(tmp_path):
    """ShardedRun.run_many (a sweep queued as one batch, ONE all-reduce of the stacked images): two gloo ranks
    against one rank, three different systems — every rank ends with every simulation's global image."""
    mp.spawn(_batch_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    systems = _batch_systems()
    for s in systems:
        s.settings.nphotons = 9001
    single = _oracle_run(systems[0]).run_many(systems, seed=123456789)
    one_by_one = [_oracle_run(s).run(9001, seed=123456789) for s in systems]
    for r in range(2):
        g = np.load(tmp_path / f"rank{r}.npz")
        for i, res in enumerate(single):
            assert np.array_equal(g["image"][i], res.image) and np.array_equal(g["counters"][i], res.counters), (r, i)
    for a, b in zip(single, one_by_one):
        assert np.array_equal(a.image, b.image) and np.array_equal(a.counters, b.counters)
    assert len({s.image.tobytes() for s in single}) == 3
