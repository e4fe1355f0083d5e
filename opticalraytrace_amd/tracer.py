"""Host driver: the Python counterpart of `program raytrace` (reference src/main.f90).

It reproduces the outer boundary of the reference for the hot path — read a
settings file, trace `nphotons` ring rays then `nphotons` point rays, write the
three raw float64 image files and append the `trans-stats.dat` row — with the two
OpenMP loops (src/main.f90:90-109, :127-162) replaced by `ort_trace` launches on
one or more MI355X.

Multi-GPU (one process per GPU, torch.distributed / RCCL): the global ray index
range [0, nphotons) of each phase is cut into contiguous shards, one per rank
(SURVEY §8e).  Draws are keyed on the GLOBAL ray index, so the summed image is
bit-identical for any number of ranks.  The only exchange is one sum-all-reduce of
the int32 image (1.29 MB) and one of the 8 int64 counters per run — the RCCL
equivalent of `!$omp atomic` on the shared image + `reduction(+:rcount,pcount)`
(src/imageMod.f90:55, src/main.f90:88).

torch is used for device memory, streams and torch.distributed only.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from . import fstr
from .capi import (C_BINNED_POINT, C_BINNED_RING, C_ISECT_POINT, C_ISECT_RING, C_LOST_POINT,
                   C_LOST_RING, IMAGE_N, NUM_COUNTERS, Context)
from .params import Settings
from .system import OpticalSystem

DEFAULT_SEED = 123456789          # src/main.f90:79


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, lo+cnt) of the global ray range for `rank` of `world`."""
    if world < 1 or not (0 <= rank < world) or n < 0:
        raise ValueError("bad shard request")
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi - lo


@dataclass
class RunResult:
    image: Optional[np.ndarray]   # int32 [2][401][401]; [0] ring, [1] point (None: a batched simulation whose image nobody asked for)
    counters: np.ndarray          # uint64 [8], see include/ort.h ORT_C_*
    nphotons: int

    @property
    def ring_transmitted(self) -> float:      # main.f90:175,180
        return 100.0 * (1.0 - (float(self.counters[C_LOST_RING]) / float(self.nphotons)))

    @property
    def point_transmitted(self) -> float:     # main.f90:175,181
        return 100.0 * (1.0 - (float(self.counters[C_LOST_POINT]) / float(self.nphotons)))

    @property
    def intersections(self) -> int:
        return int(self.counters[C_ISECT_RING]) + int(self.counters[C_ISECT_POINT])


class ShardedRun:
    """Rank-local half of a sharded run: owns the accumulators (torch tensors), cuts the global
    ray range, and sums image + counters over ranks.  How a shard is traced is the subclass's
    business (`_trace_shard`); the only tracer in this package is the HIP one below."""

    def __init__(self, image, counters, rank: int = 0, world: int = 1, process_group=None):
        self.image, self.counters = image, counters
        self.rank, self.world, self.group = rank, world, process_group

    def _trace_shard(self, phase: int, lo: int, cnt: int, seed: int) -> None:
        raise NotImplementedError

    def _synchronize(self) -> None:
        pass

    def _on_stream(self):
        """Context manager: tensor operations issued inside run on the stream the traces run on (GPU tracer)."""
        import contextlib
        return contextlib.nullcontext()

    def _flush(self) -> None:
        """Everything traced so far is in `image` / `counters` (stream-ordered)."""

    def flush(self) -> None:
        """Complete `image` / `counters` with everything traced so far (asynchronous)."""
        self._flush()

    def reset(self) -> None:
        self._flush()
        with self._on_stream():
            self.image.zero_()
            self.counters.zero_()

    def trace_phase(self, phase: int, nphotons: int, seed: int = DEFAULT_SEED) -> None:
        """This rank's shard of one phase (asynchronous on the GPU path)."""
        lo, cnt = shard_range(nphotons, self.rank, self.world)
        self._trace_shard(phase, lo, cnt, seed)

    def reduce(self, force: bool = False) -> None:
        """Sum image + counters over ranks (RCCL all-reduce over xGMI); no-op for world == 1
        unless `force` (then the collective runs on the single rank: a plumbing check)."""
        self._flush()
        if self.world > 1 or force:
            import torch.distributed as dist
            with self._on_stream():
                dist.all_reduce(self.image, op=dist.ReduceOp.SUM, group=self.group)
                dist.all_reduce(self.counters, op=dist.ReduceOp.SUM, group=self.group)

    def result(self, nphotons: int) -> RunResult:
        self._flush()
        self._synchronize()
        with self._on_stream():
            return RunResult(self.image.cpu().numpy().copy(),
                             self.counters.cpu().numpy().astype(np.uint64), nphotons)

    def run(self, nphotons: int, seed: int = DEFAULT_SEED, phases=(1, 2)) -> RunResult:
        """Both loops of src/main.f90 + the reduction; returns the global result on every rank."""
        self.reset()
        for phase in phases:
            self.trace_phase(phase, nphotons, seed)
        self.reduce()
        return self.result(nphotons)

    # ---- a batch of simulations (a sweep): hooks for where the accumulators live and how a system is staged
    def _new_accumulators(self, n_images: int, n: int):
        """(images [n_images][2][401][401] int32, counters [n][8] int64), zero-filled, where the traces write."""
        raise NotImplementedError

    def _begin_simulation(self, system, image, counters) -> None:
        raise NotImplementedError

    def _end_batch(self) -> None:
        pass

    def _to_host(self, images, counters):
        """The batch's accumulators as numpy arrays (GPU tracer: through pinned staging buffers)."""
        return images.cpu().numpy(), counters.cpu().numpy()

    def _trace_many(self, items, seed: int, phases) -> None:
        """items: (system, its image or None, the scratch image if it wants none, its counters).  Here: one simulation after
        the other; the GPU tracer hands the batch to ort_trace_batch (multi-system launches)."""
        try:
            for system, image, scratch, counters in items:
                self._begin_simulation(system, image if image is not None else scratch, counters)
                lo, cnt = shard_range(system.settings.nphotons, self.rank, self.world)
                for phase in phases:
                    self._trace_shard(phase, lo, cnt, seed)
        finally:
            self._end_batch()

    def run_many(self, systems, seed: int = DEFAULT_SEED, phases=(1, 2), want_images=True):
        """A batch of simulations (a sweep: runner.py starts one process per settings file, :26-47) queued
        back to back: per simulation the system is staged (`_begin_simulation`: on the GPU asynchronously,
        ort_set_system), the accumulators are that simulation's slice of ONE array [n_sim][2][401][401]
        (+ [n_sim][8] counters) and this rank's shard of both loops is launched; then ONE sum over the ranks of
        the whole arrays, one wait, one copy back.  Returns one RunResult per system, each bit-identical to
        `set_system(s); run()` done one at a time, on every rank.
        `want_images`: True, False, or one flag per system — a simulation whose image nobody reads (the reference's
        `make_images = .false.`, src/main.f90:183: only the transmission row is written) bins into ONE scratch image
        shared by all such simulations and comes back with `image = None`: nothing of it is allocated per simulation,
        summed over ranks or copied to the host (75 x 1.29 MB in runner.py's lens experiment)."""
        n = len(systems)
        if n == 0:
            return []
        flags = [bool(want_images)] * n if isinstance(want_images, (bool, int)) else [bool(w) for w in want_images]
        if len(flags) != n:
            raise ValueError("want_images: one flag per system")
        slot = {}                                   # simulation -> its image in the batch array; the last one is the scratch
        for i, w in enumerate(flags):
            if w:
                slot[i] = len(slot)
        n_img = len(slot)
        with self._on_stream():                 # zero-filled on the stream the traces into them run on
            images, counters = self._new_accumulators(n_img + (1 if n_img < n else 0), n)
        self._trace_many([(system, images[slot[i]] if i in slot else None, images[n_img] if i not in slot else None, counters[i])
                          for i, system in enumerate(systems)], seed, phases)
        with self._on_stream():
            if self.world > 1:
                import torch.distributed as dist
                if n_img:
                    dist.all_reduce(images[:n_img], op=dist.ReduceOp.SUM, group=self.group)
                dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=self.group)
            h_img, h_cnt = self._to_host(images[:n_img], counters)
        h_cnt = h_cnt.astype(np.uint64)
        return [RunResult(h_img[slot[i]] if i in slot else None, h_cnt[i], s.settings.nphotons) for i, s in enumerate(systems)]


class RayTracer(ShardedRun):
    """One rank's MI355X tracer.  `device` is the local HIP device index.  No CPU fallback."""

    def __init__(self, system: OpticalSystem, device: int = 0, rank: int = 0, world: int = 1,
                 process_group=None):
        from . import capi
        if not capi.torch_safe():
            raise RuntimeError("this process loaded libort_hip.so without torch (load_library(no_torch=True), the single-GPU "
                               "process entry): a RayTracer needs torch's HIP runtime to be the one the library is bound to")
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("RayTracer needs a HIP device: the trace path has no CPU fallback")
        self.torch = torch
        self.system = system
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        # accumulators live in torch tensors so RCCL can reduce them in place
        image = torch.zeros((2, IMAGE_N, IMAGE_N), dtype=torch.int32, device=self.device)
        counters = torch.zeros(NUM_COUNTERS, dtype=torch.int64, device=self.device)
        super().__init__(image, counters, rank, world, process_group)
        # the context traces on the stream that is current NOW; everything this class allocates, zeroes, reduces or
        # copies later is issued under that same stream (`_on_stream`), whatever stream the caller has made current by then
        self.stream = torch.cuda.current_stream(self.device)
        self.ctx = Context(system, device=device, stream=self.stream.cuda_stream)
        self.ctx.attach_buffers(self.image.data_ptr(), self.counters.data_ptr())
        self._pinned = {}

    def _on_stream(self):
        return self.torch.cuda.stream(self.stream)

    def close(self) -> None:
        self.ctx.close()

    def set_system(self, system: OpticalSystem) -> None:
        """Re-stage another optical system on the same context (sweeps: src/main.f90 is re-run
        once per settings file by runner.py; here only 2 KB move)."""
        self.system = system
        self.ctx.set_system(system)

    def _trace_shard(self, phase: int, lo: int, cnt: int, seed: int) -> None:
        self.ctx.trace(phase, lo, cnt, seed)

    def _flush(self) -> None:
        self.ctx.flush()           # per-XCD replicas -> the attached image tensor

    def _synchronize(self) -> None:
        self.torch.cuda.synchronize(self.device)

    def run(self, nphotons: Optional[int] = None, seed: int = DEFAULT_SEED,
            phases=(1, 2)) -> RunResult:
        n = self.system.settings.nphotons if nphotons is None else nphotons
        return super().run(n, seed, phases)

    # ---- hooks of ShardedRun.run_many
    def _new_accumulators(self, n_images: int, n: int):
        images = self.torch.zeros((n_images, 2, IMAGE_N, IMAGE_N), dtype=self.torch.int32, device=self.device)
        counters = self.torch.zeros((n, NUM_COUNTERS), dtype=self.torch.int64, device=self.device)
        return images, counters

    def _to_host(self, images, counters):
        # pinned staging buffers, kept between batches (a pageable copy goes through the runtime's own bounce buffers page
        # by page — and a first-time 97 MB host array is 24 000 page faults); one wait for both copies
        t = self.torch
        out = []
        for k, src in enumerate((images, counters)):
            pin = self._pinned.get(k)
            if pin is None or pin.numel() < src.numel():
                pin = self._pinned[k] = t.empty(max(src.numel(), 1), dtype=src.dtype, pin_memory=True)
            dst = pin[:src.numel()].view(src.shape)
            dst.copy_(src, non_blocking=True)
            out.append(dst)
        self.stream.synchronize()
        return out[0].numpy().copy(), out[1].numpy().copy()

    multi_system_launches = True      # run_many through ort_trace_batch; False: one simulation after the other (tests, A/B)

    def _trace_many(self, items, seed: int, phases) -> None:
        """ort_trace_batch per loop and per group of simulations with the same shard of rays: the simulations whose lists are
        surface programs share multi-system launches, the library traces the rest one by one (include/ort.h).  The image
        source brings its own table per simulation (ort_set_image_source): those go through the one-by-one path here."""
        if not self.multi_system_launches:
            return super()._trace_many(items, seed, phases)
        from .capi import pack_systems
        groups, single = {}, []
        for it in items:
            if it[0].settings.light_source == "image":
                single.append(it)
            else:
                groups.setdefault(shard_range(it[0].settings.nphotons, self.rank, self.world), []).append(it)
        for (lo, cnt), its in groups.items():
            packed = pack_systems([it[0] for it in its])
            imgs = [it[1].data_ptr() if it[1] is not None else 0 for it in its]
            cnts = [it[3].data_ptr() for it in its]
            for phase in phases:
                self.ctx.trace_batch(packed, phase, lo, cnt, seed, imgs, cnts)
        if single:
            super()._trace_many(single, seed, phases)

    def _begin_simulation(self, system: OpticalSystem, image, counters) -> None:
        self.set_system(system)                                          # asynchronous: the next slot of the ring
        self.ctx.attach_buffers(image.data_ptr(), counters.data_ptr())   # asynchronous: completes the previous slice

    def _end_batch(self) -> None:
        # back to the tracer's own accumulators: completes the last simulation's slice (stream-ordered)
        self.ctx.attach_buffers(self.image.data_ptr(), self.counters.data_ptr())


class LocalTracer:
    """One process, one GPU, no torch: both loops into the library's own accumulators, read back by ort_read.  This is
    what `python -m opticalraytrace_amd` uses when it is not started under torch.distributed.run — runner.py's model is
    one PROCESS per simulation (:26-47), and importing torch is most of such a process's life.  Same calls, same
    results as RayTracer with world == 1 (tests/test_gpu_host.py).  No CPU fallback."""

    def __init__(self, system: OpticalSystem, device: int = 0):
        self.system = system
        self.ctx = Context(system, device=device)

    def close(self) -> None:
        self.ctx.close()

    def run(self, nphotons: Optional[int] = None, seed: int = DEFAULT_SEED, phases=(1, 2)) -> RunResult:
        n = self.system.settings.nphotons if nphotons is None else nphotons
        self.ctx.reset()
        for phase in phases:
            self.ctx.trace(phase, 0, n, seed)
        image, counters = self.ctx.read()
        return RunResult(image, counters.astype(np.uint64), n)


# ---------------------------------------------------------------------------
# output side of the boundary: src/main.f90:45-48, :168-185; src/imageMod.f90:93-114
# ---------------------------------------------------------------------------
def output_basename(system: OpticalSystem) -> str:
    """File-name stem of src/main.f90:45-48."""
    s, b = system.settings, system.bottle
    from .system import PI
    alpha_rad = s.alpha * PI / 180.0
    iris = {"before": [True, False], "after": [False, True], "none": [False, False]}[s.iris]
    return (s.light_source.strip() + "_bottle_" + fstr.str_logical(s.use_bottle)
            + "_Ra_" + fstr.str_real(b.radiusa, 7) + "_Rb_" + fstr.str_real(b.radiusb, 7)
            + "_offset_" + fstr.str_real(b.centre[2], 7)
            + "_" + fstr.str_logical_array(iris) + "_" + fstr.str_real(s.iris_size, 7)
            + "_L2f_" + fstr.str_real(system.L2[0].f, 6) + "_L3f_" + fstr.str_real(system.L3[0].f, 6)
            + "_fo_" + fstr.str_real(s.fibre_offset, 7)
            + "_alp_" + fstr.str_real(alpha_rad * 180 / PI, 7)
            + "_bwidth_" + fstr.str_real(s.ring_width, 7)
            + "_sep_" + fstr.str_real(s.isors_offset, 7))


def write_images(image: np.ndarray, stem: str) -> Tuple[str, str, str]:
    """writeImage2D, src/imageMod.f90:93-114: three raw float64 streams, xp fastest."""
    ring = image[0].astype(np.float64)
    point = image[1].astype(np.float64)
    names = (stem + "-ring.dat", stem + "-point.dat", stem + "-total.dat")
    ring.tofile(names[0])
    point.tofile(names[1])
    (ring + point).tofile(names[2])
    return names


STATS_HEADER = ("r/%, p/%, l2%f, l3%f, bottle?, radiusA, radiusB, iris_pos, iris_radius, "
                "offset, source_type, seperation")      # main.f90:171


def stats_row(system: OpticalSystem, res: RunResult) -> str:
    """The record of src/main.f90:173-178, laid out as list-directed output of the compiler the reference is
    built with here (flang: fstr.list_directed_record — blanks, shortest-digit reals, records of 79 columns);
    equal, character for character, to what the unmodified program leaves (tests/golden/refprog_*.npz)."""
    s, b = system.settings, system.bottle
    iris = {"before": (True, False), "after": (False, True), "none": (False, False)}[s.iris]
    return fstr.list_directed_record([
        res.ring_transmitted, ",", res.point_transmitted, ",", system.L2[1].f, ",", system.L3[1].f, ",", bool(s.use_bottle), ",",
        b.radiusa, ",", b.radiusb, ",", iris[0], iris[1], ",", fstr.str_real(s.iris_size, 7), ",", b.centre[2], ",",
        s.light_source.strip() + ",", s.isors_offset])


def append_stats(folder: str, system: OpticalSystem, res: RunResult) -> str:
    """Append the trans-stats.dat record of src/main.f90:168-178 (header first if the file is new)."""
    path = os.path.join(folder, "trans-stats.dat")
    new = not os.path.exists(path)
    with open(path, "a") as f:
        if new:
            f.write(fstr.list_directed_record([STATS_HEADER]))
        f.write(stats_row(system, res))
    return path


# ---------------------------------------------------------------------------
# ray-path tracker dumps: src/stackMod.f90:24-52, src/main.f90:72-74,103-107,121-124,144-160
# ---------------------------------------------------------------------------
BLANK = "  \n"          # `write(u,*)" "`


def format_paths(path: np.ndarray, npath: np.ndarray, status: np.ndarray) -> str:
    """Text of a `*-ringtrace.dat` / `*-pointtrace.dat` file: per ray the pushed positions popped
    off the stack (last first) as `3(F10.7,1x)`, then three blank records; a ray lost in
    `telescope` leaves only the three blank records (write_empty), a ray lost in the bottle
    leaves its two points and six blank records (main.f90:150-155)."""
    from .capi import ST_HELP3, ST_LOST_BOTTLE, ST_LOST_TELESCOPE
    out = []
    for p, k, st in zip(path, npath, status):
        if st in (ST_LOST_TELESCOPE, ST_HELP3):
            out.append(BLANK * 3)
            continue
        for x, y, z in p[:k][::-1]:
            out.append(f"{x:10.7f} {y:10.7f} {z:10.7f}\n")
        out.append(BLANK * (6 if st == ST_LOST_BOTTLE else 3))
    return "".join(out)


def write_tracker_files(tracer, system: OpticalSystem, folder: str,
                        seed: int = DEFAULT_SEED) -> Tuple[str, str]:
    """`use_tracker`: dump the paths of the run's rays (at most 1e4, setupMod.f90:75)."""
    n = system.settings.nphotons
    stem = os.path.join(folder, output_basename(system))
    names = (stem + "-ringtrace.dat", stem + "-pointtrace.dat")
    for phase, name in zip((1, 2), names):
        path, npath, status = tracer.ctx.trace_paths(phase, n, seed=seed, first_ray=0)
        with open(name, "w") as f:
            f.write(format_paths(path, npath, status))
    return names


def run_settings(settings: Settings, res_dir: Optional[str] = None, data_dir: str = "data",
                 device: int = 0, verbose: bool = True, tracer: Optional[RayTracer] = None) -> RunResult:
    """One simulation = one execution of `bin/raytrace <settings>` for the hot path: build the
    system, trace both loops, append the stats row, write the images (src/main.f90).  Pass a
    `tracer` to reuse its context across many runs (sweeps): only the 2 KB system is re-staged."""
    system = OpticalSystem.from_settings(settings, res_dir)
    own = tracer is None
    if own:
        tracer = RayTracer(system, device=device)
    else:
        tracer.set_system(system)
    try:
        res = tracer.run()
        folder = os.path.join(data_dir, settings.data_folder)
        os.makedirs(folder, exist_ok=True)                  # setupMod.f90:124-131
        if settings.use_tracker:                            # main.f90:72-74, :121-124
            write_tracker_files(tracer, system, folder)
    finally:
        if own:
            tracer.close()
    write_outputs(system, res, data_dir, verbose)
    return res


def write_outputs(system: OpticalSystem, res: RunResult, data_dir: str = "data", verbose: bool = True) -> None:
    """What `program raytrace` leaves behind for one simulation (src/main.f90:168-185): the stats row, the
    two transmission lines, the three image files."""
    settings = system.settings
    folder = os.path.join(data_dir, settings.data_folder)
    os.makedirs(folder, exist_ok=True)                      # setupMod.f90:124-131
    append_stats(folder, system, res)
    if verbose:                                             # main.f90:180-181
        print(f"Ring  transmitted:  {res.ring_transmitted:8.2f}%")
        print(f"Point transmitted:  {res.point_transmitted:8.2f}%")
    if settings.make_images and not settings.use_tracker:   # main.f90:183-185; the tracker
        write_images(res.image, os.path.join(folder, output_basename(system) + "_image"))   # deselects images (setupMod.f90:76-82)


def run_settings_file(settings_path: str, res_dir: Optional[str] = None, data_dir: str = "data",
                      device: int = 0, verbose: bool = True,
                      tracer: Optional[RayTracer] = None) -> RunResult:
    """`bin/raytrace <settings>` for the hot path on one GPU: read, trace, write (src/main.f90)."""
    settings = Settings.from_file(settings_path)
    res_dir = res_dir or os.path.dirname(os.path.abspath(settings_path))
    return run_settings(settings, res_dir, data_dir, device, verbose, tracer)
