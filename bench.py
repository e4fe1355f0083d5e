#!/usr/bin/env python3
"""bench.py — ray-surface intersections / second of the MI355X trace path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the ranks are
one process per GPU over RCCL: either the driver starts them (torch.distributed.run sets
WORLD_SIZE / RANK / LOCAL_RANK) or, when it runs the plain command, this script starts them
itself as CHILD processes — before it has made any GPU call — and relays rank 0's line and
their exit code.  Rank 0 prints ONE JSON line.

Workload (default = BASELINE.json configs[1], the configuration `metric` is quoted on):
point source (phase 2 loop, src/main.f90:127-162), clearBottle-large +
planoConvex-f39.9mm + achromaticDoublet-f50.0mm, 1e7 rays per GPU, fp64.
`--workload ring1e8` = configs[2] (ring loop, src/main.f90:90-109, 1e8 rays per GPU per step),
`--workload full1e9` = configs[3]'s shape (both loops, 1e9 rays per layer per step, the rays of a
step sharded over the ranks: strong in N, which is what configs[3] asks for).
A step = one pass of the hot path over one batch: every rank emits, traces and bins its own
shard of the global index range, then image + counters are sum-all-reduced over ranks ONCE per
run of K steps, inside the timed region (RCCL; skipped for N = 1) — as the reference keeps one
shared image for the whole loop (src/main.f90:88-109).
Synthetic input = (config, seed 123456789, global ray index): rays are generated in-kernel
from the key, nothing is read from the host inside the timed region.
value = intersections all ranks evaluated (exact int64 device counter) / wall time.

Order of the legs in one process: CPU baseline (child process, before this process touches
the GPU), set-up (scratch, code objects, ~50 ms of untimed launches that bring the clocks out
of idle — see SETTLE_LAUNCHES), then the exact fp64 leg that `value` reports, then the
informational legs over the same rays: the exact path with strict libm emitters / 53-bit draws /
both (`strict`, `wide`, `strict_wide`), fp32, fast fp64.  Every leg: W untimed warm-up steps,
then exactly K timed steps between two fences.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VEC_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector (non-MFMA) peak
FP64_VEC_PEAK_TFLOPS = 78.65   # fp64 vector = half of it
FLOP_PER_INTERSECTION = 100.0  # + - x / sqrt = 1 each, ~100 per surface solve (an ASSUMED
                               # algorithmic figure; the executed instruction count is in `roofline`)
FLOP_PER_RING_EMISSION = 61.0  # ~50 flop + 5 sqrt + 6 divisions (2 sincos not counted)
BYTES_PER_RAY = 48.0           # SURVEY §8(d) contract layout: 6 x fp64 ray state resident in HBM
BYTES_PER_BINNED = 8.0         # int32 atomic read-modify-write

WORKLOADS = {
    # name: (phases per step, rays per GPU per step or None = rays per layer sharded over ranks, text)
    "point1e7": ((2,), 10_000_000, "point source (phase 2), clearBottle-large + planoConvex-f39.9mm + "
                                   "achromaticDoublet-f50.0mm, 1e7 rays per GPU (BASELINE configs[1])"),
    "ring1e8": ((1,), 100_000_000, "ring source (phase 1), clearBottle-large + planoConvex-f39.9mm + "
                                   "achromaticDoublet-f50.0mm, 1e8 rays per GPU (BASELINE configs[2])"),
    "full1e9": ((1, 2), None, "ring + point layers, full stack (bottle + plano-convex + doublet), 1e9 rays "
                              "per layer per step sharded over the ranks (BASELINE configs[3] shape)"),
}


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (never exec:
    this process may not be replaced once a GPU runtime is loaded, and it has loaded none yet)."""
    import torch                                    # device_count() does not initialise the GPU
    have = torch.cuda.device_count()
    if have < args.gpus and not (args.rehearse and have >= 1):
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s); "
                         "one rank per GPU is the only supported layout\n")
        return 2
    with socket.socket() as s:                      # a free rendezvous port on the loopback
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stdout.write(p.stdout)
    if p.returncode != 0:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank run exited with {p.returncode}\n")
    return p.returncode


def single_process(args, phases, rays_gpu, text) -> int:
    """`--gpus N --single-process`: ONE process drives N contexts (one per device) and sums image + counters
    with ort_allreduce (RCCL inside the library) — the layout of a Fortran host that owns all GPUs of a node
    (src/main.f90:88 + src/imageMod.f90:55-56 across devices).  Same steps, same rays,
    same JSON line as the one-process-per-GPU layout; only the exact fp64 leg is run."""
    import torch                                    # first: libort_hip.so must bind to torch's HIP runtime
    from opticalraytrace_amd import capi
    from opticalraytrace_amd.capi import C_BINNED_POINT, C_BINNED_RING, C_ISECT_POINT, C_ISECT_RING, Context
    from opticalraytrace_amd.params import Settings
    from opticalraytrace_amd.system import OpticalSystem
    from opticalraytrace_amd.tracer import DEFAULT_SEED, shard_range
    world = args.gpus
    have = capi.device_count()
    if have < world:
        sys.stderr.write(f"bench.py: --gpus {world} --single-process but this node shows {have} GPU(s); "
                         "one context per GPU is the only supported layout\n")
        return 2
    strong = rays_gpu is None
    total_rays = 1_000_000_000 if strong else rays_gpu * world
    system = OpticalSystem.from_settings(Settings(nphotons=min(total_rays, 2**31 - 1), make_images=True,
                                                  bottle_file="clearBottle-large.params", L2_file="planoConvex-f39.9mm.params",
                                                  L3_file="achromaticDoublet-f50.0mm.params"))
    streams = [torch.cuda.Stream(device=g) for g in range(world)]
    ctxs = [Context(system, device=g, stream=streams[g].cuda_stream) for g in range(world)]
    shards = [shard_range(total_rays, g, world) for g in range(world)]
    for c, (lo, cnt) in zip(ctxs, shards):
        c.set_timing(True)
        c.reserve(cnt)
        for ph in phases:
            c.trace(ph, 0, 64, DEFAULT_SEED)
    settle = max(2 * len(phases), min(160, 1_600_000_000 // shards[0][1]))
    for k in range(settle):
        for c, (lo, cnt) in zip(ctxs, shards):
            c.trace(phases[k % len(phases)], k * cnt, cnt, DEFAULT_SEED)

    pool = None
    if args.host_threads and world > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=world)

    def run(steps, first):
        for c in ctxs:
            c.reset()
        for c in ctxs:
            c.synchronize()
        t0 = time.perf_counter()
        if pool is None:
            for k in range(steps):
                for c, (lo, cnt) in zip(ctxs, shards):
                    for ph in phases:
                        c.trace(ph, (first + k) * total_rays + lo, cnt, DEFAULT_SEED)
        else:                                       # --host-threads: one issuing thread per device (ctypes releases the GIL in the call)
            def issue(j):
                c, (lo, cnt) = ctxs[j], shards[j]
                for k in range(steps):
                    for ph in phases:
                        c.trace(ph, (first + k) * total_rays + lo, cnt, DEFAULT_SEED)
            list(pool.map(issue, range(world)))
        run.issue_s = time.perf_counter() - t0
        for c in ctxs:
            c.flush()
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        with torch.cuda.device(0):
            ev[0].record(streams[0])
        if world > 1 or args.force_dist:
            capi.allreduce(ctxs)
        with torch.cuda.device(0):
            ev[1].record(streams[0])
        for c in ctxs:
            c.synchronize()
        return time.perf_counter() - t0, ev[0].elapsed_time(ev[1])

    run(args.warmup, 0)
    elapsed, reduce_ms = run(args.steps, args.warmup)
    img, cnt8 = ctxs[0].read()                                  # the global sums (every context holds them)
    # the reduce proves itself (outside the timed region): one extra step sharded over the devices and all-reduced ==
    # the same global range traced by device 0 alone, image and all 8 counters (src/main.f90:88, src/imageMod.f90:55-56)
    reduce_verified = None
    if world > 1 or args.force_dist:
        import numpy as np
        kv = args.warmup + args.steps
        for c in ctxs:
            c.reset()
        for c, (lo, cnt) in zip(ctxs, shards):
            for ph in phases:
                c.trace(ph, kv * total_rays + lo, cnt, DEFAULT_SEED)
        capi.allreduce(ctxs)
        sums = [c.read() for c in ctxs]
        ctxs[0].reset()
        for ph in phases:
            ctxs[0].trace(ph, kv * total_rays, total_rays, DEFAULT_SEED)
        alone = ctxs[0].read()
        reduce_verified = all(np.array_equal(si, alone[0]) and np.array_equal(sc, alone[1]) for si, sc in sums)
        if not reduce_verified:
            sys.stderr.write("bench.py: the all-reduced image of the device shards differs from device 0's trace of the same rays\n")
    isect = sum(int(cnt8[C_ISECT_RING if ph == 1 else C_ISECT_POINT]) for ph in phases)
    for ph in phases:
        assert int(img[ph - 1].sum()) == int(cnt8[C_BINNED_RING if ph == 1 else C_BINNED_POINT]), "image and counter disagree"
    culled = sum(c.work_counters()[0] for c in ctxs)
    kms = ctxs[0].kernel_times(min(args.steps * len(phases), 64))
    cnt0 = shards[0][1]
    kernels_per_call = -(-cnt0 // capi.MAX_RAYS_PER_LAUNCH)
    k_s = (sum(kms) / len(kms)) * 1e-3 / kernels_per_call
    launches = args.steps * world * len(phases) * kernels_per_call
    ring_share = sum(1 for ph in phases if ph == 1) / len(phases)
    alg = FLOP_PER_INTERSECTION * isect / launches + FLOP_PER_RING_EMISSION * (cnt0 / kernels_per_call) * ring_share
    exe = alg - (FLOP_PER_INTERSECTION + FLOP_PER_RING_EMISSION) * culled / launches
    print(json.dumps({
        "metric": "ray-surface intersections/sec", "value": isect / elapsed, "unit": "intersections/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "value_executed": (isect - culled) / elapsed, "reduce_ms": reduce_ms,
        # wall of the issue loop / ort_trace calls in it: ONE host thread serves all devices unless --host-threads
        "host_us_per_trace_call": run.issue_s / (args.steps * world * len(phases)) * 1e6, "host_threads": world if pool else 1,
        "reduce_verified": reduce_verified, "ranks_seen": capi.allreduce_ranks() if (world > 1 or args.force_dist) else 1,
        "kernel_ms_per_step_over_ranks": (lambda v: {"min": min(v), "max": max(v)})(
            [sum(k) / len(k) * len(phases) for k in (c.kernel_times(min(args.steps * len(phases), 64)) for c in ctxs)]),
        "config": {"workload": text, "name": args.workload, "host": "single-process: one context per device, ort_allreduce",
                   "rays_per_layer_per_step": total_rays, "rays_per_gpu_per_launch": cnt0, "phases": list(phases),
                   "seed": DEFAULT_SEED, "intersections_per_step": isect / args.steps, "settle_launches": settle,
                   "build_id": capi.build_id()},
        "roofline": {"bound": "valu_fp64", "achieved": alg / k_s / 1e12, "peak": FP64_VEC_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": alg / k_s / 1e12 / FP64_VEC_PEAK_TFLOPS, "frac_executed": exe / k_s / 1e12 / FP64_VEC_PEAK_TFLOPS,
                     "traffic": None, "kernel_ms": k_s * 1e3, "note": "device 0's launches; see the default layout's line for counters"},
    }), flush=True)
    for c in ctxs:
        c.close()
    return 3 if reduce_verified is False else 0


def cpu_baseline(rays: int, program_runs: int = 0):
    """Time the CPU checker on a bounded sample of the same workload (rank 0, N=1 only).

    Runs in a child process (its OpenMP runtime stays out of this one).  Prefers
    oracle/_ref (the reference's own Fortran path compiled with flang: kind
    "reference"), else the C restatement (kind "port")."""
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"),
                              "--rays", str(rays), "--program-runs", str(program_runs)], capture_output=True, text=True, timeout=1200)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:                                  # the baseline is reported, never required
        return {"value": None, "unit": "intersections/s", "cores": 0, "kind": "port",
                "sample": f"failed: {e!r}"}


def ring_cull_fraction(system) -> float:
    """Share of the ring rays whose lens-disc sample lies outside the plano-convex aperture (what the ring
    programs' segment 0 culls when the geometry allows it: csrc/ort_hip.hip ring_cull_threshold)."""
    from opticalraytrace_amd import capi
    c = capi.pack_system(system)
    ap = c.surfaces[0][0].aperture
    return max(0.0, 1.0 - ap * ap * (1.0 + 1e-6) / c.ring_lens_r2) if ap > 0 and c.ring_lens_r2 > 0 else 0.0


def profiled_counters(workload: str, build_id: str):
    """Counters of the committed rocprofv3 --pmc passes (profiles/pmc_per_launch.json, written by
    tools/stamp_profiles.py) — only when they were taken on THIS build of the kernels."""
    path = os.path.join(ROOT, "profiles", "pmc_per_launch.json")
    try:
        prof = json.load(open(path))
    except Exception:
        return None, "profiles/pmc_per_launch.json missing"
    if prof.get("build_id") != build_id:
        return None, (f"profiles/pmc_per_launch.json was taken on build {prof.get('build_id')}, this "
                      f"library is {build_id}: counters not quoted")
    return prof.get("workloads", {}).get(workload), None


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="point1e7")
    ap.add_argument("--rays", type=int, default=0, help="override: rays per GPU per step")
    ap.add_argument("--phase", type=int, default=0, help="override: 1 = ring loop, 2 = point loop")
    ap.add_argument("--cpu-rays", type=int, default=50_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast", action="store_true", help="skip the informational fast-fp64 leg")
    ap.add_argument("--no-fp32", action="store_true", help="skip the informational fp32 leg (configs[4])")
    ap.add_argument("--no-strict", action="store_true", help="skip the strict-libm / 53-bit-draw legs of the exact fp64 path")
    ap.add_argument("--sweep", action="store_true",
                    help="also time runner.py's lens experiment (75 systems): batched / one by one / process per simulation, "
                         "at 1e6 and 1e9 photons; adds the `sweep` object (several extra seconds)")
    ap.add_argument("--single-process", action="store_true",
                    help="one process, one context per GPU, ort_allreduce (the Fortran host's layout, INTEGRATION.md §C)")
    ap.add_argument("--host-threads", action="store_true",
                    help="--single-process: issue every device's launches from its own host thread (the C ABI serialises per context)")
    ap.add_argument("--rehearse", action="store_true",
                    help="development: run the N-rank flow on ONE GPU (every rank on device 0, gloo instead of RCCL); "
                         "the line says so in config.rehearsal and is not a scaling measurement")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and all-reduce even with one rank (exercises the N>1 code path on a 1-GPU box)")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        ap.error("--gpus >= 1, --steps >= 1, --warmup >= 0")

    if args.single_process:
        phases, rays_gpu, text = WORKLOADS[args.workload]
        return single_process(args, (args.phase,) if args.phase else phases, args.rays or rays_gpu, text)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}\n")
        return 2

    phases, rays_gpu, text = WORKLOADS[args.workload]
    if args.phase:
        phases = (args.phase,)
    if args.rays:
        rays_gpu = args.rays
    strong = rays_gpu is None
    total_rays = 1_000_000_000 if strong else rays_gpu * world        # per layer per step, all ranks
    custom = bool(args.phase or args.rays)
    if custom:
        text = (f"phase(s) {phases}, clearBottle-large + planoConvex-f39.9mm + achromaticDoublet-f50.0mm, "
                f"{total_rays} rays per layer per step over {world} rank(s)")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.force_dist:
        cpu = cpu_baseline(args.cpu_rays, 2 if args.sweep else 0)          # before the GPU is touched by this process

    import torch
    import torch.distributed as dist
    from opticalraytrace_amd import capi
    from opticalraytrace_amd.capi import C_BINNED_POINT, C_BINNED_RING, C_ISECT_POINT, C_ISECT_RING
    from opticalraytrace_amd.params import Settings
    from opticalraytrace_amd.system import OpticalSystem
    from opticalraytrace_amd.tracer import DEFAULT_SEED, RayTracer, shard_range

    if not torch.cuda.is_available():
        sys.stderr.write("bench.py needs an MI355X: the trace path has no CPU fallback\n")
        return 2
    if args.rehearse:
        local_rank = 0                              # every rank on the one GPU
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    settings = Settings(nphotons=min(total_rays, 2**31 - 1), make_images=True,
                        bottle_file="clearBottle-large.params",
                        L2_file="planoConvex-f39.9mm.params",
                        L3_file="achromaticDoublet-f50.0mm.params")
    system = OpticalSystem.from_settings(settings)
    tracer = RayTracer(system, device=local_rank, rank=rank, world=world)
    ctx = tracer.ctx
    ctx.set_timing(True)
    lo, cnt = shard_range(total_rays, rank, world)
    # set-up, not a step: scratch for launches of this size, and the kernels' code objects loaded
    # by a 64-ray launch in every arithmetic the legs use (a kernel's first launch otherwise pays ~2 ms)
    ctx.reserve(cnt)
    for prec in (0, 1, 2):
        ctx.set_precision(prec)
        for ph in phases:
            ctx.trace(ph, 0, 64, DEFAULT_SEED)
    ctx.set_precision(0)

    # step k traces the global ray indices [k*T, (k+1)*T) of each of its phases (T = rays per layer
    # per step over all ranks), this rank its contiguous shard of them
    def step(k):
        for ph in phases:
            ctx.trace(ph, k * total_rays + lo, cnt, DEFAULT_SEED)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_run(steps, warmup):
        """W untimed steps, then exactly K steps + the reduce between two fences; max over ranks.
        The timed steps are steps [W_main, W_main + K) of the global index range in every leg, so the
        legs trace the same rays and their images can be compared."""
        tracer.reset()
        for k in range(warmup):
            step(k)
        tracer.reduce(force=use_dist)
        fence()
        tracer.reset()
        fence()
        w0 = ctx.work_counters()
        t0 = time.perf_counter()
        for k in range(steps):
            step(args.warmup + k)
        timed_run.issue_s = time.perf_counter() - t0   # the host's share: K x phases ort_trace calls issued, nothing waited for
        tracer.flush()                              # the group's literal re-run + the fold, then the reduce
        if use_dist:                                # reduce_ms: an event pair around the all-reduce alone
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        tracer.reduce(force=use_dist)
        if use_dist:
            ev[1].record()
        fence()
        el = time.perf_counter() - t0
        timed_run.reduce_ms = ev[0].elapsed_time(ev[1]) if use_dist else 0.0
        w1 = ctx.work_counters()
        work = torch.tensor([w1[0] - w0[0], w1[1] - w0[1]], dtype=torch.int64, device="cuda")
        if use_dist:
            dist.all_reduce(work)
        timed_run.culled, timed_run.deferred = (int(x) for x in work.tolist())
        if use_dist:
            tmax = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        # per-launch durations from the HIP events each ort_trace recorded on the tracer's stream
        kms = ctx.kernel_times(min(steps * len(phases), 64))
        return el, kms

    # ---- the COLD figure: exactly what `--warmup W --steps K` give straight after the priming launches,
    # i.e. without the settle launches below.  The process has just spent seconds on the CPU (baseline,
    # imports): the GPU clocks are still ramping through these steps.  Reported as ms_per_step_cold /
    # value_cold next to the steady-clock figure `value`.
    cold_elapsed, _ = timed_run(args.steps, args.warmup)
    cold_res = tracer.result(total_rays * args.steps)
    cold_isect = sum(int(cold_res.counters[C_ISECT_RING if ph == 1 else C_ISECT_POINT]) for ph in phases)

    # ... and the clocks: after an idle period (this process has just spent seconds in the CPU baseline
    # and in imports) the first ~30 ms of GPU work run at lower clocks (tools/rampbench.py,
    # profiles/r02/ramp.log: 0.446 ms for the first 64 launches, 0.406 from then on).  A production run
    # of this path is >= 40 ms of back-to-back launches per 1e9-ray layer, so steady clocks are the
    # regime the metric is about: SETTLE_LAUNCHES (160 at 1e7 rays) untimed launches of the step's own size come first,
    # for every leg alike.  They are set-up like the priming launches above — not among the W warm-up
    # steps, not in the timed region — and are reported in config.settle_launches.
    SETTLE_LAUNCHES = max(2 * len(phases), min(160, 1_600_000_000 // cnt))   # launches of the step's own size: ~50 ms
    for k in range(SETTLE_LAUNCHES):
        ctx.trace(phases[k % len(phases)], k * cnt, cnt, DEFAULT_SEED)
    ctx.synchronize()

    def isect_binned(res):
        i = sum(int(res.counters[C_ISECT_RING if ph == 1 else C_ISECT_POINT]) for ph in phases)
        b = sum(int(res.counters[C_BINNED_RING if ph == 1 else C_BINNED_POINT]) for ph in phases)
        return i, b

    # ---- the leg `value` reports: exact fp64 — first, straight after the set-up ------------------
    elapsed, kernel_ms = timed_run(args.steps, args.warmup)
    reduce_ms, culled_total, deferred_total = timed_run.reduce_ms, timed_run.culled, timed_run.deferred
    host_us_per_call = timed_run.issue_s / (args.steps * len(phases)) * 1e6
    res = tracer.result(total_rays * args.steps)   # counters of the whole timed run, summed over ranks
    isect_total, binned_total = isect_binned(res)
    for ph in phases:
        want = int(res.counters[C_BINNED_RING if ph == 1 else C_BINNED_POINT])
        assert int(res.image[ph - 1].sum()) == want, "image and counter disagree"
    isect_per_step = isect_total / args.steps
    binned_per_step = binned_total / args.steps
    value = isect_total / elapsed

    # ---- informational legs (outside `value`): the same rays in the other arithmetics -----------
    # strict / wide / strict_wide: the exact fp64 path with the reference's own bits in front of the surfaces — glibc's
    # sin / cos / sincos in the emitters (kernel variant bit 6, src/sourceMod.f90:12-47, :250-300 through the platform libm)
    # and 53-bit uniforms (bit 5, src/random_mod.f90:39-46) — as template flags of the same surface programs
    legs = {}
    for name, prec, variant, skip in (("strict", 0, 1 | 64, args.no_strict), ("wide", 0, 1 | 32, args.no_strict),
                                      ("strict_wide", 0, 1 | 32 | 64, args.no_strict),
                                      ("fp32", 1, 1, args.no_fp32), ("fast_fp64", 2, 1, args.no_fast)):
        if skip:
            continue
        ctx.set_kernel_variant(1)
        ctx.set_precision(prec)
        ctx.set_kernel_variant(variant)
        for ph in phases:                                   # the variant's code object (set-up, as above)
            ctx.trace(ph, 0, 64, DEFAULT_SEED)
        el, kms = timed_run(args.steps, args.warmup)
        r = tracer.result(total_rays * args.steps)
        legs[name] = (el, kms, r, timed_run.culled, ctx.last_kernel_name())
    ctx.set_kernel_variant(1)
    ctx.set_precision(0)

    # ---- N > 1 proves itself (outside every timed region): ONE extra step traced the way the timed steps are — every rank
    # its shard, then the all-reduce — against the SAME global range traced by rank 0 alone.  Keyed draws make the two
    # identical bin for bin if and only if the shards tile the range and the reduce sums every rank exactly once
    # (reference: the shared image + reduction(+:rcount,pcount) of src/main.f90:88, src/imageMod.f90:55-56).
    reduce_verified, per_rank_ms = None, None
    if use_dist:
        kv = args.warmup + args.steps                  # a step of the global index range no leg has traced
        tracer.reset()
        step(kv)
        tracer.reduce(force=True)
        fence()
        summed = tracer.result(total_rays)
        tracer.reset()
        fence()
        if rank == 0:
            for ph in phases:
                ctx.trace(ph, kv * total_rays, total_rays, DEFAULT_SEED)
        alone = tracer.result(total_rays)              # (no reduce: what THIS rank traced)
        ok = True
        if rank == 0:
            import numpy as np
            ok = bool(np.array_equal(summed.image, alone.image) and np.array_equal(summed.counters, alone.counters))
            if not ok:
                sys.stderr.write(f"bench.py: the reduced image of {world} shards differs from the single-rank trace of the same rays "
                                 f"(L1 {int(abs(summed.image.astype('int64') - alone.image.astype('int64')).sum())}; counters "
                                 f"{summed.counters.tolist()} vs {alone.counters.tolist()})\n")
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        reduce_verified = bool(int(flag.item()))
        tracer.reset()
        # per-rank step time of the timed fp64 leg (mean launch x launches per step): min / max over the ranks
        mine = torch.tensor([sum(kernel_ms) / len(kernel_ms) * len(phases)], dtype=torch.float64, device="cuda")
        lo_t, hi_t = mine.clone(), mine.clone()
        dist.all_reduce(lo_t, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_t, op=dist.ReduceOp.MAX)
        per_rank_ms = {"min": float(lo_t.item()), "max": float(hi_t.item())}

    # ---- roofline of the dominant kernel (the fused trace kernel), per launch = per rank per phase
    # of a step.  kernel_ms holds one entry per ort_trace call (queued kernel + its literal re-run
    # launch + fold_kernel, one event bracket); a step of a two-phase workload is two launches.
    # ... and an ort_trace call of more than ORT_MAX_RAYS_PER_LAUNCH rays is several kernel launches.
    kernels_per_call = -(-cnt // capi.MAX_RAYS_PER_LAUNCH)
    launches_per_step = len(phases) * kernels_per_call
    call_s = (sum(kernel_ms) / len(kernel_ms)) * 1e-3               # mean ort_trace call, s
    k_s = call_s / kernels_per_call                                 # mean kernel launch, s
    rays_launch = cnt / kernels_per_call
    isect_launch = isect_per_step / world / launches_per_step
    binned_launch = binned_per_step / world / launches_per_step
    ring_share = sum(1 for ph in phases if ph == 1) / len(phases)     # share of the launches that emit ring rays
    alg_flop = FLOP_PER_INTERSECTION * isect_launch + FLOP_PER_RING_EMISSION * rays_launch * ring_share
    contract_bytes = BYTES_PER_RAY * rays_launch + BYTES_PER_BINNED * binned_launch
    ach_tf = alg_flop / k_s / 1e12
    # EXECUTED work (ring loop): a ray that segment 0 counts at the first aperture costs one hash and one
    # compare — no surface solve, no emission — so it is priced at zero flop and leaves both counts
    culled_launch = culled_total / args.steps / world / launches_per_step
    isect_exec_per_step = isect_per_step - culled_total / args.steps
    exec_flop = (FLOP_PER_INTERSECTION * (isect_launch - culled_launch)
                 + FLOP_PER_RING_EMISSION * (rays_launch * ring_share - culled_launch))
    exec_tf = exec_flop / k_s / 1e12
    build = capi.build_id()
    prof, prof_note = profiled_counters(args.workload if not custom else "custom", build)
    prof = prof or {}
    traffic = prof.get("hbm_bytes_per_launch")

    def leg_counters(leg):
        """executed-work counters of an informational leg's kernel, from the committed PMC passes of this build"""
        p, _ = profiled_counters(f"{args.workload}_{leg}" if not custom else "custom", build)
        p = p or {}
        vi, ni = p.get("valu_instructions_per_launch"), p.get("intersections_per_launch")
        return {"valu_busy": p.get("valu_busy_frac"), "valu_lane_utilisation": p.get("valu_lane_utilisation"),
                # (SQ_ACTIVE_INST_VALU x 4 cycles / SIMD cycles: a wave64 fp32 instruction occupies its SIMD for fewer than
                # the 4 cycles of an fp64 one, so the fp32 leg reads above 1)
                "valu_busy_unit": "4-cycle issue slots per SIMD cycle/4",
                "valu_instr_per_intersection": vi * 64.0 / ni if vi and ni else None,
                "instruction_classes_per_launch": {k: p[k] for k in ("fma_f32", "mul_f32", "add_f32", "trans_f32", "int32", "fma_f64",
                                                                       "mul_f64", "add_f64", "trans_f64") if k in p} or None}

    def fp_roofline(kms, r, culled, peak, bound, kernels):
        """(every queued call is cut into launches of at most 2^27 rays, whatever the arithmetic)"""
        ks = (sum(kms) / len(kms)) * 1e-3 / kernels
        i, _ = isect_binned(r)
        per_call = 1.0 / args.steps / world / len(phases)
        fl = (FLOP_PER_INTERSECTION * i * per_call + FLOP_PER_RING_EMISSION * cnt * ring_share) / kernels
        fx = (FLOP_PER_INTERSECTION * (i - culled) * per_call + FLOP_PER_RING_EMISSION * (cnt * ring_share - culled * per_call)) / kernels
        return {"bound": bound, "achieved": fl / ks / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": fl / ks / 1e12 / peak, "achieved_executed": fx / ks / 1e12, "frac_executed": fx / ks / 1e12 / peak,
                "kernel_ms": ks * 1e3, "kernel_launches_per_step": len(phases) * kernels}

    out = {
        "metric": "ray-surface intersections/sec",
        "value": value,
        "unit": "intersections/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        # the same K steps behind the same W warm-up steps WITHOUT the settle launches (clocks still ramping)
        "ms_per_step_cold": cold_elapsed / args.steps * 1e3,
        "value_cold": cold_isect / cold_elapsed,
        # what was executed: intersections minus the ring rays counted at the first aperture without a surface
        # solve (segment 0 of the ring programs; 0 for the point loop), and the all-reduce on its own
        "value_executed": isect_exec_per_step * args.steps / elapsed,
        "reduce_ms": reduce_ms,
        # wall time of the loop that ISSUES the timed steps / ort_trace calls in it (nothing is waited for inside it; when the
        # runtime's launch queue fills, it includes the wait for a free slot): what the host costs per call — at 8 GPUs the
        # per-GPU step of configs[1] is 0.3 ms, so this must stay far below it
        "host_us_per_trace_call": host_us_per_call,
        # N > 1 (or --force-dist): one extra sharded + reduced step == the same rays on rank 0 alone, image and 8 counters
        "reduce_verified": reduce_verified,
        "ranks_seen": (dist.get_world_size() if use_dist else 1),
        "kernel_ms_per_step_over_ranks": per_rank_ms,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": text, "name": "custom" if custom else args.workload,
                   "rays_per_layer_per_step": total_rays, "rays_per_gpu_per_launch": cnt,
                   "phases": list(phases), "seed": DEFAULT_SEED,
                   "sharding": f"contiguous global ray ranges over {world} rank(s); one RCCL sum of "
                               "image+counters per run of K steps, inside the timed region",
                   "intersections_per_step": isect_per_step, "binned_per_step": binned_per_step,
                   "intersections_executed_per_step": isect_exec_per_step,
                   "ring_rays_culled_per_step": culled_total / args.steps,
                   "rays_deferred_to_literal_rerun_per_step": deferred_total / args.steps,
                   "rays_per_s": total_rays * len(phases) * args.steps / elapsed,
                   "settle_launches": SETTLE_LAUNCHES,
                   **({"rehearsal": "all ranks on device 0, gloo instead of RCCL: the N-rank flow, not a scaling measurement"}
                      if args.rehearse else {}),
                   "build_id": build},
        # the BINDING bound of this path: fp64 vector-ALU issue (no MFMA: there is no contraction)
        "roofline": {
            "bound": "valu_fp64", "achieved": ach_tf, "peak": FP64_VEC_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": ach_tf / FP64_VEC_PEAK_TFLOPS,
            # the same with executed work only (culled ring rays priced at zero): equals `frac` for the point loop
            "achieved_executed": exec_tf, "frac_executed": exec_tf / FP64_VEC_PEAK_TFLOPS,
            "executed_flop_per_launch": exec_flop,
            "traffic": traffic,                                   # HBM bytes per launch, rocprofv3 PMC (or null)
            "kernel": "trace_queue_kernel<MODE_FUSED, filtered, surface program> (one launch per <= 2^27 rays, timed by the "
                      "event pair the launch itself carries; the literal re-run of deferred rays — normally none — runs "
                      "once per group of launches, fold_kernel once per run when the image is read)",
            "kernel_ms": k_s * 1e3,                               # mean duration of ONE kernel launch
            "kernel_launches_per_step": launches_per_step,        # phases x launches of <= 2^27 rays per ort_trace call
            "rays_per_kernel_launch": rays_launch,
            "flop_per_intersection": FLOP_PER_INTERSECTION,
            "flop_per_ring_emission": FLOP_PER_RING_EMISSION,
            # ring loop only: the share of its rays that end at the first aperture by their third draw alone and
            # are counted (lost, one surface solve) without being emitted — DESIGN §3.8; radius^2 / (radius + 10 mm)^2
            "ring_rays_culled_fraction": ring_cull_fraction(system) if 1 in phases else None,
            "algorithmic_flop_per_launch": alg_flop,
            # COUNTED instead of assumed: 64 x (ADD_F64 + MUL_F64 + 2 FMA_F64) wave-instructions of this very build's PMC
            # pass (profiles/pmc_per_launch.json; null when the committed counters are another build's) over the same
            # mean launch duration — what the vector units really delivered, IEEE division / sqrt expansions included
            "counted_flop_per_launch": prof.get("counted_fp64_flop_per_launch"),
            "frac_counted": (prof["counted_fp64_flop_per_launch"] * (rays_launch / prof["rays_per_launch_nominal"] if prof.get("rays_per_launch_nominal") else 1.0)
                             / k_s / 1e12 / FP64_VEC_PEAK_TFLOPS if prof.get("counted_fp64_flop_per_launch") and len(phases) == 1 else None),   # (two loops: kernel_ms is the mean over both kernels, the counters are the point kernel's)
            "non_arithmetic_share_of_valu": prof.get("non_arithmetic_share_of_valu"),   # compares, selects, moves, integer (the draw), conversions
            "sq_insts_salu_per_launch": prof.get("salu_instructions_per_launch"),
            "note": "algorithmic flop = an assumed 100 per surface solve (SURVEY §8d; + 61 per emitted ring "
                    "ray in the ring loop) / mean launch duration (HIP events on the context's stream)"
                    + ("; ring loop: the rays whose lens-disc draw already puts them outside the first aperture "
                       "(69 % for this lens) are counted there — one surface solve, lost — without being emitted "
                       "(DESIGN §3.8); the algorithmic figure prices them as the reference executes them, so this "
                       "fraction includes work avoided, not only work done faster" if 1 in phases else ""),
            # executed work, from the committed PMC passes of this very build (null otherwise)
            "valu_busy": prof.get("valu_busy_frac"),
            "valu_lane_utilisation": prof.get("valu_lane_utilisation"),
            # lane-instructions per surface solve = wave-level VALU instructions x 64 lanes / intersections
            "valu_instr_per_intersection": (prof["valu_instructions_per_launch"] * 64.0 / prof["intersections_per_launch"]
                                            if prof.get("valu_instructions_per_launch") and prof.get("intersections_per_launch") else None),
            "profile_note": prof_note,
        },
        # SURVEY §8(d)'s contract figure: the bytes the north-star layout (ray state resident in HBM)
        # WOULD move.  The fused kernel keeps a ray in registers from emission to binning and never
        # moves the 48 B/ray; shown for the contract only.
        "roofline_hbm_contract": {
            "bound": "hbm", "achieved": contract_bytes / k_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": contract_bytes / k_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": contract_bytes,
            "notional": True,
        },
        "roofline_hbm_measured": ({
            "bound": "hbm", "achieved": traffic / k_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": traffic / k_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_write_bytes": BYTES_PER_BINNED * binned_launch,
            "write_amplification": (prof.get("write_bytes_per_launch") / (BYTES_PER_BINNED * binned_launch)
                                    if prof.get("write_bytes_per_launch") and binned_launch else None),
            "note": "each scattered int32 atomic is one 32-byte memory-side write",
        } if traffic else None),
    }
    for name, text_ in (("strict", "kernel variant bit 6: the light sources call glibc 2.35's own sin / cos / sincos — every emitted ray, state, "
                                   "image and counter equals the reference's bit for bit (tests: budget 0 at full size)"),
                        ("wide", "kernel variant bit 5: 53-bit uniforms (ORT-RNG-v2w, one hash per draw) — another stream, the same statistics"),
                        ("strict_wide", "bits 5 + 6: glibc emitters on 53-bit uniforms")):
        if name not in legs:
            continue
        el, kms, r, cul, kname = legs[name]
        i_s, b_s = isect_binned(r)
        out[name] = {
            "value": i_s / el, "unit": "intersections/s", "ms_per_step": el / args.steps * 1e3, "dtype": "f64",
            "roofline": fp_roofline(kms, r, cul, FP64_VEC_PEAK_TFLOPS, "valu_fp64", kernels_per_call),
            "kernel": kname, "vs_default": (i_s / el) / value,
            "image_l1_vs_default": (int(abs(r.image.astype("int64") - res.image.astype("int64")).sum()) if name == "strict" else None),
            "binned": b_s, "intersections": i_s, "note": text_,
        }
    if "fp32" in legs:
        el, kms, r, cul, _ = legs["fp32"]
        i32, b32 = isect_binned(r)
        out["fp32"] = {
            "value": i32 / el, "unit": "intersections/s", "ms_per_step": el / args.steps * 1e3, "dtype": "f32",
            "roofline": {**fp_roofline(kms, r, cul, FP32_VEC_PEAK_TFLOPS, "valu_fp32", kernels_per_call), **leg_counters("fp32")},
            "image_l1_vs_exact": int(abs(r.image.astype("int64") - res.image.astype("int64")).sum()),
            "image_l1_vs_exact_frac_of_binned": float(abs(r.image.astype("int64") - res.image.astype("int64")).sum()) / max(binned_total, 1),
            "binned": b32, "binned_exact": binned_total, "intersections": i32, "intersections_exact": isect_total,
            "note": "ort_set_precision(1), BASELINE configs[4]: the path in single precision — hardware rcp / sqrt / rsq, fused "
                    "multiply-adds, the cheap decision forms without margins or deferrals; uniforms = top 24 bits of the same "
                    "draws; tests/test_gpu_fp32.py holds the tolerance study.  The point loop's hits are logged by the trace kernel "
                    "and binned in LDS by bin_log_kernel once per several launches and at the flush inside the timed region "
                    "(the memory-side image atomics bound the fp32 program: profiles/r04/atomics_ab.log); roofline.kernel_ms is "
                    "the trace kernel, ms_per_step holds everything",
        }
    if "fast_fp64" in legs:
        el, kms, r, cul, _ = legs["fast_fp64"]
        i2, _ = isect_binned(r)
        out["fast_fp64"] = {
            "value": i2 / el, "unit": "intersections/s", "ms_per_step": el / args.steps * 1e3,
            "roofline": {**fp_roofline(kms, r, cul, FP64_VEC_PEAK_TFLOPS, "valu_fp64", kernels_per_call), **leg_counters("fast_fp64")},
            "image_l1_vs_exact": int(abs(r.image.astype("int64") - res.image.astype("int64")).sum()),
            "note": "ort_set_precision(2): FMA contraction + Newton reciprocal/rsqrt; ~1e-15 relative from the "
                    "exact path, not bit-identical (tests/test_gpu_fastd.py)",
        }
    if cpu is not None:
        out["cpu_baseline"] = cpu
        if cpu.get("value"):
            out["gpu_over_cpu"] = value / cpu["value"]
    if args.sweep and rank == 0 and world == 1:
        from opticalraytrace_amd.sweeps import lens_experiment_rates
        tracer.close()
        out["sweep"] = {"1e6": lens_experiment_rates(1_000_000, local_rank, process_samples=3),
                        "1e9": lens_experiment_rates(1_000_000_000, local_rank, process_samples=2, repeats=1)}
        if cpu is not None and cpu.get("reference_program"):
            out["sweep"]["reference_program"] = cpu["reference_program"]
    if os.environ.get("ORT_BENCH_DUMP_KERNEL_MS"):               # development: the per-launch series
        out["kernel_ms_series"] = [round(x, 4) for x in kernel_ms]
    if rank == 0:
        print(json.dumps(out), flush=True)
    tracer.close()
    if use_dist:
        dist.destroy_process_group()
    if reduce_verified is False:
        return 3                                    # a wrong sum is a failed run, whatever the throughput
    return 0


if __name__ == "__main__":
    sys.exit(main())
