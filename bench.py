#!/usr/bin/env python3
"""bench.py — ray-surface intersections / second of the MI355X trace path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is
launched under torch.distributed.run, one rank per GPU (RCCL).  Rank 0 prints ONE
JSON line.

Workload (BASELINE.json configs[1], the configuration `metric` is quoted on):
point source (phase 2 loop, src/main.f90:127-162), clearBottle-large +
planoConvex-f39.9mm + achromaticDoublet-f50.0mm, 1e7 rays per GPU, fp64.
A step = one pass of the hot path over one batch: every rank emits, traces and
bins its own 1e7-ray shard of the global index range (weak scaling), then the
image + counters are sum-all-reduced over ranks (RCCL; skipped for N = 1).
Synthetic input = (config, seed 123456789, global ray index): rays are generated
in-kernel from the key, nothing is read from the host inside the timed region.
value = intersections all ranks evaluated (exact int64 device counter) / wall time.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VEC_PEAK_TFLOPS = 78.65   # half of the 157.3 TF fp32 vector peak (MI355X_MICROARCH.md)
FLOP_PER_INTERSECTION = 100.0
BYTES_PER_RAY = 48.0           # 6 x fp64 ray state
BYTES_PER_BINNED = 8.0         # int32 atomic read-modify-write


def cpu_baseline(rays: int):
    """Time the CPU checker on a bounded sample of the same workload (rank 0, N=1 only).

    Runs in a child process (its OpenMP runtime stays out of this one).  Prefers
    oracle/_ref (the reference's own Fortran path compiled with flang: kind
    "reference"), else the C restatement (kind "port")."""
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"),
                              "--rays", str(rays)], capture_output=True, text=True, timeout=600)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:                                  # the baseline is reported, never required
        return {"value": None, "unit": "intersections/s", "cores": 0, "kind": "port",
                "sample": f"failed: {e!r}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=10_000_000, help="rays per GPU per step")
    ap.add_argument("--phase", type=int, default=2, help="2 = point loop (configs[1]); 1 = ring loop")
    ap.add_argument("--cpu-rays", type=int, default=20_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast", action="store_true", help="skip the informational fast-fp64 leg")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and all-reduce even with one rank (exercises the N>1 code path on a 1-GPU box)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run "
                     "(one rank per GPU); see the module docstring")
        args.gpus = world

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_rays)          # before the GPU is touched by this process

    import torch
    import torch.distributed as dist
    from opticalraytrace_amd.capi import C_BINNED_POINT, C_BINNED_RING, C_ISECT_POINT, C_ISECT_RING
    from opticalraytrace_amd.params import Settings
    from opticalraytrace_amd.system import OpticalSystem
    from opticalraytrace_amd.tracer import DEFAULT_SEED, RayTracer

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the trace path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    settings = Settings(nphotons=args.rays * world, make_images=True,
                        bottle_file="clearBottle-large.params",
                        L2_file="planoConvex-f39.9mm.params",
                        L3_file="achromaticDoublet-f50.0mm.params")
    system = OpticalSystem.from_settings(settings)
    tracer = RayTracer(system, device=local_rank, rank=rank, world=world)
    tracer.ctx.set_timing(True)
    # set-up, not a step: scratch for launches of this size, and the kernels' code objects loaded
    # by a 64-ray launch (the first launch of a kernel otherwise pays ~2 ms once)
    tracer.ctx.reserve(args.rays)
    tracer.ctx.trace(args.phase, 0, 64, DEFAULT_SEED)
    phase, total_rays = args.phase, args.rays * world
    ci, cb = (C_ISECT_POINT, C_BINNED_POINT) if phase == 2 else (C_ISECT_RING, C_BINNED_RING)

    # One RUN = warmup + K steps; step k traces the global ray indices [k*T, (k+1)*T) (T = rays
    # per GPU x ranks), sharded over the ranks, accumulating into the per-GPU image.  As in the
    # reference (one shared image for the whole loop, src/main.f90:88-109) the image is reduced
    # ONCE per run — inside the timed region — not once per batch.
    def step(k):
        lo, cnt = (total_rays * rank) // world, (total_rays * (rank + 1)) // world - (total_rays * rank) // world
        tracer.ctx.trace(phase, k * total_rays + lo, cnt, DEFAULT_SEED)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    tracer.reset()
    for k in range(args.warmup):
        step(k)
    tracer.reduce(force=use_dist)
    fence()
    tracer.reset()
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    tracer.reduce(force=use_dist)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # per-launch kernel durations from the HIP events each launch recorded on the tracer's stream
    kernel_ms = tracer.ctx.kernel_times(min(args.steps, 64))

    # informational, outside `value`: the same run in the opt-in fast fp64 mode (csrc/ort_fastd.h)
    fast = None
    if not args.no_fast:
        saved_img, saved_cnt = tracer.image.clone(), tracer.counters.clone()
        tracer.ctx.set_precision(2)
        for k in range(2):
            step(k)
        fence()
        tracer.reset()
        t1 = time.perf_counter()
        for k in range(args.steps):
            step(args.warmup + k)
        tracer.reduce(force=use_dist)
        fence()
        el_fast = time.perf_counter() - t1
        if use_dist:
            tm = torch.tensor([el_fast], dtype=torch.float64, device="cuda")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            el_fast = float(tm.item())
        fr = tracer.result(total_rays * args.steps)
        fast = {"value": int(fr.counters[ci]) / el_fast, "unit": "intersections/s",
                "ms_per_step": el_fast / args.steps * 1e3,
                "image_l1_vs_exact": int((fr.image.astype("int64") - saved_img.cpu().numpy()).__abs__().sum()),
                "note": "ort_set_precision(2): FMA contraction + Newton reciprocal/rsqrt; ~1e-15 relative "
                        "from the exact path, not bit-identical (profiles/r01/fastd_study.json)"}
        tracer.ctx.set_precision(0)
        tracer.image.copy_(saved_img)
        tracer.counters.copy_(saved_cnt)

    res = tracer.result(total_rays * args.steps)   # counters of the whole timed run, summed over ranks
    isect_total = int(res.counters[ci])
    binned_total = int(res.counters[cb])
    assert int(res.image[phase - 1].sum()) == binned_total, "image and counter disagree"
    isect_per_step = isect_total / args.steps
    binned_per_step = binned_total / args.steps
    value = isect_total / elapsed

    # roofline of the dominant kernel (the fused trace kernel), per launch = per rank per step
    k_s = (sum(kernel_ms) / len(kernel_ms)) * 1e-3
    rays_launch = args.rays
    isect_launch = isect_per_step / world
    binned_launch = binned_per_step / world
    alg_bytes = BYTES_PER_RAY * rays_launch + BYTES_PER_BINNED * binned_launch
    alg_flop = FLOP_PER_INTERSECTION * isect_launch
    ach_gbs = alg_bytes / k_s / 1e9
    ach_tf = alg_flop / k_s / 1e12
    traffic = valu_busy = None
    tf = os.path.join(ROOT, "profiles", "traffic_bytes_per_launch.json")
    if os.path.exists(tf):                         # written from the rocprofv3 --pmc passes
        try:
            prof = json.load(open(tf))
            traffic = prof.get("hbm_bytes_per_launch")
            valu_busy = prof.get("valu_busy_frac")
        except Exception:
            traffic = valu_busy = None

    out = {
        "metric": "ray-surface intersections/sec",
        "value": value,
        "unit": "intersections/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "point source (phase 2), clearBottle-large + planoConvex-f39.9mm + "
                               "achromaticDoublet-f50.0mm, 1e7 rays per GPU (BASELINE configs[1])"
                               if phase == 2 and args.rays == 10_000_000 else
                               f"phase {phase}, clearBottle-large + planoConvex-f39.9mm + "
                               f"achromaticDoublet-f50.0mm, {args.rays} rays per GPU",
                   "rays_per_gpu": args.rays, "phase": phase, "seed": DEFAULT_SEED,
                   "sharding": f"contiguous global ray ranges over {world} rank(s); one RCCL sum of "
                               "image+counters per run of K steps, inside the timed region",
                   "intersections_per_step": isect_per_step, "binned_per_step": binned_per_step,
                   "rays_per_s": total_rays * args.steps / elapsed},
        "roofline": {
            "bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic,
            "kernel": "trace_queue_kernel<MODE_FUSED, filtered> (+ the literal re-run launch and fold_kernel, same event bracket)", "kernel_ms": k_s * 1e3,
            "algorithmic_bytes_per_launch": alg_bytes,
            "note": "SURVEY §8(d) contract figure 48 B/ray + 8 B/binned ray; the path is fp64-VALU "
                    "bound, see roofline_fp64 (DESIGN.md §5)",
        },
        "roofline_fp64": {
            "bound": "valu_fp64", "achieved": ach_tf, "peak": FP64_VEC_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": ach_tf / FP64_VEC_PEAK_TFLOPS, "flop_per_intersection": FLOP_PER_INTERSECTION,
            "valu_busy_frac_profiled": valu_busy,     # rocprofv3 PMC of this kernel, profiles/ (not live)
        },
    }
    if fast is not None:
        out["fast_fp64"] = fast
    if cpu is not None:
        out["cpu_baseline"] = cpu
        if cpu.get("value"):
            out["gpu_over_cpu"] = value / cpu["value"]
    if rank == 0:
        print(json.dumps(out))
    tracer.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
