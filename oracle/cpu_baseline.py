#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — CPU baseline leg of bench.py (never the thing shipped).

Times the trace loop of BASELINE configs[1] (point source, clearBottle-large +
planoConvex-f39.9mm + achromaticDoublet-f50.0mm) on the host cores for a bounded
number of rays and prints one JSON object.

  kind "reference": oracle/_ref/libort_ref.so — the reference's own Fortran path
      sources (point, bottle%forward, telescope, makeImage) compiled with flang,
      OpenMP `parallel do` over rays as src/main.f90:83-89, fed ORT-RNG-v2 draws
      (the unmodified reference's random_number is one locked generator under
      flang and does not scale).
  kind "port": oracle/libort_oracle.so — the plain-C restatement, same loop.

Only the loop is timed (no parsing, no file output), as for the GPU.  Each timing runs in a
fresh process with OMP_NUM_THREADS = the physical cores, then every hardware thread, unbound
(OMP_PROC_BIND=false), and the best of them also bound (OMP_PROC_BIND=close, OMP_PLACES=cores); the best is reported with
both counts stated (`cores`, `threads`) and its binding; every figure is the median of `--repeats`
(3) timings of the loop.  A ONE-thread point (a proportionally smaller sample of the same rays) puts
the figure in context: `parallel_efficiency` = best / (cores occupied x the one-thread figure).
Nothing in the reference's loop (src/main.f90:83-89) is changed for any of this: the thread count
and the binding are the OpenMP runtime's environment, as install.sh's -n is.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def physical_cores(cpus):
    """Distinct (socket, core) pairs among the logical CPUs this process may run on."""
    cores, cur = set(), {}
    try:
        for ln in open("/proc/cpuinfo"):
            if ":" in ln:
                k, v = [x.strip() for x in ln.split(":", 1)]
                cur[k] = v
            elif not ln.strip() and cur:
                if int(cur.get("processor", -1)) in cpus:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
    except OSError:
        pass
    return len(cores) or len(cpus)


def cpu_quota():
    """CPU time this process's cgroup may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None: a
    container can show 128 schedulable CPUs and still be capped at a share of them."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def run_once(kind: str, rays: int, phase: int, threads: int, repeats: int, bind: bool = False):
    """The timings of one thread count in a fresh process (the OpenMP team size is fixed at first use)."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close" if bind else "false")
    if bind:
        env["OMP_PLACES"] = "cores"
    else:
        env.pop("OMP_PLACES", None)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", kind, "--rays", str(rays),
                          "--phase", str(phase), "--repeats", str(repeats)], env=env, capture_output=True, text=True,
                         timeout=900)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not line:
        raise RuntimeError(f"{kind} worker failed: {out.stderr[-400:]}")
    return json.loads(line[-1])


def worker(kind: str, rays: int, phase: int, repeats: int):
    import numpy as np  # noqa: F401
    from opticalraytrace_amd.params import Settings, resource_dir
    from opticalraytrace_amd.system import OpticalSystem
    from oracle.binding import Oracle, Reference

    s = Settings(nphotons=rays, make_images=True, bottle_file="clearBottle-large.params",
                 L2_file="planoConvex-f39.9mm.params", L3_file="achromaticDoublet-f50.0mm.params")
    osys = OpticalSystem.from_settings(s)
    orc = Oracle(osys)
    seed = 123456789
    warm = min(1_000_000, rays)                             # first large call pays thread/arena start-up
    if kind == "reference":
        ref = Reference(s, resource_dir())
        ref.trace(phase, 0, warm, seed)
        times = []
        c0 = time.process_time()
        for _ in range(repeats):
            t0 = time.perf_counter()
            ref.trace(phase, 0, rays, seed)
            times.append(time.perf_counter() - t0)
        cpu = time.process_time() - c0
        _, c = orc.trace(phase, 0, rays, seed)              # untimed: the oracle counts the intersections
    else:
        orc.trace(phase, 0, warm, seed)
        times = []
        c0 = time.process_time()
        for _ in range(repeats):
            t0 = time.perf_counter()
            _, c = orc.trace(phase, 0, rays, seed)
            times.append(time.perf_counter() - t0)
        cpu = time.process_time() - c0
    # CPU seconds of all threads per second of wall time over the timed loops = the cores the OS really gave
    print(json.dumps({"seconds": sorted(times)[len(times) // 2], "all_seconds": sorted(times), "cores_delivered": cpu / sum(times),
                      "intersections": int(c[2 + phase - 1])}))


def reference_program_runs(runs: int, nphotons: int = 1_000_000):
    """runner.py's model on the CPU (runner.py:26-47): one PROCESS of the unmodified reference program
    (oracle/_ref/raytrace, all 13 sources incl. its own random_mod.f90 + main.f90, the serial build: under flang the
    OpenMP build is SLOWER — its random_number is one locked generator) per simulation of the lens experiment — wall seconds per simulation at `nphotons` photons per layer."""
    import shutil
    import subprocess
    import tempfile
    import numpy as np
    from opticalraytrace_amd.params import Settings, resource_dir
    from oracle.binding import REF_PROG
    if not os.path.exists(REF_PROG):
        return None
    tmp = tempfile.mkdtemp(prefix="ortref_sweep_")
    try:
        for d in ("bin", "res", "data"):
            os.makedirs(os.path.join(tmp, d))
        for f in os.listdir(resource_dir()):
            if f.endswith(".params"):
                shutil.copy(os.path.join(resource_dir(), f), os.path.join(tmp, "res", f))
        np.ones((512, 512)).tofile(os.path.join(tmp, "res", "ones.dat"))      # opened unconditionally (src/setupMod.f90:120-121)
        times = []
        for k in range(runs):
            f3, f2 = ("40.0", "45.0", "50.0")[k % 3], ("59.8", "49.8", "39.9")[k % 3]
            Settings(nphotons=nphotons, light_source="point", make_images=False, image_source="ones.dat", data_folder="images-lens",
                     bottle_file="clearBottle-large.params", L3_file=f"achromaticDoublet-f{f3}mm.params",
                     L2_file=f"planoConvex-f{f2}mm.params").write(os.path.join(tmp, "res", f"cfg{k}.params"))
            t0 = time.perf_counter()
            subprocess.run([REF_PROG, f"cfg{k}.params"], cwd=os.path.join(tmp, "bin"), capture_output=True, check=True, timeout=900)
            times.append(time.perf_counter() - t0)
        return {"runs": runs, "nphotons": nphotons, "seconds_per_simulation": sum(times) / len(times), "all_seconds": times,
                "simulations_per_s": len(times) / sum(times),
                "what": "oracle/_ref/raytrace: the unmodified reference program, one process per simulation (runner.py:26-47), "
                        "serial build (flang's OpenMP build is slower: one locked random_number generator, SURVEY §6)"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--program-runs", type=int, default=0, help="also time this many runs of the unmodified reference program")
    ap.add_argument("--rays", type=int, default=50_000_000)
    ap.add_argument("--phase", type=int, default=2)
    ap.add_argument("--kind", choices=["auto", "reference", "port"], default="auto")
    ap.add_argument("--repeats", type=int, default=3, help="timings per thread count; the median is reported")
    ap.add_argument("--worker", choices=["reference", "port"], default=None, help="internal: one thread count")
    args = ap.parse_args()
    if args.worker:
        return worker(args.worker, args.rays, args.phase, args.repeats)

    from oracle.binding import reference_available
    cpus = os.sched_getaffinity(0)
    n_threads, n_cores = len(cpus), physical_cores(cpus)
    kind = args.kind
    if kind == "auto":
        kind = "reference" if reference_available() else "port"
    cpu_model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # thread counts tried: one per physical core, and every hardware thread, each unbound and bound to
    # cores; the best is reported.  Plus one thread on a sample scaled down by the core count.
    quota = cpu_quota()
    tried = sorted({n_cores, n_threads} | ({max(1, int(round(quota)))} if quota and quota < n_cores else set()))
    one_rays = max(200_000, args.rays // max(n_cores, 1))
    sweep = {}
    kinds = [kind] if kind == "port" else ["reference", "port"]

    def entry(r, t, rays, bind):
        return {"threads": t, "bind": "close/cores" if bind else "none", "rays": rays, "seconds": r["seconds"],
                "cores_delivered": r["cores_delivered"],
                "all_seconds": r["all_seconds"], "value": r["intersections"] / r["seconds"], "rays_per_s": rays / r["seconds"]}

    for k in kinds:
        r = run_once(k, one_rays, args.phase, 1, 1)
        sweep[f"{k}@1"] = entry(r, 1, one_rays, False)
        for t in tried:
            r = run_once(k, args.rays, args.phase, t, args.repeats, False)
            sweep[f"{k}@{t}"] = entry(r, t, args.rays, False)
            isect = r["intersections"]
        # binding (close to cores) at the best unbound thread count only, one timing: under flang's OpenMP runtime inside
        # a CPU-quota cgroup it lands every thread on two cores (16 s instead of 2 s for the sample)
        tb = max(tried, key=lambda t: sweep[f"{k}@{t}"]["value"])
        try:
            r = run_once(k, args.rays, args.phase, tb, 1, True)
            sweep[f"{k}@{tb}b"] = entry(r, tb, args.rays, True)
        except RuntimeError:
            pass                                            # a runtime that cannot bind (cgroup cpuset): unbound only

    def best_of(k):
        keys = [x for x in sweep if x.startswith(k + "@") and x != f"{k}@1"]
        return max(keys, key=lambda x: sweep[x]["value"])

    best_key = best_of(kind)
    best = sweep[best_key]
    best_t = best["threads"]

    def efficiency(k, key):
        occupied = min(n_cores, sweep[key]["threads"], quota or n_cores)   # cores' worth of CPU time the run could use
        return sweep[key]["value"] / (occupied * sweep[f"{k}@1"]["value"])

    extra = {}
    if kind == "reference":                                 # also report the C restatement
        bp = best_of("port")
        extra = {"port_value": sweep[bp]["value"], "port_threads": sweep[bp]["threads"], "port_bind": sweep[bp]["bind"],
                 "port_one_thread_value": sweep["port@1"]["value"], "port_parallel_efficiency": efficiency("port", bp),
                 "port_sample": "oracle/libort_oracle.so (C restatement, gcc -O2 + OpenMP)"}
    what = ("oracle/_ref: the reference's own Fortran path sources compiled with flang -O2, OpenMP over rays"
            if kind == "reference" else "oracle/libort_oracle.so: C restatement, gcc -O2, OpenMP over rays")
    print(json.dumps({
        "value": best["value"], "unit": "intersections/s",
        "cores": min(n_cores, best_t),                      # physical cores the best run occupied
        "threads": best_t, "bind": best["bind"], "physical_cores_available": n_cores, "hardware_threads_available": n_threads,
        "cgroup_cpu_quota_cores": quota,                    # None: uncapped
        "cores_delivered": best["cores_delivered"],         # CPU seconds of all threads / wall second in the best run
        "kind": kind,
        "one_thread_value": sweep[f"{kind}@1"]["value"],
        # best / (cores' worth of CPU time available x the one-thread figure); available = min(physical cores occupied,
        # cgroup CPU quota): what the OpenMP loop of src/main.f90:83-89 with its
        # atomic image update (src/imageMod.f90:55) keeps of one core's rate at this thread count
        "parallel_efficiency": efficiency(kind, best_key),
        "sample": f"{args.rays} rays of phase {args.phase} (BASELINE configs[1] system, seed 123456789), "
                  f"{isect} intersections in {best['seconds']:.3f} s wall (median of {args.repeats}) on {best_t} threads "
                  f"(best of threads = {tried} x binding none / close-to-cores; one-thread point on {one_rays} rays); {what}",
        "rays_per_s": best["rays_per_s"], "seconds": best["seconds"], "cpu": cpu_model,
        "thread_sweep": sweep, **extra,
        **({"reference_program": reference_program_runs(args.program_runs)} if args.program_runs else {})}))


if __name__ == "__main__":
    main()
