#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — CPU baseline leg of bench.py (never the thing shipped).

Times the trace loop of BASELINE configs[1] (point source, clearBottle-large +
planoConvex-f39.9mm + achromaticDoublet-f50.0mm) on the host cores for a bounded
number of rays and prints one JSON object.

  kind "reference": oracle/_ref/libort_ref.so — the reference's own Fortran path
      sources (point, bottle%forward, telescope, makeImage) compiled with flang,
      OpenMP `parallel do` over rays as src/main.f90:83-89, fed ORT-RNG-v1 draws
      (the unmodified reference's random_number is one locked generator under
      flang and does not scale).
  kind "port": oracle/libort_oracle.so — the plain-C restatement, same loop.

Only the loop is timed (no parsing, no file output), as for the GPU.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=20_000_000)
    ap.add_argument("--phase", type=int, default=2)
    ap.add_argument("--kind", choices=["auto", "reference", "port"], default="auto")
    args = ap.parse_args()

    os.environ.setdefault("OMP_NUM_THREADS", str(len(os.sched_getaffinity(0))))
    cores = int(os.environ["OMP_NUM_THREADS"])      # = threads actually used
    os.environ.setdefault("OMP_PROC_BIND", "false")

    import numpy as np
    from opticalraytrace_amd.params import Settings, resource_dir
    from opticalraytrace_amd.system import OpticalSystem
    from oracle.binding import Oracle, Reference, reference_available

    s = Settings(nphotons=args.rays, make_images=True, bottle_file="clearBottle-large.params",
                 L2_file="planoConvex-f39.9mm.params", L3_file="achromaticDoublet-f50.0mm.params")
    osys = OpticalSystem.from_settings(s)
    orc = Oracle(osys)
    seed = 123456789
    # exact intersection count of the sample (the oracle counts them; the Fortran cannot)
    warm = 1_000_000                                        # first large call pays thread/arena start-up

    def time_port(n):
        orc.trace(args.phase, 0, warm, seed)
        t0 = time.perf_counter()
        _, c = orc.trace(args.phase, 0, n, seed)
        return time.perf_counter() - t0, int(c[2 + args.phase - 1])

    def time_reference(n):
        ref = Reference(s, resource_dir())
        ref.trace(args.phase, 0, warm, seed)
        t0 = time.perf_counter()
        ref.trace(args.phase, 0, n, seed)
        dt = time.perf_counter() - t0
        _, c = orc.trace(args.phase, 0, n, seed)            # untimed: the oracle counts the intersections
        return dt, int(c[2 + args.phase - 1])

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
    dt, isect = time_reference(args.rays) if kind == "reference" else time_port(args.rays)
    extra = {}
    if kind == "reference":                                 # also report the C restatement
        dtp, ip = time_port(args.rays)
        extra = {"port_value": ip / dtp,
                 "port_sample": f"oracle/libort_oracle.so (C restatement, gcc -O2 + OpenMP, {cores} threads): "
                                f"{args.rays} rays in {dtp:.3f} s wall"}
    what = ("oracle/_ref: the reference's own Fortran path sources compiled with flang -O2, OpenMP over rays"
            if kind == "reference" else "oracle/libort_oracle.so: C restatement, gcc -O2, OpenMP over rays")
    print(json.dumps({
        "value": isect / dt, "unit": "intersections/s", "cores": cores, "kind": kind,
        "sample": f"{args.rays} rays of phase {args.phase} (BASELINE configs[1] system, seed {seed}), "
                  f"{isect} intersections in {dt:.3f} s wall on {cores} threads; {what}",
        "rays_per_s": args.rays / dt, "seconds": dt, "cpu": cpu_model, **extra}))


if __name__ == "__main__":
    main()
