#!/usr/bin/env python3
"""Dev tool (GPU box): kernel time per 1e7 rays of every path the library has — the fused surface
programs (the bench), the generic walk on the same systems, bundles resident in HBM (emit_kernel +
MODE_RESIDENT), the other light sources, a scattering bottle — so that the breadth of SURVEY §8 f is
measured, not only built.  HIP events of the library (last_kernel_ms: 0 fused, 1 resident, 2 emit).
usage: python tools/pathbench.py [--rays 10000000] [--reps 12]      (log kept as profiles/rNN/pathbench.log)"""
import argparse
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SEED = 123456789


def measure(name, settings_kw, n, reps, resident=False, precision=0, variant=1):
    import torch
    from opticalraytrace_amd import capi
    from opticalraytrace_amd.params import Settings, resource_dir
    from opticalraytrace_amd.system import OpticalSystem
    from conftest import needs_extended_res, res_dir_with_image
    s = Settings(**{**dict(nphotons=n, make_images=True), **settings_kw})
    res = res_dir_with_image(resource_dir()) if needs_extended_res(s) else None
    osys = OpticalSystem.from_settings(s, res)
    rows = []
    with capi.Context(osys, device=0) as c:
        c.set_timing(True)
        c.set_precision(precision)
        c.set_kernel_variant(variant)
        bundle = torch.empty((6, n), dtype=torch.float64, device="cuda:0") if resident else None
        for phase in (1, 2):
            ms, ems = [], []
            if not resident:
                # every row at steady clocks: the rows differ in how long their set-up idles the GPU, and the first ~30 ms of
                # work after an idle period run at lower clocks (tools/rampbench.py) — ~40 ms of untimed launches first
                for k in range(max(4, min(120, int(1.2e9 // max(n, 1))))):
                    c.trace(phase, 0, n, SEED)
                c.synchronize()
            for k in range(reps + 2):
                c.reset()
                if resident:
                    c.emit(phase, 0, n, SEED, bundle.data_ptr())
                    c.trace_resident(phase, 0, n, SEED, 4 if phase == 1 else 2, bundle.data_ptr())
                    c.synchronize()
                    if k >= 2:
                        ms.append(c.last_kernel_ms(1)); ems.append(c.last_kernel_ms(2))
                else:
                    c.trace(phase, 0, n, SEED)          # the same rays every time (the image source holds nphotons rays)
                    c.synchronize()
                    if k >= 2:
                        ms.append(c.last_kernel_ms(0))
            _, cnt = c.read()
            isect = int(cnt[2 + (phase - 1)])
            t = float(np.mean(ms))
            extra = f"  emit_kernel {np.mean(ems):.4f} ms" if resident else ""
            print(f"{name:44s} phase {phase}: {t:8.4f} ms per {n:.0e} rays  {isect / n:5.2f} intersections/ray  "
                  f"{isect / t / 1e6:8.1f} G intersections/s{extra}", flush=True)
            rows.append(t)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    if not args.only:
        # the generic walk needs the development knob in the environment before the library loads: child process
        from opticalraytrace_amd import capi
        print(f"# library build {capi.build_id()}; mean kernel time of {args.reps} launches (one at a time, a wait behind each) behind ~40 ms of untimed launches of the same kind")
        for tag, env in (("main", {}), ("generic", {"ORT_DEV_NO_PROGRAMS": "1"})):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--rays", str(args.rays), "--reps", str(args.reps),
                                "--only", tag], env={**os.environ, **env})
            if r.returncode:
                return r.returncode
        return 0
    n, reps = args.rays, args.reps
    large = dict(bottle_file="clearBottle-large.params")
    measure("(clock settle, discard)", large, n, 150)
    if args.only == "generic":
        measure("generic walk (no surface program), fp64", large, n, reps)
        measure("generic walk: crs, iris before the doublet", dict(bottle_file="clearBottle-large.params", light_source="crs", crs_spot_size=0.5e-3,
                                                                   iris="before", iris_size=0.5), n, reps)
        measure("generic walk: crs, fp32", dict(bottle_file="clearBottle-large.params", light_source="crs", crs_spot_size=0.5e-3), n, reps, precision=1)
        return 0
    measure("surface programs, fused, fp64 (the bench)", large, n, reps)
    measure("surface programs, fused, fp32", large, n, reps, precision=1)
    measure("surface programs, fused, fast fp64", large, n, reps, precision=2)
    measure("surface programs, resident bundle, fp64", large, n, reps, resident=True)
    measure("iris before the doublet", dict(bottle_file="clearBottle-large.params", iris="before", iris_size=0.5), n, reps)
    measure("elliptical bottle", dict(bottle_file="clearBottle-ellipse.params"), n, reps)
    # (the spot source is a deterministic fan of a few hundred distinct rays, src/sourceMod.f90:122-159: not a bulk workload)
    measure("light source crs", dict(bottle_file="clearBottle-large.params", light_source="crs", crs_spot_size=0.5e-3), n, reps)
    measure("light source image", dict(bottle_file="clearBottle-large.params", light_source="image",
                                       image_source="synthetic-source.dat"), n, reps)
    measure("light source isors", dict(bottle_file="clearBottle-small.params", light_source="isors", isors_offset=0.5e-3), n, reps)
    # round 4: every list x every source of its phase has a program, in fp64 and fp32
    iris = dict(iris="before", iris_size=0.5)
    measure("crs, iris before the doublet", dict(bottle_file="clearBottle-large.params", light_source="crs", crs_spot_size=0.5e-3, **iris), n, reps)
    measure("isors, iris before the doublet", dict(bottle_file="clearBottle-small.params", light_source="isors", isors_offset=0.5e-3, **iris), n, reps)
    measure("image, iris before the doublet", dict(bottle_file="clearBottle-large.params", light_source="image", image_source="synthetic-source.dat", **iris), n, reps)
    measure("crs, fp32", dict(bottle_file="clearBottle-large.params", light_source="crs", crs_spot_size=0.5e-3), n, reps, precision=1)
    measure("isors, fp32", dict(bottle_file="clearBottle-small.params", light_source="isors", isors_offset=0.5e-3), n, reps, precision=1)
    measure("image, fp32", dict(bottle_file="clearBottle-large.params", light_source="image", image_source="synthetic-source.dat"), n, reps, precision=1)
    measure("scattering bottle (contents + wall)", dict(bottle_file="scatterBottle-both.params"), n // 10, reps)
    # the reference's own bits in front of the surfaces: strict libm emitters and 53-bit draws — since round 5 template flags of
    # the surface programs in exact fp64 (fp32 on the 53-bit stream: the lockstep kernel)
    measure("strict libm emitters (variant 1|64)", large, n, reps, variant=1 | 64)
    measure("53-bit draws (variant 1|32)", large, n, reps, variant=1 | 32)
    measure("strict emitters on 53-bit draws (1|32|64)", large, n, reps, variant=1 | 32 | 64)
    measure("53-bit draws, fp32 (lockstep kernel)", large, n, reps, precision=1, variant=1 | 32)
    return 0


if __name__ == "__main__":
    sys.exit(main())
