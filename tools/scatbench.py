#!/usr/bin/env python3
"""Dev tool (GPU box): the scattering bottle — pipeline (scatter_front_kernel + continuation) against the
monolithic kernel (variant bit 4), per ray count.   usage: python tools/scatbench.py [--rays 1000000,10000000]   (ORT_HIP_LIB=build/ab/x.so selects another build)"""
import argparse
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", default="1000000,10000000")
    ap.add_argument("--bottle", default="scatterBottle-both.params")
    ap.add_argument("--variants", default="1,17")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch  # noqa: F401
    from opticalraytrace_amd import capi
    from opticalraytrace_amd.params import Settings, resource_dir
    from opticalraytrace_amd.system import OpticalSystem
    from conftest import res_dir_with_image
    s = Settings(nphotons=1000, make_images=True, bottle_file=args.bottle)
    osys = OpticalSystem.from_settings(s, res_dir_with_image(resource_dir()))
    print(f"# library build {capi.build_id()}  ({os.environ.get('ORT_HIP_LIB', 'in-tree')})", flush=True)
    with capi.Context(osys, device=0) as c:
        for n in [int(x) for x in args.rays.split(",")]:
            for v in [int(x) for x in args.variants.split(",")]:
                c.set_kernel_variant(v)
                ts = []
                for k in range(args.reps + 1):
                    c.reset()
                    c.synchronize()
                    t0 = time.perf_counter()
                    c.trace(2, 0, n, 123456789)
                    c.synchronize()
                    ts.append(time.perf_counter() - t0)
                _, cnt = c.read()
                culled, deferred = c.work_counters()
                t = min(ts[1:])
                print(f"n {n:>10} variant {v:>2}: {t * 1e3:8.3f} ms wall  = {t * 1e3 / n * 1e6:7.4f} ms per 1e6 rays  "
                      f"{int(cnt[3]) / n:5.2f} intersections/ray  {int(cnt[3]) / t / 1e9:6.1f} G intersections/s  deferred {deferred}", flush=True)


if __name__ == "__main__":
    main()
