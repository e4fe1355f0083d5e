#!/usr/bin/env python3
"""Dev tool (GPU box): per-wave timeline of scatter_front_kernel from a build with -DORT_SCAT_TIMING
(ORT_HIP_LIB=build/ab/timing.so): start, last emission, end of every wave of ONE launch.
usage: ORT_HIP_LIB=build/ab/timing.so python tools/scat_timing.py [--rays N]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def host_waves(n):
    """scatter_waves() of ort_hip.hip"""
    forced = int(os.environ.get("ORT_SCAT_WAVES", 0))
    waves = forced if forced > 0 else min(max(n // 2730, 3072), 6144)
    return min(waves, (n + 127) // 128)


def summarise(out, rays):
    """out[wave] = (start, last emission, end, passes) in s_memtime ticks.  The counter is not common to the chip: waves
    are grouped by CU through their start times (the waves a CU holds at launch start within ~2e3 ticks of each other)."""
    t = out[:, :3].astype(np.int64)
    live = out[:, 3] > 1
    life, drain = (t[:, 2] - t[:, 0])[live] / 1e3, (t[:, 2] - t[:, 1])[live] / 1e3
    q = lambda x: " ".join(f"{v:9.1f}" for v in np.percentile(x, [0, 10, 50, 90, 100]))
    print(f"rays {rays} waves {len(out)}   (kilo-ticks of s_memtime; percentiles 0 10 50 90 100)")
    print("life (end-start) ", q(life))
    print("drain (end-emit) ", q(drain))
    print("passes           ", q(out[live, 3].astype(np.float64)))
    print(f"longest life {life.max():.1f}   mean life {life.mean():.1f}   mean drain {drain.mean():.1f}")
    order = np.argsort(t[:, 0])
    s = t[order, 0]
    cuts = [0] + list(np.nonzero(np.diff(s) > 5_000_000)[0] + 1) + [len(s)]
    shown = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        ii = order[a:b]
        ii = ii[out[ii, 3] > 1]
        if 8 <= len(ii) <= 13 and (t[ii, 0].max() - t[ii, 0].min()) < 3000:      # one CU, all of its waves resident from the start
            ends = np.sort((t[ii, 2] - t[ii, 0].min()) / 1e3)
            print(f"one CU, {len(ii)} waves, end (kilo-ticks after the first start): " + " ".join(f"{e:.0f}" for e in ends))
            shown += 1
            if shown == 3:
                break


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1 << 22)
    ap.add_argument("--bottle", default="scatterBottle-both.params")
    ap.add_argument("--from-file", default=None, help="summarise a saved gpurun_out/scat_times_<rays>_<waves>.npy")
    args = ap.parse_args()
    if args.from_file:
        summarise(np.load(args.from_file), args.rays)
        return
    import torch  # noqa: F401
    from opticalraytrace_amd import capi
    from opticalraytrace_amd.params import Settings, resource_dir
    from opticalraytrace_amd.system import OpticalSystem
    from conftest import res_dir_with_image
    s = Settings(nphotons=1000, make_images=True, bottle_file=args.bottle)
    osys = OpticalSystem.from_settings(s, res_dir_with_image(resource_dir()))
    lib = capi.load_library()
    with capi.Context(osys, device=0) as c:
        c.set_kernel_variant(1)
        for _ in range(3):
            c.reset(); c.trace(2, 0, args.rays, 123456789); c.synchronize()
        waves = host_waves(args.rays)
        out = np.zeros((waves, 4), dtype=np.uint64)
        rc = lib.ort_debug_scat_times(out.ctypes.data_as(C.c_void_p), C.c_int(waves))
        assert rc == 0, rc
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", f"scat_times_{args.rays}_{waves}.npy"), out)
    summarise(out, args.rays)


if __name__ == "__main__":
    main()
