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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1 << 22)
    ap.add_argument("--bottle", default="scatterBottle-both.params")
    args = ap.parse_args()
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
        waves = min(int(os.environ.get("ORT_SCAT_WAVES", 3584)), (args.rays + 127) // 128)
        out = np.zeros((waves, 4), dtype=np.uint64)
        rc = lib.ort_debug_scat_times(out.ctypes.data_as(C.c_void_p), C.c_int(waves))
        assert rc == 0, rc
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", f"scat_times_{args.rays}_{waves}.npy"), out)
    t = out[:, :3].astype(np.int64)
    t0 = t[:, 0].min()
    start, emit, end = (t[:, 0] - t0) / 1e3, (t[:, 1] - t0) / 1e3, (t[:, 2] - t0) / 1e3
    q = lambda x: " ".join(f"{v:9.1f}" for v in np.percentile(x, [0, 10, 50, 90, 100]))
    print(f"rays {args.rays} waves {waves}   (kilo-ticks of s_memtime; percentiles 0 10 50 90 100)")
    print("start            ", q(start))
    print("last emission    ", q(emit))
    print("end              ", q(end))
    print("life (end-start) ", q(end - start))
    print("drain (end-emit) ", q(end - emit))
    print("passes           ", q(out[:, 3].astype(np.float64)))
    print(f"kernel span {end.max():.1f}   mean life {np.mean(end - start):.1f}   mean drain {np.mean(end - emit):.1f}")


if __name__ == "__main__":
    main()
