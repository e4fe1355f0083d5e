#!/usr/bin/env python3
"""Dev tool (GPU box): A/B several builds of libort_hip.so in ONE process, interleaved (order rotated
per round, one discarded measurement in front of every group), at steady clocks.  Each measurement = mean kernel time of 64 back-to-back 1e7-ray launches (HIP events).
usage: python tools/abbench.py --libs build/a.so,build/b.so [--phases 2,1] [--precisions 0,2] [--rounds 5]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401,E402  (first: one HIP runtime per process)
from opticalraytrace_amd import capi  # noqa: E402
from opticalraytrace_amd.params import Settings  # noqa: E402
from opticalraytrace_amd.system import OpticalSystem  # noqa: E402


class Ctx(capi.Context):
    def __init__(self, lib, osys):
        self.lib = lib
        self.system = osys
        self._csys = capi.pack_system(osys)
        self._h = C.c_void_p()
        capi._check(lib, lib.ort_create(C.byref(self._csys), 0, None, C.byref(self._h)), "ort_create")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default=capi.library_path())
    ap.add_argument("--rays", type=int, default=10_000_000)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--phases", default="2,1")
    ap.add_argument("--precisions", default="0")
    ap.add_argument("--bottle", default="clearBottle-large.params")
    args = ap.parse_args()
    s = Settings(nphotons=args.rays, bottle_file=args.bottle, L2_file="planoConvex-f39.9mm.params",
                 L3_file="achromaticDoublet-f50.0mm.params")
    osys = OpticalSystem.from_settings(s)
    ctxs = []
    for p in args.libs.split(","):
        lib = capi.load_library(os.path.abspath(p), older_build=True)
        c = Ctx(lib, osys)
        c.set_timing(True)
        c.reserve(args.rays)
        lib.ort_build_id.restype = C.c_char_p
        print(f"# {os.path.basename(p)}: library build {lib.ort_build_id().decode()}")
        ctxs.append((os.path.basename(p), c))
    res = {}

    def measure(c, phase, prec):
        c.set_precision(prec)
        c.reset()
        for k in range(64):
            c.trace(phase, k * args.rays, args.rays, 123456789)
        ms = c.kernel_times(64)
        return sum(ms) / len(ms)

    # The first measurement behind a change of kernel (phase / precision) reads 2-4 % high whichever
    # library it is (identical copies of one library, first in the list: 0.3352 vs 0.3284 / 0.3280 ms):
    # every group starts with a discarded measurement, and the order rotates from round to round.
    for rnd in range(args.rounds + 1):
        order = ctxs[rnd % len(ctxs):] + ctxs[:rnd % len(ctxs)]
        for phase in [int(p) for p in args.phases.split(",")]:
            for prec in [int(p) for p in args.precisions.split(",")]:
                measure(order[-1][1], phase, prec)
                for name, c in order:
                    ms = measure(c, phase, prec)
                    if rnd:
                        res.setdefault((phase, prec, name), []).append(ms)
    for (phase, prec, name), v in sorted(res.items()):
        v = np.array(v)
        print(f"phase {phase} precision {prec} {name:28s}: mean {v.mean():.4f} ms  min {v.min():.4f}  max {v.max():.4f}")
    for name, c in ctxs:
        _, cnt = c.read()
        print(name, "counters", cnt)
        c.close()


if __name__ == "__main__":
    main()
