#!/usr/bin/env python3
"""Dev tool (GPU box): A/B the kernel variants in one process, interleaved rounds.
usage: python tools/kbench.py [--rays N] [--rounds R] [--lib path ...]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401,E402  (first: one HIP runtime per process)
from opticalraytrace_amd import capi  # noqa: E402
from opticalraytrace_amd.params import Settings  # noqa: E402
from opticalraytrace_amd.system import OpticalSystem  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=10_000_000)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--bottle", default="clearBottle-large.params")
    ap.add_argument("--variants", default="0,1,2,3")
    ap.add_argument("--phases", default="2,1")
    ap.add_argument("--precision", type=int, default=0)
    args = ap.parse_args()
    global VARIANTS
    VARIANTS = [int(v) for v in args.variants.split(',')]
    s = Settings(nphotons=args.rays, bottle_file=args.bottle)
    osys = OpticalSystem.from_settings(s)
    print(f"# library build {capi.build_id()}")
    ctx = capi.Context(osys)
    ctx.set_timing(True)
    ctx.set_precision(args.precision)
    res = {}
    for rnd in range(args.rounds + 1):
        for phase in [int(p) for p in args.phases.split(',')]:
            for variant in VARIANTS:
                ctx.set_kernel_variant(variant)
                ctx.reset()
                ctx.trace(phase, 0, args.rays, 123456789)
                ctx.synchronize()
                ms = ctx.last_kernel_ms(0)
                if rnd:
                    res.setdefault((phase, variant), []).append(ms)
    _, cnt = ctx.read()
    for (phase, variant), v in sorted(res.items()):
        v = np.array(v)
        print(f"phase {phase} variant {variant}: median {np.median(v):.4f} ms  min {v.min():.4f}  "
              f"-> {args.rays / np.median(v) / 1e6:.1f} Grays/s")
    print("counters", cnt)


if __name__ == "__main__":
    main()
