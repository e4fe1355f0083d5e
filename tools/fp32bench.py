#!/usr/bin/env python3
"""Dev tool (GPU box): fp32 surface programs, one ray per lane (default) against two rays per lane (variant bit 5),
interleaved, 64 back-to-back 1e7-ray launches per measurement.   usage: python tools/fp32bench.py [--rounds 4]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401,E402
from opticalraytrace_amd import capi  # noqa: E402
from opticalraytrace_amd.params import Settings  # noqa: E402
from opticalraytrace_amd.system import OpticalSystem  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=10_000_000)
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    osys = OpticalSystem.from_settings(Settings(nphotons=args.rays, bottle_file="clearBottle-large.params"))
    with capi.Context(osys) as c:
        c.set_timing(True)
        c.reserve(args.rays)
        c.set_precision(1)
        res = {}
        for rnd in range(args.rounds + 1):
            for phase in (2, 1):
                for variant in ((1, 33) if rnd % 2 else (33, 1)):
                    c.set_kernel_variant(variant)
                    c.reset()
                    for k in range(64):
                        c.trace(phase, k * args.rays, args.rays, 123456789)
                    ms = c.kernel_times(64)
                    if rnd:
                        res.setdefault((phase, variant), []).append(sum(ms) / len(ms))
        for (phase, variant), v in sorted(res.items()):
            print(f"fp32 phase {phase} {'one ray per lane ' if variant == 1 else 'two rays per lane'}: mean {np.mean(v):.4f} ms  min {np.min(v):.4f}")


if __name__ == "__main__":
    main()
