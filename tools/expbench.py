#!/usr/bin/env python3
"""Dev tool (GPU box): the experiments of csrc/ort_k_exp.hip on the fused point program against the production kernel, in ONE
process, interleaved, at steady clocks.  Each measurement = mean kernel time of 64 back-to-back launches (HIP events).
usage: python tools/expbench.py [--rays N] [--rounds R] [--configs "which:static_pct:min:max:wg_per_cu,..."]"""
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

NAMES = {0: "production (static, binned)", 1: "static, no atomic", 2: "pull V, binned", 3: "pull V, no atomic",
         4: "pull S, binned", 5: "pull S, no atomic"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=10_000_000)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--configs", default="0:0:1:1:1,1:0:1:1:1,2:60:2:32:6,3:60:2:32:6,4:60:2:32:6,5:60:2:32:6,3:0:1:4:6,5:0:1:4:6,"
                                         "3:80:2:32:6,5:80:2:32:6,3:40:2:16:6,5:40:2:16:6,3:60:2:32:5,5:60:2:32:5")
    args = ap.parse_args()
    cfgs = [tuple(int(x) for x in c.split(":")) for c in args.configs.split(",")]
    osys = OpticalSystem.from_settings(Settings(nphotons=args.rays, bottle_file="clearBottle-large.params",
                                                L2_file="planoConvex-f39.9mm.params", L3_file="achromaticDoublet-f50.0mm.params"))
    ctx = capi.Context(osys)
    lib = ctx.lib
    lib.ort_debug_set_exp.argtypes = [C.c_void_p] + [C.c_int] * 5
    lib.ort_debug_set_exp.restype = C.c_int
    ctx.set_timing(True)
    ctx.reserve(args.rays)

    def arm(cfg):
        rc = lib.ort_debug_set_exp(ctx._h, *cfg)
        assert rc == 0, lib.ort_last_error()

    def measure(cfg):
        arm(cfg)
        ctx.reset()
        for k in range(64):
            ctx.trace(2, k * args.rays, args.rays, 123456789)
        ms = ctx.kernel_times(64)
        img, cnt = ctx.read()
        return sum(ms) / len(ms), img, cnt

    res, ref = {}, None
    for rnd in range(args.rounds + 1):
        order = cfgs[rnd % len(cfgs):] + cfgs[:rnd % len(cfgs)]
        measure(order[-1])
        for cfg in order:
            ms, img, cnt = measure(cfg)
            if cfg[0] == 0 and ref is None:
                ref = (img, cnt)
            if rnd:
                res.setdefault(cfg, []).append(ms)
            if ref is not None:
                # the traced rays are the same whatever hands them out: counters identical; the image too where hits are binned
                assert np.array_equal(cnt, ref[1]), (cfg, cnt, ref[1])
                if cfg[0] in (0, 2, 4):
                    assert np.array_equal(img, ref[0]), cfg
    print(f"build {capi.build_id()}  rays per launch {args.rays}")
    for cfg in cfgs:
        v = np.array(res[cfg])
        print(f"{NAMES[cfg[0]]:30s} static {cfg[1]:3d} %  pull {cfg[2]}..{cfg[3]:2d} batches  {cfg[4]} WG/CU : mean {v.mean():.4f} ms  min {v.min():.4f}  max {v.max():.4f}")
    arm((0, 0, 1, 1, 1))
    ctx.close()


if __name__ == "__main__":
    main()
