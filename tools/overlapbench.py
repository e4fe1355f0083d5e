#!/usr/bin/env python3
"""Dev experiment (GPU box): do back-to-back launches gain from overlapping on two streams?
Wall time of 256 launches of 1e7 rays issued (a) on one context / one stream, (b) alternately on
two contexts with their own streams (their tails and ramps can overlap)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalraytrace_amd import capi
from opticalraytrace_amd.params import Settings
from opticalraytrace_amd.system import OpticalSystem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
s = Settings(nphotons=n, bottle_file="clearBottle-large.params", L2_file="planoConvex-f39.9mm.params",
             L3_file="achromaticDoublet-f50.0mm.params")
osys = OpticalSystem.from_settings(s)
st = [torch.cuda.Stream(), torch.cuda.Stream()]
cx = [capi.Context(osys, stream=x.cuda_stream) for x in st]
for c in cx:
    c.reserve(n)
for phase in (2, 1):
    for mode in ("one stream", "two streams"):
        for rep in range(3):
            for c in cx:
                c.reset()
            for k in range(64):
                cx[0].trace(phase, k * n, n, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(256):
                c = cx[k & 1] if mode == "two streams" else cx[0]
                c.trace(phase, k * n, n, 1)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 256 * 1e3
        print(f"phase {phase} {mode:12s}: {dt:.4f} ms per launch")
