#!/usr/bin/env python3
"""Dev tool (GPU box): wall time per 1e7-ray step of back-to-back ort_trace calls, with and without
the per-launch HIP events, to size the launch gap outside the kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from opticalraytrace_amd import capi  # noqa: E402
from opticalraytrace_amd.params import Settings  # noqa: E402
from opticalraytrace_amd.system import OpticalSystem  # noqa: E402

n, steps = 10_000_000, 100
ctx = capi.Context(OpticalSystem.from_settings(Settings(nphotons=n, bottle_file="clearBottle-large.params")))
for timing in (False, True, False, True):
    ctx.set_timing(timing)
    ctx.reset()
    for k in range(3):
        ctx.trace(2, k * n, n, 123456789)
    ctx.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        ctx.trace(2, (3 + k) * n, n, 123456789)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / steps
    extra = ""
    if timing:
        km = ctx.kernel_times(64)
        extra = f"  kernel bracket mean {sum(km) / len(km):.4f} ms"
    print(f"timing events {timing}: {dt * 1e3:.4f} ms per step{extra}")
