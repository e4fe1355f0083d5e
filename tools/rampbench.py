#!/usr/bin/env python3
"""Kernel time per 1e7-ray step as a function of time since the GPU left idle (clock ramp):
prints the mean of every group of 64 back-to-back launches.  usage: rampbench.py [groups] [idle_s]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalraytrace_amd.params import Settings
from opticalraytrace_amd.system import OpticalSystem
from opticalraytrace_amd.tracer import DEFAULT_SEED, RayTracer

groups = int(sys.argv[1]) if len(sys.argv) > 1 else 12
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
s = Settings(nphotons=10_000_000, make_images=True, bottle_file="clearBottle-large.params",
             L2_file="planoConvex-f39.9mm.params", L3_file="achromaticDoublet-f50.0mm.params")
t = RayTracer(OpticalSystem.from_settings(s), device=0)
t.ctx.set_timing(True)
t.ctx.reserve(10_000_000)
t.ctx.trace(2, 0, 64, DEFAULT_SEED)
t.ctx.synchronize()
time.sleep(idle)
t0 = time.perf_counter()
for g in range(groups):
    for k in range(64):
        t.ctx.trace(2, (g * 64 + k) * 10_000_000, 10_000_000, DEFAULT_SEED)
    ms = t.ctx.kernel_times(64)
    print(f"group {g:2d}  t={time.perf_counter() - t0:6.3f}s  mean {sum(ms) / len(ms):.4f} ms  min {min(ms):.4f}  max {max(ms):.4f}", flush=True)
t.close()
