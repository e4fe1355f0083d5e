#!/usr/bin/env python3
"""Dev tool (GPU box): how often each filtered predicate raises `rare` (needs a -DORT_DBG_RARE build:
ORT_HIP_LIB=build/libort_dbg.so python tools/rare_sites.py)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401,E402
from opticalraytrace_amd import capi  # noqa: E402
from opticalraytrace_amd.params import Settings  # noqa: E402
from opticalraytrace_amd.system import OpticalSystem  # noqa: E402

SITES = ["sqrt range", "div3 shared", "quadratic", "fresnel", "aperture", "NA", "bin"]
n = 10_000_000
ctx = capi.Context(OpticalSystem.from_settings(Settings(nphotons=n, bottle_file="clearBottle-large.params")))
lib = capi.load_library()
if len(sys.argv) > 1:
    ctx.set_precision(int(sys.argv[1]))
buf = (ctypes.c_ulonglong * 16)()
prev = [0] * 16
for phase in (2, 1):
    ctx.reset()
    ctx.trace(phase, 0, n, 123456789)
    ctx.synchronize()
    assert lib.ort_debug_rare(buf) == 0
    now = list(buf)
    print(f"phase {phase}: {n} rays:", {SITES[i]: now[i] - prev[i] for i in range(len(SITES))},
          "first raise at surface:", [now[8 + k] - prev[8 + k] for k in range(8)])
    prev = now
