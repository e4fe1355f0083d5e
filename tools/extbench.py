import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from opticalraytrace_amd import capi
from conftest import make_system
n = 10_000_000
for name in ("large", "large_crs", "small_scatter_c"):
    s, osys = make_system(name)
    ctx = capi.Context(osys)
    ctx.set_timing(True)
    for phase in (1, 2):
        ts = []
        for rep in range(5):
            ctx.reset(); ctx.trace(phase, 0, n, 123456789); ctx.synchronize()
            ts.append(ctx.kernel_times(1)[0])
        print(name, "phase", phase, "median ms", sorted(ts)[2])
    ctx.close()
