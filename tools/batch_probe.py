#!/usr/bin/env python3
"""Dev tool (GPU box): the 75 simulations of the lens experiment one launch each (kernel time by HIP events, rays culled by
segment 0) against the multi-system launches of the same batch, and a batch of 75 copies of ONE system."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                                       # noqa: E402
import torch                                                             # noqa: E402
from opticalraytrace_amd.capi import Context, build_id, pack_systems     # noqa: E402
from opticalraytrace_amd.params import Settings                          # noqa: E402
from opticalraytrace_amd.sweeps import L2_FOCALS, L3_FOCALS, LENS_BOTTLES  # noqa: E402
from opticalraytrace_amd.system import OpticalSystem                     # noqa: E402

SEED = 123456789


def batch_ms(ctx, systems, n, phase, reps=5):
    counters = torch.zeros((len(systems), 8), dtype=torch.int64, device="cuda")
    packed = pack_systems(systems)
    ptrs = [counters[i].data_ptr() for i in range(len(systems))]
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        ctx.trace_batch(packed, phase, 0, n, SEED, [0] * len(systems), ptrs)
        ctx.synchronize()
        out.append(ctx.last_kernel_ms(1))
    return min(out), ctx.last_kernel_name()


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
    print(f"# library build {build_id()}  {n} rays per simulation and loop")
    systems = [OpticalSystem.from_settings(Settings(nphotons=n, light_source="point", make_images=False, bottle_file=b, use_bottle=u,
                                                    L3_file=f"achromaticDoublet-f{f3}mm.params", L2_file=f"planoConvex-f{f2}mm.params"))
               for f3 in L3_FOCALS for f2 in L2_FOCALS for b, u in LENS_BOTTLES]
    with Context(systems[0]) as ctx:
        ctx.set_timing(True)
        for phase in (1, 2):
            per, culled = [], []
            for s in systems:
                ctx.set_system(s)
                ctx.reset()
                ctx.trace(phase, 0, n, SEED)
                ctx.trace(phase, 0, n, SEED)
                ctx.synchronize()
                per.append(ctx.last_kernel_ms(0))
                culled.append(ctx.work_counters()[0] / 2 / n)
            per, culled = np.array(per), np.array(culled)
            print(f"phase {phase}: one launch per simulation: sum {per.sum():.3f} ms  min {per.min():.4f}  max {per.max():.4f}  "
                  f"culled share min {culled.min():.3f} max {culled.max():.3f}")
            ms, name = batch_ms(ctx, systems, n, phase)
            print(f"phase {phase}: multi-system launches of the 75: {ms:.3f} ms   ({name})")
            ms, name = batch_ms(ctx, [systems[0]] * 75, n, phase)
            print(f"phase {phase}: 75 copies of simulation 0 in one launch: {ms:.3f} ms   ({name})")
            ctx.set_system(systems[0])
            ctx.reset()
            ctx.trace(phase, 0, 75 * n, SEED); ctx.trace(phase, 0, 75 * n, SEED)
            ctx.synchronize()
            print(f"phase {phase}: simulation 0 with 75 x the rays, one launch: {ctx.last_kernel_ms(0):.3f} ms")


if __name__ == "__main__":
    main()
