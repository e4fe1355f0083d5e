#!/usr/bin/env python3
"""Dev tool (GPU box): where the time of the FIRST batched lens experiment of a fresh context goes (profiles/r05: 66 ms against
11 ms for every later one).  Times, with a stream wait behind each: context creation, a warm-up simulation, then the pieces of
two consecutive run_many-like batches (accumulators, pack, ort_trace_batch per loop, copy back)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                             # noqa: E402
from opticalraytrace_amd.capi import pack_systems, build_id              # noqa: E402
from opticalraytrace_amd.params import Settings                          # noqa: E402
from opticalraytrace_amd.sweeps import L2_FOCALS, L3_FOCALS, LENS_BOTTLES  # noqa: E402
from opticalraytrace_amd.system import OpticalSystem                     # noqa: E402
from opticalraytrace_amd.tracer import DEFAULT_SEED, RayTracer           # noqa: E402


class Clock:
    def __init__(self):
        self.t = time.perf_counter()

    def lap(self, what):
        torch.cuda.synchronize()
        now = time.perf_counter()
        print(f"  {what:58s} {1e3 * (now - self.t):8.2f} ms")
        self.t = now


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
    print(f"build {build_id()}  nphotons {n}")
    c = Clock()
    systems = [OpticalSystem.from_settings(Settings(nphotons=n, light_source="point", make_images=False, bottle_file=b, use_bottle=u,
                                                    L3_file=f"achromaticDoublet-f{f3}mm.params", L2_file=f"planoConvex-f{f2}mm.params"))
               for f3 in L3_FOCALS for f2 in L2_FOCALS for b, u in LENS_BOTTLES]
    c.lap("75 systems built (host)")
    t = RayTracer(systems[0])
    c.lap("RayTracer (context, accumulators)")
    t.run(n)
    c.lap("warm-up simulation (both loops, one system)")
    t.run(n)
    c.lap("the same again")
    for rep in range(3):
        print(f"batch {rep}")
        counters = torch.zeros((75, 8), dtype=torch.int64, device="cuda")
        c.lap("accumulators (torch.zeros)")
        packed = pack_systems(systems)
        c.lap("pack_systems")
        for phase in (1, 2):
            t.ctx.trace_batch(packed, phase, 0, n, DEFAULT_SEED, [0] * 75, [counters[i].data_ptr() for i in range(75)])
            host = time.perf_counter() - c.t
            c.lap(f"ort_trace_batch phase {phase} (host part {1e3 * host:.2f} ms) {t.ctx.last_kernel_name()[:40]}")
        pin = torch.empty((75, 8), dtype=torch.int64, pin_memory=True) if rep == 0 else pin
        c.lap("pinned staging buffer" if rep == 0 else "-")
        pin.copy_(counters, non_blocking=True)
        c.lap("copy back")
    t.close()


if __name__ == "__main__":
    main()
