#!/usr/bin/env python3
"""Dev tool (GPU box): parity soak over many random optical systems (tests/random_systems.py).
Per system: explicit-input rays HIP vs oracle bit for bit (status, bins, intersection and draw
counts, final state), the queued filtered kernel's image == the literal lockstep kernel's, and — round 3 — the
PRODUCTION kernel's image and counters == the oracle's for the same keyed rays (two rays of emission budget).
usage: python tools/parity_soak.py [--systems 300] [--rays 100000] [--first 100]   -> one line per
system + a summary (kept as profiles/rNN/parity_soak.log)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
from opticalraytrace_amd import capi  # noqa: E402
from opticalraytrace_amd.system import OpticalSystem  # noqa: E402
from oracle.binding import Oracle  # noqa: E402  (the checker; this is a test tool)
from parity import SEED, emit_draws  # noqa: E402
from random_systems import random_system  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--systems", type=int, default=300)
    ap.add_argument("--first", type=int, default=100)
    ap.add_argument("--rays", type=int, default=100_000)
    args = ap.parse_args()
    bad = 0
    rays = 0
    worst = 0
    t0 = time.time()
    print(f"# library build {capi.build_id()}, {args.systems} systems from seed {args.first}, {args.rays} rays per phase")
    for seed in range(args.first, args.first + args.systems):
        settings, res = random_system(seed)
        osys = OpticalSystem.from_settings(settings, res)
        orc = Oracle(osys)
        with capi.Context(osys, device=0) as ctx:
            notes = []
            n = args.rays + seed % 64
            for phase in (1, 2):
                u = np.random.default_rng(seed * 2 + phase).random((9, n))
                em = orc.trace_rays(phase, n, u=u)["emitted"]
                base = emit_draws(settings, phase)
                want = orc.trace_rays(phase, n, pos_dir_in=em, u=u, draw_base=base)
                got = ctx.trace_rays(phase, n, pos_dir_in=em, u=u, draw_base=base)
                for key in ("status", "bin_xy", "n_isect", "n_draws", "pos_dir"):
                    if not np.array_equal(got[key], want[key]):
                        notes.append(f"phase {phase} {key} differs")
                rays += n
            imgs = []
            for variant in (0, 1):
                ctx.set_kernel_variant(variant)
                ctx.reset()
                ctx.trace(1, 0, 2 * n, SEED)
                ctx.trace(2, 0, 2 * n, SEED)
                imgs.append(ctx.read())
            if not (np.array_equal(imgs[0][0], imgs[1][0]) and np.array_equal(imgs[0][1], imgs[1][1])):
                notes.append("queued filtered image != lockstep literal image")
            want_img = np.zeros((2, 401, 401), np.int32)
            want_cnt = np.zeros(8, np.uint64)
            orc.trace(1, 0, 2 * n, SEED, want_img, want_cnt)
            orc.trace(2, 0, 2 * n, SEED, want_img, want_cnt)
            l1 = int(np.abs(imgs[1][0].astype(np.int64) - want_img).sum())
            dc = int(np.abs(imgs[1][1].astype(np.int64) - want_cnt.astype(np.int64)).max())
            if l1 > 4 or dc > 2:
                notes.append(f"production image vs oracle: L1 {l1}, counter delta {dc}")
            worst = max(worst, l1)
            binned = int(imgs[1][1][4]) + int(imgs[1][1][5])
            # round 4: the fp32 queued kernels (segment 0 on the draw's word, the hit log) == the fp32 lockstep kernel;
            # strict libm emitters on the 53-bit stream == the oracle on it, without any budget
            ctx.set_precision(1)
            f32 = []
            for variant in (0, 1):
                ctx.set_kernel_variant(variant)
                ctx.reset()
                ctx.trace(1, 0, 2 * n, SEED)
                ctx.trace(2, 0, 2 * n, SEED)
                f32.append(ctx.read())
            if not (np.array_equal(f32[0][0], f32[1][0]) and np.array_equal(f32[0][1], f32[1][1])):
                notes.append("fp32 queued image != fp32 lockstep image")
            ctx.set_precision(0)
            ctx.set_kernel_variant(1 | 32 | 64)
            ctx.reset()
            ctx.trace(1, 0, n, SEED)
            ctx.trace(2, 0, n, SEED)
            wimg, wcnt = ctx.read()
            ctx.set_kernel_variant(1)
            orc.set_wide_draws(True)
            try:
                want_img[:] = 0; want_cnt[:] = 0
                orc.trace(1, 0, n, SEED, want_img, want_cnt)
                orc.trace(2, 0, n, SEED, want_img, want_cnt)
            finally:
                orc.set_wide_draws(False)
            if not (np.array_equal(wimg, want_img) and np.array_equal(wcnt, want_cnt)):
                notes.append("strict emitters on the 53-bit stream != oracle")
        bad += bool(notes)
        print(f"seed {seed:4d} iris {settings.iris:6s} bottle {int(settings.use_bottle)} binned {binned:7d} "
              f"{'OK' if not notes else 'MISMATCH: ' + '; '.join(notes)}", flush=True)
    print(f"# {args.systems} systems, {rays} explicit rays, {bad} systems with a mismatch, largest production-vs-oracle image L1 "
          f"distance {worst}, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
