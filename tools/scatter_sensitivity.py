#!/usr/bin/env python3
"""Which libm function's last bit does the in-bottle scattering walk amplify?  (CPU only.)

The GPU's log / atan2 / acos / sin / cos differ from glibc's by an ulp for some arguments; the
walk (tauint + Henyey-Greenstein `stokes`, reference src/surfaces.f90:13-50, src/stokes.f90:7-166)
divides by sint * sinbt and takes acos of a value near +-1.  This builds the oracle with
-DORC_PERTURB, moves ONE function's result by one ulp in half of its calls, and reports how far
the final ray states move against the unperturbed oracle: the fraction of rays beyond 1e-10 /
1e-8 relative, and the fraction whose discrete outcome changes.
usage: python tools/scatter_sensitivity.py [n_rays]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_system  # noqa: E402
import oracle.binding as ob  # noqa: E402

so = os.path.join(ROOT, "build", "libort_oracle_perturb.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.run(["gcc", "-O2", "-std=gnu11", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-DORC_PERTURB",
                "-shared", "-o", so, os.path.join(ROOT, "oracle", "ort_oracle.c"), "-lm"], check=True)
ob.ORACLE_SO = so
ob.build_oracle = lambda force=False: so
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
for name in ("small_scatter_c", "small_scatter_bc"):
    _, osys = make_system(name)
    orc = ob.Oracle(osys)
    mask = C.c_int.in_dll(orc.lib, "orc_perturb_mask")
    mask.value = 0
    base = orc.trace_rays(2, n, seed=123456789)
    reach = base["status"] <= 2
    print(f"{name}: {n} rays, {reach.sum()} reach the image plane, mean draws {base['n_draws'].mean():.1f}")
    for bit, fn in ((1, "log (tauint)"), (2, "atan2 (phip)"), (4, "acos (cosdph)"), (8, "sin/cos(ri1)"), (16, "sin/cos(phi)"), (31, "all")):
        mask.value = bit
        p = orc.trace_rays(2, n, seed=123456789)
        flips = (p["status"] != base["status"]) | (p["n_draws"] != base["n_draws"])
        both = reach & ~flips
        a, b = p["pos_dir"][:, both], base["pos_dir"][:, both]
        scale = np.maximum(np.abs(b), np.abs(b).max(axis=1, keepdims=True) * 1e-6)
        err = (np.abs(a - b) / scale).max(axis=0)
        print(f"  {fn:14s}: outcome flips {flips.mean():.2e}   state > 1e-10: {np.mean(err > 1e-10):.2e}   "
              f"> 1e-8: {np.mean(err > 1e-8):.2e}   median {np.median(err):.1e}   max {err.max():.1e}")
    mask.value = 0
