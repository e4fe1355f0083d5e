#!/usr/bin/env python3
"""Dev tool (GPU box): the scattering walk ray by ray against the CPU checker, bit for bit.
Emission taken from the checker (explicit input rays, keyed draws from draw 2 on) and, second, emitted on the GPU:
how many rays differ at all, and by how much.   usage: python tools/scatter_exact.py [n_rays]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    import torch  # noqa: F401
    from opticalraytrace_amd import capi
    from opticalraytrace_amd.params import Settings, resource_dir
    from opticalraytrace_amd.system import OpticalSystem
    from oracle.binding import Oracle
    from conftest import res_dir_with_image
    for bottle in ("scatterBottle-contents.params", "scatterBottle-both.params"):
        osys = OpticalSystem.from_settings(Settings(nphotons=1000, make_images=True, bottle_file=bottle), res_dir_with_image(resource_dir()))
        orc = Oracle(osys)
        want = orc.trace_rays(2, n, seed=123456789, first_ray=0)
        with capi.Context(osys, device=0) as c:
            for label, kw in (("explicit emission", dict(pos_dir_in=want["emitted"], draw_base=2)), ("emitted on the GPU", {})):
                got = c.trace_rays(2, n, seed=123456789, first_ray=0, **kw)
                same = (got["status"] == want["status"]) & (got["n_draws"] == want["n_draws"]) & (got["n_isect"] == want["n_isect"])
                reach = same & (want["status"] <= 2)
                a, b = got["pos_dir"][:, reach], want["pos_dir"][:, reach]
                ne = (a != b).any(axis=0)
                scale = np.maximum(np.abs(b), np.abs(b).max(axis=1, keepdims=True) * 1e-6)
                err = (np.abs(a - b) / scale).max(axis=0)
                print(f"{bottle} [{label}]: rays {n}, outcome differs {int((~same).sum())}, reach {int(reach.sum())}, "
                      f"state not bit-identical {int(ne.sum())}, beyond 1e-10: {int((err > 1e-10).sum())}, max rel err {err.max():.3g}, "
                      f"max draws {int(want['n_draws'].max())}", flush=True)


if __name__ == "__main__":
    main()
