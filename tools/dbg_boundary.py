import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import numpy as np
from conftest import make_system
from test_gpu_parity import _special_rays
from opticalraytrace_amd.capi import Context
from oracle.binding import Oracle
np.set_printoptions(precision=17, linewidth=200)
for name in ("small",):
    _, osys = make_system(name)
    ctx = Context(osys); orc = Oracle(osys)
    a, u = _special_rays(osys)
    n = a.shape[1]
    for phase in (2,):
        want = orc.trace_rays(phase, n, pos_dir_in=a, u=u, draw_base=0)
        for variant in (1, 3):
            ctx.set_kernel_variant(variant)
            got = ctx.trace_rays(phase, n, pos_dir_in=a, u=u, draw_base=0)
            bad = np.nonzero(got["status"] != want["status"])[0]
            print(name, phase, "variant", variant, "bad", bad)
            for i in bad[:6]:
                print(" ray", i, "in", a[:, i], "u", u[:, i])
                print("   got  st", got["status"][i], "nis", got["n_isect"][i], "nd", got["n_draws"][i], got["pos_dir"][:, i])
                print("   want st", want["status"][i], "nis", want["n_isect"][i], "nd", want["n_draws"][i], want["pos_dir"][:, i])
