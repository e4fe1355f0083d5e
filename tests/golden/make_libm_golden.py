#!/usr/bin/env python3
"""Writes tests/golden/libm_glibc235.npz: known answers of glibc 2.35's sin / cos / sincos / log / atan2 / acos
(x86-64, the variants selected on a CPU with FMA + AVX2) — arguments the tracer forms plus the boundaries of
every range the algorithms switch at, results as computed by THIS container's libm.so.6 through ctypes (one C call per
value: nothing a compiler could merge).  csrc/ort_libm.h is pinned to these numbers (tests/test_libm_exact.py), and
a host whose libm answers differently is reported as "not the pinned libm" instead of failing the parity suite."""
import ctypes as C
import math
import os
import numpy as np

libm = C.CDLL("libm.so.6")
for f in ("sin", "cos", "log", "acos"):
    getattr(libm, f).restype = C.c_double
    getattr(libm, f).argtypes = [C.c_double]
libm.atan2.restype = C.c_double
libm.atan2.argtypes = [C.c_double, C.c_double]
libm.sincos.restype = None
libm.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]


def around(e, k=12):
    out = [e]
    lo, hi = e, e
    for _ in range(k):
        lo = math.nextafter(lo, -math.inf); hi = math.nextafter(hi, math.inf)
        out += [lo, hi]
    return out


def main():
    rng = np.random.default_rng(20261005)
    n = 1500
    ang = list(2 * math.pi * rng.random(n)) + list(2 * math.pi * rng.integers(0, 2**32, n) * 2.0**-32) + list((rng.random(n) * 6 - 2) * math.pi)
    for e in (2.0**-27, 2.0**-26, 0.126, 0.855469, 2.426265, math.pi / 2, math.pi, 1.5 * math.pi, 2 * math.pi, 105414335.0, 0.0, 1e-300, 1e5, 3e7):
        ang += around(e) + [-x for x in around(e)]
    ang = np.array(ang)
    lg = list(rng.random(n)) + list(rng.integers(1, 2**32, n) * 2.0**-32) + list(1 + (rng.random(n) - 0.5) * 0.25) + [0.0]
    for e in (1.0, 1 - 2.0**-4, 1 + float.fromhex("0x1.09p-4"), 0.5, 2.0**-32, 2.0**-64, 2.0):
        lg += [x for x in around(e) if x >= 0]
    lg = np.array(lg)
    ac = list(rng.random(n) * 2 - 1) + list(np.cos(2 * math.pi * rng.random(n))) + list(np.copysign(1 - rng.random(n) * rng.random(n) * 0.04, rng.random(n) - 0.5))
    for e in (0.0, 2.0**-55, 0.125, 0.25, 0.5, 0.75, 0.921875, 0.953125, 0.96875, 1.0):
        ac += [x for x in around(e) + [-y for y in around(e)] if abs(x) <= 1]
    ac = np.array(ac)
    st, ph = np.sqrt(rng.random(n)), 2 * math.pi * rng.random(n)
    ay = list(st * np.sin(ph)) + list(rng.random(n) * 2 - 1)
    ax = list(st * np.cos(ph)) + list(rng.random(n) * 2 - 1)
    x0 = rng.random(n) * 2 - 1
    ax += list(x0); ay += list(x0 * (0.0625 + (rng.random(n) - 0.5) * 1e-3))
    k = rng.integers(16, 257, n) / 256.0
    ax += list(x0); ay += list(x0 * (k + (rng.random(n) - 0.5) * 2.0**-8))
    ay += list(x0); ax += list(x0 * (k + (rng.random(n) - 0.5) * 2.0**-8))
    ax += list(x0); ay += list(-x0)
    for y in (0.0, -0.0, 1.0, -1.0, 0.5, 2.0**-60, -2.0**-60, 1e-300, 3.0):
        for x in (0.0, -0.0, 1.0, -1.0, 0.5, 2.0**-60, -2.0**-60, 1e-300, -3.0):
            ay.append(y); ax.append(x)
    ay, ax = np.array(ay), np.array(ax)
    s, c = C.c_double(), C.c_double()
    scs, scc = np.empty_like(ang), np.empty_like(ang)
    for i, x in enumerate(ang):
        libm.sincos(float(x), C.byref(s), C.byref(c))
        scs[i], scc[i] = s.value, c.value
    out = dict(ang=ang, sin=np.array([libm.sin(float(x)) for x in ang]), cos=np.array([libm.cos(float(x)) for x in ang]), sincos_s=scs, sincos_c=scc,
               log_x=lg, log=np.array([libm.log(float(x)) for x in lg]),
               acos_x=ac, acos=np.array([libm.acos(float(x)) for x in ac]),
               atan2_y=ay, atan2_x=ax, atan2=np.array([libm.atan2(float(y), float(x)) for y, x in zip(ay, ax)]))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libm_glibc235.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
