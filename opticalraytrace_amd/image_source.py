"""The `image` light source on the host: reference `init_emit_image`
(src/sourceMod.f90:363-408) — read the 512 x 512 float64 source image (bpm.py output,
bpm.py:204-205), scale it to the number of rays with stochastic rounding — and the lookup
table `emit_image` (:303-323) implies.

`emit_image` walks the histogram in a fixed order (second index outer, first index inner),
emits one ray from the first cell whose count is not used up and decrements it.  In serial
order ray i therefore starts from the cell whose cumulative count first exceeds i: a CDF and
a binary search per ray, which is what the kernel does (the per-thread decrementing copy of
the OpenMP build is the same walk restarted per thread).  Rays beyond the histogram total
(its sum differs from nphotons by the rounding noise) have no cell: the reference re-uses a
stale ray there; here they are counted as lost.

The reference draws the 262 144 rounding uniforms from the still unseeded runtime generator;
here draw k of the (i, j) loop is ORT-RNG-v2(seed, phase 0, ray 0, k).
"""
from __future__ import annotations

import numpy as np

from .params import ParamsError
from .rng import uniforms

N = 512
PIXEL = 5000e-6 / 512.0            # dx, src/sourceMod.f90:337
HALF = 2500e-6


def load_image(path: str) -> np.ndarray:
    """File content in a C-order view v with v[i, j] = imgout(i, j) AFTER the reference's
    `imgout = transpose(imgout)` (:386): `read(u) imgout` fills the first index fastest."""
    img = np.fromfile(path, dtype=np.float64)
    if img.size != N * N:
        raise ParamsError(f"{path}: image source must hold 512 x 512 float64 values, found {img.size}")
    return img.reshape(N, N)


def histogram(img: np.ndarray, nphotons: int, seed: int) -> np.ndarray:
    """imgin of init_emit_image, returned in the ORDER emit_image SCANS it (flat, 262144)."""
    # sum(imgout) in array element order of the transposed array (first index fastest)
    tot = float(np.cumsum(img.T.reshape(-1))[-1])
    tmp = (float(nphotons) * img) / tot                       # tmp(i, j)
    base = np.trunc(tmp)
    diff = tmp - base
    u = uniforms(seed, 0, 0, np.arange(N * N)).reshape(N, N)  # draw k = i*512 + j (:396-397)
    counts = (base + ((u < diff) & (diff > 0))).astype(np.int64)   # imgin(i, j)
    # emit_image: do i2 (second index) ; do j2 (first index): img(j2, i2)  ->  s = i2*512 + j2
    return counts.T.reshape(-1).astype(np.int32)


def cdf(counts_scan: np.ndarray) -> np.ndarray:
    c = np.zeros(N * N + 1, dtype=np.int64)
    np.cumsum(np.maximum(counts_scan, 0), out=c[1:])
    return c
