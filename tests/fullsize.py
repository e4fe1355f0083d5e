"""Full-size image parity: a production trace of N keyed rays against the oracle's image of the same
rays, with every difference traced back to the ray that caused it.

`compare_full_size` is independent of who traces: it takes two objects with

    image(phase, lo, n) -> (int32 [401][401] layer image, uint64 [8] counters)   rays [lo, lo + n)
    rays(phase, lo, n)  -> dict(status, bin_xy, n_isect, emitted)                 per-ray outcomes

`got` is the side under test (the HIP library through the C ABI: the production kernel for
`image`, the parity entry ort_trace_rays for `rays`), `want` the checker (oracle/).  The caller has
already produced `got_total`, the image + counters of the whole range traced the way production
traces it (one ort_trace call per BASELINE step, whatever launches that is cut into).

Method.  The oracle walks the range in chunks and keeps each chunk's layer image (643 KB per
chunk).  If the summed oracle image and counters equal `got_total`, that is the result: identical.
Otherwise the side under test re-traces chunk by chunk (cheap on the GPU), every chunk whose image
or counters differ is bisected down to LEAF rays, and the leaf is compared ray by ray.  A differing
ray is classified:

  emission   the two sides emitted the ray differently (sin / cos of the device library vs glibc,
             <= 2 ulp): a discrete outcome may legitimately flip — budgeted, reported
  defect     identical emitted ray, different outcome — or the per-ray entry agrees with the oracle
             while the production image of the leaf does not: never tolerated
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np

LEAF = 1 << 14


@dataclass
class Divergence:
    phase: int
    ray: int
    kind: str                 # "emission" | "defect"
    detail: str


@dataclass
class FullSizeReport:
    rays: int = 0
    image_l1: int = 0
    counter_delta: List[int] = field(default_factory=lambda: [0] * 8)
    divergences: List[Divergence] = field(default_factory=list)
    chunks_differing: int = 0
    want_image: object = None          # the checker's layer image (int64 [401][401]) and counters (int64 [8]) of the range
    want_counters: object = None

    @property
    def defects(self):
        return [d for d in self.divergences if d.kind == "defect"]

    def summary(self) -> str:
        lines = [f"{self.rays} rays: image L1 {self.image_l1}, counter deltas {self.counter_delta}, "
                 f"{len(self.divergences)} diverging rays ({len(self.defects)} defects) in "
                 f"{self.chunks_differing} chunks"]
        lines += [f"  phase {d.phase} ray {d.ray}: {d.kind}: {d.detail}" for d in self.divergences[:20]]
        return "\n".join(lines)


def _layer_from_rays(r) -> np.ndarray:
    img = np.zeros((401, 401), np.int64)
    b = r["status"] == 0
    np.add.at(img, (r["bin_xy"][1][b] + 200, r["bin_xy"][0][b] + 200), 1)
    return img


def _leaf(got, want, phase, lo, n, rep: FullSizeReport) -> None:
    g, w = got.rays(phase, lo, n), want.rays(phase, lo, n)
    differ = (g["status"] != w["status"]) | (g["n_isect"] != w["n_isect"])
    b = w["status"] == 0
    differ |= b & ((g["bin_xy"][0] != w["bin_xy"][0]) | (g["bin_xy"][1] != w["bin_xy"][1]))
    for i in np.nonzero(differ)[0]:
        same_emission = np.array_equal(g["emitted"][:, i], w["emitted"][:, i])
        what = (f"status {int(g['status'][i])} vs {int(w['status'][i])}, bin {g['bin_xy'][:, i].tolist()} vs "
                f"{w['bin_xy'][:, i].tolist()}, intersections {int(g['n_isect'][i])} vs {int(w['n_isect'][i])}, "
                f"emitted delta {np.abs(g['emitted'][:, i] - w['emitted'][:, i]).max():.3e}")
        rep.divergences.append(Divergence(phase, lo + int(i), "defect" if same_emission else "emission", what))
    # the production kernel must equal the per-ray entry of its own library on this leaf
    img, _ = got.image(phase, lo, n)
    if not np.array_equal(img.astype(np.int64), _layer_from_rays(g)):
        rep.divergences.append(Divergence(phase, lo, "defect",
                                          f"production image of rays [{lo}, {lo + n}) differs from the image of its own per-ray outcomes"))
    elif not differ.any():
        rep.divergences.append(Divergence(phase, lo, "defect",
                                          f"leaf [{lo}, {lo + n}) differs between the sides but no ray does"))


def _bisect(got, want, phase, lo, n, rep: FullSizeReport) -> None:
    if n <= LEAF:
        _leaf(got, want, phase, lo, n, rep)
        return
    h = n // 2
    for a, m in ((lo, h), (lo + h, n - h)):
        gi, gc = got.image(phase, a, m)
        wi, wc = want.image(phase, a, m)
        if not (np.array_equal(gi, wi) and np.array_equal(gc, wc)):
            _bisect(got, want, phase, a, m, rep)


def compare_full_size(got, want, phase: int, lo: int, n: int, got_total, chunk: int = 1 << 23) -> FullSizeReport:
    """See the module text.  got_total = (layer image int32 [401][401], counters uint64 [8]) of rays
    [lo, lo + n) of `phase` as production traced them."""
    rep = FullSizeReport(rays=n)
    kept = []
    total = np.zeros((401, 401), np.int64)
    ctot = np.zeros(8, np.int64)
    for a in range(lo, lo + n, chunk):
        m = min(chunk, lo + n - a)
        wi, wc = want.image(phase, a, m)
        kept.append((a, m, wi.copy(), wc.copy()))
        total += wi
        ctot += wc.astype(np.int64)
    rep.want_image, rep.want_counters = total, ctot
    gi, gc = got_total
    rep.image_l1 = int(np.abs(gi.astype(np.int64) - total).sum())
    rep.counter_delta = (gc.astype(np.int64) - ctot).tolist()
    if rep.image_l1 == 0 and not any(rep.counter_delta):
        return rep
    for a, m, wi, wc in kept:
        ci, cc = got.image(phase, a, m)
        if np.array_equal(ci, wi) and np.array_equal(cc, wc):
            continue
        rep.chunks_differing += 1
        _bisect(got, want, phase, a, m, rep)
    if not rep.divergences:
        rep.divergences.append(Divergence(phase, lo, "defect",
                                          "the whole-range trace differs from the oracle but no chunk traced alone does "
                                          "(launch cuts / deferral groups / ray keys across launches)"))
    return rep


class OracleSide:
    """The checker as a `compare_full_size` side (oracle/binding.py: Oracle)."""

    def __init__(self, orc, seed):
        self.orc, self.seed = orc, seed

    def image(self, phase, lo, n):
        img = np.zeros((2, 401, 401), np.int32)
        cnt = np.zeros(8, np.uint64)
        self.orc.trace(phase, lo, n, self.seed, img, cnt)
        return img[phase - 1], cnt

    def rays(self, phase, lo, n):
        return self.orc.trace_rays(phase, n, seed=self.seed, first_ray=lo)


class HipSide:
    """The library under test: production kernel for images, ort_trace_rays per ray."""

    def __init__(self, ctx, seed):
        self.ctx, self.seed = ctx, seed

    def image(self, phase, lo, n):
        self.ctx.reset()
        self.ctx.trace(phase, lo, n, self.seed)
        img, cnt = self.ctx.read()
        return img[phase - 1], cnt

    def rays(self, phase, lo, n):
        return self.ctx.trace_rays(phase, n, seed=self.seed, first_ray=lo)
