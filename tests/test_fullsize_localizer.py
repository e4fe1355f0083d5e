"""The full-size comparison of tests/fullsize.py itself (CPU, oracle on both sides): identical sides
report nothing; a side that loses ONE ray of 300 000 is caught and the ray is named."""
import numpy as np

from conftest import make_system
from fullsize import OracleSide, compare_full_size
from parity import SEED


class LosesOneRay(OracleSide):
    def __init__(self, orc, seed, victim):
        super().__init__(orc, seed)
        self.victim = victim

    def image(self, phase, lo, n):
        j = self.victim
        if not (lo <= j < lo + n):
            return super().image(phase, lo, n)
        img = np.zeros((401, 401), np.int32)
        cnt = np.zeros(8, np.uint64)
        for a, m in ((lo, j - lo), (j + 1, lo + n - j - 1)):
            if m:
                i, c = super().image(phase, a, m)
                img += i
                cnt += c
        return img, cnt

    def rays(self, phase, lo, n):
        r = super().rays(phase, lo, n)
        if lo <= self.victim < lo + n:
            r["status"][self.victim - lo] = 4
            r["n_isect"][self.victim - lo] = 0
        return r


def test_identical_sides_and_one_lost_ray():
    from oracle.binding import Oracle
    _, osys = make_system("large")
    orc = Oracle(osys)
    want = OracleSide(orc, SEED)
    lo, n = 1000, 300_000
    total = want.image(2, lo, n)
    rep = compare_full_size(want, want, 2, lo, n, total, chunk=1 << 16)
    assert rep.image_l1 == 0 and not any(rep.counter_delta) and not rep.divergences

    st = orc.trace_rays(2, 4096, seed=SEED, first_ray=lo + 200_000)["status"]
    victim = lo + 200_000 + int(np.nonzero(st == 0)[0][0])           # a ray that is binned
    bad = LosesOneRay(orc, SEED, victim)
    rep = compare_full_size(bad, want, 2, lo, n, bad.image(2, lo, n), chunk=1 << 16)
    assert rep.image_l1 == 1 and rep.chunks_differing == 1
    assert [(d.ray, d.kind) for d in rep.divergences] == [(victim, "defect")], rep.summary()
