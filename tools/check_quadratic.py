#!/usr/bin/env python3
"""CPU replay of the guards of solve_and_pick<filtered> (csrc/ort_device.h) against the literal
solveQuadratic + root choice (reference src/surfaces.f90:227-260, :75-86) with numpy doubles:
wherever the guards do not flag, hit/miss and the returned root must agree bit for bit.  Three
operand regimes: any bit pattern; moderate exponents; hb and c over the whole exponent range."""
import numpy as np

rng = np.random.default_rng(1)
n = 20_000_000


def rnd():
    return rng.integers(0, 2**64, n, dtype=np.uint64).view(np.float64)


def sgn():
    return rng.choice([-1.0, 1.0], n)


bad_total = 0
for rep in range(3):
    with np.errstate(all="ignore"):
        a, hb, c = np.abs(rnd()), rnd(), rnd()
        if rep == 1:
            a = np.ldexp(1 + rng.random(n), rng.integers(-120, 120, n))
            hb = np.ldexp(1 + rng.random(n), rng.integers(-340, 340, n)) * sgn()
            c = np.ldexp(1 + rng.random(n), rng.integers(-340, 340, n)) * sgn()
        if rep == 2:
            a = np.ldexp(1 + rng.random(n), rng.integers(-110, 110, n))
            hb = np.ldexp(1 + rng.random(n), rng.integers(-1074, 1023, n)) * sgn()
            c = np.ldexp(1 + rng.random(n), rng.integers(-1074, 1023, n)) * sgn()
        hh = hb * hb
        D = hh - a * c
        neg = D < 0
        sq = np.sqrt(D)
        bpos = hb > 0
        q = -(hb + np.where(bpos, sq, -sq))
        qpos, cneg = ~bpos, c < 0
        use = qpos & cneg
        num, den = np.where(use, q, c), np.where(use, a, q)
        common = (np.abs(D) > 1e-10 * hh) & (np.abs(D) < 2.0**900) & (a > 2.0**-100) & (a < 2.0**100) & (np.abs(c) > 2.0**-300)
        ok = common & (neg | ((np.abs(q) > 2.0**-300) & (np.abs(q) < 2.0**300)))
        tf, hf = num / den, (qpos | cneg) & ~neg
        b = 2 * hb
        disc = b * b - 4 * a * c
        negl = disc < 0
        sql = np.sqrt(disc)
        ql = np.where(b > 0, -0.5 * (b + sql), -0.5 * (b - sql))
        dz = disc == 0
        xd = -0.5 * b / a
        t0, t1 = np.where(dz, xd, ql / a), np.where(dz, xd, c / ql)
        sw = t0 > t1
        lo, hi = np.where(sw, t1, t0), np.where(sw, t0, t1)
        lneg = lo < 0
        tl, hl = np.where(lneg, hi, lo), ~(lneg & (hi < 0)) & ~negl
        bad = ok & ((hf != hl) | (hl & (tf.view(np.uint64) != tl.view(np.uint64))))
        print(f"regime {rep}: mismatches {bad.sum()} of {ok.sum()} unflagged ({(ok & hl).sum()} hits)")
        bad_total += int(bad.sum())
raise SystemExit(1 if bad_total else 0)
