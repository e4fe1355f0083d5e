"""Statistical comparison of two Monte-Carlo images of the same set-up that were produced
with DIFFERENT random streams (the unmodified reference program uses the Fortran
runtime's random_number; this build uses ORT-RNG-v2)."""
import numpy as np


def layer_moments(layer):
    """total, centroid (x, y) and rms radius in bins of one 401x401 layer (xp fastest)."""
    layer = np.asarray(layer, np.float64).reshape(401, 401)
    tot = layer.sum()
    yy, xx = np.mgrid[-200:201, -200:201]
    if tot == 0:
        return 0.0, 0.0, 0.0, 0.0
    cx, cy = (layer * xx).sum() / tot, (layer * yy).sum() / tot
    r2 = (layer * ((xx - cx) ** 2 + (yy - cy) ** 2)).sum() / tot
    return tot, cx, cy, np.sqrt(r2)


def radial_histogram(layer, nbins=20, rmax=200.0):
    layer = np.asarray(layer, np.float64).reshape(401, 401)
    yy, xx = np.mgrid[-200:201, -200:201]
    r = np.sqrt((xx + 0.5) ** 2 + (yy + 0.5) ** 2)
    h, _ = np.histogram(r, bins=nbins, range=(0, rmax), weights=layer)
    return h


def assert_same_distribution(a, b, n_a, n_b, what=""):
    """Two images from independent streams: totals within 5 sigma (binomial), centroid and rms
    radius within 5 standard errors, radial histogram chi-square per dof < 3."""
    ta, cxa, cya, ra = layer_moments(a)
    tb, cxb, cyb, rb = layer_moments(b)
    pa, pb = ta / n_a, tb / n_b
    sig = np.sqrt(pa * (1 - pa) / n_a + pb * (1 - pb) / n_b) + 1e-12
    assert abs(pa - pb) < 5 * sig, f"{what}: binned fraction {pa:.5f} vs {pb:.5f} (sigma {sig:.2e})"
    if min(ta, tb) < 200:
        return
    se = max(ra, rb) * np.sqrt(1 / ta + 1 / tb)
    assert abs(cxa - cxb) < 5 * se and abs(cya - cyb) < 5 * se, f"{what}: centroid"
    assert abs(ra - rb) < 5 * se, f"{what}: rms radius {ra:.3f} vs {rb:.3f} (se {se:.3f})"
    ha, hb = radial_histogram(a), radial_histogram(b)
    m = (ha + hb) > 20
    fa, fb = ha[m] / ta, hb[m] / tb
    var = ha[m] / ta ** 2 + hb[m] / tb ** 2
    chi2 = ((fa - fb) ** 2 / var).sum() / max(1, m.sum() - 1)
    assert chi2 < 3.0, f"{what}: radial chi2/dof {chi2:.2f}"
