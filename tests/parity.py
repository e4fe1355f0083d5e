"""Shared comparison helpers for the parity tests (oracle / reference / HIP path)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = 123456789            # src/main.f90:79
REL_TOL = 1e-10             # north_star: results within 1e-10 relative fp64


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def merge_status(st):
    """Map the fine statuses onto what the Fortran driver can observe:
    0 binned, 1 reached the image plane but not binned, 3 bottle, 4 telescope (5 Help3 -> 4)."""
    st = np.asarray(st).copy()
    st[st == 2] = 1
    st[st == 5] = 4
    return st


def sparse_image(idx, cnt):
    img = np.zeros(2 * 401 * 401, np.int32)
    img[idx] = cnt
    return img.reshape(2, 401, 401)


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    scale = np.maximum(np.abs(b), 1e-300)
    return np.max(np.abs(a - b) / np.maximum(scale, np.max(np.abs(b)) * 1e-6 + 1e-300)) if a.size else 0.0


def assert_rays_equal(got, want, exact=True, what=""):
    """Every ray: same status, same bin, same draw count; state bit-exact or within REL_TOL."""
    sg, sw = merge_status(got["status"]), merge_status(want["status"])
    assert np.array_equal(sg, sw), f"{what}: status differs for rays {np.nonzero(sg != sw)[0][:10]}"
    binned = sw == 0
    assert np.array_equal(np.asarray(got["bin_xy"])[:, binned], np.asarray(want["bin_xy"])[:, binned]), \
        f"{what}: bins differ"
    if "n_draws" in got and "n_draws" in want:
        assert np.array_equal(got["n_draws"], want["n_draws"]), f"{what}: draw counts differ"
    reach = sw <= 1       # the Fortran driver leaves pos/dir mid-flight values too; compare all
    for key in ("pos_dir",):
        g, w = np.asarray(got[key]), np.asarray(want[key])
        if exact:
            assert np.array_equal(g, w), \
                f"{what}: {key} not bit-exact, max rel err {rel_err(g, w):.3e}"
        else:
            assert rel_err(g[:, reach], w[:, reach]) <= REL_TOL, \
                f"{what}: {key} rel err {rel_err(g[:, reach], w[:, reach]):.3e} > {REL_TOL}"


# draws the emitter of (light source, phase) consumes before the first surface
# (crs phase 1 draws a variable number: its explicit-input fixtures restart at draw 0)
EMIT_DRAWS = {("point", 1): 4, ("point", 2): 2, ("spot", 1): 4, ("spot", 2): 0,
              ("crs", 1): 0, ("crs", 2): 2, ("image", 1): 4, ("image", 2): 4,
              ("isors", 1): 0, ("isors", 2): 2}


def emit_draws(settings, phase):
    return EMIT_DRAWS[(settings.light_source, phase)]
