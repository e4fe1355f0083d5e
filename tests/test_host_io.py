"""Output side of the boundary: file name, raw image streams, stats row, sharding arithmetic."""
import os

import numpy as np
import pytest

from conftest import make_system
from opticalraytrace_amd import fstr
from opticalraytrace_amd.tracer import (RunResult, append_stats, output_basename, shard_range,
                                        write_images)


def test_str_rules():
    assert fstr.str_real(0.035, 7) == "0.03500"          # first 7 chars of f100.16
    assert fstr.str_real(-0.002, 7) == "-0.0020"
    assert fstr.str_real(0.0399, 6) == "0.0399"
    assert fstr.str_real(1.0) == "1.0000000000000000"
    assert fstr.str_int(7, 3) == "007" and fstr.str_int(12345, 3) == "123" and fstr.str_int(42) == "42"
    assert fstr.str_logical(True) == "T" and fstr.str_logical_array([False, True]) == "_F_T"


def test_output_basename_matches_reference_program():
    """Name written by the unmodified reference program for this set-up (src/main.f90:45-48)."""
    _, o = make_system("large")
    assert output_basename(o) == ("point_bottle_T_Ra_0.03500_Rb_0.03500_offset_-0.0020__F_F_1.00000"
                                  "_L2f_0.0399_L3f_0.0500_fo_0.00000_alp_5.00000_bwidth_0.00050_sep_0.00000")


def test_image_files_layout(tmp_path):
    img = np.zeros((2, 401, 401), np.int32)
    img[0, 200 + 3, 200 - 7] = 5          # layer ring, yp = 3, xp = -7
    img[1, 0, 400] = 2                    # layer point, yp = -200, xp = 200
    names = write_images(img, str(tmp_path / "x_image"))
    assert [os.path.basename(n) for n in names] == ["x_image-ring.dat", "x_image-point.dat",
                                                    "x_image-total.dat"]
    for n in names:
        assert os.path.getsize(n) == 401 * 401 * 8     # raw float64, no header (imageMod.f90:102-112)
    ring = np.fromfile(names[0], np.float64)
    assert ring[(-7 + 200) + 401 * (3 + 200)] == 5.0 and ring.sum() == 5.0   # xp fastest
    tot = np.fromfile(names[2], np.float64)
    assert tot.sum() == 7.0 and tot[400 + 401 * 0] == 2.0


def test_stats_row(tmp_path):
    _, o = make_system("small")
    cnt = np.zeros(8, np.uint64)
    cnt[0], cnt[1] = 975388, 399300                     # the losses of the reference program's own 1e6-ray run of this set-up
    res = RunResult(np.zeros((2, 401, 401), np.int32), cnt, 1000000)
    assert abs(res.ring_transmitted - 2.4612) < 1e-9 and abs(res.point_transmitted - 60.07) < 1e-9
    p = append_stats(str(tmp_path), o, res)
    append_stats(str(tmp_path), o, res)
    text = open(p).read()
    # character for character what the unmodified reference program (flang build) wrote for these very values
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refprog_small.npz"))
    want = str(g["stats"])
    header, row = want[:want.index("\n 2.46") + 1], want[want.index("\n 2.46") + 1:]
    assert text == header + row + row
    f = [x.strip() for x in row.replace("\n", "").split(",")]
    assert len(f) == 12 and f[4] == "T" and f[10] == "point" and float(f[5]) == 0.0175


def test_shard_ranges_tile_the_index_range():
    for n in (0, 1, 7, 10_000_000, 2 ** 31 - 1):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            for (lo, c), (lo2, _) in zip(parts, parts[1:]):
                assert lo + c == lo2
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_bench_refuses_more_ranks_than_gpus_before_touching_a_gpu():
    """`python bench.py --gpus N` as the driver issues it: on a box with fewer GPUs (here: none) a
    message and exit status 2 — no traceback, no JSON line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if os.path.exists("/dev/kfd"):
        pytest.skip("this box has GPUs: covered by tests/test_gpu_distributed.py")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "GPU(s)" in p.stderr and "Traceback" not in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_list_directed_reals_and_records():
    """fstr.list_directed_*: the layout of flang's `write(u,*)` (shortest round-trip digits; F form without a leading zero
    when the value rounded to one digit is in [0.1, 1e15), else d.dddE+-XX; records of 79 columns) — against 54 values a
    flang-built program printed here (tests/golden/flang_list_directed.json) and the reference program's own files."""
    import json
    R = fstr.list_directed_real
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flang_list_directed.json")))
    assert len(gold["records"]) >= 50
    for r in gold["records"]:
        x = float.fromhex(r["value"])
        assert R(x) == r["text"], (x, R(x), r["text"])
    for x, want in ((0.034800000000001496, "3.4800000000001496E-02"), (49.250400000000006, "49.250400000000006"),
                    (0.0399, "3.99E-02"), (0.05, "5.E-02"), (-2e-3, "-2.E-03"), (0.0, "0."), (60.07, "60.07"),
                    (100.0, "100."), (0.5, ".5"), (0.1, ".1"), (1e-10, "1.E-10"), (1234.5, "1234.5"), (-0.25, "-.25")):
        assert R(x) == want, (x, R(x), want)
        assert float(R(x).replace("E", "e")) == x
    rec = fstr.list_directed_record
    assert rec([1.5, ",", True, False, "ab", "cd", 2.0]) == " 1.5 , T F abcd 2.\n"
    long = "x" * 100
    out = rec([long])
    assert out == " " + "x" * 78 + "\n " + "x" * 22 + "\n" and all(len(ln) <= 79 for ln in out.splitlines())
    out = rec([1.0 / 3.0] * 6)                     # 6 x 18 characters (" .3333333333333333"): the fifth item starts a new record
    assert [len(ln) for ln in out.splitlines()] == [72, 36]
