"""`python -m opticalraytrace_amd <settings>` = `bin/raytrace <settings>` at the process boundary
(install.sh:70-72, src/setupMod.f90:51-56): same inputs, same output files, a non-zero exit where the
reference would `error stop` (what runner.py:47 `check=True` relies on)."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from opticalraytrace_amd.params import Settings, resource_dir

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tree(tmp_path, **over):
    """The reference's run tree: <tmp>/{bin,res,data}; res holds the .params files and a settings
    file in runner.py's format (runner.py:106-108)."""
    for d in ("bin", "res", "data"):
        os.makedirs(tmp_path / d, exist_ok=True)
    for f in os.listdir(resource_dir()):
        if f.endswith(".params"):
            shutil.copy(os.path.join(resource_dir(), f), tmp_path / "res" / f)
    s = Settings(**{**dict(nphotons=200_000, make_images=True, data_folder="images"), **over})
    s.write(str(tmp_path / "res" / "test_0.params"))
    return s


def _run(cwd, *args):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return subprocess.run([sys.executable, "-m", "opticalraytrace_amd", *args], cwd=cwd, env=env,
                          capture_output=True, text=True, timeout=600)


def test_bad_inputs_exit_non_zero_with_a_message(tmp_path):
    _tree(tmp_path)
    bin_dir = str(tmp_path / "bin")
    p = _run(bin_dir, "nothing.params")                               # open(status="old") fails in the reference
    assert p.returncode == 1 and "nothing.params" in p.stderr and "Traceback" not in p.stderr
    # a 14-line bottle file: the reference dies with "End of file" (src/lens.f90:205)
    base = open(tmp_path / "res" / "clearBottle-small.params").read().splitlines()[:12]
    with open(tmp_path / "res" / "clearBottle-14.params", "w") as f:
        f.write("\n".join(base + ["0.", "0."]) + "\n")
    Settings(nphotons=1000, bottle_file="clearBottle-14.params").write(str(tmp_path / "res" / "bad.params"))
    p = _run(bin_dir, "bad.params")
    assert p.returncode == 1 and "exactly 12" in p.stderr
    Settings(nphotons=1000, light_source="laser").write(str(tmp_path / "res" / "src.params"))
    p = _run(bin_dir, "src.params")
    assert p.returncode == 1 and "source type" in p.stderr            # setupMod.f90:98
    assert not os.listdir(tmp_path / "data")                          # nothing was written


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="this box has a GPU")
def test_without_a_gpu_the_run_fails_loudly(tmp_path):
    _tree(tmp_path, nphotons=1000)
    p = _run(str(tmp_path / "bin"), "test_0.params")
    assert p.returncode == 3 and "HIP" in p.stderr and "fallback" in p.stderr


@pytest.mark.gpu
def test_entry_point_writes_what_run_settings_writes(tmp_path, hip_library):
    from opticalraytrace_amd.tracer import run_settings
    s = _tree(tmp_path, bottle_file="clearBottle-large.params", iris="before", iris_size=0.8)
    p = _run(str(tmp_path / "bin"), "test_0.params")                  # as install.sh does: from bin/, bare name
    assert p.returncode == 0, p.stderr[-2000:]
    assert "Ring  transmitted:" in p.stdout and "Point transmitted:" in p.stdout
    got_dir = tmp_path / "data" / "images"
    names = sorted(f for f in os.listdir(got_dir) if f.endswith(".dat") and "image" in f)
    assert len(names) == 3 and names[0].startswith("point_bottle_T_Ra_")
    want_root = tmp_path / "want"
    res = run_settings(s, str(tmp_path / "res"), str(want_root), verbose=False)
    for n in names:
        a = np.fromfile(got_dir / n)
        b = np.fromfile(want_root / "images" / n)
        assert a.size == 401 * 401 and np.array_equal(a, b), n
    assert np.fromfile(got_dir / names[2]).sum() == res.image.sum()   # -total = ring + point
    rows = open(got_dir / "trans-stats.dat").read().splitlines()
    assert len(rows) == 4 and rows == open(want_root / "images" / "trans-stats.dat").read().splitlines()   # header + record, two lines each
    # explicit paths instead of the bin/ layout
    p = _run(str(tmp_path), "res/test_0.params", "--data", "data2", "--quiet")
    assert p.returncode == 0 and p.stdout == ""
    assert sorted(f for f in os.listdir(tmp_path / "data2" / "images") if "image" in f) == names
    # a process on its own never loads torch (tracer.LocalTracer: runner.py starts one process per simulation, and the
    # import would be most of its life) ...
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    code = ("import sys; from opticalraytrace_amd.__main__ import main; rc = main(['res/test_0.params', '--data', 'data3', '--quiet']); "
            "print('torch' in sys.modules); sys.exit(rc)")
    p = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip() == "False", (p.stdout, p.stderr[-1000:])
    # ... and with torch (ORT_NO_TORCH=0: RayTracer, the tracer of the multi-rank run) writes the same bytes
    p = subprocess.run([sys.executable, "-c", code.replace("data3", "data4")], cwd=str(tmp_path), env=dict(env, ORT_NO_TORCH="0"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip() == "True", (p.stdout, p.stderr[-1000:])
    for n in names:
        assert np.array_equal(np.fromfile(tmp_path / "data3" / "images" / n), np.fromfile(tmp_path / "data4" / "images" / n)), n


@pytest.mark.gpu
def test_entry_point_options_beyond_the_reference(tmp_path, hip_library):
    """--fp32, --strict-libm, --wide-draws: the arithmetic / libm / draw-stream choices of the library at the process
    boundary.  Same files, transmissions within a tenth of a percent of the default run's (other rays, same statistics)."""
    _tree(tmp_path, bottle_file="clearBottle-small.params")
    base = None
    for k, flags in enumerate(([], ["--fp32"], ["--strict-libm"], ["--wide-draws"], ["--strict-libm", "--wide-draws"])):
        p = _run(str(tmp_path), "res/test_0.params", "--data", f"d{k}", *flags)
        assert p.returncode == 0, (flags, p.stderr[-1500:])
        t = [float(w.rstrip("%")) for w in p.stdout.split() if w.endswith("%")]
        assert len(t) == 2
        base = base or t
        assert abs(t[0] - base[0]) < 0.1 and abs(t[1] - base[1]) < 0.5, (flags, t, base)
        assert len([f for f in os.listdir(tmp_path / f"d{k}" / "images") if "image" in f]) == 3
    # strict libm changes no outcome on this fixture (identical transmissions); the 53-bit stream traces other rays
    a = np.fromfile(tmp_path / "d0" / "images" / sorted(os.listdir(tmp_path / "d0" / "images"))[0])
    b = np.fromfile(tmp_path / "d3" / "images" / sorted(os.listdir(tmp_path / "d3" / "images"))[0])
    assert not np.array_equal(a, b)


@pytest.mark.gpu
def test_entry_point_with_the_tracker_on(tmp_path, hip_library):
    """use_tracker (src/main.f90:72-74, :121-124, :183): the two path dumps are written, the images are not, the stats
    record is — by the torch-free process and by run_settings alike, byte for byte."""
    from opticalraytrace_amd.tracer import run_settings
    s = _tree(tmp_path, nphotons=2000, use_tracker=True, light_source="spot", bottle_file="clearBottle-small.params")
    p = _run(str(tmp_path / "bin"), "test_0.params")
    assert p.returncode == 0, p.stderr[-2000:]
    got = tmp_path / "data" / "images"
    names = sorted(os.listdir(got))
    dumps = [n for n in names if n.endswith("trace.dat")]
    assert len(dumps) == 2 and not [n for n in names if "_image" in n] and "trans-stats.dat" in names, names
    run_settings(s, str(tmp_path / "res"), str(tmp_path / "want"), verbose=False)
    for n in dumps + ["trans-stats.dat"]:
        assert open(got / n).read() == open(tmp_path / "want" / "images" / n).read(), n
    assert os.path.getsize(got / dumps[0]) > 1000


@pytest.mark.gpu
def test_entry_point_under_a_two_rank_launcher(tmp_path, hip_library):
    """`torch.distributed.run --nproc-per-node 2 -m opticalraytrace_amd <settings>`: every rank traces its shard, the
    image is summed over the ranks, rank 0 alone writes — the same three image files and stats record as the one-process
    run, byte for byte.  Rehearsed on one GPU (both ranks on device 0, ORT_DIST_BACKEND=gloo: RCCL needs a device per
    rank — that leg belongs to a multi-GPU node)."""
    import socket
    _tree(tmp_path, bottle_file="clearBottle-large.params", nphotons=300_001)
    p = _run(str(tmp_path), "res/test_0.params", "--data", "one", "--quiet")
    assert p.returncode == 0, p.stderr[-2000:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), ORT_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    q = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", "opticalraytrace_amd", "res/test_0.params", "--data", "two", "--device", "0"],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert q.returncode == 0, q.stderr[-3000:]
    assert q.stdout.count("Ring  transmitted:") == 1                     # rank 0 alone reports
    a, b = tmp_path / "one" / "images", tmp_path / "two" / "images"
    assert sorted(os.listdir(a)) == sorted(os.listdir(b)) and len(os.listdir(a)) == 4
    for f in os.listdir(a):
        assert open(a / f, "rb").read() == open(b / f, "rb").read(), f
