"""Where the time of a batched sweep goes (runner.py:232-261 at 1e6 photons).

    python tools/sweep_profile.py [nphotons]

Prints (1) the simulations/s of lens_experiment_rates, (2) wall-clock sections of ONE batched lens experiment
(queueing = settings + system build, run_many's issue loop, the wait, the copy back, the output files) and
(3) the top of a cProfile of the same.  Development tool: its output goes to profiles/rNN/sweep_profile.log.
"""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from opticalraytrace_amd import sweeps, tracer as tracer_mod          # noqa: E402
from opticalraytrace_amd.capi import build_id                          # noqa: E402


def sections(nphotons):
    marks = {}
    orig_run_many = tracer_mod.ShardedRun.run_many
    orig_write = sweeps.write_outputs

    def timed_run_many(self, systems, *a, **k):
        t0 = time.perf_counter()
        out = orig_run_many(self, systems, *a, **k)
        marks["run_many"] = marks.get("run_many", 0.0) + time.perf_counter() - t0
        return out

    def timed_write(*a, **k):
        t0 = time.perf_counter()
        orig_write(*a, **k)
        marks["write_outputs"] = marks.get("write_outputs", 0.0) + time.perf_counter() - t0

    tracer_mod.ShardedRun.run_many = timed_run_many
    sweeps.write_outputs = timed_write
    try:
        with tempfile.TemporaryDirectory(prefix="ort_sweep_") as tmp:
            sw = sweeps.Sweep(nphotons=nphotons, data_dir=tmp, batched=True)
            try:
                sw.run("warm.params", light_source="point", make_images=False, data_folder="warm")
                sw.flush()
                marks.clear()
                for rep in range(3):
                    t0 = time.perf_counter()
                    sw.lens_experiment()
                    el = time.perf_counter() - t0
                    print(f"  rep {rep}: {75 / el:8.1f} simulations/s  total {el * 1e3:7.2f} ms  run_many {marks.get('run_many', 0) * 1e3:7.2f} ms  "
                          f"write_outputs {marks.get('write_outputs', 0) * 1e3:7.2f} ms  "
                          f"queueing (settings + systems) {(el - marks.get('run_many', 0) - marks.get('write_outputs', 0)) * 1e3:7.2f} ms")
                    marks.clear()
                pr = cProfile.Profile()
                pr.enable()
                sw.lens_experiment()
                pr.disable()
                s = io.StringIO()
                pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
                print(s.getvalue())
            finally:
                sw.close()
    finally:
        tracer_mod.ShardedRun.run_many = orig_run_many
        sweeps.write_outputs = orig_write


if __name__ == "__main__":
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
    print(f"build {build_id()}  nphotons {n}")
    r = sweeps.lens_experiment_rates(n)
    for mode in ("batched", "queued", "one_by_one"):
        m = r[mode]
        print(f"{mode:12s} {m['simulations_per_s']:8.1f} simulations/s  ({m['seconds'] * 1e3:.2f} ms for {m['simulations']}); "
              f"first call of a fresh context {m['first_call_simulations_per_s']:8.1f} /s ({m['first_call_seconds'] * 1e3:.2f} ms)")
    sections(n)
