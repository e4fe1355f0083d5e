"""Experiment sweeps of the reference's `runner.py`.

    python -m opticalraytrace_amd.sweeps -s -p -b -i -o -l --isors [--nphotons N] [--data-dir DIR]

Same flags, same loops, same settings defaults and the same `data/<folder>/` outputs as
runner.py (`-s` :113-133, `-p` :136-155, `-b` :209-228, `-i` :158-186, `-o` :189-208, `-l` :232-261,
and `iSORS_vs_Bessel` :267-320, which runner.py defines but wires to no flag: `--isors` here);
the simulations of an experiment are QUEUED on one reused GPU context — system re-staged asynchronously,
one device array of images for the whole batch, one wait and one copy back at the end (RayTracer.run_many)
— instead of one `./install.sh -n 32 -f <settings>` process each (runner.py:26-47); `--one-by-one` runs
them one `run_settings` call at a time (same files, bit for bit).  (Exception: a simulation that brings a new `image`
source table waits for the queue — ort_set_image_source is synchronous, the context holds one table.)  A settings file
that cannot be built into a system fails at the `run()` call that names it; if a batch fails on the device the queue is
re-run one simulation at a time, so every simulation before the failing one still leaves its files.  `-b` needs the Bessel image
`bessel-smear.dat` (bpm.py's output, not shipped by the reference) in the res directory and fails
with a clear message without it.
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import Dict, Iterable, List, Optional, Tuple

from .params import Settings, resource_dir
from .tracer import RayTracer, RunResult, run_settings, write_outputs

# runner.py:371-372 / :384-385
BOTTLES = [("clearBottle-large.params", True), ("clearBottle-small.params", True),
           ("clearBottle-ellipse.params", True), ("clearBottle-small.params", False)]
LENS_BOTTLES = BOTTLES[:3]
IRISES = ["before", "after", "none"]                       # runner.py:172
IRIS_SIZES = [1.0, 0.8, 0.6, 0.4, 0.2]                     # runner.py:173
L2_FOCALS = ["59.8", "49.8", "39.9", "34.9", "29.9"]       # runner.py:249
L3_FOCALS = ["40.0", "45.0", "50.0", "60.0", "75.0"]       # runner.py:251
OFFSETS_MM = list(range(4, 17, 2))                         # runner.py:200


def _experiment(fn):
    """An experiment queues its simulations and ends with the batch run."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        try:
            out = fn(self, *a, **k)
        except BaseException:
            # the experiment failed while queueing: what was queued before the failing call still runs — as runner.py
            # would have left the outputs of every simulation before the bad one — but a failure of THAT must not
            # replace the experiment's own exception
            try:
                self.flush()
            except Exception:
                pass
            raise
        self.flush()
        return out
    return wrapped


class Sweep:
    MAX_BATCH = 128        # simulations per device array: 128 x 1.29 MB of images

    def __init__(self, nphotons: int = 1_000_000_000, res_dir: Optional[str] = None,
                 data_dir: str = "data", device: int = 0, verbose: bool = False,
                 settings_dir: Optional[str] = None, batched: bool = True, multi_system: bool = True):
        self.nphotons, self.res_dir, self.data_dir = nphotons, res_dir or resource_dir(), data_dir
        self.device, self.verbose, self.settings_dir = device, verbose, settings_dir
        self.batched = batched
        self.multi_system = multi_system     # a batch goes through ort_trace_batch (multi-system launches); False: queued one after the other
        self.tracer: Optional[RayTracer] = None
        self.results: List[Tuple[str, Settings, RunResult]] = []
        self._pending: List[Tuple[str, Settings, object]] = []     # (name, settings, the system built from them)

    def close(self) -> None:
        try:
            self.flush()
        finally:
            if self.tracer is not None:
                self.tracer.close()
                self.tracer = None

    def _tracer_for(self, s: Settings) -> RayTracer:
        if self.tracer is None:
            from .system import OpticalSystem
            self.tracer = RayTracer(OpticalSystem.from_settings(s, self.res_dir), device=self.device)
            self.tracer.multi_system_launches = self.multi_system
        return self.tracer

    def run(self, name: str, **over) -> Optional[RunResult]:
        """make_settings + run_sim of runner.py for one parameter set.  Batched: the simulation is queued and
        runs at the next flush() (every experiment ends with one); its result then appears in `results`."""
        s = Settings(nphotons=self.nphotons, **over)
        s.validate()
        if self.settings_dir:                              # runner.py:106-110: keep the settings text
            os.makedirs(self.settings_dir, exist_ok=True)
            s.write(os.path.join(self.settings_dir, name))
        # the tracker dumps ray paths through the synchronous per-ray entry: such a simulation runs on its own
        if not self.batched or s.use_tracker:
            self.flush()
            res = run_settings(s, self.res_dir, self.data_dir, self.device, self.verbose, self._tracer_for(s))
            self.results.append((name, s, res))
            return res
        # the system is built (lens / bottle files read and checked) HERE, so that a bad settings file fails at the call
        # that names it, as the reference program would have — not later, inside a batch of 128
        from .system import OpticalSystem
        self._pending.append((name, s, OpticalSystem.from_settings(s, self.res_dir)))
        if len(self._pending) >= self.MAX_BATCH:
            self.flush()
        return None

    def flush(self) -> None:
        """Run the queued simulations as one batch, then leave each one's files, in queue order.  If the library reports
        an error for the batch (OrtError; anything else is a programming error and is raised as it is) the queue is
        re-run one simulation at a time, so that — like runner.py, which runs them one by one — every simulation before
        the failing one leaves its outputs; the error of the failing one is raised after that, and what had not run by
        then stays queued (`_pending`)."""
        if not self._pending:
            return
        from .capi import OrtError
        pending, self._pending = self._pending, []
        tracer = self._tracer_for(pending[0][1])
        try:
            # (a simulation without images — make_images false, src/main.f90:183 — brings back its counters only)
            results = tracer.run_many([system for _, _, system in pending], want_images=[s.make_images for _, s, _ in pending])
        except OrtError as e:
            # A failed call leaves the queue as it was for the caller to see; the simulations are then run one at a time so
            # that — like runner.py, which runs them one by one — every simulation before the failing one leaves its files.
            # After a device fault (ORT_E_HIP) the HIP error is sticky for this context: a new one is built first; if the
            # device itself is gone that fails too and the sweep ends there with what had run.
            done = 0
            self._pending = pending
            try:
                if "ORT_E_HIP" in str(e):
                    tracer.close()
                    self.tracer = None
                    tracer = self._tracer_for(pending[0][1])
                for name, s, system in pending:
                    tracer.set_system(system)
                    res = tracer.run(s.nphotons)
                    write_outputs(system, res, self.data_dir, self.verbose)
                    self.results.append((name, s, res))
                    done += 1
            finally:
                self._pending = pending[done:]      # what did not run is still queued
            return
        for (name, s, system), res in zip(pending, results):
            write_outputs(system, res, self.data_dir, self.verbose)
            self.results.append((name, s, res))

    # ---- runner.py experiments -------------------------------------------------
    @_experiment
    def spot_diagrams(self, bottles=BOTTLES) -> None:      # -s, runner.py:113-133
        for i, (bottle, use) in enumerate(bottles):
            keep, self.nphotons = self.nphotons, 100
            try:
                self.run(f"test_{i}.params", use_tracker=True, light_source="spot", bottle_file=bottle,
                         use_bottle=use, data_folder="spot-diag")
            finally:
                self.nphotons = keep

    @_experiment
    def point_images(self, bottles=BOTTLES) -> None:       # -p, runner.py:136-155
        for i, (bottle, use) in enumerate(bottles):
            self.run(f"test_{i}.params", light_source="point", make_images=True,
                     bottle_file=bottle, use_bottle=use, data_folder="images")

    @_experiment
    def bessel_images(self, bottles=BOTTLES) -> None:      # -b, runner.py:209-228
        from .params import ParamsError
        src = os.path.join(self.res_dir, Settings().image_source)
        if not os.path.exists(src):
            raise ParamsError(f"{src}: the image source file (bpm.py's Bessel beam, 512 x 512 float64) is "
                              "missing — the reference opens it with status='old' (src/sourceMod.f90:381)")
        for i, (bottle, use) in enumerate(bottles):
            self.run(f"test_{i}.params", light_source="image", make_images=True,
                     bottle_file=bottle, use_bottle=use, data_folder="images")

    @_experiment
    def isors_vs_bessel(self) -> None:                     # iSORS_vs_Bessel, runner.py:267-320
        """isors source against the point source with the bottle moved so that the Bessel ring has
        the same spatial offset; 7 offsets 0 ... 1.5 mm each."""
        import math
        ring_width, alpha_deg, n_axicon, l2_file = 0.5e-3, 5.0, 1.45, "planoConvex-f39.9mm.params"
        from .params import GlassBottle, PlanoConvex
        l2fb = PlanoConvex.from_file(os.path.join(self.res_dir, l2_file), 785e-9).fb
        # runner.py reads radius a from clearBottle-small_0.0mm.params — a 14-line file the reference's
        # own reader cannot open; the same bottle is clearBottle-small.params
        bottle0 = "clearBottle-small_0.0mm.params"
        if not os.path.exists(os.path.join(self.res_dir, bottle0)):
            bottle0 = "clearBottle-small.params"
        radius_a = GlassBottle.from_file(os.path.join(self.res_dir, bottle0), 785e-9).radiusa
        alpha = alpha_deg * math.pi / 180.0
        init_dist = 97.3e-3                                # distance from axicon to L1 (runner.py:295)
        common = dict(use_tracker=False, make_images=True, image_diameter=1e-2, data_folder="iSORS_vs_Bessel",
                      use_bottle=True, ring_width=ring_width, alpha=alpha_deg, n_axicon=n_axicon, L2_file=l2_file)
        k = 0
        for source in ("isors", "point"):
            for j in range(7):
                offset = 1.5e-3 * j / 6.0                  # np.linspace(0., 1.5e-3, 7)
                if source == "isors":
                    bottle = bottle0
                else:                                      # create_bottle_file(prop_offset), runner.py:322-347
                    prop = (l2fb * (offset + ring_width)) / (init_dist * math.tan(alpha * (n_axicon - 1))) - radius_a
                    bottle = self._isors_bottle_file(prop, j)
                self.run(f"test_{k}.params", light_source=source, isors_offset=offset, bottle_file=bottle, **common)
                k += 1

    def _isors_bottle_file(self, z: float, j: int) -> str:
        """clearBottle-small_iSORS.params of runner.py:322-347, written next to the outputs (the
        packaged res directory is not touched; one file per offset, so that the kept settings files
        stay runnable); an absolute path, which the readers accept."""
        lines = ["2.d-3", "17.5d-3", "17.5d-3", "0.0", "0.0", repr(float(z)), "1.5130", "0.003169", "0.003962",
                 "1.35265", "0.00306", "0.00002"]
        d = os.path.abspath(self.settings_dir or self.data_dir)
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, f"clearBottle-small_iSORS_{j}.params")
        with open(path, "w") as f:
            f.write("\n".join(lines) + "\n")
        return path

    @_experiment
    def iris_experiment(self, bottles=BOTTLES) -> None:    # -i, runner.py:158-186
        for i, (bottle, use) in enumerate(bottles):
            for iris in IRISES:
                for size in IRIS_SIZES:
                    self.run(f"test_{i}_{iris}_{size}.params", light_source="point", make_images=True,
                             bottle_file=bottle, use_bottle=use, iris=iris, iris_size=size,
                             data_folder="iris")
                    if iris == "none":
                        break                              # one size is enough without an iris (:184-186)

    @_experiment
    def offset_experiment(self) -> None:                   # -o, runner.py:189-208
        # runner.py asks for -4 ... -16 mm, but res/ only ships files down to -14 mm: the
        # reference's own sweep dies on its 7th simulation (open status="old").  Here the
        # missing file is reported and skipped.
        for i, off in enumerate(OFFSETS_MM):
            if not os.path.exists(os.path.join(self.res_dir, f"clearBottle-large_-{off}mm.params")):
                print(f"offset experiment: clearBottle-large_-{off}mm.params is not shipped; skipped",
                      file=sys.stderr)
                continue
            self.run(f"test_{i}.params", light_source="point", make_images=True,
                     bottle_file=f"clearBottle-large_-{off}mm.params", data_folder="images-offset")

    @_experiment
    def lens_experiment(self, bottles=LENS_BOTTLES) -> None:   # -l, runner.py:232-261
        for k, f3 in enumerate(L3_FOCALS):
            for j, f2 in enumerate(L2_FOCALS):
                for i, (bottle, use) in enumerate(bottles):
                    self.run(f"test_{i}_{j}_{k}.params", light_source="point", make_images=False,
                             bottle_file=bottle, use_bottle=use, data_folder="images-lens",
                             L3_file=f"achromaticDoublet-f{f3}mm.params",
                             L2_file=f"planoConvex-f{f2}mm.params")


def lens_experiment_rates(nphotons: int, device: int = 0, modes=("batched", "queued", "one_by_one"), process_samples: int = 0, repeats: int = 3) -> dict:
    """Simulations per second of runner.py's lens experiment (75 systems, :232-261) on one GPU: `batched` = queued as one
    batch through ort_trace_batch (multi-system launches: one launch per surface program and loop over all 75), `queued` =
    the same batch queued one simulation after the other on the context (asynchronous ort_set_system / ort_attach_buffers /
    ort_trace, one wait), `one_by_one` = one `run_settings` at a time on a reused context, and — `process_samples`
    simulations of it — one `python -m opticalraytrace_amd <settings>` PROCESS per simulation, which is runner.py's own model
    (:26-47).  Per mode: the FIRST experiment of a fresh context (`first_call_*`: scratch allocation, the other bottles'
    code objects, clocks coming out of idle) and the median of `repeats` more (`simulations_per_s`).  Outputs (the stats
    rows; the lens experiment writes no images) go to a scratch directory."""
    import subprocess
    import tempfile
    import time
    out = {"experiment": "lens_experiment (runner.py:232-261): 5 doublets x 5 plano-convex lenses x 3 bottles", "nphotons": nphotons}
    with tempfile.TemporaryDirectory(prefix="ort_sweep_") as tmp:
        for mode in modes:
            sw = Sweep(nphotons=nphotons, data_dir=os.path.join(tmp, mode), device=device, batched=(mode != "one_by_one"),
                       multi_system=(mode == "batched"))
            times = []
            try:
                sw.run("warm.params", light_source="point", make_images=False, data_folder="warm")   # context, code objects
                sw.flush()
                for _ in range(1 + max(repeats, 0)):
                    t0 = time.perf_counter()
                    sw.lens_experiment()
                    times.append(time.perf_counter() - t0)
            finally:
                sw.close()
            n = (len(sw.results) - 1) // len(times)
            steady = sorted(times[1:])[len(times[1:]) // 2] if len(times) > 1 else times[0]
            out[mode] = {"simulations": n, "seconds": steady, "simulations_per_s": n / steady, "rays_per_s": 2.0 * nphotons * n / steady,
                         "first_call_seconds": times[0], "first_call_simulations_per_s": n / times[0], "all_seconds": times}
        if process_samples:
            sw = Sweep(nphotons=nphotons, data_dir=os.path.join(tmp, "p"), settings_dir=os.path.join(tmp, "settings"), batched=True)
            sw.tracer = None
            names = []
            for k, f3 in enumerate(L3_FOCALS[:process_samples]):
                s = Settings(nphotons=nphotons, light_source="point", make_images=False, bottle_file=LENS_BOTTLES[0][0],
                             data_folder="images-lens", L3_file=f"achromaticDoublet-f{f3}mm.params",
                             L2_file=f"planoConvex-f{L2_FOCALS[k % len(L2_FOCALS)]}mm.params")
                os.makedirs(sw.settings_dir, exist_ok=True)
                s.write(os.path.join(sw.settings_dir, f"p_{k}.params"))
                names.append(f"p_{k}.params")
            times = []
            for name in names:
                t0 = time.perf_counter()
                subprocess.run([sys.executable, "-m", "opticalraytrace_amd", os.path.join(sw.settings_dir, name), "--res", sw.res_dir,
                                "--data", os.path.join(tmp, "p"), "--quiet", "--device", str(device)], check=True,
                               cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
                times.append(time.perf_counter() - t0)
            out["process_per_simulation"] = {"samples": len(times), "seconds_per_simulation": sum(times) / len(times),
                                             "simulations_per_s": len(times) / sum(times),
                                             "what": "python -m opticalraytrace_amd <settings>: interpreter + context + run + files per simulation"}
    return out


def main(argv: Optional[Iterable[str]] = None) -> int:
    ap = argparse.ArgumentParser(usage="%(prog)s [OPTION]", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-s", "--spot", action="store_true", help="Create spot diagrams.")
    ap.add_argument("-p", "--point", action="store_true", help="Create point/ring images.")
    ap.add_argument("-b", "--bessel", action="store_true", help="Create bessel/ring diagrams.")
    ap.add_argument("--isors", action="store_true", help="iSORS vs Bessel comparison (runner.py iSORS_vs_Bessel).")
    ap.add_argument("-o", "--offset", action="store_true", help="Run offset experiment on large bottle.")
    ap.add_argument("-i", "--iris", action="store_true", help="Run iris experiment on bottles.")
    ap.add_argument("-l", "--lens", action="store_true", help="Run lens experiments.")
    ap.add_argument("-a", "--all", action="store_true", help="Run all experiments of this path.")
    ap.add_argument("--nphotons", type=int, default=1_000_000_000)     # runner.py:69
    ap.add_argument("--data-dir", default="data")
    ap.add_argument("--res-dir", default=None)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--one-by-one", action="store_true", help="one simulation at a time (default: queued in batches)")
    args = ap.parse_args(argv)
    sw = Sweep(args.nphotons, args.res_dir, args.data_dir, args.device, verbose=True,
               settings_dir=os.path.join(args.data_dir, "settings"), batched=not args.one_by_one)
    from .params import ParamsError
    try:
        if args.bessel or args.all:
            try:
                sw.bessel_images()
            except ParamsError as e:
                print(f"-b: {e}", file=sys.stderr)
                if not args.all:
                    return 2
        if args.isors:
            sw.isors_vs_bessel()
        if args.spot or args.all:
            sw.spot_diagrams()
        if args.point or args.all:
            sw.point_images()
        if args.offset or args.all:
            sw.offset_experiment()
        if args.iris or args.all:
            sw.iris_experiment()
        if args.lens or args.all:
            sw.lens_experiment()
    finally:
        sw.close()
    print(f"{len(sw.results)} simulations")
    return 0


if __name__ == "__main__":
    sys.exit(main())
