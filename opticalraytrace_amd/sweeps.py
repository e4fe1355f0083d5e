"""Experiment sweeps of the reference's `runner.py` that stay on the hot path.

    python -m opticalraytrace_amd.sweeps -p -i -o -l [--nphotons N] [--data-dir DIR]

Same flags, same loops, same settings defaults and the same `data/<folder>/` outputs as
runner.py (`-p` :136-155, `-i` :158-186, `-o` :189-208, `-l` :232-261); every simulation is
one `run_settings` call on ONE reused GPU context instead of one
`./install.sh -n 32 -f <settings>` process (runner.py:26-47).  `-s` (spot diagrams: spot source
+ ray-path tracker dumps, runner.py:113-133) is supported too; `-b` (Bessel image source) and
the iSORS comparison use emitters outside this path and are refused with a message.
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import Dict, Iterable, List, Optional, Tuple

from .params import Settings, resource_dir
from .tracer import RayTracer, RunResult, run_settings

# runner.py:371-372 / :384-385
BOTTLES = [("clearBottle-large.params", True), ("clearBottle-small.params", True),
           ("clearBottle-ellipse.params", True), ("clearBottle-small.params", False)]
LENS_BOTTLES = BOTTLES[:3]
IRISES = ["before", "after", "none"]                       # runner.py:172
IRIS_SIZES = [1.0, 0.8, 0.6, 0.4, 0.2]                     # runner.py:173
L2_FOCALS = ["59.8", "49.8", "39.9", "34.9", "29.9"]       # runner.py:249
L3_FOCALS = ["40.0", "45.0", "50.0", "60.0", "75.0"]       # runner.py:251
OFFSETS_MM = list(range(4, 17, 2))                         # runner.py:200


class Sweep:
    def __init__(self, nphotons: int = 1_000_000_000, res_dir: Optional[str] = None,
                 data_dir: str = "data", device: int = 0, verbose: bool = False,
                 settings_dir: Optional[str] = None):
        self.nphotons, self.res_dir, self.data_dir = nphotons, res_dir or resource_dir(), data_dir
        self.device, self.verbose, self.settings_dir = device, verbose, settings_dir
        self.tracer: Optional[RayTracer] = None
        self.results: List[Tuple[str, Settings, RunResult]] = []

    def close(self) -> None:
        if self.tracer is not None:
            self.tracer.close()
            self.tracer = None

    def run(self, name: str, **over) -> RunResult:
        """make_settings + run_sim of runner.py for one parameter set."""
        s = Settings(nphotons=self.nphotons, **over)
        s.validate()
        if self.settings_dir:                              # runner.py:106-110: keep the settings text
            os.makedirs(self.settings_dir, exist_ok=True)
            s.write(os.path.join(self.settings_dir, name))
        if self.tracer is None:
            from .system import OpticalSystem
            self.tracer = RayTracer(OpticalSystem.from_settings(s, self.res_dir), device=self.device)
        res = run_settings(s, self.res_dir, self.data_dir, self.device, self.verbose, self.tracer)
        self.results.append((name, s, res))
        return res

    # ---- runner.py experiments -------------------------------------------------
    def spot_diagrams(self, bottles=BOTTLES) -> None:      # -s, runner.py:113-133
        for i, (bottle, use) in enumerate(bottles):
            keep, self.nphotons = self.nphotons, 100
            try:
                self.run(f"test_{i}.params", use_tracker=True, light_source="spot", bottle_file=bottle,
                         use_bottle=use, data_folder="spot-diag")
            finally:
                self.nphotons = keep

    def point_images(self, bottles=BOTTLES) -> None:       # -p, runner.py:136-155
        for i, (bottle, use) in enumerate(bottles):
            self.run(f"test_{i}.params", light_source="point", make_images=True,
                     bottle_file=bottle, use_bottle=use, data_folder="images")

    def iris_experiment(self, bottles=BOTTLES) -> None:    # -i, runner.py:158-186
        for i, (bottle, use) in enumerate(bottles):
            for iris in IRISES:
                for size in IRIS_SIZES:
                    self.run(f"test_{i}_{iris}_{size}.params", light_source="point", make_images=True,
                             bottle_file=bottle, use_bottle=use, iris=iris, iris_size=size,
                             data_folder="iris")
                    if iris == "none":
                        break                              # one size is enough without an iris (:184-186)

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

    def lens_experiment(self, bottles=LENS_BOTTLES) -> None:   # -l, runner.py:232-261
        for k, f3 in enumerate(L3_FOCALS):
            for j, f2 in enumerate(L2_FOCALS):
                for i, (bottle, use) in enumerate(bottles):
                    self.run(f"test_{i}_{j}_{k}.params", light_source="point", make_images=False,
                             bottle_file=bottle, use_bottle=use, data_folder="images-lens",
                             L3_file=f"achromaticDoublet-f{f3}mm.params",
                             L2_file=f"planoConvex-f{f2}mm.params")


def main(argv: Optional[Iterable[str]] = None) -> int:
    ap = argparse.ArgumentParser(usage="%(prog)s [OPTION]", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-s", "--spot", action="store_true", help="Create spot diagrams.")
    ap.add_argument("-p", "--point", action="store_true", help="Create point/ring images.")
    ap.add_argument("-b", "--bessel", action="store_true", help="(not on this path)")
    ap.add_argument("-o", "--offset", action="store_true", help="Run offset experiment on large bottle.")
    ap.add_argument("-i", "--iris", action="store_true", help="Run iris experiment on bottles.")
    ap.add_argument("-l", "--lens", action="store_true", help="Run lens experiments.")
    ap.add_argument("-a", "--all", action="store_true", help="Run all experiments of this path.")
    ap.add_argument("--nphotons", type=int, default=1_000_000_000)     # runner.py:69
    ap.add_argument("--data-dir", default="data")
    ap.add_argument("--res-dir", default=None)
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    if args.bessel:
        print("-b needs the image emitter (SURVEY §8 f2): not on the MI355X hot path", file=sys.stderr)
        return 2
    sw = Sweep(args.nphotons, args.res_dir, args.data_dir, args.device, verbose=True,
               settings_dir=os.path.join(args.data_dir, "settings"))
    try:
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
