"""`python -m opticalraytrace_amd <settings>` — the process-level drop-in for `bin/raytrace <settings>`.

The reference is started as `cd bin && ./raytrace <file>` (install.sh:70-72); it reads
`../res/<file>` (src/setupMod.f90:51-56), the lens / bottle files next to it, and writes under
`../data/<folder>/` (src/setupMod.f90:124-131, src/main.f90:168-185): three raw float64 images, one
appended `trans-stats.dat` row, two "transmitted" lines on stdout.  Same contract here, the two
OpenMP loops running on the MI355X:

    python -m opticalraytrace_amd settings.params            # from bin/: ../res/settings.params -> ../data/
    python -m opticalraytrace_amd res/test_0.params --data data
    python -m opticalraytrace_amd test_0.params --res res --data data --device 1

Exit status: 0, or non-zero with a message on stderr where the reference would `error stop`
(unreadable / truncated input files, an unknown light source, ...) — which is what
`runner.py:47` (`subprocess.run(..., check=True)`) relies on.  Started under
`torch.distributed.run` (WORLD_SIZE > 1) every rank traces its shard of the rays, the image is
summed over RCCL and rank 0 writes the files; a process on its own does not load torch at all (ORT_NO_TORCH=0
makes it).  There is no CPU fallback: without a HIP device the run fails (exit status 3)."""
from __future__ import annotations

import argparse
import os
import sys

from .params import ParamsError, Settings


def _locate(settings: str, res: str | None):
    """(settings path, res dir): a bare name is looked up in --res (default ../res, as the
    reference does from bin/); an existing path brings its own directory as res."""
    if res is not None:
        p = settings if os.path.isabs(settings) or os.path.exists(settings) else os.path.join(res, settings)
        return p, res
    if os.path.exists(settings):
        return settings, os.path.dirname(os.path.abspath(settings))
    return os.path.join("..", "res", settings), os.path.join("..", "res")


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m opticalraytrace_amd", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("settings", help="settings file (20 positional lines, src/setupMod.f90:57-133)")
    ap.add_argument("--res", default=None, help="directory of the settings / .params files (default ../res)")
    ap.add_argument("--data", default=None, help="output root (default ../data, or data/ next to an explicit --res)")
    ap.add_argument("--device", type=int, default=None, help="HIP device (default LOCAL_RANK or 0)")
    ap.add_argument("--quiet", action="store_true")
    # beyond the reference's program (which has none of these: fp64, its runtime's draws, its libm):
    ap.add_argument("--fp32", action="store_true", help="trace in single precision (BASELINE configs[4]; ~2 x the rate, images within shot noise)")
    ap.add_argument("--strict-libm", action="store_true",
                    help="the light sources call glibc's own sin / cos: emitted rays equal the compiled reference's bit for bit (fp64 only; ~1.07 x the time)")
    ap.add_argument("--wide-draws", action="store_true",
                    help="53-bit uniforms, as random_number fills ran2's real(8) (src/random_mod.f90:39-46), instead of 32-bit ones (~1.09 x the time in fp64)")
    args = ap.parse_args(argv)
    if args.fp32 and args.strict_libm:
        ap.error("--strict-libm belongs to the exact fp64 path (the reference's arithmetic): not with --fp32")

    path, res_dir = _locate(args.settings, args.res)
    data_dir = args.data or (os.path.join("..", "data") if args.res is None and not os.path.exists(args.settings)
                             else "data")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    device = args.device if args.device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    try:
        settings = Settings.from_file(path)
        settings.validate()
        from .system import OpticalSystem
        system = OpticalSystem.from_settings(settings, res_dir)    # reads the lens / bottle files
    except (ParamsError, OSError) as e:
        print(f"raytrace: {e}", file=sys.stderr)
        return 1
    try:
        # a process of its own for ONE simulation (runner.py:26-47): no torch — its import is ~1 s of such a process — the
        # library's own accumulators (tracer.LocalTracer).  Decided HERE, for this call: nothing is written to the
        # environment (children would inherit it), and a process that already holds torch keeps it.
        local = world == 1 and "torch" not in sys.modules and os.environ.get("ORT_NO_TORCH", "1") == "1"
        from . import capi
        from .capi import OrtError
        if local:
            capi.load_library(no_torch=True)
        from .tracer import (LocalTracer, RayTracer, append_stats, output_basename, write_images,
                             write_tracker_files)
        group = None
        if world > 1:
            import torch
            import torch.distributed as dist
            # ORT_DIST_BACKEND=gloo (development): rehearse the multi-rank run where RCCL cannot be used — e.g. all
            # ranks on one GPU (`--device 0` on every rank)
            backend = os.environ.get("ORT_DIST_BACKEND", "nccl")
            torch.cuda.set_device(device)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        if local:
            tracer = LocalTracer(system, device=device)
        else:
            tracer = RayTracer(system, device=device, rank=rank, world=world, process_group=group)
        try:
            if args.fp32:
                tracer.ctx.set_precision(1)
            if args.strict_libm or args.wide_draws:
                tracer.ctx.set_kernel_variant(1 | (64 if args.strict_libm else 0) | (32 if args.wide_draws else 0))
            res = tracer.run()
            folder = os.path.join(data_dir, settings.data_folder)
            if rank == 0:
                os.makedirs(folder, exist_ok=True)                  # setupMod.f90:124-131
                if settings.use_tracker and world == 1:             # main.f90:72-74 (serial only there too)
                    write_tracker_files(tracer, system, folder)
        finally:
            tracer.close()
        if rank == 0:
            append_stats(folder, system, res)
            if not args.quiet:                                      # main.f90:180-181
                print(f"Ring  transmitted:  {res.ring_transmitted:8.2f}%")
                print(f"Point transmitted:  {res.point_transmitted:8.2f}%")
            if settings.make_images and not settings.use_tracker:   # main.f90:183-185
                write_images(res.image, os.path.join(folder, output_basename(system) + "_image"))
        if world > 1:
            dist.destroy_process_group()
    except (OrtError, RuntimeError) as e:
        print(f"raytrace: {e}", file=sys.stderr)
        return 3
    return 0


if __name__ == "__main__":
    sys.exit(main())
