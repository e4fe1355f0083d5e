"""opticalraytrace_amd — the per-ray hot path of lewisfish/OpticalRayTrace on MI355X.

  params   readers for the reference's settings / .params files, lens + bottle models
  system   run set-up (derived constants, both wavelengths) and the surface list
  capi     ctypes binding of the C ABI (include/ort.h, csrc/libort_hip.so)
  tracer   host driver mirroring `program raytrace`, multi-GPU sharding + RCCL reduce
"""
from .params import (AchromaticDoublet, GlassBottle, ParamsError, PlanoConvex, Settings,  # noqa: F401
                     resource_dir)
from .system import OpticalSystem  # noqa: F401

__all__ = ["Settings", "OpticalSystem", "PlanoConvex", "AchromaticDoublet", "GlassBottle",
           "ParamsError", "resource_dir"]
