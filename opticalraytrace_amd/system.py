"""Run set-up of the reference, on the host: settings -> lenses at both wavelengths
-> derived constants -> the flat surface table the kernels stage into LDS.

Reference file:line followed here
  * src/setupMod.f90:115-121   which objects are built, with which offsets
  * src/main.f90:51-81         cosThetaMax, bottle clamp, ring radii, img_plane_1
  * src/main.f90:113-116       L2/L3 rebuilt at 843 nm for the point phase
  * src/optics_system.f90:28-49  surface order of `telescope`
  * src/lens.f90:230-350, :425-481, :531-645  per-surface constants
"""
from __future__ import annotations

import functools
import math
import os
from dataclasses import dataclass
from typing import List, Optional

from .params import (AchromaticDoublet, GlassBottle, ParamsError, PlanoConvex, Settings,
                     resource_dir)

PI = 4.0 * math.atan(1.0)          # src/constants.f90:5
TWOPI = 2.0 * 4.0 * math.atan(1.0)
L1_FB = 97.3e-3                    # literal in src/main.f90:66 (fb of the never-loaded L1)
POINT_WAVELENGTH = 843e-9          # src/main.f90:114
NA_SINE = 0.22                     # src/imageMod.f90:40

# surface kinds / flags: numeric values of include/ort.h
SURF_PLANE, SURF_SPHERE, SURF_CYLINDER, SURF_ELLIPSE, SURF_IRIS, SURF_IMAGE = range(6)
F_SKIP_ON_REFLECT, F_MISS_IS_HELP3, F_BOTTLE, F_TRACK, F_SCATTER = 1, 2, 4, 8, 16
MAX_SURFACES = 12


@functools.lru_cache(maxsize=16)
def acos_threshold(na: float) -> float:
    """min{x in [0,1] : acos(x) <= na} over the doubles, by bisection on the bit pattern."""
    import struct

    def f2i(x: float) -> int:
        return struct.unpack("<q", struct.pack("<d", x))[0]

    def i2f(i: int) -> float:
        return struct.unpack("<d", struct.pack("<q", i))[0]

    lo, hi = f2i(0.0), f2i(1.0)          # acos(lo) > na, acos(hi) = 0 <= na
    if not (math.acos(0.0) > na >= 0.0):
        raise ValueError("na outside (0, pi/2)")
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if math.acos(i2f(mid)) <= na:
            hi = mid
        else:
            lo = mid
    # acos must be monotone around the threshold for the compare to be equivalent
    for k in range(1, 65):
        assert math.acos(i2f(hi + k)) <= na and math.acos(i2f(lo - k + 1 - 1)) > na
    return i2f(hi)


@dataclass
class Surface:
    kind: int
    cz: float
    radius: float = 0.0
    n1: float = 1.0
    n2: float = 1.0
    aperture: float = -1.0
    flags: int = 0
    cx: float = 0.0
    cy: float = 0.0
    radius_b: float = 0.0
    name: str = ""
    mua: float = 0.0
    mus: float = 0.0
    hgg: float = 0.0
    scat_radius: float = 0.0


@dataclass
class OpticalSystem:
    """State of `program raytrace` just before its first loop (src/main.f90:43-81)."""
    settings: Settings
    bottle: GlassBottle
    L2: List[PlanoConvex]            # [phase-1]: settings wavelength, 843 nm
    L3: List[AchromaticDoublet]
    cos_theta_max: float
    distance: float
    bessel_diameter: float
    r1: float
    r2: float
    img_plane: float
    bottle_moved: bool = False
    crs_spot_size: float = 0.0       # spot_size after setupMod.f90:136
    image_source_path: str = ""      # setupMod.f90:120-121 (read only for the `image` source)
    image_seed: int = 123456789      # keys the histogram's rounding draws (image_source.py)

    # ------------------------------------------------------------------
    @classmethod
    def from_settings(cls, settings: Settings, res_dir: Optional[str] = None) -> "OpticalSystem":
        res_dir = res_dir or resource_dir()
        s = settings
        if s.light_source not in ("point", "spot", "crs", "image", "isors"):
            raise ParamsError("No such source type!")              # setupMod.f90:98
        wl = s.wavelength
        bottle = GlassBottle.from_file(os.path.join(res_dir, s.bottle_file), wl)
        L2, L3 = [], []
        for w in (wl, POINT_WAVELENGTH):
            l2 = PlanoConvex.from_file(os.path.join(res_dir, s.L2_file), w)
            l3 = AchromaticDoublet.from_file(os.path.join(res_dir, s.L3_file), w,
                                             2.0 * l2.fb + l2.thickness)
            L2.append(l2)
            L3.append(l3)
        a = L2[0]
        alpha = s.alpha * PI / 180.0                           # setupMod.f90:61
        angle = math.atan(a.radius / a.fb)                     # main.f90:51
        cos_theta_max = math.cos(angle)
        # setupMod.f90:135-136, evaluated before the clamp below as in the reference
        offset0 = bottle.radiusa + bottle.centre[2]
        crs_spot_size = (s.crs_spot_size * (a.fb - offset0)) / a.fb
        moved = False
        if a.fb <= bottle.radiusa + bottle.centre[2]:          # main.f90:54-58
            bottle.centre[2] = a.fb - bottle.radiusa - 2e-3
            moved = True
        if s.light_source == "isors":                          # main.f90:60-64
            distance = bottle.radiusa + s.isors_offset
        else:
            distance = bottle.radiusa + bottle.centre[2]
        bessel = distance * L1_FB * math.tan(alpha * (s.n_axicon - 1)) / a.fb   # main.f90:66
        r1 = bessel - s.ring_width                             # main.f90:68-70
        half = bessel / 2.0
        r2 = half * half
        r1 = r1 * r1
        img_plane = 2.0 * (a.fb + L3[0].fb) + a.thickness + L3[0].thickness     # main.f90:81
        return cls(s, bottle, L2, L3, cos_theta_max, distance, bessel, r1, r2, img_plane, moved,
                   crs_spot_size, os.path.join(res_dir, s.image_source))

    # ------------------------------------------------------------------
    def surfaces(self, phase: int) -> List[Surface]:
        """The ordered surface list one ray of `phase` meets (1 ring, 2 point)."""
        s, b = self.settings, self.bottle
        l2, l3 = self.L2[phase - 1], self.L3[phase - 1]
        out: List[Surface] = []
        if phase == 2 and s.use_bottle:                        # main.f90:145-148
            cx, cy, cz = b.centre
            if b.ellipse:                                      # lens.f90:249-253, :300-301
                out.append(Surface(SURF_ELLIPSE, cz, b.radiusa - b.thickness, b.ncontents, b.nbottle,
                                   flags=F_SKIP_ON_REFLECT | F_BOTTLE, cx=cx, cy=cy,
                                   radius_b=b.radiusb - b.thickness, name="bottle inner"))
                out.append(Surface(SURF_ELLIPSE, cz, b.radiusa / 2.0, b.nbottle, 1.0,
                                   flags=F_SKIP_ON_REFLECT | F_BOTTLE, cx=cx, cy=cy,
                                   radius_b=b.radiusb / 2.0, name="bottle outer"))
            else:                                              # lens.f90:255, :303
                out.append(Surface(SURF_CYLINDER, cz, b.radiusa - b.thickness, b.ncontents, b.nbottle,
                                   flags=F_SKIP_ON_REFLECT | F_BOTTLE, cx=cx, cy=cy, name="bottle inner"))
                out.append(Surface(SURF_CYLINDER, cz, b.radiusa, b.nbottle, 1.0,
                                   flags=F_SKIP_ON_REFLECT | F_BOTTLE, cx=cx, cy=cy, name="bottle outer"))
            # in-bottle scattering (lens.f90:218-219): contents before the inner wall (g = .65,
            # :262-282), glass before the outer wall (g = 0.9, :312-333); tauint always walks
            # inside the CIRCULAR cylinder radiusa - thickness / radiusa
            if b.mua_c + b.mus_c != 0.0:
                out[0].flags |= F_SCATTER
                out[0].mua, out[0].mus, out[0].hgg = b.mua_c, b.mus_c, 0.65
                out[0].scat_radius = b.radiusa - b.thickness
            if b.mua_b + b.mus_b != 0.0:
                out[1].flags |= F_SCATTER
                out[1].mua, out[1].mus, out[1].hgg = b.mua_b, b.mus_b, 0.9
                out[1].scat_radius = b.radiusa
        # plano-convex, flat face first: lens.f90:446-459 (reflection flag ignored)
        out.append(Surface(SURF_PLANE, l2.flat_z, 0.0, l2.n1, l2.n2, aperture=l2.radius, name="L2 flat"))
        out.append(Surface(SURF_SPHERE, l2.centre_z, l2.curve_radius, l2.n2, l2.n1,
                           flags=F_SKIP_ON_REFLECT, name="L2 curved"))       # lens.f90:462-479
        if s.iris == "before":                                 # lens.f90:551-565
            out.append(Surface(SURF_IRIS, l3.centre1_z - l3.R1, aperture=l3.radius * s.iris_size,
                               name="iris before"))
        out.append(Surface(SURF_SPHERE, l3.centre1_z, l3.R1, l3.n1, l3.n2, aperture=l3.radius * 1.0,
                           flags=F_SKIP_ON_REFLECT, name="L3 face 1"))       # lens.f90:568-590
        out.append(Surface(SURF_SPHERE, l3.centre2_z, l3.R2, l3.n2, l3.n3,
                           flags=F_SKIP_ON_REFLECT, name="L3 face 2"))       # lens.f90:595-610
        out.append(Surface(SURF_SPHERE, l3.centre3_z, l3.R3, l3.n3, l3.n1,
                           flags=F_SKIP_ON_REFLECT | F_MISS_IS_HELP3, name="L3 face 3"))  # :616-628
        if s.iris == "after":                                  # lens.f90:632-644
            out.append(Surface(SURF_IRIS, l3.centre3_z + l3.R3, aperture=l3.radius * s.iris_size,
                               name="iris after"))
        out.append(Surface(SURF_IMAGE, self.img_plane + s.fibre_offset, name="image plane"))  # optics_system.f90:48
        # ray-path tracker pushes (stackMod): after bottle%forward (main.f90:147), after each
        # lens of telescope and at the image plane (optics_system.f90:29,39,50)
        last = {}
        for k, sf in enumerate(out):
            group = ("bottle" if sf.name.startswith("bottle") else
                     "L2" if sf.name.startswith("L2") else
                     "image" if sf.kind == SURF_IMAGE else "L3")
            last[group] = k
        for k in last.values():
            out[k].flags |= F_TRACK
        assert len(out) <= MAX_SURFACES
        return out

    def queue_split(self, phase: int) -> int:
        """Where the queued kernel cuts the surface list (index of the first surface of segment
        2): just after the aperture stop that removes most rays — the plano flat face for the
        ring source (69 % of ring rays miss it), the first doublet face for the point
        source (a third of the remaining rays miss it).  Scheduling only; results are identical
        for any value."""
        return self.queue_split_of(self.surfaces(phase), phase)

    @staticmethod
    def queue_split_of(surfaces, phase: int) -> int:
        """queue_split for a list `surfaces(phase)` has already built (pack_system: one list per phase, not two)."""
        names = [s.name for s in surfaces]
        stop = "L2 flat" if phase == 1 else "L3 face 1"
        return names.index(stop) + 1

    @property
    def bin_width(self) -> float:
        return self.settings.image_diameter / 401.0            # imageMod.f90:45

    @property
    def na_angle(self) -> float:
        return math.asin(NA_SINE)                              # imageMod.f90:40

    @property
    def na_cos_min(self) -> float:
        """Smallest double x with acos(x) <= asin(0.22) under this host's libm.

        src/imageMod.f90:39-44 rejects a ray when acos(x) > asin(0.22); acos is
        monotone, so the same decision is `x < na_cos_min`, one compare per ray on
        the device with the host libm's (= the reference's) rounding of acos.
        """
        return acos_threshold(self.na_angle)

    def max_intersections(self, phase: int) -> int:
        return len(self.surfaces(phase))
