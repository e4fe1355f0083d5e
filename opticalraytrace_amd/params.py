"""Readers for the reference's text inputs and the lens/bottle models built from them.

Host-side counterpart of (reference file:line)
  * settings file           src/setupMod.f90:57-133   (20 positional lines)
  * plano-convex .params    src/lens.f90:129-167      (12 lines)
  * achromatic doublet      src/lens.f90:73-126       (21 lines)
  * glass bottle            src/lens.f90:170-227      (12 or >=16 lines)
  * dispersion formulas     src/lens.f90:647-695
  * settings writer/defaults runner.py:67-108

The Fortran side uses list-directed reads (`read(u,*) x`): the first blank- or
comma-delimited token of each non-empty record is the value, `d`/`D` exponents
are legal, everything after it is a free-text comment.  All `real` are fp64
(src/Makefile:2), so Python floats reproduce the arithmetic bit for bit
(+ - * / sqrt are correctly rounded on both sides).
"""
from __future__ import annotations

import functools
import math
import os
import re
from dataclasses import dataclass, field, asdict
from typing import List, Optional

_NUM_RE = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eEdD][+-]?\d+)?$")


class ParamsError(ValueError):
    """Bad or truncated input file (the reference dies with a Fortran I/O `error stop`)."""


@functools.lru_cache(maxsize=8192)        # (a sweep parses the same few hundred tokens for every one of its systems)
def parse_real(tok: str) -> float:
    """Fortran list-directed real: `785d-9`, `1.d-2`, `5`, `0.`."""
    t = tok.strip()
    if not _NUM_RE.match(t):
        raise ParamsError(f"not a Fortran real: {tok!r}")
    return float(t.replace("d", "e").replace("D", "e"))


def parse_logical(tok: str) -> bool:
    """Fortran list-directed logical: optional '.', then T/F, rest ignored."""
    t = tok.strip().lstrip(".").lower()
    if t.startswith("t"):
        return True
    if t.startswith("f"):
        return False
    raise ParamsError(f"not a Fortran logical: {tok!r}")


def first_tokens(path: str) -> List[str]:
    """First token of every non-blank record (list-directed reads skip blank records).  A sweep reads the same
    few lens / bottle files for every one of its systems: the tokens are kept per (path, size, mtime)."""
    try:
        st = os.stat(path)
        key = (path, st.st_size, st.st_mtime_ns)
    except OSError:
        key = None
    if key is not None and key in _TOKENS:
        return list(_TOKENS[key])
    toks = _first_tokens(path)
    if key is not None:
        if len(_TOKENS) > 512:
            _TOKENS.clear()
        _TOKENS[key] = tuple(toks)
    return toks


_TOKENS: dict = {}


def _first_tokens(path: str) -> List[str]:
    toks: List[str] = []
    with open(path, "r") as f:
        for line in f:
            s = line.strip()
            if not s:
                continue
            tok = re.split(r"[\s,]+", s, maxsplit=1)[0]
            if tok.startswith("'") or tok.startswith('"'):
                q = tok[0]
                end = s.find(q, 1)
                tok = s[1:end] if end > 0 else tok.strip(q)
            toks.append(tok)
    return toks


# --------------------------------------------------------------------------
# dispersion formulas, src/lens.f90:647-695 (wave in metres, converted to um)
# --------------------------------------------------------------------------
def sellmeier(wave: float, b1, b2, b3, c1, c2, c3) -> float:
    w = wave * 1e6
    wave2 = w * w
    a = (b1 * wave2) / (wave2 - c1)
    b = (b2 * wave2) / (wave2 - c2)
    c = (b3 * wave2) / (wave2 - c3)
    return math.sqrt(1.0 + (a + b + c))


def cauchy(wave: float, a, b, c) -> float:
    w = wave * 1e6
    return a + b * (1.0 / (w * w)) + c * (1.0 / ((w * w) * (w * w)))


def dispersion(wave: float, a, b, c) -> float:
    w = wave * 1e6
    wave2 = w * w
    return a - b * wave2 + (c / wave2)


# --------------------------------------------------------------------------
# lens / bottle models
# --------------------------------------------------------------------------
@dataclass
class PlanoConvex:
    """type plano_convex, src/lens.f90:14-20; constructor :129-167."""
    thickness: float
    curve_radius: float
    diameter: float
    f: float
    fb: float
    n1: float
    n2: float
    radius: float
    centre_z: float

    @classmethod
    def from_file(cls, path: str, wavelength: float, offset: float = 0.0) -> "PlanoConvex":
        t = first_tokens(path)
        if len(t) < 12:
            raise ParamsError(f"{path}: plano-convex file needs 12 values, found {len(t)}")
        v = [parse_real(x) for x in t[:12]]
        thickness, curve_radius, diameter, f, fb, n1 = v[:6]
        n2 = sellmeier(wavelength, *v[6:12])
        return cls(thickness, curve_radius, diameter, f, fb, n1, n2,
                   radius=diameter / 2.0,
                   centre_z=offset + (fb + thickness) - curve_radius)

    @property
    def flat_z(self) -> float:
        """z of the flat face, src/lens.f90:447."""
        return self.centre_z + self.curve_radius - self.thickness


@dataclass
class AchromaticDoublet:
    """type achromatic_doublet, src/lens.f90:27-33; constructor :73-126."""
    thickness1: float
    thickness2: float
    R1: float
    R2: float
    R3: float
    diameter: float
    f: float
    fb: float
    n1: float
    n2: float
    n3: float
    radius: float
    thickness: float
    centre1_z: float
    centre2_z: float
    centre3_z: float

    @classmethod
    def from_file(cls, path: str, wavelength: float, offset: float = 0.0) -> "AchromaticDoublet":
        t = first_tokens(path)
        if len(t) < 21:
            raise ParamsError(f"{path}: doublet file needs 21 values, found {len(t)}")
        v = [parse_real(x) for x in t[:21]]
        th1, th2, R1, R2, R3, diameter, f, fb, n1 = v[:9]
        n2 = sellmeier(wavelength, *v[9:15])
        n3 = sellmeier(wavelength, *v[15:21])
        thickness = th1 + th2
        return cls(th1, th2, R1, R2, R3, diameter, f, fb, n1, n2, n3,
                   radius=diameter / 2.0, thickness=thickness,
                   centre1_z=offset + fb + R1,
                   centre2_z=offset + fb + th1 - R2,
                   centre3_z=offset + fb + thickness - R3)


@dataclass
class GlassBottle:
    """type glass_bottle, src/lens.f90:40-48; constructor :170-227."""
    thickness: float
    radiusa: float
    radiusb: float
    centre: List[float]
    nbottle: float
    ncontents: float
    mua_b: float = 0.0
    mus_b: float = 0.0
    mua_c: float = 0.0
    mus_c: float = 0.0

    @property
    def ellipse(self) -> bool:
        return self.radiusa != self.radiusb

    @property
    def scatters(self) -> bool:
        return (self.mua_b + self.mus_b) != 0.0 or (self.mua_c + self.mus_c) != 0.0

    @classmethod
    def from_file(cls, path: str, wavelength: float) -> "GlassBottle":
        t = first_tokens(path)
        n = len(t)
        if n < 12:
            raise ParamsError(f"{path}: bottle file needs 12 values, found {n}")
        if 12 < n < 16:
            # src/lens.f90:195-208: a 13th value commits the reader to three more
            # reads; the reference aborts with "End of file" (e.g. the shipped
            # clearBottle-small_0.0mm.params has 14 lines).
            raise ParamsError(f"{path}: bottle file has {n} values; the reference accepts "
                              "exactly 12 (no scattering) or >= 16 (mua_b, mus_b, mua_c, mus_c)")
        v = [parse_real(x) for x in t[:12]]
        thickness, ra, rb, x, y, z = v[:6]
        mu = [parse_real(q) for q in t[12:16]] if n >= 16 else [0.0] * 4
        return cls(thickness, ra, rb, [x, y, z],
                   nbottle=dispersion(wavelength, *v[6:9]),
                   ncontents=cauchy(wavelength, *v[9:12]),
                   mua_b=mu[0], mus_b=mu[1], mua_c=mu[2], mus_c=mu[3])


# --------------------------------------------------------------------------
# settings file
# --------------------------------------------------------------------------
SOURCE_TYPES = ("image", "spot", "point", "isors", "crs")
IRIS_POSITIONS = ("before", "after", "none")

# key order and defaults of runner.py:67-86
_SETTINGS_KEYS = ["ring_width", "wavelength", "nphotons", "alpha", "n axicon", "use_bottle",
                  "use_tracker", "make_images", "image_diameter", "fibre_offset", "light_source",
                  "iris", "iris_size", "bottle_file", "L2_file", "L3_file", "image_source",
                  "data_folder", "isors_offset", "crs_spot_size"]


@dataclass
class Settings:
    """The 20 positional values of src/setupMod.f90:57-133 (defaults: runner.py:67-86)."""
    ring_width: float = 0.5e-3
    wavelength: float = 785e-9
    nphotons: int = 1000000000
    alpha: float = 5.0               # degrees in the file; radians after setupMod.f90:61
    n_axicon: float = 1.45
    use_bottle: bool = True
    use_tracker: bool = False
    make_images: bool = False
    image_diameter: float = 1e-2
    fibre_offset: float = 0.0
    light_source: str = "point"
    iris: str = "none"
    iris_size: float = 1.0
    bottle_file: str = "clearBottle-large.params"
    L2_file: str = "planoConvex-f39.9mm.params"
    L3_file: str = "achromaticDoublet-f50.0mm.params"
    image_source: str = "bessel-smear.dat"
    data_folder: str = "settings"
    isors_offset: float = 0.0
    crs_spot_size: float = 0.0

    @classmethod
    def from_file(cls, path: str) -> "Settings":
        t = first_tokens(path)
        if len(t) < 20:
            raise ParamsError(f"{path}: settings file needs 20 values, found {len(t)}")
        s = cls(
            ring_width=parse_real(t[0]), wavelength=parse_real(t[1]),
            nphotons=int(parse_real(t[2])), alpha=parse_real(t[3]), n_axicon=parse_real(t[4]),
            use_bottle=parse_logical(t[5]), use_tracker=parse_logical(t[6]),
            make_images=parse_logical(t[7]), image_diameter=parse_real(t[8]),
            fibre_offset=parse_real(t[9]), light_source=t[10], iris=t[11],
            iris_size=parse_real(t[12]), bottle_file=t[13], L2_file=t[14], L3_file=t[15],
            image_source=t[16], data_folder=t[17], isors_offset=parse_real(t[18]),
            crs_spot_size=parse_real(t[19]))
        s.validate()
        return s

    def validate(self) -> None:
        if self.light_source not in SOURCE_TYPES:
            raise ParamsError("No such source type!")          # setupMod.f90:98
        if self.iris not in IRIS_POSITIONS:
            raise ParamsError("No such iris position!")        # setupMod.f90:110
        if self.nphotons > 10000 and self.use_tracker:
            raise ParamsError("Too many photons for tracker use!")  # setupMod.f90:75
        if not (0 <= self.nphotons <= 2147483647):
            raise ParamsError("nphotons must fit a default INTEGER (setupMod.f90:8)")

    def write(self, path: str) -> None:
        """Settings text in the layout of runner.py:99-108 (value padded to col 35, `# key`)."""
        vals = [self.ring_width, _e2d(self.wavelength), self.nphotons, self.alpha, self.n_axicon,
                str(self.use_bottle).lower(), str(self.use_tracker).lower(),
                str(self.make_images).lower(), _e2d(self.image_diameter), self.fibre_offset,
                self.light_source, self.iris.lower(), self.iris_size, self.bottle_file,
                self.L2_file, self.L3_file, self.image_source, self.data_folder,
                self.isors_offset, self.crs_spot_size]
        with open(path, "w") as f:
            for key, val in zip(_SETTINGS_KEYS, vals):
                sv = str(val)
                f.write(sv + " " * (35 - len(sv)) + "# " + key + "\n")


def _e2d(x: float) -> str:
    return repr(float(x)).replace("e", "d")


def resource_dir() -> str:
    """Directory of the .params data files shipped with this package."""
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "res")
