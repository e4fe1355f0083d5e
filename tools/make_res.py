#!/usr/bin/env python3
"""Regenerate opticalraytrace_amd/res/*.params from the values in the reference's
res/ directory (run in the build container only; /root/reference is read-only).

The .params format is the drop-in contract (first token per line is the value,
see opticalraytrace_amd/params.py); the numbers are measured lens/bottle data.
Only the values are taken over — they are re-printed as Python float literals —
and the per-line labels are written here.  Files the reference itself cannot read
(13-15 value bottle files) are skipped.
"""
import glob
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from opticalraytrace_amd.params import first_tokens, parse_real  # noqa: E402

REF = "/root/reference/res"
OUT = os.path.join(os.path.dirname(__file__), "..", "opticalraytrace_amd", "res")

PLANO = ["centre thickness [m]", "radius of curvature of the convex face [m]", "lens diameter [m]",
         "effective focal length f [m]", "back focal length fb [m]", "index of the surrounding medium",
         "Sellmeier B1 (Schott N-BK7)", "Sellmeier B2", "Sellmeier B3",
         "Sellmeier C1 [um^2]", "Sellmeier C2 [um^2]", "Sellmeier C3 [um^2]"]
DOUBLET = ["thickness of element 1 [m]", "thickness of element 2 [m]", "radius of face 1 [m]",
           "radius of face 2 (cemented) [m]", "radius of face 3 [m]", "lens diameter [m]",
           "effective focal length f [m]", "back focal length fb [m]", "index of the surrounding medium",
           "element 1 Sellmeier B1 (N-LAK22)", "element 1 B2", "element 1 B3",
           "element 1 C1 [um^2]", "element 1 C2 [um^2]", "element 1 C3 [um^2]",
           "element 2 Sellmeier B1 (N-SF6)", "element 2 B2", "element 2 B3",
           "element 2 C1 [um^2]", "element 2 C2 [um^2]", "element 2 C3 [um^2]"]
BOTTLE = ["wall thickness [m]", "outer radius a, along z [m]", "outer radius b, along y [m] (a != b: elliptical)",
          "centre x [m]", "centre y [m]", "centre z [m]",
          "glass dispersion A (soda-lime, clear)", "glass dispersion B", "glass dispersion C",
          "contents Cauchy A (ethanol)", "contents Cauchy B", "contents Cauchy C",
          "wall absorption mua [1/m]", "wall scattering mus [1/m]",
          "contents absorption mua [1/m]", "contents scattering mus [1/m]"]


def main():
    os.makedirs(OUT, exist_ok=True)
    for path in sorted(glob.glob(os.path.join(REF, "*.params"))):
        name = os.path.basename(path)
        if name == "settings.params":
            continue
        toks = first_tokens(path)
        n = len(toks)
        if name.startswith("clearBottle"):
            if 12 < n < 16:
                print("skip (unreadable by the reference):", name, n)
                continue
            labels = BOTTLE
        elif name.startswith("achromaticDoublet"):
            labels, n = DOUBLET, 21
        else:
            labels, n = PLANO, 12
        with open(os.path.join(OUT, name), "w") as f:
            for tok, lab in zip(toks[:n], labels):
                f.write(f"{repr(parse_real(tok)):<24}! {lab}\n")
        print("wrote", name, n)


if __name__ == "__main__":
    main()
