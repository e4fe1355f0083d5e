#!/usr/bin/env python3
"""Writes tests/golden/flang_list_directed.json: how the compiler the reference is built with here (AMD flang 22, ROCm 7.2,
-fdefault-real-8) prints REAL(8) values with `write(*,*) x` — the form of every number in the reference's trans-stats.dat
(src/main.f90:168-178).  A throw-away Fortran program is compiled and run in this container; the fixture holds the values
(as hex floats) and the text flang printed.  fstr.list_directed_real is pinned to it (tests/test_host_io.py)."""
import json
import os
import subprocess
import tempfile

FLANG = "/opt/rocm/lib/llvm/bin/flang"
VALUES = [0.0, 1e14, 1e15, 9.4e14, 9.5e14, 9.99999e15, 1e16, 1.23456789e16, 1e17, 0.1, 0.0999999, 0.099, 0.0951, 0.0949, 0.09, 0.05,
          1e-5, 123456.789, 0.5, 0.25, 0.95, 0.999, 99.5, 100.0, 1e3, 12345678901234.5, 123456789012345.6, 1.5e15, 9e15, 2.46, 60.07, 0.03,
          49.25, 49.250400000000006, 3.9899999999999998e-2, 1e-10, -2e-3, -0.5, -12.75, -1e15, 0.075001, 0.0399, 5e-2, 0.123456789012345,
          1.0, 10.0, 7.0, 1234.5678, 1e-300, 1e300, 6.02214076e23, 0.3333333333333333, 2.0 / 3.0, 100.0 / 3.0]


def main():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "ld.f90")
        with open(src, "w") as f:
            f.write("program ld\n    implicit none\n    real(8) :: v\n    integer :: i\n    do i = 1, %d\n        read(*, '(Z16)') v\n"
                    "        write(*,*) v\n    end do\nend program\n" % len(VALUES))
        exe = os.path.join(d, "ld")
        subprocess.run([FLANG, "-O1", "-o", exe, src], check=True, capture_output=True)
        import struct
        inp = "".join("%016X\n" % struct.unpack("<Q", struct.pack("<d", v))[0] for v in VALUES)
        out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(out) == len(VALUES), out
    rec = [{"value": float(v).hex(), "text": t[1:] if t.startswith(" ") else t} for v, t in zip(VALUES, out)]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flang_list_directed.json")
    json.dump({"compiler": "AMD flang 22.0.0git (ROCm 7.2.0)", "records": rec}, open(path, "w"), indent=0)
    print("wrote", path, len(rec))
    for r in rec:
        print(float.fromhex(r["value"]), repr(r["text"]))


if __name__ == "__main__":
    main()
