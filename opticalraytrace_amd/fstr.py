"""Number -> string rules of the reference's `str()` (src/utils.f90:47-56, :252-420),
needed to reproduce the output file name of src/main.f90:45-48."""
from typing import Iterable, Optional


def str_real(x: float, length: Optional[int] = None) -> str:
    """str_R8 (src/utils.f90:~372): write '(f100.16)', adjustl, trim, keep the first `length` chars."""
    s = f"{x:.16f}"
    if length is not None:
        s = s[:length].strip()
    return s


def str_int(i: int, length: Optional[int] = None) -> str:
    """str_I32 / str_I64: zero-padded on the left to `length`, or truncated to it."""
    s = str(int(i))
    if length is None:
        return s
    if length >= len(s):
        return "0" * (length - len(s)) + s
    return s[:length].strip()


def str_logical(a: bool) -> str:
    """str_logical: format L1."""
    return "T" if a else "F"


def str_logical_array(a: Iterable[bool]) -> str:
    """str_logicalarray: '_' + L1 for every element, e.g. [F, F] -> '_F_F'."""
    return "".join("_" + str_logical(v) for v in a)
