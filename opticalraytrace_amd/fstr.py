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


# ---------------------------------------------------------------------------
# list-directed output (`write(u,*) ...`) as the compiler the reference is built with here (AMD flang
# 22, ROCm 7.2) lays it out — what the stats row of src/main.f90:168-178 looks like in a file the
# unmodified program wrote.  Pinned by two kinds of fixture: the reference program's own files
# (tests/golden/refprog_*.npz) and tests/golden/flang_list_directed.json — 54 values printed by a flang-built
# program in this container (make_list_directed_golden.py), covering every rule below and its edges:
#   * a REAL(8) is its shortest round-trip digits: `0.` for zero; F form WITHOUT a leading zero (`49.250400000000006`,
#     `100.`, `.5`, `-.25`) when the value ROUNDED TO ONE SIGNIFICANT DIGIT lies in [0.1, 1e15) — so `.0951` and
#     `940000000000000.` but `9.49E-02` and `9.5E+14` —, else `d.dddE+-XX` (`3.99E-02`, `5.E-02`, `-2.E-03`, `1.E-300`);
#   * numeric and logical items are preceded by one blank; a character item only when the item before it
#     was not a character item;
#   * a record holds at most 79 characters: an item that does not fit starts a new record — a character
#     item then without its blank —, and a character item longer than a record is cut at 79 with the
#     rest on the next record behind a blank.
# (gfortran, the reference Makefile's compiler, pads differently and is not available here to pin.)
# ---------------------------------------------------------------------------
LIST_DIRECTED_WIDTH = 79


def list_directed_real(x: float) -> str:
    if x == 0.0:
        return "0."
    r = repr(abs(x))                            # the shortest digits that round-trip
    if "e" in r or "E" in r:
        mant, ex = r.lower().split("e")
        ex = int(ex)
    else:
        mant, ex = r, 0
    ip, _, fp = mant.partition(".")
    alld = (ip + fp).lstrip("0")
    # decimal exponent of the first significant digit
    if ip.strip("0"):
        e10 = len(ip.lstrip("0")) - 1 + ex
    else:
        e10 = -(len(fp) - len(fp.lstrip("0")) + 1) + ex
    alld = alld.rstrip("0") or "0"
    sign = "-" if x < 0 else ""
    # flang decides between the two forms on the value rounded to ONE significant digit: 0.0951 -> 0.1 (F), 9.5e14 -> 1e15 (E)
    from decimal import Decimal, ROUND_HALF_EVEN
    one = Decimal(abs(x)).scaleb(-e10).quantize(Decimal(1), rounding=ROUND_HALF_EVEN)
    e1 = e10 + (1 if one >= 10 else 0)
    if -1 <= e1 < 15:
        if e10 >= 0:
            body = alld[:e10 + 1].ljust(e10 + 1, "0") + "." + alld[e10 + 1:]
        else:                                   # |x| < 1: no leading zero
            body = "." + "0" * (-e10 - 1) + alld
        return sign + body
    return f"{sign}{alld[0]}.{alld[1:]}E{'+' if e10 >= 0 else '-'}{abs(e10):02d}"


def list_directed_record(items) -> str:
    """items: floats, bools (-> T / F) and strs, in write order -> the text incl. newlines."""
    lines, line, prev_char = [], "", False
    for it in items:
        is_char = isinstance(it, str)
        text = it if is_char else ("T" if it else "F") if isinstance(it, bool) else list_directed_real(float(it))
        sep = "" if (is_char and prev_char and line) else " "
        if is_char and not line and lines:
            sep = ""
        if line and len(line) + len(sep) + len(text) > LIST_DIRECTED_WIDTH:
            lines.append(line)
            line = ""
            sep = "" if is_char else " "
        piece = sep + text
        while len(line) + len(piece) > LIST_DIRECTED_WIDTH:          # one character item longer than a record
            room = LIST_DIRECTED_WIDTH - len(line)
            lines.append(line + piece[:room])
            line, piece = "", " " + piece[room:]
        line += piece
        prev_char = is_char
    lines.append(line)
    return "\n".join(lines) + "\n"
