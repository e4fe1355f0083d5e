// TEST INFRASTRUCTURE — never shipped, never measured as the product.
// pinned_libm.cpp — the six libm entries the trace path calls (sin, cos, sincos, log, atan2, acos), taken NOT from the
// machine's libm but from csrc/ort_libm.h compiled for the host: glibc 2.35's algorithms (x86-64, FMA variants) restated
// operation for operation and pinned to 54 000 committed known answers (tests/golden/libm_glibc235.npz,
// tests/test_libm_exact.py).  libort_oracle_pinned.so = ort_oracle.c built with -DORC_PINNED_LIBM + this file: the checker
// for a machine whose own libm is not glibc 2.35 — there the device (which reproduces 2.35 wherever it runs) and an
// oracle on the host's libm would differ in the last bit of these calls for the host's reason (oracle/binding.py chooses).
#include "../opticalraytrace_amd/csrc/ort_libm.h"

extern "C" {
double ortp_sin(double x) { return ort::glibc::sin(x); }
double ortp_cos(double x) { return ort::glibc::cos(x); }
void ortp_sincos(double x, double *s, double *c)
{
    const ort::glibc::SinCos r = ort::glibc::sincos(x);
    *s = r.s; *c = r.c;
}
double ortp_log(double x) { return ort::glibc::log(x); }
double ortp_atan2(double y, double x) { return ort::glibc::atan2(y, x); }
double ortp_acos(double x) { return ort::glibc::acos(x); }
}
