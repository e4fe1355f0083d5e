// libm_host_api.cpp — csrc/ort_libm.h compiled for the host behind a C entry, for tests/test_libm_exact.py:
// evaluates one of the restated glibc functions (straight or predicated form) over arrays.
#include "../../opticalraytrace_amd/csrc/ort_libm.h"

extern "C" int ortlm_eval(int fn, const double *a, const double *b, double *out, double *out2, long n)
{
    namespace g = ort::glibc;
    using g::SinCos;
    for (long i = 0; i < n; ++i) {
        switch (fn) {
        case 0: out[i] = g::sin(a[i]); break;
        case 1: out[i] = g::cos(a[i]); break;
        case 2: { const SinCos r = g::sincos(a[i]); out[i] = r.s; out2[i] = r.c; break; }
        case 3: out[i] = g::log(a[i]); break;
        case 4: out[i] = g::atan2(a[i], b[i]); break;
        case 5: out[i] = g::acos(a[i]); break;
        case 6: { const SinCos r = g::sincos_p<false>(a[i]); out[i] = r.s; out2[i] = r.c; break; }
        case 7: { const SinCos r = g::sincos_p<true>(a[i]); out[i] = r.s; out2[i] = r.c; break; }
        case 8: out[i] = g::log_p(a[i]); break;
        case 9: out[i] = g::atan2_p(a[i], b[i]); break;
        case 10: out[i] = g::acos_p(a[i]); break;
        default: return 1;
        }
    }
    return 0;
}
