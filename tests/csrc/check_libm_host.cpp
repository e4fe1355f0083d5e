// check_libm_host.cpp — csrc/ort_libm.h (glibc 2.35's sin / cos / sincos / log / atan2 / acos restated for the
// device) compiled for the HOST and compared with the host's libm, bit for bit (arguments: libm_args.h).
//   check_libm_host [n_per_function = 20000000] [seed]
// prints one line per function: calls, mismatches, first mismatching argument; exit status 1 on any mismatch.
// Build: g++ -O2 -std=c++17 -mfma -ffp-contract=off (tests/test_libm_exact.py).
#include <stdio.h>
#include <stdlib.h>
#include "../../opticalraytrace_amd/csrc/ort_libm.h"
#include "libm_args.h"

using namespace libm_args;

int main(int argc, char **argv)
{
    const long long n = argc > 1 ? atoll(argv[1]) : 20000000ll;
    Rng rng(argc > 2 ? strtoull(argv[2], 0, 0) : 0x9E3779B97F4A7C15ull);
    Tally tsin{"sin"}, tcos{"cos"}, tsc{"sincos"}, tlog{"log"}, tat{"atan2"}, tac{"acos"};
    Tally psc{"sincos_p"}, ppair{"sin,cos_p"}, plog{"log_p"}, pat{"atan2_p"}, pac{"acos_p"};      // the predicated forms
    const long long chunk = 1 << 22;
    for (long long done = 0; done < n; done += chunk) {
        const long long m = n - done < chunk ? n - done : chunk;
        for (double x : angles(m, rng)) {
            tsin.add(ort::glibc::sin(x), libm_sin(x), x);
            tcos.add(ort::glibc::cos(x), libm_cos(x), x);
            const ort::glibc::SinCos g = ort::glibc::sincos(x);
            double ws, wc;
            libm_sincos(x, &ws, &wc);
            tsc.add(g.s, ws, x); tsc.add(g.c, wc, x);
            const ort::glibc::SinCos gp = ort::glibc::sincos_p<false>(x), gq = ort::glibc::sincos_p<true>(x);
            psc.add(gp.s, ws, x); psc.add(gp.c, wc, x);
            ppair.add(gq.s, libm_sin(x), x); ppair.add(gq.c, libm_cos(x), x);
        }
        for (double x : logs(m, rng)) { tlog.add(ort::glibc::log(x), libm_log(x), x); plog.add(ort::glibc::log_p(x), libm_log(x), x); }
        for (double x : acoss(m, rng)) { tac.add(ort::glibc::acos(x), libm_acos(x), x); pac.add(ort::glibc::acos_p(x), libm_acos(x), x); }
        std::vector<double> ys, xs;
        atan2s(m, rng, ys, xs);
        for (size_t i = 0; i < ys.size(); ++i) {
            tat.add(ort::glibc::atan2(ys[i], xs[i]), libm_atan2(ys[i], xs[i]), ys[i], xs[i]);
            pat.add(ort::glibc::atan2_p(ys[i], xs[i]), libm_atan2(ys[i], xs[i]), ys[i], xs[i]);
        }
    }
    return tsin.report() | tcos.report() | tsc.report() | tlog.report() | tat.report() | tac.report() |
           psc.report() | ppair.report() | plog.report() | pat.report() | pac.report();
}
