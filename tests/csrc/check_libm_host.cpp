// check_libm_host.cpp — csrc/ort_libm.h (glibc 2.35's sin / cos / sincos / log / atan2 / acos restated for the
// device) compiled for the HOST and compared with the host's libm, bit for bit, over the arguments the tracer
// forms: sweeps of uniforms (w 2^-32 and 53-bit), angles in [0, 2 pi], cosines in [-1, 1], unit-vector
// components, plus the boundaries of every range the algorithms switch at.
//   check_libm_host [n_per_function = 20000000] [seed]
// prints one line per function: calls, mismatches, first mismatching argument; exit status 1 on any mismatch.
// Build: g++ -O2 -std=c++17 -mfma -ffp-contract=off (tests/test_libm_exact.py).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "../../opticalraytrace_amd/csrc/ort_libm.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t next64()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double u53() { return (double)(next64() >> 11) * 0x1p-53; }
static inline double u32() { return (double)(next64() >> 32) * 0x1p-32; }
static inline bool same(double a, double b)
{
    uint64_t x, y; memcpy(&x, &a, 8); memcpy(&y, &b, 8);
    return x == y || (a != a && b != b);
}

struct Tally { const char *name; long long n = 0, bad = 0; double a0 = 0, a1 = 0, got = 0, want = 0; };
static void report(const Tally &t, int &fail)
{
    printf("%-8s calls %lld mismatches %lld", t.name, t.n, t.bad);
    if (t.bad) { printf("  first: arg %a %a got %a want %a", t.a0, t.a1, t.got, t.want); fail = 1; }
    printf("\n");
}
// the host libm through volatile pointers: with direct calls the compiler merges sin(x) and cos(x) of one argument
// into ONE sincos(x) call — another function with other last bits (ort_libm.h)
static double (*volatile libm_sin)(double) = ::sin;
static double (*volatile libm_cos)(double) = ::cos;
static double (*volatile libm_log)(double) = ::log;
static double (*volatile libm_acos)(double) = ::acos;
static double (*volatile libm_atan2)(double, double) = ::atan2;
static void (*volatile libm_sincos)(double, double *, double *) = ::sincos;
#define CHECK1(T, f, x) do { const double x_ = (x); const double g_ = ort::glibc::f(x_), w_ = libm_##f(x_); T.n++; \
    if (!same(g_, w_)) { if (!T.bad) { T.a0 = x_; T.got = g_; T.want = w_; } T.bad++; } } while (0)

int main(int argc, char **argv)
{
    const long long n = argc > 1 ? atoll(argv[1]) : 20000000ll;
    if (argc > 2) rng_state = strtoull(argv[2], 0, 0);
    const double twopi = 2. * 3.14159265358979323846;
    int fail = 0;
    Tally tsin{"sin"}, tcos{"cos"}, tsc{"sincos"}, tlog{"log"}, tat{"atan2"}, tac{"acos"};
    auto sincos_check = [&](double x) {
        const ort::glibc::SinCos g = ort::glibc::sincos(x);
        double ws, wc;
        libm_sincos(x, &ws, &wc);
        tsc.n++;
        if (!same(g.s, ws) || !same(g.c, wc)) { if (!tsc.bad) { tsc.a0 = x; tsc.got = same(g.s, ws) ? g.c : g.s; tsc.want = same(g.s, ws) ? wc : ws; } tsc.bad++; }
    };
    auto atan2_check = [&](double y, double x) {
        const double g = ort::glibc::atan2(y, x), w = libm_atan2(y, x);
        tat.n++;
        if (!same(g, w)) { if (!tat.bad) { tat.a0 = y; tat.a1 = x; tat.got = g; tat.want = w; } tat.bad++; }
    };
    // range boundaries of sin / cos / sincos, +- a few ulps
    const double edges[] = {0x1p-27, 0x1p-26, 0.126, 0.855469, 2.426265, 1.5707963267948966, 3.141592653589793, 4.71238898038469,
                            6.283185307179586, 0.7853981633974483, 2.356194490192345, 3.9269908169872414, 5.497787143782138, 105414350., 1e-300, 0.};
    for (double e : edges)
        for (int s = -1; s <= 1; s += 2)
            for (int d = -40; d <= 40; ++d) {
                double x = e;
                for (int j = 0; j < (d < 0 ? -d : d); ++j) x = nextafter(x, d < 0 ? -1e300 : 1e300);
                x *= s;
                CHECK1(tsin, sin, x); CHECK1(tcos, cos, x); sincos_check(x);
            }
    for (long long i = 0; i < n; ++i) {
        // angles as the tracer forms them: twopi * u (32-bit uniforms), and anything in [-2 pi, 4 pi]
        const double a = (i & 1) ? twopi * u32() : (i & 2) ? (u53() * 6. - 2.) * 3.14159265358979323846 : twopi * u53();
        CHECK1(tsin, sin, a); CHECK1(tcos, cos, a); sincos_check(a);
        if ((i & 255) == 0) { const double sm = a * 0x1p-20 * u53(); CHECK1(tsin, sin, sm); CHECK1(tcos, cos, sm); sincos_check(sm); }
        if ((i & 1023) == 0) { const double big = a * 1e6 * u53(); CHECK1(tsin, sin, big); CHECK1(tcos, cos, big); sincos_check(big); }
    }
    // log: uniforms in (0, 1) (32- and 53-bit), s = x^2 + y^2 < 1, around 1, tiny, 0
    {
        const double le[] = {1.0, 1.0 - 0x1p-4, 1.0 + 0x1.09p-4, 0.5, 0.25, 0x1p-32, 0x1p-31, 0x1p-64, 2.0, 0x1.6p-1, 0x1.6p0};
        for (double e : le)
            for (int d = -40; d <= 40; ++d) {
                double x = e;
                for (int j = 0; j < (d < 0 ? -d : d); ++j) x = nextafter(x, d < 0 ? 0. : 1e300);
                CHECK1(tlog, log, x);
            }
        CHECK1(tlog, log, 0.0);
        for (long long i = 0; i < n; ++i) {
            const double x = (i & 1) ? u32() : (i & 2) ? u53() : (i & 4) ? 1.0 + (u53() - 0.5) * 0.25 : u53() * u53() * 4.;
            CHECK1(tlog, log, x);
        }
    }
    // acos: [-1, 1], denser near +-1 and 0, every range boundary
    {
        const double ae[] = {0., 0x1p-55, 0.125, 0.25, 0.5, 0.75, 0.921875, 0.953125, 0.96875, 1.0};
        for (double e : ae)
            for (int s = -1; s <= 1; s += 2)
                for (int d = -40; d <= 40; ++d) {
                    double x = e;
                    for (int j = 0; j < (d < 0 ? -d : d); ++j) x = nextafter(x, d < 0 ? -1. : 1.);
                    if (fabs(x) <= 1.) CHECK1(tac, acos, s * x);
                }
        for (long long i = 0; i < n; ++i) {
            double x = u53() * 2. - 1.;
            if ((i & 3) == 1) x = copysign(1. - u53() * u53() * 0.04, x);
            if ((i & 3) == 2) x = ::cos(twopi * u53());
            if ((i & 1023) == 3) x *= 0x1p-30 * u53();
            CHECK1(tac, acos, x);
        }
    }
    // atan2: components of unit vectors (sin t cos p, sin t sin p), arbitrary pairs, ratios around 1/16 and the
    // table nodes, equal magnitudes, zeros, tiny ratios
    {
        const double vals[] = {0., -0., 1., -1., 0.5, 0x1p-60, -0x1p-60, 0x1p-30, 1e-300, 0.0625, 3., -3.};
        for (double y : vals) for (double x : vals) atan2_check(y, x);
        for (long long i = 0; i < n; ++i) {
            double y, x;
            const int mode = (int)(i & 7);
            if (mode < 3) { const double st = sqrt(u53()), p = twopi * u53(); y = st * ::sin(p); x = st * ::cos(p); }
            else if (mode == 3) { y = u53() * 2. - 1.; x = u53() * 2. - 1.; }
            else if (mode == 4) { x = u53() * 2. - 1.; y = x * (0.0625 + (u53() - 0.5) * 1e-3) * ((i & 8) ? 1 : -1); }
            else if (mode == 5) { x = u53() * 2. - 1.; y = (i & 8) ? x : -x; if (i & 16) y = nextafter(y, 0.); }
            else if (mode == 6) { x = (u53() * 2. - 1.); y = x * u53() * ((i & 8) ? 0x1p-20 : 0x1p-58); if (i & 16) { const double t = x; x = y; y = t; } }
            else { const double k = (double)(16 + (next64() % 241)) / 256.; x = u53() * 2. - 1.; y = x * (k + (u53() - 0.5) * 0x1p-8); if (i & 8) { const double t = x; x = y; y = t; } }
            atan2_check(y, x);
        }
    }
    report(tsin, fail); report(tcos, fail); report(tsc, fail); report(tlog, fail); report(tat, fail); report(tac, fail);
    return fail;
}
