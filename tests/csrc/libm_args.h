// libm_args.h — the arguments tests/csrc/check_libm_host.cpp and check_libm_gpu.hip evaluate csrc/ort_libm.h on:
// what the tracer forms (angles twopi * u of 32- and 53-bit uniforms, uniforms in (0, 1), cosines, components of
// unit vectors) plus +-40 ulps around every boundary at which glibc's algorithms switch ranges.  Host code only.
#pragma once
#include <stdint.h>
#include <math.h>
#include <vector>

namespace libm_args {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double u53() { return (double)(next() >> 11) * 0x1p-53; }
    double u32() { return (double)(next() >> 32) * 0x1p-32; }
};

static const double kTwoPi = 2. * 3.14159265358979323846;

inline void around(std::vector<double> &out, double e, double lo, double hi, bool both_signs)
{
    for (int d = -40; d <= 40; ++d) {
        double x = e;
        for (int j = 0; j < (d < 0 ? -d : d); ++j) x = nextafter(x, d < 0 ? lo : hi);
        out.push_back(x);
        if (both_signs) out.push_back(-x);
    }
}

// angles for sin / cos / sincos
inline std::vector<double> angles(long long n, Rng &r)
{
    std::vector<double> a;
    const double edges[] = {0x1p-27, 0x1p-26, 0.126, 0.855469, 2.426265, 1.5707963267948966, 3.141592653589793, 4.71238898038469,
                            6.283185307179586, 0.7853981633974483, 2.356194490192345, 3.9269908169872414, 5.497787143782138, 105414335., 1e-300, 0.};     // (high word 0x419921FB = 105414336: where glibc switches to its Payne-Hanek reduction, the fallback here)
    for (double e : edges) around(a, e, -1e300, 1e300, true);
    for (long long i = 0; i < n; ++i) {
        const double x = (i & 1) ? kTwoPi * r.u32() : (i & 2) ? (r.u53() * 6. - 2.) * 3.14159265358979323846 : kTwoPi * r.u53();
        a.push_back(x);
        if ((i & 255) == 0) a.push_back(x * 0x1p-20 * r.u53());
        if ((i & 1023) == 0) a.push_back(x * 1e6 * r.u53());
    }
    return a;
}

// log: uniforms in (0, 1), s = x^2 + y^2 < 1, around 1, tiny, 0
inline std::vector<double> logs(long long n, Rng &r)
{
    std::vector<double> a;
    const double le[] = {1.0, 1.0 - 0x1p-4, 1.0 + 0x1.09p-4, 0.5, 0.25, 0x1p-32, 0x1p-31, 0x1p-64, 2.0, 0x1.6p-1, 0x1.6p0};
    for (double e : le) around(a, e, 0., 1e300, false);
    a.push_back(0.0);
    for (long long i = 0; i < n; ++i)
        a.push_back((i & 1) ? r.u32() : (i & 2) ? r.u53() : (i & 4) ? 1.0 + (r.u53() - 0.5) * 0.25 : r.u53() * r.u53() * 4.);
    return a;
}

// acos: [-1, 1], denser near +-1 and 0, every range boundary
inline std::vector<double> acoss(long long n, Rng &r)
{
    std::vector<double> a, t;
    const double ae[] = {0., 0x1p-55, 0.125, 0.25, 0.5, 0.75, 0.921875, 0.953125, 0.96875, 1.0};
    for (double e : ae) around(t, e, -1., 1., true);
    for (double x : t) if (fabs(x) <= 1.) a.push_back(x);
    for (long long i = 0; i < n; ++i) {
        double x = r.u53() * 2. - 1.;
        if ((i & 3) == 1) x = copysign(1. - r.u53() * r.u53() * 0.04, x);
        if ((i & 3) == 2) x = ::cos(kTwoPi * r.u53());
        if ((i & 1023) == 3) x *= 0x1p-30 * r.u53();
        a.push_back(x);
    }
    return a;
}

// atan2 (y, x): components of unit vectors, arbitrary pairs, ratios around 1/16 and the table nodes, equal
// magnitudes, zeros, tiny ratios
inline void atan2s(long long n, Rng &r, std::vector<double> &ys, std::vector<double> &xs)
{
    const double vals[] = {0., -0., 1., -1., 0.5, 0x1p-60, -0x1p-60, 0x1p-30, 1e-300, 0.0625, 3., -3.};
    for (double y : vals) for (double x : vals) { ys.push_back(y); xs.push_back(x); }
    for (long long i = 0; i < n; ++i) {
        double y, x;
        const int mode = (int)(i & 7);
        if (mode < 3) { const double st = sqrt(r.u53()), p = kTwoPi * r.u53(); y = st * ::sin(p); x = st * ::cos(p); }
        else if (mode == 3) { y = r.u53() * 2. - 1.; x = r.u53() * 2. - 1.; }
        else if (mode == 4) { x = r.u53() * 2. - 1.; y = x * (0.0625 + (r.u53() - 0.5) * 1e-3) * ((i & 8) ? 1 : -1); }
        else if (mode == 5) { x = r.u53() * 2. - 1.; y = (i & 8) ? x : -x; if (i & 16) y = nextafter(y, 0.); }
        else if (mode == 6) { x = (r.u53() * 2. - 1.); y = x * r.u53() * ((i & 8) ? 0x1p-20 : 0x1p-58); if (i & 16) { const double t = x; x = y; y = t; } }
        else { const double k = (double)(16 + (r.next() % 241)) / 256.; x = r.u53() * 2. - 1.; y = x * (k + (r.u53() - 0.5) * 0x1p-8); if (i & 8) { const double t = x; x = y; y = t; } }
        ys.push_back(y); xs.push_back(x);
    }
}

// the host's libm through volatile pointers: with direct calls the compiler merges sin(x) and cos(x) of one
// argument into ONE sincos(x) call — another function with other last bits (ort_libm.h)
static double (*volatile libm_sin)(double) = ::sin;
static double (*volatile libm_cos)(double) = ::cos;
static double (*volatile libm_log)(double) = ::log;
static double (*volatile libm_acos)(double) = ::acos;
static double (*volatile libm_atan2)(double, double) = ::atan2;
static void (*volatile libm_sincos)(double, double *, double *) = ::sincos;

inline bool same(double a, double b)
{
    uint64_t x, y;
    __builtin_memcpy(&x, &a, 8); __builtin_memcpy(&y, &b, 8);
    return x == y || (a != a && b != b);
}

struct Tally {
    const char *name;
    long long n = 0, bad = 0;
    double a0 = 0, a1 = 0, got = 0, want = 0;
    void add(double got_, double want_, double arg0, double arg1 = 0.)
    {
        n++;
        if (!same(got_, want_)) { if (!bad) { a0 = arg0; a1 = arg1; got = got_; want = want_; } bad++; }
    }
    int report() const
    {
        printf("%-8s calls %lld mismatches %lld", name, n, bad);
        if (bad) printf("  first: arg %a %a got %a want %a", a0, a1, got, want);
        printf("\n");
        return bad ? 1 : 0;
    }
};

}  // namespace libm_args
