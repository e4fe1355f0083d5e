// SPDX-License-Identifier: LGPL-2.1-or-later
// Derived from the GNU C Library 2.35 (sysdeps/ieee754/dbl-64: s_sin.c, s_sincos.c, e_log.c, e_atan2.c, e_asin.c and their
// tables — IBM Accurate Mathematical Library, Copyright (C) 2001-2022 Free Software Foundation, Inc.), restated for HIP.
// This file is free software under the GNU Lesser General Public License, version 2.1 or any later version: see
// THIRD_PARTY_NOTICES.md and LICENSES/LGPL-2.1.txt at the repository root.
//
// ort_libm.h — the reference's transcendental functions, bit for bit.
//
// The reference (Fortran, all `real` fp64) calls sin / cos / log / atan2 / acos of the platform's libm:
// built with flang or gfortran on x86-64 Linux these resolve to glibc.  glibc's double-precision functions
// are NOT correctly rounded (documented errors 0.50 - 0.56 ulp), so "the reference's result" is whatever
// glibc's own algorithm produces, and neither a correctly rounded function nor another good libm (the ROCm
// device library: <= 1 - 2 ulp) reproduces it: the in-bottle scattering walk (src/stokes.f90:7-166,
// src/surfaces.f90:13-50) amplifies a last-bit difference in atan2 / sin / cos of the azimuth to 1e-10 in
// 2e-5 of its rays.  This file therefore RESTATES glibc's algorithms, operation for operation, so that the
// device returns glibc's bits.
//
// Third-party dependency restated: GNU C Library 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.11), x86-64, the variants
// its ifunc resolvers select on a CPU with FMA + AVX2 (every x86-64 server CPU since 2013/2015; the test
// session checks the host's libm against known answers and says so if it is another one):
//   sin, cos         sysdeps/ieee754/dbl-64/s_sin.c      built with -mfma -mavx2  (__sin_fma, __cos_fma)
//   sincos           sysdeps/ieee754/dbl-64/s_sincos.c   NO multiarch variant in 2.35: built without FMA
//   log              sysdeps/ieee754/dbl-64/e_log.c      (__ieee754_log_fma)
//   atan2            sysdeps/ieee754/dbl-64/e_atan2.c    (__ieee754_atan2_fma)
//   acos             sysdeps/ieee754/dbl-64/e_asin.c     (__ieee754_acos_fma)
// Which products the compiler fused into FMAs decides the last bit, so the operation sequences below were
// taken from the machine code of /lib/x86_64-linux-gnu/libm-2.35.a (objdump of s_sin-fma.o, e_log-fma.o,
// e_atan2-fma.o, e_asin-fma.o, s_sincos.o), the tables from the same archive (tools/make_libm_tables.py ->
// ort_libm_tables.h).  Every fused operation is an explicit __builtin_fma; everything else relies on
// -ffp-contract=off (the build's setting).  tests/csrc/check_libm_host.cpp compiles THIS header for the host
// and compares every function with the host's libm over 1e8+ arguments (bit for bit);
// tests/csrc/check_libm_gpu.hip does the same for the device code.
//
// Domain: what the tracer passes — finite arguments, |x| < 1.05e8 for sin / cos / sincos, x >= 0 for log,
// |x| <= 1 for acos, finite atan2 operands.  Anything else (NaN, infinities, huge angles, subnormals where
// glibc takes a special path) goes to ORT_LIBM_FALLBACK, the platform's own function: never reached by a ray.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)              // the product: device functions, tables in constant memory
#include <hip/hip_runtime.h>
#define ORT_LIBM_FN __device__ inline
#define ORT_LIBM_MFN __device__ inline
#define ORT_LIBM_TABLE __device__ __constant__ const
#else                                // tests/csrc/check_libm_host.cpp: the same text compiled for the host
#include <math.h>
#define ORT_LIBM_FN static inline
#define ORT_LIBM_MFN inline
#define ORT_LIBM_TABLE static const
#endif
#include "ort_libm_tables.h"

namespace ort {
namespace glibc {

ORT_LIBM_FN uint64_t bits(double x) { return __builtin_bit_cast(uint64_t, x); }
ORT_LIBM_FN double dbl(uint64_t b) { return __builtin_bit_cast(double, b); }
ORT_LIBM_FN double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
ORT_LIBM_FN double abs_(double x) { return dbl(bits(x) & 0x7fffffffffffffffull); }
ORT_LIBM_FN double copysign_(double x, double s) { return dbl((bits(x) & 0x7fffffffffffffffull) | (bits(s) & 0x8000000000000000ull)); }
ORT_LIBM_FN int32_t hi_word(double x) { return (int32_t)(bits(x) >> 32); }
ORT_LIBM_FN uint32_t lo_word(double x) { return (uint32_t)bits(x); }
#define ORT_LM(b) (::ort::glibc::dbl(b##ull))

// Where a function reads the lookup tables.  TabGlobal: where they are, constant memory — a per-lane gather through
// the vector cache: one wavefront-wide load of 64 scattered rows costs the cache ~50 cycles, and atan2 (7 loads) and
// acos (12) are bound by that, not by their arithmetic (tools/ubench_libm.hip: 2.4 x their instruction count).
// TabLds (device): a workgroup's copy of the four big tables in LDS (stage_tables), read with ds_read_b64.
struct TabGlobal {
    ORT_LIBM_MFN double sincos(int i) const { return dbl(kGlibcSinCosTab[i]); }
    ORT_LIBM_MFN double logd(int i) const { return dbl(kGlibcLogData[i]); }
    ORT_LIBM_MFN double cij(int i) const { return dbl(kGlibcAtanCij[i]); }
    ORT_LIBM_MFN double powtwo(int i) const { return dbl(kGlibcAcosPowtwo[i]); }
    ORT_LIBM_MFN double inroot(int i) const { return dbl(kGlibcAcosInroot[i]); }
    ORT_LIBM_MFN double asncs(int i) const { return dbl(kGlibcAcosAsncs[i]); }
};
constexpr int kLdsSinCos = 0, kLdsCij = kLdsSinCos + 440, kLdsPowtwo = kLdsCij + 241 * 7, kLdsInroot = kLdsPowtwo + 28,
              kLdsAsncs = kLdsInroot + 128, kLdsTableWords = kLdsAsncs + 2568;           // 4851 doubles = 38 808 bytes
#if defined(__HIPCC__)
struct TabLds {
    const __attribute__((address_space(3))) uint64_t *base;      // kLdsTableWords words, filled by stage_tables
    ORT_LIBM_MFN double sincos(int i) const { return dbl(base[kLdsSinCos + i]); }
    ORT_LIBM_MFN double logd(int i) const { return dbl(kGlibcLogData[i]); }      // 128 rows of 16 bytes: a few cache lines, stays global
    ORT_LIBM_MFN double cij(int i) const { return dbl(base[kLdsCij + i]); }
    ORT_LIBM_MFN double powtwo(int i) const { return dbl(base[kLdsPowtwo + i]); }
    ORT_LIBM_MFN double inroot(int i) const { return dbl(base[kLdsInroot + i]); }
    ORT_LIBM_MFN double asncs(int i) const { return dbl(base[kLdsAsncs + i]); }
};
// cooperative copy into a __shared__ uint64_t[kLdsTableWords]; the caller synchronises the workgroup afterwards
ORT_LIBM_FN void stage_tables(uint64_t *lds, int tid, int nthreads)
{
    for (int i = tid; i < 440; i += nthreads) lds[kLdsSinCos + i] = kGlibcSinCosTab[i];
    for (int i = tid; i < 241 * 7; i += nthreads) lds[kLdsCij + i] = kGlibcAtanCij[i];
    for (int i = tid; i < 28; i += nthreads) lds[kLdsPowtwo + i] = kGlibcAcosPowtwo[i];
    for (int i = tid; i < 128; i += nthreads) lds[kLdsInroot + i] = kGlibcAcosInroot[i];
    for (int i = tid; i < 2568; i += nthreads) lds[kLdsAsncs + i] = kGlibcAcosAsncs[i];
}
#endif

struct SinCos { double s, c; };

// The scalar constants of all five algorithms in ONE table in constant memory.  A function fetches its base
// through kbase() — on the device an opaque (asm volatile) copy of the address, so that the loads are scalar
// loads (uniform address: s_load into SGPRs, which feed the vector instructions as operands) that STAY inside
// the function: as 64-bit literals the compiler parks every constant in a VGPR pair for the whole kernel (50
// constants = 100 VGPRs: 160 bytes of scratch per lane in the first port), and plain constant loads it hoists out
// of the kernel's stage loop into SGPRs it then spills.
enum {
    K_sc_BIG, K_sc_SN3, K_sc_SN5, K_sc_CS2, K_sc_CS4, K_sc_CS6, K_sc_S1, K_sc_S2, K_sc_S3, K_sc_S4, K_sc_S5, K_sc_HP0, K_sc_HP1, K_sc_HPINV, K_sc_TOINT, K_sc_MP1, K_sc_MP2, K_sc_PP3, K_sc_PP4, K_sc_T126, K_at_D3, K_at_D5, K_at_D7, K_at_D9, K_at_D11, K_at_D13, K_at_HPI, K_at_HPI1, K_at_OPI, K_at_OPI1, K_ac_F1, K_ac_F2, K_ac_F3, K_ac_F4, K_ac_F5, K_ac_F6, K_ac_RT0, K_ac_RT1, K_ac_RT2, K_ac_RT3, K_ac_HP0, K_ac_HP1, K_ac_PI, K_lg_TWO27, K_lg_MTWO27, K_at_TWO8, K_at_TWO52, K_at_INV16, K_ac_ONEHALF, K_COUNT
};
ORT_LIBM_TABLE uint64_t kGlibcK[K_COUNT] = {
    0x42c8000000000000ull,   // sc::BIG
    0xbfc5555555555515ull,   // sc::SN3
    0x3f811110e829872full,   // sc::SN5
    0x3fe0000000000000ull,   // sc::CS2
    0xbfa5555555555535ull,   // sc::CS4
    0x3f56c16bedd9e239ull,   // sc::CS6
    0xbfc5555555555555ull,   // sc::S1
    0x3f81111111110eceull,   // sc::S2
    0xbf2a01a019db08b8ull,   // sc::S3
    0x3ec71de27b9a7ed9ull,   // sc::S4
    0xbe5addffc2fcdf59ull,   // sc::S5
    0x3ff921fb54442d18ull,   // sc::HP0
    0x3c91a62633145c07ull,   // sc::HP1
    0x3fe45f306dc9c883ull,   // sc::HPINV
    0x4338000000000000ull,   // sc::TOINT
    0x3ff921fb58000000ull,   // sc::MP1
    0xbe4dde973c000000ull,   // sc::MP2
    0xbc8cb3b398000000ull,   // sc::PP3
    0xbacd747f23e32ed7ull,   // sc::PP4
    0x3fc020c49ba5e354ull,   // sc::T126
    0xbfd5555555555555ull,   // at::D3
    0x3fc99999999997fdull,   // at::D5
    0xbfc24924923f7603ull,   // at::D7
    0x3fbc71c6e5129a3bull,   // at::D9
    0xbfb7458022b13c25ull,   // at::D11
    0x3fb375f08b31cbceull,   // at::D13
    0x3ff921fb54442d18ull,   // at::HPI
    0x3c91a62633145c07ull,   // at::HPI1
    0x400921fb54442d18ull,   // at::OPI
    0x3ca1a62633145c07ull,   // at::OPI1
    0x3fc55555555554f9ull,   // ac::F1
    0x3fb333333336127dull,   // ac::F2
    0x3fa6db6dae42c0e4ull,   // ac::F3
    0x3f9f1c7e04f4ad99ull,   // ac::F4
    0x3f96e442c822d419ull,   // ac::F5
    0x3f9292d80f453c72ull,   // ac::F6
    0x3fefffffffecc1ddull,   // ac::RT0
    0x3fdfffffff757304ull,   // ac::RT1
    0x3fd800496769c91aull,   // ac::RT2
    0x3fd4006318d1dab9ull,   // ac::RT3
    0x3ff921fb54442d18ull,   // ac::HP0
    0x3c91a62633145c07ull,   // ac::HP1
    0x400921fb54442d18ull,   // ac::PI
    0x41a0000000000000ull,   // lg::TWO27
    0xc1a0000000000000ull,   // lg::MTWO27
    0x4070000000000000ull,   // at::TWO8
    0x4330000000000000ull,   // at::TWO52
    0x3fb0000000000000ull,   // at::INV16
    0x3ff8000000000000ull,   // ac::ONEHALF
};
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) uint64_t *kptr;      // constant address space: scalar loads
ORT_LIBM_FN kptr kbase()
{
    kptr p = (kptr)kGlibcK;
    asm volatile("" : "+s"(p));
    return p;
}
#else
typedef const uint64_t *kptr;
ORT_LIBM_FN kptr kbase() { return kGlibcK; }
#endif
#define ORT_SCK(n) (::ort::glibc::dbl(K[::ort::glibc::K_sc_##n]))
#define ORT_ATK(n) (::ort::glibc::dbl(K[::ort::glibc::K_at_##n]))
#define ORT_ACK(n) (::ort::glibc::dbl(K[::ort::glibc::K_ac_##n]))
#define ORT_LGK(n) (::ort::glibc::dbl(K[::ort::glibc::K_lg_##n]))


// ---------------------------------------------------------------------------------------------------------
// sin / cos / sincos: s_sin.c.  x is cut into X = a multiple of 1/128 (table: sin X, cos X as double-double
// sn + ssn, cs + ccs) and a remainder |x - X| <= 1/256 with short Taylor polynomials; |x| < 0.126 uses a
// Taylor polynomial alone; 0.855 <= |x| < 2.43 goes through pi/2 - |x|; beyond that a Cody-Waite reduction
// by pi/2 in four pieces (mp1, mp2, pp3, pp4).
// ---------------------------------------------------------------------------------------------------------


// SINCOS_TABLE_LOOKUP: u = big + |x|; its low word counts 1/128ths
template <class TB> ORT_LIBM_FN void sincos_lookup(const TB &T, double u, double &sn, double &ssn, double &cs, double &ccs)
{
    const int k = (int)(lo_word(u) << 2);
    sn = T.sincos(k); ssn = T.sincos(k + 1);
    cs = T.sincos(k + 2); ccs = T.sincos(k + 3);
}

// FMA = true: the contraction of s_sin-fma.o (sin, cos); false: the separately rounded operations of s_sincos.o
template <bool FMA> ORT_LIBM_FN double taylor_sin(kptr K, double x, double dx)       // TAYLOR_SIN(x*x, x, dx), |x| < 0.126
{
    const double xx = x * x;
    if (FMA) {
        const double p = fma_(xx, fma_(xx, fma_(xx, fma_(xx, ORT_SCK(S5), ORT_SCK(S4)), ORT_SCK(S3)), ORT_SCK(S2)), ORT_SCK(S1));
        const double t = fma_(xx, fma_(p, x, -(0.5 * dx)), dx);
        return x + t;
    }
    const double p = ((((ORT_SCK(S5) * xx + ORT_SCK(S4)) * xx + ORT_SCK(S3)) * xx + ORT_SCK(S2)) * xx) + ORT_SCK(S1);
    const double t = ((p * x - 0.5 * dx) * xx + dx);
    return x + t;
}

template <bool FMA, class TB> ORT_LIBM_FN double do_sin_big(kptr K, const TB &T, double x, double dx)       // do_sin for |x| >= 0.126
{
    const double xold = x;
    if (x <= 0) dx = -dx;
    const double ax = abs_(x), u = ORT_SCK(BIG) + ax;
    x = ax - (u - ORT_SCK(BIG));
    double sn, ssn, cs, ccs;
    sincos_lookup(T, u, sn, ssn, cs, ccs);
    const double xx = x * x;
    double cor;
    if (FMA) {
        const double s = x + fma_(x * xx, fma_(xx, ORT_SCK(SN5), ORT_SCK(SN3)), dx);
        const double c = fma_(x, dx, xx * fma_(xx, fma_(xx, ORT_SCK(CS6), ORT_SCK(CS4)), ORT_SCK(CS2)));
        cor = fma_(s, cs, fma_(-c, sn, fma_(s, ccs, ssn)));
    } else {
        const double s = x + (dx + x * xx * (ORT_SCK(SN3) + xx * ORT_SCK(SN5)));
        const double c = x * dx + xx * (ORT_SCK(CS2) + xx * (ORT_SCK(CS4) + xx * ORT_SCK(CS6)));
        cor = (ssn + s * ccs - sn * c) + cs * s;
    }
    return copysign_(sn + cor, xold);
}
template <bool FMA, class TB> ORT_LIBM_FN double do_sin(kptr K, const TB &T, double x, double dx)
{
    if (abs_(x) < ORT_SCK(T126)) return taylor_sin<FMA>(K, x, dx);
    return do_sin_big<FMA>(K, T, x, dx);
}

template <bool FMA, class TB> ORT_LIBM_FN double do_cos(kptr K, const TB &T, double x, double dx)
{
    if (x < 0) dx = -dx;
    const double ax = abs_(x), u = ORT_SCK(BIG) + ax;
    x = ax - (u - ORT_SCK(BIG)) + dx;
    double sn, ssn, cs, ccs;
    sincos_lookup(T, u, sn, ssn, cs, ccs);
    const double xx = x * x;
    double cor;
    if (FMA) {
        const double s = fma_(x * xx, fma_(xx, ORT_SCK(SN5), ORT_SCK(SN3)), x);
        const double c = xx * fma_(xx, fma_(xx, ORT_SCK(CS6), ORT_SCK(CS4)), ORT_SCK(CS2));
        cor = fma_(-s, sn, fma_(-c, cs, fma_(-s, ssn, ccs)));
    } else {
        const double s = x + x * xx * (ORT_SCK(SN3) + xx * ORT_SCK(SN5));
        const double c = xx * (ORT_SCK(CS2) + xx * (ORT_SCK(CS4) + xx * ORT_SCK(CS6)));
        cor = (ccs - s * ssn - cs * c) - sn * s;
    }
    return cs + cor;
}

// reduce_sincos: x = n pi/2 + (a + da), 2.426265 <= |x| < 105414350
template <bool FMA> ORT_LIBM_FN int reduce_sincos(kptr K, double x, double &a, double &da)
{
    double t, xn, y, t2, db, b;
    if (FMA) {
        t = fma_(x, ORT_SCK(HPINV), ORT_SCK(TOINT));
        xn = t - ORT_SCK(TOINT);
        y = fma_(-xn, ORT_SCK(MP2), fma_(-xn, ORT_SCK(MP1), x));
        t2 = fma_(-xn, ORT_SCK(PP3), y);
        db = fma_(-ORT_SCK(PP3), xn, y - t2);
        b = fma_(-xn, ORT_SCK(PP4), t2);
        db = db + fma_(-xn, ORT_SCK(PP4), t2 - b);
    } else {
        t = (x * ORT_SCK(HPINV) + ORT_SCK(TOINT));
        xn = t - ORT_SCK(TOINT);
        y = (x - xn * ORT_SCK(MP1)) - xn * ORT_SCK(MP2);
        double t1 = xn * ORT_SCK(PP3);
        t2 = y - t1;
        db = (y - t2) - t1;
        t1 = xn * ORT_SCK(PP4);
        b = t2 - t1;
        db += (t2 - b) - t1;
    }
    a = b; da = db;
    return (int)(lo_word(t) & 3u);
}

template <bool FMA, class TB> ORT_LIBM_FN double do_sincos(kptr K, const TB &T, double a, double da, int n)
{
    double r = (n & 1) ? do_cos<FMA>(K, T, a, da) : do_sin<FMA>(K, T, a, da);
    return (n & 2) ? -r : r;
}

#ifndef ORT_LIBM_FALLBACK_SIN
#define ORT_LIBM_FALLBACK_SIN(x) ::sin(x)
#define ORT_LIBM_FALLBACK_COS(x) ::cos(x)
#define ORT_LIBM_FALLBACK_LOG(x) ::log(x)
#define ORT_LIBM_FALLBACK_ATAN2(y, x) ::atan2(y, x)
#define ORT_LIBM_FALLBACK_ACOS(x) ::acos(x)
#endif

// __sin (s_sin.c), FMA build
template <class TB> ORT_LIBM_FN double sin(const TB &T, double x)
{
    const kptr K = kbase();
    const int32_t k = hi_word(x) & 0x7fffffff;
    if (k < 0x3e500000) return x;                                    // |x| < 2^-26
    if (k < 0x3feb6000) return do_sin<true>(K, T, x, 0.0);                 // |x| < 0.855469
    if (k < 0x400368fd) {                                            // |x| < 2.426265
        const double t = ORT_SCK(HP0) - abs_(x);
        return copysign_(do_cos<true>(K, T, t, ORT_SCK(HP1)), x);
    }
    if (k < 0x419921FB) {                                            // |x| < 105414350
        double a, da;
        const int n = reduce_sincos<true>(K, x, a, da);
        return do_sincos<true>(K, T, a, da, n);
    }
    return ORT_LIBM_FALLBACK_SIN(x);
}

// __cos (s_sin.c), FMA build
template <class TB> ORT_LIBM_FN double cos(const TB &T, double x)
{
    const kptr K = kbase();
    const int32_t k = hi_word(x) & 0x7fffffff;
    if (k < 0x3e400000) return 1.0;                                  // |x| < 2^-27
    if (k < 0x3feb6000) return do_cos<true>(K, T, x, 0.0);
    if (k < 0x400368fd) {
        const double y = ORT_SCK(HP0) - abs_(x);
        const double a = y + ORT_SCK(HP1);
        const double da = (y - a) + ORT_SCK(HP1);
        return do_sin<true>(K, T, a, da);
    }
    if (k < 0x419921FB) {
        double a, da;
        const int n = reduce_sincos<true>(K, x, a, da);
        return do_sincos<true>(K, T, a, da, n + 1);
    }
    return ORT_LIBM_FALLBACK_COS(x);
}

// __sincos (s_sincos.c): the same pieces built WITHOUT fused multiply-adds (glibc 2.35 has no multiarch
// variant of the double sincos), and with its own split of the middle range — so sincos(x) differs from
// (sin(x), cos(x)) in the last bit for ~0.1 % of the arguments.  The reference's compilers turn most of
// its sin/cos pairs into ONE sincos call (which ones: ort_device.h at each call site).
template <class TB> ORT_LIBM_FN SinCos sincos(const TB &T, double x)
{
    const kptr K = kbase();
    const int32_t k = hi_word(x) & 0x7fffffff;
    if (k < 0x400368fd) {
        if (k < 0x3e400000) return {x, 1.0};
        if (k < 0x3feb6000) return {do_sin<false>(K, T, x, 0.0), do_cos<false>(K, T, x, 0.0)};
        const double y = ORT_SCK(HP0) - abs_(x);
        const double a = y + ORT_SCK(HP1);
        const double da = (y - a) + ORT_SCK(HP1);
        return {copysign_(do_cos<false>(K, T, a, da), x), do_sin<false>(K, T, a, da)};
    }
    if (k < 0x419921FB) {
        double a, da;
        const int n = reduce_sincos<false>(K, x, a, da);
        return {do_sincos<false>(K, T, a, da, n), do_sincos<false>(K, T, a, da, n + 1)};
    }
    return {ORT_LIBM_FALLBACK_SIN(x), ORT_LIBM_FALLBACK_COS(x)};
}

// ---------------------------------------------------------------------------------------------------------
// log: e_log.c (Szabolcs Nagy's), FMA build.  x = 2^k z, z in [0x1.6p-1, 0x1.6p0); table of 128 (1/c, log c)
// with c near the centre of each subinterval; r = z/c - 1 exactly by FMA; degree-5 polynomial.  Near 1
// (1 - 2^-4 <= x < 1 + 0x1.09p-4) a degree-11 polynomial in r = x - 1 with a split r*r.
// ---------------------------------------------------------------------------------------------------------
template <class TB> ORT_LIBM_FN double log(const TB &T, double x)
{
    const kptr K = kbase();
    const uint64_t ix = bits(x);
    const uint32_t top = (uint32_t)(ix >> 48);
#define ORT_LOGD(i) T.logd(i)
    if (ix - 0x3fee000000000000ull < 0x3090000000000ull) {
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r = x - 1.0;
        const double r2 = r * r, r3 = r * r2;
        // B[j] = poly1[j] = data[7 + j]
        const double p1 = fma_(r2, ORT_LOGD(10), fma_(r, ORT_LOGD(9), ORT_LOGD(8)));            // B1 + r B2 + r2 B3
        const double p2 = fma_(r2, ORT_LOGD(13), fma_(r, ORT_LOGD(12), ORT_LOGD(11)));          // B4 + r B5 + r2 B6
        double p3 = fma_(r2, ORT_LOGD(16), fma_(r, ORT_LOGD(15), ORT_LOGD(14)));                // B7 + r B8 + r2 B9
        p3 = fma_(r3, ORT_LOGD(17), p3);                                                        // + r3 B10
        const double P = fma_(fma_(p3, r3, p2), r3, p1);
        const double t = fma_(r, ORT_LGK(TWO27), r);
        const double rhi = fma_(ORT_LGK(MTWO27), r, t);
        const double rlo = r - rhi;
        const double rhi2 = rhi * rhi;
        const double B0 = ORT_LOGD(7);                                                          // -0.5
        const double hi = fma_(rhi2, B0, r);
        double lo = fma_(rhi2, B0, r - hi);
        lo = fma_(B0 * rlo, rhi + r, lo);
        const double y = fma_(P, r3, lo);
        return hi + y;
    }
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
        if (ix * 2 == 0) return -__builtin_huge_val();               // log(+-0) = -inf
        return ORT_LIBM_FALLBACK_LOG(x);                             // negative, NaN, inf, subnormal
    }
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> 45) & 127u);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
    const double invc = ORT_LOGD(18 + 2 * i), logc = ORT_LOGD(18 + 2 * i + 1);
    const double z = dbl(iz);
    const double r = fma_(z, invc, -1.0);
    const double kd = (double)k;
    const double w = fma_(kd, ORT_LOGD(0), logc);                    // kd Ln2hi + logc
    const double hi = r + w;
    double lo = fma_(kd, ORT_LOGD(1), (w - hi) + r);                 // + kd Ln2lo
    const double r2 = r * r;
    // A[j] = poly[j] = data[2 + j]
    lo = fma_(r2, ORT_LOGD(2), lo);
    const double q = fma_(fma_(r, ORT_LOGD(6), ORT_LOGD(5)), r2, fma_(r, ORT_LOGD(4), ORT_LOGD(3)));
    return fma_(r * r2, q, lo) + hi;
#undef ORT_LOGD
}

// ---------------------------------------------------------------------------------------------------------
// atan2: e_atan2.c (IBM Accurate Mathematical Library, slow paths removed in glibc 2.28), FMA build.
// u = min(|x|,|y|) / max(|x|,|y|) with its rounding error du (EMULV by FMA); u < 1/16: odd polynomial; else
// table cij[i] = {u_i, atan u_i, 5 polynomial coefficients} at i = round(256 u) - 16; the four quadrant
// combinations add / subtract pi/2 or pi as double-doubles.
// ---------------------------------------------------------------------------------------------------------


template <class TB> ORT_LIBM_FN double atan2(const TB &T, double y, double x)
{
    const kptr K = kbase();
    const uint64_t bx = bits(x), by = bits(y);
    const int32_t ux = (int32_t)(bx >> 32), uy = (int32_t)(by >> 32);
    const uint32_t dy = (uint32_t)by;
    if ((ux & 0x7ff00000) == 0x7ff00000 || (uy & 0x7ff00000) == 0x7ff00000) return ORT_LIBM_FALLBACK_ATAN2(y, x);   // NaN, inf
    if (uy == 0 && dy == 0) return (ux < 0) ? ORT_ATK(OPI) : 0.0;                                // y = +0
    if ((uint32_t)uy == 0x80000000u && dy == 0) return (ux < 0) ? -ORT_ATK(OPI) : -0.0;          // y = -0
    if (x == 0.0) return (uy < 0) ? -ORT_ATK(HPI) : ORT_ATK(HPI);                                // x = +-0
    double ax = (x < 0) ? -x : x, ay = (y < 0) ? -y : y;
    const int de = (uy & 0x7ff00000) - (ux & 0x7ff00000);
    if (de >= 59768832) return (y > 0) ? ORT_ATK(HPI) : -ORT_ATK(HPI);
    if (de <= -59768832) {
        if (x > 0) {
            const double z = ay / ax;
            if (z < 0x1p-1022) return ORT_LIBM_FALLBACK_ATAN2(y, x);
            return copysign_(z, y);
        }
        return (y > 0) ? ORT_ATK(OPI) : -ORT_ATK(OPI);
    }
    if (ax < 0x1p-500 || ay < 0x1p-500) { ax *= 0x1p500; ay *= 0x1p500; }
    if (ax > 0x1p500 || ay > 0x1p500) { ax *= 0x1p-500; ay *= 0x1p-500; }
    double u, du;
    const bool y_small = ay < ax;              // (i) / (iv): u = ay/ax; else u = ax/ay
    const bool x_small = ax < ay;              // x <= 0: (iii) if |x| < |y|, else (iv) — |x| = |y| is (iv) with u = 1
    {
        const double num = y_small ? ay : ax, den = y_small ? ax : ay;
        u = num / den;
        const double v = den * u;
        const double vv = fma_(den, u, -v);
        du = ((num - v) - vv) / den;
    }
    const bool small = u < 0.0625;
    double z;
    if (small) {
        const double v = u * u;
        const double p = fma_(v, fma_(v, fma_(v, fma_(v, fma_(v, ORT_ATK(D13), ORT_ATK(D11)), ORT_ATK(D9)), ORT_ATK(D7)), ORT_ATK(D5)), ORT_ATK(D3));
        if (x > 0 && y_small) {                          // (i) atan(ay/ax)
            const double zz = fma_(u * v, p, du);
            z = u + zz;
        } else {
            const double zz = (u * v) * p;
            // (ii) pi/2 - u, (iii) pi/2 + u, (iv) pi - u: ESUB / EADD of the constant and u, then the low parts
            const bool plus = !(x > 0) && x_small;       // (iii)
            const bool pi = !(x > 0) && !x_small;        // (iv)
            const double C = pi ? ORT_ATK(OPI) : ORT_ATK(HPI), C1 = pi ? ORT_ATK(OPI1) : ORT_ATK(HPI1);
            if (plus) {
                const double t2 = u + C;                                     // EADD(hpi, u, t2, cor)
                const double cor = (C > abs_(u)) ? (C - t2) + u : (u - t2) + C;
                const double t3 = ((C1 + cor) + du) + zz;
                z = t2 + t3;
            } else {
                const double t2 = C - u;                                     // ESUB(C, u, t2, cor)
                const double cor = (C > abs_(u)) ? (C - t2) - u : C - (u + t2);
                const double t3 = ((C1 + cor) - du) - zz;
                z = t2 + t3;
            }
        }
    } else {
        const int i = (int)(fma_(u, ORT_ATK(TWO8), ORT_ATK(TWO52)) - ORT_ATK(TWO52)) - 16;
#define ORT_CIJ(j) T.cij(7 * i + (j))
        if (x > 0 && y_small) {                          // (i)
            const double t3 = u - ORT_CIJ(0);
            const double v = t3 + du;                                        // EADD(t3, du, v, dv)
            const double dv = (abs_(t3) > abs_(du)) ? (t3 - v) + du : (du - v) + t3;
            const double t1 = ORT_CIJ(1), t2 = ORT_CIJ(2);
            const double q = fma_(v, fma_(v, fma_(v, ORT_CIJ(6), ORT_CIJ(5)), ORT_CIJ(4)), ORT_CIJ(3));
            const double zz = fma_(v, t2, fma_(dv, t2, (v * v) * q));
            z = t1 + zz;
        } else {
            const double v = (u - ORT_CIJ(0)) + du;
            const double q = fma_(v, fma_(v, fma_(v, fma_(v, ORT_CIJ(6), ORT_CIJ(5)), ORT_CIJ(4)), ORT_CIJ(3)), ORT_CIJ(2));
            if (x > 0) {                                 // (ii) pi/2 - atan(ax/ay)
                const double zz = fma_(-v, q, ORT_ATK(HPI1));
                z = (ORT_ATK(HPI) - ORT_CIJ(1)) + zz;
            } else if (x_small) {                        // (iii) pi/2 + atan(ax/ay)
                const double zz = fma_(v, q, ORT_ATK(HPI1));
                z = (ORT_ATK(HPI) + ORT_CIJ(1)) + zz;
            } else {                                     // (iv) pi - atan(ay/ax)
                const double zz = fma_(-v, q, ORT_ATK(OPI1));
                z = (ORT_ATK(OPI) - ORT_CIJ(1)) + zz;
            }
        }
#undef ORT_CIJ
    }
    return copysign_(z, y);
}

// ---------------------------------------------------------------------------------------------------------
// acos: e_asin.c (IBM), FMA build.  |x| < 1/8: pi/2 - x - x^3 P(x^2); 1/8 <= |x| < 0.96875: six ranges of
// table-driven polynomials around grid points (asncs: blocks of 11 - 15 doubles); 0.96875 <= |x| < 1:
// 2 asin(sqrt((1 - |x|)/2)) with an inline square root (inroot / powtwo seeds + one correction).
// ---------------------------------------------------------------------------------------------------------


template <class TB> ORT_LIBM_FN double acos(const TB &T, double x)
{
    const kptr K = kbase();
    const int32_t m = hi_word(x), k = m & 0x7fffffff;
    if (k < 0x3c880000) return ORT_ACK(HP0);
    if (k < 0x3fc00000) {
        const double x2 = x * x;
        const double p = fma_(x2, fma_(x2, fma_(x2, fma_(x2, fma_(x2, ORT_ACK(F6), ORT_ACK(F5)), ORT_ACK(F4)), ORT_ACK(F3)), ORT_ACK(F2)), ORT_ACK(F1));
        const double r = ORT_ACK(HP0) - x;
        const double cor = fma_(-p, x * x2, ((ORT_ACK(HP0) - r) - x) + ORT_ACK(HP1));
        return r + cor;
    }
    if (k < 0x3fef0000) {
        // block n of asncs: [0] grid point, [1] first-order coefficient, [2 .. top] higher coefficients,
        // [top + 1] low part of the constant, [top + 2] the constant
        int n, top;
        if (k < 0x3fd00000) { n = 11 * ((k & 0x000fffff) >> 15); top = 6; }
        else if (k < 0x3fe00000) { n = 11 * ((k & 0x000fffff) >> 14) + 352; top = 6; }
        else if (k < 0x3fe80000) { n = 1056 + ((k & 0x000fe000) >> 11) * 3; top = 7; }
        else if (k < 0x3fed8000) { n = 992 + ((k & 0x000fe000) >> 13) * 13; top = 8; }
        else if (k < 0x3fee8000) { n = 884 + ((k & 0x000fe000) >> 13) * 14; top = 9; }
        else { n = 768 + ((k & 0x000fe000) >> 13) * 15; top = 10; }
#define ORT_ASN(j) T.asncs(n + (j))
        const double xs = (m > 0) ? x : -x;
        const double xx = xs - ORT_ASN(0);
        double q = ORT_ASN(top);
        for (int j = top - 1; j >= 2; --j) q = fma_(xx, q, ORT_ASN(j));
        double t = fma_(xx * xx, q, ORT_ASN(top + 1));
        t = fma_(xx, ORT_ASN(1), t);
        const double c = ORT_ASN(top + 2);
#undef ORT_ASN
        if (m > 0) return (ORT_ACK(HP1) - t) + (ORT_ACK(HP0) - c);
        return (t + ORT_ACK(HP1)) + (c + ORT_ACK(HP0));
    }
    if (k < 0x3ff00000) {
        const double z = ((m > 0) ? (1.0 - x) : (x + 1.0)) * 0.5;
        const int32_t kz = hi_word(z);
        double t = T.inroot((kz >> 14) & 0x7f) * T.powtwo(511 - (kz >> 21));
        const double r = fma_(-(t * t), z, 1.0);
        t = fma_(r, fma_(r, fma_(r, ORT_ACK(RT3), ORT_ACK(RT2)), ORT_ACK(RT1)), ORT_ACK(RT0)) * t;
        const double c = z * t;
        const double h = fma_(-c, t * 0.5, 1.5);                    // 1.5 - 0.5 t c
        const double y = fma_(ORT_LGK(MTWO27), c, fma_(c, ORT_LGK(TWO27), c));      // c rounded to 26 bits
        const double cc = fma_(-y, y, z) / fma_(h, c, y);
        const double p = fma_(z, fma_(z, fma_(z, fma_(z, fma_(z, ORT_ACK(F6), ORT_ACK(F5)), ORT_ACK(F4)), ORT_ACK(F3)), ORT_ACK(F2)), ORT_ACK(F1)) * z;
        const double e = p * (y + cc);
        if (m >= 0) {
            const double res = (cc + e) + y;
            return res + res;
        }
        const double res = ((ORT_ACK(HP1) - cc) - e) + (ORT_ACK(HP0) - y);
        return res + res;
    }
    if (k == 0x3ff00000 && lo_word(x) == 0) return (m > 0) ? 0.0 : ORT_ACK(PI);
    return ORT_LIBM_FALLBACK_ACOS(x);
}


// =========================================================================================================
// The same functions as PREDICATED DATAFLOW — what the kernels call.  The straight versions above branch on the
// argument's range; in a wavefront of 64 rays every range is present, so a GPU would run all of them one after
// the other behind exec masks (the first port of the scattering walk did: 1.7 x slower, 160 bytes of scratch
// per lane).  Here every lane evaluates every range's (short) formula once, in a fixed order, and selects;
// whatever two ranges share is computed once (one do_sin and one do_cos per sincos; one polynomial chain for the
// four quadrant forms of atan2; one Horner loop for the five table ranges of acos).  Operation for operation
// the selected result is the straight version's: tests/csrc/check_libm_host.cpp and check_libm_gpu.hip hold
// BOTH against the host's libm.  Arguments outside the domain (top of this file) and the rare special cases
// take the straight version behind a wave-uniform branch (ORT_LIBM_ANY).
// =========================================================================================================
#if defined(ORT_LIBM_NO_RARE)         // development (tools/ubench_libm.hip): the main paths alone, to count their instructions
#define ORT_LIBM_ANY(c) (false)
#elif defined(__HIP_DEVICE_COMPILE__)
#define ORT_LIBM_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
#else
#define ORT_LIBM_ANY(c) (c)
#endif
ORT_LIBM_FN double sel(bool c, double a, double b) { return c ? a : b; }
ORT_LIBM_FN double neg_if_(double x, bool c) { return dbl(bits(x) ^ (c ? 0x8000000000000000ull : 0ull)); }

// do_sin(A, DA) and do_cos(A, DA) of one argument pair each, sharing the table row where the two arguments are
// the same number (As == Ac: every case but the middle range of the FMA sin / cos pair)
template <bool FMA, class TB> ORT_LIBM_FN void do_sin_cos(kptr K, const TB &T, double As, double DAs, double Ac, double DAc, double &S, double &C)
{
    // do_sin: |As| < 0.126 Taylor, else table
    const double st = taylor_sin<FMA>(K, As, DAs);
    const double dxs = neg_if_(DAs, As <= 0);
    const double axs = abs_(As), us = ORT_SCK(BIG) + axs;
    const double xs = axs - (us - ORT_SCK(BIG));
    double sn, ssn, cs, ccs;
    sincos_lookup(T, us, sn, ssn, cs, ccs);
    {
        const double xx = xs * xs;
        double cor;
        if (FMA) {
            const double s = xs + fma_(xs * xx, fma_(xx, ORT_SCK(SN5), ORT_SCK(SN3)), dxs);
            const double c = fma_(xs, dxs, xx * fma_(xx, fma_(xx, ORT_SCK(CS6), ORT_SCK(CS4)), ORT_SCK(CS2)));
            cor = fma_(s, cs, fma_(-c, sn, fma_(s, ccs, ssn)));
        } else {
            const double s = xs + (dxs + xs * xx * (ORT_SCK(SN3) + xx * ORT_SCK(SN5)));
            const double c = xs * dxs + xx * (ORT_SCK(CS2) + xx * (ORT_SCK(CS4) + xx * ORT_SCK(CS6)));
            cor = (ssn + s * ccs - sn * c) + cs * s;
        }
        S = sel(axs < ORT_SCK(T126), st, copysign_(sn + cor, As));
    }
    // do_cos
    const double dxc = neg_if_(DAc, Ac < 0);
    const double axc = abs_(Ac), uc = ORT_SCK(BIG) + axc;
    if (ORT_LIBM_ANY(lo_word(uc) != lo_word(us))) sincos_lookup(T, uc, sn, ssn, cs, ccs);    // (never in sincos(): As == Ac)
    {
        const double x = axc - (uc - ORT_SCK(BIG)) + dxc;
        const double xx = x * x;
        double cor;
        if (FMA) {
            const double s = fma_(x * xx, fma_(xx, ORT_SCK(SN5), ORT_SCK(SN3)), x);
            const double c = xx * fma_(xx, fma_(xx, ORT_SCK(CS6), ORT_SCK(CS4)), ORT_SCK(CS2));
            cor = fma_(-s, sn, fma_(-c, cs, fma_(-s, ssn, ccs)));
        } else {
            const double s = x + x * xx * (ORT_SCK(SN3) + xx * ORT_SCK(SN5));
            const double c = xx * (ORT_SCK(CS2) + xx * (ORT_SCK(CS4) + xx * ORT_SCK(CS6)));
            cor = (ccs - s * ssn - cs * c) - sn * s;
        }
        C = cs + cor;
    }
}

// sincos(x) (FMA = false: glibc's sincos) or the pair (sin(x), cos(x)) (FMA = true: glibc's sin and cos, which treat
// the middle range differently from each other and from sincos)
template <bool FMA, class TB> ORT_LIBM_FN SinCos sincos_p(const TB &T, double x)
{
    const kptr K = kbase();
    const int32_t k = hi_word(x) & 0x7fffffff;
    const bool r1 = k < 0x3feb6000, r2 = !r1 && k < 0x400368fd;
    if (ORT_LIBM_ANY(!(k < 0x419921FB))) {                       // huge, inf, NaN: the straight versions (fallback inside)
        if (!(k < 0x419921FB)) return FMA ? SinCos{sin(T, x), cos(T, x)} : sincos(T, x);
    }
    // range 3: x = n pi/2 + (a3 + da3)
    double a3, da3;
    const int n = reduce_sincos<FMA>(K, x, a3, da3);
    // range 2: pi/2 - |x| as (a2 + da2); glibc's sin uses (hp0 - |x|, hp1) instead, for its do_cos
    const double y2 = ORT_SCK(HP0) - abs_(x);
    const double a2 = y2 + ORT_SCK(HP1);
    const double da2 = (y2 - a2) + ORT_SCK(HP1);
    const double As = sel(r1, x, sel(r2, a2, a3)), DAs = sel(r1, 0.0, sel(r2, da2, da3));
    double Ac = As, DAc = DAs;
    if (FMA) { Ac = sel(r2, y2, As); DAc = sel(r2, ORT_SCK(HP1), DAs); }
    double S, C;
    do_sin_cos<FMA>(K, T, As, DAs, Ac, DAc, S, C);
    // which of the two is the sine: range 1 S; range 2 C with x's sign (C > 0); range 3 by n's parity, sign by n & 2
    const bool odd = (n & 1) != 0;
    const bool swap = r2 || (!r1 && odd);
    const bool neg_s = r2 ? (hi_word(x) < 0) : (!r1 && (n & 2) != 0);
    const bool neg_c = !r1 && !r2 && ((n + 1) & 2) != 0;
    SinCos r = {neg_if_(sel(swap, C, S), neg_s), neg_if_(sel(swap, S, C), neg_c)};
    // |x| < 2^-27: (x, 1); glibc's sin() returns x below 2^-26 already
    const bool tiny_s = k < (FMA ? 0x3e500000 : 0x3e400000), tiny_c = k < 0x3e400000;
    r.s = sel(tiny_s, x, r.s);
    r.c = sel(tiny_c, 1.0, r.c);
    return r;
}

template <class TB> ORT_LIBM_FN double log_p(const TB &T, double x)
{
    const kptr K = kbase();
    const uint64_t ix = bits(x);
    const uint32_t top = (uint32_t)(ix >> 48);
    const bool odd = (top - 0x0010u >= 0x7ff0u - 0x0010u) && ix * 2 != 0;     // negative, NaN, inf, subnormal
    if (ORT_LIBM_ANY(odd)) { if (odd) return log(T, x); }
#define ORT_LOGD(i) T.logd(i)
    const bool near = ix - 0x3fee000000000000ull < 0x3090000000000ull;
    double rn;
    {
        const double r = x - 1.0;
        const double r2 = r * r, r3 = r * r2;
        const double p1 = fma_(r2, ORT_LOGD(10), fma_(r, ORT_LOGD(9), ORT_LOGD(8)));
        const double p2 = fma_(r2, ORT_LOGD(13), fma_(r, ORT_LOGD(12), ORT_LOGD(11)));
        double p3 = fma_(r2, ORT_LOGD(16), fma_(r, ORT_LOGD(15), ORT_LOGD(14)));
        p3 = fma_(r3, ORT_LOGD(17), p3);
        const double P = fma_(fma_(p3, r3, p2), r3, p1);
        const double t = fma_(r, ORT_LGK(TWO27), r);
        const double rhi = fma_(ORT_LGK(MTWO27), r, t);
        const double rlo = r - rhi;
        const double rhi2 = rhi * rhi;
        const double B0 = ORT_LOGD(7);
        const double hi = fma_(rhi2, B0, r);
        double lo = fma_(rhi2, B0, r - hi);
        lo = fma_(B0 * rlo, rhi + r, lo);
        rn = hi + fma_(P, r3, lo);
        rn = sel(ix == 0x3ff0000000000000ull, 0.0, rn);
    }
    double rm;
    {
        const uint64_t tmp = ix - 0x3fe6000000000000ull;
        const int i = (int)((tmp >> 45) & 127u);
        const int k = (int)((int64_t)tmp >> 52);
        const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
        const double invc = ORT_LOGD(18 + 2 * i), logc = ORT_LOGD(18 + 2 * i + 1);
        const double r = fma_(dbl(iz), invc, -1.0);
        const double kd = (double)k;
        const double w = fma_(kd, ORT_LOGD(0), logc);
        const double hi = r + w;
        double lo = fma_(kd, ORT_LOGD(1), (w - hi) + r);
        const double r2 = r * r;
        lo = fma_(r2, ORT_LOGD(2), lo);
        const double q = fma_(fma_(r, ORT_LOGD(6), ORT_LOGD(5)), r2, fma_(r, ORT_LOGD(4), ORT_LOGD(3)));
        rm = fma_(r * r2, q, lo) + hi;
    }
#undef ORT_LOGD
    return sel(ix * 2 == 0, -__builtin_huge_val(), sel(near, rn, rm));
}

template <class TB> ORT_LIBM_FN double atan2_p(const TB &T, double y, double x)
{
    const kptr K = kbase();
    const uint64_t bx = bits(x), by = bits(y);
    const int32_t ex = (int32_t)(bx >> 32) & 0x7ff00000, ey = (int32_t)(by >> 32) & 0x7ff00000;
    const int de = ey - ex;
    // zeros, NaN, inf, exponents 57 apart, operands that glibc rescales (below 2^-500 or above 2^500): straight version
    const bool special = (bx << 1) == 0 || (by << 1) == 0 || ex == 0x7ff00000 || ey == 0x7ff00000 || de >= 59768832 || de <= -59768832 ||
                         ex < 0x20b00000 || ey < 0x20b00000 || ex >= 0x5f300000 || ey >= 0x5f300000;
    if (ORT_LIBM_ANY(special)) { if (special) return atan2(T, y, x); }
    const double ax = abs_(x), ay = abs_(y);
    const bool y_small = ay < ax, x_small = ax < ay, xpos = x > 0;
    const double num = sel(y_small, ay, ax), den = sel(y_small, ax, ay);
    const double u = num / den;
    const double vq = den * u;
    const double du = ((num - vq) - fma_(den, u, -vq)) / den;
    const bool case1 = xpos && y_small;                      // (i)   atan(ay/ax)
    const bool plus = !xpos && x_small;                      // (iii) pi/2 + atan(ax/ay)
    const bool pi = !xpos && !x_small;                       // (iv)  pi - atan(ay/ax);   else (ii) pi/2 - atan(ax/ay)
    const double C = sel(pi, ORT_ATK(OPI), ORT_ATK(HPI)), C1 = sel(pi, ORT_ATK(OPI1), ORT_ATK(HPI1));
    double zs;
    {   // u < 1/16
        const double v = u * u;
        const double p = fma_(v, fma_(v, fma_(v, fma_(v, fma_(v, ORT_ATK(D13), ORT_ATK(D11)), ORT_ATK(D9)), ORT_ATK(D7)), ORT_ATK(D5)), ORT_ATK(D3));
        const double uv = u * v;
        const double z1 = u + fma_(uv, p, du);
        // (ii) (iv): C - u ...; (iii): C + u ...: one form on sign-flipped u, du, zz (a - b == a + (-b) bit for bit; C > |u| here)
        const double su = neg_if_(u, !plus), sdu = neg_if_(du, !plus), szz = neg_if_(uv * p, !plus);
        const double t2 = C + su;
        const double cor = (C - t2) + su;
        const double t3 = ((C1 + cor) + sdu) + szz;
        zs = sel(case1, z1, t2 + t3);
    }
    double zt;
    {   // table: cij[i] = {u_i, atan u_i, c2 .. c6}
        int i = (int)(fma_(u, ORT_ATK(TWO8), ORT_ATK(TWO52)) - ORT_ATK(TWO52)) - 16;
        i = i < 0 ? 0 : i;                                   // (lanes with u < 1/16: result unused)
#define ORT_CIJ(j) T.cij(7 * i + (j))
        const double t3 = u - ORT_CIJ(0);
        const double v = t3 + du;
        const double dv = sel(abs_(t3) > abs_(du), (t3 - v) + du, (du - v) + t3);
        const double c1 = ORT_CIJ(1), c2 = ORT_CIJ(2);
        const double q3 = fma_(v, fma_(v, fma_(v, ORT_CIJ(6), ORT_CIJ(5)), ORT_CIJ(4)), ORT_CIJ(3));
#undef ORT_CIJ
        const double z1 = c1 + fma_(v, c2, fma_(dv, c2, (v * v) * q3));
        const double q = fma_(v, q3, c2);
        const double zo = (C + neg_if_(c1, !plus)) + fma_(neg_if_(v, !plus), q, C1);
        zt = sel(case1, z1, zo);
    }
    return copysign_(sel(u < 0.0625, zs, zt), y);
}

template <class TB> ORT_LIBM_FN double acos_p(const TB &T, double x)
{
    const kptr K = kbase();
    const int32_t m = hi_word(x), k = m & 0x7fffffff;
    const bool odd = k >= 0x3ff00000;                        // |x| >= 1, inf, NaN
    if (ORT_LIBM_ANY(odd)) { if (odd) return acos(T, x); }
    const bool pos = m > 0;
    // |x| < 1/8
    double rb;
    {
        const double x2 = x * x;
        const double p = fma_(x2, fma_(x2, fma_(x2, fma_(x2, fma_(x2, ORT_ACK(F6), ORT_ACK(F5)), ORT_ACK(F4)), ORT_ACK(F3)), ORT_ACK(F2)), ORT_ACK(F1));
        const double r = ORT_ACK(HP0) - x;
        rb = r + fma_(-p, x * x2, ((ORT_ACK(HP0) - r) - x) + ORT_ACK(HP1));
        rb = sel(k < 0x3c880000, ORT_ACK(HP0), rb);
    }
    // 1/8 <= |x| < 0.96875: the block of asncs and the degree of its polynomial
    double rt;
    {
        const bool in = k >= 0x3fc00000 && k < 0x3fef0000;
        int n = 11 * ((k >> 15) & 0x1f), top = 6;
        if (k >= 0x3fd00000) n = 11 * ((k >> 14) & 0x3f) + 352;
        if (k >= 0x3fe00000) { n = 1056 + ((k >> 11) & 0x1fc) * 3; top = 7; }
        if (k >= 0x3fe80000) { n = 992 + ((k >> 13) & 0x7f) * 13; top = 8; }
        if (k >= 0x3fed8000) { n = 884 + ((k >> 13) & 0x7f) * 14; top = 9; }
        if (k >= 0x3fee8000) { n = 768 + ((k >> 13) & 0x7f) * 15; top = 10; }
        n = in ? n : 0;
#define ORT_ASN(j) T.asncs(n + (j))
        const double xx = abs_(x) - ORT_ASN(0);
        // Horner from the longest block's degree: a step above a shorter block's own degree is fma(xx, 0, 0) = 0, and the
        // step AT its degree fma(xx, 0, a_top) = a_top: the shorter chains, bit for bit
        double q = 0.0;
#pragma unroll
        for (int j = 10; j >= 2; --j) q = fma_(xx, q, j <= top ? ORT_ASN(j) : 0.0);
        double t = fma_(xx * xx, q, ORT_ASN(top + 1));
        t = fma_(xx, ORT_ASN(1), t);
        const double c = ORT_ASN(top + 2);
#undef ORT_ASN
        // x > 0: (hp1 - t) + (hp0 - c);  x < 0: (hp1 + t) + (hp0 + c)
        rt = (ORT_ACK(HP1) + neg_if_(t, pos)) + (ORT_ACK(HP0) + neg_if_(c, pos));
    }
    // 0.96875 <= |x| < 1
    double rh;
    {
        const double z = (1.0 - abs_(x)) * 0.5;
        const int32_t kz = hi_word(z);
        const bool in = k >= 0x3fef0000;
        const int ir = in ? ((kz >> 14) & 0x7f) : 0, ip = in ? 511 - (kz >> 21) : 0;
        double t = T.inroot(ir) * T.powtwo(ip);
        const double r = fma_(-(t * t), z, 1.0);
        t = fma_(r, fma_(r, fma_(r, ORT_ACK(RT3), ORT_ACK(RT2)), ORT_ACK(RT1)), ORT_ACK(RT0)) * t;
        const double c = z * t;
        const double h = fma_(-c, t * 0.5, 1.5);
        const double yy = fma_(ORT_LGK(MTWO27), c, fma_(c, ORT_LGK(TWO27), c));
        const double cc = fma_(-yy, yy, z) / fma_(h, c, yy);
        const double p = fma_(z, fma_(z, fma_(z, fma_(z, fma_(z, ORT_ACK(F6), ORT_ACK(F5)), ORT_ACK(F4)), ORT_ACK(F3)), ORT_ACK(F2)), ORT_ACK(F1)) * z;
        const double e = p * (yy + cc);
        const double rp = (cc + e) + yy;
        const double rn = ((ORT_ACK(HP1) - cc) - e) + (ORT_ACK(HP0) - yy);
        const double res = sel(m >= 0, rp, rn);
        rh = res + res;
    }
    return sel(k < 0x3fc00000, rb, sel(k < 0x3fef0000, rt, rh));
}


// the same with the tables where they are (constant memory)
ORT_LIBM_FN double sin(double x) { return sin(TabGlobal{}, x); }
ORT_LIBM_FN double cos(double x) { return cos(TabGlobal{}, x); }
ORT_LIBM_FN SinCos sincos(double x) { return sincos(TabGlobal{}, x); }
ORT_LIBM_FN double log(double x) { return log(TabGlobal{}, x); }
ORT_LIBM_FN double atan2(double y, double x) { return atan2(TabGlobal{}, y, x); }
ORT_LIBM_FN double acos(double x) { return acos(TabGlobal{}, x); }
template <bool FMA> ORT_LIBM_FN SinCos sincos_p(double x) { return sincos_p<FMA>(TabGlobal{}, x); }
ORT_LIBM_FN double log_p(double x) { return log_p(TabGlobal{}, x); }
ORT_LIBM_FN double atan2_p(double y, double x) { return atan2_p(TabGlobal{}, y, x); }
ORT_LIBM_FN double acos_p(double x) { return acos_p(TabGlobal{}, x); }

}  // namespace glibc
}  // namespace ort
