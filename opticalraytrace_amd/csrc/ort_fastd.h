// ort_fastd.h — `fastd`: fp64 with relaxed rounding, the arithmetic of the "fast fp64" mode
// (ort_set_precision(ctx, 2)).
//
// The traced code is a template on its real type T (ort_device.h).  T = double evaluates every
// + - * / sqrt as the reference does (separately, correctly rounded: bit-exact results).
// T = fastd keeps fp64 range and ~1e-16 relative accuracy per operation but lets the hardware
// do what it is good at: multiply-adds are fused (FMA contraction), x / y is x * (1/y) with the
// v_rcp_f64 seed refined by two Newton steps, sqrt(x) is a Goldschmidt iteration on the
// v_rsq_f64 seed, and normalising a vector multiplies by rsqrt directly.  Each operation is
// accurate to ~1-2 ulp instead of 0.5 ulp; a traced ray differs from the exact path by ~1e-13
// relative (measured, tests/test_gpu_fastd.py) — three orders inside the 1e-10 of BASELINE's
// north star — but it is NOT bit-identical, so discrete outcomes can flip for ~1e-6 of the rays.
// It is an opt-in; the default and everything the parity tests call "exact" is T = double.
#pragma once
#include <hip/hip_runtime.h>

namespace ort {

// the fastd overloads below must not hide the global double / float functions inside this namespace
using ::sqrt;
using ::fabs;
using ::floor;
using ::log;
using ::acos;
using ::atan2;

struct fastd {
    double v;
    fastd() = default;
    __host__ __device__ constexpr fastd(double x) : v(x) {}
    __host__ __device__ explicit operator double() const { return v; }
    __host__ __device__ explicit operator int() const { return (int)v; }
    __host__ __device__ explicit operator float() const { return (float)v; }
};

#define ORT_FASTD_BIN(op)                                                  \
    __device__ inline fastd operator op(fastd a, fastd b)                  \
    {                                                                      \
        _Pragma("clang fp contract(fast)") return fastd(a.v op b.v);       \
    }
ORT_FASTD_BIN(+)
ORT_FASTD_BIN(-)
ORT_FASTD_BIN(*)
#undef ORT_FASTD_BIN
__device__ inline fastd operator-(fastd a) { return fastd(-a.v); }
#define ORT_FASTD_CMP(op) \
    __device__ inline bool operator op(fastd a, fastd b) { return a.v op b.v; }
ORT_FASTD_CMP(<)
ORT_FASTD_CMP(>)
ORT_FASTD_CMP(<=)
ORT_FASTD_CMP(>=)
ORT_FASTD_CMP(==)
ORT_FASTD_CMP(!=)
#undef ORT_FASTD_CMP

// 1/y: v_rcp_f64 seed (>= 22 bits) + two Newton steps (~1 ulp)
__device__ inline double rcp_nr2(double y)
{
    double r = __builtin_amdgcn_rcp(y);
    double e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-y, r, 1.0);
    return __builtin_fma(r, e, r);
}

// 1/sqrt(s): v_rsq_f64 seed + two Newton steps (~1 ulp)
__device__ inline double rsq_nr2(double s)
{
    double y = __builtin_amdgcn_rsq(s);
    double h = 0.5 * s * y;
    double e = __builtin_fma(-h, y, 0.5);
    y = __builtin_fma(y, e, y);
    h = 0.5 * s * y;
    e = __builtin_fma(-h, y, 0.5);
    return __builtin_fma(y, e, y);
}

__device__ inline fastd operator/(fastd a, fastd b) { return fastd(a.v * rcp_nr2(b.v)); }

// sqrt: Goldschmidt on the rsq seed (g -> sqrt x, h -> 1/(2 sqrt x)); sqrt(0) = 0, sqrt(<0) = NaN
__device__ inline fastd sqrt(fastd x)
{
    const double y = __builtin_amdgcn_rsq(x.v);
    double g = x.v * y, h = 0.5 * y;
    const double r0 = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r0, g);
    h = __builtin_fma(h, r0, h);
    const double d = __builtin_fma(-g, g, x.v);
    g = __builtin_fma(d, h, g);
    return fastd(x.v == 0.0 ? 0.0 : g);
}

__device__ inline fastd fabs(fastd a) { return fastd(::fabs(a.v)); }
__device__ inline fastd floor(fastd a) { return fastd(::floor(a.v)); }
__device__ inline fastd log(fastd a) { return fastd(::log(a.v)); }
__device__ inline fastd acos(fastd a) { return fastd(::acos(a.v)); }
__device__ inline fastd atan2(fastd a, fastd b) { return fastd(::atan2(a.v, b.v)); }

}  // namespace ort
