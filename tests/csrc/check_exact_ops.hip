// Test program (GPU): the hand-expanded fp64 operations of csrc/ort_device.h against the
// compiler's IEEE operations, bit for bit, over random operands of every exponent.
//   sqrt_f<true>(x)        vs  sqrt(x)            where the lane's `rare` flag stays clear
//   div3_shared(v, t)      vs  v.x / t, v.y / t, v.z / t   where `shared` comes back true
//   vnormalise_f<true>(v)  vs  v / sqrt(v.v)      where `rare` stays clear
// and reports how many operands raised the flag (they take the literal path in the tracer).
// Prints "mismatches <n>" per operation; exit code 0 iff all are 0.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../include/ort.h"
#include "../../opticalraytrace_amd/csrc/ort_device.h"

__device__ inline uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// mode 0: any bit pattern (all exponents, NaN, inf, subnormal); mode 1: magnitudes the tracer
// meets (1e-12 .. 1e3, both signs, exact zeros in x); mode 2: exponents -340 .. 340
__device__ inline double operand(uint64_t key, int mode)
{
    const uint64_t b = mix(key);
    if (mode == 0) return __longlong_as_double((long long)b);
    if (mode == 2) {                                         // 2^-340 .. 2^340: straddles the guards' thresholds
        const double mm = 1.0 + (double)(b >> 12) * 0x1p-52;
        const int ee = (int)((b >> 1) & 1023) % 681 - 340;
        return (b & 1) ? -ldexp(mm, ee) : ldexp(mm, ee);
    }
    const double m = 1.0 + (double)(b >> 12) * 0x1p-52;
    const int e = (int)((b >> 4) & 63) - 40;                 // 2^-40 .. 2^23
    double v = ldexp(m, e);
    if ((b & 15) == 0 && (key & 3) == 0) v = 0.0;        // exact zeros: the x operand only
    return (b & 8) ? -v : v;
}

__device__ inline bool same(double a, double b)
{
    return __double_as_longlong(a) == __double_as_longlong(b) || (a != a && b != b);
}

__global__ void check(uint64_t n, int mode, unsigned long long *bad)
{
    unsigned long long bs = 0, bd = 0, bn = 0, rs = 0, rd = 0, rn = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const double x = operand(4 * i + 0, mode), y = operand(4 * i + 1, mode);
        const double z = operand(4 * i + 2, mode), t = operand(4 * i + 3, mode);
        bool rare = false;
        const double sq = ort::sqrt_f<true, double>(x, true, rare);
        if (rare) rs++;
        else if (!same(sq, sqrt(x))) bs++;
        bool shared;
        const ort::Vec q = ort::div3_shared(ort::Vec{x, y, z}, t, shared);
        if (!shared) rd++;
        else if (!same(q.x, x / t) || !same(q.y, y / t) || !same(q.z, z / t)) bd++;
        // a normalisation as the tracer does it: t = |v|; an exactly-zero x is announced by the
        // caller (cylinder normals), any other exact zero must raise the flag
        rare = false;
        const ort::Vec u = ort::vnormalise_f<true, double>(ort::Vec{x, y, z}, true, rare, x == 0.0);
        const double len = sqrt(x * x + y * y + z * z);
        if (rare) rn++;
        else if (!same(u.x, x / len) || !same(u.y, y / len) || !same(u.z, z / len)) bn++;
        // the self-contained forms (emitters): never flagged, always exact
        const ort::Vec w = ort::div3(ort::Vec{x, y, z}, t);
        if (!same(w.x, x / t) || !same(w.y, y / t) || !same(w.z, z / t)) bd++;
    }
    atomicAdd(&bad[0], bs); atomicAdd(&bad[1], bd); atomicAdd(&bad[2], bn);
    atomicAdd(&bad[3], rs); atomicAdd(&bad[4], rd); atomicAdd(&bad[5], rn);
}

// Accuracy of the decision-only approximations (ort_device.h: rcp_approx, rsq_approx) and of the
// hardware seeds under them: max relative error over 2^26 operands in [2^-20, 2^20].
__global__ void approx_err(uint64_t n, double *out)
{
    double e_rcp = 0, e_rsq = 0, e_seed_rcp = 0, e_seed_rsq = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = mix(i + 12345);
        const double x = ldexp(1.0 + (double)(b >> 12) * 0x1p-52, (int)((b >> 4) & 63) % 41 - 20);
        const double r = 1.0 / x, q = 1.0 / sqrt(x);
        e_rcp = fmax(e_rcp, fabs(ort::rcp_approx(x) - r) / r);
        e_rsq = fmax(e_rsq, fabs(ort::rsq_approx(x) - q) / q);
        e_seed_rcp = fmax(e_seed_rcp, fabs(__builtin_amdgcn_rcp(x) - r) / r);
        e_seed_rsq = fmax(e_seed_rsq, fabs(__builtin_amdgcn_rsq(x) - q) / q);
    }
    // block-free reduction: one slot per thread, maxima taken on the host
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    out[4 * t + 0] = e_rcp; out[4 * t + 1] = e_rsq; out[4 * t + 2] = e_seed_rcp; out[4 * t + 3] = e_seed_rsq;
}

// Round-2 forms.  (a) vnormalise_est: vectors of length t0 (1 + x), |x| up to 2^-30 (the guard is
// 2^-32 on the square: about half of the larger ones must be flagged, none may mismatch), t0 over
// the tracer's radii (mode 1) or 2^-200 .. 2^200 (mode 2; mode 0: 2^-20 .. 2^20 with x of a few
// ulps).  (b) div_plain inside its guard.  (c) solve_and_pick<filtered> against the literal
// solveQuadratic + root choice wherever it does not flag.
__global__ void check2(uint64_t n, int mode, unsigned long long *bad)
{
    unsigned long long bn = 0, rn = 0, bd = 0, nd = 0, bq = 0, rq = 0, hq = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = mix(8 * i + 5);
        // (a)
        const int et = mode == 2 ? (int)(b & 511) % 401 - 200 : (mode == 1 ? (int)(b & 15) - 12 : (int)(b & 63) % 41 - 20);
        const double t0 = ldexp(1.0 + (double)(mix(8 * i + 6) >> 12) * 0x1p-52, et);
        double dx = operand(8 * i + 0, 1), dy = operand(8 * i + 1, 1), dz = operand(8 * i + 2, 1);
        const bool cyl = (b >> 20) & 1;
        if (cyl) dx = 0.0;
        const double len = sqrt(dx * dx + dy * dy + dz * dz);
        const uint64_t bx = mix(8 * i + 7);
        const double mag = mode == 0 ? ldexp((double)(bx & 255), -52) : ldexp((double)(bx >> 12) * 0x1p-52, -30 - (int)(bx & 31));
        const double x = (bx & 2048) ? mag : -mag;
        const double sc = t0 * (1.0 + x) / len;
        const ort::Vec v = {dx * sc, dy * sc, dz * sc};
        bool rare = false;
        const ort::Vec u = ort::vnormalise_est<true, double>(v, t0, 0.5 / t0, (0.5 / t0) / (2.0 * (t0 * t0)), 0x1p-32 * (t0 * t0), true, rare, cyl);
        const double l2 = sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
        if (rare || !(len > 0.0)) rn++;
        else if (!same(u.x, v.x / l2) || !same(u.y, v.y / l2) || !same(u.z, v.z / l2)) bn++;
        // (b)
        const double p = operand(8 * i + 3, mode == 1 ? 1 : 2), q = operand(8 * i + 4, mode == 1 ? 1 : 2);
        if (fabs(p) > 0x1p-300 && fabs(p) < 0x1p300 && fabs(q) > 0x1p-300 && fabs(q) < 0x1p300) {
            nd++;
            if (!same(ort::div_plain(p, q), p / q)) bd++;
        }
        // (c) a ray against a sphere: a = d.d, hb = d.L, c = L.L - R^2 with operands of the mode
        {
            const int m = mode == 1 ? 1 : mode;
            const double a = mode == 1 ? 1.0 + ldexp((double)(bx >> 40), -60) - 0x1p-37 : fabs(operand(8 * i + 3, m));
            const double hb = operand(8 * i + 4, m), c = operand(8 * i + 2, m);
            double tf, tl;
            bool hf, hl, rf = false, rl = false;
            ort::solve_and_pick<true, double>(a, hb, c, true, tf, hf, rf);
            ort::solve_and_pick<false, double>(a, hb, c, true, tl, hl, rl);
            if (rf) rq++;
            else if (hf != hl || (hl && !same(tf, tl))) bq++;
            else if (hl) hq++;
        }
    }
    atomicAdd(&bad[0], bn); atomicAdd(&bad[1], rn); atomicAdd(&bad[2], bd); atomicAdd(&bad[3], nd);
    atomicAdd(&bad[4], bq); atomicAdd(&bad[5], rq); atomicAdd(&bad[6], hq);
}

int main()
{
    {
        unsigned long long *d2, h2[7];
        if (hipMalloc(&d2, sizeof(h2)) != hipSuccess) { printf("no device\n"); return 2; }
        for (int mode = 0; mode < 3; ++mode) {
            (void)hipMemset(d2, 0, sizeof(h2));
            const uint64_t n = 1ull << 28;
            check2<<<4096, 256>>>(n, mode, d2);
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
            (void)hipMemcpy(h2, d2, sizeof(h2), hipMemcpyDeviceToHost);
            printf("est  %d operands %llu : mismatches normalise_est %llu div_plain %llu quadratic %llu ; flagged normalise_est %llu "
                   "quadratic %llu ; compared div_plain %llu quadratic hits %llu\n",
                   mode, (unsigned long long)n, h2[0], h2[2], h2[4], h2[1], h2[5], h2[3], h2[6]);
            if (h2[0] || h2[2] || h2[4]) return 1;
            // the checks must not be vacuous: most operand sets are compared, not flagged
            if (h2[1] > n / 2 || h2[3] < n / 4 || (mode == 1 && h2[6] < n / 8)) { printf("vacuous\n"); return 1; }
        }
        (void)hipFree(d2);
    }
    unsigned long long *d_bad, h[6];
    if (hipMalloc(&d_bad, sizeof(h)) != hipSuccess) { printf("no device\n"); return 2; }
    int rc = 0;
    for (int mode = 0; mode < 3; ++mode) {
        (void)hipMemset(d_bad, 0, sizeof(h));
        const uint64_t n = 1ull << 28;
        check<<<4096, 256>>>(n, mode, d_bad);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
        (void)hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d operands %llu : mismatches sqrt %llu div3 %llu normalise %llu ; flagged sqrt %llu div3 %llu normalise %llu\n",
               mode, (unsigned long long)n, h[0], h[1], h[2], h[3], h[4], h[5]);
        if (h[0] || h[1] || h[2]) rc = 1;
        // the tracer's magnitudes must essentially never be flagged (zeros are: sqrt(0), |v| = 0)
        if (mode == 1 && (h[4] > n / 1000 || h[5] > n / 1000)) rc = 1;
    }
    (void)hipFree(d_bad);
    {
        const int blocks = 1024, threads = 256;
        double *d_e;
        std::vector<double> h_e((size_t)4 * blocks * threads);
        if (hipMalloc(&d_e, h_e.size() * sizeof(double)) != hipSuccess) return 2;
        approx_err<<<blocks, threads>>>(1ull << 26, d_e);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
        (void)hipMemcpy(h_e.data(), d_e, h_e.size() * sizeof(double), hipMemcpyDeviceToHost);
        double m[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < h_e.size(); ++i) m[i & 3] = m[i & 3] > h_e[i] ? m[i & 3] : h_e[i];
        printf("max relative error: rcp_approx %.3g rsq_approx %.3g (bound 2^-40 = 9.09e-13); seeds v_rcp_f64 %.3g v_rsq_f64 %.3g\n",
               m[0], m[1], m[2], m[3]);
        if (!(m[0] < 0x1p-40) || !(m[1] < 0x1p-40)) rc = 1;
        (void)hipFree(d_e);
    }
    return rc;
}
