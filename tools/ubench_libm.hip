// Dev tool (GPU box): cost of the glibc-exact device functions (csrc/ort_libm.h, predicated forms) next to the
// device library's and the tracer's own sincos, in a saturated loop: ns per call per lane-wave and relative cost.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o build/ubench_libm tools/ubench_libm.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../include/ort.h"
#include "../opticalraytrace_amd/csrc/ort_device.h"

__device__ inline double u01(uint64_t &s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(s >> 11) * 0x1p-53;
}
template <int F> __global__ void bench(double *out, int iters)
{
    __shared__ uint64_t LT[F >= 20 ? ort::glibc::kLdsTableWords : 1];
    if (F >= 20) { ort::glibc::stage_tables(LT, threadIdx.x, blockDim.x); __syncthreads(); }
    const ort::glibc::TabLds TL = {(const __attribute__((address_space(3))) uint64_t *)LT};
    uint64_t s = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    double acc = 0.;
    for (int i = 0; i < iters; ++i) {
        const double u = u01(s);
        if (F == 0) acc += u;                                                     // the loop itself
        if (F == 1) { const ort::glibc::SinCos r = ort::glibc::sincos_p<false>(6.283185307179586 * u); acc += r.s + r.c; }
        if (F == 2) { const ort::SinCosT<double> r = ort::sincos_small(6.283185307179586 * u); acc += r.s + r.c; }
        if (F == 3) { double sn, cs; sincos(6.283185307179586 * u, &sn, &cs); acc += sn + cs; }
        if (F == 4) acc += ort::glibc::atan2_p(2. * u - 1., 2. * u01(s) - 1.);
        if (F == 5) acc += atan2(2. * u - 1., 2. * u01(s) - 1.);
        if (F == 6) acc += ort::glibc::acos_p(2. * u - 1.);
        if (F == 7) acc += acos(2. * u - 1.);
        if (F == 8) acc += ort::glibc::log_p(u);
        if (F == 9) acc += log(u);
        if (F == 10) { const ort::glibc::SinCos r = ort::glibc::sincos_p<true>(6.283185307179586 * u); acc += r.s + r.c; }
        if (F == 11) { acc += (2. * u - 1.) / (u01(s) + 0.5); }                   // one IEEE division (+ a draw)
        if (F == 12) { acc += sqrt(u); }
        if (F == 21) { const ort::glibc::SinCos r = ort::glibc::sincos_p<false>(TL, 6.283185307179586 * u); acc += r.s + r.c; }
        if (F == 24) acc += ort::glibc::atan2_p(TL, 2. * u - 1., 2. * u01(s) - 1.);
        if (F == 26) acc += ort::glibc::acos_p(TL, 2. * u - 1.);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int F> double run(double *d, int iters)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    bench<F><<<256 * 8, 256>>>(d, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    bench<F><<<256 * 8, 256>>>(d, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main()
{
    double *d;
    (void)hipMalloc(&d, 256 * 8 * 256 * 8);
    const int iters = 2000;
    const char *names[] = {"loop", "glibc sincos_p", "own sincos_small", "ocml sincos", "glibc atan2_p", "ocml atan2", "glibc acos_p", "ocml acos",
                           "glibc log_p", "ocml log", "glibc sin,cos pair", "fp64 division", "fp64 sqrt"};
    double t[13];
    t[0] = run<0>(d, iters); t[1] = run<1>(d, iters); t[2] = run<2>(d, iters); t[3] = run<3>(d, iters); t[4] = run<4>(d, iters);
    t[5] = run<5>(d, iters); t[6] = run<6>(d, iters); t[7] = run<7>(d, iters); t[8] = run<8>(d, iters); t[9] = run<9>(d, iters);
    t[10] = run<10>(d, iters); t[11] = run<11>(d, iters); t[12] = run<12>(d, iters);
    // 2048 workgroups x 4 waves = 8192 waves on 1024 SIMDs: 8 waves per SIMD share its issue port
    const double l1 = run<21>(d, iters), l4 = run<24>(d, iters), l6 = run<26>(d, iters);
    printf("tables in LDS: sincos_p %.1f  atan2_p %.1f  acos_p %.1f  (net of the loop)\n", (l1 - t[0]) * 1e6 / iters / 8, (l4 - t[0]) * 1e6 / iters / 8, (l6 - t[0]) * 1e6 / iters / 8);
    for (int i = 0; i < 13; ++i)
        printf("%-20s %8.3f ms   %7.1f ns per call per SIMD-slot (net of the loop: %6.1f)\n", names[i], t[i], t[i] * 1e6 / iters / 8, (t[i] - t[0]) * 1e6 / iters / 8);
    return 0;
}
