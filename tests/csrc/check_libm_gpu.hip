// check_libm_gpu.hip — the DEVICE code of csrc/ort_libm.h (glibc 2.35's sin / cos / sincos / log / atan2 / acos
// restated) against the host's libm, bit for bit: the arguments of libm_args.h are evaluated on the GPU and by the
// host's libm.  check_libm_gpu [n_per_function = 30000000] [seed]; exit status 1 on any mismatch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../opticalraytrace_amd/csrc/ort_libm.h"
#include "libm_args.h"

using namespace libm_args;

// PRED: the predicated forms the kernels call (sincos_p<false> = sincos, sincos_p<true> = the pair sin, cos); else the
// straight ones
template <bool PRED>
__global__ void eval1(const double *x, size_t n, double *o_sin, double *o_cos, double *o_scs, double *o_scc)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (PRED) {
            const ort::glibc::SinCos q = ort::glibc::sincos_p<true>(x[i]), r = ort::glibc::sincos_p<false>(x[i]);
            o_sin[i] = q.s; o_cos[i] = q.c; o_scs[i] = r.s; o_scc[i] = r.c;
        } else {
            o_sin[i] = ort::glibc::sin(x[i]);
            o_cos[i] = ort::glibc::cos(x[i]);
            const ort::glibc::SinCos r = ort::glibc::sincos(x[i]);
            o_scs[i] = r.s; o_scc[i] = r.c;
        }
    }
}
template <bool PRED> __global__ void eval_log(const double *x, size_t n, double *o)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        o[i] = PRED ? ort::glibc::log_p(x[i]) : ort::glibc::log(x[i]);
}
template <bool PRED> __global__ void eval_acos(const double *x, size_t n, double *o)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        o[i] = PRED ? ort::glibc::acos_p(x[i]) : ort::glibc::acos(x[i]);
}
template <bool PRED> __global__ void eval_atan2(const double *y, const double *x, size_t n, double *o)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        o[i] = PRED ? ort::glibc::atan2_p(y[i], x[i]) : ort::glibc::atan2(y[i], x[i]);
}

#define HIP_OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); return 2; } } while (0)

struct DevBuf {
    double *p = nullptr;
    size_t cap = 0;
    int fit(size_t n) { if (n <= cap) return 0; if (p) (void)hipFree(p); cap = n; return hipMalloc(&p, n * sizeof(double)) == hipSuccess ? 0 : 1; }
};

int main(int argc, char **argv)
{
    const long long n = argc > 1 ? atoll(argv[1]) : 30000000ll;
    Rng rng(argc > 2 ? strtoull(argv[2], 0, 0) : 0x243F6A8885A308D3ull);
    Tally straight[6] = {{"sin"}, {"cos"}, {"sincos"}, {"log"}, {"atan2"}, {"acos"}};
    Tally pred[6] = {{"sin_p"}, {"cos_p"}, {"sincos_p"}, {"log_p"}, {"atan2_p"}, {"acos_p"}};
    DevBuf in0, in1, o0, o1, o2, o3;
    std::vector<double> h0, h1, h2, h3;
    const long long chunk = 1 << 22;
    for (long long done = 0; done < n; done += chunk) {
        const long long m = n - done < chunk ? n - done : chunk;
        const bool PRED = ((done / chunk) & 1) != 0;              // chunks alternate between the two forms
        Tally *T = PRED ? pred : straight;
        Tally &tsin = T[0], &tcos = T[1], &tsc = T[2], &tlog = T[3], &tat = T[4], &tac = T[5];
        {
            const std::vector<double> a = angles(m, rng);
            const size_t k = a.size();
            if (in0.fit(k) || o0.fit(k) || o1.fit(k) || o2.fit(k) || o3.fit(k)) return 2;
            HIP_OK(hipMemcpy(in0.p, a.data(), k * 8, hipMemcpyHostToDevice));
            if (PRED) eval1<true><<<2048, 256>>>(in0.p, k, o0.p, o1.p, o2.p, o3.p); else eval1<false><<<2048, 256>>>(in0.p, k, o0.p, o1.p, o2.p, o3.p);
            HIP_OK(hipGetLastError());
            h0.resize(k); h1.resize(k); h2.resize(k); h3.resize(k);
            HIP_OK(hipMemcpy(h0.data(), o0.p, k * 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(h1.data(), o1.p, k * 8, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(h2.data(), o2.p, k * 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(h3.data(), o3.p, k * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < k; ++i) {
                tsin.add(h0[i], libm_sin(a[i]), a[i]); tcos.add(h1[i], libm_cos(a[i]), a[i]);
                double ws, wc;
                libm_sincos(a[i], &ws, &wc);
                tsc.add(h2[i], ws, a[i]); tsc.add(h3[i], wc, a[i]);
            }
        }
        {
            const std::vector<double> a = logs(m, rng);
            const size_t k = a.size();
            if (in0.fit(k) || o0.fit(k)) return 2;
            HIP_OK(hipMemcpy(in0.p, a.data(), k * 8, hipMemcpyHostToDevice));
            if (PRED) eval_log<true><<<2048, 256>>>(in0.p, k, o0.p); else eval_log<false><<<2048, 256>>>(in0.p, k, o0.p);
            HIP_OK(hipGetLastError());
            h0.resize(k);
            HIP_OK(hipMemcpy(h0.data(), o0.p, k * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < k; ++i) tlog.add(h0[i], libm_log(a[i]), a[i]);
        }
        {
            const std::vector<double> a = acoss(m, rng);
            const size_t k = a.size();
            if (in0.fit(k) || o0.fit(k)) return 2;
            HIP_OK(hipMemcpy(in0.p, a.data(), k * 8, hipMemcpyHostToDevice));
            if (PRED) eval_acos<true><<<2048, 256>>>(in0.p, k, o0.p); else eval_acos<false><<<2048, 256>>>(in0.p, k, o0.p);
            HIP_OK(hipGetLastError());
            h0.resize(k);
            HIP_OK(hipMemcpy(h0.data(), o0.p, k * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < k; ++i) tac.add(h0[i], libm_acos(a[i]), a[i]);
        }
        {
            std::vector<double> ys, xs;
            atan2s(m, rng, ys, xs);
            const size_t k = ys.size();
            if (in0.fit(k) || in1.fit(k) || o0.fit(k)) return 2;
            HIP_OK(hipMemcpy(in0.p, ys.data(), k * 8, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(in1.p, xs.data(), k * 8, hipMemcpyHostToDevice));
            if (PRED) eval_atan2<true><<<2048, 256>>>(in0.p, in1.p, k, o0.p); else eval_atan2<false><<<2048, 256>>>(in0.p, in1.p, k, o0.p);
            HIP_OK(hipGetLastError());
            h0.resize(k);
            HIP_OK(hipMemcpy(h0.data(), o0.p, k * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < k; ++i) tat.add(h0[i], libm_atan2(ys[i], xs[i]), ys[i], xs[i]);
        }
    }
    int rc = 0;
    for (int j = 0; j < 6; ++j) rc |= straight[j].report() | pred[j].report();
    return rc;
}
