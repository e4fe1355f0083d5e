// Dev micro-benchmark (GPU box): issue cost and dependent latency of the fp64 / int
// instructions that dominate the trace kernel, measured with s_memtime on one wave
// per SIMD and on four waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 ubench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 512;

enum Op { FMA, ADD, MUL, RCP, RSQ, SQRTHW, DIVSCALE, DIVFMAS, DIVFIXUP, LDEXP, MULLO, MADU64, FULLDIV, FULLSQRT, CVT, NOPS };
const char *names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64(hw)",
                       "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_ldexp_f64", "v_mul_lo_u32",
                       "v_mad_u64_u32", "a/b (IEEE f64)", "sqrt (IEEE f64)", "v_cvt_f64_u32"};

template <int OP>
__device__ inline double op1(double x, double y)
{
    if (OP == FMA) return __builtin_fma(x, y, y);
    if (OP == ADD) return x + y;
    if (OP == MUL) return x * y;
    if (OP == RCP) return __builtin_amdgcn_rcp(x);
    if (OP == RSQ) return __builtin_amdgcn_rsq(x);
    if (OP == SQRTHW) return __builtin_amdgcn_sqrt(x);
    if (OP == DIVSCALE) { bool f; return __builtin_amdgcn_div_scale(x, y, true, &f); }
    if (OP == DIVFMAS) return __builtin_amdgcn_div_fmas(x, y, y, false);
    if (OP == DIVFIXUP) return __builtin_amdgcn_div_fixup(x, y, y);
    if (OP == LDEXP) return __builtin_ldexp(x, 1);
    if (OP == FULLDIV) return y / x;
    if (OP == FULLSQRT) return sqrt(x);
    return x;
}

template <int OP, int CHAINS>
__global__ void k(double *out, unsigned long long *cycles, double seed)
{
    double v[CHAINS];
    unsigned u[CHAINS];
    unsigned long long w[CHAINS];
    for (int c = 0; c < CHAINS; ++c) { v[c] = seed + threadIdx.x * 1e-3 + c; u[c] = threadIdx.x + c + 3; w[c] = u[c]; }
    double y = seed * 0.999;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == MULLO) u[c] = u[c] * 2654435761u + 1u;     // mul_lo + add
                else if (OP == MADU64) w[c] = (unsigned long long)(unsigned)w[c] * 2654435761ull + w[c];
                else if (OP == CVT) v[c] += (double)(u[c] += 7u);
                else v[c] = op1<OP>(v[c], y);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += v[c] + u[c] + (double)w[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, int CHAINS>
int run(const char *name, int waves_per_simd)
{
    int blocks = 256;                         // one block per CU
    int threads = 64 * 4 * waves_per_simd;    // waves spread over the 4 SIMDs
    double *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, blocks * threads * sizeof(double)));
    CHECK(hipMalloc(&cyc, blocks * (threads / 64) * sizeof(unsigned long long)));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.2345);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * (threads / 64));
    CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= h.size();
    double per_wave_instr = mean / (ITERS * 8.0 * CHAINS);                 // s_memtime ticks (100 MHz?) per instr per wave
    printf("%-18s chains=%d waves/SIMD=%d : %8.3f ticks per instr per wave, %8.3f ticks per instr per SIMD\n", name, CHAINS,
           waves_per_simd, per_wave_instr, per_wave_instr / waves_per_simd);
    hipFree(out); hipFree(cyc);
    return 0;
}

#define RUN_ALL(OP) \
    run<OP, 1>(names[OP], 1); run<OP, 4>(names[OP], 1); run<OP, 1>(names[OP], 4); run<OP, 4>(names[OP], 4);

int main()
{
    printf("ticks are s_memtime units (shader clock cycles)\n");
    RUN_ALL(FMA) RUN_ALL(ADD) RUN_ALL(MUL) RUN_ALL(RCP) RUN_ALL(RSQ) RUN_ALL(SQRTHW) RUN_ALL(DIVSCALE)
    RUN_ALL(DIVFMAS) RUN_ALL(DIVFIXUP) RUN_ALL(LDEXP) RUN_ALL(MULLO) RUN_ALL(MADU64) RUN_ALL(CVT)
    RUN_ALL(FULLDIV) RUN_ALL(FULLSQRT)
    return 0;
}
