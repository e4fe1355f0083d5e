// Dev micro-benchmark #3 (GPU box): does packing fp32 pay?  Issue cost of v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 (two floats per lane) against v_fma_f32 / v_mul_f32 / v_add_f32 and v_fma_f64, and of the
// per-element instructions a two-rays-per-lane kernel still needs (v_rcp_f32, v_sqrt_f32, v_cmp_f32, v_cndmask).
// Same method and units as ubench.hip / ubench2.hip (inline asm chains, s_memtime ticks).
// Build: hipcc -O3 --offload-arch=gfx950 -o build/ubench3 tools/ubench3.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

constexpr int ITERS = 512;
enum Op { FMA64, FMA32, MUL32, ADD32, PKFMA, PKMUL, PKADD, RCP32, SQRT32, CMP32, CMP64VCC, CMP64SGPR, CMP64SGPR4, NOPS };
const char *names[] = {"v_fma_f64", "v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_rcp_f32", "v_sqrt_f32",
                       "v_cmp_gt_f32", "v_cmp_f64 -> vcc", "v_cmp_f64 -> sgpr", "v_cmp_f64 -> 4 sgprs"};
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP, int CHAINS>
__global__ void k(double *out, unsigned long long *cycles, double seed)
{
    double v[CHAINS];
    float f[CHAINS];
    f2 p[CHAINS];
    unsigned long long acc = 0;
    for (int c = 0; c < CHAINS; ++c) { v[c] = seed + threadIdx.x * 1e-3 + c; f[c] = (float)v[c]; p[c] = f2{f[c], f[c] + 1.f}; }
    double y = seed * 0.999;
    float yf = 0.999f;
    f2 yp = {0.999f, 0.998f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[c]) : "v"(y));
                else if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[c]) : "v"(yf));
                else if (OP == MUL32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[c]) : "v"(yf));
                else if (OP == ADD32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(yf));
                else if (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[c]) : "v"(yp));
                else if (OP == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[c]) : "v"(yp));
                else if (OP == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(yp));
                else if (OP == RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[c]));
                else if (OP == SQRT32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[c]));
                else if (OP == CMP64VCC) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(v[c]), "v"(y) : "vcc");
                else if (OP == CMP64SGPR) asm volatile("v_cmp_gt_f64 s[40:41], %0, %1" : : "v"(v[c]), "v"(y) : "s40", "s41");
                else if (OP == CMP64SGPR4) {
                    if ((c & 3) == 0) asm volatile("v_cmp_gt_f64 s[40:41], %0, %1" : : "v"(v[c]), "v"(y) : "s40", "s41");
                    else if ((c & 3) == 1) asm volatile("v_cmp_gt_f64 s[42:43], %0, %1" : : "v"(v[c]), "v"(y) : "s42", "s43");
                    else if ((c & 3) == 2) asm volatile("v_cmp_gt_f64 s[44:45], %0, %1" : : "v"(v[c]), "v"(y) : "s44", "s45");
                    else asm volatile("v_cmp_gt_f64 s[46:47], %0, %1" : : "v"(v[c]), "v"(y) : "s46", "s47");
                }
                else if (OP == CMP32) { unsigned long long m; asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m) : "v"(f[c]), "v"(yf)); acc ^= m; }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = (double)acc;
    for (int c = 0; c < CHAINS; ++c) s += v[c] + f[c] + p[c].x + p[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, int CHAINS>
void run(int waves_per_simd)
{
    int blocks = 256, threads = 64 * 4 * waves_per_simd;
    double *out; unsigned long long *cyc;
    (void)hipMalloc(&out, blocks * threads * sizeof(double));
    (void)hipMalloc(&cyc, blocks * (threads / 64) * sizeof(unsigned long long));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.2345);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * (threads / 64));
    (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= h.size();
    double per = mean / (ITERS * 8.0 * CHAINS);
    printf("%-14s chains=%d waves/SIMD=%d : %8.3f ticks per instr per wave, %8.3f per SIMD\n", names[OP], CHAINS, waves_per_simd, per, per / waves_per_simd);
    (void)hipFree(out); (void)hipFree(cyc);
}

#define RUN_ALL(OP) run<OP, 1>(1); run<OP, 4>(4);

int main()
{
    RUN_ALL(FMA64) RUN_ALL(FMA32) RUN_ALL(MUL32) RUN_ALL(ADD32) RUN_ALL(PKFMA) RUN_ALL(PKMUL) RUN_ALL(PKADD) RUN_ALL(RCP32) RUN_ALL(SQRT32) RUN_ALL(CMP32) RUN_ALL(CMP64VCC) RUN_ALL(CMP64SGPR) RUN_ALL(CMP64SGPR4)
    return 0;
}
