// Dev micro-benchmark #2 (GPU box): issue cost of the cheap-looking instructions of the trace
// kernel — compares, selects, 32-bit integer ops, 64-bit shifts — pinned with inline asm so the
// compiler cannot fold the dependent chains.  Same method and units as ubench.hip.
// Build: hipcc -O3 --offload-arch=gfx950 -o build/ubench2 tools/ubench2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

constexpr int ITERS = 512;
enum Op { FMA, CMPVCC, CNDSGPR, CMPSEL, CMPSELNOP, CNDMASK, XOR, ADDU32, MULLO, MULHI, MADU64, CMPF64, CMPU32, MAXF64, FLOORF64, LSHR64, CMPCLASS, CVTF64U32, FMA32, NOPS };
const char *names[] = {"v_fma_f64", "v_cmp_gt_f64 vcc", "v_cndmask(sgpr mask)", "cmp+cndmask dep", "cmp+4 fma+cndmask", "v_cndmask_b32", "v_xor_b32", "v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32",
                       "v_mad_u64_u32", "v_cmp_gt_f64", "v_cmp_gt_u32", "v_max_f64", "v_floor_f64", "v_lshrrev_b64",
                       "v_cmp_class_f64", "v_cvt_f64_u32", "v_fma_f32"};

template <int OP, int CHAINS>
__global__ void k(double *out, unsigned long long *cycles, double seed)
{
    double v[CHAINS];
    unsigned u[CHAINS];
    float f[CHAINS];
    unsigned long long w[CHAINS];
    unsigned long long acc = 0;
    for (int c = 0; c < CHAINS; ++c) { v[c] = seed + threadIdx.x * 1e-3 + c; u[c] = threadIdx.x + c + 3; w[c] = u[c]; f[c] = (float)v[c]; }
    double y = seed * 0.999;
    unsigned kk = 2654435761u + threadIdx.x;
    unsigned long long smask = 0x5555555555555555ull ^ (unsigned long long)__builtin_amdgcn_readfirstlane((int)seed);
    float yf = 0.999f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == FMA) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[c]) : "v"(y));
                else if (OP == CMPVCC) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(v[c]), "v"(y) : "vcc");
                else if (OP == CNDSGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[40:41]" : "+v"(u[c]) : "v"(kk));
                else if (OP == CMPSEL) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(u[c]) : "v"(v[c]), "v"(y), "v"(kk) : "vcc");
                else if (OP == CMPSELNOP) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n v_fma_f64 %1, %1, %2, %2\n v_fma_f64 %1, %1, %2, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(u[c]), "+v"(v[c]) : "v"(y), "v"(kk) : "vcc");
                else if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(kk));
                else if (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[c]) : "v"(kk));
                else if (OP == ADDU32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(kk));
                else if (OP == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[c]) : "v"(kk));
                else if (OP == MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[c]) : "v"(kk));
                else if (OP == MADU64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[c]) : "v"(kk), "v"(u[c]) : "vcc");
                else if (OP == CMPF64) { unsigned long long m; asm volatile("v_cmp_gt_f64 %0, %1, %2" : "=s"(m) : "v"(v[c]), "v"(y)); acc ^= m; }
                else if (OP == CMPU32) { unsigned long long m; asm volatile("v_cmp_gt_u32 %0, %1, %2" : "=s"(m) : "v"(u[c]), "v"(kk)); acc ^= m; }
                else if (OP == CMPCLASS) { unsigned long long m; asm volatile("v_cmp_class_f64 %0, %1, %2" : "=s"(m) : "v"(v[c]), "v"(kk)); acc ^= m; }
                else if (OP == MAXF64) asm volatile("v_max_f64 %0, %0, %1" : "+v"(v[c]) : "v"(y));
                else if (OP == FLOORF64) asm volatile("v_floor_f64 %0, %0" : "+v"(v[c]));
                else if (OP == LSHR64) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(w[c]));
                else if (OP == CVTF64U32) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(v[c]) : "v"(u[c]));
                else if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[c]) : "v"(yf));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = (double)acc;
    for (int c = 0; c < CHAINS; ++c) s += v[c] + u[c] + (double)w[c] + f[c];
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
    printf("%-18s chains=%d waves/SIMD=%d : %8.3f ticks per instr per wave, %8.3f per SIMD\n", names[OP], CHAINS, waves_per_simd, per, per / waves_per_simd);
    (void)hipFree(out); (void)hipFree(cyc);
}

#define RUN_ALL(OP) run<OP, 1>(1); run<OP, 4>(4);

int main()
{
    RUN_ALL(FMA) RUN_ALL(CMPVCC) RUN_ALL(CNDSGPR) RUN_ALL(CMPSEL) RUN_ALL(CMPSELNOP) RUN_ALL(FMA32) RUN_ALL(CNDMASK) RUN_ALL(XOR) RUN_ALL(ADDU32) RUN_ALL(MULLO) RUN_ALL(MULHI) RUN_ALL(MADU64)
    RUN_ALL(CMPF64) RUN_ALL(CMPU32) RUN_ALL(CMPCLASS) RUN_ALL(MAXF64) RUN_ALL(FLOORF64) RUN_ALL(LSHR64) RUN_ALL(CVTF64U32)
    return 0;
}
