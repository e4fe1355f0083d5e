// ort_launch.h — what the host side (ort_hip.hip) sees of the kernel families: one launcher per translation unit that
// instantiates kernels (ort_k_*.hip).  Splitting the instantiations keeps a full build at the time of its largest unit
// (make -j) and an A/B of one family at one unit.  Every launcher puts ONE kernel on `stream` (hipExtLaunchKernelGGL
// with the caller's event pair, either of which may be null) and returns the instantiation's name as the source spells
// it — what ort_last_kernel_name() hands to the tests and to bench.py — or nullptr when the family holds no kernel for
// the request (the caller then takes the next, more general family).
#pragma once
#include "ort_trace.h"

namespace ortk {

struct LaunchCfg {
    int grid;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
};

// surface programs (trace_queue_kernel<MODE, true, false, T, PROG, false, RNG>); mode = MODE_FUSED / MODE_RESIDENT
const char *launch_program_f64(int prog, int mode, const LaunchCfg &cfg, const TraceArgs &a);                 // ort_k_prog64.hip
const char *launch_program_f64_strict(int prog, int mode, const LaunchCfg &cfg, const TraceArgs &a);          // ort_k_strict.hip   RNG_STRICT
const char *launch_program_f64_wide(int prog, int mode, bool strict, const LaunchCfg &cfg, const TraceArgs &a);   // ort_k_wide.hip     RNG_WIDE (| RNG_STRICT)
const char *launch_program_f32(int prog, int mode, const LaunchCfg &cfg, const TraceArgs &a);                 // ort_k_prog32.hip
const char *launch_program_fast(int prog, int mode, const LaunchCfg &cfg, const TraceArgs &a);                // ort_k_fast.hip
// everything that walks a list it does not know at compile time: the generic queued kernels and the lockstep kernel
// (trace_kernel: the literal re-run, the parity / debug entry, variant bit 0 clear), per arithmetic
struct GenericReq {
    int precision;       // 0 exact fp64, 1 fp32, 2 fast fp64
    int mode;            // MODE_FUSED / MODE_RESIDENT / MODE_DEBUG
    bool queued, filt, anysrc, scat, wide;
};
const char *launch_generic(const GenericReq &q, const LaunchCfg &cfg, const TraceArgs &a);                     // ort_k_generic.hip (fp64), ort_k_prog32.hip, ort_k_fast.hip
const char *launch_generic_f64(const GenericReq &q, const LaunchCfg &cfg, const TraceArgs &a);
const char *launch_generic_f32(const GenericReq &q, const LaunchCfg &cfg, const TraceArgs &a);
const char *launch_generic_fast(const GenericReq &q, const LaunchCfg &cfg, const TraceArgs &a);
// the scattering pipeline (ort_k_scatter.hip): the front kernel on cfg.grid workgroups of 64 kScatWaves threads, and the
// continuation (trace_queue_kernel<MODE_CONTINUE>; program = the list behind the bottle is PROG_POINT's)
const char *launch_scatter_front(bool anysrc, bool wide, const LaunchCfg &cfg, const TraceArgs &a);
const char *launch_continue(bool program, bool wide, const LaunchCfg &cfg, const TraceArgs &a);

// multi-system launches (ort_k_batch.hip): `d_batch` = n_sys TraceArgs on the device, one per simulation; cfg.grid = workgroups
// per simulation.  batch_has_program: does the unit hold trace_batch_kernel<prog>?
bool batch_has_program(int prog);
const char *launch_batch(int prog, const LaunchCfg &cfg, int n_sys, const TraceArgs *d_batch);
const char *launch_batch_rerun(const LaunchCfg &cfg, int n_sys, const TraceArgs *d_batch);

// development experiments on the fused point program (ort_k_exp.hip; ort_debug_set_exp)
const char *launch_exp(int which, const LaunchCfg &cfg, const TraceArgs &a);

#define ORT_KLAUNCH(K) (hipExtLaunchKernelGGL(K, dim3(cfg.grid), dim3(kBlock), 0, cfg.stream, cfg.ev0, cfg.ev1, 0, a), #K)

}  // namespace ortk
