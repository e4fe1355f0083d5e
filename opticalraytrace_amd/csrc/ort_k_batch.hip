// ort_k_batch.hip — MULTI-SYSTEM launches (runner.py's experiment loops, :113-261, one process per settings file
// there): trace_batch_kernel<PROG> traces one loop of every simulation of a group that shares surface program PROG in ONE
// launch (gridDim.x = simulations, gridDim.y = workgroups of each; ort_trace.h), trace_batch_rerun_kernel closes the deferral lists of all of them in one
// more.  Exact fp64, default emitters and ORT-RNG-v2 — the arithmetic and draws of trace_queue_kernel<MODE_FUSED, ..., PROG>,
// the same body; the image source (one table per context) and everything that is not a surface program go one by one.
#include "ort_launch.h"
namespace ortk {

// the literal re-run of every simulation of a batch: simulation blockIdx.y's listed rays (normally none: the workgroups
// read a zero count and return)
__global__ __launch_bounds__(kBlock, ORT_MIN_WAVES) void trace_batch_rerun_kernel(const TraceArgs *batch)
{
    TraceArgs a;
    load_batch_args(a, batch, blockIdx.y);
    a.listed = 1;
    trace_body<MODE_FUSED, false, double, true, true>(a);
}

#define ORT_BATCH_PROGRAMS(X) ORT_PROGRAMS(X) ORT_RING_LISTS(X, SRC_CRS) ORT_RING_LISTS(X, SRC_ISORS)

bool batch_has_program(int prog)
{
    switch (prog) {
#define ORT_CASE(P) case P:
        ORT_BATCH_PROGRAMS(ORT_CASE)
#undef ORT_CASE
        return true;
    default:
        return false;
    }
}

const char *launch_batch(int prog, const LaunchCfg &cfg, int n_sys, const TraceArgs *d_batch)
{
#define ORT_CASE(P)                                                                                                                        \
    case P:                                                                                                                                \
        hipExtLaunchKernelGGL((trace_batch_kernel<P>), dim3(n_sys, cfg.grid), dim3(kBlock), 0, cfg.stream, cfg.ev0, cfg.ev1, 0, d_batch);  \
        return "trace_batch_kernel<" #P ">";
    switch (prog) {
        ORT_BATCH_PROGRAMS(ORT_CASE)
    default:
        return nullptr;
    }
#undef ORT_CASE
}

const char *launch_batch_rerun(const LaunchCfg &cfg, int n_sys, const TraceArgs *d_batch)
{
    hipExtLaunchKernelGGL(trace_batch_rerun_kernel, dim3(cfg.grid, n_sys), dim3(kBlock), 0, cfg.stream, cfg.ev0, cfg.ev1, 0, d_batch);
    return "trace_batch_rerun_kernel";
}

}  // namespace ortk
