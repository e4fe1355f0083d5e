// ort_k_scatter.hip — the scattering pipeline: scatter_front_kernel (ort_scatter.h) and the continuation
// trace_queue_kernel<MODE_CONTINUE> behind it, on ORT-RNG-v2 and on the 53-bit stream.
#include "ort_launch.h"
#include "ort_scatter.h"
namespace ortk {
#define ORT_SLAUNCH(K) (hipExtLaunchKernelGGL(K, dim3(cfg.grid), dim3(64 * kScatWaves), 0, cfg.stream, cfg.ev0, cfg.ev1, 0, a), #K)
const char *launch_scatter_front(bool anysrc, bool wide, const LaunchCfg &cfg, const TraceArgs &a)
{
    if (wide) return anysrc ? ORT_SLAUNCH((scatter_front_kernel<true, true>)) : ORT_SLAUNCH((scatter_front_kernel<false, true>));
    return anysrc ? ORT_SLAUNCH((scatter_front_kernel<true, false>)) : ORT_SLAUNCH((scatter_front_kernel<false, false>));
}
const char *launch_continue(bool program, bool wide, const LaunchCfg &cfg, const TraceArgs &a)
{
    if (wide) {
        if (program) return ORT_KLAUNCH((trace_queue_kernel<MODE_CONTINUE, true, false, double, PROG_POINT_WALKED, false, RNG_WIDE>));
        return ORT_KLAUNCH((trace_queue_kernel<MODE_CONTINUE, true, false, double, PROG_GENERIC, false, RNG_WIDE>));
    }
    if (program) return ORT_KLAUNCH((trace_queue_kernel<MODE_CONTINUE, true, false, double, PROG_POINT_WALKED, false>));
    return ORT_KLAUNCH((trace_queue_kernel<MODE_CONTINUE, true, false, double, PROG_GENERIC, false>));
}
}  // namespace ortk
