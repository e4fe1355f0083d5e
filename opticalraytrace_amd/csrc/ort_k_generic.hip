// ort_k_generic.hip — the exact-fp64 kernels that walk a surface list they do not know at compile time: the generic
// queued walk (lists no program matches, the spot source, the monolithic scattering walk of variant bit 4, literal
// predicates of variant bit 1) and the lockstep kernel (the literal re-run of deferred rays, the parity / debug entry,
// variant bit 0 clear).  WIDE: the queued filtered walks on the 53-bit stream (the lockstep kernel chooses its stream at run time).
#include "ort_launch.h"
namespace ortk {
const char *launch_generic_f64(const GenericReq &q, const LaunchCfg &cfg, const TraceArgs &a)
{
    if (q.mode == MODE_DEBUG) {
        if (q.filt) return ORT_KLAUNCH((trace_kernel<MODE_DEBUG, true, double, true>));
        return ORT_KLAUNCH((trace_kernel<MODE_DEBUG, false, double, true>));
    }
    const bool fused = q.mode == MODE_FUSED;
    if (!q.queued) {
        if (fused) return q.filt ? ORT_KLAUNCH((trace_kernel<MODE_FUSED, true, double, true>)) : ORT_KLAUNCH((trace_kernel<MODE_FUSED, false, double, true>));
        return q.filt ? ORT_KLAUNCH((trace_kernel<MODE_RESIDENT, true, double, true>)) : ORT_KLAUNCH((trace_kernel<MODE_RESIDENT, false, double, true>));
    }
    if (q.wide) {                                            // filtered, clear media (launch_trace sends everything else to the lockstep kernel)
        if (!q.filt || q.scat) return nullptr;
        if (q.anysrc) {
            if (fused) return ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, true, double, PROG_GENERIC, false, RNG_WIDE>));
            return ORT_KLAUNCH((trace_queue_kernel<MODE_RESIDENT, true, true, double, PROG_GENERIC, false, RNG_WIDE>));
        }
        if (fused) return ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, false, double, PROG_GENERIC, false, RNG_WIDE>));
        return ORT_KLAUNCH((trace_queue_kernel<MODE_RESIDENT, true, false, double, PROG_GENERIC, false, RNG_WIDE>));
    }
    if (q.anysrc || !q.filt) {
        // alternate emitters and the A/B variants share the extended instantiations
        if (fused) {
            if (q.filt && !q.scat) return ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, true, double, PROG_GENERIC, false>));   // other emitters, clear media
            if (q.filt) return ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, true, double>));
            return ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, false, true, double>));
        }
        if (q.filt && !q.scat) return ORT_KLAUNCH((trace_queue_kernel<MODE_RESIDENT, true, true, double, PROG_GENERIC, false>));
        if (q.filt) return ORT_KLAUNCH((trace_queue_kernel<MODE_RESIDENT, true, true, double>));
        return ORT_KLAUNCH((trace_queue_kernel<MODE_RESIDENT, false, true, double>));
    }
    // the default emitters (ring / point) without scattering: the lean generic walk
    if (fused) return ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, false, double>));
    return ORT_KLAUNCH((trace_queue_kernel<MODE_RESIDENT, true, false, double>));
}

const char *launch_generic(const GenericReq &q, const LaunchCfg &cfg, const TraceArgs &a)
{
    if (q.precision == 2) return launch_generic_fast(q, cfg, a);
    if (q.precision == 1) return launch_generic_f32(q, cfg, a);
    return launch_generic_f64(q, cfg, a);
}
}  // namespace ortk
