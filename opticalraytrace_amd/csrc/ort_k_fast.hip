// ort_k_fast.hip — every fast-fp64 kernel (opt-in, ort_set_precision(2), ort_fastd.h: fused multiply-adds, Newton divide /
// Goldschmidt sqrt; ~1e-15 relative from the exact path, not bit-identical).
#include "ort_k_program.h"
namespace ortk {
const char *launch_program_fast(int prog, int mode, const LaunchCfg &cfg, const TraceArgs &a)
{
    return launch_program_t<fastd, 0, true, false>(prog, mode, cfg, a);
}
const char *launch_generic_fast(const GenericReq &q, const LaunchCfg &cfg, const TraceArgs &a)
{
    if (q.mode == MODE_DEBUG) return ORT_KLAUNCH((trace_kernel<MODE_DEBUG, true, fastd, true>));
    if (!q.queued) {                                         // the literal re-run of deferred rays
        if (q.mode == MODE_FUSED) return ORT_KLAUNCH((trace_kernel<MODE_FUSED, false, fastd, true>));
        return ORT_KLAUNCH((trace_kernel<MODE_RESIDENT, false, fastd, true>));
    }
    if (q.anysrc) {
        if (q.mode == MODE_FUSED) return ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, true, fastd>));
        return ORT_KLAUNCH((trace_queue_kernel<MODE_RESIDENT, true, true, fastd>));
    }
    if (q.mode == MODE_FUSED) return ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, false, fastd>));
    return ORT_KLAUNCH((trace_queue_kernel<MODE_RESIDENT, true, false, fastd>));
}
}  // namespace ortk
