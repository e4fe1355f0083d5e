// ort_k_prog32.hip — every fp32 kernel (BASELINE configs[4], ort_set_precision(1)): the surface programs (fused and resident
// for the default sources, fused for crs / isors / image), the generic queued walk and the lockstep kernel.  Always the
// filtered FORMS, never a deferral (ort_device.h kLoose) — in every kernel, so that they agree bit for bit.
#include "ort_k_program.h"
namespace ortk {
const char *launch_program_f32(int prog, int mode, const LaunchCfg &cfg, const TraceArgs &a)
{
    return launch_program_t<float, 0, true, true>(prog, mode, cfg, a);
}
const char *launch_generic_f32(const GenericReq &q, const LaunchCfg &cfg, const TraceArgs &a)
{
    if (q.mode == MODE_DEBUG) return ORT_KLAUNCH((trace_kernel<MODE_DEBUG, true, float, true>));
    if (q.queued && !q.anysrc) {                             // default emitters, clear media: the lean generic walk
        if (q.mode == MODE_FUSED) return ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, false, float>));
        return ORT_KLAUNCH((trace_queue_kernel<MODE_RESIDENT, true, false, float>));
    }
    if (q.mode == MODE_FUSED) return ORT_KLAUNCH((trace_kernel<MODE_FUSED, true, float, true>));
    return ORT_KLAUNCH((trace_kernel<MODE_RESIDENT, true, float, true>));
}
}  // namespace ortk
