// ort_k_exp.hip — EXPERIMENTS on the headline kernel (the fused point program, exact fp64), reached only through the
// development knob ORT_DEV_EXP (ort_hip.hip: launch_one); no product path selects them.  Round 5, review item 2: the
// one combination round 4 left untried — persistent waves on eight per-XCD work heads WITHOUT the image atomics the
// returning pulls queued behind (NOBIN = the upper bound of what logging the hits instead of binning them could give).
//   1  static ranges, image atomic left out            (what the atomics cost under the static plan)
//   2  pulled ranges (vector atomic), hits binned      (round 4's losing variant, rebuilt as the reference point)
//   3  pulled ranges (vector atomic), atomic left out  (the untried pair)
//   4  pulled ranges (scalar atomic), hits binned
//   5  pulled ranges (scalar atomic), atomic left out
// Results: profiles/r05/pull_nobin.log.
#include "ort_launch.h"
namespace ortk {
const char *launch_exp(int which, const LaunchCfg &cfg, const TraceArgs &a)
{
    switch (which) {
    case 1: return ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, false, double, PROG_POINT, false, 0, SCHED_STATIC, true>));
    case 2: return ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, false, double, PROG_POINT, false, 0, SCHED_PULL_V, false>));
    case 3: return ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, false, double, PROG_POINT, false, 0, SCHED_PULL_V, true>));
    case 4: return ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, false, double, PROG_POINT, false, 0, SCHED_PULL_S, false>));
    case 5: return ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, false, double, PROG_POINT, false, 0, SCHED_PULL_S, true>));
    default: return nullptr;
    }
}
}  // namespace ortk
