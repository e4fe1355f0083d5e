// ort_k_prog64.hip — the surface programs in exact fp64 with the default emitters and ORT-RNG-v2: the production kernels
// (BASELINE configs[1]-[3]).  Fused and resident for the default sources, fused for crs / isors / image.
#include "ort_k_program.h"
namespace ortk {
const char *launch_program_f64(int prog, int mode, const LaunchCfg &cfg, const TraceArgs &a)
{
    return launch_program_t<double, 0, true, true>(prog, mode, cfg, a);
}
}  // namespace ortk
