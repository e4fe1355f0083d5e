// ort_k_wide.hip — the fused surface programs in exact fp64 on the 53-BIT STREAM ORT-RNG-v2w (kernel variant bit 5: one
// hash per draw, all of a double's mantissa, as random_number fills ran2's real(8), src/random_mod.f90:39-46), with the
// default and with the strict libm emitters.
#include "ort_k_program.h"
namespace ortk {
const char *launch_program_f64_wide(int prog, int mode, bool strict, const LaunchCfg &cfg, const TraceArgs &a)
{
    if (strict) return launch_program_t<double, RNG_WIDE | RNG_STRICT, false, true>(prog, mode, cfg, a);
    return launch_program_t<double, RNG_WIDE, false, true>(prog, mode, cfg, a);
}
}  // namespace ortk
