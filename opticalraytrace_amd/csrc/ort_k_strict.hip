// ort_k_strict.hip — the fused surface programs in exact fp64 with STRICT LIBM EMITTERS (kernel variant bit 6): the light
// sources call glibc 2.35's own sin / cos / sincos (ort_libm.h), so every emitted ray — and with it every state, image and
// counter — equals the reference's bit for bit (src/sourceMod.f90:12-47, :250-300 through the platform libm).
#include "ort_k_program.h"
namespace ortk {
const char *launch_program_f64_strict(int prog, int mode, const LaunchCfg &cfg, const TraceArgs &a)
{
    return launch_program_t<double, RNG_STRICT, false, true>(prog, mode, cfg, a);
}
}  // namespace ortk
