// ort_k_program.h — the switch over the surface programs, shared by the units that instantiate them for one arithmetic
// and one RNG setting each (ort_k_prog64 / _strict / _wide / _prog32 / _fast .hip).
#pragma once
#include <stdio.h>
#include "ort_launch.h"

namespace ortk {

// RESIDENT: also the resident-bundle form of the default-source programs (the other sources exist fused only)
// the name a program launch reports: trace_queue_kernel<mode, arithmetic, program, strict, wide>
template <class T, int RNG>
inline const char *program_name(int prog, int mode)
{
    static thread_local char buf[160];
    static const char *const lists[] = {"PROG_GENERIC", "PROG_POINT", "PROG_RING", "PROG_POINT_IRIS_B", "PROG_POINT_IRIS_A", "PROG_RING_IRIS_B",
                                        "PROG_RING_IRIS_A", "PROG_POINT_BARE", "PROG_POINT_ELLIPSE"};
    static const char *const srcs[] = {"", " | SRC_CRS", " | SRC_ISORS", " | SRC_IMAGE", " | SRC_HANDED_OVER"};
    const int list = prog & PROG_LIST_MASK, src = prog >> 4;
    snprintf(buf, sizeof buf, "trace_queue_kernel<%s, %s, %s%s, strict=%d, wide=%d>", mode == MODE_FUSED ? "MODE_FUSED" : "MODE_RESIDENT",
             std::is_same<T, float>::value ? "float" : (std::is_same<T, fastd>::value ? "fastd" : "double"),
             list <= 8 ? lists[list] : "?", src <= 4 ? srcs[src] : "?", (RNG & RNG_STRICT) ? 1 : 0, (RNG & RNG_WIDE) ? 1 : 0);
    return buf;
}

template <class T, int RNG, bool RESIDENT, bool SOURCES>
inline const char *launch_program_t(int prog, int mode, const LaunchCfg &cfg, const TraceArgs &a)
{
    if (mode != MODE_FUSED && !(RESIDENT && mode == MODE_RESIDENT)) return nullptr;
    const char *const name = program_name<T, RNG>(prog, mode);
#define ORT_CASE(P)                                                                                                        \
    case P:                                                                                                                \
        if (mode == MODE_FUSED) return (void)ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, false, T, P, false, RNG>)), name;       \
        if constexpr (RESIDENT) return (void)ORT_KLAUNCH((trace_queue_kernel<MODE_RESIDENT, true, false, T, P, false, RNG>)), name;    \
        return nullptr;
#define ORT_CASE_FUSED(P)                                                                                                  \
    case P:                                                                                                                \
        if (mode == MODE_FUSED) return (void)ORT_KLAUNCH((trace_queue_kernel<MODE_FUSED, true, false, T, P, false, RNG>)), name;       \
        return nullptr;
    switch (prog) {
        ORT_PROGRAMS(ORT_CASE)
    default:
        break;
    }
    if constexpr (SOURCES) {
        switch (prog) {
            ORT_SOURCE_PROGRAMS(ORT_CASE_FUSED)
        default:
            break;
        }
    }
#undef ORT_CASE
#undef ORT_CASE_FUSED
    return nullptr;
}

}  // namespace ortk
