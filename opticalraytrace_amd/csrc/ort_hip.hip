// ort_hip.hip — the C ABI of include/ort.h (gfx950 / CDNA4, wave64) and the small kernels around the trace kernels
// (fold, hit-log binning, emit).  The trace kernels themselves are templates (ort_trace.h, ort_scatter.h over the device
// functions of ort_device.h) instantiated family by family in ort_k_*.hip and reached through ort_launch.h.
//
// No MFMA: there is no contraction anywhere on this path; the kernel is bound by
// fp64 VALU issue (IEEE divide / sqrt expansions), not by HBM.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <new>
#include <dlfcn.h>
#include <rccl/rccl.h>       // types and enums only: the library is resolved at run time (ort_allreduce)
#include "../../include/ort.h"
#include "ort_device.h"
#include "ort_launch.h"

using namespace ort;
using namespace ortk;

namespace {

// image[layer] += sum of the replicas' layer; replicas are left zero for the next launch.
// Thread j sums slot j of the 8 replicas (coalesced) and adds it to the bin the slot belongs to.
__global__ __launch_bounds__(256) void fold_kernel(int32_t *image, int32_t *replicas, int phase)
{
    const int nb = ORT_IMAGE_N * ORT_IMAGE_N;
    int32_t *img = image + (size_t)(phase - 1) * nb;
    int32_t *rep = replicas + (size_t)(phase - 1) * kSlots;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < kSlots; j += gridDim.x * blockDim.x) {
        const uint32_t bin = (j * kSlotMulInv) & (kSlots - 1);
        if (bin >= (uint32_t)nb) continue;        // 39 % of the slots belong to no bin and are never written
        int s = 0;
#pragma unroll
        for (int r = 0; r < kReplicas; ++r) {
            int32_t *p = rep + (size_t)r * kReplicaInts + j;
            int v = *p;
            if (v) { s += v; *p = 0; }
        }
        if (s) img[bin] += s;
    }
}

// The image atomics are performed at the memory side — one DRAM read-modify-write per hit, whatever the scope of the
// atomic, the number of replicas or their layout beyond "one hot bin per line" (profiles/r04/atomics_ab.log) — at ~21
// per ns for the whole device.  The fp64 kernels produce 14 hits per ns and do not notice; the fp32 point program would
// produce 32 and was BOUND by them (0.202 ms per 1e7 rays against 0.132 with the atomic compiled out).  So the fp32
// queued kernels write every hit to a log with plain coalesced stores (2 B per hit, sorted into kBinTiles parts by
// bin % kBinTiles), and this kernel bins the log of a launch in LDS: workgroup (u, t) takes the waves [u wpu, (u + 1) wpu)
// of the traced grid and part t (126 KB of LDS counters for its 32 161 bins — the focal blob spreads evenly over the
// parts), reads the directory slice once, then every wave of it streams the traced waves' entries; the counts are
// then ADDED — plain loads and stores, no atomic — to slab u
// of the layer, which this workgroup alone touches during the launch (launches are ordered by the stream).
// fold_slabs_kernel sums the kBinUnits slabs into the image when it is next needed (flush_replicas).
__global__ __launch_bounds__(kBinThreads) void bin_log_kernel(const uint16_t *log, uint64_t stride, const uint32_t *dir, uint32_t nwaves,
                                                               uint32_t wpu, int32_t *slabs)
{
    __shared__ int32_t H[kBinTile];
    __shared__ unsigned long long D[kBinDirMax];
    const uint32_t tile = blockIdx.y;
    const uint32_t w0 = blockIdx.x * wpu, w1 = w0 + wpu < nwaves ? w0 + wpu : nwaves;
    const uint32_t nd = w1 > w0 ? w1 - w0 : 0u;
    for (uint32_t j = threadIdx.x; j < (uint32_t)kBinTile; j += kBinThreads) H[j] = 0;
    for (uint32_t j = threadIdx.x; j < nd; j += kBinThreads) {
        const uint32_t *e = dir + (size_t)(w0 + j) * kBinDirWords;
        D[j] = ((unsigned long long)e[kBinTiles] << 32) | e[tile];            // first entry of the wave's region << 32 | hits of this tile
    }
    __syncthreads();
    const uint16_t *part = log + (size_t)tile * stride;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // A wave streams TWELVE traced waves' regions at a time, eight entries per lane and load (a region starts at a multiple
    // of 64 entries; the parts are padded): the loads of a step are in flight together — the kernel is a chain of memory
    // round trips (directory, entries, slab), not of work
    constexpr uint32_t W = kBinThreads / 64, G = 12;
    for (uint32_t r = wave; r < nd; r += W * G) {
        uint32_t base[G], cnt[G], most = 0;
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t rg = r + g * W;
            const unsigned long long e = rg < nd ? D[rg] : 0ull;
            base[g] = (uint32_t)(e >> 32); cnt[g] = (uint32_t)e;
            most = cnt[g] > most ? cnt[g] : most;
        }
        for (uint32_t j = lane * 8u; j < most; j += 512u) {
            uint4 v[G];
#pragma unroll
            for (uint32_t g = 0; g < G; ++g)
                v[g] = j < cnt[g] ? *reinterpret_cast<const uint4 *>(part + base[g] + j) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (uint32_t g = 0; g < G; ++g) {
                const uint32_t w4[4] = {v[g].x, v[g].y, v[g].z, v[g].w};
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k)
                    if (j + k < cnt[g]) atomicAdd(&H[(w4[k >> 1] >> (16u * (k & 1u))) & 0xffffu], 1);
            }
        }
    }
    __syncthreads();
    const uint32_t nb = ORT_IMAGE_N * ORT_IMAGE_N;
    int32_t *slab = slabs + (size_t)blockIdx.x * nb;
    // slab += H: the loads of a thread's bins first, all in flight together, then the stores (a load-add-store loop
    // waits for one memory round trip per bin)
    constexpr int PER = (kBinTile + kBinThreads - 1) / kBinThreads;
    int add[PER], have[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t j = threadIdx.x + (uint32_t)k * kBinThreads;
        add[k] = j < (uint32_t)kBinTile ? H[j] : 0;
        have[k] = 0;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t bin = (threadIdx.x + (uint32_t)k * kBinThreads) * (uint32_t)kBinTiles + tile;
        if (add[k] != 0 && bin < nb) have[k] = __builtin_nontemporal_load(slab + bin);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t bin = (threadIdx.x + (uint32_t)k * kBinThreads) * (uint32_t)kBinTiles + tile;
        if (add[k] != 0 && bin < nb) slab[bin] = have[k] + add[k];
    }
}

// image[layer] += sum of the layer's slabs; the slabs are left zero
__global__ __launch_bounds__(256) void fold_slabs_kernel(int32_t *image, int32_t *slabs, int phase)
{
    const uint32_t nb = ORT_IMAGE_N * ORT_IMAGE_N;
    int32_t *img = image + (size_t)(phase - 1) * nb;
    int32_t *sl = slabs + (size_t)(phase - 1) * kBinUnits * nb;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nb; j += gridDim.x * blockDim.x) {
        int s = 0;
        for (int u = 0; u < kBinUnits; ++u) {
            int32_t *p = sl + (size_t)u * nb + j;
            const int v = *p;
            if (v) { s += v; *p = 0; }
        }
        if (s) img[j] += s;
    }
}

__global__ __launch_bounds__(kBlock) void emit_kernel(const ort_system *sys, int phase,
                                                      uint64_t first_ray, uint64_t n, uint64_t rng_base,
                                                      double *pos_dir, const long long *img_cdf, int strict, int wide)
{
    __shared__ ort_system S;
    stage_system(S, sys);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        KeyedDrawsT<true> d;
        d.init_keyed(rng_base, first_ray + i, 0);
        d.set_wide(wide != 0);
        Ray r;
        emit<double, true>(S, phase, r, d, first_ray + i, img_cdf, strict != 0);
        pos_dir[0 * n + i] = (double)r.pos.x; pos_dir[1 * n + i] = (double)r.pos.y; pos_dir[2 * n + i] = (double)r.pos.z;
        pos_dir[3 * n + i] = (double)r.dir.x; pos_dir[4 * n + i] = (double)r.dir.y; pos_dir[5 * n + i] = (double)r.dir.z;
    }
}

thread_local char g_err[512] = "";

int fail(int code, const char *what, hipError_t e = hipSuccess)
{
    if (e != hipSuccess) snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    else snprintf(g_err, sizeof g_err, "%s", what);
    return code;
}

#define HIP_TRY(expr)                                              \
    do {                                                           \
        hipError_t e_ = (expr);                                    \
        if (e_ != hipSuccess) return fail(ORT_E_HIP, #expr, e_);   \
    } while (0)

int grid_for(uint64_t n)
{
    static int max_blocks = 0;
    if (!max_blocks) {
        const char *e = getenv("ORT_DEV_MAX_BLOCKS");        // development knob
        max_blocks = (e && atoi(e) > 0) ? atoi(e) : kMaxBlocks;
    }
    uint64_t b = (n + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > (uint64_t)max_blocks) b = max_blocks;
    return (int)b;
}

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return (e && atoi(e) > 0) ? atoi(e) : dflt;
}

// Rays per launch of the queued kernels: ORT_MAX_RAYS_PER_LAUNCH = 2^27 — every launch pays ~25 us of ramp and drain, and
// launches of 2^25 rays cost the ring loop 8 % and the 1e9-ray layers 4 % against these (profiles/r04: ring1e8 1.90e11 ->
// 2.07e11).  ORT_DEV_CHUNK_LOG2 (development knob, read once) makes them smaller: the tests cut small traces into several launches.
uint64_t chunk_rays()
{
    static const uint64_t chunk = [] {
        const int lg = env_int("ORT_DEV_CHUNK_LOG2", 27);
        const uint64_t c = 1ull << (lg < 6 ? 6 : (lg > 27 ? 27 : lg));
        return c < kChunkRaysMax ? c : kChunkRaysMax;
    }();
    return chunk;
}

// Ray ranges of a queued launch of n rays.  A launch of equal ranges ends with a partly filled last
// round of workgroups: the chip runs at 2 waves per SIMD instead of 5-6 for the last ~10 % of the
// time (measured: two launches overlapped on two streams took 10 % less than back to back).  So the
// bulk of the rays (kHeadPercent) goes to kHeadBlocks = 256 CUs x 5 workgroups in long ranges, and
// the rest to many workgroups of kTailBatches 64-ray batches per wave, which the dispatcher hands
// to whichever CU has room (the fp64 program kernels fit 6 per CU, so the first of them start
// beside the long ones): the chip drains within one short workgroup.  Swept on the GPU at 1e7 rays
// per launch (profiles/r02/range_plan_sweep.log): 1280 / 86 % / 6; 1024 or 1536 long workgroups
// +2 %, 2560 / 92 % / 4 +3 %, equal ranges +5 %; at 2^25 rays per launch the choice is within
// 0.5 %.  Small launches keep equal ranges.  Scheduling only: results do not depend on it.
// (The fp32 kernels fit 8 workgroups per CU; more long workgroups gained 3 % while the point program was bound by its image
// atomics and lose 12 % in the ring loop now that it is not — profiles/r04/headsweep.log, hb.log: one plan for all.)
// Round 5 (profiles/r05/tailbatch_sweep.log, plansweep.log): the LENGTH of a short range has to grow with the launch where a
// ray is cheap — the ring loop, whose segment 0 counts 69 % of the rays without emitting them: a wave with 6 batches holds
// ~120 candidates, runs its passes on partly filled wavefronts and pays the kernel's prologue for them.  `light` = the launch
// culls (1: fp64 arithmetics, 2: fp32, whose rays are cheaper still): 6 batches up to ~6e6 (fp32: 2.4e6) rays, then one more
// per 1e6 (4e5) rays, at most 32 — ring1e8 0.720 -> 0.666 ms, 2^27 rays 0.952 -> 0.874 (fp32 0.578 -> 0.472); the point loop
// (every ray emitted, 6.3 surface solves) is indifferent above 2e7 rays (12 batches: -0.5 %) and wants 6 at 1e7.
constexpr int kHeadBlocks = 1280, kHeadPercent = 86, kTailBatches = 6, kEqualBlocks = 1024;
constexpr uint64_t kTwoLevelRays = 2500000;
int tail_batches_for(uint64_t n, int light)
{
    static const int forced = env_int("ORT_DEV_TAIL_BATCHES", 0);          // development knob
    if (forced > 0) return forced;
    if (light == 0) return n > 16000000ull ? 2 * kTailBatches : kTailBatches;
    const uint64_t per = light == 2 ? 400000ull : 1000000ull;
    const uint64_t tb = n / per;
    return (int)(tb < (uint64_t)kTailBatches ? (uint64_t)kTailBatches : (tb > 32 ? 32 : tb));
}
int plan_ranges(TraceArgs &a, int light = 0)
{
    static const int head_blocks = env_int("ORT_DEV_HEAD_BLOCKS", kHeadBlocks), head_pct = env_int("ORT_DEV_HEAD_PERCENT", kHeadPercent);
    const int tail_batches = tail_batches_for(a.n_rays, light);
    const uint64_t n = a.n_rays;
    // Small launches: equal ranges on at most kEqualBlocks workgroups — ONE round of the chip with ranges as long as they
    // can be (a launch of 1e6 rays is 15 625 batches for 6 144 wave slots): on 3 072 workgroups, two rounds of 128-ray ranges,
    // the same launch took 0.053 ms in the ring loop and 0.067 in the point loop against 0.034 / 0.056 (profiles/r05/small_launches.log);
    // the two-level plan takes over where it wins, at ~2.5e6 rays.
    static const int equal_cap = env_int("ORT_DEV_MAX_BLOCKS", kEqualBlocks);
    if (n < kTwoLevelRays || head_pct >= 100) {               // equal ranges
        int grid = equal_cap;
        const uint64_t batches = (n + 63) / 64, blocks = (batches + kWavesPerBlock - 1) / kWavesPerBlock;
        if (blocks < (uint64_t)grid) grid = (int)blocks;
        const uint64_t nwaves = (uint64_t)grid * kWavesPerBlock;
        a.head_blocks = (uint32_t)grid;
        a.head_rays = n;
        a.head_chunk = (((n + nwaves - 1) / nwaves) + 63) & ~63ull;
        a.tail_chunk = 64;
        if (a.head_chunk > kMaxRange) {                      // (development knobs only: a small ORT_DEV_MAX_BLOCKS) more workgroups
            grid = (int)((n + kMaxRange * kWavesPerBlock - 1) / (kMaxRange * kWavesPerBlock));
            a.head_blocks = (uint32_t)grid;
            a.head_chunk = kMaxRange;
        }
        return grid;
    }
    const uint64_t hw = (uint64_t)head_blocks * kWavesPerBlock;
    a.head_blocks = (uint32_t)head_blocks;
    // (an absolute cap on the rays of the short workgroups instead of a share was measured and is worse at every launch
    // size: profiles/r05/tailsweep.log)
    a.head_chunk = ((n / 100 * (uint64_t)head_pct / hw) + 63) & ~63ull;
    if (a.head_chunk > kMaxRange) a.head_chunk = kMaxRange;          // (2^27 rays over 1 280 workgroups: 22 592)
    a.head_rays = a.head_chunk * hw < n ? a.head_chunk * hw : n;
    a.tail_chunk = 64ull * (uint64_t)tail_batches;
    const uint64_t rest = n - a.head_rays, tb = a.tail_chunk * kWavesPerBlock;
    return head_blocks + (int)((rest + tb - 1) / tb);
}

int redo_blocks()
{
    static int n = 0;
    if (!n) {
        const char *e = getenv("ORT_DEV_REDO_BLOCKS");       // development knob
        n = (e && atoi(e) > 0) ? atoi(e) : kRedoBlocks;
    }
    return n;
}

int check_system(const ort_system *sys)
{
    if (!sys) return fail(ORT_E_INVALID, "system is NULL");
    if (sys->abi_version != ORT_ABI_VERSION) return fail(ORT_E_INVALID, "ort_system.abi_version mismatch");
    for (int p = 0; p < 2; ++p) {
        int n = sys->n_surfaces[p];
        if (n < 1 || n > ORT_MAX_SURFACES) return fail(ORT_E_INVALID, "n_surfaces out of range");
        for (int k = 0; k < n; ++k) {
            int kind = sys->surfaces[p][k].kind;
            if (kind < ORT_SURF_PLANE || kind > ORT_SURF_IMAGE) return fail(ORT_E_INVALID, "bad surface kind");
            if ((kind == ORT_SURF_IMAGE) != (k == n - 1))
                return fail(ORT_E_INVALID, "the image plane must be the last surface, and only the last");
        }
        if (sys->split[p] < 0 || sys->split[p] > n) return fail(ORT_E_INVALID, "split out of range");
        if (sys->emitter[p] < ORT_EMIT_RING || sys->emitter[p] > ORT_EMIT_ISORS_NORING) return fail(ORT_E_INVALID, "bad emitter");
        if (sys->emitter[p] == ORT_EMIT_ISORS_NORING) {      // it reads the bottle's glass and contents from the point loop's list
            const ort_surface &in = sys->surfaces[1][0], &out = sys->surfaces[1][1];
            const bool wall = (in.kind == ORT_SURF_CYLINDER || in.kind == ORT_SURF_ELLIPSE) && in.kind == out.kind;
            if (sys->n_surfaces[1] < 2 || !wall || !(in.flags & ORT_F_BOTTLE) || !(out.flags & ORT_F_BOTTLE) || in.n2 != out.n1)
                return fail(ORT_E_INVALID, "ORT_EMIT_ISORS_NORING needs the point loop's list to start with the bottle's two walls");
        }
    }
    return ORT_OK;
}

}  // namespace

// Staged systems live in a ring of kSysSlots device slots fed from pinned host slots: ort_set_system copies
// asynchronously into the NEXT slot and later launches read that one, so a sweep can queue system after
// system without ever waiting for the stream (launches already queued keep reading the slot they were given).
constexpr int kSysSlots = 16;

struct ort_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    DevSystem *d_sys;            // the current slot of d_sys_ring
    DevSystem *d_sys_ring, *h_sys_ring;          // kSysSlots each; h_: pinned staging
    int sys_slot;
    hipEvent_t sys_ev[kSysSlots];                // recorded when a slot stops being current: its readers precede it
    bool sys_ev_set[kSysSlots];
    int32_t *d_image, *own_image;
    int32_t *d_replicas;         // kReplicas x 2 layers x kSlots: hits not yet folded into the image
    bool fold_pending[2];        // per layer: the replicas hold hits (fold_kernel runs when the image is needed)
    // fp32 queued launches (bin_log_kernel): the hit log of one launch, its directory, the slabs [2][kBinUnits][bins]
    uint16_t *d_hit_log;         // kBinTiles parts of hit_log_cap entries
    uint32_t *d_hit_dir;         // kBinDirWords words per traced wave
    uint64_t hit_log_cap, hit_dir_cap;
    uint64_t hit_used, hit_waves;            // entries per part / directory entries the launches since the last binning hold
    int hit_phase;                           // their layer (1 or 2), 0: the log is empty
    int32_t *d_slabs;
    bool slab_pending[2];
    hipEvent_t launch_ev[2];     // start / stop events the next kernel launch carries itself (null: none)
    char last_kernel[192];       // the instantiation the last trace launch ran (ort_last_kernel_name)
    // deferral group: consecutive fused launches of one phase / seed / system whose deferred rays share
    // the re-run list; the literal re-run is launched when the group closes (close_group)
    bool group_open;
    int group_mode;
    TraceArgs group_args;        // the group's first launch (first_ray = the base the list entries refer to)
    uint64_t group_rays;         // rays launched in the group so far (bound on the list's fill)
    uint32_t *d_redo_list;       // re-run list of the queued filtered kernel, redo_cap entries
    size_t redo_cap;
    unsigned int *d_redo_ctl;    // [2]: entries, re-run workgroups done; zero between launches
    long long *d_img_cdf;        // image-source table (ORT_IMAGE_SOURCE_CELLS + 1) or null
    unsigned long long *d_counters, *own_counters;
    unsigned long long *d_work;  // [ORT_NUM_WORK], see ort_work_counters
    bool timing;
    int variant;                 // bit mask, see ort_set_kernel_variant
    int precision;               // 0 fp64 (reference arithmetic), 1 fp32 (study path)
    int emitter[2];              // host copy of ort_system.emitter
    bool scatter[2];             // per phase: some surface of its list carries ORT_F_SCATTER
    int prog[2];                 // per phase: PROG_* the staged system matches (match_program)
    bool cont_prog[2];           // per phase: the surfaces from scat_k0 - 1 on are PROG_POINT_WALKED's (matches_behind)
    int scat_k0[2];              // per phase: > 0: the scattering pipeline applies, its continuation starts at this surface
    double *d_cont_pos_dir;      // hand-over bundle of the scattering pipeline (scatter_front_kernel), cont_cap entries
    double *d_cont_t;
    uint64_t *d_cont_draw;
    int32_t *d_cont_nis;
    uint64_t cont_cap;
    unsigned long long *d_scat_ctl;      // kScatCtlWords: the pipeline's work heads and slot counter (TraceArgs.scat_ctl)
    int n_cus;                   // compute units of the device (one workgroup of scatter_front_kernel each)
    double ring_cull;            // squared-radius threshold of the staged system (ring_cull_threshold), +inf: no culling
    float ring_cullf;
    uint32_t ring_cull_word, ring_cull_wordf;      // TraceArgs.cull_word / cull_wordf (cull_word_of)
    uint64_t ring_cull_wide;                       // TraceArgs.cull_wide (cull_wide_of)
    // multi-system launches (ort_trace_batch): per-simulation systems and arguments (device tables + pinned staging), the
    // simulations' re-run lists (bat_list_stride entries each) and control words, a scratch image for the simulations that
    // want none when they are traced one by one, the event that guards the staging buffers
    DevSystem *d_bat_sys, *h_bat_sys;
    TraceArgs *d_bat_args, *h_bat_args;
    int bat_cap;
    uint32_t *d_bat_list;
    uint64_t bat_list_cap;
    unsigned int *d_bat_ctl;
    int bat_ctl_cap;
    int32_t *d_bat_scratch;
    hipEvent_t bat_ev;
    bool bat_ev_set;
    ort_system cur_sys;          // host copy of the staged system (ort_trace_batch restages it behind simulations it runs one by one)
    // development (ort_debug_set_exp, ort_k_exp.hip): experiment on the fused point program, 0 = none; the pulled variants'
    // static share of a launch (percent), pull bounds (batches of 64 rays), workgroups per CU; their work heads: a ring of
    // kPullSets sets of eight 128-byte lines, zero when handed to a launch
    int exp_which, exp_static_pct, exp_min, exp_max, exp_wg_per_cu;
    unsigned int *d_pull_ctl;
    int pull_set;
    hipEvent_t ev[3][2];
    hipEvent_t ring[kTimingRing][2];   // fused-trace launches, most recent kTimingRing
    unsigned long long ring_count;
    bool ev_valid[3];
};

// does the surface list of `phase` equal program P field for field (kinds, flags other than the
// tracker's, aperture presence, queue point, default emitter)?
template <int P>
static bool matches(const ort_system *sys)
{
    const int p = Prog<P>::phase - 1;
    if (sys->n_surfaces[p] != Prog<P>::n || sys->split[p] != Prog<P>::split) return false;
    if (sys->emitter[p] != Prog<P>::emitter) return false;
    for (int k = 0; k < Prog<P>::n; ++k) {
        const ort_surface &s = sys->surfaces[p][k];
        if (s.kind != Prog<P>::kind[k] || (int)(s.flags & ~ORT_F_TRACK) != Prog<P>::flags[k] ||
            (s.aperture >= 0.0) != (Prog<P>::ap[k] != 0))
            return false;
        // OPT_ON_AXIS: the program kernels take pos.x - cx, pos.y - cy of a sphere for pos.x, pos.y
        if (s.kind == ORT_SURF_SPHERE && !(s.cx == 0.0 && s.cy == 0.0 && !std::signbit(s.cx) && !std::signbit(s.cy))) return false;
    }
    return true;
}

// Threshold on rr for the ring programs' segment 0: radius^2 (1 + margin) of the plano-convex aperture if
// the geometry the argument needs holds for this system, else +inf.  With the aim point T on the plane
// z = zt (ring_lens_z), the start point P (|P.xy| <= sqrt(r2), P.z in [zmin, zmax] on the bottle) and
// the flat face at zp, the ray crosses the face at P + (T - P) s, s = (zp - P.z) / (zt - P.z):
// |s - 1| <= |zp - zt| / (zt - zmax).  Required: zp within 1e-12 |zt| of zt and zt - zmax >= 1e-3 |zt|
// (so |s - 1| <= 1e-9), the start points inside the aperture (|P.xy| < radius), a real bottle point
// under every start point.  The crossing point's squared radius then differs from rr by less than
// 3e-9 rr + rounding (1e-14 in fp64, 1e-5 in fp32, which also moves zp by an ulp): margins 1e-6 / 1e-3.
// The largest 32-bit draw word w whose ray is NOT culled: rr(w) > cull is false.  rr(w) is formed with the operations of
// emit_ring and of the kernel's arithmetic (fp64 — also the fast fp64, whose product + 0 is the same value —: u = w 2^-32;
// fp32: u = (w >> 8) 2^-24), compiled here with the same -ffp-contract=off; it does not decrease with w, so a bisection
// finds the border.  0xffffffff when no word is culled.
template <class T> static uint32_t cull_word_of(T r2, T cull)
{
    auto dies = [&](uint32_t w) {
        volatile T u;
        if (sizeof(T) == 8) u = (T)((double)w * 0x1p-32);
        else u = (T)((float)(w >> 8) * 0x1.0p-24f);
        volatile T span = r2 - T(0.);
        volatile T prod = u * span;
        volatile T rr = T(0.) + prod;
        return rr > cull;
    };
    if (!dies(0xffffffffu)) return 0xffffffffu;
    if (dies(0u)) return 0xffffffffu;              // (cannot happen: rr(0) = 0 and cull > 0) nothing is culled
    uint32_t lo = 0u, hi = 0xffffffffu;            // lo survives, hi dies
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (dies(mid)) hi = mid; else lo = mid;
    }
    return lo;
}

// the same border for the 53-bit draws of ORT-RNG-v2w (fp64): the largest x = h >> 11 whose ray is not culled, u = x 2^-53
// formed as unit53 forms it (two exact halves, an exact sum).  ~0 when nothing is culled.
static uint64_t cull_wide_of(double r2, double cull)
{
    auto dies = [&](uint64_t x) {
        volatile double u = (double)(uint32_t)(x >> 32) * 0x1p-21 + (double)(uint32_t)x * 0x1p-53;
        volatile double span = r2 - 0.;
        volatile double prod = u * span;
        volatile double rr = 0. + prod;
        return rr > cull;
    };
    const uint64_t top = (1ull << 53) - 1ull;
    if (!dies(top) || dies(0ull)) return ~0ull;
    uint64_t lo = 0ull, hi = top;                  // lo survives, hi dies
    while (hi - lo > 1ull) {
        const uint64_t mid = lo + (hi - lo) / 2ull;
        if (dies(mid)) hi = mid; else lo = mid;
    }
    return lo;
}

static void ring_cull_threshold(const ort_system *sys, bool is_ring_program, double *cull, float *cullf)
{
    *cull = HUGE_VAL; *cullf = HUGE_VALF;
    if (!is_ring_program || getenv("ORT_DEV_NO_RING_CULL")) return;
    const ort_surface &s0 = sys->surfaces[0][0];
    if (s0.kind != ORT_SURF_PLANE || !(s0.aperture > 0.0)) return;
    const double zt = sys->ring_lens_z, zp = s0.cz, ra = sys->ring_bottle_ra, rb = sys->ring_bottle_rb;
    const double rmax = sqrt(sys->ring_r1 > sys->ring_r2 ? sys->ring_r1 : sys->ring_r2);     // |P.xy| <= rmax
    const double q = sys->ring_ellipse ? rmax * ra / rb : rmax;
    if (!(fabs(zp - zt) <= 1e-12 * fabs(zt)) || !(ra > 0.0) || !(ra * ra > q * q * 1.000001)) return;
    const double zmax = sys->ring_bottle_z + ra;
    if (!(zt - zmax >= 1e-3 * fabs(zt)) || !(rmax < s0.aperture) || !(sys->ring_lens_r2 > 0.0)) return;
    *cull = s0.aperture * s0.aperture * (1.0 + 1e-6);
    *cullf = (float)(s0.aperture * s0.aperture * (1.0 + 1e-3));
}

// the continuation of the scattering pipeline as a surface program: the list from surface k0 - 1 on (the wall the rays are
// handed over at, whose step the continuation finishes) equals P's — kinds, aperture presence, queue point, flags other than
// the tracker's and ORT_F_SCATTER — with k0 = 1 or 2 (the bottle's walls are steps 0 and 1 of the point loop)
template <int P>
static bool matches_behind(const ort_system *sys, int k0)
{
    const int p = Prog<P>::phase - 1;
    if (k0 < 1 || k0 > 2 || sys->n_surfaces[p] != Prog<P>::n || sys->split[p] != Prog<P>::split) return false;
    for (int k = k0 - 1; k < Prog<P>::n; ++k) {
        const ort_surface &s = sys->surfaces[p][k];
        if (s.kind != Prog<P>::kind[k] || (int)(s.flags & ~(ORT_F_TRACK | ORT_F_SCATTER)) != Prog<P>::flags[k] ||
            (s.aperture >= 0.0) != (Prog<P>::ap[k] != 0))
            return false;
        if (k >= k0 && (s.flags & ORT_F_SCATTER)) return false;
        if (s.kind == ORT_SURF_SPHERE && !(s.cx == 0.0 && s.cy == 0.0 && !std::signbit(s.cx) && !std::signbit(s.cy))) return false;
    }
    return true;
}

static bool list_scatters(const ort_system *sys, int p)
{
    for (int k = 0; k < sys->n_surfaces[p]; ++k)
        if (sys->surfaces[p][k].flags & ORT_F_SCATTER) return true;
    return false;
}

// the surface program the list of `phase` (1 / 2) matches field for field, PROG_GENERIC if none
static int program_of(const ort_system *sys, int phase)
{
    if (getenv("ORT_DEV_NO_PROGRAMS")) return PROG_GENERIC;                        // development knob (A/B)
    int prog = PROG_GENERIC;
#define ORT_MATCH(P) if (Prog<P>::phase == phase && matches<P>(sys)) prog = P;
    ORT_PROGRAMS(ORT_MATCH)
    ORT_SOURCE_PROGRAMS(ORT_MATCH)
#undef ORT_MATCH
    return prog;
}

// segment 0 of the ring programs: the borders of the cull in the three arithmetics' draw words
struct RingCull {
    double cull; float cullf;
    uint32_t word, wordf;
    uint64_t wide;
};
static RingCull ring_cull_of(const ort_system *sys, int prog_ring)
{
    RingCull r;
    ring_cull_threshold(sys, prog_ring != PROG_GENERIC && sys->emitter[0] == ORT_EMIT_RING, &r.cull, &r.cullf);
    r.word = cull_word_of<double>(sys->ring_lens_r2, r.cull);
    r.wordf = cull_word_of<float>((float)sys->ring_lens_r2, r.cullf);
    r.wide = cull_wide_of(sys->ring_lens_r2, r.cull);
    return r;
}

static void note_system(ort_ctx *c, const ort_system *sys)
{
    c->cur_sys = *sys;
    c->emitter[0] = sys->emitter[0]; c->emitter[1] = sys->emitter[1];
    for (int p = 0; p < 2; ++p) c->scatter[p] = list_scatters(sys, p);
    // the scattering pipeline takes the surfaces up to the last scattering one: they must be the walls of one
    // bottle (one kind — circular or elliptical cylinder —, no aperture stop) with something left behind them
    for (int p = 0; p < 2; ++p) {
        int last = -1;
        for (int k = 0; k < sys->n_surfaces[p]; ++k)
            if (sys->surfaces[p][k].flags & ORT_F_SCATTER) last = k;
        bool ok = last >= 0 && last + 1 < sys->n_surfaces[p] && last < 255;
        for (int k = 0; ok && k <= last; ++k) {
            const ort_surface &s = sys->surfaces[p][k];
            ok = (s.kind == ORT_SURF_CYLINDER || s.kind == ORT_SURF_ELLIPSE) && s.kind == sys->surfaces[p][0].kind && !(s.aperture >= 0.0);
        }
        c->scat_k0[p] = ok ? last + 1 : 0;
        c->cont_prog[p] = false;
    }
    c->cont_prog[1] = c->scat_k0[1] > 0 && matches_behind<PROG_POINT_WALKED>(sys, c->scat_k0[1]) && !getenv("ORT_DEV_NO_PROGRAMS");
    c->prog[0] = program_of(sys, 1); c->prog[1] = program_of(sys, 2);
    const RingCull rc = ring_cull_of(sys, c->prog[0]);
    c->ring_cull = rc.cull; c->ring_cullf = rc.cullf;
    c->ring_cull_word = rc.word; c->ring_cull_wordf = rc.wordf; c->ring_cull_wide = rc.wide;
}

// system + derived per-surface constants -> the next device slot, asynchronously (the staging copy is the
// slot's own pinned host buffer).  A slot is reused kSysSlots systems later; by then the event recorded when it
// was retired has normally long passed.
// what a context keeps of a system on the device: the caller's record, the derived per-surface constants, both once more in fp32
static void fill_dev_system(DevSystem &h, const ort_system *sys)
{
    h.sys = *sys;
    for (int p = 0; p < 2; ++p)
        for (int k = 0; k < ORT_MAX_SURFACES; ++k) h.aux[p][k] = make_aux<double>(sys->surfaces[p][k]);
    axial_start<double>(h.aux[1][0], sys->surfaces[1][0], sys->point_offset);           // OPT_AXIAL_START (point programs, step 0)
    convert_system(h.sysf, *sys);
    for (int p = 0; p < 2; ++p)
        for (int k = 0; k < ORT_MAX_SURFACES; ++k) h.auxf[p][k] = make_aux<float>(h.sysf.surfaces[p][k]);
    axial_start<float>(h.auxf[1][0], h.sysf.surfaces[1][0], h.sysf.point_offset);
}

static int upload_system(ort_ctx *c, const ort_system *sys, bool first = false)
{
    int slot = 0;
    if (!first) {
        HIP_TRY(hipEventRecord(c->sys_ev[c->sys_slot], c->stream));
        c->sys_ev_set[c->sys_slot] = true;
        slot = (c->sys_slot + 1) % kSysSlots;
        if (c->sys_ev_set[slot]) HIP_TRY(hipEventSynchronize(c->sys_ev[slot]));
    }
    DevSystem &h = c->h_sys_ring[slot];
    fill_dev_system(h, sys);
    c->sys_slot = slot;
    c->d_sys = c->d_sys_ring + slot;
    HIP_TRY(hipMemcpyAsync(c->d_sys, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
    return ORT_OK;
}

// Replicas -> image.  The fold is deferred until somebody needs the image (ort_read, ort_reset,
// ort_allreduce, ort_flush, a change of accumulators): a run of K back-to-back ort_trace calls pays
// for one fold, not K (each is a launch that reads 8 MB: ~2 % of a 1e7-ray launch).
static void launch_one(ort_ctx *c, int mode, const TraceArgs &a, int grid, bool queued, bool filt, bool anysrc);

// The rays the queued launches of the open group deferred (ort_device.h: a filtered predicate too
// close to call) are traced by the literal lockstep kernel — ONE launch per group instead of one
// behind every queued launch (an empty re-run launch cost ~10 us + a launch gap, 3 % of a 1e7-ray
// step).  The list can hold every ray of the group (launch_trace closes the group before it
// could overflow), so nothing is ever dropped.
static int close_group(ort_ctx *c)
{
    if (!c->group_open) return ORT_OK;
    c->group_open = false;
    TraceArgs a = c->group_args;
    a.listed = 1;
    a.n_rays = c->group_rays;
    launch_one(c, c->group_mode, a, redo_blocks(), false, false, true);
    HIP_TRY(hipGetLastError());
    return ORT_OK;
}

static int bin_pending(ort_ctx *c);
static int flush_replicas(ort_ctx *c)
{
    { const int rc = close_group(c); if (rc) return rc; }
    for (int p = 0; p < 2; ++p) {
        if (!c->fold_pending[p]) continue;
        hipLaunchKernelGGL(fold_kernel, dim3(256), dim3(256), 0, c->stream, c->d_image, c->d_replicas, p + 1);
        HIP_TRY(hipGetLastError());
        c->fold_pending[p] = false;
    }
    { const int rc = bin_pending(c); if (rc) return rc; }
    for (int p = 0; p < 2; ++p) {
        if (!c->slab_pending[p]) continue;
        hipLaunchKernelGGL(fold_slabs_kernel, dim3(256), dim3(256), 0, c->stream, c->d_image, c->d_slabs, p + 1);
        HIP_TRY(hipGetLastError());
        c->slab_pending[p] = false;
    }
    return ORT_OK;
}

extern "C" {

int ort_abi_version(void) { return ORT_ABI_VERSION; }

#ifndef ORT_BUILD_ID
#define ORT_BUILD_ID "unstamped"
#endif
const char *ort_build_id(void) { return ORT_BUILD_ID; }

const char *ort_last_error(void) { return g_err; }

int ort_device_count(int *count)
{
    if (!count) return fail(ORT_E_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(ORT_E_NODEVICE, "hipGetDeviceCount", e); }
    *count = n;
    return ORT_OK;
}

// everything of ort_create that can fail after the context exists (the caller destroys it on failure)
static int create_on_device(ort_ctx *c, const ort_system *sys)
{
    HIP_TRY(hipMalloc(&c->d_sys_ring, kSysSlots * sizeof(DevSystem)));
    HIP_TRY(hipHostMalloc(&c->h_sys_ring, kSysSlots * sizeof(DevSystem), hipHostMallocDefault));
    for (int k = 0; k < kSysSlots; ++k) HIP_TRY(hipEventCreateWithFlags(&c->sys_ev[k], hipEventDisableTiming));
    HIP_TRY(hipMalloc(&c->own_image, ORT_IMAGE_BINS * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&c->own_counters, ORT_NUM_COUNTERS * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&c->d_replicas, kReplicas * kReplicaInts * sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(c->d_replicas, 0, kReplicas * kReplicaInts * sizeof(int32_t), c->stream));
    HIP_TRY(hipMalloc(&c->d_work, ORT_NUM_WORK * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(c->d_work, 0, ORT_NUM_WORK * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipMalloc(&c->d_redo_ctl, 2 * sizeof(unsigned int)));
    HIP_TRY(hipMemsetAsync(c->d_redo_ctl, 0, 2 * sizeof(unsigned int), c->stream));
    HIP_TRY(hipMalloc(&c->d_scat_ctl, kScatCtlWords * sizeof(unsigned long long)));
    HIP_TRY(hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, c->device));
    if (c->n_cus < 1) c->n_cus = 1;
    c->d_image = c->own_image;
    c->d_counters = c->own_counters;
    for (int k = 0; k < 3; ++k) {
        HIP_TRY(hipEventCreate(&c->ev[k][0]));
        HIP_TRY(hipEventCreate(&c->ev[k][1]));
    }
    for (int k = 0; k < kTimingRing; ++k) {
        HIP_TRY(hipEventCreate(&c->ring[k][0]));
        HIP_TRY(hipEventCreate(&c->ring[k][1]));
    }
    note_system(c, sys);
    const int rc = upload_system(c, sys, true);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->d_image, 0, ORT_IMAGE_BINS * sizeof(int32_t), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, ORT_NUM_COUNTERS * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ORT_OK;
}

int ort_create(const ort_system *sys, int device, void *stream, ort_ctx **out)
{
    if (!out) return fail(ORT_E_INVALID, "out is NULL");
    *out = nullptr;
    int rc = check_system(sys);
    if (rc) return rc;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(ORT_E_NODEVICE, "no HIP device: the trace path has no CPU fallback");
    if (device < 0 || device >= n) return fail(ORT_E_NODEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    ort_ctx *c = new (std::nothrow) ort_ctx();
    if (!c) return fail(ORT_E_NOMEM, "host allocation failed");
    memset(c, 0, sizeof *c);
    c->device = device;
    c->variant = 1;
    // NULL = the device's default (null) stream, like every HIP API: work is then ordered with
    // whatever else the caller runs there (torch's default stream, RCCL's stream dependencies)
    c->stream = (hipStream_t)stream;
    c->own_stream = false;
    rc = create_on_device(c, sys);
    if (rc) {
        // a half-built context owns device memory and events: release them, keep the message
        char msg[sizeof g_err];
        memcpy(msg, g_err, sizeof msg);
        ort_destroy(c);
        memcpy(g_err, msg, sizeof msg);
        return rc;
    }
    *out = c;
    return ORT_OK;
}

int ort_destroy(ort_ctx *c)
{
    if (!c) return ORT_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 2; ++j) if (c->ev[k][j]) (void)hipEventDestroy(c->ev[k][j]);
    for (int k = 0; k < kTimingRing; ++k)
        for (int j = 0; j < 2; ++j) if (c->ring[k][j]) (void)hipEventDestroy(c->ring[k][j]);
    for (int k = 0; k < kSysSlots; ++k) if (c->sys_ev[k]) (void)hipEventDestroy(c->sys_ev[k]);
    (void)hipFree(c->d_sys_ring); (void)hipHostFree(c->h_sys_ring); (void)hipFree(c->own_image); (void)hipFree(c->own_counters); (void)hipFree(c->d_replicas); (void)hipFree(c->d_img_cdf);
    (void)hipFree(c->d_redo_list); (void)hipFree(c->d_redo_ctl); (void)hipFree(c->d_work);
    (void)hipFree(c->d_cont_pos_dir); (void)hipFree(c->d_cont_draw); (void)hipFree(c->d_cont_nis); (void)hipFree(c->d_cont_t);
    (void)hipFree(c->d_scat_ctl);
    (void)hipFree(c->d_hit_log); (void)hipFree(c->d_hit_dir); (void)hipFree(c->d_slabs);
    (void)hipFree(c->d_pull_ctl);
    (void)hipFree(c->d_bat_sys); (void)hipHostFree(c->h_bat_sys); (void)hipFree(c->d_bat_args); (void)hipHostFree(c->h_bat_args);
    (void)hipFree(c->d_bat_list); (void)hipFree(c->d_bat_ctl); (void)hipFree(c->d_bat_scratch);
    if (c->bat_ev) (void)hipEventDestroy(c->bat_ev);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return ORT_OK;
}

int ort_set_system(ort_ctx *c, const ort_system *sys)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    int rc = check_system(sys);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = close_group(c); if (rc) return rc; }     // its re-run must see the system its launches saw
    note_system(c, sys);
    return upload_system(c, sys);
}

int ort_set_image_source(ort_ctx *c, const int64_t *cdf)
{
    if (!c || !cdf) return fail(ORT_E_INVALID, "NULL argument");
    if (cdf[0] != 0) return fail(ORT_E_INVALID, "cdf[0] must be 0");
    for (int s = 0; s < ORT_IMAGE_SOURCE_CELLS; ++s)
        if (cdf[s + 1] < cdf[s]) return fail(ORT_E_INVALID, "cdf must be non-decreasing");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = close_group(c); if (rc) return rc; }     // its re-run emits from the table its launches used
    const size_t bytes = (size_t)(ORT_IMAGE_SOURCE_CELLS + 1) * sizeof(long long);
    if (!c->d_img_cdf) HIP_TRY(hipMalloc(&c->d_img_cdf, bytes));
    HIP_TRY(hipMemcpyAsync(c->d_img_cdf, cdf, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ORT_OK;
}

int ort_reset(ort_ctx *c)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    const int rc = flush_replicas(c);                      // leaves the replicas zero
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->d_image, 0, ORT_IMAGE_BINS * sizeof(int32_t), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, ORT_NUM_COUNTERS * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_work, 0, ORT_NUM_WORK * sizeof(unsigned long long), c->stream));
    return ORT_OK;
}

int ort_flush(ort_ctx *c)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    return flush_replicas(c);
}

// development: does the experiment armed by ort_debug_set_exp apply to this launch (the fused point program, exact fp64, ORT-RNG-v2)?
static bool exp_applies(const ort_ctx *c, int mode, const TraceArgs &a)
{
    return c->exp_which > 0 && mode == MODE_FUSED && c->precision == 0 && c->prog[a.phase - 1] == PROG_POINT && !a.strict && !a.wide &&
           !a.listed && (c->variant & 3) == 1 && !c->scatter[a.phase - 1];
}

// the pulled variants' plan of a launch: kWavesPerBlock x wg_per_cu x CUs persistent waves, each with a static first range
// (static_pct of the launch in equal shares), the rest handed out by eight heads (TraceArgs.pull_*)
constexpr int kPullSets = 256;
static int plan_pull(ort_ctx *c, TraceArgs &a, int *grid)
{
    if (!c->d_pull_ctl) {
        HIP_TRY(hipMalloc(&c->d_pull_ctl, (size_t)kPullSets * 8 * 128));
        HIP_TRY(hipMemsetAsync(c->d_pull_ctl, 0, (size_t)kPullSets * 8 * 128, c->stream));
        c->pull_set = 0;
    }
    if (c->pull_set == kPullSets) {
        HIP_TRY(hipMemsetAsync(c->d_pull_ctl, 0, (size_t)kPullSets * 8 * 128, c->stream));
        c->pull_set = 0;
    }
    const uint64_t n = a.n_rays, batches = (n + 63) / 64;
    uint64_t blocks = (uint64_t)c->n_cus * (uint64_t)c->exp_wg_per_cu;
    if (blocks * kWavesPerBlock > batches) blocks = (batches + kWavesPerBlock - 1) / kWavesPerBlock;
    const uint64_t nwaves = blocks * kWavesPerBlock;
    a.head_blocks = (uint32_t)blocks;
    a.head_chunk = ((n / 100 * (uint64_t)c->exp_static_pct / nwaves) + 63) & ~63ull;
    if (c->exp_static_pct == 0) a.head_chunk = 0;
    a.head_rays = a.head_chunk * nwaves < n ? a.head_chunk * nwaves : n;
    a.tail_chunk = 64;
    const uint64_t rest_b = (n - a.head_rays + 63) / 64;
    a.pull_share_b = (uint32_t)((rest_b + 7) / 8);
    a.pull_wph = (uint32_t)((nwaves + 7) / 8);
    a.pull_min = (uint32_t)c->exp_min; a.pull_max = (uint32_t)c->exp_max;
    a.pull_ctl = c->d_pull_ctl + (size_t)c->pull_set * 8 * 32;
    c->pull_set++;
    *grid = (int)blocks;
    return ORT_OK;
}

// One kernel of the trace family on `grid` workgroups: the most specific family that holds a kernel for the request
// (ort_launch.h) — a surface program the staged system matches, else the generic walk / the lockstep kernel.
static void launch_one(ort_ctx *c, int mode, const TraceArgs &a, int grid, bool queued, bool filt, bool anysrc)
{
    const bool scat = c->scatter[a.phase - 1];
    const LaunchCfg cfg = {grid, c->stream, c->launch_ev[0], c->launch_ev[1]};
    const char *name = nullptr;
    // a program's draw indices are compile-time constants: they assume the emitter's own number of draws in front of the
    // first surface (resident bundles may come with another draw_base)
    int prog = c->prog[a.phase - 1];
    if (mode != MODE_FUSED && a.draw_base != (a.phase == 1 ? 4 : 2)) prog = PROG_GENERIC;
    const bool program = prog != PROG_GENERIC && queued && mode != MODE_DEBUG && !scat && (filt || c->precision == 1);
    if (program && exp_applies(c, mode, a)) name = launch_exp(c->exp_which, cfg, a);
    else if (program) {
        if (c->precision == 2) name = launch_program_fast(prog, mode, cfg, a);
        else if (c->precision == 1) name = launch_program_f32(prog, mode, cfg, a);
        else if (a.wide) name = launch_program_f64_wide(prog, mode, a.strict != 0, cfg, a);
        else if (a.strict) name = launch_program_f64_strict(prog, mode, cfg, a);
        else name = launch_program_f64(prog, mode, cfg, a);
    }
    if (!name) {
        // fp32: always the filtered forms (ort_device.h kLoose); its queued generic walk exists for the default emitters in
        // clear media only — everything else takes the lockstep kernel
        const GenericReq q = {c->precision, mode, queued, filt, anysrc, scat, a.wide != 0};
        name = launch_generic(q, cfg, a);
    }
    // (the literal re-run that closes a group is not "the kernel the trace ran": the name stays the queued launch's)
    if (!a.listed) snprintf(c->last_kernel, sizeof c->last_kernel, "%s", name ? name : "(none)");
}

// The re-run list holds one entry per ray of a deferral group.  A group must be able to take every
// ray of a launch (launches cover at most chunk_rays()); beyond that the list is sized for several
// launches, so that a run of back-to-back launches closes a group — one literal re-run launch —
// only now and then: 8 launches of the largest size seen, at most 2^27 entries (512 MB of the 288 GB).
constexpr uint64_t kListMax = 1ull << 27;
static int reserve_list(ort_ctx *c, uint64_t n_rays)
{
    const uint64_t chunk = n_rays < chunk_rays() ? n_rays : chunk_rays();
    uint64_t want = 8 * chunk < kListMax ? 8 * chunk : kListMax;
    if (want < chunk) want = chunk;
    if (want > c->redo_cap) {
        { const int rc = close_group(c); if (rc) return rc; }
        HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_redo_list);
        c->d_redo_list = nullptr; c->redo_cap = 0;
        // (several launches' worth is a convenience — one re-run launch per eight queued ones; a device short of memory
        // gets the list of ONE launch, which is all a group needs)
        if (hipMalloc(&c->d_redo_list, want * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            c->d_redo_list = nullptr;
            want = chunk;
            if (hipMalloc(&c->d_redo_list, want * sizeof(uint32_t)) != hipSuccess) {
                (void)hipGetLastError();
                c->d_redo_list = nullptr;
                return fail(ORT_E_NOMEM, "device memory: the re-run list of one launch (4 bytes per ray) does not fit");
            }
        }
        c->redo_cap = want;
    }
    return ORT_OK;
}

// Scratch of the scattering pipeline: one hand-over entry (68 bytes) per ray of a launch; launches of that path
// cover at most 2^27 rays (9 GB of the 288 GB, allocated only as far as a trace needs it).
// Every launch ends with the longest random walks of its rays — one scattering event per pass of a wave, ~45 events
// for the longest of 4e6 rays: ~0.3 ms per launch that nothing else fills (time = 0.3 ms x launches + 0.116 ms per 1e6 rays,
// profiles/r04/scatbench.log: 4e7 rays cost 0.138 / 0.131 / 0.125 ms per 1e6 rays in launches of 2^24 / 2^25 / 2^26).
// So the launches are as large as the scratch may be: a 1e9-ray layer takes 8 of them, not 30.
constexpr int kScatterChunkLog2 = 27;
static uint64_t scatter_chunk()
{
    static const uint64_t chunk = 1ull << env_int("ORT_DEV_SCAT_CHUNK_LOG2", kScatterChunkLog2);     // development knob
    return chunk;
}
// Workgroups of scatter_front_kernel for a launch of n rays: one per CU (12 persistent wavefronts each), fewer when the
// launch has fewer than 12 batches of 64 rays per CU.  What a wave takes per pull: 1/32 of its share of the launch, at
// least one batch, at most four (the end of a launch is as ragged as one pull is long).
static unsigned scatter_groups(const ort_ctx *c, uint64_t n)
{
    const uint64_t batches = (n + 63) / 64, want = (batches + kScatWaves - 1) / kScatWaves;
    const uint64_t most = (uint64_t)env_int("ORT_DEV_SCAT_GROUPS", c->n_cus);         // development knob
    return (unsigned)(want < most ? (want ? want : 1) : most);
}
static uint32_t scatter_grab(uint64_t n, unsigned groups)
{
    static const int forced = env_int("ORT_DEV_SCAT_GRAB", 0);                        // development knob (batches per pull)
    const uint64_t per_wave = (n + 63) / 64 / ((uint64_t)groups * kScatWaves);
    uint64_t g = per_wave / 32;
    if (g < 1) g = 1;
    if (g > 4) g = 4;
    if (forced > 0) g = (uint64_t)forced;
    return (uint32_t)(64 * g);
}
// fp32 queued launches.  The hit log holds the launches since the last binning (eight launches of the largest size seen so
// far, at most kHitLogEntries 16-bit entries per part: 1.34 GB of the 288 GB), the directory eight words per traced
// wave, the slabs (2 layers x kBinUnits x 643 KB = 66 MB) what bin_log_kernel has accumulated since the last fold.
// Binning is LAZY like the fold: bin_log_kernel costs ~20 us whatever the log holds (a chain of memory round trips, and
// one scattered store per non-empty bin and unit), so the launches of a run append to the log and one kernel bins them
// when the log or the directory is full, when the other layer's launch comes, or when the image is needed
// (flush_replicas).  Tracing and binning are ordered by the stream.
constexpr uint64_t kHitLogEntries = kChunkRaysMax + 64;      // one launch of the largest size, or many smaller ones
constexpr uint64_t kHitDirEntries = (uint64_t)kBinUnits * kBinDirMax;
static int bin_pending(ort_ctx *c)
{
    if (!c->hit_phase) return ORT_OK;
    const uint64_t nwaves = c->hit_waves;
    const uint32_t units = nwaves < (uint64_t)kBinUnits ? (uint32_t)nwaves : (uint32_t)kBinUnits;
    const uint32_t wpu = (uint32_t)((nwaves + units - 1) / units);           // <= kBinDirMax: launch_trace bins before it could not be
    hipLaunchKernelGGL(bin_log_kernel, dim3(units, kBinTiles), dim3(kBinThreads), 0, c->stream, (const uint16_t *)c->d_hit_log,
                       (uint64_t)c->hit_log_cap, (const uint32_t *)c->d_hit_dir, (uint32_t)nwaves, wpu,
                       c->d_slabs + (size_t)(c->hit_phase - 1) * kBinUnits * ORT_IMAGE_N * ORT_IMAGE_N);
    HIP_TRY(hipGetLastError());
    c->slab_pending[c->hit_phase - 1] = true;
    c->hit_used = 0; c->hit_waves = 0; c->hit_phase = 0;
    return ORT_OK;
}

// room for a launch of `rays` rays on `nwaves` waves of layer `phase` behind what the log holds (bins it first if not)
static int reserve_hit_log(ort_ctx *c, int phase, uint64_t rays, uint64_t nwaves)
{
    if (nwaves > kHitDirEntries) return fail(ORT_E_INVALID, "fp32 hit log: the launch has more waves than the binning kernel's directory holds");
    if (c->hit_phase && (c->hit_phase != phase || c->hit_used + rays + 8 > c->hit_log_cap || c->hit_waves + nwaves > kHitDirEntries)) {
        const int rc = bin_pending(c);
        if (rc) return rc;
    }
    if (rays + 8 > c->hit_log_cap || !c->d_hit_dir || !c->d_slabs) {
        HIP_TRY(hipStreamSynchronize(c->stream));            // (the log is empty here; a kernel may still be reading the old buffers)
        if (rays + 8 > c->hit_log_cap) {
            (void)hipFree(c->d_hit_log); c->d_hit_log = nullptr; c->hit_log_cap = 0;
            // room for eight launches of the largest size seen (one binning per eight launches), at most kHitLogEntries per part:
            // a 1 000-ray run holds a log of kilobytes, not the 1.34 GB of a 2^27-ray launch
            uint64_t want = 8 * (rays + 64);
            if (want > kHitLogEntries) want = kHitLogEntries;
            if (want < rays + 8) want = rays + 8;
            const uint64_t cap = (want + 63) & ~63ull;          // (+8: bin_log_kernel reads eight entries at a time)
            uint64_t got = cap;
            if (hipMalloc(&c->d_hit_log, (size_t)kBinTiles * got * sizeof(uint16_t)) != hipSuccess) {
                (void)hipGetLastError();                           // short of memory: the log of this one launch
                c->d_hit_log = nullptr;
                got = (rays + 8 + 63) & ~63ull;
                if (hipMalloc(&c->d_hit_log, (size_t)kBinTiles * got * sizeof(uint16_t)) != hipSuccess) {
                    (void)hipGetLastError();
                    c->d_hit_log = nullptr;
                    return fail(ORT_E_NOMEM, "device memory: the fp32 hit log of one launch (10 bytes per ray) does not fit");
                }
            }
            c->hit_log_cap = got;
        }
        if (!c->d_hit_dir) HIP_TRY(hipMalloc(&c->d_hit_dir, kHitDirEntries * kBinDirWords * sizeof(uint32_t)));
        if (!c->d_slabs) {
            const size_t bytes = (size_t)2 * kBinUnits * ORT_IMAGE_N * ORT_IMAGE_N * sizeof(int32_t);
            HIP_TRY(hipMalloc(&c->d_slabs, bytes));
            HIP_TRY(hipMemsetAsync(c->d_slabs, 0, bytes, c->stream));
        }
    }
    return ORT_OK;
}

// does this launch go to an fp32 instantiation of trace_queue_kernel (launch_one's own decision)?
static bool fp32_queued_launch(const ort_ctx *c, int mode, int phase, bool queued, bool anysrc)
{
    if (c->precision != 1 || !queued || mode == MODE_DEBUG) return false;
    if (!anysrc) return true;
    return mode == MODE_FUSED && !c->scatter[phase - 1] && c->prog[phase - 1] > PROG_LIST_MASK;
}

static int reserve_handover(ort_ctx *c, uint64_t n_rays)
{
    // every ray of a launch can be handed over, and every wavefront leaves up to one chunk of slots partly used
    const uint64_t rays = n_rays < scatter_chunk() ? n_rays : scatter_chunk();
    const uint64_t want = rays + (uint64_t)scatter_groups(c, rays) * kScatWaves * kHandChunk;
    if (want <= c->cont_cap) return ORT_OK;
    { const int rc = close_group(c); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_cont_pos_dir); (void)hipFree(c->d_cont_draw); (void)hipFree(c->d_cont_nis); (void)hipFree(c->d_cont_t);
    c->d_cont_pos_dir = nullptr; c->d_cont_draw = nullptr; c->d_cont_nis = nullptr; c->d_cont_t = nullptr; c->cont_cap = 0;
    HIP_TRY(hipMalloc(&c->d_cont_pos_dir, 6 * want * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_cont_t, want * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_cont_draw, want * sizeof(uint64_t)));
    HIP_TRY(hipMalloc(&c->d_cont_nis, want * sizeof(int32_t)));
    c->cont_cap = want;
    return ORT_OK;
}

static int launch_trace(ort_ctx *c, int mode, TraceArgs &a0, int evk)
{
    a0.sys = &c->d_sys->sys; a0.aux = c->d_sys->aux[a0.phase - 1];
    a0.sysf = &c->d_sys->sysf; a0.auxf = c->d_sys->auxf[a0.phase - 1];
    a0.image = c->d_image; a0.counters = c->d_counters; a0.work = c->d_work;
    const bool use_rep = (c->variant & 4) == 0 && mode != MODE_DEBUG;
    a0.replicas = use_rep ? c->d_replicas : nullptr;
    a0.img_cdf = c->d_img_cdf;
    a0.in_stride = a0.n_rays;
    const bool culling = (c->variant & 8) == 0;                 // bit 3: A/B knob, ring rays are all emitted
    a0.strict = ((c->variant & 64) != 0 && c->precision == 0) ? 1 : 0;
    a0.wide = (c->variant & 32) ? 1 : 0;
    a0.cull_word = culling ? c->ring_cull_word : 0xffffffffu;
    a0.cull_wordf = culling ? c->ring_cull_wordf : 0xffffffffu;
    a0.cull_wide = culling ? c->ring_cull_wide : ~0ull;
    if (a0.first_ray > ORT_MAX_RAY_INDEX || a0.n_rays > ORT_MAX_RAY_INDEX - a0.first_ray)
        return fail(ORT_E_INVALID, "ray indices reach beyond 2^40 (ORT_MAX_RAY_INDEX): the keyed draw counter holds 40 bits of ray index");
    if (a0.n_rays == 0) return ORT_OK;
    const bool filt = (c->variant & 2) == 0 && c->precision != 1;      // fp32: literal predicates, nothing is deferred
    // 53-bit draws: the queued kernels of the exact fp64 path have WIDE instantiations (surface programs, the generic filtered
    // walk in clear media, the scattering pipeline); every other combination takes the lockstep kernel, which chooses its
    // stream at run time
    const bool pipeline_ok = c->scat_k0[a0.phase - 1] > 0 && (c->variant & 16) == 0;
    const bool wide_queued = c->precision == 0 && filt && (!c->scatter[a0.phase - 1] || (pipeline_ok && mode == MODE_FUSED));
    const bool queued = (c->variant & 1) && mode != MODE_DEBUG && (!a0.wide || wide_queued);
    const bool anysrc_emitter = c->emitter[a0.phase - 1] != (a0.phase == 1 ? ORT_EMIT_RING : ORT_EMIT_POINT);
    const bool anysrc = anysrc_emitter || c->scatter[a0.phase - 1];
    // The queued filtered kernel defers the rays that sit on a decision boundary to a list, which
    // the literal lockstep kernel traces right after it.  Every ray of a launch could be on it
    // (an axial beam meets every flat face at costt == 1), so a launch covers at most chunk_rays().
    const bool deferring = queued && filt;
    const uint64_t total = a0.n_rays;
    // scattering media, exact fp64, the default kernel variant: the three-stage pipeline (scatter_front_kernel)
    const bool pipeline = mode == MODE_FUSED && deferring && c->precision == 0 && c->scatter[a0.phase - 1] && pipeline_ok;
    // fp32 hit log: worth its second kernel (~10 us + the launch gap) where a large share of the rays is binned — the point loop
    // (42 % of its rays: -28 % per step); the ring loop bins 1 % of its rays and keeps the atomics.  ORT_DEV_HIT_LOG = 1 never,
    // 2 the point loop (default), 3 both loops (development knob; the image is the same bit for bit)
    static const int hit_log_mode = env_int("ORT_DEV_HIT_LOG", 2);
    const bool logging = fp32_queued_launch(c, mode, a0.phase, queued, anysrc) && hit_log_mode >= (a0.phase == 2 ? 2 : 3);
    // (a queued launch indexes its rays with 32 bits; a logging launch fits the hit log: kHitLogEntries)
    const uint64_t step = pipeline ? scatter_chunk() : (queued ? chunk_rays() : total);
    if (pipeline) {
        const int rc = reserve_handover(c, total);
        if (rc) return rc;
        a0.cont_pos_dir = c->d_cont_pos_dir; a0.cont_t = c->d_cont_t; a0.cont_draw = c->d_cont_draw; a0.cont_nis = c->d_cont_nis;
        a0.cont_cap = c->cont_cap; a0.cont_k0 = c->scat_k0[a0.phase - 1];
        a0.scat_ctl = c->d_scat_ctl;
    }
    if (deferring) {
        const int rc = reserve_list(c, total);
        if (rc) return rc;
        a0.redo_list = c->d_redo_list;
        a0.redo_ctl = c->d_redo_ctl;
    }
    const int slot = (int)(c->ring_count % kTimingRing);
    // fused traces (evk 0) are timed by the ring's event pair.  A trace that is ONE kernel launch
    // hands the pair to the launch itself (hipExtLaunchKernel: start / stop of that dispatch, no
    // packet of its own); every separate event record is a packet the command processor handles
    // between two kernels (~2 us each at 1e7 rays per launch).
    const bool one_launch = total <= step && !pipeline;        // the pipeline is two kernels per launch
    const bool ext_timed = c->timing && evk == 0 && one_launch;
    if (c->timing && evk > 0) HIP_TRY(hipEventRecord(c->ev[evk][0], c->stream));
    if (c->timing && evk == 0 && !ext_timed) HIP_TRY(hipEventRecord(c->ring[slot][0], c->stream));
    for (uint64_t off = 0; off < total; off += step) {
        TraceArgs a = a0;
        a.n_rays = total - off < step ? total - off : step;
        a.first_ray = a0.first_ray + off;
        if (a.pos_dir_in) a.pos_dir_in += off;              // same component stride (in_stride)
        // queued: every wave walks 64-aligned contiguous ranges (plan_ranges); lockstep: grid-stride
        // (what a ray of this launch costs: the ring programs' segment 0 counts most of them without emitting them)
        const bool culls = a.phase == 1 && c->prog[0] != PROG_GENERIC && c->emitter[0] == ORT_EMIT_RING && mode == MODE_FUSED &&
                           (c->precision == 1 ? a.cull_wordf : a.cull_word) != 0xffffffffu;
        int grid = queued ? plan_ranges(a, culls ? (c->precision == 1 ? 2 : 1) : 0) : grid_for(a.n_rays);
        if (queued && c->exp_which >= 2 && exp_applies(c, mode, a)) { const int rc = plan_pull(c, a, &grid); if (rc) return rc; }
        if (deferring && mode == MODE_FUSED) {
            // fused launches share the re-run list of their group; the literal re-run comes when the
            // group closes (close_group)
            const TraceArgs &g = c->group_args;
            const bool fits = c->group_open && c->group_mode == mode && g.phase == a.phase && g.rng_base == a.rng_base &&
                              a.first_ray >= g.first_ray && (a.first_ray - g.first_ray) + a.n_rays <= (1ull << 32) &&
                              c->group_rays + a.n_rays <= c->redo_cap;
            if (!fits) {
                { const int rc = close_group(c); if (rc) return rc; }
                c->group_args = a;
                c->group_mode = mode;
                c->group_rays = 0;
                c->group_open = true;
            }
            a.defer_base = a.first_ray - c->group_args.first_ray;
            c->group_rays += a.n_rays;
        } else {
            const int rc = close_group(c);                  // anything else runs behind the group's re-run
            if (rc) return rc;
        }
        if (ext_timed) { c->launch_ev[0] = c->ring[slot][0]; c->launch_ev[1] = c->ring[slot][1]; }
        if (pipeline) {
            const unsigned groups = scatter_groups(c, a.n_rays);
            a.scat_grab = scatter_grab(a.n_rays, groups);
            a.scat_share = (((a.n_rays + kScatHeads - 1) / kScatHeads) + 63) & ~63ull;
            HIP_TRY(hipMemsetAsync(c->d_scat_ctl, 0, kScatCtlWords * sizeof(unsigned long long), c->stream));
            const LaunchCfg front = {(int)groups, c->stream, nullptr, nullptr}, cont = {grid, c->stream, nullptr, nullptr};
            (void)launch_scatter_front(anysrc_emitter, a.wide != 0, front, a);
            HIP_TRY(hipGetLastError());
            snprintf(c->last_kernel, sizeof c->last_kernel, "scatter_front_kernel + %s", launch_continue(c->cont_prog[a.phase - 1], a.wide != 0, cont, a));
        } else if (logging) {
            const uint64_t nwaves = (uint64_t)grid * kWavesPerBlock;
            { const int rc = reserve_hit_log(c, a.phase, a.n_rays, nwaves); if (rc) return rc; }
            a.hit_log = c->d_hit_log; a.hit_stride = c->hit_log_cap;
            a.hit_base = (uint32_t)c->hit_used;                              // a multiple of 64: the waves' regions stay aligned
            a.hit_dir = c->d_hit_dir + c->hit_waves * kBinDirWords;
            launch_one(c, mode, a, grid, queued, filt, anysrc);
            c->hit_used = (c->hit_used + a.n_rays + 63) & ~63ull;
            c->hit_waves += nwaves;
            c->hit_phase = a.phase;
        } else
        launch_one(c, mode, a, grid, queued, filt, anysrc);
        c->launch_ev[0] = c->launch_ev[1] = nullptr;
        HIP_TRY(hipGetLastError());                         // a failed launch is reported where it happened
        if (deferring && mode != MODE_FUSED) {              // resident bundles: re-run at once (the bundle is the caller's)
            a.listed = 1;
            launch_one(c, mode, a, redo_blocks(), false, false, true);
            HIP_TRY(hipGetLastError());
        }
    }
    if (use_rep) c->fold_pending[a0.phase - 1] = true;     // folded when the image is next needed (flush_replicas)
    if (c->timing && evk > 0) { HIP_TRY(hipEventRecord(c->ev[evk][1], c->stream)); c->ev_valid[evk] = true; }
    if (c->timing && evk == 0) {
        if (!ext_timed) HIP_TRY(hipEventRecord(c->ring[slot][1], c->stream));
        c->ring_count++;
        c->ev_valid[0] = true;
    }
    return ORT_OK;
}

int ort_reserve(ort_ctx *c, uint64_t n_rays)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    return reserve_list(c, n_rays);
}

int ort_trace(ort_ctx *c, int phase, uint64_t first_ray, uint64_t n_rays, uint64_t seed)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    if (phase != 1 && phase != 2) return fail(ORT_E_INVALID, "phase must be 1 (ring) or 2 (point)");
    HIP_TRY(hipSetDevice(c->device));
    TraceArgs a;
    memset(&a, 0, sizeof a);
    a.first_ray = first_ray; a.n_rays = n_rays; a.rng_base = stream_base(seed, phase);
    a.phase = phase;
    return launch_trace(c, MODE_FUSED, a, 0);
}

// ---- multi-system launches ------------------------------------------------------------------------------------------
// A simulation of the batch goes into a multi-system launch when the context runs the default exact path (fp64, queued,
// filtered, ORT-RNG-v2, default emitters' arithmetic), its list of `phase` is a surface program ort_k_batch.hip holds, clear
// media, not the image source (one table per context) and at most kBatchRaysMax rays (a simulation's re-run list holds
// every ray of it).  Everything else is traced one by one through ort_set_system / ort_attach_buffers / ort_trace, which is
// also what the batch means.
constexpr uint64_t kBatchRaysMax = 1ull << 24;        // beyond it a simulation fills the chip by itself
constexpr uint64_t kBatchListMax = 1ull << 28;        // re-run list entries of one multi-system launch (1 GB)
constexpr int kBatchTargetBlocks = 256 * 6 * 4;       // workgroups of a launch: four rounds of six per CU
constexpr int kBatchRedoBlocks = 4;                   // workgroups per simulation of the batched re-run (normally they read a zero count and return)

// Ray ranges of ONE simulation of a multi-system launch of `cnt` simulations of a.n_rays rays each: as plan_ranges cuts a launch
// of its own, with the long workgroups shared out — 1 280 of them in the whole launch (at least one per simulation), 86 % of a
// simulation's rays in them, the rest in short ranges whose length follows the launch's TOTAL size (tail_batches_for).  The
// dispatch order is simulation-fastest (trace_batch_kernel: blockIdx.x), so the long workgroups of all simulations start first
// and the launch drains on short ones.  (Round 5's first plan — equal ranges on ~6 144 workgroups, simulation-major dispatch —
// took 6 % longer for the 75 ring loops of the lens experiment and 3 % for 75 point loops; a group of 25 point loops was 4 %
// faster with it: profiles/r05/batch_probe.log and HISTORY.md.  One plan for single and multi-system launches.)
// Returns the workgroups per simulation (gridDim.y).
static int plan_batch(TraceArgs &a, int cnt, int light)
{
    const uint64_t n = a.n_rays, per_block = 64ull * kWavesPerBlock;
    uint64_t hb = ((uint64_t)kHeadBlocks + cnt / 2) / cnt;
    if (hb < 1) hb = 1;
    const uint64_t batches = (n + 63) / 64, blocks_most = (batches + kWavesPerBlock - 1) / kWavesPerBlock;
    if (n < hb * per_block * 4) {                            // small simulations: equal ranges, the launch aims at kBatchTargetBlocks workgroups
        uint64_t by = ((uint64_t)kBatchTargetBlocks + cnt - 1) / cnt;
        if (by > blocks_most) by = blocks_most;
        if (by < 1) by = 1;
        const uint64_t nwaves = by * kWavesPerBlock;
        a.head_blocks = (uint32_t)by; a.head_rays = n;
        a.head_chunk = (((n + nwaves - 1) / nwaves) + 63) & ~63ull;       // (< kMaxRange: n < 4 hb per_block)
        a.tail_chunk = 64;
        return (int)by;
    }
    const uint64_t hw = hb * kWavesPerBlock;
    a.head_blocks = (uint32_t)hb;
    a.head_chunk = ((n / 100 * (uint64_t)kHeadPercent / hw) + 63) & ~63ull;
    if (a.head_chunk > kMaxRange) {                           // (one long workgroup per simulation and > 3e5 rays in it: more of them)
        hb = (n / 100 * (uint64_t)kHeadPercent + kMaxRange * kWavesPerBlock - 1) / (kMaxRange * kWavesPerBlock);
        a.head_blocks = (uint32_t)hb;
        a.head_chunk = kMaxRange;
    }
    a.head_rays = a.head_chunk * (uint64_t)a.head_blocks * kWavesPerBlock < n ? a.head_chunk * (uint64_t)a.head_blocks * kWavesPerBlock : n;
    a.tail_chunk = 64ull * (uint64_t)tail_batches_for(n * (uint64_t)cnt, light);
    const uint64_t rest = n - a.head_rays, tb = a.tail_chunk * kWavesPerBlock;
    return (int)(a.head_blocks + (rest + tb - 1) / tb);
}

static int reserve_batch(ort_ctx *c, int n_sys, uint64_t list_stride)
{
    if (n_sys > c->bat_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_bat_sys); (void)hipHostFree(c->h_bat_sys); (void)hipFree(c->d_bat_args); (void)hipHostFree(c->h_bat_args);
        c->d_bat_sys = c->h_bat_sys = nullptr; c->d_bat_args = c->h_bat_args = nullptr; c->bat_cap = 0;
        const int cap = n_sys < 64 ? 64 : n_sys;
        HIP_TRY(hipMalloc(&c->d_bat_sys, (size_t)cap * sizeof(DevSystem)));
        HIP_TRY(hipHostMalloc(&c->h_bat_sys, (size_t)cap * sizeof(DevSystem), hipHostMallocDefault));
        HIP_TRY(hipMalloc(&c->d_bat_args, (size_t)cap * sizeof(TraceArgs)));
        HIP_TRY(hipHostMalloc(&c->h_bat_args, (size_t)cap * sizeof(TraceArgs), hipHostMallocDefault));
        c->bat_cap = cap;
    }
    if (n_sys > c->bat_ctl_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_bat_ctl); c->d_bat_ctl = nullptr; c->bat_ctl_cap = 0;
        const int cap = n_sys < 64 ? 64 : n_sys;
        HIP_TRY(hipMalloc(&c->d_bat_ctl, (size_t)cap * 2 * sizeof(unsigned int)));
        HIP_TRY(hipMemsetAsync(c->d_bat_ctl, 0, (size_t)cap * 2 * sizeof(unsigned int), c->stream));   // the re-run leaves them zero
        c->bat_ctl_cap = cap;
    }
    const uint64_t want = (uint64_t)n_sys * list_stride;
    if (want > c->bat_list_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_bat_list); c->d_bat_list = nullptr; c->bat_list_cap = 0;
        HIP_TRY(hipMalloc(&c->d_bat_list, want * sizeof(uint32_t)));
        c->bat_list_cap = want;
    }
    if (!c->bat_ev) HIP_TRY(hipEventCreateWithFlags(&c->bat_ev, hipEventDisableTiming));
    return ORT_OK;
}

int ort_trace_batch(ort_ctx *c, int n, const ort_system *systems, int phase, uint64_t first_ray, uint64_t n_rays, uint64_t seed,
                    void *const *d_images, void *const *d_counters)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    if (phase != 1 && phase != 2) return fail(ORT_E_INVALID, "phase must be 1 (ring) or 2 (point)");
    if (n < 0 || (n > 0 && (!systems || !d_counters))) return fail(ORT_E_INVALID, "bad batch: n < 0, or systems / d_counters NULL");
    if (first_ray > ORT_MAX_RAY_INDEX || n_rays > ORT_MAX_RAY_INDEX - first_ray)
        return fail(ORT_E_INVALID, "ray indices reach beyond 2^40 (ORT_MAX_RAY_INDEX): the keyed draw counter holds 40 bits of ray index");
    for (int i = 0; i < n; ++i) {
        const int rc = check_system(&systems[i]);
        if (rc) { char msg[sizeof g_err]; snprintf(msg, sizeof msg, "system %d of the batch: %.400s", i, g_err); return fail(rc, msg); }
        if (!d_counters[i]) return fail(ORT_E_INVALID, "d_counters[i] is NULL: every simulation of a batch has counters of its own");
        if (systems[i].emitter[phase - 1] == ORT_EMIT_IMAGE)
            return fail(ORT_E_INVALID, "the image source holds ONE table per context (ort_set_image_source): trace such simulations with ort_set_system + ort_trace");
    }
    if (n == 0 || n_rays == 0) return ORT_OK;
    HIP_TRY(hipSetDevice(c->device));
    const int p = phase - 1;
    const bool ctx_ok = c->precision == 0 && (c->variant & ~4) == 1 && n_rays <= kBatchRaysMax && !c->exp_which;
    // the program of every simulation (PROG_GENERIC: it is traced one by one), batched ones in the order of their programs
    int *prog = (int *)malloc((size_t)n * 3 * sizeof(int)), *order = prog + n, *grid_y = prog + 2 * n;    // grid_y[k]: workgroups per simulation of the group that starts at order[k]
    if (!prog) return fail(ORT_E_NOMEM, "host allocation failed");
    int nb = 0;
    for (int i = 0; i < n; ++i) {
        prog[i] = (ctx_ok && !list_scatters(&systems[i], p)) ? program_of(&systems[i], phase) : PROG_GENERIC;
        if (!batch_has_program(prog[i])) prog[i] = PROG_GENERIC;
        if (prog[i] != PROG_GENERIC) order[nb++] = i;
    }
    for (int a = 1; a < nb; ++a) {                                        // stable insertion sort by program (n is a few hundred at most)
        const int v = order[a];
        int b = a;
        while (b > 0 && prog[order[b - 1]] > prog[v]) { order[b] = order[b - 1]; --b; }
        order[b] = v;
    }
    int rc = ORT_OK;
#define BATCH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(ORT_E_HIP, #expr, e_); goto done; } } while (0)
    if (c->timing) BATCH_TRY(hipEventRecord(c->ev[1][0], c->stream));
    if (nb > 0) {
        const uint64_t stride = (n_rays + 63) & ~63ull;                   // a simulation's re-run list: every ray of it can be deferred
        const int per_launch = (int)((kBatchListMax / stride) < (uint64_t)nb ? (kBatchListMax / stride) : (uint64_t)nb);
        const uint64_t rng_base = stream_base(seed, phase);
        for (int at = 0; at < nb; at += per_launch) {
            const int m = nb - at < per_launch ? nb - at : per_launch;
            rc = reserve_batch(c, m, stride);
            if (rc) goto done;
            if (c->bat_ev_set) BATCH_TRY(hipEventSynchronize(c->bat_ev));          // the staging buffers' last copies have been read
            for (int j = 0; j < m; ++j) {
                const int i = order[at + j];
                const ort_system *sys = &systems[i];
                fill_dev_system(c->h_bat_sys[j], sys);
                DevSystem *d = c->d_bat_sys + j;
                TraceArgs &a = c->h_bat_args[j];
                memset(&a, 0, sizeof a);
                a.sys = &d->sys; a.aux = d->aux[p]; a.sysf = &d->sysf; a.auxf = d->auxf[p];
                a.image = (int32_t *)(d_images ? d_images[i] : nullptr);      // NULL: the hit is counted, not binned
                a.replicas = nullptr;
                a.counters = (unsigned long long *)d_counters[i];
                a.work = c->d_work;
                a.first_ray = first_ray; a.n_rays = n_rays; a.rng_base = rng_base;
                a.phase = phase; a.in_stride = n_rays;
                a.redo_list = c->d_bat_list + (uint64_t)j * stride;
                a.redo_ctl = c->d_bat_ctl + 2 * j;
                const RingCull cull = phase == 1 ? ring_cull_of(sys, prog[i]) : RingCull{HUGE_VAL, HUGE_VALF, 0xffffffffu, 0xffffffffu, ~0ull};
                a.cull_word = cull.word; a.cull_wordf = cull.wordf; a.cull_wide = cull.wide;
            }
            // the ray ranges of a simulation's workgroups: plan_batch, one plan per group (launch) of simulations
            for (int g0 = 0; g0 < m;) {
                int g1 = g0;
                while (g1 < m && prog[order[at + g1]] == prog[order[at + g0]]) ++g1;
                const bool culls = phase == 1 && c->h_bat_args[g0].cull_word != 0xffffffffu;
                const int by = plan_batch(c->h_bat_args[g0], g1 - g0, culls ? 1 : 0);
                for (int j = g0; j < g1; ++j) {
                    TraceArgs &a = c->h_bat_args[j];
                    const TraceArgs &p0 = c->h_bat_args[g0];
                    a.head_blocks = p0.head_blocks; a.head_rays = p0.head_rays; a.head_chunk = p0.head_chunk; a.tail_chunk = p0.tail_chunk;
                }
                grid_y[at + g0] = by;
                g0 = g1;
            }
            BATCH_TRY(hipMemcpyAsync(c->d_bat_sys, c->h_bat_sys, (size_t)m * sizeof(DevSystem), hipMemcpyHostToDevice, c->stream));
            BATCH_TRY(hipMemcpyAsync(c->d_bat_args, c->h_bat_args, (size_t)m * sizeof(TraceArgs), hipMemcpyHostToDevice, c->stream));
            BATCH_TRY(hipEventRecord(c->bat_ev, c->stream));
            c->bat_ev_set = true;
            for (int g0 = 0; g0 < m;) {
                int g1 = g0;
                while (g1 < m && prog[order[at + g1]] == prog[order[at + g0]]) ++g1;
                const LaunchCfg cfg = {grid_y[at + g0], c->stream, nullptr, nullptr};
                const char *name = launch_batch(prog[order[at + g0]], cfg, g1 - g0, c->d_bat_args + g0);
                BATCH_TRY(hipGetLastError());
                snprintf(c->last_kernel, sizeof c->last_kernel, "%s x %d", name ? name : "(none)", g1 - g0);
                g0 = g1;
            }
            const LaunchCfg redo = {kBatchRedoBlocks, c->stream, nullptr, nullptr};
            (void)launch_batch_rerun(redo, m, c->d_bat_args);
            BATCH_TRY(hipGetLastError());
        }
    }
    if (nb < n) {
        // the others one by one, as the batch is defined; then the context's own system and accumulators again
        const ort_system saved = c->cur_sys;
        int32_t *img = c->d_image == c->own_image ? nullptr : c->d_image;
        unsigned long long *cnt = c->d_counters == c->own_counters ? nullptr : c->d_counters;
        for (int i = 0; i < n && !rc; ++i) {
            if (prog[i] != PROG_GENERIC) continue;
            void *image = d_images ? d_images[i] : nullptr;
            if (!image) {
                if (!c->d_bat_scratch) BATCH_TRY(hipMalloc(&c->d_bat_scratch, ORT_IMAGE_BINS * sizeof(int32_t)));
                image = c->d_bat_scratch;            // (nobody reads it: whatever it holds)
            }
            rc = ort_set_system(c, &systems[i]);
            if (!rc) rc = ort_attach_buffers(c, image, d_counters[i]);
            if (!rc) rc = ort_trace(c, phase, first_ray, n_rays, seed);
        }
        if (!rc) rc = ort_set_system(c, &saved);
        if (!rc) rc = ort_attach_buffers(c, img, cnt);
    }
    if (!rc && c->timing) { BATCH_TRY(hipEventRecord(c->ev[1][1], c->stream)); c->ev_valid[1] = true; }
done:
#undef BATCH_TRY
    free(prog);
    return rc;
}

int ort_trace_resident(ort_ctx *c, int phase, uint64_t first_ray, uint64_t n_rays, uint64_t seed,
                       int draw_base, const double *d_pos_dir)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    if (phase != 1 && phase != 2) return fail(ORT_E_INVALID, "phase must be 1 (ring) or 2 (point)");
    if (!d_pos_dir && n_rays) return fail(ORT_E_INVALID, "d_pos_dir is NULL");
    if (draw_base < 0) return fail(ORT_E_INVALID, "draw_base < 0");
    HIP_TRY(hipSetDevice(c->device));
    TraceArgs a;
    memset(&a, 0, sizeof a);
    a.first_ray = first_ray; a.n_rays = n_rays; a.rng_base = stream_base(seed, phase);
    a.phase = phase; a.draw_base = draw_base; a.pos_dir_in = d_pos_dir;
    return launch_trace(c, MODE_RESIDENT, a, 1);
}

int ort_emit(ort_ctx *c, int phase, uint64_t first_ray, uint64_t n_rays, uint64_t seed, double *d_pos_dir)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    if (phase != 1 && phase != 2) return fail(ORT_E_INVALID, "phase must be 1 (ring) or 2 (point)");
    if (!d_pos_dir && n_rays) return fail(ORT_E_INVALID, "d_pos_dir is NULL");
    if (first_ray > ORT_MAX_RAY_INDEX || n_rays > ORT_MAX_RAY_INDEX - first_ray)
        return fail(ORT_E_INVALID, "ray indices reach beyond 2^40 (ORT_MAX_RAY_INDEX): the keyed draw counter holds 40 bits of ray index");
    if (n_rays == 0) return ORT_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (c->timing) HIP_TRY(hipEventRecord(c->ev[2][0], c->stream));
    hipLaunchKernelGGL(emit_kernel, dim3(grid_for(n_rays)), dim3(kBlock), 0, c->stream,
                       &c->d_sys->sys, phase, first_ray, n_rays, stream_base(seed, phase), d_pos_dir,
                       (const long long *)c->d_img_cdf, ((c->variant & 64) != 0 && c->precision == 0) ? 1 : 0,
                       (c->variant & 32) ? 1 : 0);
    HIP_TRY(hipGetLastError());
    if (c->timing) { HIP_TRY(hipEventRecord(c->ev[2][1], c->stream)); c->ev_valid[2] = true; }
    return ORT_OK;
}

int ort_trace_rays(ort_ctx *c, int phase, int64_t n, const double *pos_dir_in, int nu, const double *u,
                   int draw_base, uint64_t seed, uint64_t first_ray, double *pos_dir_out,
                   double *emitted_out, int32_t *status, int32_t *bin_xy, int32_t *n_isect,
                   int32_t *n_draws)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    if (phase != 1 && phase != 2) return fail(ORT_E_INVALID, "phase must be 1 (ring) or 2 (point)");
    if (n < 0 || draw_base < 0 || nu < 0 || (u && nu == 0)) return fail(ORT_E_INVALID, "bad sizes");
    if (n == 0) return ORT_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t N = (size_t)n;
    double *d_in = nullptr, *d_u = nullptr, *d_out = nullptr, *d_em = nullptr;
    int32_t *d_st = nullptr, *d_bin = nullptr, *d_nis = nullptr, *d_nd = nullptr;
    int rc = ORT_OK;
#define TRY_GOTO(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(ORT_E_HIP, #expr, e_); goto done; } } while (0)
    if (pos_dir_in) {
        TRY_GOTO(hipMalloc(&d_in, 6 * N * sizeof(double)));
        TRY_GOTO(hipMemcpyAsync(d_in, pos_dir_in, 6 * N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    if (u) {
        TRY_GOTO(hipMalloc(&d_u, (size_t)nu * N * sizeof(double)));
        TRY_GOTO(hipMemcpyAsync(d_u, u, (size_t)nu * N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    if (pos_dir_out) TRY_GOTO(hipMalloc(&d_out, 6 * N * sizeof(double)));
    if (emitted_out) TRY_GOTO(hipMalloc(&d_em, 6 * N * sizeof(double)));
    if (status) TRY_GOTO(hipMalloc(&d_st, N * sizeof(int32_t)));
    if (bin_xy) TRY_GOTO(hipMalloc(&d_bin, 2 * N * sizeof(int32_t)));
    if (n_isect) TRY_GOTO(hipMalloc(&d_nis, N * sizeof(int32_t)));
    if (n_draws) TRY_GOTO(hipMalloc(&d_nd, N * sizeof(int32_t)));
    {
        TraceArgs a;
        memset(&a, 0, sizeof a);
        a.first_ray = first_ray; a.n_rays = (uint64_t)n; a.rng_base = stream_base(seed, phase);
        a.phase = phase; a.draw_base = draw_base; a.pos_dir_in = d_in; a.u = d_u; a.nu = nu;
        a.pos_dir_out = d_out; a.emitted_out = d_em; a.status = d_st; a.bin_xy = d_bin;
        a.n_isect = d_nis; a.n_draws = d_nd;
        rc = launch_trace(c, MODE_DEBUG, a, -1);
        if (rc) goto done;
    }
    if (d_out) TRY_GOTO(hipMemcpyAsync(pos_dir_out, d_out, 6 * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (d_em) TRY_GOTO(hipMemcpyAsync(emitted_out, d_em, 6 * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (d_st) TRY_GOTO(hipMemcpyAsync(status, d_st, N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (d_bin) TRY_GOTO(hipMemcpyAsync(bin_xy, d_bin, 2 * N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (d_nis) TRY_GOTO(hipMemcpyAsync(n_isect, d_nis, N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (d_nd) TRY_GOTO(hipMemcpyAsync(n_draws, d_nd, N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TRY_GOTO(hipStreamSynchronize(c->stream));
done:
#undef TRY_GOTO
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_in); (void)hipFree(d_u); (void)hipFree(d_out); (void)hipFree(d_em);
    (void)hipFree(d_st); (void)hipFree(d_bin); (void)hipFree(d_nis); (void)hipFree(d_nd);
    return rc;
}

int ort_trace_paths(ort_ctx *c, int phase, int64_t n, uint64_t seed, uint64_t first_ray,
                    double *path, int32_t *npath, int32_t *status)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    if (phase != 1 && phase != 2) return fail(ORT_E_INVALID, "phase must be 1 (ring) or 2 (point)");
    if (n < 0 || !path || !npath) return fail(ORT_E_INVALID, "bad arguments");
    if (n == 0) return ORT_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t N = (size_t)n;
    double *d_path = nullptr;
    int32_t *d_np = nullptr, *d_st = nullptr;
    int rc = ORT_OK;
#define TRY_GOTO(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(ORT_E_HIP, #expr, e_); goto done; } } while (0)
    TRY_GOTO(hipMalloc(&d_path, N * ORT_MAX_PATH * 3 * sizeof(double)));
    TRY_GOTO(hipMalloc(&d_np, N * sizeof(int32_t)));
    TRY_GOTO(hipMalloc(&d_st, N * sizeof(int32_t)));
    TRY_GOTO(hipMemsetAsync(d_path, 0, N * ORT_MAX_PATH * 3 * sizeof(double), c->stream));
    {
        TraceArgs a;
        memset(&a, 0, sizeof a);
        a.first_ray = first_ray; a.n_rays = (uint64_t)n; a.rng_base = stream_base(seed, phase);
        a.phase = phase; a.path = d_path; a.npath = d_np; a.status = d_st;
        rc = launch_trace(c, MODE_DEBUG, a, -1);
        if (rc) goto done;
    }
    TRY_GOTO(hipMemcpyAsync(path, d_path, N * ORT_MAX_PATH * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TRY_GOTO(hipMemcpyAsync(npath, d_np, N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (status) TRY_GOTO(hipMemcpyAsync(status, d_st, N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TRY_GOTO(hipStreamSynchronize(c->stream));
done:
#undef TRY_GOTO
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_path); (void)hipFree(d_np); (void)hipFree(d_st);
    return rc;
}

int ort_read(ort_ctx *c, int32_t *image, uint64_t *counters)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = flush_replicas(c); if (rc) return rc; }
    if (image)
        HIP_TRY(hipMemcpyAsync(image, c->d_image, ORT_IMAGE_BINS * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (counters)
        HIP_TRY(hipMemcpyAsync(counters, c->d_counters, ORT_NUM_COUNTERS * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ORT_OK;
}

int ort_attach_buffers(ort_ctx *c, void *d_image, void *d_counters)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    if ((d_image == nullptr) != (d_counters == nullptr))
        return fail(ORT_E_INVALID, "attach both buffers or neither");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = flush_replicas(c); if (rc) return rc; }     // pending hits belong to the old accumulators
    c->d_image = d_image ? (int32_t *)d_image : c->own_image;
    c->d_counters = d_counters ? (unsigned long long *)d_counters : c->own_counters;
    return ORT_OK;
}

// RCCL, resolved lazily: a process that never reduces never loads it, and a process that already
// holds an RCCL (PyTorch-ROCm bundles one under the same SONAME) gets that very library back
// from dlopen instead of a second copy.
namespace {
struct Rccl {
    void *lib;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int *);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
    int n;                                   // communicators currently held ...
    int devices[ORT_MAX_DEVICES];            // ... for these devices, in this order
    ncclComm_t comms[ORT_MAX_DEVICES];
    int fault;                               // tests (ort_debug_fault_allreduce): the next collective is given an invalid datatype
} g_rccl;

// the communicators this process holds are given back (ort_comm_destroy; a failed collective: the next call starts afresh)
void rccl_drop()
{
    for (int i = 0; i < g_rccl.n; ++i) (void)g_rccl.CommDestroy(g_rccl.comms[i]);
    g_rccl.n = 0;
}

int rccl_load()
{
    if (g_rccl.lib) return ORT_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(ORT_E_NOCOMM, "librccl.so.1 cannot be loaded (ort_allreduce needs RCCL)");
#define ORT_SYM(field, name)                                                               \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));              \
    if (!g_rccl.field) return fail(ORT_E_NOCOMM, "librccl.so.1 lacks " name)
    ORT_SYM(CommInitAll, "ncclCommInitAll");
    ORT_SYM(CommDestroy, "ncclCommDestroy");
    ORT_SYM(CommCount, "ncclCommCount");
    ORT_SYM(GroupStart, "ncclGroupStart");
    ORT_SYM(GroupEnd, "ncclGroupEnd");
    ORT_SYM(AllReduce, "ncclAllReduce");
    ORT_SYM(GetErrorString, "ncclGetErrorString");
#undef ORT_SYM
    g_rccl.lib = h;
    return ORT_OK;
}

int rccl_fail(const char *what, ncclResult_t r)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return ORT_E_NOCOMM;
}
#define RCCL_TRY(expr)                                                   \
    do {                                                                 \
        ncclResult_t r_ = (expr);                                        \
        if (r_ != ncclSuccess) return rccl_fail(#expr, r_);              \
    } while (0)
}  // namespace

int ort_allreduce(ort_ctx **ctxs, int n)
{
    if (!ctxs || n < 1 || n > ORT_MAX_DEVICES) return fail(ORT_E_INVALID, "ctxs NULL or n outside 1..ORT_MAX_DEVICES");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail(ORT_E_INVALID, "a context is NULL");
        for (int j = 0; j < i; ++j)
            if (ctxs[j]->device == ctxs[i]->device) return fail(ORT_E_INVALID, "two contexts on one device: RCCL ranks are devices");
    }
    int rc = rccl_load();
    if (rc) return rc;
    bool same = g_rccl.n == n;
    for (int i = 0; same && i < n; ++i) same = g_rccl.devices[i] == ctxs[i]->device;
    if (!same) {
        rccl_drop();
        int devs[ORT_MAX_DEVICES];
        for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
        RCCL_TRY(g_rccl.CommInitAll(g_rccl.comms, n, devs));
        for (int i = 0; i < n; ++i) g_rccl.devices[i] = devs[i];
        g_rccl.n = n;
    }
    // one group: image (int32 x 321 602 = 1.29 MB) and counters (uint64 x 8) of every device, each on
    // its context's stream, so the sums are ordered after the traces already queued there
    for (int i = 0; i < n; ++i) {
        HIP_TRY(hipSetDevice(ctxs[i]->device));
        const int frc = flush_replicas(ctxs[i]);
        if (frc) return frc;
    }
    RCCL_TRY(g_rccl.GroupStart());
    // The group is ENDED on every path: a call that fails inside it (RCCL records the error, ncclGroupEnd then drops what
    // was queued and returns it) must not leave the process with an open group that swallows every later RCCL call.
    // A failed collective can leave a communicator unusable: the cached ones are destroyed, the next call initialises new ones.
    ncclResult_t bad = ncclSuccess;
    const char *what = nullptr;
    const bool fault = g_rccl.fault != 0;      // tests only (ort_debug_fault_allreduce): armed for ONE call
    g_rccl.fault = 0;
    for (int i = 0; i < n && bad == ncclSuccess; ++i) {
        ort_ctx *c = ctxs[i];
        bad = g_rccl.AllReduce(c->d_image, c->d_image, ORT_IMAGE_BINS, (fault && i == 0) ? (ncclDataType_t)99 : ncclInt32, ncclSum, g_rccl.comms[i], c->stream);
        if (bad != ncclSuccess) { what = "ncclAllReduce(image)"; break; }
        bad = g_rccl.AllReduce(c->d_counters, c->d_counters, ORT_NUM_COUNTERS, ncclUint64, ncclSum, g_rccl.comms[i], c->stream);
        if (bad != ncclSuccess) what = "ncclAllReduce(counters)";
    }
    const ncclResult_t end = g_rccl.GroupEnd();
    if (bad != ncclSuccess || end != ncclSuccess) rccl_drop();
    if (bad != ncclSuccess) return rccl_fail(what, bad);
    if (end != ncclSuccess) return rccl_fail("ncclGroupEnd", end);
    return ORT_OK;
}

int ort_comm_destroy(void)
{
    if (g_rccl.lib) rccl_drop();
    return ORT_OK;
}

int ort_allreduce_ranks(int *n_ranks)
{
    if (!n_ranks) return fail(ORT_E_INVALID, "NULL argument");
    *n_ranks = 0;
    if (!g_rccl.lib || g_rccl.n == 0) return ORT_OK;          // no ort_allreduce yet
    int n = 0;
    RCCL_TRY(g_rccl.CommCount(g_rccl.comms[0], &n));            // what RCCL itself says, not what it was asked for
    *n_ranks = n;
    return ORT_OK;
}

int ort_device_image(ort_ctx *c, void **d_image)
{
    if (!c || !d_image) return fail(ORT_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = flush_replicas(c); if (rc) return rc; }
    *d_image = c->d_image;
    return ORT_OK;
}

int ort_device_counters(ort_ctx *c, void **d_counters)
{
    if (!c || !d_counters) return fail(ORT_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = flush_replicas(c); if (rc) return rc; }     // the deferred rays of an open group are counted by its re-run
    *d_counters = c->d_counters;
    return ORT_OK;
}

int ort_work_counters(ort_ctx *c, uint64_t *work)
{
    if (!c || !work) return fail(ORT_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = close_group(c); if (rc) return rc; }
    HIP_TRY(hipMemcpyAsync(work, c->d_work, ORT_NUM_WORK * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ORT_OK;
}

int ort_synchronize(ort_ctx *c)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = flush_replicas(c); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ORT_OK;
}

int ort_set_kernel_variant(ort_ctx *c, int variant)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    if (variant < 0 || variant > 127) return fail(ORT_E_INVALID, "variant must be in 0..127");
    if ((variant & 64) && c->precision != 0)
        return fail(ORT_E_INVALID, "kernel variant bit 6 (strict libm emitters) belongs to the exact fp64 path: ort_set_precision(0) first");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = close_group(c); if (rc) return rc; }
    c->variant = variant;
    return ORT_OK;
}

int ort_kernel_times(ort_ctx *c, float *ms, int capacity, int *count)
{
    if (!c || !ms || !count || capacity < 0) return fail(ORT_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long have = c->ring_count < (unsigned long long)kTimingRing ? c->ring_count : kTimingRing;
    int n = (int)(have < (unsigned long long)capacity ? have : (unsigned long long)capacity);
    for (int i = 0; i < n; ++i) {                    // oldest of the kept launches first
        int slot = (int)((c->ring_count - (unsigned long long)n + i) % kTimingRing);
        HIP_TRY(hipEventSynchronize(c->ring[slot][1]));
        HIP_TRY(hipEventElapsedTime(&ms[i], c->ring[slot][0], c->ring[slot][1]));
    }
    *count = n;
    return ORT_OK;
}

int ort_set_precision(ort_ctx *c, int precision)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    if (precision < 0 || precision > 2) return fail(ORT_E_INVALID, "precision must be 0 (fp64 exact), 1 (fp32) or 2 (fp64 fast)");
    if (precision != 0 && (c->variant & 64))
        return fail(ORT_E_INVALID, "strict libm emitters (kernel variant bit 6) are set: they belong to the exact fp64 path — clear the bit first");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = close_group(c); if (rc) return rc; }     // the re-run uses the arithmetic of its launches
    c->precision = precision;
    return ORT_OK;
}

int ort_last_kernel_name(ort_ctx *c, char *buf, int capacity)
{
    if (!c || !buf || capacity < 1) return fail(ORT_E_INVALID, "bad argument");
    snprintf(buf, (size_t)capacity, "%s", c->last_kernel);
    return ORT_OK;
}

int ort_set_timing(ort_ctx *c, int enable)
{
    if (!c) return fail(ORT_E_INVALID, "ctx is NULL");
    c->timing = enable != 0;
    return ORT_OK;
}

int ort_last_kernel_ms(ort_ctx *c, int kind, float *ms)
{
    if (!c || !ms) return fail(ORT_E_INVALID, "NULL argument");
    if (kind < 0 || kind > 2) return fail(ORT_E_INVALID, "kind must be 0, 1 or 2");
    *ms = -1.f;
    if (!c->ev_valid[kind]) return ORT_OK;
    HIP_TRY(hipSetDevice(c->device));
    hipEvent_t e0 = c->ev[kind][0], e1 = c->ev[kind][1];
    if (kind == 0) {                                   // the most recent slot of the ring
        const int slot = (int)((c->ring_count - 1) % kTimingRing);
        e0 = c->ring[slot][0]; e1 = c->ring[slot][1];
    }
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(ms, e0, e1));
    return ORT_OK;
}

}  // extern "C"

// TEST HOOK, not part of include/ort.h: the next ort_allreduce of this process hands its first collective an invalid
// datatype (tests/test_gpu_distributed.py: the group is ended, the error is reported, the call after it works)
extern "C" int ort_debug_fault_allreduce(int arm)
{
    g_rccl.fault = arm != 0;
    return ORT_OK;
}

// DEVELOPMENT HOOK, not part of include/ort.h (tools/expbench.py): arm one of ort_k_exp.hip's experiments on the fused
// point program of this context (which = 0: none)
extern "C" int ort_debug_set_exp(ort_ctx *c, int which, int static_pct, int pull_min, int pull_max, int wg_per_cu)
{
    if (!c || which < 0 || which > 5 || static_pct < 0 || static_pct > 100 || pull_min < 1 || pull_max < pull_min || wg_per_cu < 1 || wg_per_cu > 8)
        return fail(ORT_E_INVALID, "bad experiment");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = close_group(c); if (rc) return rc; }
    c->exp_which = which; c->exp_static_pct = static_pct; c->exp_min = pull_min; c->exp_max = pull_max; c->exp_wg_per_cu = wg_per_cu;
    return ORT_OK;
}

#ifdef ORT_DBG_RARE
// development build only: per-site counts of raised `rare` flags since the library was loaded
extern "C" int ort_debug_rare(unsigned long long out[16])
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ort::ort_dbg_rare), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

#ifdef ORT_SCAT_TIMING
extern "C" int ort_debug_scat_times(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scat_times), (size_t)n * 4 * sizeof(unsigned long long));
}
#endif
