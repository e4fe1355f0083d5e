// ort_hip.hip — HIP kernels (gfx950 / CDNA4, wave64) and the C ABI of include/ort.h.
//
// Kernels (device functions: ort_device.h; the arithmetic type T is double = the reference's
// arithmetic, bit-exact; float = fp32 study path; fastd = opt-in fast fp64, ort_fastd.h)
//   trace_queue_kernel<MODE, FILT, EXT, T, PROG>   the production kernel.  One wavefront = one ray
//       bundle over contiguous ranges of global ray indices (plan_ranges); a ray lives in VGPRs
//       from emission to binning; survivors of the first surface segment are compacted through a
//       wave-private LDS queue so that the second segment runs on full wavefronts; hits are
//       binned with global int32 atomics into one of 8 image replicas (fold_kernel adds them
//       into the image when it is next needed); counters are reduced per workgroup.
//         MODE_FUSED     emit in-kernel (src/main.f90:90-109 / :127-162 whole loop body)
//         MODE_RESIDENT  ray bundle read from HBM, SoA fp64 [6][n], coalesced
//         FILT           filtered predicates (decisions from bounded approximations); a ray that
//                        lands inside a margin is not decided here: its index goes to the
//                        re-run list and it leaves the kernel without side effect
//         EXT (ANYSRC)   also compiles the rarely used emitters (spot, crs, image, isors); SCAT the
//                        in-bottle scattering walk (213+ VGPRs); the default instantiation
//                        leaves both out, and a phase is given only what its own list needs
//         PROG           the surface list as template constants (Prog<P>: the default point /
//                        ring systems, their iris variants, no bottle, elliptical bottle), steps
//                        unrolled, the system read through scalar loads from its device copy;
//                        the ring programs put a segment 0 in front (rays that are certain to
//                        miss the first aperture are counted, not emitted).  PROG_GENERIC stages
//                        the 3 KB ort_system into LDS once per workgroup and walks any list
//   trace_kernel<MODE, FILT, T, EXT>         plain lockstep thread-per-ray walk: the literal
//       re-run of the listed rays when a group of queued launches closes (normally an empty
//       list), the parity / debug entry (MODE_DEBUG: per-ray outputs, tracker paths, no side
//       effect) and the A/B baseline of the queued kernel
//   fold_kernel, emit_kernel
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

using namespace ort;

namespace {

constexpr int kBlock = 256;
#ifndef ORT_MIN_WAVES
#define ORT_MIN_WAVES 1
#endif
constexpr int kTimingRing = 64;         // launches kept by ort_kernel_times
constexpr int kReplicas = 8;            // image replicas, one per XCD
// A replica stores one layer in 2^18 slots; bin i lives in slot (i * kSlotMul) mod 2^18 (a bijection:
// the multiplier is odd), so neighbouring bins — the focal blob — land in unrelated 64-byte lines.
constexpr int kSlotBits = 18;
constexpr uint32_t kSlots = 1u << kSlotBits;
constexpr uint32_t kSlotMul = 0x379B1u, kSlotMulInv = 0x32F51u;
static_assert(((kSlotMul * kSlotMulInv) & (kSlots - 1)) == 1u, "kSlotMulInv must invert kSlotMul modulo 2^18");
static_assert(ORT_IMAGE_N * ORT_IMAGE_N <= (int)kSlots, "a layer must fit the slot table");
constexpr size_t kReplicaInts = 2 * (size_t)kSlots;   // both layers of one replica
constexpr int kMaxBlocks = 256 * 12;    // grid cap of the lockstep kernels and of small queued launches (equal ranges)
constexpr uint64_t kChunkRaysMax = ORT_MAX_RAYS_PER_LAUNCH;   // rays per launch of the queued kernel (bounds the re-run list: 4 B per ray); see chunk_rays()
constexpr int kRedoBlocks = 128;        // grid of the literal re-run kernel (it normally finds an empty list and returns)

enum { MODE_FUSED = 0, MODE_RESIDENT = 1, MODE_DEBUG = 2, MODE_CONTINUE = 3 };

struct TraceArgs {
    const ort_system *sys;       // device copy (DevSystem.sys)
    const SurfAuxT<double> *aux; // DevSystem.aux[phase - 1]: read by the program kernels through scalar loads
    const SystemT<float> *sysf;  // the same system and constants in single precision (fp32 path, program kernels)
    const SurfAuxT<float> *auxf;
    int32_t *image;              // [2][401][401]
    int32_t *replicas;           // [kReplicas][2][kSlots] or null: see bin_hit, fold_kernel
    unsigned long long *counters;
    unsigned long long *work;    // [ORT_NUM_WORK]: executed-work counters (ort_work_counters), not part of the result
    uint64_t first_ray, n_rays, rng_base;
    int phase, draw_base;
    // ray ranges of the queued kernel's waves (host: plan_ranges): workgroups [0, head_blocks) cut
    // rays [0, head_rays) into head_chunk per wave, the rest cut the remainder into tail_chunk per wave
    uint32_t head_blocks;
    uint64_t head_rays, head_chunk, tail_chunk;
    // the re-run list of the queued filtered kernel (see trace_queue_kernel): ray indices relative
    // to first_ray; ctl[0] = entries, ctl[1] = workgroups of the re-run kernel that are done
    uint32_t *redo_list;
    unsigned int *redo_ctl;
    // ring programs (trace_queue_kernel, segment 0): a ring ray whose lens-disc sample rr (its third draw x
    // ring_lens_r2) exceeds the host's threshold (ring_cull_threshold) misses the plano-convex aperture for certain and is
    // counted without being emitted.  rr does not decrease with the draw's 32-bit word, so the test is taken on the word:
    // the ray dies iff word > cull_word (fp32 arithmetic: cull_wordf; host: cull_word_of); 0xffffffff: nothing is culled
    uint32_t cull_word, cull_wordf;
    int strict;                  // kernel variant bit 6: the emitters call glibc's own sin / cos / sincos (ort_device.h: sincos_em)
    int wide;                    // kernel variant bit 5: 53-bit draws (ORT-RNG-v2w, ort_device.h); lockstep kernel only
    uint64_t defer_base;         // queued kernel: list entry of ray i of this launch = defer_base + i (the entries of all
                                 // launches of a group are relative to the group's first ray, see close_group)
    int listed;                  // trace_kernel: iterate redo_list instead of [0, n_rays)
    // resident / debug inputs
    uint64_t in_stride;          // component stride of pos_dir_in (the bundle's ray count)
    const double *pos_dir_in;    // SoA [6][in_stride] or null
    const double *u;             // [nu][n] or null
    int nu;
    // debug outputs (any may be null)
    double *pos_dir_out, *emitted_out;
    int32_t *status, *bin_xy, *n_isect, *n_draws;
    // scattering pipeline (scatter_front_kernel -> trace_queue_kernel<MODE_CONTINUE>): the rays that leave the
    // scattering surfaces alive: state SoA [6][cont_cap], keyed draw counter (ray << 24) + draws consumed
    // (kNoRay: the slot holds no ray), intersections evaluated so far.  A wave of the front kernel fills the slots
    // of its own ray range from the bottom and marks the rest empty: no shared counter (one returning atomic per
    // hand-over on ONE address serialised the whole kernel: 1.4 ms per 4e6 rays, 73 % of the wave cycles waiting)
    // A ray is handed over when it ARRIVES at the last scattering wall: the continuation starts with the rest of that
    // surface's step (back test, move by cont_t, normal, Fresnel) in its own filtered arithmetic.
    double *cont_pos_dir;
    double *cont_t;
    uint64_t *cont_draw;
    int32_t *cont_nis;
    uint64_t cont_cap;
    int cont_k0;                 // first surface of the continuation (= last scattering surface + 1)
    // work distribution of scatter_front_kernel (kScatCtlWords words, zero before every launch): eight heads, one per
    // XCD (head x hands out the rays [x scat_share, (x + 1) scat_share) of the launch, scat_grab at a time), and the
    // count of hand-over slots allocated so far (what the continuation walks)
    unsigned long long *scat_ctl;
    uint64_t scat_share;
    uint32_t scat_grab;
    const long long *img_cdf;    // image-source table or null
    // fp32 queued kernels: hits are LOGGED, not binned (see bin_log_kernel).  The log has kBinTiles parts of hit_stride
    // 16-bit entries, part t for the bins with bin % kBinTiles == t (entry = bin / kBinTiles).  A wave writes its hits to
    // the bottom of its own region of each part — entries [lo, lo + hits_t) of its ray range [lo, hi): a ray ends at
    // most once — and leaves hits_0 .. hits_4 and lo in its eight words of the directory
    uint16_t *hit_log;
    uint64_t hit_stride;
    uint32_t hit_base;           // this launch's first entry in every part (the log holds several launches: bin_pending)
    uint32_t *hit_dir;           // this launch's first directory entry
    double *path;                // [n][ORT_MAX_PATH][3] or null (tracker)
    int32_t *npath;
};

// cooperative copy of the system into LDS, 8 bytes per thread per pass
__device__ inline void stage_system(ort_system &dst, const ort_system *src)
{
    constexpr int words = sizeof(ort_system) / 8;
    static_assert(sizeof(ort_system) % 8 == 0, "ort_system must be a whole number of 8-byte words");
    const uint64_t *s = reinterpret_cast<const uint64_t *>(src);
    uint64_t *d = reinterpret_cast<uint64_t *>(&dst);
    for (int i = threadIdx.x; i < words; i += blockDim.x) d[i] = s[i];
    __syncthreads();
}

// fp32 path: the same system in single precision (each value rounded to nearest once)
__host__ __device__ inline void convert_surface(SurfaceT<float> &b, const ort_surface &a)
{
    b.cx = (float)a.cx; b.cy = (float)a.cy; b.cz = (float)a.cz; b.radius = (float)a.radius;
    b.radius_b = (float)a.radius_b; b.n1 = (float)a.n1; b.n2 = (float)a.n2; b.eta = (float)a.eta;
    b.aperture = (float)a.aperture; b.kind = a.kind; b.flags = a.flags;
    b.mua = (float)a.mua; b.mus = (float)a.mus; b.hgg = (float)a.hgg; b.scat_radius = (float)a.scat_radius;
}
__host__ __device__ inline void convert_globals(SystemT<float> &dst, const ort_system &src)
{
    dst.n_surfaces[0] = src.n_surfaces[0]; dst.n_surfaces[1] = src.n_surfaces[1];
    dst.split[0] = src.split[0]; dst.split[1] = src.split[1];
    dst.ring_ellipse = src.ring_ellipse; dst.pad = 0;
    dst.cos_theta_max = (float)src.cos_theta_max;
    dst.ring_r1 = (float)src.ring_r1; dst.ring_r2 = (float)src.ring_r2;
    dst.ring_lens_r2 = (float)src.ring_lens_r2; dst.ring_lens_z = (float)src.ring_lens_z;
    dst.ring_bottle_ra = (float)src.ring_bottle_ra; dst.ring_bottle_rb = (float)src.ring_bottle_rb;
    dst.ring_bottle_z = (float)src.ring_bottle_z;
    dst.bin_width = (float)src.bin_width; dst.inv_bin_width = (float)src.inv_bin_width;
    dst.na_cos_min = (float)src.na_cos_min; dst.twopi = (float)src.twopi;
    dst.spot_dphi = (float)src.spot_dphi; dst.spot_dtheta = (float)src.spot_dtheta;
    dst.crs_sigma = (float)src.crs_sigma; dst.crs_radius = (float)src.crs_radius;
    dst.crs_cy = (float)src.crs_cy; dst.crs_cz = (float)src.crs_cz;
    dst.img_lens_r2 = (float)src.img_lens_r2; dst.img_lens_z = (float)src.img_lens_z;
    dst.point_offset = (float)src.point_offset;
    dst.isors_sigma = (float)src.isors_sigma; dst.isors_k = (float)src.isors_k; dst.isors_height = (float)src.isors_height;
    dst.isors_base_pos = (float)src.isors_base_pos; dst.isors_z = (float)src.isors_z;
    dst.isors_rad1 = (float)src.isors_rad1; dst.isors_rad2 = (float)src.isors_rad2;
    dst.isors_cy = (float)src.isors_cy; dst.isors_cz = (float)src.isors_cz;
    dst.isors_lens_r2 = (float)src.isors_lens_r2; dst.isors_lens_z = (float)src.isors_lens_z;
    dst.emitter[0] = src.emitter[0]; dst.emitter[1] = src.emitter[1];
}
inline void convert_system(SystemT<float> &dst, const ort_system &src)      // host
{
    for (int i = 0; i < 2 * ORT_MAX_SURFACES; ++i)
        convert_surface(dst.surfaces[i / ORT_MAX_SURFACES][i % ORT_MAX_SURFACES], src.surfaces[i / ORT_MAX_SURFACES][i % ORT_MAX_SURFACES]);
    convert_globals(dst, src);
}
// ... and converted once per workgroup while staging (generic kernels)
__device__ inline void stage_system(SystemT<float> &dst, const ort_system *src)
{
    for (int i = threadIdx.x; i < 2 * ORT_MAX_SURFACES; i += blockDim.x)
        convert_surface(dst.surfaces[i / ORT_MAX_SURFACES][i % ORT_MAX_SURFACES], src->surfaces[i / ORT_MAX_SURFACES][i % ORT_MAX_SURFACES]);
    if (threadIdx.x == 0) convert_globals(dst, *src);
    __syncthreads();
}

// the phase's per-surface derived constants (ort_device.h: SurfAuxT), one thread per surface
template <class T, class Surf>
__device__ inline void stage_aux(SurfAuxT<T> *aux, const Surf *surf, int ns)
{
    if ((int)threadIdx.x < ns) aux[threadIdx.x] = make_aux<T>(surf[threadIdx.x]);
    __syncthreads();
}

// Where this workgroup bins its hits.  With one image, every wave of the chip queues its atomics
// on the same few thousand 64-byte lines of the focal blob (measured in round 1: +0.19 ms on a
// 0.75 ms launch).  So the hits go to one of kReplicas private copies, one per XCD, read from the
// hardware (HW_REG_XCC_ID; blockIdx % 8 only says which blocks share an XCD while the placement is
// round-robin, which it is not while the tail of a previous kernel occupies some XCDs).
// A hit is a no-return atomic that the L2 forwards to the memory side (the 8 L2s are not coherent
// with each other: TCC_EA0_ATOMIC counts one request, and one 32-byte DRAM write, per hit).  The
// blob is ~17 000 bins, half of the hits on 1000 of them: stored row by row that is ~100 very hot
// lines per replica, and how those happen to fall on the memory channels decided the launch time
// — the same kernel took 0.40 or 0.48 ms (fast fp64: 0.34 - 0.55 ms) depending on where the
// context's buffers lay (TCC_EA0_WRREQ_STALL x 3 in the slow placements).  So within a replica the
// bins are scattered over 2^18 slots by a multiplicative hash: every hot bin gets a line of its own
// and the load spreads over all channels, whatever the placement.  fold_kernel undoes the hash.
// Integer adds commute: the image is bit-identical either way.
__device__ inline int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7; }   // hwreg(HW_REG_XCC_ID, 0, 4)

__device__ inline void bin_hit(int32_t *layer, int xp, int yp, bool replicated)
{
    const uint32_t bin = (uint32_t)((xp + 200) + ORT_IMAGE_N * (yp + 200));         // imageMod.f90:55-56
#ifdef ORT_DEV_NO_BIN                                                                // A/B build: what the image atomics cost
    if (xp != 0x7fffffff) return;
#endif
    atomicAdd(&layer[replicated ? (bin * kSlotMul) & (kSlots - 1) : bin], 1);
}

__device__ inline int32_t *hist_layer(const TraceArgs &a)
{
    if (a.replicas) return a.replicas + (size_t)xcc_id() * kReplicaInts + (size_t)(a.phase - 1) * kSlots;
    return a.image + (size_t)(a.phase - 1) * ORT_IMAGE_N * ORT_IMAGE_N;
}

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
// Integer adds commute: the image equals the one the atomics would have produced, bit for bit.
constexpr int kBinUnits = 51, kBinTiles = 5, kBinTile = (ORT_IMAGE_N * ORT_IMAGE_N + kBinTiles - 1) / kBinTiles, kBinDirMax = 2048;
constexpr int kBinThreads = 1024, kBinDirWords = 8;
static_assert(kBinTile <= 65536, "a log entry (bin / kBinTiles) must fit 16 bits");
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

// One lockstep pass of a wave over surfaces [k0, k1): every lane steps with its `st`
// predicate; the loop leaves as soon as no lane of the wave is alive (uniform branch).
// KEEP: see surface_step — false where only st/xp/yp/nis of an ended ray are read afterwards.
template <bool FILT, class T, bool EXT, bool KEEP, class Sys, class Surf, class D>
__device__ inline void walk_pass(const Sys &S, const Surf *surf, const SurfAuxT<T> *aux, int k0, int k1,
                                 RayT<T> &r, D &draws, int &nis, int &st, int &xp, int &yp, bool &rare)
{
    for (int k = k0; k < k1; ++k) {
        if (!wave_any_live(st)) break;
#ifdef ORT_DBG_RARE
        const bool before = rare;
#endif
        surface_step<FILT, T, EXT, KEEP>(S, surf[k], aux[k], r, draws, nis, st, xp, yp, rare);
#ifdef ORT_DBG_RARE
        if (rare && !before && k < 8) atomicAdd(&ort::ort_dbg_rare[8 + k], 1ull);   // first raise, by surface
#endif
    }
}

// What a context keeps on the device: the system as the caller staged it, plus the derived
// per-surface constants (SurfAuxT; formed on the host by the same IEEE operations).
struct DevSystem {
    ort_system sys;
    SurfAuxT<double> aux[2][ORT_MAX_SURFACES];
    SystemT<float> sysf;                         // fp32 path: converted once on the host (round to nearest,
    SurfAuxT<float> auxf[2][ORT_MAX_SURFACES];   // as v_cvt_f32_f64 would), constants formed in fp32
};

// The program kernels know each step's surface index at compile time, so they read its record
// straight from the device copy through the CONSTANT address space: uniform address + constant
// memory = scalar loads (s_load_dwordx*) into SGPRs.  The values then feed the vector
// instructions as scalar operands instead of occupying VGPRs (an LDS read lands in VGPRs), which
// is what lets the unrolled kernel fit 128 VGPRs without spilling.
template <class T> struct ConstPtrs {       // fp64 and fast fp64 read the fp64 records
    typedef const __attribute__((address_space(4))) ort_system *sys_t;
    typedef const __attribute__((address_space(4))) ort_surface *surf_t;
    typedef const __attribute__((address_space(4))) SurfAuxT<double> *aux_t;
    typedef ort_surface Surf;
};
template <> struct ConstPtrs<float> {
    typedef const __attribute__((address_space(4))) SystemT<float> *sys_t;
    typedef const __attribute__((address_space(4))) SurfaceT<float> *surf_t;
    typedef const __attribute__((address_space(4))) SurfAuxT<float> *aux_t;
    typedef SurfaceT<float> Surf;
};

template <class T>
__device__ inline typename ConstPtrs<T>::Surf load_surface(typename ConstPtrs<T>::surf_t p)
{
    typename ConstPtrs<T>::Surf s;
    s.cx = p->cx; s.cy = p->cy; s.cz = p->cz; s.radius = p->radius; s.radius_b = p->radius_b;
    s.n1 = p->n1; s.n2 = p->n2; s.eta = p->eta; s.aperture = p->aperture;
    s.mua = p->mua; s.mus = p->mus; s.hgg = p->hgg; s.scat_radius = p->scat_radius;
    s.kind = p->kind; s.flags = p->flags;
    return s;
}

template <class T>
__device__ inline SurfAuxT<T> load_aux(typename ConstPtrs<T>::aux_t p)
{
    SurfAuxT<T> a;
    a.r2 = T(p->r2); a.ap2 = T(p->ap2); a.ap_tol = T(p->ap_tol); a.eta2 = T(p->eta2);
    a.ell_sa = T(p->ell_sa); a.ell_sb = T(p->ell_sb);
    a.rh = T(p->rh); a.rk = T(p->rk); a.r2_tol = T(p->r2_tol);
    a.ap_lo = T(p->ap_lo); a.ap_hi = T(p->ap_hi);
    a.ax_ly = T(p->ax_ly); a.ax_lz = T(p->ax_lz); a.ax_c = T(p->ax_c);      // (read by step 0 of the point programs only)
    return a;
}

// Surface programs known at compile time.  The reference's two loops walk a fixed list in
// their default set-up (bottle present, no iris, circular bottle): with the kinds, flags and
// aperture presence as template constants the per-step dispatch (readfirstlane + scalar
// branches + the blocks they cut the schedule into) disappears and the steps are laid out
// back to back: -8 % kernel time.  The host selects a program only when the staged system
// matches it field for field (match_program); everything else runs the generic walk.
// A program = a surface LIST (low four bits) + the light source in front of it (bits 4..): 0 the phase's default emitter
// (ring / point), SRC_* another one.  Every list is instantiated with every source its phase has.
enum { PROG_GENERIC = 0, PROG_POINT, PROG_RING, PROG_POINT_IRIS_B, PROG_POINT_IRIS_A, PROG_RING_IRIS_B, PROG_RING_IRIS_A,
       PROG_POINT_BARE, PROG_POINT_ELLIPSE, PROG_LIST_MASK = 15 };
enum { SRC_CRS = 1 << 4, SRC_ISORS = 2 << 4, SRC_IMAGE = 3 << 4, SRC_HANDED_OVER = 4 << 4 };
constexpr int PROG_CRS = PROG_RING | SRC_CRS, PROG_ISORS = PROG_RING | SRC_ISORS, PROG_IMAGE = PROG_POINT | SRC_IMAGE,
              PROG_POINT_WALKED = PROG_POINT | SRC_HANDED_OVER;
// every list with its phase's default emitter (ring / point): X(name) — instantiated fused and resident, in every arithmetic
#define ORT_RING_LISTS(X, S) X(PROG_RING | S) X(PROG_RING_IRIS_B | S) X(PROG_RING_IRIS_A | S)
#define ORT_POINT_LISTS(X, S) X(PROG_POINT | S) X(PROG_POINT_IRIS_B | S) X(PROG_POINT_IRIS_A | S) X(PROG_POINT_BARE | S) X(PROG_POINT_ELLIPSE | S)
#define ORT_PROGRAMS(X) ORT_POINT_LISTS(X, 0) ORT_RING_LISTS(X, 0)
// ... and with the other bulk light sources (runner.py's crs / iSORS / Bessel-image experiments, with and without an iris):
// crs and isors in front of the ring loop's lists, the image source in front of the point loop's; fused, fp64 and fp32
// (resident bundles and fast fp64 of those sources run the generic walk)
#define ORT_SOURCE_PROGRAMS(X) ORT_RING_LISTS(X, SRC_CRS) ORT_RING_LISTS(X, SRC_ISORS) ORT_POINT_LISTS(X, SRC_IMAGE)

namespace prog {
constexpr int CYL = ORT_SURF_CYLINDER, ELL = ORT_SURF_ELLIPSE, PLN = ORT_SURF_PLANE, SPH = ORT_SURF_SPHERE, IRS = ORT_SURF_IRIS, IMG = ORT_SURF_IMAGE;
constexpr int SK = ORT_F_SKIP_ON_REFLECT, BT = ORT_F_BOTTLE | ORT_F_SKIP_ON_REFLECT, H3 = ORT_F_SKIP_ON_REFLECT | ORT_F_MISS_IS_HELP3;
}
template <int P> struct Prog;
// point loop, src/main.f90:127-162: bottle (2 cylinders), plano-convex (flat, curved), doublet (3 faces), image
template <> struct Prog<PROG_POINT> {
    static constexpr int phase = 2, n = 8, split = 5;
    static constexpr int emitter = ORT_EMIT_POINT;
    static constexpr int kind[n] = {prog::CYL, prog::CYL, prog::PLN, prog::SPH, prog::SPH, prog::SPH, prog::SPH, prog::IMG};
    static constexpr int flags[n] = {prog::BT, prog::BT, 0, prog::SK, prog::SK, prog::SK, prog::H3, 0};
    static constexpr int ap[n] = {0, 0, 1, 0, 1, 0, 0, 0};
};
// ring loop, src/main.f90:90-109: plano-convex, doublet, image
template <> struct Prog<PROG_RING> {
    static constexpr int phase = 1, n = 6, split = 1;
    static constexpr int emitter = ORT_EMIT_RING;
    static constexpr int kind[n] = {prog::PLN, prog::SPH, prog::SPH, prog::SPH, prog::SPH, prog::IMG};
    static constexpr int flags[n] = {0, prog::SK, prog::SK, prog::SK, prog::H3, 0};
    static constexpr int ap[n] = {1, 0, 1, 0, 0, 0};
};
// the same with the iris in front of the doublet (src/lens.f90:551-565) ...
template <> struct Prog<PROG_POINT_IRIS_B> {
    static constexpr int phase = 2, n = 9, split = 6;
    static constexpr int emitter = ORT_EMIT_POINT;
    static constexpr int kind[n] = {prog::CYL, prog::CYL, prog::PLN, prog::SPH, prog::IRS, prog::SPH, prog::SPH, prog::SPH, prog::IMG};
    static constexpr int flags[n] = {prog::BT, prog::BT, 0, prog::SK, 0, prog::SK, prog::SK, prog::H3, 0};
    static constexpr int ap[n] = {0, 0, 1, 0, 1, 1, 0, 0, 0};
};
template <> struct Prog<PROG_RING_IRIS_B> {
    static constexpr int phase = 1, n = 7, split = 1;
    static constexpr int emitter = ORT_EMIT_RING;
    static constexpr int kind[n] = {prog::PLN, prog::SPH, prog::IRS, prog::SPH, prog::SPH, prog::SPH, prog::IMG};
    static constexpr int flags[n] = {0, prog::SK, 0, prog::SK, prog::SK, prog::H3, 0};
    static constexpr int ap[n] = {1, 0, 1, 1, 0, 0, 0};
};
// ... and behind it (src/lens.f90:632-644)
template <> struct Prog<PROG_POINT_IRIS_A> {
    static constexpr int phase = 2, n = 9, split = 5;
    static constexpr int emitter = ORT_EMIT_POINT;
    static constexpr int kind[n] = {prog::CYL, prog::CYL, prog::PLN, prog::SPH, prog::SPH, prog::SPH, prog::SPH, prog::IRS, prog::IMG};
    static constexpr int flags[n] = {prog::BT, prog::BT, 0, prog::SK, prog::SK, prog::SK, prog::H3, 0, 0};
    static constexpr int ap[n] = {0, 0, 1, 0, 1, 0, 0, 1, 0};
};
template <> struct Prog<PROG_RING_IRIS_A> {
    static constexpr int phase = 1, n = 7, split = 1;
    static constexpr int emitter = ORT_EMIT_RING;
    static constexpr int kind[n] = {prog::PLN, prog::SPH, prog::SPH, prog::SPH, prog::SPH, prog::IRS, prog::IMG};
    static constexpr int flags[n] = {0, prog::SK, prog::SK, prog::SK, prog::H3, 0, 0};
    static constexpr int ap[n] = {1, 0, 1, 0, 0, 1, 0};
};

// the point loop without the bottle (use_bottle = false, src/main.f90:147)
template <> struct Prog<PROG_POINT_BARE> {
    static constexpr int phase = 2, n = 6, split = 3;
    static constexpr int emitter = ORT_EMIT_POINT;
    static constexpr int kind[n] = {prog::PLN, prog::SPH, prog::SPH, prog::SPH, prog::SPH, prog::IMG};
    static constexpr int flags[n] = {0, prog::SK, prog::SK, prog::SK, prog::H3, 0};
    static constexpr int ap[n] = {1, 0, 1, 0, 0, 0};
};

// the point loop through an elliptical bottle (src/lens.f90:221-225)
template <> struct Prog<PROG_POINT_ELLIPSE> {
    static constexpr int phase = 2, n = 8, split = 5;
    static constexpr int emitter = ORT_EMIT_POINT;
    static constexpr int kind[n] = {prog::ELL, prog::ELL, prog::PLN, prog::SPH, prog::SPH, prog::SPH, prog::SPH, prog::IMG};
    static constexpr int flags[n] = {prog::BT, prog::BT, 0, prog::SK, prog::SK, prog::SK, prog::H3, 0};
    static constexpr int ap[n] = {0, 0, 1, 0, 1, 0, 0, 0};
};

// a list behind another light source: the crs source (point_on_bottle, src/sourceMod.f90:50-89, src/main.f90:99), the isors
// source (iSORS, :162-247, main.f90:97), the image source (emit_image, :303-361, main.f90:133) — or, SRC_HANDED_OVER, behind
// a scattering bottle (trace_queue_kernel<MODE_CONTINUE>: rays handed over by scatter_front_kernel, each at its own draw;
// whatever source emitted them)
constexpr int ORT_EMIT_HANDED_OVER = -2;
constexpr int source_emitter(int src)
{
    return src == SRC_CRS ? ORT_EMIT_CRS : src == SRC_ISORS ? ORT_EMIT_ISORS : src == SRC_IMAGE ? ORT_EMIT_IMAGE : ORT_EMIT_HANDED_OVER;
}
template <int P> struct Prog : Prog<(P & PROG_LIST_MASK)> {
    static_assert(P > PROG_LIST_MASK, "a list without a Prog<> specialisation");
    static constexpr int emitter = source_emitter(P & ~PROG_LIST_MASK);
};

template <int P> constexpr bool prog_is_ring()             // phase-1 list (plano-convex first)
{
    if constexpr (P == PROG_GENERIC) return false;
    else return Prog<P>::phase == 1;
}
// segment 0 (the cull by the third draw) belongs to the ring EMITTER
template <int P> constexpr bool prog_culls()
{
    if constexpr (P == PROG_GENERIC) return false;
    else return Prog<P>::phase == 1 && Prog<P>::emitter == ORT_EMIT_RING;
}
// does every ray reach step K at the same draw index?  ring (4 draws), point (2), image (4): yes; crs and isors
// draw a variable number (polar Box-Muller, src/random_mod.f90:59-85): their steps count draws per lane
template <int P> constexpr bool prog_static_draws()
{
    if constexpr (P == PROG_GENERIC) return false;
    else return Prog<P>::emitter == ORT_EMIT_RING || Prog<P>::emitter == ORT_EMIT_POINT || Prog<P>::emitter == ORT_EMIT_IMAGE;
}
// the rays of the program start where `point` puts them, in front of a circular cylinder (OPT_AXIAL_START)
template <int P> constexpr bool prog_starts_on_axis()
{
    if constexpr (P == PROG_GENERIC) return false;
    else return Prog<P>::emitter == ORT_EMIT_POINT && Prog<P>::kind[0] == ORT_SURF_CYLINDER;
}

// the step the queue point of trace_queue_kernel lies in, + 1.  Prog<P>::split (the host's choice for
// the list: behind the stop that removes most rays) — except in the fused ring programs: their
// segment 0 has already removed the rays the first stop would, every ray that reaches segment 1
// passes it, and the next stop (the doublet's first face, step 2) ends nearly all of them
template <int P, int MODE> constexpr int queue_step()
{
    if constexpr (P == PROG_GENERIC) return 0;
    else if constexpr (prog_is_ring<P>() && MODE == MODE_FUSED) {
        for (int k = 1; k < Prog<P>::n; ++k)
            if (Prog<P>::ap[k] != 0 && Prog<P>::kind[k] != ORT_SURF_IRIS && Prog<P>::kind[k] != ORT_SURF_IMAGE) return k + 1;
        return Prog<P>::split;
    } else return Prog<P>::split;
}

// draws a ray has consumed before step K of program P: the emitter's (point 2, ring 4:
// src/sourceMod.f90:31-37, :266-286) plus one per refracting surface passed (an iris and the image
// plane draw nothing)
template <int P> constexpr int draw_index(int K)
{
    if (!prog_static_draws<P>()) return -1;              // surface_step DK < 0: per-lane draw counter
    int d = Prog<P>::emitter == ORT_EMIT_POINT ? 2 : 4;
    for (int j = 0; j < K; ++j)
        if (Prog<P>::kind[j] != ORT_SURF_IRIS && Prog<P>::kind[j] != ORT_SURF_IMAGE) d++;
    return d;
}

// steps [K, K1) of program P, each entered only while some lane of the wave is alive.  FRESH: no
// hash is at hand for the next odd draw (the walk starts behind the queue)
// OPT: what every step may assume (ort_device.h OPT_*); a step's status carries its intersection count
// (surface_step NISK)
// (OPT_COUNT_STEPS: the rays did not start at step 0 with a count of 0 — they are counted per step and lane)
constexpr int OPT_COUNT_STEPS = 8;
template <int K, int OPT = 0> constexpr int nisk() { return !(OPT & OPT_COUNT_STEPS) ? K + 1 : -1; }

template <bool FILT, class T, bool KEEP, int P, int K, int K1, bool FRESH, int OPT, class Sys, class D>
__device__ inline void walk_fixed(const Sys &S, typename ConstPtrs<T>::surf_t surf, typename ConstPtrs<T>::aux_t aux, RayT<T> &r, D &draws,
                                  int &nis, int &st, int &xp, int &yp, bool &rare)
{
    if constexpr (K < K1) {
        if (wave_any_live(st)) {
            const typename ConstPtrs<T>::Surf s = load_surface<T>(surf + K);
            const SurfAuxT<T> ax = load_aux<T>(aux + K);
#ifdef ORT_ISA_MARKERS      // tools/isa_budget.py: comment lines that delimit the steps in the listing
            asm volatile("; ORT_STEP_BEGIN %0" ::"n"(K));
#endif
            constexpr bool draws_here = Prog<P>::kind[K] != ORT_SURF_IRIS && Prog<P>::kind[K] != ORT_SURF_IMAGE;
            constexpr int OPTK = K == 0 ? OPT : (OPT & ~OPT_AXIAL_START);       // the emitter's position holds at the first surface only
            if constexpr (draws_here && Prog<P>::ap[K] != 0 && !KEEP) {
                // a refracting step with an aperture stop, in halves (surface_step PART): when the stop
                // ends every ray of the wavefront — the doublet's first face does that to 9 of 10
                // wavefronts of the ring loop — the normalisation and the Fresnel arithmetic are skipped
                surface_step<FILT, T, false, KEEP, Prog<P>::kind[K], Prog<P>::flags[K], Prog<P>::ap[K], draw_index<P>(K), FRESH, 1, nisk<K, OPT>(), OPTK>(
                    S, s, ax, r, draws, nis, st, xp, yp, rare);
                if (wave_any_live(st))
                    surface_step<FILT, T, false, KEEP, Prog<P>::kind[K], Prog<P>::flags[K], Prog<P>::ap[K], draw_index<P>(K), FRESH, 2, nisk<K, OPT>(), OPTK>(
                        S, s, ax, r, draws, nis, st, xp, yp, rare);
            } else {
                surface_step<FILT, T, false, KEEP, Prog<P>::kind[K], Prog<P>::flags[K], Prog<P>::ap[K], draw_index<P>(K), FRESH, 0, nisk<K, OPT>(), OPTK>(
                    S, s, ax, r, draws, nis, st, xp, yp, rare);
            }
#ifdef ORT_ISA_MARKERS
            asm volatile("; ORT_STEP_END %0" ::"n"(K));
#endif
            walk_fixed<FILT, T, KEEP, P, K + 1, K1, FRESH && !draws_here, OPT>(S, surf, aux, r, draws, nis, st, xp, yp, rare);
        }
    }
}

// one half (PART 1 / 2, ort_device.h: surface_step) of step K of program P: the step the queue point
// of trace_queue_kernel sits in
template <bool FILT, class T, int P, int K, int PART, int OPT, class Sys, class D>
__device__ inline void step_part(const Sys &S, typename ConstPtrs<T>::surf_t surf, typename ConstPtrs<T>::aux_t aux, RayT<T> &r, D &draws,
                                 int &nis, int &st, int &xp, int &yp, bool &rare)
{
    if (wave_any_live(st)) {
        const typename ConstPtrs<T>::Surf s = load_surface<T>(surf + K);
        const SurfAuxT<T> ax = load_aux<T>(aux + K);
#ifdef ORT_ISA_MARKERS
        if (PART == 1) asm volatile("; ORT_STEP_BEGIN %0" ::"n"(K));
#endif
        surface_step<FILT, T, false, false, Prog<P>::kind[K], Prog<P>::flags[K], Prog<P>::ap[K], draw_index<P>(K), PART == 2, PART, nisk<K, OPT>(), (K == 0 ? OPT : (OPT & ~OPT_AXIAL_START))>(
            S, s, ax, r, draws, nis, st, xp, yp, rare);
#ifdef ORT_ISA_MARKERS
        if (PART == 2) asm volatile("; ORT_STEP_END %0" ::"n"(K));
#endif
    }
}

// The segment [k0, k1) with the reference's outcome for every lane.  FILT: one pass with the
// filtered predicates; if any lane raised `rare` (ort_device.h) the wave runs the segment again
// from its initial state — `restore(r, draws, st)` re-creates it: reloaded or re-emitted, so no
// register is held for it through the hot pass — with the literal formulas, and the flagged
// lanes take that run's results.  One rare branch per segment instead of one per predicate.
template <bool FILT, class T, bool EXT, bool KEEP, class Sys, class Surf, class D, class Restore>
__device__ inline void walk(const Sys &S, const Surf *surf, const SurfAuxT<T> *aux, int k0, int k1,
                            RayT<T> &r, D &draws, int &nis, int &st, int &xp, int &yp, Restore restore)
{
    bool rare = false;
    if constexpr (!FILT) {
        walk_pass<false, T, EXT, KEEP>(S, surf, aux, k0, k1, r, draws, nis, st, xp, yp, rare);
    } else {
        const int nis0 = nis, xp0 = xp, yp0 = yp;
        walk_pass<true, T, EXT, KEEP>(S, surf, aux, k0, k1, r, draws, nis, st, xp, yp, rare);
        if (!kLoose<T> && wave_rare(rare)) {             // (fp32: the filtered forms stand)
            RayT<T> r2;
            D d2 = draws;
            int st2, nis2 = nis0, xp2 = xp0, yp2 = yp0;
            bool unused = false;
            restore(r2, d2, st2);
            walk_pass<false, T, EXT, KEEP>(S, surf, aux, k0, k1, r2, d2, nis2, st2, xp2, yp2, unused);
            r.pos = vselect(rare, r2.pos, r.pos);
            r.dir = vselect(rare, r2.dir, r.dir);
            draws.take(rare, d2);
            nis = rare ? nis2 : nis; st = rare ? st2 : st;
            xp = rare ? xp2 : xp; yp = rare ? yp2 : yp;
        }
    }
}

// FILT: filtered predicates (ort_device.h); false = every predicate evaluated literally.
template <int MODE, bool FILT, class T, bool ANYSRC>
__global__ __launch_bounds__(kBlock, ORT_MIN_WAVES) void trace_kernel(TraceArgs a)
{
    __shared__ typename SysTypes<T>::Sys S;
    __shared__ unsigned int blk[4];       // lost, isect, binned, help3
    // the re-run launch normally finds its list empty (no workgroup appends while it runs, so
    // every workgroup reads the same count)
    if (MODE != MODE_DEBUG && a.listed && a.redo_ctl[0] == 0) return;
    __shared__ SurfAuxT<T> AUX[ORT_MAX_SURFACES];
    stage_system(S, a.sys);
    stage_aux(AUX, S.surfaces[a.phase - 1], S.n_surfaces[a.phase - 1]);
    if (MODE != MODE_DEBUG) {
        if (threadIdx.x < 4) blk[threadIdx.x] = 0;
        __syncthreads();
    }
    unsigned int lost = 0, isect = 0, binned = 0, help3 = 0;
    int32_t *layer = hist_layer(a);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    // listed: the rays to trace are the entries of the re-run list (the queued filtered kernel
    // appended the rays that sat on a decision boundary); otherwise all of [0, n_rays)
    const bool listed = MODE != MODE_DEBUG && a.listed;
    const uint64_t n = listed ? (uint64_t)a.redo_ctl[0] : a.n_rays;
    const uint64_t ns_in = a.in_stride;                  // component stride of the input bundle

    const int ns = S.n_surfaces[a.phase - 1];
    const typename SysTypes<T>::Surf *surf = S.surfaces[a.phase - 1];
    // whole waves iterate together (the tail wave keeps its out-of-range lanes dead)
    const uint64_t base0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63ull;
    for (uint64_t wbase = base0; wbase < n; wbase += stride) {
        const uint64_t j = wbase + (threadIdx.x & 63);
        const bool act = j < n;
        const uint64_t jc = act ? j : n - 1;             // clamped index for loads of idle lanes
        const uint64_t ic = listed ? (uint64_t)a.redo_list[jc] : jc;
        const uint64_t i = ic;                           // output index (debug entry: never listed)
        RayT<T> r = {{T(0.), T(0.), T(0.)}, {T(0.), T(0.), T(1.)}}, em;
        int nis = 0, xp = -9999, yp = -9999, st = act ? -1 : ORT_ST_NA_REJECT;
        const bool have_in = MODE != MODE_FUSED && a.pos_dir_in;
        if (have_in) {
            r.pos = {(T)a.pos_dir_in[0 * ns_in + ic], (T)a.pos_dir_in[1 * ns_in + ic], (T)a.pos_dir_in[2 * ns_in + ic]};
            r.dir = {(T)a.pos_dir_in[3 * ns_in + ic], (T)a.pos_dir_in[4 * ns_in + ic], (T)a.pos_dir_in[5 * ns_in + ic]};
        }
        int kdraws = 0;
        if (MODE == MODE_DEBUG) {
            Draws d;
            if (a.u) d.init_table(a.u + ic, (int64_t)n, a.nu, a.draw_base);
            else d.init_keyed(a.rng_base, a.first_ray + ic, a.draw_base, a.wide != 0);
            if (!have_in) {
                const Draws d_none = d;
                const int est = emit<T, ANYSRC>(S, a.phase, r, d, a.first_ray + ic, a.img_cdf, a.strict != 0);
                st = est < 0 ? st : est;
                const bool exhausted = est == ORT_ST_LOST_TELESCOPE;
                d.take(exhausted, d_none);               // an exhausted image source emits nothing and draws nothing
                if (exhausted) r = {{T(0.), T(0.), T(0.)}, {T(0.), T(0.), T(0.)}};
            }
            em = r;
            if (a.path) {
                // tracker: the walk of `walk`, recording the pushes of src/stackMod.f90
                int np = 0;
                double *pp = a.path + (size_t)ic * ORT_MAX_PATH * 3;
                auto push = [&](bool c) {
                    if (c && act && np < ORT_MAX_PATH) {
                        pp[np * 3 + 0] = (double)r.pos.x; pp[np * 3 + 1] = (double)r.pos.y; pp[np * 3 + 2] = (double)r.pos.z;
                        np++;
                    }
                };
                push(true);                                          // main.f90:103,144
                for (int k = 0; k < ns; ++k) {
                    if (!wave_any_live(st)) break;
                    const bool was_live = st < 0;
                    bool unused = false;                 // the tracker walks with the literal predicates
                    surface_step<false, T, true>(S, surf[k], AUX[k], r, d, nis, st, xp, yp, unused);
                    const bool track = (__builtin_amdgcn_readfirstlane((int)surf[k].flags) & ORT_F_TRACK) != 0;
                    push(was_live && (st >= 0 || track));   // where it ended, or a tracked surface passed alive
                }
                if (act) a.npath[ic] = np;
            } else {
                const Draws d0 = d;
                const int st0 = st;
                walk<FILT, T, ANYSRC, true>(S, surf, AUX, 0, ns, r, d, nis, st, xp, yp,
                                            [&](RayT<T> &rr, Draws &dd, int &ss) { rr = em; dd = d0; ss = st0; });
            }
            kdraws = d.k;
        } else {
            KeyedDrawsT<true> d;
            d.init_keyed(a.rng_base, a.first_ray + ic, have_in ? a.draw_base : 0);
            d.set_wide(a.wide != 0);
            if (!have_in) {
                const int est = emit<T, ANYSRC>(S, a.phase, r, d, a.first_ray + ic, a.img_cdf, a.strict != 0);
                st = est < 0 ? st : est;
            }
            const RayT<T> r0 = r;
            const KeyedDrawsT<true> d0 = d;
            const int st0 = st;
            walk<FILT, T, ANYSRC, false>(S, surf, AUX, 0, ns, r, d, nis, st, xp, yp,
                                         [&](RayT<T> &rr, KeyedDrawsT<true> &dd, int &ss) { rr = r0; dd = d0; ss = st0; });
        }
        if (!act) continue;
        if (MODE == MODE_DEBUG) {
            if (a.pos_dir_out) {
                a.pos_dir_out[0 * n + i] = (double)r.pos.x; a.pos_dir_out[1 * n + i] = (double)r.pos.y;
                a.pos_dir_out[2 * n + i] = (double)r.pos.z; a.pos_dir_out[3 * n + i] = (double)r.dir.x;
                a.pos_dir_out[4 * n + i] = (double)r.dir.y; a.pos_dir_out[5 * n + i] = (double)r.dir.z;
            }
            if (a.emitted_out) {
                a.emitted_out[0 * n + i] = (double)em.pos.x; a.emitted_out[1 * n + i] = (double)em.pos.y;
                a.emitted_out[2 * n + i] = (double)em.pos.z; a.emitted_out[3 * n + i] = (double)em.dir.x;
                a.emitted_out[4 * n + i] = (double)em.dir.y; a.emitted_out[5 * n + i] = (double)em.dir.z;
            }
            if (a.status) a.status[i] = st;
            if (a.bin_xy) { a.bin_xy[i] = xp; a.bin_xy[n + i] = yp; }
            if (a.n_isect) a.n_isect[i] = nis;
            if (a.n_draws) a.n_draws[i] = kdraws;
        } else {
            isect += (unsigned)nis;
            if (st == ORT_ST_BINNED) {
                binned++;
                bin_hit(layer, xp, yp, a.replicas != nullptr);
            } else if (st >= ORT_ST_LOST_BOTTLE) {
                lost++;                                                       // optics_system.f90:32,42; main.f90:151
                if (st == ORT_ST_HELP3) help3++;
            }
        }
    }
    if (MODE != MODE_DEBUG) {
        atomicAdd(&blk[0], lost); atomicAdd(&blk[1], isect);
        atomicAdd(&blk[2], binned); atomicAdd(&blk[3], help3);
        __syncthreads();
        if (threadIdx.x < 4 && blk[threadIdx.x])
            atomicAdd(&a.counters[2 * threadIdx.x + (a.phase - 1)], (unsigned long long)blk[threadIdx.x]);
        // the last workgroup to finish leaves the list empty for the next launch (every workgroup
        // has read ctl[0] before it counts itself done)
        if (listed && threadIdx.x == 0 && atomicAdd(&a.redo_ctl[1], 1u) == gridDim.x - 1) {
            atomicAdd(&a.work[ORT_W_DEFERRED], (unsigned long long)a.redo_ctl[0]);
            a.redo_ctl[0] = 0;
            a.redo_ctl[1] = 0;
        }
    }
}

// ---------------------------------------------------------------------------
// Queued variant ("wavefront per ray bundle"): every wave is an independent worker
// over a contiguous range of global ray indices.  The surface list is cut into two
// segments at S.split (host-chosen: just after the aperture stop that removes most
// rays).  Segment 1 runs in lockstep on 64 fresh rays; the survivors are appended to
// a wave-private ray queue in LDS (SoA: pos, dir in the kernel's precision, draw state; 128 slots).  As soon
// as 64 rays are queued the wave runs segment 2 on a FULL wavefront.  Dead lanes of
// segment 1 therefore never ride along through segment 2 — the lanes stay busy
// although rays die at different surfaces.  No workgroup barrier is involved: a
// queue is only ever touched by the wave that owns it.  Per-ray arithmetic and draw
// order are exactly those of the lockstep kernel, so results are bit-identical.
// With filtered predicates the kernel holds no literal formula at all: see `defer` below.
// ---------------------------------------------------------------------------
constexpr uint64_t kNoRay = ~0ull;   // hand-over bundle of the scattering pipeline: a slot without a ray
// control words of the scattering pipeline (TraceArgs.scat_ctl), each on a 128-byte line of its own
constexpr int kScatCtlStride = 16, kScatHeads = 8, kScatSlotsWord = kScatHeads * kScatCtlStride, kScatCtlWords = (kScatHeads + 1) * kScatCtlStride;
constexpr int kScatWaves = 12;              // wavefronts per workgroup of scatter_front_kernel = per CU (LDS and 168 VGPRs allow no more)
constexpr unsigned kHandChunk = 256;        // hand-over slots a wave allocates at a time
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kQueueCap = 128;      // >= 63 leftover + 64 new survivors
constexpr int kQueueFields = 6;     // px py pz dx dy dz (+ the draw state: its own array)

__device__ inline int lane_prefix(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

template <int MODE, bool FILT, bool ANYSRC, class T, int PROG = PROG_GENERIC, bool SCAT = ANYSRC>
__global__ __launch_bounds__(kBlock, PROG != PROG_GENERIC ? 4 : ORT_MIN_WAVES) void trace_queue_kernel(TraceArgs a)
{
    static_assert(PROG == PROG_GENERIC || (!ANYSRC && !SCAT && FILT), "programs exist for the lean kernels only (filtered forms)");
    // fp32 (kLoose): the filtered forms decide every lane — nothing is deferred, `rare` is not looked at
    constexpr bool DEFER = FILT && !kLoose<T>;
    __shared__ typename SysTypes<T>::Sys S;
    // 26.6 KB per workgroup of a program kernel in fp64 (6 workgroups per CU's 160 KB), 14 KB in fp32
    constexpr bool fixed = PROG != PROG_GENERIC;        // surface program known at compile time
    using QT = typename std::conditional<std::is_same<T, float>::value, float, double>::type;
    constexpr bool sdraws = prog_static_draws<PROG>();  // every lane at the same, compile-time draw index (ProgDraws)
    using QD = typename std::conditional<sdraws, uint32_t, uint64_t>::type;  // static draws: the ray's index in the launch
    __shared__ QT Q[kWavesPerBlock][kQueueFields][kQueueCap];
    __shared__ QD QDRAW[kWavesPerBlock][kQueueCap];
    // ring programs, fused: segment 0 (below) culls the rays that are certain to miss the first aperture
    // before anything is emitted; the others wait here (ray index in the launch) for a full wave
    constexpr bool PRE = prog_culls<PROG>() && MODE == MODE_FUSED;
    __shared__ uint32_t CQ[kWavesPerBlock][PRE ? kQueueCap : 1];
    // intersections evaluated before the queue point: `split` for every survivor unless a surface
    // scatters (extended instantiation), so only that one carries the count through the queue
    constexpr bool CARRY = SCAT || MODE == MODE_CONTINUE;     // the count differs from ray to ray at the queue point
    __shared__ int QN[kWavesPerBlock][CARRY ? kQueueCap : 1];
    __shared__ unsigned int blk[5];       // lost, isect, binned, help3, culled
    __shared__ SurfAuxT<T> AUX[PROG == PROG_GENERIC ? ORT_MAX_SURFACES : 1];
    // a program kernel reads everything it needs of the system (surface records, emitter and image
    // constants) through scalar loads from the device copy: nothing to stage, no barrier at its start
    if (PROG == PROG_GENERIC) {
        stage_system(S, a.sys);
        stage_aux(AUX, S.surfaces[a.phase - 1], S.n_surfaces[a.phase - 1]);
    }
    if (threadIdx.x < 5) blk[threadIdx.x] = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // uniform: the wave's range and loop control stay scalar
    QT (*q)[kQueueCap] = Q[wave];
    QD *qd = QDRAW[wave];
    uint32_t *cq = CQ[wave];
    int *qn = QN[wave];
    using DrawsT = typename std::conditional<sdraws, ProgDraws, KeyedDraws>::type;
    int phase = a.phase, ns, split;
    if constexpr (fixed) {
        phase = Prog<PROG>::phase; ns = Prog<PROG>::n; split = queue_step<PROG, MODE>();   // host: match_program
        static_assert(queue_step<PROG, MODE>() < Prog<PROG>::n, "segment 1 of a program must end in front of its image plane (the fp32 hit log relies on it)");
    } else {
        ns = S.n_surfaces[phase - 1];
        split = S.split[phase - 1];
        if (split <= 0 || split >= ns) split = ns;      // no queue point: one segment
    }
    const int ph = phase - 1;
    const typename SysTypes<T>::Surf *surf = S.surfaces[ph];
    // program kernels: scalar loads from the device copy in the kernel's own precision
    typename ConstPtrs<T>::sys_t csys;
    typename ConstPtrs<T>::surf_t csurf;
    typename ConstPtrs<T>::aux_t caux;
    if constexpr (std::is_same<T, float>::value) {
        csys = (typename ConstPtrs<T>::sys_t)a.sysf;
        csurf = (typename ConstPtrs<T>::surf_t)a.sysf->surfaces[ph];
        caux = (typename ConstPtrs<T>::aux_t)a.auxf;
    } else {
        csys = (typename ConstPtrs<T>::sys_t)a.sys;
        csurf = (typename ConstPtrs<T>::surf_t)a.sys->surfaces[ph];
        caux = (typename ConstPtrs<T>::aux_t)a.aux;
    }
    int32_t *layer = hist_layer(a);
    // MODE_CONTINUE: the slots of the hand-over bundle, one per ray of the launch, some of them empty
    // MODE_CONTINUE: the hand-over slots the front kernel allocated (a count it left on the device), some of them empty
    uint64_t n = a.n_rays;
    if (MODE == MODE_CONTINUE) {
        const uint64_t used = (uint64_t)a.scat_ctl[kScatSlotsWord];
        n = used < a.cont_cap ? used : a.cont_cap;
    }
    const uint64_t ns_in = MODE == MODE_CONTINUE ? a.cont_cap : a.in_stride;
    const int k0 = MODE == MODE_CONTINUE ? a.cont_k0 : 0;     // first surface walked here
    if (MODE == MODE_CONTINUE && split <= k0) split = ns;     // no queue point behind the start: one segment

    // contiguous, 64-aligned range of ray indices for this wave: long ranges for the workgroups of
    // the first rounds, short ones for the last workgroups (plan_ranges), so that the chip drains evenly
    uint64_t lo, hi;
    if (MODE == MODE_CONTINUE) {                            // the slot count is only known here: equal ranges over the grid
        const uint64_t nw = (uint64_t)gridDim.x * kWavesPerBlock, wid = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
        const uint64_t chunk = (((n + nw - 1) / nw) + 63) & ~63ull;
        lo = wid * chunk; if (lo > n) lo = n;
        hi = lo + chunk;  if (hi > n) hi = n;
    } else {
        const bool head = blockIdx.x < a.head_blocks;
        const uint64_t wid = (uint64_t)(head ? blockIdx.x : blockIdx.x - a.head_blocks) * kWavesPerBlock + wave;
        const uint64_t chunk = head ? a.head_chunk : a.tail_chunk;
        const uint64_t end = head ? a.head_rays : n;
        lo = (head ? 0 : a.head_rays) + wid * chunk; if (lo > end) lo = end;
        hi = lo + chunk;  if (hi > end) hi = end;
    }

    // what the steps of a program kernel may assume (ort_device.h): fused rays have unit directions (they
    // were emitted here), the lens spheres of every program are centred on the axis (host: matches<P>)
    // (axial start: exact fp64 only — fast fp64 contracts L.L into fmas that the constants of the host do not replay; fp32 keeps its literal steps)
    constexpr bool axial = MODE == MODE_FUSED && prog_starts_on_axis<PROG>() && FILT && std::is_same<T, double>::value;
    constexpr int OPT = fixed ? ((MODE == MODE_FUSED ? OPT_UNIT_DIR : 0) | OPT_ON_AXIS | (axial ? OPT_AXIAL_START : 0) |
                                 (MODE == MODE_CONTINUE ? OPT_COUNT_STEPS : 0)) : 0;
    constexpr bool tagged = fixed && MODE != MODE_CONTINUE;     // st = ORT_ST_* | intersections << 8
    // fp32: the hits go to the launch's log instead of the image (bin_log_kernel: the memory-side atomics bound this kernel)
    constexpr bool LOG = kLoose<T> && MODE != MODE_CONTINUE;
    const bool logging = LOG && a.hit_log != nullptr;           // the host's choice per launch (launch_trace)
    unsigned int lost = 0, isect = 0, binned = 0, help3 = 0, culled = 0;
    unsigned int hits[kBinTiles] = {0, 0, 0, 0, 0};            // LOG: entries this wave has written to each part (wave-uniform)
    auto finish = [&](int st_in, int nis_in, int xp, int yp) {
        const int st = tagged ? status_code(st_in) : st_in;
        const int nis = tagged ? status_isect(st_in) : nis_in;
        isect += (unsigned)nis;
        if (st == ORT_ST_BINNED) {
            binned++;
            if (!logging) bin_hit(layer, xp, yp, a.replicas != nullptr);
        } else if (st >= ORT_ST_LOST_BOTTLE) {
            lost++;                                                          // optics_system.f90:32,42; main.f90:151
            if (st == ORT_ST_HELP3) help3++;
        }
    };
    // A ray that raised `rare` (ort_device.h: it sat on a decision boundary of a filtered
    // predicate) leaves this kernel without any side effect: its index goes to the re-run list
    // and trace_kernel<literal> traces it from the start afterwards.  ~4e-6 of the rays.
    auto defer = [&](uint64_t i) { a.redo_list[atomicAdd(&a.redo_ctl[0], 1u)] = (uint32_t)(a.defer_base + i); };
    // LOG: called by the whole wave (uniform control flow) with `hit` = this lane's ray ended binned in this pass
    auto log_hits = [&](bool hit, int st_in, int xp, int yp) {
        hit = hit && (tagged ? status_code(st_in) : st_in) == ORT_ST_BINNED;
        const uint32_t bin = (uint32_t)((xp + 200) + ORT_IMAGE_N * (yp + 200));   // imageMod.f90:55-56
        const uint32_t part = bin % (uint32_t)kBinTiles, idx = bin / (uint32_t)kBinTiles;
#pragma unroll
        for (int t = 0; t < kBinTiles; ++t) {
            const bool mine = hit && part == (uint32_t)t;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
            if (mine) a.hit_log[(size_t)t * a.hit_stride + a.hit_base + lo + hits[t] + (unsigned)lane_prefix(m)] = (uint16_t)idx;
            hits[t] += (unsigned)__popcll(m);
        }
    };

    uint64_t next = lo;
    int img_hint = -1;           // image source: the cell of the previous batch's first ray (emit_image)
    int qcount = 0, qhead = 0;
    int ccount = 0, chead = 0;
    const uint64_t z0 = zray_of(a.rng_base, a.first_ray);      // ProgDraws::init_index (a launch holds < 2^32 rays)
    // segment 0: zray of ray next + lane = znext (wave-uniform) + zlane
    uint64_t znext = z0 + (kGolden << 23) * next;
    const uint64_t zlane = (kGolden << 23) * (uint64_t)lane;
    unsigned int culled_wave = 0;     // rays segment 0 culled (wave-uniform): each is lost after one intersection
    for (;;) {
        const bool have_new = next < hi;
        const bool cand_ready = PRE && (ccount >= 64 || (!have_new && ccount > 0));
        if (qcount >= 64 || (!have_new && !cand_ready && qcount > 0)) {
            // ---- segment 2 on up to 64 queued rays
            const int m = qcount < 64 ? qcount : 64;
            const bool act = lane < m;
            const int slot = (qhead + lane) & (kQueueCap - 1);
            qhead = (qhead + m) & (kQueueCap - 1);
            qcount -= m;
            // (every lane loads: the slots of the lanes beyond m hold rays of earlier passes — or, in a wave's first
            // partial pass, whatever the LDS held — whose arithmetic is discarded: st >= 0 keeps them out of every
            // decision, side effect and deferral)
            RayT<T> r = {{T(0.), T(0.), T(0.)}, {T(0.), T(0.), T(1.)}};
            DrawsT d;
            QD dw = 0;                                   // the queued image of the draw state
            int nis = 0, xp = 0, yp = 0, st = act ? -1 : ORT_ST_NA_REJECT;
            {
                r.pos = {T(q[0][slot]), T(q[1][slot]), T(q[2][slot])};
                r.dir = {T(q[3][slot]), T(q[4][slot]), T(q[5][slot])};
                dw = qd[slot];
                nis = CARRY ? qn[slot] : split;
            }
            if constexpr (sdraws) d.init_index(z0, dw, 0);
            else d.unpack(dw, a.rng_base);
            bool rare = false;
            if constexpr (fixed) {
                // the queue point lies INSIDE step split - 1, behind its aperture test (step_part)
                step_part<FILT, T, PROG, queue_step<PROG, MODE>() - 1, 2, OPT>(*csys, csurf, caux, r, d, nis, st, xp, yp, rare);
                walk_fixed<FILT, T, false, PROG, queue_step<PROG, MODE>(), Prog<PROG>::n, false, OPT>(*csys, csurf, caux, r, d, nis, st, xp, yp, rare);
            } else walk_pass<FILT, T, SCAT, false>(S, surf, AUX, split, ns, r, d, nis, st, xp, yp, rare);
            if (act) {
                if (DEFER && rare) defer(sdraws ? (uint64_t)dw : d.ray_of_packed(dw, a.rng_base) - a.first_ray);
                else finish(st, nis, xp, yp);
            }
            if constexpr (LOG) { if (logging) log_hits(act, st, xp, yp); }      // (fp32 defers nothing)
            __builtin_amdgcn_wave_barrier();
        } else if (cand_ready || (!PRE && have_new)) {
            // ---- segment 1 on 64 fresh rays (ring programs: on up to 64 rays that passed segment 0)
            uint64_t i;
            bool act;
            if constexpr (PRE) {
                const int m = ccount < 64 ? ccount : 64;
                act = lane < m;
                i = act ? (uint64_t)cq[(chead + lane) & (kQueueCap - 1)] : lo;
                chead = (chead + m) & (kQueueCap - 1);
                ccount -= m;
            } else {
                i = next + (uint64_t)lane;
                act = i < hi;
                next += 64;
            }
            const uint64_t ic = act ? i : (PRE ? lo : hi - 1);   // idle lanes recompute a ray of this wave, unused
            RayT<T> r;
            DrawsT d;
            int nis = 0, xp = 0, yp = 0, st = act ? -1 : ORT_ST_NA_REJECT;
            bool rare = false;
            uint64_t ridx = i;                           // the ray's index in the launch (what a deferral lists)
            if constexpr (MODE == MODE_CONTINUE) {
                const uint64_t c0 = a.cont_draw[ic];
                act = act && c0 != kNoRay;
                st = act ? -1 : ORT_ST_NA_REJECT;
                d.unpack(c0, a.rng_base);
                ridx = (c0 >> 24) - a.first_ray;
                nis = a.cont_nis[ic];
                r.pos = {T(a.cont_pos_dir[0 * ns_in + ic]), T(a.cont_pos_dir[1 * ns_in + ic]), T(a.cont_pos_dir[2 * ns_in + ic])};
                r.dir = {T(a.cont_pos_dir[3 * ns_in + ic]), T(a.cont_pos_dir[4 * ns_in + ic]), T(a.cont_pos_dir[5 * ns_in + ic])};
                // the rest of the step of surface k0 - 1, where the ray arrived after its walk (lens.f90:283-297, :334-348):
                // back test, move, normal, Fresnel — surface_step's tail for a wall of the bottle (no aperture stop)
                typename SysTypes<T>::Surf sw;
                SurfAuxT<T> axw;
                if constexpr (fixed) { sw = load_surface<T>(csurf + (k0 - 1)); axw = load_aux<T>(caux + (k0 - 1)); }
                else { sw = surf[k0 - 1]; axw = AUX[k0 - 1]; }
                const unsigned wflags = (unsigned)__builtin_amdgcn_readfirstlane((int)sw.flags);
                const int wlost = (wflags & ORT_F_BOTTLE) ? ORT_ST_LOST_BOTTLE : ORT_ST_LOST_TELESCOPE;
                const bool back = act && (wflags & ORT_F_SCATTER) != 0 && r.dir.z < T(0.);
                const bool on = act && !back;
                r.pos = vadd(r.pos, vscale(r.dir, T(a.cont_t[ic])));
                const VecT<T> Nraw = {T(0.0), sw.cy - r.pos.y, sw.cz - r.pos.z};
                // (a ray that left its walk by the reference's `out` test is not on the wall: no estimate of |N| holds)
                const VecT<T> Nw = vnormalise_f<FILT, T>(Nraw, on, rare, true);
                const T uw = d.template peek_as<T>();
                d.advance(on);
                const bool refl = reflect_refract<FILT, false, T>(r.dir, Nw, sw.n1, sw.n2, sw.eta, axw.eta2, uw, on, rare);
                const bool diesw = on && refl && (wflags & ORT_F_SKIP_ON_REFLECT) != 0;
                st = (back || diesw) ? wlost : st;
            } else if (MODE == MODE_RESIDENT) {
                if constexpr (sdraws) d.init_index(z0, (uint32_t)ic, a.draw_base);
                else d.init_keyed(a.rng_base, a.first_ray + ic, a.draw_base);
                r.pos = {T(a.pos_dir_in[0 * ns_in + ic]), T(a.pos_dir_in[1 * ns_in + ic]), T(a.pos_dir_in[2 * ns_in + ic])};
                r.dir = {T(a.pos_dir_in[3 * ns_in + ic]), T(a.pos_dir_in[4 * ns_in + ic]), T(a.pos_dir_in[5 * ns_in + ic])};
            } else {
                if constexpr (sdraws) d.init_index(z0, (uint32_t)ic, 0);
                else d.init_keyed(a.rng_base, a.first_ray + ic, 0);
                int est;
                if constexpr (fixed) est = emit<T, false, FILT, Prog<PROG>::emitter>(*csys, phase, r, d, a.first_ray + ic, a.img_cdf, rare, &img_hint);
                else est = emit<T, ANYSRC, FILT && !ANYSRC>(S, phase, r, d, a.first_ray + ic, a.img_cdf, rare, nullptr, a.strict != 0);
                st = est < 0 ? st : est;
            }
            if constexpr (fixed && MODE == MODE_CONTINUE) {
                // the list behind the bottle wall the rays were handed over at: its second wall first if only the contents scatter
                if (k0 == 1) walk_fixed<FILT, T, false, PROG, 1, 2, false, OPT>(*csys, csurf, caux, r, d, nis, st, xp, yp, rare);
                walk_fixed<FILT, T, false, PROG, 2, queue_step<PROG, MODE>() - 1, false, OPT>(*csys, csurf, caux, r, d, nis, st, xp, yp, rare);
                step_part<FILT, T, PROG, queue_step<PROG, MODE>() - 1, 1, OPT>(*csys, csurf, caux, r, d, nis, st, xp, yp, rare);
            } else if constexpr (fixed) {
                walk_fixed<FILT, T, false, PROG, 0, queue_step<PROG, MODE>() - 1, false, OPT>(*csys, csurf, caux, r, d, nis, st, xp, yp, rare);
                step_part<FILT, T, PROG, queue_step<PROG, MODE>() - 1, 1, OPT>(*csys, csurf, caux, r, d, nis, st, xp, yp, rare);
            } else walk_pass<FILT, T, SCAT, false>(S, surf, AUX, k0, split, r, d, nis, st, xp, yp, rare);
            const bool deferred = DEFER && rare && act;
            const bool survive = act && st < 0 && !deferred;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(survive);
            if (survive) {
                const int slot = (qhead + qcount + lane_prefix(mask)) & (kQueueCap - 1);
                q[0][slot] = (QT)r.pos.x; q[1][slot] = (QT)r.pos.y; q[2][slot] = (QT)r.pos.z;
                q[3][slot] = (QT)r.dir.x; q[4][slot] = (QT)r.dir.y; q[5][slot] = (QT)r.dir.z;
                if constexpr (sdraws) qd[slot] = (uint32_t)i;
                else qd[slot] = d.pack();
                if (CARRY) qn[slot] = nis;
            } else if (deferred) {
                defer(ridx);
            } else if (act) {
                finish(st, nis, xp, yp);
            }
            // (a surface program's segment 1 ends in front of its image plane: no ray is binned here)
            if constexpr (LOG && !fixed) { if (logging) log_hits(act && !survive, st, xp, yp); }
            qcount += __popcll(mask);
            __builtin_amdgcn_wave_barrier();
        } else if (PRE && have_new) {
            // ---- segment 0 (ring programs) on 64 fresh ray indices.  `ring` aims every ray at a point of
            // the plane z = L2%fb with squared radius rr = ranu(0., (radius + 10e-3)**2), its third draw
            // (src/sourceMod.f90:283-286), and that plane IS the plano-convex lens's flat face
            // (centre%z + curve_radius - thickness = fb, src/lens.f90:163, :448): the ray crosses the
            // face at its aim point up to rounding (host: ring_cull_threshold bounds it by 1e-8 of rr),
            // so rr > radius^2 (1 + 1e-6) means r > this%radius at lens.f90:451 whatever the other
            // three draws are: the ray ends there after ONE surface solve.  69 % of the ring rays; they
            // are counted (lost, one intersection) and never emitted.  Results are unchanged
            // (tests: with culling == without it, bit for bit, on every system of the parity suite).
            // The pass is ~30 vector instructions for 64 rays: the ray's hash input is a wave-uniform 64-bit sum (scalar
            // unit) plus a per-lane constant; the test rr > cull, rr = 0 + u3 (ring_lens_r2 - 0) as emit_ring forms it, is
            // monotone in the draw's 32-bit word and is taken on the word (cull_word: the largest word that survives, found
            // by the host with the kernel's own arithmetic); the culled rays are counted per wave, not per lane.
            if constexpr (PRE) {
                const uint32_t left = (uint32_t)(hi - next);                  // >= 1, and a launch holds < 2^32 rays
                const bool act = (uint32_t)lane < left;
                const uint64_t h = mix64((znext + zlane) + kGolden * 2ull);   // pair 1 of ray next + lane: its draws 2 and 3
                const uint32_t w = (uint32_t)(h >> 32);                       // draw 2, the third (ProgDraws::at<T, 2>)
                bool dies;
                if constexpr (std::is_same<T, float>::value) dies = w > a.cull_wordf;
                else dies = w > a.cull_word;
                const bool cand = act && !dies;
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(cand);
                if (cand) cq[(chead + ccount + lane_prefix(mask)) & (kQueueCap - 1)] = (uint32_t)next + (uint32_t)lane;
                const int passed = __popcll(mask);
                culled_wave += (unsigned)((left < 64u ? (int)left : 64) - passed);
                ccount += passed;
                znext += (kGolden << 23) * 64ull;
            }
            next += 64;
            __builtin_amdgcn_wave_barrier();
        } else {
            break;
        }
    }
    if (logging && lane == 0) {
        uint32_t *e = a.hit_dir + ((size_t)blockIdx.x * kWavesPerBlock + wave) * kBinDirWords;
#pragma unroll
        for (int t = 0; t < kBinTiles; ++t) e[t] = hits[t];
        e[kBinTiles] = a.hit_base + (uint32_t)lo;
    }
    if (PRE && lane == 0) { lost += culled_wave; isect += culled_wave; culled = culled_wave; }   // src/optics_system.f90:42 (lost), one intersection each
    atomicAdd(&blk[0], lost); atomicAdd(&blk[1], isect);
    atomicAdd(&blk[2], binned); atomicAdd(&blk[3], help3);
    if constexpr (PRE) atomicAdd(&blk[4], culled);
    __syncthreads();
    if (threadIdx.x < 4 && blk[threadIdx.x])
        atomicAdd(&a.counters[2 * threadIdx.x + (a.phase - 1)], (unsigned long long)blk[threadIdx.x]);
    if (PRE && threadIdx.x == 4 && blk[4]) atomicAdd(&a.work[ORT_W_CULLED], (unsigned long long)blk[4]);
}


// ---------------------------------------------------------------------------
// In-bottle scattering (src/lens.f90:262-282, :312-333, src/surfaces.f90:13-50,
// src/stokes.f90:7-166) as its own kernel in front of the lean walk.
//
// The random walk is a loop of unknown length per ray around ~800 instructions of log / atan2 / acos /
// sin / cos; compiled INTO the surface walk it costs every lane of every step 200+ VGPRs (2 waves per SIMD)
// and runs in lockstep until the last of 64 rays has left its walk (lanes busy: ~20 %).  Here the surfaces up to
// the last scattering one (the bottle's two walls) are cut into three stages, each run on FULL wavefronts fed
// from wave-private LDS queues — the scheme of trace_queue_kernel with a cycle in it:
//   E  64 fresh rays: emit, ENTER surface 0
//   W  64 walking rays: one scattering event (move, absorb?, stokes, next leg by tauint); a ray that goes on
//      walking returns to the walk queue, one that reaches the wall (or leaves the cylinder) goes to the arrival queue
//   A  64 arrived rays: the rest of the surface step (back test, move, normal, Fresnel), then ENTER the next
//      surface, or — behind the last scattering surface — hand the ray over
//   ENTER surface k: intersect; if the medium in front of it scatters, the first leg (tauint)
// Rays that survive are appended to the hand-over bundle in HBM (state, keyed draw counter, intersections so far)
// and trace_queue_kernel<MODE_CONTINUE> walks the remaining surfaces at its own register budget.  Every
// operation of a ray is the one the monolithic kernel (and the lockstep kernel) performs, in the same order with
// the same draws.  The quadratics of the walls and of every leg, and the normal + Fresnel step at an inner wall, are
// evaluated in their filtered forms (ort_device.h: the same bits, or the ray is listed for the literal re-run); the walk
// itself (stokes, log) is literal — so rays, images and counters are bit-identical to theirs (tests: the pipeline against
// the lockstep kernel; both against the CPU checker).
// The transcendental functions of the walk are glibc's own algorithms (ort_libm.h: the reference's results bit for
// bit); their lookup tables (sin/cos, atan2, acos: 38 KB) are staged ONCE per workgroup into LDS — gathered from
// constant memory they cost the vector cache ~50 cycles per wavefront-wide load and bound the walk (tools/ubench_libm.hip).
// One workgroup of kScatWaves = 12 wavefronts per CU (151 KB of LDS: tables + 12 pools of 9 KB + the staged system; the
// kernel's 168 VGPRs allow 3 waves per SIMD), every wave an independent worker: no barrier after the staging.
// Work distribution: PERSISTENT waves pull batches of rays from eight heads, one per XCD (HW_REG_XCC_ID; head x hands
// out an eighth of the launch's ray range, scat_grab rays per returning atomic), and steal from the next head when
// theirs is dry; a wave ends when all eight are.  With static ranges the 12 waves of a CU ended at 0.51 / 0.81 / 1.13 M
// cycles (a SIMD serves its oldest wave first) and the last third of every launch ran at one wave per SIMD
// (profiles/r03/scatbench.log).  Keyed draws make the result independent of which wave traces which ray.
// Hand-over slots are allocated kHandChunk at a time from one counter (scat_ctl[kScatSlotsWord]); a wave fills its
// chunk from the bottom and marks what is left empty when it ends, so every allocated slot is written and the
// continuation walks exactly the allocated count.
// ---------------------------------------------------------------------------
constexpr int kSQCap = 128;        // at most 128 rays in flight per wave: stage E runs only while <= 64 are (a power of two)
// ONE pool of ray slots per wave and three rings of slot numbers over it — walking, arrived, free: a ray keeps its
// slot from stage to stage, only its number moves between the rings
struct ScatPool {
    double f[7][kSQCap];           // px py pz dx dy dz t (length of the next leg)
    uint64_t c[kSQCap];            // keyed draw counter
    uint32_t m[kSQCap];            // intersections so far << 8 | surface index
    uint8_t ring[3][kSQCap];       // slot numbers: RING_WALK, RING_ARRIVED, RING_FREE
};
constexpr int RING_WALK = 0, RING_ARRIVED = 1, RING_FREE = 2;
static_assert(sizeof(ScatPool) * kScatWaves + glibc::kLdsTableWords * 8 + sizeof(ort_system) + 256 <= 160 * 1024, "scatter_front_kernel: LDS of one CU");

// ENTER surface k (per lane) for the lanes `on`: src/lens.f90:255-261 / :303-311 up to the first tauint.
// Circular walls: the two quadratics in their filtered forms (ort_device.h: the same bits, or the lane raises `rare`);
// a lane that did is left exactly as it came (`ended` untouched, neither walking nor arrived): the caller defers it.
__device__ inline void scat_enter(const ort_surface *surf, int k, int kind0, bool on, const Ray &r, KeyedDraws &d,
                                  int &nis, double &t, bool &walking, bool &arrived, int &ended, bool &rare)
{
    const ort_surface &s = surf[k];
    const int ended0 = ended;
    nis += on ? 1 : 0;
    double tt;
    bool hit, unused = false;
    if (kind0 == ORT_SURF_ELLIPSE) intersect_ellipse<false, double>(r, s.cy, s.cz, s.radius, s.radius_b, 0., 0., on, tt, hit, unused);
    else intersect_quadric<true, double>(r, s.cx, s.cy, s.cz, s.radius, s.radius * s.radius, true, on, tt, hit, rare);
    const unsigned flags = s.flags;
    const int lost = (flags & ORT_F_BOTTLE) ? ORT_ST_LOST_BOTTLE : ORT_ST_LOST_TELESCOPE;
    ended = (on && !hit) ? ((flags & ORT_F_MISS_IS_HELP3) ? ORT_ST_HELP3 : lost) : ended;
    const bool go = on && hit;
    const bool scat = (flags & ORT_F_SCATTER) != 0;
    double dist;
    bool at_wall, ok;
    tauint<double, KeyedDraws, true>(r, s.mua, s.mus, s.cy, s.cz, s.scat_radius, go && scat, d, dist, at_wall, ok, nis, &rare);
    ended = (go && scat && !ok) ? ORT_ST_NO_INTERSECTION : ended;
    t = go ? (scat ? dist : tt) : t;
    const bool alive = go && (!scat || ok);
    const bool bad = on && rare;
    walking = alive && scat && !at_wall && !bad;
    arrived = alive && !(alive && scat && !at_wall) && !bad;
    ended = bad ? ended0 : ended;
    rare = bad;
}

#ifdef ORT_SCAT_TIMING
__device__ unsigned long long g_scat_times[4 * 16384];     // dev builds: start, last emission, end, passes per wave
#endif
__device__ inline uint64_t uniform64(uint64_t v)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// ANYSRC = false: the point source only (the default of the loop the bottle belongs to, src/main.f90:136)
template <bool ANYSRC>
__global__ __launch_bounds__(64 * kScatWaves) void scatter_front_kernel(TraceArgs a)
{
    __shared__ ort_system S;
    __shared__ ScatPool POOLS[kScatWaves];
    __shared__ uint64_t LT[glibc::kLdsTableWords];       // glibc's sin/cos, atan2 and acos tables (ort_libm.h: TabLds)
    __shared__ double ALB[ORT_MAX_SURFACES];
    __shared__ unsigned int blk[4];
    stage_system(S, a.sys);
    glibc::stage_tables(LT, (int)threadIdx.x, (int)blockDim.x);
    if (threadIdx.x < 4) blk[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    ScatPool &P = POOLS[wave];
    for (int j = lane; j < kSQCap; j += 64) P.ring[RING_FREE][j] = (uint8_t)j;      // every slot starts free
    const int ph = a.phase - 1;
    const ort_surface *surf = S.surfaces[ph];
    // albedo of every surface's medium (lens.f90:266, :317), the division done once per workgroup instead of once per event
    if (threadIdx.x < ORT_MAX_SURFACES) ALB[threadIdx.x] = surf[threadIdx.x].mus / (surf[threadIdx.x].mus + surf[threadIdx.x].mua);
    __syncthreads();
    const glibc::TabLds tabs = {(const __attribute__((address_space(3))) uint64_t *)LT};
    const int kind0 = __builtin_amdgcn_readfirstlane(surf[0].kind);      // host: the same for every surface in front of cont_k0
    const int klast = a.cont_k0 - 1;

    unsigned int lost = 0, isect = 0, help3 = 0;
    auto end_ray = [&](int st, int nis) {                 // a ray that ends inside the bottle (nothing is binned here)
        isect += (unsigned)nis;
        lost++;                                           // every status a ray can end with here counts as lost
        if (st == ORT_ST_HELP3) help3++;
    };
    // rings: wcount + acount + fcount + (slots held by the lanes of the running stage) = kSQCap
    int wcount = 0, whead = 0, acount = 0, ahead = 0, fcount = kSQCap, fhead = 0;
    auto give = [&](int which, int &count, int head, bool cond, int slot) {        // append the slot numbers of the lanes `cond`
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(cond);
        if (cond) P.ring[which][(head + count + lane_prefix(mask)) & (kSQCap - 1)] = (uint8_t)slot;
        count += __popcll(mask);
    };
    auto take = [&](int which, int &count, int &head, bool &act, int &slot) {      // the first (up to) 64 slot numbers
        const int m = count < 64 ? count : 64;
        act = lane < m;
        slot = act ? (int)P.ring[which][(head + lane) & (kSQCap - 1)] : 0;
        head = (head + m) & (kSQCap - 1);
        count -= m;
    };
    auto store = [&](bool cond, int slot, const Ray &r, double t, const KeyedDraws &d, int nis, int k) {
        if (cond) {
            P.f[0][slot] = r.pos.x; P.f[1][slot] = r.pos.y; P.f[2][slot] = r.pos.z;
            P.f[3][slot] = r.dir.x; P.f[4][slot] = r.dir.y; P.f[5][slot] = r.dir.z;
            P.f[6][slot] = t;
            P.c[slot] = d.c;
            P.m[slot] = ((uint32_t)nis << 8) | (uint32_t)k;
        }
    };
    auto load = [&](bool act, int slot, Ray &r, double &t, KeyedDraws &d, int &nis, int &k) {
        r = {{0., 0., 0.}, {0., 0., 1.}};
        t = 0.; nis = 0; k = 0;
        d.base = a.rng_base; d.c = 0;
        if (act) {
            r.pos = {P.f[0][slot], P.f[1][slot], P.f[2][slot]};
            r.dir = {P.f[3][slot], P.f[4][slot], P.f[5][slot]};
            t = P.f[6][slot];
            d.c = P.c[slot];
            const uint32_t mm = P.m[slot];
            nis = (int)(mm >> 8); k = (int)(mm & 0xffu);
        }
    };
    // where the rays of a stage go: back to the walk ring, to the arrival ring (state stored in their own slot), or out
    // of the pool (ended / handed over: the slot is free again)
    auto route = [&](bool held, int slot, bool to_walk, bool to_arrived, const Ray &r, double t, const KeyedDraws &d, int nis, int k) {
        store(to_walk || to_arrived, slot, r, t, d, nis, k);
        give(RING_WALK, wcount, whead, to_walk, slot);
        give(RING_ARRIVED, acount, ahead, to_arrived, slot);
        give(RING_FREE, fcount, fhead, held && !to_walk && !to_arrived, slot);
    };
    // behind the last scattering surface: the next free slots of this wave's current chunk of the hand-over bundle; a
    // full chunk is followed by a new one from the launch-wide counter (one returning atomic per kHandChunk rays)
    uint64_t hbase = 0;
    unsigned hused = kHandChunk;                          // no chunk yet
    auto new_chunk = [&]() {
        unsigned long long v = 0;
        if (lane == 0) v = atomicAdd(&a.scat_ctl[kScatSlotsWord], (unsigned long long)kHandChunk);
        return uniform64(v);
    };
    auto hand_over = [&](bool cond, const Ray &r, double t, const KeyedDraws &d, int nis) {
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(cond);
        const unsigned cnt = (unsigned)__popcll(mask);
        if (cnt == 0) return;                             // (wave-uniform)
        uint64_t hnext = hbase;
        if (hused + cnt > kHandChunk) hnext = new_chunk();        // the batch spills into a new chunk
        if (cond) {
            const unsigned pos = hused + (unsigned)lane_prefix(mask);
            const uint64_t j = pos < kHandChunk ? hbase + pos : hnext + (pos - kHandChunk), cap = a.cont_cap;
            if (j < cap) {                                // (always: the host sizes the bundle for every chunk a launch can take)
                a.cont_pos_dir[0 * cap + j] = r.pos.x; a.cont_pos_dir[1 * cap + j] = r.pos.y; a.cont_pos_dir[2 * cap + j] = r.pos.z;
                a.cont_pos_dir[3 * cap + j] = r.dir.x; a.cont_pos_dir[4 * cap + j] = r.dir.y; a.cont_pos_dir[5 * cap + j] = r.dir.z;
                a.cont_t[j] = t;
                a.cont_draw[j] = d.c;
                a.cont_nis[j] = nis;
            }
        }
        hused += cnt;
        if (hused > kHandChunk) { hbase = hnext; hused -= kHandChunk; }
    };

#ifdef ORT_SCAT_TIMING
    const unsigned long long t_start = __builtin_readcyclecounter();
    unsigned long long t_emit = t_start, n_pass = 0;
#endif
    // a ray that raised `rare` leaves the pipeline without a trace (it has not ended, nothing of it was counted) and is
    // listed for the literal re-run from its emission, like the deferred rays of trace_queue_kernel
    auto defer = [&](bool cond, const KeyedDraws &d) {
        if (cond) a.redo_list[atomicAdd(&a.redo_ctl[0], 1u)] = (uint32_t)(a.defer_base + ((d.c >> 24) - a.first_ray));
    };
    // the rays this wave has pulled and not yet emitted, the head it pulls from, and whether all eight heads are dry
    uint64_t bnext = 0, bhi = 0;
    int hx = xcc_id() & (kScatHeads - 1), tried = 0;
    bool dry = false;
    for (;;) {
        while (bnext >= bhi && !dry) {                    // (wave-uniform) pull: at most eight failures in a wave's life
            const uint64_t base = (uint64_t)hx * a.scat_share;
            uint64_t end = base + a.scat_share;  if (end > a.n_rays) end = a.n_rays;
            unsigned long long v = 0;
            if (lane == 0 && base < end) v = atomicAdd(&a.scat_ctl[hx * kScatCtlStride], (unsigned long long)a.scat_grab);
            const uint64_t got = base + uniform64(v);
            if (base < end && got < end) {
                bnext = got;
                bhi = got + a.scat_grab < end ? got + a.scat_grab : end;
                tried = 0;
            } else {
                hx = (hx + 1) & (kScatHeads - 1);
                dry = ++tried >= kScatHeads;
            }
        }
        const bool have_new = bnext < bhi;
#ifdef ORT_SCAT_TIMING
        n_pass++;
        if (have_new) t_emit = __builtin_readcyclecounter();
#endif
        // a full wavefront of walking rays first, then of arrived ones; fresh rays while at most 64 are in flight (the
        // pool holds 128); otherwise the fuller ring runs on a partial wavefront
        const bool may_emit = have_new && wcount + acount <= 64;
        if (wcount >= 64 || (wcount > 0 && acount < 64 && !may_emit && wcount >= acount)) {
            // ---- W: one scattering event (src/lens.f90:264-281 / :315-332) for up to 64 walking rays
            bool act;
            Ray r;
            double t;
            KeyedDraws d;
            int nis, k;
            int slot;
            take(RING_WALK, wcount, whead, act, slot);
            load(act, slot, r, t, d, nis, k);
#ifdef ORT_ISA_MARKERS
            asm volatile("; ORT_STAGE_BEGIN W");
#endif
            const ort_surface &s = surf[k];
            int ended = -1;
            bool walking = act;
            r.pos = vselect(walking, vadd(r.pos, vscale(r.dir, t)), r.pos);
            const double albedo = ALB[k];
            const double u = d.peek();
            d.advance(walking);
            const bool absorbed = walking && !(u < albedo);
            ended = absorbed ? ORT_ST_LOST_BOTTLE : ended;
            walking = walking && !absorbed;
            stokes_hg<double>(r.dir, s.hgg, S.twopi, walking, d, tabs);
            double dist;
            bool at_wall, ok, rare = false;
            tauint<double, KeyedDraws, true>(r, s.mua, s.mus, s.cy, s.cz, s.scat_radius, walking, d, dist, at_wall, ok, nis, &rare);
            const bool bad = walking && rare;
            walking = walking && !rare;
            const bool lostw = walking && !ok;
            ended = lostw ? ORT_ST_NO_INTERSECTION : ended;
            t = (walking && ok) ? dist : t;
            const bool out = sqrt(r.pos.x * r.pos.x + r.pos.z * r.pos.z) >= s.scat_radius;        // sic: x, z
            const bool still = walking && ok && !out && !at_wall;
            const bool arrived = walking && ok && !still;
#ifdef ORT_ISA_MARKERS
            asm volatile("; ORT_STAGE_END W");
#endif
            defer(bad, d);
            route(act, slot, still, arrived && k < klast, r, t, d, nis, k);
            hand_over(arrived && k >= klast, r, t, d, nis);
            if (act && ended >= 0) end_ray(ended, nis);
            __builtin_amdgcn_wave_barrier();
        } else if (acount >= 64 || (acount > 0 && !may_emit)) {
            // ---- A: the rest of the surface step for up to 64 rays that reached the wall, then the next surface
            bool act;
            Ray r;
            double t;
            KeyedDraws d;
            int nis, k;
            int slot;
            take(RING_ARRIVED, acount, ahead, act, slot);
            load(act, slot, r, t, d, nis, k);
            const ort_surface &s = surf[k];
            const unsigned flags = s.flags;
            int ended = -1;
            const bool back = act && (flags & ORT_F_SCATTER) != 0 && r.dir.z < 0.;       // lens.f90:283, :334
            ended = back ? ORT_ST_LOST_BOTTLE : ended;
            const bool live = act && !back;
            r.pos = vselect(live, vadd(r.pos, vscale(r.dir, t)), r.pos);
            const Vec Nraw = {0.0, s.cy - r.pos.y, s.cz - r.pos.z};                      // lens.f90:288-290
            // normal and Fresnel step in their filtered forms (a ray that left its walk by the `out` test is not on the wall:
            // no estimate of |N| holds, hence vnormalise_f)
            bool rare = false;
            const Vec N = vnormalise_f<true, double>(Nraw, live, rare, true);
            const double u = d.peek();
            d.advance(live);
            const bool reflected = reflect_refract<true, true, double>(r.dir, N, s.n1, s.n2, s.eta, s.eta * s.eta, u, live, rare);
            const bool bad1 = live && rare;
            const bool dies = live && !rare && reflected && (flags & ORT_F_SKIP_ON_REFLECT) != 0;
            ended = dies ? ((flags & ORT_F_BOTTLE) ? ORT_ST_LOST_BOTTLE : ORT_ST_LOST_TELESCOPE) : ended;
            const bool enter = live && !rare && !dies;  // (rays arriving at the LAST wall never come here: handed over)
            bool walking = false, arrived = false, rare2 = false;
            const int k1 = enter ? k + 1 : k;
            scat_enter(surf, k1, kind0, enter, r, d, nis, t, walking, arrived, ended, rare2);
            defer(bad1 || rare2, d);
            route(act, slot, walking, arrived && k1 < klast, r, t, d, nis, k1);
            hand_over(arrived && k1 >= klast, r, t, d, nis);
            if (act && ended >= 0) end_ray(ended, nis);
            __builtin_amdgcn_wave_barrier();
        } else if (may_emit) {
            // ---- E: 64 fresh rays
            const uint64_t i = bnext + (uint64_t)lane;
            const bool act = i < bhi;
            bnext += 64;
            const uint64_t ic = act ? i : bhi - 1;
            Ray r;
            KeyedDraws d;
            d.init_keyed(a.rng_base, a.first_ray + ic, 0);
            int ended = emit<double, ANYSRC>(S, a.phase, r, d, a.first_ray + ic, a.img_cdf, a.strict != 0);
            int nis = 0;
            double t = 0.;
            bool walking, arrived, rare = false;
            scat_enter(surf, 0, kind0, act && ended < 0, r, d, nis, t, walking, arrived, ended, rare);
            defer(rare, d);
            bool held;
            int slot;
            take(RING_FREE, fcount, fhead, held, slot);      // 64 of them: at most 64 rays are in flight (may_emit)
            route(held, slot, walking, arrived && klast > 0, r, t, d, nis, 0);
            hand_over(arrived && klast <= 0, r, t, d, nis);
            if (act && ended >= 0) end_ray(ended, nis);
            __builtin_amdgcn_wave_barrier();
        } else {
            break;                                           // nothing in flight, every head dry
        }
    }
#ifdef ORT_SCAT_TIMING
    {
        const unsigned wid = blockIdx.x * kScatWaves + wave;
        if (lane == 0 && wid < 16384) {
            g_scat_times[4 * wid + 0] = t_start; g_scat_times[4 * wid + 1] = t_emit;
            g_scat_times[4 * wid + 2] = __builtin_readcyclecounter(); g_scat_times[4 * wid + 3] = n_pass;
        }
    }
#endif
    if (hused < kHandChunk)                                  // what is left of the wave's last chunk holds no ray
        for (uint64_t j = hbase + hused + (uint64_t)lane; j < hbase + kHandChunk && j < a.cont_cap; j += 64) a.cont_draw[j] = kNoRay;
    atomicAdd(&blk[0], lost); atomicAdd(&blk[1], isect); atomicAdd(&blk[3], help3);
    __syncthreads();
    if (threadIdx.x < 4 && blk[threadIdx.x])
        atomicAdd(&a.counters[2 * threadIdx.x + (a.phase - 1)], (unsigned long long)blk[threadIdx.x]);
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
        const char *e = getenv("ORT_MAX_BLOCKS");        // development knob
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
// 2.07e11).  ORT_CHUNK_LOG2 (development knob, read once) makes them smaller: the tests cut small traces into several launches.
uint64_t chunk_rays()
{
    static const uint64_t chunk = [] {
        const int lg = env_int("ORT_CHUNK_LOG2", 27);
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
constexpr int kHeadBlocks = 1280, kHeadPercent = 86, kTailBatches = 6;
int plan_ranges(TraceArgs &a)
{
    static const int head_blocks = env_int("ORT_HEAD_BLOCKS", kHeadBlocks), head_pct = env_int("ORT_HEAD_PERCENT", kHeadPercent),
                     tail_batches = env_int("ORT_TAIL_BATCHES", kTailBatches);
    const uint64_t n = a.n_rays, per_block = 64ull * kWavesPerBlock;
    if (n < (uint64_t)head_blocks * per_block * 4 || head_pct >= 100) {       // equal ranges
        int grid = grid_for(n);
        const uint64_t batches = (n + 63) / 64, blocks = (batches + kWavesPerBlock - 1) / kWavesPerBlock;
        if (blocks < (uint64_t)grid) grid = (int)blocks;
        const uint64_t nwaves = (uint64_t)grid * kWavesPerBlock;
        a.head_blocks = (uint32_t)grid;
        a.head_rays = n;
        a.head_chunk = (((n + nwaves - 1) / nwaves) + 63) & ~63ull;
        a.tail_chunk = 64;
        return grid;
    }
    const uint64_t hw = (uint64_t)head_blocks * kWavesPerBlock;
    a.head_blocks = (uint32_t)head_blocks;
    a.head_chunk = ((n / 100 * (uint64_t)head_pct / hw) + 63) & ~63ull;
    a.head_rays = a.head_chunk * hw < n ? a.head_chunk * hw : n;
    a.tail_chunk = 64ull * (uint64_t)tail_batches;
    const uint64_t rest = n - a.head_rays, tb = a.tail_chunk * kWavesPerBlock;
    return head_blocks + (int)((rest + tb - 1) / tb);
}

int redo_blocks()
{
    static int n = 0;
    if (!n) {
        const char *e = getenv("ORT_REDO_BLOCKS");       // development knob
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

static void ring_cull_threshold(const ort_system *sys, bool is_ring_program, double *cull, float *cullf)
{
    *cull = HUGE_VAL; *cullf = HUGE_VALF;
    if (!is_ring_program || getenv("ORT_NO_RING_CULL")) return;
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

static void note_system(ort_ctx *c, const ort_system *sys)
{
    c->emitter[0] = sys->emitter[0]; c->emitter[1] = sys->emitter[1];
    for (int p = 0; p < 2; ++p) {
        c->scatter[p] = false;
        for (int k = 0; k < sys->n_surfaces[p]; ++k)
            if (sys->surfaces[p][k].flags & ORT_F_SCATTER) c->scatter[p] = true;
    }
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
    c->cont_prog[1] = c->scat_k0[1] > 0 && matches_behind<PROG_POINT_WALKED>(sys, c->scat_k0[1]) && !getenv("ORT_NO_PROGRAMS");
    c->prog[0] = c->prog[1] = PROG_GENERIC;
#define ORT_MATCH(P) if (matches<P>(sys)) c->prog[Prog<P>::phase - 1] = P;
    ORT_PROGRAMS(ORT_MATCH)
    ORT_SOURCE_PROGRAMS(ORT_MATCH)
#undef ORT_MATCH
    if (getenv("ORT_NO_PROGRAMS")) c->prog[0] = c->prog[1] = PROG_GENERIC;      // development knob (A/B)
    ring_cull_threshold(sys, c->prog[0] != PROG_GENERIC && sys->emitter[0] == ORT_EMIT_RING, &c->ring_cull, &c->ring_cullf);
    c->ring_cull_word = cull_word_of<double>(sys->ring_lens_r2, c->ring_cull);
    c->ring_cull_wordf = cull_word_of<float>((float)sys->ring_lens_r2, c->ring_cullf);
}

// system + derived per-surface constants -> the next device slot, asynchronously (the staging copy is the
// slot's own pinned host buffer).  A slot is reused kSysSlots systems later; by then the event recorded when it
// was retired has normally long passed.
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
    h.sys = *sys;
    for (int p = 0; p < 2; ++p)
        for (int k = 0; k < ORT_MAX_SURFACES; ++k) h.aux[p][k] = make_aux<double>(sys->surfaces[p][k]);
    axial_start<double>(h.aux[1][0], sys->surfaces[1][0], sys->point_offset);           // OPT_AXIAL_START (point programs, step 0)
    convert_system(h.sysf, *sys);
    for (int p = 0; p < 2; ++p)
        for (int k = 0; k < ORT_MAX_SURFACES; ++k) h.auxf[p][k] = make_aux<float>(h.sysf.surfaces[p][k]);
    axial_start<float>(h.auxf[1][0], h.sysf.surfaces[1][0], h.sysf.point_offset);
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

// The lean queued kernel (default emitters, no scattering, filtered predicates): specialised for
// the surface program the staged system matches, generic otherwise.
template <class T, bool FILT = true>
static void launch_lean(ort_ctx *c, int mode, const TraceArgs &a, int grid)
{
#define ORT_LAUNCH(K) hipExtLaunchKernelGGL(K, dim3(grid), dim3(kBlock), 0, c->stream, c->launch_ev[0], c->launch_ev[1], 0, a)
    // a program's draw indices are compile-time constants: they assume the emitter's own number of
    // draws in front of the first surface (resident bundles may come with another draw_base)
    int prog = c->prog[a.phase - 1];
    if (mode != MODE_FUSED && a.draw_base != (a.phase == 1 ? 4 : 2)) prog = PROG_GENERIC;
    if (a.strict) prog = PROG_GENERIC;                   // strict libm emitters live in the generic kernels (variant bit 6)
#define ORT_CASE(P)                                                                                        \
    case P:                                                                                                \
        if (mode == MODE_FUSED) ORT_LAUNCH((trace_queue_kernel<MODE_FUSED, FILT, false, T, P>));           \
        else ORT_LAUNCH((trace_queue_kernel<MODE_RESIDENT, FILT, false, T, P>));                           \
        break;
    switch (prog) {
        ORT_PROGRAMS(ORT_CASE)
    default:
        if (mode == MODE_FUSED) ORT_LAUNCH((trace_queue_kernel<MODE_FUSED, FILT, false, T>));
        else ORT_LAUNCH((trace_queue_kernel<MODE_RESIDENT, FILT, false, T>));
    }
#undef ORT_CASE
#undef ORT_LAUNCH
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

// One kernel of the trace family on `grid` workgroups.
static void launch_one(ort_ctx *c, int mode, const TraceArgs &a, int grid, bool queued, bool filt, bool anysrc)
{
    const bool scat = c->scatter[a.phase - 1];
#define ORT_LAUNCH(K) hipExtLaunchKernelGGL(K, dim3(grid), dim3(kBlock), 0, c->stream, c->launch_ev[0], c->launch_ev[1], 0, a)
    if (c->precision == 2) {
        // fast fp64 (ort_fastd.h): FMA contraction, Newton divide / Goldschmidt sqrt; ~1e-13 from exact
        if (mode == MODE_DEBUG) ORT_LAUNCH((trace_kernel<MODE_DEBUG, true, fastd, true>));
        else if (!queued) {                                   // the literal re-run of deferred rays
            if (mode == MODE_FUSED) ORT_LAUNCH((trace_kernel<MODE_FUSED, false, fastd, true>));
            else ORT_LAUNCH((trace_kernel<MODE_RESIDENT, false, fastd, true>));
        } else if (anysrc) {
            if (mode == MODE_FUSED) ORT_LAUNCH((trace_queue_kernel<MODE_FUSED, true, true, fastd>));
            else ORT_LAUNCH((trace_queue_kernel<MODE_RESIDENT, true, true, fastd>));
        } else {
            launch_lean<fastd>(c, mode, a, grid);
        }
    } else if (c->precision == 1) {
        // fp32 path (BASELINE configs[4]): always the filtered forms, never a deferral (ort_device.h kLoose) — in every kernel,
        // so that they agree bit for bit; queued program kernels for the default emitters in clear media, the lockstep
        // kernel for everything else
        if (mode == MODE_DEBUG) ORT_LAUNCH((trace_kernel<MODE_DEBUG, true, float, true>));
        else if (queued && !anysrc) launch_lean<float, true>(c, mode, a, grid);
        else if (queued && mode == MODE_FUSED && !scat && c->prog[a.phase - 1] > PROG_LIST_MASK) {
            switch (c->prog[a.phase - 1]) {                   // the other bulk light sources: their own program kernels
#define ORT_CASE(P) case P: ORT_LAUNCH((trace_queue_kernel<MODE_FUSED, true, false, float, P>)); break;
                ORT_SOURCE_PROGRAMS(ORT_CASE)
#undef ORT_CASE
            }
        }
        else if (mode == MODE_FUSED) ORT_LAUNCH((trace_kernel<MODE_FUSED, true, float, true>));
        else ORT_LAUNCH((trace_kernel<MODE_RESIDENT, true, float, true>));
    } else if (mode == MODE_DEBUG) {
        if (filt) ORT_LAUNCH((trace_kernel<MODE_DEBUG, true, double, true>));
        else ORT_LAUNCH((trace_kernel<MODE_DEBUG, false, double, true>));
    } else if (anysrc && filt && queued && mode == MODE_FUSED && !scat && c->prog[a.phase - 1] > PROG_LIST_MASK && !a.strict) {
        // the other bulk light sources in front of a default surface list: their own program kernels
        switch (c->prog[a.phase - 1]) {
#define ORT_CASE(P) case P: ORT_LAUNCH((trace_queue_kernel<MODE_FUSED, true, false, double, P>)); break;
            ORT_SOURCE_PROGRAMS(ORT_CASE)
#undef ORT_CASE
        }
    } else if (anysrc || !filt || !queued) {
        // alternate emitters and the A/B variants share the generic instantiations
        if (mode == MODE_FUSED) {
            if (queued) {
                if (filt && !scat) ORT_LAUNCH((trace_queue_kernel<MODE_FUSED, true, true, double, PROG_GENERIC, false>));   // other emitters, clear media
                else if (filt) ORT_LAUNCH((trace_queue_kernel<MODE_FUSED, true, true, double>));
                else ORT_LAUNCH((trace_queue_kernel<MODE_FUSED, false, true, double>));
            }
            else { if (filt) ORT_LAUNCH((trace_kernel<MODE_FUSED, true, double, true>)); else ORT_LAUNCH((trace_kernel<MODE_FUSED, false, double, true>)); }
        } else {
            if (queued) {
                if (filt && !scat) ORT_LAUNCH((trace_queue_kernel<MODE_RESIDENT, true, true, double, PROG_GENERIC, false>));
                else if (filt) ORT_LAUNCH((trace_queue_kernel<MODE_RESIDENT, true, true, double>));
                else ORT_LAUNCH((trace_queue_kernel<MODE_RESIDENT, false, true, double>));
            }
            else { if (filt) ORT_LAUNCH((trace_kernel<MODE_RESIDENT, true, double, true>)); else ORT_LAUNCH((trace_kernel<MODE_RESIDENT, false, double, true>)); }
        }
    } else {
        // the default emitters (ring / point) without scattering have their own, leaner instantiation
        launch_lean<double>(c, mode, a, grid);
    }
#undef ORT_LAUNCH
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
        HIP_TRY(hipMalloc(&c->d_redo_list, want * sizeof(uint32_t)));
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
    static const uint64_t chunk = 1ull << env_int("ORT_SCAT_CHUNK_LOG2", kScatterChunkLog2);     // development knob
    return chunk;
}
// Workgroups of scatter_front_kernel for a launch of n rays: one per CU (12 persistent wavefronts each), fewer when the
// launch has fewer than 12 batches of 64 rays per CU.  What a wave takes per pull: 1/32 of its share of the launch, at
// least one batch, at most four (the end of a launch is as ragged as one pull is long).
static unsigned scatter_groups(const ort_ctx *c, uint64_t n)
{
    const uint64_t batches = (n + 63) / 64, want = (batches + kScatWaves - 1) / kScatWaves;
    const uint64_t most = (uint64_t)env_int("ORT_SCAT_GROUPS", c->n_cus);         // development knob
    return (unsigned)(want < most ? (want ? want : 1) : most);
}
static uint32_t scatter_grab(uint64_t n, unsigned groups)
{
    static const int forced = env_int("ORT_SCAT_GRAB", 0);                        // development knob (batches per pull)
    const uint64_t per_wave = (n + 63) / 64 / ((uint64_t)groups * kScatWaves);
    uint64_t g = per_wave / 32;
    if (g < 1) g = 1;
    if (g > 4) g = 4;
    if (forced > 0) g = (uint64_t)forced;
    return (uint32_t)(64 * g);
}
// fp32 queued launches.  The hit log holds the launches since the last binning (kHitLogEntries 16-bit entries per part:
// 1.34 GB of the 288 GB), the directory eight words per traced
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
            const uint64_t want = rays + 8 > kHitLogEntries ? rays + 8 : kHitLogEntries;
            const uint64_t cap = (want + 63) & ~63ull;          // (+8: bin_log_kernel reads eight entries at a time)
            HIP_TRY(hipMalloc(&c->d_hit_log, (size_t)kBinTiles * cap * sizeof(uint16_t)));
            c->hit_log_cap = cap;
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
    if (a0.first_ray > ORT_MAX_RAY_INDEX || a0.n_rays > ORT_MAX_RAY_INDEX - a0.first_ray)
        return fail(ORT_E_INVALID, "ray indices reach beyond 2^40 (ORT_MAX_RAY_INDEX): the keyed draw counter holds 40 bits of ray index");
    if (a0.n_rays == 0) return ORT_OK;
    const bool queued = (c->variant & 1) && mode != MODE_DEBUG && !a0.wide;     // 53-bit draws: the lockstep kernel
    const bool filt = (c->variant & 2) == 0 && c->precision != 1;      // fp32: literal predicates, nothing is deferred
    const bool anysrc_emitter = c->emitter[a0.phase - 1] != (a0.phase == 1 ? ORT_EMIT_RING : ORT_EMIT_POINT);
    const bool anysrc = anysrc_emitter || c->scatter[a0.phase - 1];
    // The queued filtered kernel defers the rays that sit on a decision boundary to a list, which
    // the literal lockstep kernel traces right after it.  Every ray of a launch could be on it
    // (an axial beam meets every flat face at costt == 1), so a launch covers at most chunk_rays().
    const bool deferring = queued && filt;
    const uint64_t total = a0.n_rays;
    // scattering media, exact fp64, the default kernel variant: the three-stage pipeline (scatter_front_kernel)
    const bool pipeline = mode == MODE_FUSED && deferring && c->precision == 0 && c->scatter[a0.phase - 1] &&
                          c->scat_k0[a0.phase - 1] > 0 && (c->variant & 16) == 0;
    // fp32 hit log: worth its second kernel (~10 us + the launch gap) where a large share of the rays is binned — the point loop
    // (42 % of its rays: -28 % per step); the ring loop bins 1 % of its rays and keeps the atomics.  ORT_HIT_LOG = 1 never,
    // 2 the point loop (default), 3 both loops (development knob; the image is the same bit for bit)
    static const int hit_log_mode = env_int("ORT_HIT_LOG", 2);
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
        const int grid = queued ? plan_ranges(a) : grid_for(a.n_rays);
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
            if (anysrc_emitter) hipLaunchKernelGGL(scatter_front_kernel<true>, dim3(groups), dim3(64 * kScatWaves), 0, c->stream, a);
            else hipLaunchKernelGGL(scatter_front_kernel<false>, dim3(groups), dim3(64 * kScatWaves), 0, c->stream, a);
            HIP_TRY(hipGetLastError());
            if (c->cont_prog[a.phase - 1])
                hipLaunchKernelGGL((trace_queue_kernel<MODE_CONTINUE, true, false, double, PROG_POINT_WALKED, false>), dim3(grid), dim3(kBlock), 0,
                                   c->stream, a);
            else
                hipLaunchKernelGGL((trace_queue_kernel<MODE_CONTINUE, true, false, double, PROG_GENERIC, false>), dim3(grid), dim3(kBlock), 0,
                                   c->stream, a);
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
} g_rccl;

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
        for (int i = 0; i < g_rccl.n; ++i) (void)g_rccl.CommDestroy(g_rccl.comms[i]);
        g_rccl.n = 0;
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
    // ORT_FAULT_ALLREDUCE (tests): the first collective is given an invalid datatype.
    ncclResult_t bad = ncclSuccess;
    const char *what = nullptr;
    const bool fault = getenv("ORT_FAULT_ALLREDUCE") != nullptr;
    for (int i = 0; i < n && bad == ncclSuccess; ++i) {
        ort_ctx *c = ctxs[i];
        bad = g_rccl.AllReduce(c->d_image, c->d_image, ORT_IMAGE_BINS, (fault && i == 0) ? (ncclDataType_t)99 : ncclInt32, ncclSum, g_rccl.comms[i], c->stream);
        if (bad != ncclSuccess) { what = "ncclAllReduce(image)"; break; }
        bad = g_rccl.AllReduce(c->d_counters, c->d_counters, ORT_NUM_COUNTERS, ncclUint64, ncclSum, g_rccl.comms[i], c->stream);
        if (bad != ncclSuccess) what = "ncclAllReduce(counters)";
    }
    const ncclResult_t end = g_rccl.GroupEnd();
    if (bad != ncclSuccess) return rccl_fail(what, bad);
    if (end != ncclSuccess) return rccl_fail("ncclGroupEnd", end);
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
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = close_group(c); if (rc) return rc; }     // the re-run uses the arithmetic of its launches
    c->precision = precision;
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
