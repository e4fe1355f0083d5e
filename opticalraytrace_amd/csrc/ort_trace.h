// ort_trace.h — the trace kernels of libort_hip.so as templates, shared by the translation units that instantiate them
// (ort_k_*.hip: one family of instantiations each, so that a build runs in parallel and an A/B of one family rebuilds
// one unit) and by the host side (ort_hip.hip), which only sees TraceArgs and the launchers of ort_launch.h.
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
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/ort.h"
#include "ort_device.h"

namespace ortk {
using namespace ort;

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
    // SCHED_PULL_* (experiments): eight work heads (one 128-byte line each, zero before the launch), head x hands out batches
    // [0, pull_share_b) of 64 rays of the rays [head_rays + x share, head_rays + (x + 1) share); waves per head; grab bounds
    unsigned int *pull_ctl;
    uint32_t pull_share_b, pull_wph, pull_min, pull_max;
    uint32_t cull_word, cull_wordf;
    uint64_t cull_wide;          // the same border for the 53-bit draws of ORT-RNG-v2w (fp64): the ray dies iff (h >> 11) > cull_wide
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

constexpr int kBinUnits = 51, kBinTiles = 5, kBinTile = (ORT_IMAGE_N * ORT_IMAGE_N + kBinTiles - 1) / kBinTiles, kBinDirMax = 2048;
constexpr int kBinThreads = 1024, kBinDirWords = 8;
static_assert(kBinTile <= 65536, "a log entry (bin / kBinTiles) must fit 16 bits");

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
template <int MODE, bool FILT, class T, bool ANYSRC, bool BATCH = false>
__device__ __forceinline__ void trace_body(const TraceArgs &a)
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
            KeyedDrawsT<1> d;
            d.init_keyed(a.rng_base, a.first_ray + ic, have_in ? a.draw_base : 0);
            d.set_wide(a.wide != 0);
            if (!have_in) {
                const int est = emit<T, ANYSRC>(S, a.phase, r, d, a.first_ray + ic, a.img_cdf, a.strict != 0);
                st = est < 0 ? st : est;
            }
            const RayT<T> r0 = r;
            const KeyedDrawsT<1> d0 = d;
            const int st0 = st;
            walk<FILT, T, ANYSRC, false>(S, surf, AUX, 0, ns, r, d, nis, st, xp, yp,
                                         [&](RayT<T> &rr, KeyedDrawsT<1> &dd, int &ss) { rr = r0; dd = d0; ss = st0; });
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
                if (!BATCH || a.image != nullptr) bin_hit(layer, xp, yp, a.replicas != nullptr);
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

template <int MODE, bool FILT, class T, bool ANYSRC>
__global__ __launch_bounds__(kBlock, ORT_MIN_WAVES) void trace_kernel(TraceArgs a)
{
    trace_body<MODE, FILT, T, ANYSRC, false>(a);
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
#ifndef ORT_SCAT_WAVES
#define ORT_SCAT_WAVES 12
#endif
constexpr int kScatWaves = ORT_SCAT_WAVES;              // wavefronts per workgroup of scatter_front_kernel = per CU (LDS and 168 VGPRs allow no more)
constexpr unsigned kHandChunk = 256;        // hand-over slots a wave allocates at a time
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kQueueCap = 128;      // >= 63 leftover + 64 new survivors
constexpr uint64_t kMaxRange = 65472;   // rays of one wave's static range (a queued ray index is 16 bits relative to the range's first ray)
constexpr int kQueueFields = 6;     // px py pz dx dy dz (+ the draw state: its own array)

__device__ inline int lane_prefix(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// RNG (bit mask, exact fp64): RNG_STRICT = the program's emitter calls glibc's own sin / cos / sincos (kernel variant bit 6: the
// emitted ray equals the reference's bit for bit; the generic kernels take the same choice at run time, TraceArgs.strict);
// RNG_WIDE = 53-bit draws, stream ORT-RNG-v2w (kernel variant bit 5: one hash per draw)
constexpr int RNG_STRICT = 1, RNG_WIDE = 2;
// SCHED (experiments of ort_k_exp.hip; the production instantiations are SCHED_STATIC): how a wave comes by its ray ranges —
// SCHED_STATIC: plan_ranges' two-level static plan; SCHED_PULL_V / SCHED_PULL_S: PERSISTENT waves that pull batches from
// eight work heads, one per XCD (TraceArgs.pull_ctl; as scatter_front_kernel does), with a returning vector atomic of lane 0
// (its result comes back in order behind the wave's earlier image atomics: vmcnt) or with a SCALAR atomic (s_atomic_add:
// lgkmcnt, independent of the vector memory queue).  NOBIN: the image atomic is left out (what the hits cost; the upper
// bound of what logging them instead could gain).
constexpr int SCHED_STATIC = 0, SCHED_PULL_V = 1, SCHED_PULL_S = 2;
// BATCH (trace_batch_kernel below): the arguments are ONE simulation's of a multi-system launch — hits go straight into that
// simulation's image (no replicas), or nowhere when it wants none (image == nullptr: the hit is counted, not binned)
template <int MODE, bool FILT, bool ANYSRC, class T, int PROG = PROG_GENERIC, bool SCAT = ANYSRC, int RNG = 0, int SCHED = SCHED_STATIC,
          bool NOBIN = false, bool BATCH = false>
__device__ __forceinline__ void trace_queue_body(const TraceArgs &a)
{
    static_assert(SCHED == SCHED_STATIC || (MODE == MODE_FUSED && PROG != PROG_GENERIC), "pulled ranges: fused program kernels");
    static_assert(PROG == PROG_GENERIC || (!ANYSRC && !SCAT && FILT), "programs exist for the lean kernels only (filtered forms)");
    static_assert(RNG == 0 || std::is_same<T, double>::value, "strict emitters and 53-bit draws belong to the exact fp64 path");
    static_assert(!(RNG & RNG_STRICT) || (PROG != PROG_GENERIC && MODE == MODE_FUSED), "STRICT is a constant of the fused program kernels");
    constexpr bool STRICT = (RNG & RNG_STRICT) != 0, WIDE = (RNG & RNG_WIDE) != 0;
    // fp32 (kLoose): the filtered forms decide every lane — nothing is deferred, `rare` is not looked at
    constexpr bool DEFER = FILT && !kLoose<T>;
    __shared__ typename SysTypes<T>::Sys S;
    // 26.6 KB per workgroup of a program kernel in fp64 (6 workgroups per CU's 160 KB), 14 KB in fp32
    constexpr bool fixed = PROG != PROG_GENERIC;        // surface program known at compile time
    using QT = typename std::conditional<std::is_same<T, float>::value, float, double>::type;
    constexpr bool sdraws = prog_static_draws<PROG>();  // every lane at the same, compile-time draw index (ProgDraws)
    // static draws: what is queued of a ray's draw state is its index — as 16 bits RELATIVE to the wave's range where the range is
    // static (plan_ranges keeps a range below kMaxRange rays): 1 KB less per queue, which is what lets a CU hold six
    // workgroups of a ring program (two queues: 28.7 KB -> 26.6 KB) instead of five
    constexpr bool SHORTQ = sdraws && SCHED == SCHED_STATIC && MODE != MODE_CONTINUE;
    using QI = typename std::conditional<SHORTQ, uint16_t, uint32_t>::type;
    using QD = typename std::conditional<sdraws, QI, uint64_t>::type;
    __shared__ QT Q[kWavesPerBlock][kQueueFields][kQueueCap];
    __shared__ QD QDRAW[kWavesPerBlock][kQueueCap];
    // ring programs, fused: segment 0 (below) culls the rays that are certain to miss the first aperture
    // before anything is emitted; the others wait here (ray index in the launch) for a full wave
    constexpr bool PRE = prog_culls<PROG>() && MODE == MODE_FUSED;
    __shared__ QI CQ[kWavesPerBlock][PRE ? kQueueCap : 1];
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
    QI *cq = CQ[wave];
    int *qn = QN[wave];
    using PD = ProgDrawsT<WIDE>;
    using DrawsT = typename std::conditional<sdraws, PD, KeyedDrawsT<WIDE ? 2 : 0>>::type;
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
    } else if constexpr (SCHED != SCHED_STATIC) {
        // every wave's first range is static (head_rays of the launch in head_chunk per wave; 0: none), the rest is pulled
        const uint64_t wid = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
        lo = wid * a.head_chunk; if (lo > a.head_rays) lo = a.head_rays;
        hi = lo + a.head_chunk;  if (hi > a.head_rays) hi = a.head_rays;
    } else {
        // BATCH: blockIdx.x is the SIMULATION (the fastest index of the dispatch order: the long workgroups of all simulations
        // of the launch come first, their short ones last), blockIdx.y the workgroup within it
        const uint32_t bid = BATCH ? blockIdx.y : blockIdx.x;
        const bool head = bid < a.head_blocks;
        const uint64_t wid = (uint64_t)(head ? bid : bid - a.head_blocks) * kWavesPerBlock + wave;
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
            if constexpr (!NOBIN) { if (!logging && (!BATCH || a.image != nullptr)) bin_hit(layer, xp, yp, a.replicas != nullptr); }
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
    const uint64_t z0 = PD::zray_at(a.rng_base, a.first_ray);      // ProgDraws::init_index (a launch holds < 2^32 rays)
    // segment 0: zray of ray next + lane = znext (wave-uniform) + zlane
    uint64_t znext = z0 + (kGolden << PD::kRayShift) * next;
    const uint64_t zlane = (kGolden << PD::kRayShift) * (uint64_t)lane;
    unsigned int culled_wave = 0;     // rays segment 0 culled (wave-uniform): each is lost after one intersection
    // SCHED_PULL_*: the head this wave pulls from (its XCD's), whether all eight are dry, the batches the head had left at the last pull
    int hx = xcc_id() & 7, tried = 0;
    bool dry = false;
    uint32_t left_b = a.pull_share_b;
    for (;;) {
        if constexpr (SCHED != SCHED_STATIC) {
            while (next >= hi && !dry) {                  // (wave-uniform) pull: at most eight failures in a wave's life
                // guided: 1 / (2 waves per head) of what the head had left, pull_min .. pull_max batches of 64 rays
                uint32_t grab = left_b / (2u * a.pull_wph);
                grab = grab < a.pull_min ? a.pull_min : (grab > a.pull_max ? a.pull_max : grab);
                uint32_t got;
                unsigned int *head = a.pull_ctl + hx * 32;                // one 128-byte line per head
                if constexpr (SCHED == SCHED_PULL_S) {
                    uint32_t v = grab;
                    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(head) : "memory");
                    got = v;
                } else {
                    uint32_t v = 0;
                    if (lane == 0) v = atomicAdd(head, grab);
                    got = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
                }
                if (got < a.pull_share_b) {
                    const uint64_t base = a.head_rays + (uint64_t)hx * a.pull_share_b * 64ull;
                    lo = base + (uint64_t)got * 64ull;
                    uint64_t end = base + (uint64_t)a.pull_share_b * 64ull;  if (end > n) end = n;
                    hi = lo + (uint64_t)grab * 64ull;  if (hi > end) hi = end;
                    if (lo > end) lo = end;
                    next = lo;
                    znext = z0 + (kGolden << PD::kRayShift) * next;
                    left_b = a.pull_share_b - got;
                    tried = 0;
                } else {
                    hx = (hx + 1) & 7;
                    left_b = a.pull_share_b;
                    dry = ++tried >= 8;
                }
            }
        }
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
            typename std::conditional<sdraws, uint32_t, uint64_t>::type dw = 0;      // the queued image of the draw state (static draws: the index in the launch)
            int nis = 0, xp = 0, yp = 0, st = act ? -1 : ORT_ST_NA_REJECT;
            {
                r.pos = {T(q[0][slot]), T(q[1][slot]), T(q[2][slot])};
                r.dir = {T(q[3][slot]), T(q[4][slot]), T(q[5][slot])};
                if constexpr (SHORTQ) dw = (uint32_t)lo + (uint32_t)qd[slot];
                else dw = qd[slot];
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
                i = act ? (SHORTQ ? lo : 0ull) + (uint64_t)cq[(chead + lane) & (kQueueCap - 1)] : lo;
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
                if constexpr (fixed) est = emit<T, false, FILT, Prog<PROG>::emitter>(*csys, phase, r, d, a.first_ray + ic, a.img_cdf, rare, &img_hint, STRICT);
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
                if constexpr (sdraws) qd[slot] = (QI)(SHORTQ ? i - lo : i);
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
                bool dies;
                if constexpr (WIDE) {
                    // v2w: draw 2 has a hash of its own, u = (h >> 11) 2^-53; the test is taken on those 53 bits (cull_wide)
                    const uint64_t h = mix64((znext + zlane) + kGolden * 3ull);
                    dies = (h >> 11) > a.cull_wide;
                } else {
                    const uint64_t h = mix64((znext + zlane) + kGolden * 2ull);   // pair 1 of ray next + lane: its draws 2 and 3
                    const uint32_t w = (uint32_t)(h >> 32);                       // draw 2, the third (ProgDraws::at<T, 2>)
                    if constexpr (std::is_same<T, float>::value) dies = w > a.cull_wordf;
                    else dies = w > a.cull_word;
                }
                const bool cand = act && !dies;
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(cand);
                if (cand) cq[(chead + ccount + lane_prefix(mask)) & (kQueueCap - 1)] = (QI)((SHORTQ ? (uint32_t)(next - lo) : (uint32_t)next) + (uint32_t)lane);
                const int passed = __popcll(mask);
                culled_wave += (unsigned)((left < 64u ? (int)left : 64) - passed);
                ccount += passed;
                znext += (kGolden << PD::kRayShift) * 64ull;
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

template <int MODE, bool FILT, bool ANYSRC, class T, int PROG = PROG_GENERIC, bool SCAT = ANYSRC, int RNG = 0, int SCHED = SCHED_STATIC,
          bool NOBIN = false>
__global__ __launch_bounds__(kBlock, PROG != PROG_GENERIC ? 4 : ORT_MIN_WAVES) void trace_queue_kernel(TraceArgs a)
{
    trace_queue_body<MODE, FILT, ANYSRC, T, PROG, SCAT, RNG, SCHED, NOBIN, false>(a);
}

// ---------------------------------------------------------------------------
// Multi-system launches (runner.py:113-261 runs one PROCESS per settings file): ONE launch traces one loop of
// MANY simulations that share a surface program.  gridDim.x = simulations of the launch, blockIdx.x -> that simulation's
// TraceArgs in a device table (its own staged system, image, counters, re-run list, ray ranges over blockIdx.y — long ranges
// on its first workgroups, short ones on the rest, as plan_ranges cuts a launch of its own), read through
// scalar loads like a kernel argument; everything per ray is the fused program kernel's (the same body).  At 1e6 rays a
// simulation is ~40 us of work inside ~25 us of ramp and drain of a launch of its own: batched, the chip sees one launch of
// n x 1e6 rays.  The rays a simulation defers go to ITS list; trace_batch_rerun_kernel closes all of them in one launch.
// ---------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) TraceArgs *batch_args_t;
__device__ __forceinline__ void load_batch_args(TraceArgs &a, const TraceArgs *batch, uint32_t simulation)
{
    const batch_args_t p = (batch_args_t)batch + simulation;
    __builtin_memcpy(&a, p, sizeof(TraceArgs));
}

template <int PROG>
__global__ __launch_bounds__(kBlock, 4) void trace_batch_kernel(const TraceArgs *batch)
{
    TraceArgs a;
    load_batch_args(a, batch, blockIdx.x);
    trace_queue_body<MODE_FUSED, true, false, double, PROG, false, 0, SCHED_STATIC, false, true>(a);
}

}  // namespace ortk
