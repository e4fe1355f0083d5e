// ort_scatter.h — the front kernel of the scattering pipeline, see the comment block below.
#pragma once
#include "ort_trace.h"

namespace ortk {

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
template <class KD>
__device__ inline void scat_enter(const ort_surface *surf, int k, int kind0, bool on, const Ray &r, KD &d,
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
    tauint<double, KD, true>(r, s.mua, s.mus, s.cy, s.cz, s.scat_radius, go && scat, d, dist, at_wall, ok, nis, &rare);
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
// WIDE: 53-bit draws (ORT-RNG-v2w, kernel variant bit 5) — the same kernel on the other stream
template <bool ANYSRC, bool WIDE = false>
__global__ __launch_bounds__(64 * kScatWaves) void scatter_front_kernel(TraceArgs a)
{
    using KD = KeyedDrawsT<WIDE ? 2 : 0>;
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
    auto store = [&](bool cond, int slot, const Ray &r, double t, const KD &d, int nis, int k) {
        if (cond) {
            P.f[0][slot] = r.pos.x; P.f[1][slot] = r.pos.y; P.f[2][slot] = r.pos.z;
            P.f[3][slot] = r.dir.x; P.f[4][slot] = r.dir.y; P.f[5][slot] = r.dir.z;
            P.f[6][slot] = t;
            P.c[slot] = d.c;
            P.m[slot] = ((uint32_t)nis << 8) | (uint32_t)k;
        }
    };
    auto load = [&](bool act, int slot, Ray &r, double &t, KD &d, int &nis, int &k) {
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
    auto route = [&](bool held, int slot, bool to_walk, bool to_arrived, const Ray &r, double t, const KD &d, int nis, int k) {
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
    auto hand_over = [&](bool cond, const Ray &r, double t, const KD &d, int nis) {
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
    auto defer = [&](bool cond, const KD &d) {
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
            KD d;
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
            tauint<double, KD, true>(r, s.mua, s.mus, s.cy, s.cz, s.scat_radius, walking, d, dist, at_wall, ok, nis, &rare);
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
            KD d;
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
            KD d;
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

}  // namespace ortk
