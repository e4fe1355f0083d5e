// ort_device.h — per-ray device functions of the MI355X trace path (gfx950, wave64).
//
// Numerics contract: everything that feeds the ray STATE is a separately rounded
// IEEE-754 fp64 operation in the order the reference writes it (this file is
// compiled with -ffp-contract=off; fp64 '/' and sqrt lower to correctly rounded
// sequences), so that for identical input rays and identical uniforms the result
// equals the reference's bit for bit.  Reference = lewisfish/OpticalRayTrace, all
// `real` fp64 (src/Makefile:2).  Each function cites the reference lines it implements.
//
// Code shape: PREDICATED DATAFLOW, not per-lane control flow.  A CU has ONE scalar
// unit for its four SIMDs, and every lane-divergent `if` costs scalar instructions
// (v_cmp -> s_and_saveexec -> s_cbranch_execz ... s_or exec).  Written with per-lane
// branches this path issued 1 scalar instruction per 3 vector instructions and the
// scalar unit, not the VALUs, set the pace (round-1 PMC: SQ_ACTIVE_INST_SCA 73 % of
// elapsed vs VALU 60 %).  So: every lane executes every surface step with a `live`
// predicate and commits state through selects; the only branches are WAVE-UNIFORM
// (surface kind via readfirstlane), plus the exec-masked side effects (image atomic,
// LDS queue traffic).  The filtered predicates below do not branch to their literal
// formulas either: they raise a per-lane flag and the caller deals with it per segment.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/ort.h"
#include "ort_fastd.h"
#include "ort_libm.h"


namespace ort {

// ----------------------------------------------------------------------------
// ORT-RNG-v2: counter-based uniforms keyed on (seed, phase, global ray, draw).
// Replaces the semantics of ran2() = one U[0,1) per call (src/random_mod.f90:39-46);
// the reference's own generator is the Fortran runtime's and is not reproducible
// across compilers or thread counts (SURVEY §7 "hard parts").
//   c = (ray << 24) + k                      linear draw counter of draw k of a ray (k < 2^24)
//   h = mix64(base + GOLDEN * ((c >> 1) + 1)),   base = mix64(seed ^ (GOLDEN * phase))
//   u = (k even ? h >> 32 : h & 0xffffffff) * 2^-32
// One SplitMix64 finaliser serves TWO consecutive draws (v1 spent one per draw: ~22 of the ~210
// vector instructions per surface).  In a surface program every live lane of a wave is at the
// same draw index, known at compile time, so the odd draws reuse the hash of the even ones.
// ----------------------------------------------------------------------------
constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;

__host__ __device__ inline uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

__host__ __device__ inline uint64_t stream_base(uint64_t seed, int phase)
{
    return mix64(seed ^ (kGolden * (uint64_t)phase));
}

// the hash input of pair j of a ray: zray + GOLDEN * (j + 1), zray = base + GOLDEN * (ray << 23)
// (= base + GOLDEN * (((ray << 24) + 2 j) >> 1)).  GOLDEN is odd, so the ray index comes back from
// zray through its inverse modulo 2^64.
constexpr uint64_t kGoldenInv = 0xF1DE83E19937733Dull;
static_assert(kGolden * kGoldenInv == 1ull, "kGoldenInv must invert kGolden modulo 2^64");
__host__ __device__ inline uint64_t zray_of(uint64_t base, uint64_t ray) { return base + kGolden * (ray << 23); }
__host__ __device__ inline uint64_t ray_of_zray(uint64_t zray, uint64_t base) { return ((zray - base) * kGoldenInv) >> 23; }

// a 32-bit draw as a double in [0,1) = w * 2^-32, exactly, without an int->fp64 conversion:
// 2^52 + w has w in the low mantissa word (exponent word 0x433..., ulp = 1)
__device__ inline double bits_to_unit(uint32_t w)
{
    return (__hiloint2double(0x43300000, (int)w) - 0x1p52) * 0x1p-32;
}
// fp32 path: the top 24 bits of the SAME draw, so a ray sees (to 2^-24) the same uniform
__device__ inline float bits_to_unit_f32(uint32_t w) { return (float)(w >> 8) * 0x1.0p-24f; }
template <class T> __device__ inline T unit_from(uint32_t w)
{
    if constexpr (sizeof(T) == 8) return T(bits_to_unit(w));
    else return bits_to_unit_f32(w);
}
__device__ inline uint32_t draw_word(uint64_t h, bool odd) { return odd ? (uint32_t)h : (uint32_t)(h >> 32); }

// ORT-RNG-v2w, the 53-bit stream (kernel variant bit 5; ran2() fills a real(8), src/random_mod.f90:39-46, and the
// runtime's random_number gives it 53 random bits): ONE hash per draw, all of a double's mantissa from it,
//   h = mix64(base + GOLDEN * (c + 1)),   u = (h >> 11) * 2^-53          (fp32 path: the top 24 bits of h)
// — every value k * 2^-53 of [0, 1) can occur, where v2's 32-bit draws stop at multiples of 2^-32.  The two halves
// of h >> 11 convert exactly and their sum is exact.  Offered by the kernels that count draws per lane (the
// lockstep kernel and the parity entry); the surface programs and the scattering pipeline are built on v2's pairs.
__host__ __device__ inline uint64_t wide_hash(uint64_t base, uint64_t c) { return mix64(base + kGolden * (c + 1ull)); }
__host__ __device__ inline double unit53(uint64_t h)
{
    const uint64_t x = h >> 11;
    return (double)(uint32_t)(x >> 32) * 0x1p-21 + (double)(uint32_t)x * 0x1p-53;
}
template <class T> __device__ inline T wide_unit_from(uint64_t h)
{
    if constexpr (sizeof(T) == 8) return T(unit53(h));
    else return (float)(uint32_t)(h >> 40) * 0x1.0p-24f;
}

// the i1 ballot builtin: an s_and of the compare mask with exec.  (HIP's __ballot(int) first
// materialises the predicate as 0/1 in a VGPR and compares it again: two VALU instructions.)
__device__ inline bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// "is any ray of the wavefront still alive" (st < 0) with the status word opaque at this point: the compiler
// otherwise re-derives the test from the selects that produced st (one compare + mask logic per select)
__device__ inline bool wave_any_live(int &st)
{
    asm("" : "+v"(st));
    return __builtin_amdgcn_ballot_w64(st < 0) != 0ull;
}

// Per-ray draw source of the parity / debug entry.  peek() is the next uniform, advance(c)
// consumes it where c.  Keyed stream, or an explicit table: draw k at table[k*stride].
struct Draws {
    uint64_t c;            // keyed: (ray << 24) + k
    uint64_t base;
    const double *table;
    int64_t stride;
    int len;
    int k;                 // draws consumed so far
    bool wide;             // keyed: ORT-RNG-v2w (wave-uniform)

    __device__ inline void init_keyed(uint64_t b, uint64_t ray, int first_draw, bool w = false)
    {
        table = nullptr; stride = 0; len = 0; k = first_draw; base = b; wide = w;
        c = (ray << 24) + (uint64_t)first_draw;
    }
    __device__ inline void init_table(const double *t, int64_t s, int l, int first_draw)
    {
        table = t; stride = s; len = l; k = first_draw; c = 0; base = 0; wide = false;
    }
    __device__ inline double peek() const
    {
        if (table) return k < len ? table[(int64_t)k * stride] : 0.5;
        if (wide) return unit53(wide_hash(base, c));
        return bits_to_unit(draw_word(mix64(base + kGolden * ((c >> 1) + 1ull)), (c & 1ull) != 0));
    }
    __device__ inline void advance(bool cnd)
    {
        k += cnd ? 1 : 0;
        c += cnd ? 1ull : 0ull;
    }
    __device__ inline double next() { double u = peek(); advance(true); return u; }
    __device__ inline void take(bool cnd, const Draws &o) { k = cnd ? o.k : k; c = cnd ? o.c : c; }   // same stream
    template <class T> __device__ inline T peek_as() const { return (T)peek(); }
    template <class T> __device__ inline T next_as() { return (T)next(); }
    template <class T> __device__ inline void next_pair_as(bool cnd, T &u1, T &u2)
    {
        u1 = peek_as<T>(); advance(cnd);
        u2 = peek_as<T>(); advance(cnd);
    }
};

// keyed-only variant of the bulk kernels that walk a list they do not know at compile time
// (generic walk, lockstep kernel): the linear counter per lane, one hash per draw
// WM: 0 = the stream is v2 (the queued kernels and the scattering pipeline by default: nothing is paid for a choice);
// 1 = chosen at run time (set_wide: v2, or the 53-bit v2w of kernel variant bit 5) — the lockstep kernel and emit_kernel;
// 2 = v2w, known at compile time (the WIDE instantiations of the queued kernels and of the scattering pipeline)
template <int WM> struct WideChoice {
    bool wide_ = false;
    __device__ inline bool wide() const { return wide_; }
    __device__ inline void set_wide(bool w) { wide_ = w; }
};
template <> struct WideChoice<0> {
    __device__ inline constexpr bool wide() const { return false; }
    __device__ inline void set_wide(bool) {}
};
template <> struct WideChoice<2> {
    __device__ inline constexpr bool wide() const { return true; }
    __device__ inline void set_wide(bool) {}
};
template <int WM> struct KeyedDrawsT : WideChoice<WM> {
    uint64_t c;            // (ray << 24) + k
    uint64_t base;
    __device__ inline void init_keyed(uint64_t b, uint64_t ray, int first_draw)
    {
        base = b;
        c = (ray << 24) + (uint64_t)first_draw;
    }
    __device__ inline uint32_t word() const
    {
        return draw_word(mix64(base + kGolden * ((c >> 1) + 1ull)), (c & 1ull) != 0);
    }
    __device__ inline double peek() const { return peek_as<double>(); }
    __device__ inline void advance(bool cnd) { c += cnd ? 1ull : 0ull; }
    __device__ inline void take(bool cnd, const KeyedDrawsT &o) { c = cnd ? o.c : c; }
    __device__ inline double next() { const double u = peek(); c += 1ull; return u; }
    template <class T> __device__ inline T peek_as() const
    {
        if constexpr (WM == 1) {
            // ONE hash and one conversion for both streams (the choice is wave-uniform, but a compiler that turns it into
            // selects would otherwise hash twice per draw): x = the draw as a 53-bit integer — v2w: h >> 11; v2: its 32-bit
            // word << 21, and w 2^21 2^-53 = w 2^-32 exactly as bits_to_unit forms it
            const bool wd = this->wide();
            const uint64_t h = mix64(base + kGolden * ((wd ? c : c >> 1) + 1ull));
            const uint64_t x = wd ? h >> 11 : (uint64_t)draw_word(h, (c & 1ull) != 0) << 21;
            if constexpr (sizeof(T) == 8) return T((double)(uint32_t)(x >> 32) * 0x1p-21 + (double)(uint32_t)x * 0x1p-53);
            else return (float)(uint32_t)(x >> 29) * 0x1.0p-24f;         // the top 24 bits of either
        } else if constexpr (WM == 2) return wide_unit_from<T>(wide_hash(base, c));
        else return unit_from<T>(word());
    }
    template <class T> __device__ inline T next_as()
    {
        const T u = peek_as<T>();
        c += 1ull;
        return u;
    }
    // two consecutive draws for the lanes `cnd` (a rejection loop's pair, src/random_mod.f90:68-69).  Draws 2j and
    // 2j + 1 are the two halves of ONE hash: when every lane of the wave stands at an even draw — it does in the
    // Box-Muller loops of the crs and isors sources, which start at draw 2 / 0 and consume pairs — one hash serves both
    // (v2w: one hash per draw)
    template <class T> __device__ inline void next_pair_as(bool cnd, T &u1, T &u2)
    {
        if (WM != 2 && !(WM == 1 && this->wide()) && !wave_any((c & 1ull) != 0)) {
            const uint64_t h = mix64(base + kGolden * ((c >> 1) + 1ull));
            u1 = unit_from<T>((uint32_t)(h >> 32));
            u2 = unit_from<T>((uint32_t)h);
            c += cnd ? 2ull : 0ull;
        } else {
            u1 = peek_as<T>(); advance(cnd);
            u2 = peek_as<T>(); advance(cnd);
        }
    }
    __device__ inline uint64_t ray() const { return c >> 24; }
    // queue image of the state (one 64-bit word)
    __device__ inline uint64_t pack() const { return c; }
    __device__ inline void unpack(uint64_t w, uint64_t b) { c = w; base = b; }
    __device__ inline uint64_t ray_of_packed(uint64_t w, uint64_t) const { return w >> 24; }
};
using KeyedDraws = KeyedDrawsT<0>;
using KeyedDrawsWide = KeyedDrawsT<2>;

// Draw source of the surface-program kernels: every live lane of a wave is at the same draw
// index K, a compile-time constant of the program step, so nothing is counted per lane: the
// state is the ray's zray, and at<K>() hashes pair K/2 at the even draw and keeps the hash for
// the odd one (FRESH: the hash is not at hand — first draw after the queue — and is formed again).
// WIDE: the 53-bit stream ORT-RNG-v2w (kernel variant bit 5) — one hash per draw, h = mix64(zray + GOLDEN (K + 1)) with
// zray = base + GOLDEN (ray << 24), i.e. wide_hash(base, (ray << 24) + K): the draw the lockstep kernel and the checker hand out
template <bool WIDE> struct ProgDrawsT {
    static constexpr int kRayShift = WIDE ? 24 : 23;
    uint64_t zray;
    uint64_t h;
    int k;                 // sequential interface of the emitters: folds to constants in their straight-line code
    static __host__ __device__ inline uint64_t zray_at(uint64_t base, uint64_t ray) { return base + kGolden * (ray << kRayShift); }
    __device__ inline void init_keyed(uint64_t base, uint64_t ray, int first_draw)
    {
        zray = zray_at(base, ray);
        h = 0;
        k = first_draw;
    }
    // ray `idx` of a launch whose first ray has zray z0 (= zray_at(base, first_ray), wave-uniform):
    // zray_at(base, first_ray + idx) = z0 + (GOLDEN << 23) idx modulo 2^64 — a 64 x 32-bit product
    // per lane instead of a 64-bit add, a shift and a 64 x 64-bit product
    __device__ inline void init_index(uint64_t z0, uint32_t idx, int first_draw)
    {
        zray = z0 + (kGolden << kRayShift) * (uint64_t)idx;
        h = 0;
        k = first_draw;
    }
    template <class T, int K, bool FRESH> __device__ inline T at()
    {
        if constexpr (WIDE) {
            h = mix64(zray + kGolden * (uint64_t)(K + 1));
            return wide_unit_from<T>(h);
        } else {
            if constexpr ((K & 1) == 0 || FRESH) h = mix64(zray + kGolden * (uint64_t)(K / 2 + 1));
            return unit_from<T>(draw_word(h, (K & 1) != 0));
        }
    }
    template <class T> __device__ inline T next_as()
    {
        T u;
        if constexpr (WIDE) {
            h = mix64(zray + kGolden * (uint64_t)(k + 1));
            u = wide_unit_from<T>(h);
        } else {
            if ((k & 1) == 0) h = mix64(zray + kGolden * (uint64_t)(k / 2 + 1));
            u = unit_from<T>(draw_word(h, (k & 1) != 0));
        }
        k += 1;
        return u;
    }
    __device__ inline uint64_t pack() const { return zray; }
    __device__ inline void unpack(uint64_t w, uint64_t) { zray = w; h = 0; k = 0; }
    __device__ inline uint64_t ray_of_packed(uint64_t w, uint64_t base) const { return ((w - base) * kGoldenInv) >> kRayShift; }
    // the dynamic interface is never instantiated for a program (surface_step takes the static one)
    template <class T> __device__ inline T peek_as() const { return T(0.5); }
    __device__ inline void advance(bool) {}
    template <class T> __device__ inline void next_pair_as(bool, T &u1, T &u2) { u1 = u2 = T(0.5); }
};
using ProgDraws = ProgDrawsT<false>;

// Division and square root of the traced arithmetic, by type.
//   double  the compiler's correctly rounded IEEE operations (the reference's arithmetic)
//   fastd   ort_fastd.h (its own operator/ and sqrt)
//   float   the fp32 path of BASELINE configs[4], which has no reference to be bit-exact against
//           (the reference is fp64 only, src/Makefile:2) and therefore none of the exact path's corset:
//           hardware reciprocal / square root / reciprocal square root as they come (~1 ulp fp32), fused
//           multiply-adds wherever a product feeds a sum (mad, dot3, ...: -ffp-contract=off keeps the compiler
//           from fusing the SHARED templates, so the float forms say it), the filtered decision forms without
//           their margins or deferrals (kLoose).  tests/test_gpu_fp32.py holds its deviation from fp64.
__device__ inline double div_t(double a, double b)
{
    return a / b;
}
__device__ inline fastd div_t(fastd a, fastd b) { return a / b; }
__device__ inline float div_t(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }     // v_rcp_f32: 1 ulp
__device__ inline double sqrt_t(double x)
{
    return sqrt(x);
}
__device__ inline fastd sqrt_t(fastd x) { return sqrt(x); }
__device__ inline float sqrt_t(float x) { return __builtin_amdgcn_sqrtf(x); }
#define ORT_SQRT(x) sqrt_t(x)
#define ORT_DIV(a, b) div_t(a, b)

// ORT_RARE(site, cond): a lane raises its `rare` flag.  -DORT_DBG_RARE (development) also counts
// the raises per site in ort_dbg_rare[].
#ifdef ORT_DBG_RARE
__device__ unsigned long long ort_dbg_rare[16];
#define ORT_RARE(site, cond) do { const bool c_ = (cond); rare = rare || c_; if (c_) atomicAdd(&ort_dbg_rare[site], 1ull); } while (0)
#else
// (bitwise, not `||`: a short-circuit makes the compiler branch around the compares of `cond` —
// two s_and_saveexec / s_cbranch_execz pairs per surface step and a boolean carried through a VGPR)
#define ORT_RARE(site, cond) do { rare = rare | (bool)(cond); } while (0)
#endif
// "does any lane need the literal formula": the guard of every rare path
__device__ inline bool wave_rare(bool p) { return wave_any(p); }

// sqrt(x) at the hot sites.  FILT (fp64): the compiler's own fp64 square-root expansion —
// v_rsq_f64 seed, one coupled Goldschmidt step, two residual corrections — WITHOUT its input
// scaling (only active below 2^-767) and its class fix-up (x = 0, inf): the same instructions
// on the same operands, hence bit-identical to sqrt(x), whenever 2^-700 < |x| < 2^700 (x < 0
// gives NaN either way).  A lane outside that range that needs the value raises `rare`
// (tests/csrc/check_exact_ops.hip compares the two over 2^28 operands of every exponent).
template <bool FILT, class T>
__device__ inline T sqrt_f(T x, bool need, bool &rare)
{
    if constexpr (FILT && std::is_same<T, double>::value) {
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y;
        double h = y * 0.5;
        const double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        double d = __builtin_fma(-g, g, x);
        g = __builtin_fma(d, h, g);
        d = __builtin_fma(-g, g, x);
        g = __builtin_fma(d, h, g);
        const bool plain = (fabs(x) > 0x1p-700) & (fabs(x) < 0x1p700);      // false for NaN too
        ORT_RARE(0, need & !plain);
        return g;
    }
    return ORT_SQRT(x);
}

// ----------------------------------------------------------------------------
// 3-vector algebra, src/vector_class.f90:48-186
// ----------------------------------------------------------------------------
// T = double is the reference's arithmetic (all `real` are fp64).  T = float is the fp32 study
// path of BASELINE configs[4]: same operations, single precision, literal predicates only.
template <class T> struct VecT { T x, y, z; };
using Vec = VecT<double>;

template <class T> __device__ inline VecT<T> vsub(VecT<T> a, VecT<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> __device__ inline VecT<T> vadd(VecT<T> a, VecT<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> __device__ inline VecT<T> vscale(VecT<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <class T> __device__ inline T vdot(VecT<T> a, VecT<T> b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
// kLoose<T>: an arithmetic without a bit-exact contract (fp32): products are fused into the sums they feed, and a
// lane inside a filtered predicate's margin is simply decided by the cheap form (no `rare`, no deferral).
template <class T> constexpr bool kLoose = std::is_same<T, float>::value;
// a * b + c, (a . b), b + a s, I alpha + N beta: separately rounded in the reference's order for the exact types, fused for fp32
template <class T> __device__ inline T mad(T a, T b, T c)
{
    if constexpr (kLoose<T>) return __builtin_fmaf(a, b, c);
    else return a * b + c;
}
template <class T> __device__ inline T dot3(T ax, T ay, T az, T bx, T by, T bz)
{
    if constexpr (kLoose<T>) return __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
    else return (ax * bx) + (ay * by) + (az * bz);
}
template <class T> __device__ inline VecT<T> vmad(VecT<T> a, T s, VecT<T> b)      // b + a s
{
    if constexpr (kLoose<T>) return {__builtin_fmaf(a.x, s, b.x), __builtin_fmaf(a.y, s, b.y), __builtin_fmaf(a.z, s, b.z)};
    else return vadd(b, vscale(a, s));
}
template <class T> __device__ inline VecT<T> vlin2(VecT<T> a, T s, VecT<T> b, T t)      // a s + b t
{
    if constexpr (kLoose<T>) return {__builtin_fmaf(b.x, t, a.x * s), __builtin_fmaf(b.y, t, a.y * s), __builtin_fmaf(b.z, t, a.z * s)};
    else return vadd(vscale(a, s), vscale(b, t));
}
template <> __device__ inline float vdot<float>(VecT<float> a, VecT<float> b) { return dot3(a.x, a.y, a.z, b.x, b.y, b.z); }
template <class T> __device__ inline VecT<T> vselect(bool c, VecT<T> a, VecT<T> b) { return {c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }

// x/t, y/t, z/t — three IEEE-754 divisions by the same denominator — with the
// reciprocal refinement done once.  This is the compiler's own fp64 division
// sequence (v_div_scale / v_rcp / 2 Newton steps / v_div_fmas / v_div_fixup)
// with the denominator-only half shared.  It applies when v_div_scale leaves the
// denominator unscaled for every numerator (d_i == t: always, outside the
// subnormal / overflow corners); then each quotient is bit-for-bit `x / t`.
// v_div_scale returns NaN when an operand is zero or NaN: exactly the cases in which
// v_div_fixup discards the iteration and forms the result from the operands' classes, so a
// NaN d_i (an exactly-zero numerator: the x component of every cylinder normal) does not
// break sharing — the test is the ORDERED `d_i <> t`.
__device__ inline Vec div3_shared(Vec v, double t, bool &shared)
{
    bool f0, f1, f2, fd;
    const double d0 = __builtin_amdgcn_div_scale(v.x, t, false, &fd);
    const double d1 = __builtin_amdgcn_div_scale(v.y, t, false, &fd);
    const double d2 = __builtin_amdgcn_div_scale(v.z, t, false, &fd);
    shared = !(d0 < t || d0 > t) && !(d1 < t || d1 > t) && !(d2 < t || d2 > t);
    double r = __builtin_amdgcn_rcp(t);
    double e = __builtin_fma(-t, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-t, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double n0 = __builtin_amdgcn_div_scale(v.x, t, true, &f0);
    const double n1 = __builtin_amdgcn_div_scale(v.y, t, true, &f1);
    const double n2 = __builtin_amdgcn_div_scale(v.z, t, true, &f2);
    const double m0 = n0 * r, m1 = n1 * r, m2 = n2 * r;
    Vec q;
    q.x = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(__builtin_fma(-t, m0, n0), r, m0, f0), t, v.x);
    q.y = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(__builtin_fma(-t, m1, n1), r, m1, f1), t, v.y);
    q.z = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(__builtin_fma(-t, m2, n2), r, m2, f2), t, v.z);
    return q;
}

// the three quotients, self-contained (emitters): lanes where the shared form does not apply
// divide plainly, behind a wave-uniform branch
__device__ inline VecT<float> div3(VecT<float> v, float t)
{
    const float r = __builtin_amdgcn_rcpf(t);
    return {v.x * r, v.y * r, v.z * r};
}
__device__ inline VecT<fastd> div3(VecT<fastd> v, fastd t)
{
    const double r = rcp_nr2(t.v);
    return {fastd(v.x.v * r), fastd(v.y.v * r), fastd(v.z.v * r)};
}
__device__ inline Vec div3(Vec v, double t)
{
    bool shared;
    Vec q = div3_shared(v, t, shared);
    if (wave_any(!shared)) {
        q.x = shared ? q.x : v.x / t;
        q.y = shared ? q.y : v.y / t;
        q.z = shared ? q.z : v.z / t;
    }
    return q;
}

// magnitude_fn (:175-186): NORMALISES, by three divisions
template <class T> __device__ inline VecT<T> vnormalise(VecT<T> a)
{
    T tmp = ORT_SQRT(a.x * a.x + a.y * a.y + a.z * a.z);
    return div3(a, tmp);
}
template <> __device__ inline VecT<float> vnormalise<float>(VecT<float> a)
{
    const float y = __builtin_amdgcn_rsqf(dot3(a.x, a.y, a.z, a.x, a.y, a.z));       // v_rsq_f32: multiply by 1/|a|
    return {a.x * y, a.y * y, a.z * y};
}
template <> __device__ inline VecT<fastd> vnormalise<fastd>(VecT<fastd> a)
{
    const double y = rsq_nr2((a.x * a.x + a.y * a.y + a.z * a.z).v);      // multiply by 1/|a| directly
    return {fastd(a.x.v * y), fastd(a.y.v * y), fastd(a.z.v * y)};
}

// Normalisation at the hot sites.  FILT (fp64): sqrt_f, then the three quotients v_i / t from
// the compiler's own division sequence with everything that only serves operands needing
// v_div_scale's rescaling, or v_div_fixup's special cases, left out: one refined reciprocal of
// t, then per component  m = v r,  e = fma(-t, m, v),  q = fma(e, r, m)  — exactly what
// v_div_scale / v_div_fmas / v_div_fixup reduce to when nothing is scaled and nothing is special.
// That holds when t = |v| lies in (2^-350, 2^350) (sqrt_f's range test) and every component is
// above 2^-300 in magnitude or exactly zero (gives +0 like 0/t); |v_i| <= t bounds the quotients
// by 1.  A lane outside that raises `rare`.
// tests/csrc/check_exact_ops.hip compares with the IEEE divisions over 2^28 operand sets.
// the three quotients v_i / t of that scheme alone: requires t in (2^-350, 2^350) (the caller's
// sqrt_f guarantees it for t = sqrt(s)) and |v_i| <= t up to rounding (components of the vector
// whose length t is)
// x_zero: the caller knows a.x is the literal zero (the x of a cylinder normal; wave-uniform).
// Any other component that is not above 2^-300 — an exact zero included: a ray lying exactly in a
// coordinate plane through the centre — raises `rare` (three compares instead of six).
template <bool FILT, class T>
__device__ inline VecT<T> div3_f(VecT<T> a, T t, bool need, bool &rare, bool x_zero = false)
{
    if constexpr (FILT && std::is_same<T, double>::value) {
        double r = __builtin_amdgcn_rcp(t);
        double e = __builtin_fma(-t, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-t, r, 1.0);
        r = __builtin_fma(r, e, r);
        const double mx = a.x * r, my = a.y * r, mz = a.z * r;
        Vec q;
        q.x = __builtin_fma(__builtin_fma(-t, mx, a.x), r, mx);
        q.y = __builtin_fma(__builtin_fma(-t, my, a.y), r, my);
        q.z = __builtin_fma(__builtin_fma(-t, mz, a.z), r, mz);
        const bool odd = (!x_zero & !(fabs(a.x) > 0x1p-300)) | !(fabs(a.y) > 0x1p-300) | !(fabs(a.z) > 0x1p-300);
        ORT_RARE(1, need & odd);
        return q;
    }
    return div3(a, t);
}

template <bool FILT, class T>
__device__ inline VecT<T> vnormalise_f(VecT<T> a, bool need, bool &rare, bool x_zero = false)
{
    if constexpr (FILT && std::is_same<T, double>::value) {
        // sqrt_f's range: div3_f's guard puts |a.y|, |a.z| above 2^-300, so s > 2^-600; the
        // upper end is tested here
        const double s = a.x * a.x + a.y * a.y + a.z * a.z;
        bool unused = false;
        const double t = sqrt_f<true, double>(s, false, unused);
        ORT_RARE(0, need & !(s < 0x1p700));
        return div3_f<true, double>(a, t, need, rare, x_zero);
    }
    return vnormalise(a);
}

// Normalisation of a vector whose length is KNOWN up to rounding: the normal of a sphere or a
// circular cylinder at a point of the surface (|N| = radius), a direction that was already divided
// by its length (|v| = 1).  With t0 the known length, h0 = 1/(2 t0), k0 = h0/(2 t0^2):
//   e  = s - t0^2                  (one fma: exact up to 2^-53 |e|)
//   g1 = t0 + e h0                 sqrt(s) to |x|^2/8 + one rounding,  x = e/t0^2, |x| < 2^-32
//   h1 = h0 - e k0                 1/(2 sqrt s) to 2^-52
//   t  = g1 + (s - g1^2) h1        the residual correction that ends the compiler's own square-root
//                                  expansion: the correctly rounded sqrt(s), i.e. the bits of sqrt(s)
//   r  = 2 h1 refined by one Newton step on t: 1/t to within an ulp, as after the compiler's two
//        steps on the v_rcp seed; then the three quotients exactly as div3_f
// No v_rsq, no v_rcp, 8 instructions instead of 15 (+ 2 quarter-rate seeds) in front of the
// quotients.  A lane whose s is not within 2^-32 of t0^2 (it is not on the surface: only garbage
// lanes and rays the quadratic flagged already) raises `rare`.
// tests/csrc/check_exact_ops.hip compares with v / sqrt(v.v) over 2^28 operand sets per mode.
template <bool FILT, class T>
__device__ inline VecT<T> vnormalise_est(VecT<T> a, T t0, T h0, T k0, T s_tol, bool need, bool &rare, bool x_zero = false)
{
    if constexpr (FILT && std::is_same<T, double>::value) {
        const double s = a.x * a.x + a.y * a.y + a.z * a.z;
        const double e = __builtin_fma(-t0, t0, s);
        // t0, h0, k0 are wave-uniform (SGPR pairs) and a vector instruction reads at most ONE scalar operand: left to
        // itself the compiler copies two of them into VGPRs (two v_mov_b64 per call).  h0 is the one both need.
        double vh0, g1, h1;
        if (__builtin_constant_p(h0)) {                      // literal arguments (emit_ring: 1, 0.5, 0.25): inline constants
            g1 = __builtin_fma(e, h0, t0);
            h1 = __builtin_fma(-e, k0, h0);
        } else {
            asm("v_mov_b64 %0, %1" : "=v"(vh0) : "s"(h0));
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(g1) : "v"(e), "v"(vh0), "s"(t0));
            asm("v_fma_f64 %0, -%1, %2, %3" : "=v"(h1) : "v"(e), "s"(k0), "v"(vh0));
        }
        const double d = __builtin_fma(-g1, g1, s);
        const double t = __builtin_fma(d, h1, g1);
        const double y = h1 + h1;
        const double e2 = __builtin_fma(-t, y, 1.0);
        const double r = __builtin_fma(y, e2, y);
        const double mx = a.x * r, my = a.y * r, mz = a.z * r;
        Vec q;
        q.x = __builtin_fma(__builtin_fma(-t, mx, a.x), r, mx);
        q.y = __builtin_fma(__builtin_fma(-t, my, a.y), r, my);
        q.z = __builtin_fma(__builtin_fma(-t, mz, a.z), r, mz);
        // the x of a cylinder normal (a literal +0, known at compile time in a program kernel): 0 r = +0, fma(-t, +0, +0) =
        // +0, fma(+0, r, +0) = +0 for every finite r > 0 — and a lane whose r is anything else fails the test on e below
        if (__builtin_constant_p(x_zero) && x_zero) q.x = 0.0;
        const bool odd = (!x_zero & !(fabs(a.x) > 0x1p-300)) | !(fabs(a.y) > 0x1p-300) | !(fabs(a.z) > 0x1p-300);
        ORT_RARE(1, need & (odd | !(fabs(e) < s_tol)));
        return q;
    }
    return vnormalise(a);
}

// x / y for operands that need neither v_div_scale's rescaling nor v_div_fixup's special cases
// (the caller's guard keeps |x|, |y| and |x / y| between 2^-700 and 2^700, y away from the
// subnormals): the compiler's division sequence with those two stages left out — bit for bit x / y.
__device__ inline double div_plain(double x, double y)
{
    double r = __builtin_amdgcn_rcp(y);
    double e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double m = x * r;
    return __builtin_fma(__builtin_fma(-y, m, x), r, m);
}

template <class T> struct RayT { VecT<T> pos, dir; };
using Ray = RayT<double>;

// What a surface step may ASSUME (OPT bit mask; 0 = nothing, the generic walk and every entry that takes
// rays from the caller).  Each bit removes instructions whose result is known in advance; none changes a
// result (tests: every kernel variant against the lockstep kernel and the CPU checker, bit for bit).
//   OPT_UNIT_DIR   the direction is a unit vector up to rounding: it was emitted by `point` / `ring`
//                  (normalised there) and every surface passed since left I alpha + N beta with
//                  |I| = |N| = 1 and alpha^2 (1 - c1^2) + ... = 1 (refract: eta^2 (1 - c1^2) + c2^2 = 1 by
//                  c2^2 = k; reflect: 1 - 4 c1^2 + 4 c1^2), each to a few ulps — true for the fused
//                  program kernels, whose rays never come from outside.  Then a = dir.dir needs no range
//                  guard in solve_and_pick, and the NA test of make_image may take dir.z for dir.z / |dir|.
//   OPT_AXIAL_START  the ray stands where `point` put it, (0, 0, 0 + offset), and this is the first surface of its loop
//                  (a cylinder: the bottle): pos - centre and c = L.L - radius**2 are the system's (axial_start), not the ray's
//   OPT_ON_AXIS    the surface's centre has cx = cy = +0.0 exactly (host: match_program): pos.x - cx and
//                  pos.y - cy are pos.x and pos.y, bit for bit (x - (+0) = x for every x, -0 included).
constexpr int OPT_UNIT_DIR = 1, OPT_ON_AXIS = 2, OPT_AXIAL_START = 4;

// ----------------------------------------------------------------------------
// Filtered predicates (FILT = true, the production setting).
//
// What only feeds a DECISION — reflect or refract (u <= R), inside the aperture,
// which root of the quadratic, NA acceptance, which image bin — is evaluated with a
// cheap approximation whose error bound is orders of magnitude smaller than the
// margin it is tested against.  A lane that lands inside the margin (probability
// ~1e-10 per test), and every special case (costt >= 1 at normal incidence, k within
// 1e-6 of total reflection, tangent rays, zero or non-finite operands), raises the
// per-lane flag `rare` and carries on with an unspecified value.  The CALLER sees to it
// that such a ray is traced with FILT = false (the reference's literal formulas) instead:
// the queued kernel appends its index to a list that the literal lockstep kernel traces
// afterwards (ort_hip.hip, `defer`); the lockstep kernel re-runs the segment in place
// (`walk`).  The decision taken is therefore always the reference's and outcomes stay
// bit-identical, while the hot path holds no rare branch at all (one behind every
// predicate cost 14 % of the kernel time: each is a scheduling barrier plus scalar
// work, and two of them — total internal reflection at the rim of the plano-convex lens,
// the exactly-zero x component of every cylinder normal — were not rare) and ~4 of the
// ~13 fp64 divide / square-root expansions per surface leave it.  FILT = false evaluates
// every predicate literally (re-run path, A/B, tests).
// ----------------------------------------------------------------------------
// 1/y with relative error < 2^-40 for finite normal y: v_rcp_f64 seed + one Newton step
__device__ inline double rcp_approx(double y)
{
    double r = __builtin_amdgcn_rcp(y);
    double e = __builtin_fma(-y, r, 1.0);
    return __builtin_fma(e, r, r);
}

// 1/sqrt(s), relative error < 2^-40: v_rsq_f64 seed + one Newton step
__device__ inline double rsq_approx(double s)
{
    double y = __builtin_amdgcn_rsq(s);
    double h = 0.5 * s * y;
    double e = __builtin_fma(-h, y, 0.5);
    return __builtin_fma(y, e, y);
}
__device__ inline fastd rcp_approx(fastd y) { return fastd(rcp_approx(y.v)); }
__device__ inline float rcp_approx(float y) { return __builtin_amdgcn_rcpf(y); }
__device__ inline float rsq_approx(float s) { return __builtin_amdgcn_rsqf(s); }
// a b + c in one rounding, for decision-only arithmetic (the traced state never uses it)
__device__ inline double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ inline fastd fmad(fastd a, fastd b, fastd c) { return fastd(__builtin_fma(a.v, b.v, c.v)); }
__device__ inline float fmad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ inline fastd rsq_approx(fastd s) { return fastd(rsq_approx(s.v)); }

// keep ? x : -x.  For fp64 the compiler flips the sign with a v_xor and then selects (two instructions); the
// select instruction itself can negate a source (the `neg` modifier works on bit 31 of a 32-bit source: the high
// dword of a double): one v_cndmask.  Pure bit manipulation: the same value as the plain expression, NaNs included.
__device__ inline double neg_unless(double x, bool keep)
{
    const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
    const int hi = __double2hiint(x);
    int out;
    asm("v_cndmask_b32_e64 %0, -%1, %1, %2" : "=v"(out) : "v"(hi), "s"(m));
    return __hiloint2double(out, __double2loint(x));
}
// flip ? -x : x
__device__ inline double neg_if(double x, bool flip)
{
    const unsigned long long m = __builtin_amdgcn_ballot_w64(flip);
    const int hi = __double2hiint(x);
    int out;
    asm("v_cndmask_b32_e64 %0, %1, -%1, %2" : "=v"(out) : "v"(hi), "s"(m));
    return __hiloint2double(out, __double2loint(x));
}
__device__ inline float neg_if(float x, bool flip) { return flip ? -x : x; }
__device__ inline fastd neg_if(fastd x, bool flip) { return fastd(neg_if(x.v, flip)); }
__device__ inline float neg_unless(float x, bool keep) { return keep ? x : -x; }
__device__ inline fastd neg_unless(fastd x, bool keep) { return fastd(neg_unless(x.v, keep)); }

// The staged system as the kernels see it.  For T = double this is ort_system itself (the
// device copy is staged byte for byte); for T = float a converted copy is staged.
template <class T> struct SurfaceT {
    T cx, cy, cz, radius, radius_b, n1, n2, eta, aperture, mua, mus, hgg, scat_radius;
    int32_t kind;
    uint32_t flags;
};
template <class T> struct SystemT {
    int32_t n_surfaces[2], split[2], ring_ellipse, pad;
    SurfaceT<T> surfaces[2][ORT_MAX_SURFACES];
    T cos_theta_max, ring_r1, ring_r2, ring_lens_r2, ring_lens_z, ring_bottle_ra, ring_bottle_rb, ring_bottle_z;
    T bin_width, inv_bin_width, na_cos_min, twopi;
    T spot_dphi, spot_dtheta, crs_sigma, crs_radius, crs_cy, crs_cz, img_lens_r2, img_lens_z;
    T point_offset;
    T isors_sigma, isors_k, isors_height, isors_base_pos, isors_z, isors_rad1, isors_rad2, isors_cy, isors_cz, isors_lens_r2, isors_lens_z;
    int32_t emitter[2];
};
template <class T> struct SysTypes { using Sys = SystemT<T>; using Surf = SurfaceT<T>; };
template <> struct SysTypes<double> { using Sys = ort_system; using Surf = ort_surface; };
template <> struct SysTypes<fastd> { using Sys = ort_system; using Surf = ort_surface; };   // doubles convert implicitly

// Per-surface values every ray of a workgroup would otherwise recompute (wave-uniform operands,
// but fp64 arithmetic lives in the vector unit): formed once per workgroup next to the staged
// system, by the very operations the per-ray code would use, so nothing changes bit-wise.  Only
// the filtered path reads them; the literal path recomputes from the surface record.
template <class T> struct SurfAuxT { T r2, ap2, ap_tol, eta2, ell_sa, ell_sb, rh, rk, r2_tol, ap_lo, ap_hi, ax_ly, ax_lz, ax_c; };
template <class T, class Surf>
__host__ __device__ inline SurfAuxT<T> make_aux(const Surf &s)
{
    SurfAuxT<T> a;
    const T r = s.radius, rb = s.radius_b, A = s.aperture, e = s.eta;
    a.r2 = r * r;                          // intersect_sphere / _cylinder: radius**2
    a.ap2 = A * A;                         // aperture test on the squares
    a.ap_tol = T(1e-12) * a.ap2;
    a.ap_lo = a.ap2 - a.ap_tol;            // outside_aperture: decided when s2 < ap_lo or s2 > ap_hi
    a.ap_hi = a.ap2 + a.ap_tol;
    a.eta2 = e * e;                        // refract: eta**2
    a.ell_sa = T(1.) / (r * r);            // intersect_ellipse: 1/semia**2, 1/semib**2 (inf for other kinds: unused)
    a.ell_sb = T(1.) / (rb * rb);
    // vnormalise_est: the normal of a sphere / cylinder has length radius up to rounding, so its
    // square root and reciprocal start from these estimates instead of v_rsq / v_rcp seeds
    a.rh = T(0.5) / r;                     // 1 / (2 |N|)
    a.rk = a.rh / (T(2.) * a.r2);          // d(1/(2 sqrt s))/ds at s = r^2, negated
    a.r2_tol = T(0x1p-32) * a.r2;          // |N.N - r^2| beyond this: not a point of the surface, literal path
    a.ax_ly = a.ax_lz = a.ax_c = T(0.);    // OPT_AXIAL_START: set by axial_start() for the first surface of the point loop
    return a;
}

// OPT_AXIAL_START: the point source starts every ray at pos = (0, 0, 0 + offset) (emit_point), so at the FIRST
// surface of its loop L = pos - centre and c = L.L - radius**2 are the same for every ray: formed here once per
// system, operation for operation as intersect_quadric forms them per ray (cylinder: no x terms).
template <class T, class Surf>
__host__ __device__ inline void axial_start(SurfAuxT<T> &a, const Surf &s, T point_offset)
{
    const T py = T(0.0), pz = T(0.0) + point_offset;
    const T Lx = T(0.0), Ly = py - s.cy, Lz = pz - s.cz;
    a.ax_ly = Ly;
    a.ax_lz = Lz;
    a.ax_c = ((Lx * Lx) + (Ly * Ly) + (Lz * Lz)) - a.r2;
}

template <class T> __device__ inline bool aperture_present(T a);
template <> __device__ inline bool aperture_present<double>(double a)
{
    return __builtin_amdgcn_readfirstlane(__double2hiint(a)) >= 0;
}
template <> __device__ inline bool aperture_present<fastd>(fastd a)
{
    return __builtin_amdgcn_readfirstlane(__double2hiint(a.v)) >= 0;
}
template <> __device__ inline bool aperture_present<float>(float a)
{
    return __builtin_amdgcn_readfirstlane(__float_as_int(a)) >= 0;
}

// ----------------------------------------------------------------------------
// solveQuadratic (src/surfaces.f90:227-260) + root choice (:75-86), predicated.  The callers
// pass hb = b/2 (the reference forms b = 2.*(...)).
// FILT works on the quantities scaled by exact powers of two — D = hb^2 - a c = discrim/4,
// sqrt(D) = sqrt(discrim)/2, q = -(hb +- sqrt(D)) — which round exactly like the reference's
// (scaling by 2 or 4 commutes with rounding away from the subnormal range, and D is kept inside
// (1e-200, 1e200)), so q, and the quotient returned, are the reference's bit for bit with three
// multiplications fewer.  Away from tangency the order of the two roots follows from signs alone
// (q^2 - a c = |hb| sqrt(D) + D >= 2.5e-11 q^2 under the filter, a > 0, and rounding is
// monotone), so only the quotient the reference ends up returning is divided out, with its
// operands chosen by selects:
//   q > 0 : roots c/q <= q/a ; c/q < 0 <=> c < 0  -> t = c < 0 ? q/a : c/q   (always a hit)
//   q < 0 : q/a < 0 ; hit <=> c/q >= 0 <=> c <= 0 -> t = c/q
// Range guards: on the two operands of the quotient, see `ok`.
// ----------------------------------------------------------------------------
template <bool FILT, class T, bool UNIT = false>
__device__ inline void solve_and_pick(T a, T hb, T c, bool live, T &t, bool &hit, bool &rare)
{
    // (fp32, kLoose: the filtered FORM — one quotient, the root picked by signs — without its margins: `rare` is never
    // consulted for that type)
    if constexpr (FILT) {
        const T hh = hb * hb;
        const T D = kLoose<T> ? mad(-a, c, hh) : hh - a * c;
        const bool nneg = !(D < T(0.0));              // :243 — D < 0: no real root (one compare serves `ok` and `hit`)
        const bool neg = !nneg;
        bool unused = false;
        const T sq = sqrt_f<true, T>(D, false, unused);   // range: see `ok`; NaN when neg (misses)
        const bool qpos = !(hb > T(0.0));             // q = -(hb + s) < 0 for hb > 0, = s - hb > 0 otherwise (s > 0)
        const T q = (-hb) - neg_if(sq, qpos);         // :249-253: -(hb + s), as (-hb) - s (negation commutes with rounding): no sign flip afterwards
        const bool cneg = c < T(0.0);
        const bool use_qa = qpos && cneg;
        const T num = use_qa ? q : c, den = use_qa ? a : q;
        // The one quotient is formed without v_div_scale's rescaling and v_div_fixup's special cases
        // (div_plain), which is x / y bit for bit while the operands stay well inside the normal range.
        // Tested for every lane: |D| > 1e-10 hb^2 (not tangent), |D| < 2^900 (hb^2 and a c far from
        // overflow: D and the reference's b^2 - 4 a c then have the same sign — the scaling by 4 is
        // exact), a in (2^-100, 2^100), |c| > 2^-300 (a c is a normal number; c = 0 — a ray that
        // starts exactly on the surface — and a quotient c/q that would underflow to a signed zero,
        // which the reference's root ordering treats as a root at distance 0, go to the literal
        // path).  For a lane with real roots also |q| = |hb| + sqrt(D) in (2^-300, 2^300): sqrt(D) <=
        // |q| and D > 2^-640 are inside sqrt_f's range, |c| < 2^701, every quotient in (2^-1000, 2^1000).
        // (tools/check_quadratic.py replays these guards on the CPU over all exponents.)
        // UNIT: a = dir.dir = 1 to a few ulps (OPT_UNIT_DIR): its two range tests are vacuous
        const bool a_ok = UNIT ? true : ((a > T(0x1p-100)) & (a < T(0x1p100)));
        const bool common = (fabs(D) > T(1e-10) * hh) & (fabs(D) < T(0x1p900)) & a_ok & (fabs(c) > T(0x1p-300));
        const bool ok = common & (neg | ((fabs(q) > T(0x1p-300)) & (fabs(q) < T(0x1p300))));
        if constexpr (std::is_same<T, double>::value) t = div_plain(num, den);
        else t = ORT_DIV(num, den);
        hit = (qpos || cneg) && nneg;
        ORT_RARE(2, live & !ok);                     // tangent, degenerate, on the surface, out of range, or NaN
    } else {
        const T b = T(2.0) * hb;
        const T discrim = b * b - T(4.0) * a * c;
        const bool neg = discrim < T(0.0);            // :243
        const T sq = ORT_SQRT(discrim);
        const T q = (b > T(0.0)) ? T(-0.5) * (b + sq) : T(-0.5) * (b - sq);
        const bool dz = discrim == T(0.0);            // :245-247
        const T xd = ORT_DIV(T(-0.5) * b, a);
        const T t0 = dz ? xd : ORT_DIV(q, a);
        const T t1 = dz ? xd : ORT_DIV(c, q);
        const bool sw = t0 > t1;                      // :75-79
        const T lo = sw ? t1 : t0, hi = sw ? t0 : t1;
        const bool lneg = lo < T(0.0);                // :80-83
        t = lneg ? hi : lo;
        hit = !(lneg && hi < T(0.0)) && !neg;
    }
}

// intersect_sphere (src/surfaces.f90:52-89) and intersect_cylinder (:91-130) in
// one body: the x-axis cylinder is the sphere with the x terms removed
// (a = dz^2+dy^2 etc. — fp addition commutes, so the sums are bit-identical).
// r2 = radius**2 (SurfAuxT); the literal path forms it itself.
template <bool FILT, class T, int OPT = 0>
__device__ inline void intersect_quadric(const RayT<T> &r, T cx, T cy, T cz, T radius, T r2,
                                         bool cylinder, bool live, T &t, bool &hit, bool &rare,
                                         T ax_ly = T(0.), T ax_lz = T(0.), T ax_c = T(0.))
{
    constexpr bool axis = (OPT & 2) != 0;                // OPT_ON_AXIS: cx = cy = +0.0
    constexpr bool start = (OPT & 4) != 0;               // OPT_AXIAL_START: L and c are the system's (axial_start), a cylinder
    static_assert(!start || FILT, "the axial start serves the filtered program kernels");
    T Lx = cylinder ? T(0.0) : (axis ? r.pos.x : r.pos.x - cx);
    T Ly = start ? ax_ly : (axis ? r.pos.y : r.pos.y - cy);
    T Lz = start ? ax_lz : r.pos.z - cz;
    T dx = cylinder ? T(0.0) : r.dir.x;
    T a = dot3(dx, r.dir.y, r.dir.z, dx, r.dir.y, r.dir.z);
    T hb = dot3(dx, r.dir.y, r.dir.z, Lx, Ly, Lz);
    T c = start ? ax_c : dot3(Lx, Ly, Lz, Lx, Ly, Lz) - (FILT ? r2 : radius * radius);
    solve_and_pick<FILT, T, (OPT & 1) != 0>(a, hb, c, live, t, hit, rare);
}

// intersect_ellipse, src/surfaces.f90:133-176 (sa, sb = 1/semia**2, 1/semib**2: SurfAuxT)
template <bool FILT, class T, int OPT = 0>
__device__ inline void intersect_ellipse(const RayT<T> &r, T cy, T cz, T semia, T semib, T aux_sa, T aux_sb,
                                         bool live, T &t, bool &hit, bool &rare)
{
    T sa = FILT ? aux_sa : ORT_DIV(T(1.), semia * semia);
    T sb = FILT ? aux_sb : ORT_DIV(T(1.), semib * semib);
    T Ly = r.pos.y - cy;
    T Lz = r.pos.z - cz;
    T a = sa * (r.dir.z * r.dir.z) + sb * (r.dir.y * r.dir.y);
    T hb = sa * r.dir.z * Lz + sb * r.dir.y * Ly;
    T c = sa * (Lz * Lz) + sb * (Ly * Ly) - T(1);
    solve_and_pick<FILT>(a, hb, c, live, t, hit, rare);      // a = sa dz^2 + sb dy^2 is not |dir|^2: guards stay
}

// fresnel, src/surfaces.f90:336-372, as one expression (eta = n1/n2 rounded once on
// the host).  costt > 1 (rounding at normal incidence) makes sintt NaN, every
// comparison false, tir NaN -> 1: the reference then ALWAYS reflects; kept.
template <class T>
__device__ inline T fresnel(T costt, T n1, T n2, T eta)
{
    T sintt = ORT_SQRT(T(1.) - costt * costt);
    T sint2 = eta * sintt;
    T cost2 = ORT_SQRT(T(1.) - sint2 * sint2);
    T r1 = fabs(ORT_DIV(n1 * costt - n2 * cost2, n1 * costt + n2 * cost2));
    T r2 = fabs(ORT_DIV(n1 * cost2 - n2 * costt, n1 * cost2 + n2 * costt));
    T tir = T(0.5) * (r1 * r1 + r2 * r2);
    tir = (tir != tir || tir > T(1.) || tir < T(0.)) ? T(1.) : tir;     // :366-369
    return (sint2 > T(1.)) ? T(1.0) : ((costt == T(1.)) ? T(0.) : tir);  // :353-358
}

// reflect_refract (src/surfaces.f90:262-282) with reflect (:285-300) and refract
// (:303-333), predicated: the direction is committed where `live`.  The caller
// supplies the uniform u and consumes the draw.  Returns true where the ray reflected.
// FILT: R is only ever compared with u, so it is formed with two approximate
// reciprocals, with refract's own c2 standing in for fresnel's cost2 (the same
// quantity, rounded along another path; they differ by < 1e-13 once k > 1e-6) and with
// refract's eta c1 - c2 as the first numerator.  |R' - R| < 1e-12, the margin is 1e-10.
// DIES: a reflected ray ends at this surface (ORT_F_SKIP_ON_REFLECT) and the caller reads nothing of an
// ended ray's state (KEEP = false): only the refracted direction is formed — alpha = eta as it stands, beta
// one sign select — and a reflecting lane leaves with garbage in I.
template <bool FILT, bool KEEP, class T, bool DIES = false>
__device__ inline bool reflect_refract(VecT<T> &I, VecT<T> N, T n1, T n2, T eta, T eta2, T u, bool live, bool &rare)
{
    static_assert(!DIES || !KEEP, "DIES leaves garbage in reflecting lanes");
    const T c1s = vdot(N, I);                            // == vdot(I, N): the products commute
    const T c1 = fabs(c1s);                              // costt (fresnel) and |c1| (refract)
    // refract's radicand, refract's order (:327); eta2 = eta**2 (SurfAuxT)
    const T k = kLoose<T> ? mad(-(FILT ? eta2 : eta * eta), mad(-c1, c1, T(1.0)), T(1.0))
                          : T(1.0) - (FILT ? eta2 : eta * eta) * (T(1.0) - c1 * c1);
    bool unused = false;
    const T c2 = sqrt_f<FILT, T>(k, false, unused);      // NaN beyond total reflection: unused there
    const T ec1 = eta * c1;
    const T m = ec1 - c2;                                // refract's coefficient of the normal (:329)
    bool reflected;
    if constexpr (FILT) {
        // fresnel's two amplitude ratios with numerator and denominator divided by n2:
        // (n1 c1 - n2 c2)/(n1 c1 + n2 c2) = (eta c1 - c2)/(eta c1 + c2), likewise the other
        // and u < R is tested on the cross-multiplied form (every factor below is positive)
        //   2 u B^2 E^2 < A^2 E^2 + C^2 B^2,   A = eta c1 - c2, B = eta c1 + c2, C = eta c2 - c1, E = eta c2 + c1
        // — ten multiply-adds, no reciprocal; relative error of either side < 1e-15.
        const T ec2 = eta * c2;
        const T B = ec1 + c2, E = ec2 + c1, Cn = ec2 - c1;
        const T B2 = B * B, E2 = E * E, A2 = m * m, C2 = Cn * Cn;
        const T P = B2 * E2;
        const T rhs = fmad(A2, E2, C2 * B2);                    // = 2 R P
        const T diff = fmad(T(-0.5), rhs, u * P);               // = (u - R) P
        // Total internal reflection is an everyday outcome at the rim of the plano-convex lens
        // (6 % of the ring rays): k < -1e-6 means eta^2 (1 - c1^2) > 1 + 1e-6, so fresnel's
        // `sint2 > 1` (:353) holds whatever its rounding and it returns 1: u <= 1 reflects.
        // (diff is NaN there and is not consulted.)  NaN anywhere else -> undecided; c1 >= 1 and
        // k within 1e-6 of zero -> literal path.
        // The same decisions in four compares instead of five.  With c1 < 1 and |k| > 1e-6: k > 0 makes c2,
        // P, rhs and diff finite numbers; k < 0 makes c2 = sqrt(k) NaN and with it diff.  So "diff is NaN"
        // IS total reflection there, and the two tests on diff are written so that a NaN takes the
        // reflecting side: decided unless |diff| <= margin (unordered: decided), reflected unless diff >= 0.
        const bool decided = (c1 < T(1.0)) & (fabs(k) > T(1e-6)) & !(fabs(diff) <= T(1e-10) * P);
        reflected = !(diff >= T(0.0));
        ORT_RARE(3, live & !decided);
    } else {
        reflected = u <= fresnel(c1, n1, n2, eta);       // :275
    }
    // Both outcomes are I*alpha + N*beta, bit for bit:
    //   reflect (:297)      I - (2 c1s) N         = I*1   + N*(-(2 c1s))     (x*1 and a + (-b) are exact)
    //   refract (:320-329)  eta I + (eta c1 - c2) Nt,  Nt = N or -N  = I*eta + N*(+-(eta c1 - c2))
    // so the two scalars are selected, not the six components.
    const T mm = neg_unless(m, c1s < T(0.));
    const T alpha = DIES ? eta : (reflected ? T(1.) : eta);
    const T beta = DIES ? mm : (reflected ? -(T(2.) * c1s) : mm);
    const VecT<T> out = vlin2(I, alpha, N, beta);
    I = KEEP ? vselect(live, out, I) : out;
    return reflected;
}

// aperture test `sqrt(x^2+y^2) > A` (lens.f90:450-454, :576-580, :559-563): decided on
// the squares unless they agree to 1e-12 (then the reference's square root is taken)
template <bool FILT, class T>
__device__ inline bool outside_aperture(T x, T y, T A, T A2, T A2tol, T A2lo, T A2hi, bool live, bool &rare)
{
    const T s2 = x * x + y * y;
    if constexpr (FILT) {                               // A2 = A*A, A2tol = 1e-12 A2 (SurfAuxT)
        // the same margin as two bounds formed once per surface: A2lo = A2 - A2tol, A2hi = A2 + A2tol
        // (SurfAuxT); inside [A2lo, A2hi], or NaN: the reference's square root decides
        const bool out = s2 > A2hi;
        ORT_RARE(4, live & !out & !(s2 < A2lo));
        return out;
    } else {
        return ORT_SQRT(s2) > A;
    }
}

// ----------------------------------------------------------------------------
// In-bottle scattering: tauint (src/surfaces.f90:13-50), the
// Henyey-Greenstein direction update `stokes` (src/stokes.f90:7-166, hgg /= 0 at both call
// sites) and the random walk of src/lens.f90:262-282 / :312-333 — all predicated: `on` marks
// the lanes still walking; every draw is consumed only by those lanes.  Always literal
// (no filtered predicates inside the walk).
// ----------------------------------------------------------------------------
// sine and cosine BY VALUE: results handed back through pointers end up as stack slots whenever
// the optimiser merges two call sites (pointer phis defeat SROA: the fast-fp64 and fp32 kernels
// carried 40 / 20 bytes of scratch per lane for exactly that)
template <class T> struct SinCosT { T s, c; };
template <class T> __device__ inline SinCosT<T> sincos_v(T x);

// The libm calls of the scattering walk (tauint: log; stokes: atan2, sincos twice, acos) and of rang (log).
// T = double: the reference's OWN results — glibc's, whose algorithms ort_libm.h restates operation for operation
// (they are not correctly rounded, so no other accurate implementation returns the same bits, and the walk
// amplifies a last-bit difference of the azimuth to 1e-10 in 2e-5 of its rays).  The reference's compilers turn
// both sin/cos pairs of stokes into sincos() calls (flang: stokes.o imports atan2, acos, sincos only), i.e.
// glibc's non-FMA build of the algorithm.  fast fp64 and fp32 have no bit-exact contract: device library / own.
__device__ inline double log_ref(double x) { return glibc::log_p(x); }
__device__ inline fastd log_ref(fastd x) { return fastd(::log(x.v)); }
__device__ inline float log_ref(float x) { return ::logf(x); }
// TB: where glibc's lookup tables are read (ort_libm.h: TabGlobal = constant memory, TabLds = the workgroup's LDS copy
// — the scattering pipeline's kernel, whose walk is otherwise bound by the table gathers)
template <class TB> __device__ inline double atan2_ref(const TB &tb, double y, double x) { return glibc::atan2_p(tb, y, x); }
template <class TB> __device__ inline fastd atan2_ref(const TB &, fastd y, fastd x) { return fastd(::atan2(y.v, x.v)); }
template <class TB> __device__ inline float atan2_ref(const TB &, float y, float x) { return ::atan2f(y, x); }
template <class TB> __device__ inline double acos_ref(const TB &tb, double x) { return glibc::acos_p(tb, x); }
template <class TB> __device__ inline fastd acos_ref(const TB &, fastd x) { return fastd(::acos(x.v)); }
template <class TB> __device__ inline float acos_ref(const TB &, float x) { return ::acosf(x); }
template <class TB> __device__ inline SinCosT<double> sincos_ref(const TB &tb, double x)
{
    const glibc::SinCos r = glibc::sincos_p<false>(tb, x);
    return {r.s, r.c};
}
template <class TB> __device__ inline SinCosT<fastd> sincos_ref(const TB &, fastd x) { return sincos_v<fastd>(x); }
template <class TB> __device__ inline SinCosT<float> sincos_ref(const TB &, float x) { return sincos_v<float>(x); }

// one leg: optical depth tau = -log(u) against the distance to the cylinder wall.
// ok = false is the reference's `error stop "no intersection"`.
// FILT (scatter_front_kernel): the distance from the filtered quadratic — the same bits, or `rare` (the ray is re-run)
template <class T, class D, bool FILT = false>
__device__ inline void tauint(const RayT<T> &r, T mua, T mus, T cy, T cz, T radius, bool on, D &draws,
                              T &dist, bool &at_wall, bool &ok, int &nis, bool *rare = nullptr)
{
    const T mu_tot = mua + mus;
    const T u = draws.template peek_as<T>();
    draws.advance(on);
    const T tau = -log_ref(u);
    T d;
    bool hit;
    bool unused = false;
    if constexpr (FILT) intersect_quadric<true, T>(r, T(0.), cy, cz, radius, radius * radius, true, on, d, hit, *rare);
    else intersect_quadric<false, T>(r, T(0.), cy, cz, radius, T(0.), true, on, d, hit, unused);
    nis += on ? 1 : 0;
    const T tauradius = d * mu_tot;
    const bool inside = tau < tauradius;
    dist = inside ? ORT_DIV(tau, mu_tot) : d;
    at_wall = !inside;
    ok = hit;
}

template <class T, class D, class TB = glibc::TabGlobal>
__device__ inline void stokes_hg(VecT<T> &dir, T hgg, T twopi, bool on, D &draws, const TB &tb = TB{})
{
    const T pi = twopi * T(0.5);
    const T costp = dir.z;
    const T sintp = ORT_SQRT(T(1.) - costp * costp);
    const T g2 = hgg * hgg;
#ifdef ORT_ISA_MARKERS
    asm volatile("; ORT_FN_BEGIN atan2");
#endif
    const T phip = atan2_ref(tb, dir.y, dir.x);
#ifdef ORT_ISA_MARKERS
    asm volatile("; ORT_FN_END atan2");
#endif
    const T u1 = draws.template peek_as<T>();
    draws.advance(on);
    const T w = ORT_DIV(T(1.) - g2, T(1.) - hgg + T(2.) * hgg * u1);
    T bmu = ORT_DIV((T(1.) + g2) - w * w, T(2.) * hgg);
    T cosb2 = bmu * bmu;
    const bool clamp = fabs(bmu) > T(1.);
    bmu = clamp ? (bmu > T(1.) ? T(1.) : T(-1.)) : bmu;
    cosb2 = clamp ? T(1.) : cosb2;
    const T sinbt = ORT_SQRT(T(1.) - cosb2);
    const T u2 = draws.template peek_as<T>();
    draws.advance(on);
    const T ri1 = twopi * u2;
    const bool upper = ri1 > pi;                         // :75 — the two branches mirror each other
    const T ang = upper ? twopi - ri1 : ri1;
    T sa, ca;
#ifdef ORT_ISA_MARKERS
    asm volatile("; ORT_FN_BEGIN sincos1");
#endif
    { const SinCosT<T> sc_ = sincos_ref(tb, ang); sa = sc_.s; ca = sc_.c; }
#ifdef ORT_ISA_MARKERS
    asm volatile("; ORT_FN_END sincos1");
#endif
    const bool keep = (bmu == T(1.)) || (bmu == T(-1.));   // goto 100: direction unchanged
    T cost = costp * bmu + sintp * sinbt * ca;
    const bool mid = fabs(cost) < T(1.);
    const T sint_m = fabs(ORT_SQRT(T(1.) - cost * cost));
    const T bott = sint_m * sinbt;
    const T sint = mid ? sint_m : T(0.);
    const T sini2 = mid ? ORT_DIV(sa * sintp, sint_m) : T(0.);
    const T cosi2 = mid ? ORT_DIV(costp, bott) - ORT_DIV(cost * bmu, bott) : (cost >= T(1.) ? T(-1.) : T(1.));
    T cosdph = -cosi2 * ca + sini2 * sa * bmu;
    cosdph = fabs(cosdph) > T(1.) ? (cosdph > T(1.) ? T(1.) : T(-1.)) : cosdph;
#ifdef ORT_ISA_MARKERS
    asm volatile("; ORT_FN_BEGIN acos");
#endif
    const T ac = acos_ref(tb, cosdph);
#ifdef ORT_ISA_MARKERS
    asm volatile("; ORT_FN_END acos");
#endif
    T phi = upper ? phip + ac : phip - ac;
    phi = phi > twopi ? phi - twopi : phi;
    phi = phi < T(0.) ? phi + twopi : phi;
    T sp, cp;
    { const SinCosT<T> sc_ = sincos_ref(tb, phi); sp = sc_.s; cp = sc_.c; }
    const VecT<T> nd = {sint * cp, sint * sp, cost};
    dir = vselect(on && !keep, nd, dir);
}

// walk of lens.f90:262-282 / :312-333.  `t` enters as the distance to the wall (from the
// surface's own intersection) and leaves as the length of the last leg; lanes that end here get
// `ended` = ORT_ST_LOST_BOTTLE (absorbed / heading back) or ORT_ST_NO_INTERSECTION.
template <class T, class Surf, class D>
__device__ inline void scatter_walk(const Surf &s, T twopi, RayT<T> &r, T &t, bool on, D &draws, int &nis,
                                    int &ended)
{
    T dist;
    bool at_wall, ok;
    tauint<T>(r, s.mua, s.mus, s.cy, s.cz, s.scat_radius, on, draws, dist, at_wall, ok, nis);
    ended = (on && !ok) ? ORT_ST_NO_INTERSECTION : ended;
    t = on ? dist : t;
    bool alive = on && ok;                               // not yet ended inside the walk
    bool walking = alive && !at_wall;
    const T albedo = ORT_DIV(T(s.mus), T(s.mus + s.mua));
    while (wave_any(walking)) {
        r.pos = vselect(walking, vadd(r.pos, vscale(r.dir, t)), r.pos);
        const T u = draws.template peek_as<T>();
        draws.advance(walking);
        const bool absorbed = walking && !(u < albedo);
        ended = absorbed ? ORT_ST_LOST_BOTTLE : ended;
        alive = alive && !absorbed;
        walking = walking && !absorbed;
        stokes_hg<T>(r.dir, s.hgg, twopi, walking, draws);
        tauint<T>(r, s.mua, s.mus, s.cy, s.cz, s.scat_radius, walking, draws, dist, at_wall, ok, nis);
        const bool lostw = walking && !ok;
        ended = lostw ? ORT_ST_NO_INTERSECTION : ended;
        alive = alive && !lostw;
        t = (walking && ok) ? dist : t;
        const bool out = ORT_SQRT(r.pos.x * r.pos.x + r.pos.z * r.pos.z) >= s.scat_radius;   // sic: x, z
        walking = walking && ok && !out && !at_wall;
    }
    const bool back = alive && (r.dir.z < T(0.));
    ended = back ? ORT_ST_LOST_BOTTLE : ended;
}

// sin and cos of an angle of a few radians (every call site passes an angle in [0, 2 pi] or a
// small fan angle): k = nearest multiple of pi/2, reduced argument by a two-term Cody-Waite
// subtraction with FMAs (exact to ~1e-32 for k <= 8), then the classic minimax kernels on
// [-pi/4, pi/4] (coefficients: Sun fdlibm k_sin.c / k_cos.c), error < 1 ulp — the same class as
// the device library's sincos (<= 2 ulp from the reference's glibc), at a third of its
// instructions (it carries a Payne-Hanek path for huge arguments).  The reduction stays exact
// enough (k * 2^-106) far beyond any angle the tracer forms.
// A 64-bit constant as a SCALAR value: fp64 vector instructions take no 64-bit literal, and left
// to itself the compiler parks such constants in VGPRs for the whole kernel (26 VGPRs for the
// polynomial coefficients below).  Two s_mov_b32 give the register allocator a uniform value it
// keeps in SGPRs (one scalar operand per vector instruction is free).
__device__ inline double scalar_const(double v)
{
    const uint64_t bits = __builtin_bit_cast(uint64_t, v);
    uint32_t lo, hi;
    asm("s_mov_b32 %0, %1" : "=s"(lo) : "i"((uint32_t)bits));
    asm("s_mov_b32 %0, %1" : "=s"(hi) : "i"((uint32_t)(bits >> 32)));
    return __hiloint2double((int)hi, (int)lo);
}
#define ORT_SC(v) scalar_const(v)

// the coefficients live in constant memory: a uniform address, so they arrive through scalar loads
// as 64-bit SGPR pairs that the FMAs read directly (a literal would be parked in a VGPR pair for
// the whole kernel; an SGPR pair assembled from two s_mov_b32 is first copied to VGPRs — two v_mov
// per coefficient)
__device__ __constant__ const double kSinCosTab[16] = {
    6.36619772367581382433e-01,                      // 0  2/pi
    1.57079632679489655800e+00, 6.12323399573676603587e-17,      // 1, 2  pi/2 head, tail
    1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06,     // 3..8  sin: S6 .. S1
    -1.98412698298579493134e-04, 8.33333333332248946124e-03, -1.66666666666666324348e-01,
    -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07,    // 9..14 cos: C6 .. C1
    2.48015872894767294178e-05, -1.38888888888741095749e-03, 4.16666666666666019037e-02,
    0.0};
// a * b + c with c a scalar (SGPR pair) operand of the instruction itself.  Left to the compiler, a
// uniform addend is first copied into the destination VGPR pair (two v_mov_b32) so that the
// two-address v_fmac_f64 can be used.
__device__ inline double fma_sc(double a, double b, double c_uniform)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_uniform));
    return r;
}

__device__ inline SinCosT<double> sincos_small(double x)
{
    typedef const __attribute__((address_space(4))) double *ctab_t;
    const ctab_t C = (ctab_t)kSinCosTab;
    const double k = __builtin_rint(x * C[0]);        // x * 2/pi
    double r = __builtin_fma(-k, C[1], x);
    r = __builtin_fma(-k, C[2], r);
    const double z = r * r;
    double ps = __builtin_fma(z, C[3], C[4]);
    ps = fma_sc(z, ps, C[5]);
    ps = fma_sc(z, ps, C[6]);
    ps = fma_sc(z, ps, C[7]);
    ps = fma_sc(z, ps, C[8]);
    const double sr = __builtin_fma(r * z, ps, r);
    double pc = __builtin_fma(z, C[9], C[10]);
    pc = fma_sc(z, pc, C[11]);
    pc = fma_sc(z, pc, C[12]);
    pc = fma_sc(z, pc, C[13]);
    pc = fma_sc(z, pc, C[14]);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + z * z * pc);                  // k_cos.c's compensated form
    const int n = (int)k;
    const bool swap = (n & 1) != 0;
    const double ss = swap ? cr : sr, cc = swap ? sr : cr;
    return {(n & 2) ? -ss : ss, ((n + 1) & 2) ? -cc : cc};
}
// The fp32 study path: the same scheme in single precision (two-term Cody-Waite reduction with
// fmaf, fdlibm's k_sinf / k_cosf polynomials): < 1 ulp (fp32) on [0, 2 pi]
__device__ inline SinCosT<float> sincos_small_f32(float x)
{
    const float k = __builtin_rintf(x * 6.36619772367581382433e-01f);
    float r = __builtin_fmaf(-k, 1.57079637050628662109375f, x);
    r = __builtin_fmaf(-k, -4.37113900018624283e-8f, r);
    const float z = r * r;
    float ps = __builtin_fmaf(z, 2.7557314297e-06f, -1.9841270114e-04f);
    ps = __builtin_fmaf(z, ps, 8.3333337680e-03f);
    ps = __builtin_fmaf(z, ps, -1.6666667163e-01f);
    const float sr = __builtin_fmaf(r * z, ps, r);
    float pc = __builtin_fmaf(z, -2.7557314297e-07f, 2.4801587642e-05f);
    pc = __builtin_fmaf(z, pc, -1.3888889225e-03f);
    pc = __builtin_fmaf(z, pc, 4.1666667908e-02f);
    const float hz = 0.5f * z;
    const float w = 1.0f - hz;
    const float cr = w + (((1.0f - w) - hz) + z * z * pc);
    const int n = (int)k;
    const bool swap = (n & 1) != 0;
    const float ss = swap ? cr : sr, cc = swap ? sr : cr;
    return {(n & 2) ? -ss : ss, ((n + 1) & 2) ? -cc : cc};
}
// every call site passes an angle in [0, 2 pi] (twopi * u, a bounded fan angle, a direction's
// azimuth); no fallback to the device library here: its constants alone would sit in 18 VGPRs
template <> __device__ inline SinCosT<double> sincos_v<double>(double x) { return sincos_small(x); }
template <> __device__ inline SinCosT<fastd> sincos_v<fastd>(fastd x)
{
    const SinCosT<double> v = sincos_small(x.v);
    return {fastd(v.s), fastd(v.c)};
}
template <> __device__ inline SinCosT<float> sincos_v<float>(float x) { return sincos_small_f32(x); }

// sin / cos of the EMITTERS.  Default: sincos_small (35 instructions, within an ulp of glibc's: emitted rays agree with the
// reference's to 1e-12, outcomes are identical — the bulk loops are emitter-bound, and glibc's own algorithm costs 130).
// strict (kernel variant bit 6, wave-uniform; exact fp64 only): glibc's results bit for bit (ort_libm.h), through the very
// entry the reference's compiled code calls at that site — sincos() (the non-FMA build) for every pair the compilers merge,
// the separate sin() and cos() (FMA builds) for the first pair of `ring` and the polar angle of `create_spot`
// (the reference's sourceMod.o imports sin, cos, sincos; the CPU checker of the tests mirrors it call by call).  Then an
// emitted ray equals the checker's in every bit, and with it everything downstream.
template <class T> __device__ inline SinCosT<T> sincos_em(bool, T x) { return sincos_v<T>(x); }
template <> __device__ inline SinCosT<double> sincos_em<double>(bool strict, double x)
{
    if (strict) { const glibc::SinCos r = glibc::sincos_p<false>(x); return {r.s, r.c}; }
    return sincos_small(x);
}
template <class T> __device__ inline SinCosT<T> sin_cos_em(bool, T x) { return sincos_v<T>(x); }      // sin(x), cos(x) as two calls
template <> __device__ inline SinCosT<double> sin_cos_em<double>(bool strict, double x)
{
    if (strict) { const glibc::SinCos r = glibc::sincos_p<true>(x); return {r.s, r.c}; }
    return sincos_small(x);
}

// ----------------------------------------------------------------------------
// emitters (straight-line)
// ----------------------------------------------------------------------------
// point, src/sourceMod.f90:12-47 (called without offset, src/main.f90:136).  FILT (the queued
// kernels): square roots and normalisations in their range-guarded exact forms (sqrt_f, div3_f),
// a ray outside their ranges raises `rare` and is emitted and traced literally afterwards.
template <class T, bool FILT = false, class Sys, class D>
__device__ inline void emit_point(const Sys &S, RayT<T> &r, D &draws, bool &rare, bool strict = false)
{
    T phi = S.twopi * draws.template next_as<T>();
    T sinp, cosp;
    { const SinCosT<T> sc_ = sincos_em<T>(strict, phi); sinp = sc_.s; cosp = sc_.c; }
    T ran = draws.template next_as<T>();
    T cost = (T(1.0) - ran) + ran * S.cos_theta_max;
    T sint = sqrt_f<FILT, T>(T(1.0) - cost * cost, true, rare);
    // every emitter ends with the same six stores in the same order (pos, then dir): when the
    // optimiser merges the emitters' tails it then merges VALUES, not the addresses of r's fields
    r.pos = {T(0.0), T(0.0), T(0.0) + T(S.point_offset)};       // :45; offset = bottle%centre%z for the isors source (main.f90:140)
    r.dir = {sint * cosp, sint * sinp, cost};
}

// ring, src/sourceMod.f90:250-300
template <class T, bool FILT = false, class Sys, class D>
__device__ inline void emit_ring(const Sys &S, RayT<T> &r, D &draws, bool &rare, bool strict = false)
{
    T rr = S.ring_r1 + draws.template next_as<T>() * (S.ring_r2 - S.ring_r1);     // ranu(r1, r2)
    T theta = draws.template next_as<T>() * S.twopi;
    T st, ct;
    { const SinCosT<T> sc_ = sin_cos_em<T>(strict, theta); st = sc_.s; ct = sc_.c; }      // :272-273: cos(theta), sin(theta) stay two calls
    T sq = sqrt_f<FILT, T>(rr, true, rare);
    T posx = sq * ct;
    T posy = sq * st;
    T Ra = S.ring_bottle_ra;
    T q = S.ring_ellipse ? ORT_DIV(posy * Ra, T(S.ring_bottle_rb)) : posy;            // :277 vs :279
    T posz = S.ring_bottle_z + sqrt_f<FILT, T>(Ra * Ra - q * q, true, rare);
    rr = T(0.) + draws.template next_as<T>() * (S.ring_lens_r2 - T(0.));           // ranu(0., (radius+10e-3)**2)
    theta = draws.template next_as<T>() * S.twopi;
    { const SinCosT<T> sc_ = sincos_em<T>(strict, theta); st = sc_.s; ct = sc_.c; }        // :287-288: merged into sincos()
    sq = sqrt_f<FILT, T>(rr, true, rare);
    T ex = sq * ct - posx;
    T ey = sq * st - posy;
    T ez = S.ring_lens_z - posz;
    T dist = sqrt_f<FILT, T>(ex * ex + ey * ey + ez * ez, true, rare);
    // dir = vector(nxp, nyp, nzp); dir = dir%magnitude() (:292-298): the second normalisation acts on a unit vector
    r.pos = {posx, posy, posz};
    r.dir = vnormalise_est<FILT, T>(div3_f<FILT, T>(VecT<T>{ex, ey, ez}, dist, true, rare), T(1.), T(0.5), T(0.25), T(0x1p-32), true, rare);
}

// create_spot, src/sourceMod.f90:122-159: deterministic fan, no draws; n = 1-based loop index
template <class T, class Sys>
__device__ inline void emit_spot(const Sys &S, RayT<T> &r, uint64_t ray, bool strict = false)
{
    const int n = (int)(ray + 1);
    T phi = S.spot_dphi * (T)(n % 10);
    T theta = S.spot_dtheta * (T)(n / 10);
    T sinp, cosp, sint_unused, cost;
    { const SinCosT<T> sc_ = sincos_em<T>(strict, phi); sinp = sc_.s; cosp = sc_.c; }
    { const SinCosT<T> sc_ = sin_cos_em<T>(strict, theta); sint_unused = sc_.s; cost = sc_.c; }      // cos(theta) alone: the cos() entry
    T sint = ORT_SQRT(T(1.) - cost * cost);
    r.pos = {T(0.), T(0.), T(0.)};
    r.dir = {sint * cosp, sint * sinp, cost};
}

// point_on_bottle, src/sourceMod.f90:50-89 (the "crs" source of phase 1): cone direction as
// `point`, start position = a Gaussian spot (rang, src/random_mod.f90:59-85: polar Box-Muller,
// a variable number of draws) dropped along -z onto the cylinder radiusa + thickness.
// FILT (the crs program): the drop onto the cylinder through the filtered quadratic — the same bits, or `rare`
template <class T, bool FILT = false, class Sys, class D>
__device__ inline void emit_crs(const Sys &S, RayT<T> &r, D &draws, bool &rare, bool strict = false)
{
    T phi = S.twopi * draws.template next_as<T>();
    T sinp, cosp;
    { const SinCosT<T> sc_ = sincos_em<T>(strict, phi); sinp = sc_.s; cosp = sc_.c; }
    T ran = draws.template next_as<T>();
    T cost = (T(1.0) - ran) + ran * S.cos_theta_max;
    T sint = ORT_SQRT(T(1.0) - cost * cost);
    T x = T(0.), y = T(0.), s = T(1.);
    bool more = true;
    while (wave_any(more)) {                               // do while(s >= 1.)
        T u1, u2;
        draws.template next_pair_as<T>(more, u1, u2);
        T xn = T(-1.) + u1 * (T(1.) - T(-1.));             // ranu(-1., 1.)
        T yn = T(-1.) + u2 * (T(1.) - T(-1.));
        x = more ? xn : x;
        y = more ? yn : y;
        s = more ? y * y + x * x : s;
        more = more && (s >= T(1.));
    }
    T cst = ORT_SQRT(ORT_DIV(T(-2.) * log_ref(s), s));
    T tmp1 = T(0.) + S.crs_sigma * (x * cst);
    T tmp2 = T(0.) + S.crs_sigma * (y * cst);
    RayT<T> drop = {{tmp1, tmp2, T(1.0)}, {T(0.), T(0.), T(-1.)}};
    T t;
    bool hit;
    if constexpr (FILT) intersect_quadric<true, T>(drop, T(0.), T(S.crs_cy), T(S.crs_cz), T(S.crs_radius), T(S.crs_radius) * T(S.crs_radius), true, true, t, hit, rare);
    else { bool unused = false; intersect_quadric<false, T>(drop, T(0.), S.crs_cy, S.crs_cz, S.crs_radius, T(0.), true, true, t, hit, unused); }
    t = hit ? t : T(0.);                                   // the reference leaves t undefined on a miss
    r.pos = vadd(drop.pos, vscale(drop.dir, t));
    r.dir = {sint * cosp, sint * sinp, cost};
}

// rang, src/random_mod.f90:59-85: polar Box-Muller, a variable number of draws (predicated loop)
template <class T, class D>
__device__ inline void rang(D &draws, T sigma, T &gx, T &gy)
{
    T x = T(0.), y = T(0.), s = T(1.);
    bool more = true;
    while (wave_any(more)) {                               // do while(s >= 1.)
        T u1, u2;
        draws.template next_pair_as<T>(more, u1, u2);
        T xn = T(-1.) + u1 * (T(1.) - T(-1.));             // ranu(-1., 1.)
        T yn = T(-1.) + u2 * (T(1.) - T(-1.));
        x = more ? xn : x;
        y = more ? yn : y;
        s = more ? y * y + x * x : s;
        more = more && (s >= T(1.));
    }
    T cst = ORT_SQRT(ORT_DIV(T(-2.) * log_ref(s), s));
    gx = T(0.) + sigma * (x * cst);
    gy = T(0.) + sigma * (y * cst);
}

// iSORS(ring = .true.), src/sourceMod.f90:162-247 (the "isors" source of phase 1, src/main.f90:97):
// a Gaussian beam dropped along -z onto an axicon (glass cone, n = 1.4), refracted or reflected
// there (one draw; the flag is not looked at), carried base_pos / dir%z on, put beside the bottle
// and brought to its inner wall; then aimed at a uniform point of the lens disc.  Returns false
// where the reference aborts (`error stop "no intersection with bottle!"`, :216-218: the beam
// reflected at the axicon and flies away from the bottle): the ray ends as ORT_ST_NO_INTERSECTION
// with no further draw.  Literal arithmetic throughout (intersect_cone = src/surfaces.f90:179-224) — except FILT (the
// isors program): the cone's normal, the Fresnel step at it, the bottle quadratic and the aim at the lens in their
// filtered forms (the same bits, or `rare`); the cone's own quadratic has a < 0 and stays literal.
// RING = false: iSORS(ring = .false.), which no call site of the reference reaches (src/main.f90:141 is commented out):
// instead of stopping at the bottle's inner wall the beam goes THROUGH the bottle from outside — bottle_backward_sub,
// src/lens.f90:352-423: outer wall (the full semi-axes: ring_bottle_ra / _rb), Fresnel step air -> glass, inner wall
// (isors_rad1 / _rad2), Fresnel step glass -> contents, each step only if the one before it happened (a miss or a
// reflection returns early; the caller does not look at the flag) — is carried to the plane z = bottle%centre%z and aimed
// at the lens disc of radius L1%radius + 10 mm (ring_lens_r2).  Literal arithmetic; the glass and the contents are those of
// the point loop's bottle (surfaces[1][0]: the inner wall's n1 = contents, n2 = glass; host: check_system).
template <class T, bool FILT = false, bool RING = true, class Sys, class D>
__device__ inline bool emit_isors(const Sys &S, RayT<T> &r, D &draws, bool &rare, bool strict = false)
{
    static_assert(RING || !FILT, "iSORS(ring = .false.) has no filtered form");
    T gx, gy;
    rang<T>(draws, T(S.isors_sigma), gx, gy);
    const T height = T(S.isors_height), k = T(S.isors_k);
    VecT<T> pos = {T(0.) + gx, T(0.) + gy, T(0.) + T(2.) * height};
    VecT<T> dir = {T(0.), T(0.), T(-1.)};
    // intersect_cone: centre = 0
    const T Lz = pos.z - T(0.);
    const T a = dir.x * dir.x + dir.y * dir.y - (k * (dir.z * dir.z));
    const T hb = (dir.x * pos.x) + (dir.y * pos.y) - (k * dir.z * (Lz - height));      // b = 2.*(...)
    const T c = pos.x * pos.x + pos.y * pos.y - (k * ((Lz - height) * (Lz - height)));
    T t;
    bool cone, unused = false;
    solve_and_pick<false, T>(a, hb, c, true, t, cone, unused);
    bool ok = true;
    {
        const VecT<T> hitp = vadd(pos, vscale(dir, t));
        VecT<T> N = {ORT_DIV(T(2.) * (hitp.x - T(0.)), k), ORT_DIV(T(2.) * (hitp.y - T(0.)), k), -(T(2.) * (hitp.z - T(0.))) + T(2.) * height};
        if constexpr (FILT) N = vnormalise_f<true, T>(vscale(N, T(-1.)), cone, rare);
        else N = vnormalise(vscale(N, T(-1.)));
        const T u = draws.template peek_as<T>();
        draws.advance(cone);
        VecT<T> d2 = dir;
        if constexpr (FILT) (void)reflect_refract<true, true, T>(d2, N, T(1.4), T(1.), T(1.4) / T(1.), (T(1.4) / T(1.)) * (T(1.4) / T(1.)), u, cone, rare);
        else (void)reflect_refract<false, true, T>(d2, N, T(1.4), T(1.), T(1.4) / T(1.), T(0.), u, true, unused);
        const T tt = ORT_DIV(T(S.isors_base_pos), d2.z);
        VecT<T> p2 = vadd(hitp, vscale(d2, tt));
        p2.z = T(S.isors_z);
        if constexpr (!RING) {
            const bool ell = __builtin_amdgcn_readfirstlane(S.ring_ellipse) != 0;
            const T cy = T(S.isors_cy), cz = T(S.isors_cz);
            const T nb = T(S.surfaces[1][0].n2), nc = T(S.surfaces[1][0].n1);
            auto wall = [&](const RayT<T> &pr, T ra, T rb, T &tw, bool &hit) {
                if (ell) intersect_ellipse<false, T>(pr, cy, cz, ra, rb, T(0.), T(0.), true, tw, hit, unused);
                else intersect_quadric<false, T>(pr, T(0.), cy, cz, ra, T(0.), true, true, tw, hit, unused);
            };
            T t1, t2;
            bool hit1, hit2;
            wall(RayT<T>{p2, d2}, T(S.ring_bottle_ra), T(S.ring_bottle_rb), t1, hit1);          // lens.f90:362-369
            const bool at1 = cone && hit1;
            const VecT<T> q1 = vselect(at1, vadd(p2, vscale(d2, t1)), p2);
            const VecT<T> N1 = vnormalise(VecT<T>{T(0.), q1.y - cy, q1.z - cz});                  // orig - centre, orig%x = centre%x
            const T ua = draws.template peek_as<T>();
            draws.advance(at1);
            VecT<T> d3 = d2;
            const bool refl1 = reflect_refract<false, true, T>(d3, N1, T(1.), nb, ORT_DIV(T(1.), nb), T(0.), ua, true, unused);
            d3 = vselect(at1, d3, d2);
            const bool in1 = at1 && !refl1;                                                       // :378-381
            wall(RayT<T>{q1, d3}, T(S.isors_rad1), T(S.isors_rad2), t2, hit2);                     // :384-391
            const bool at2 = in1 && hit2;
            const VecT<T> q2 = vselect(at2, vadd(q1, vscale(d3, t2)), q1);
            const VecT<T> N2 = vnormalise(VecT<T>{T(0.), q2.y - cy, q2.z - cz});
            const T ub = draws.template peek_as<T>();
            draws.advance(at2);
            VecT<T> d4 = d3;
            (void)reflect_refract<false, true, T>(d4, N2, nb, nc, ORT_DIV(nb, nc), T(0.), ub, true, unused);
            d4 = vselect(at2, d4, d3);
            const T tz = ORT_DIV(cz - q2.z, d4.z);                                                // sourceMod.f90: to the plane z = centre%z
            const VecT<T> q3 = vadd(q2, vscale(d4, tz));
            pos = vselect(cone, q3, pos);
            dir = vselect(cone, d4, dir);
        } else {
        // bottle inner wall: circular (rad1 = rad2) or elliptical cylinder about x
        const RayT<T> probe = {p2, d2};
        T tb;
        bool hitb;
        if (__builtin_amdgcn_readfirstlane(S.ring_ellipse))
            intersect_ellipse<false, T>(probe, T(S.isors_cy), T(S.isors_cz), T(S.isors_rad1), T(S.isors_rad2), T(0.), T(0.), true, tb, hitb, unused);
        else if constexpr (FILT)
            intersect_quadric<true, T>(probe, T(0.), T(S.isors_cy), T(S.isors_cz), T(S.isors_rad1), T(S.isors_rad1) * T(S.isors_rad1), true, cone, tb, hitb, rare);
        else
            intersect_quadric<false, T>(probe, T(0.), T(S.isors_cy), T(S.isors_cz), T(S.isors_rad1), T(0.), true, true, tb, hitb, unused);
        ok = !cone || hitb;
        const VecT<T> p3 = vadd(p2, vscale(d2, tb));
        pos = vselect(cone, vselect(hitb, p3, p2), pos);    // at the abort the ray sits beside the bottle
        dir = vselect(cone, d2, dir);
        }
    }
    const T aim_r2 = RING ? T(S.isors_lens_r2) : T(S.ring_lens_r2);                 // L1%radius**2 / (L1%radius + 10e-3)**2
    T rr = T(0.) + draws.template peek_as<T>() * (aim_r2 - T(0.));                   // ranu(0., ...)
    draws.advance(ok);
    T theta = draws.template peek_as<T>() * S.twopi;
    draws.advance(ok);
    T st, ct;
    { const SinCosT<T> sc_ = sincos_em<T>(strict, theta); st = sc_.s; ct = sc_.c; }
    T sq = ORT_SQRT(rr);
    T ex = sq * ct - pos.x, ey = sq * st - pos.y, ez = T(S.isors_lens_z) - pos.z;
    VecT<T> aimed;
    if constexpr (FILT) {
        // (as emit_ring: the second normalisation acts on a unit vector)
        const T dist = sqrt_f<true, T>(ex * ex + ey * ey + ez * ez, ok, rare);
        aimed = vnormalise_est<true, T>(div3_f<true, T>(VecT<T>{ex, ey, ez}, dist, ok, rare), T(1.), T(0.5), T(0.25), T(0x1p-32), ok, rare);
    } else {
        const T dist = ORT_SQRT(ex * ex + ey * ey + ez * ez);
        aimed = vnormalise(div3(VecT<T>{ex, ey, ez}, dist));
    }
    r.pos = pos;
    r.dir = vselect(ok, aimed, dir);
    return ok;
}

// emit_image + emit, src/sourceMod.f90:303-361: ray `ray` (serial order) starts in the histogram
// cell s with cdf[s] <= ray < cdf[s+1] (binary search in the 2 MB table, L2-resident), at a
// uniform point of the 9.8 um cell, aimed at a uniform point of the lens disc.  Returns false
// when the histogram is exhausted (the reference re-uses a stale ray there).
// `hint` (the bulk program kernel): a wave emits successive batches of INCREASING ray indices, so the cell of one
// batch's first ray bounds the next batch's cells from below, and they lie a few cells further on (1e9 rays over
// 2^18 cells): a galloping search from the hint takes ~6 dependent loads instead of 18.  Same cell either way:
// the largest s with cdf[s] <= ray.
template <class T, class Sys, class D>
__device__ inline bool emit_image(const Sys &S, const long long *cdf, RayT<T> &r, D &draws, uint64_t ray, int *hint = nullptr, bool strict = false)
{
    const long long key = (long long)ray;
    const bool have = cdf != nullptr && key < cdf[ORT_IMAGE_SOURCE_CELLS];
    int lo = 0, hi = ORT_IMAGE_SOURCE_CELLS;                 // cdf[lo] <= key < cdf[hi]
    if (cdf != nullptr) {
        if (hint != nullptr && *hint >= 0) {
            lo = *hint;                                      // wave-uniform, cdf[lo] <= key
            int span = 1;
            for (;;) {
                const int probe = lo + span < ORT_IMAGE_SOURCE_CELLS ? lo + span : ORT_IMAGE_SOURCE_CELLS;
                const bool up = lo + span < ORT_IMAGE_SOURCE_CELLS && cdf[probe] <= key;
                if (!wave_any(up)) break;
                lo = up ? probe : lo;
                span = up ? span * 2 : span;
            }
            hi = lo + span < ORT_IMAGE_SOURCE_CELLS ? lo + span : ORT_IMAGE_SOURCE_CELLS;
            while (wave_any(hi - lo > 1)) {
                const int mid = (lo + hi) >> 1;
                const bool up = cdf[mid] <= key;
                lo = up ? mid : lo;
                hi = up ? hi : mid;
            }
        } else {
            for (int it = 0; it < 18; ++it) {                // 2^18 cells
                const int mid = (lo + hi) >> 1;
                const bool up = cdf[mid] <= key;
                lo = up ? mid : lo;
                hi = up ? hi : mid;
            }
        }
        if (hint != nullptr) *hint = __builtin_amdgcn_readfirstlane(lo);     // lane 0 holds the batch's smallest ray index
    }
    const int i = lo % 512 + 1, j = lo / 512 + 1;            // first, second index of imgin
    const T dx = T(5000e-6) / T(512.);
    T a = ((T)i - T(1.)) * dx, b = (T)i * dx;
    T x = (a + draws.template next_as<T>() * (b - a)) - T(2500e-6);
    a = ((T)j - T(1.)) * dx; b = (T)j * dx;
    T y = (a + draws.template next_as<T>() * (b - a)) - T(2500e-6);
    T rr = T(0.) + draws.template next_as<T>() * (S.img_lens_r2 - T(0.));
    T theta = draws.template next_as<T>() * S.twopi;
    T st, ct;
    { const SinCosT<T> sc_ = sincos_em<T>(strict, theta); st = sc_.s; ct = sc_.c; }
    T sq = ORT_SQRT(rr);
    T ex = sq * ct - x, ey = sq * st - y, ez = S.img_lens_z - T(0.);
    T dist = ORT_SQRT(ex * ex + ey * ey + ez * ez);
    r.pos = {x, y, T(0.)};
    r.dir = vnormalise(div3(VecT<T>{ex, ey, ez}, dist));
    return have;
}

// the phase's emitter (wave-uniform choice; src/main.f90:95-101, :132-142).  Returns -1, or the
// ORT_ST_* status of a ray the source cannot emit (image source exhausted: the reference re-uses a
// stale ray; isors: the reference aborts).
// ANYSRC = false instantiates only the two default emitters (ring for phase 1, point for phase
// 2): the bulk kernels are compiled once for that case so that the rarely used emitters do not
// cost registers (143 vs 121 VGPRs, i.e. 3 vs 4 waves per SIMD) on the path that is benchmarked.
// EMITTER >= 0 (surface programs): the emitter is the compile-time constant ORT_EMIT_* of the program.
template <class T, bool ANYSRC, bool FILT = false, int EMITTER = -1, class Sys, class D>
__device__ inline int emit(const Sys &S, int phase, RayT<T> &r, D &draws, uint64_t ray, const long long *cdf, bool &rare,
                           int *img_hint = nullptr, bool strict = false)
{
    if constexpr (EMITTER >= 0) {
        bool unused = false;                              // spot and image are literal throughout
        // (`strict` is a literal of the program kernel — trace_queue_kernel's STRICT — and folds away after inlining)
        if constexpr (EMITTER == ORT_EMIT_RING) emit_ring<T, FILT>(S, r, draws, rare, strict);
        else if constexpr (EMITTER == ORT_EMIT_POINT) emit_point<T, FILT>(S, r, draws, rare, strict);
        else if constexpr (EMITTER == ORT_EMIT_SPOT) emit_spot<T>(S, r, ray, strict);
        else if constexpr (EMITTER == ORT_EMIT_CRS) emit_crs<T, FILT>(S, r, draws, rare, strict);
        else if constexpr (EMITTER == ORT_EMIT_ISORS) return emit_isors<T, FILT>(S, r, draws, rare, strict) ? -1 : ORT_ST_NO_INTERSECTION;
        else return emit_image<T>(S, cdf, r, draws, ray, img_hint, strict) ? -1 : ORT_ST_LOST_TELESCOPE;
        (void)unused;
        return -1;
    }
    if (!ANYSRC) {
        if (phase == 1) emit_ring<T, FILT>(S, r, draws, rare, strict);
        else emit_point<T, FILT>(S, r, draws, rare, strict);
        return -1;
    }
    const int e = __builtin_amdgcn_readfirstlane(S.emitter[phase - 1]);
    bool unused = false;                                  // the other emitters are literal throughout
    if (e == ORT_EMIT_RING) emit_ring<T>(S, r, draws, unused, strict);
    else if (e == ORT_EMIT_POINT) emit_point<T>(S, r, draws, unused, strict);
    else if (e == ORT_EMIT_SPOT) emit_spot<T>(S, r, ray, strict);
    else if (e == ORT_EMIT_CRS) emit_crs<T>(S, r, draws, unused, strict);
    else if (e == ORT_EMIT_ISORS) return emit_isors<T>(S, r, draws, unused, strict) ? -1 : ORT_ST_NO_INTERSECTION;
    else if (e == ORT_EMIT_ISORS_NORING) return emit_isors<T, false, false>(S, r, draws, unused, strict) ? -1 : ORT_ST_NO_INTERSECTION;
    else return emit_image<T>(S, cdf, r, draws, ray, nullptr, strict) ? -1 : ORT_ST_LOST_TELESCOPE;
    return -1;
}
template <class T, bool ANYSRC, class Sys, class D>
__device__ inline int emit(const Sys &S, int phase, RayT<T> &r, D &draws, uint64_t ray, const long long *cdf, bool strict = false)
{
    bool unused = false;
    return emit<T, ANYSRC, false>(S, phase, r, draws, ray, cdf, unused, nullptr, strict);
}

// ----------------------------------------------------------------------------
// makeImage2D, src/imageMod.f90:19-58, predicated.  The acceptance test
// acos(x) <= asin(0.22) is evaluated as x >= na_cos_min, where na_cos_min is the
// smallest double whose libm acos is <= asin(0.22), found on the host (no
// transcendental per ray, and the decision is the host libm's, i.e. the
// reference's).  NaN / x > 1 fall through as accepted, exactly as
// `if(angle > na) return` does with a NaN angle.  Returns the ORT_ST_* status.
// ----------------------------------------------------------------------------
template <bool FILT, class T, bool UNIT = false, class Sys>
__device__ inline int make_image(const Sys &S, const RayT<T> &r, bool live, int &xp, int &yp, bool &rare)
{
    bool reject;
    T fx, fy;
    if constexpr (FILT) {
        // x = dir_z / |dir| up to 1e-13; the literal form below rounds it five more times.
        // UNIT (OPT_UNIT_DIR): |dir| = 1 to a few ulps, so dir_z itself is x to 1e-14 (margin 1e-10)
        // (fp32: the same form in every kernel — without margins and deferrals a decision is only the same everywhere if
        // its arithmetic is)
        T xa;
        if constexpr (UNIT && !kLoose<T>) xa = r.dir.z;
        else xa = r.dir.z * rsq_approx(vdot(r.dir, r.dir));
        reject = xa < T(S.na_cos_min);
        // floor(x / binwid) from one multiply.  |q| > 1e3: off the +-200 grid whatever the rounding.
        // Otherwise the product and the quotient differ by < 1e3 * 4.4e-16, so floor agrees unless q
        // is within 1e-9 of an integer.
        const T qx = r.pos.x * T(S.inv_bin_width), qy = r.pos.y * T(S.inv_bin_width);
        fx = floor(qx); fy = floor(qy);
        const T gx = qx - fx, gy = qy - fy;
        const bool na_decided = fabs(xa - T(S.na_cos_min)) > T(1e-10);
        const bool far = (fabs(qx) > T(1e3)) | (fabs(qy) > T(1e3));
        const bool bin_decided = far | ((gx > T(1e-9)) & (gx < T(1. - 1e-9)) & (gy > T(1e-9)) & (gy < T(1. - 1e-9)));   // NaN: undecided
        ORT_RARE(5, live & !na_decided);
        ORT_RARE(6, live & !reject & !bin_decided);
    } else {
        VecT<T> d = vnormalise(r.dir);
        d = vscale(d, T(-1.));
        const T top = (T(0.) * d.x) + (T(0.) * d.y) + (T(-1.) * d.z);
        const T bottom = ORT_SQRT(vdot(d, d)) * T(1.0);
        reject = ORT_DIV(top, bottom) < T(S.na_cos_min);
        fx = floor(ORT_DIV(r.pos.x, T(S.bin_width)));
        fy = floor(ORT_DIV(r.pos.y, T(S.bin_width)));
    }
    const bool off = (r.pos.x > T(1000) || r.pos.y > T(1000)) ||            // :48
                     !(fabs(fx) <= T(200.)) || !(fabs(fy) <= T(200.));       // :52
    const bool binned = live && !reject && !off;
    xp = binned ? (int)fx : xp;
    yp = binned ? (int)fy : yp;
    return reject ? ORT_ST_NA_REJECT : (off ? ORT_ST_OFF_GRID : ORT_ST_BINNED);
}

// status word of the program kernels (surface_step NISK): ORT_ST_* in the low byte, intersections above
__device__ inline int status_code(int st) { return st & 0xff; }
__device__ inline int status_isect(int st) { return st >> 8; }

// ----------------------------------------------------------------------------
// One surface of the staged list for every lane of the wave.  `st` < 0 marks a
// live ray; a ray that ends here gets its final ORT_ST_* status.  nis counts
// surface solves (the metric's unit of work).  The surface record is
// the same for all lanes (wave-uniform index), so kind/flags are branched on as
// scalars.
//   bottle   src/lens.f90:230-350      plano   :425-481     doublet :531-645
//   image    src/optics_system.f90:48-49 + imageMod
// ----------------------------------------------------------------------------
// EXT = false compiles the step without the in-bottle scattering walk (the lean instantiation
// of the bulk kernels; the host picks it when no surface carries ORT_F_SCATTER).
// KEEP = false lets lanes whose ray has ended carry garbage in r (the bulk kernels read only
// st/xp/yp/nis of such lanes); KEEP = true freezes r where the ray ended (debug / tracker output).
// DK >= 0 (surface programs): this step's draw index is the compile-time constant DK for every live
// lane (ProgDraws::at); DK < 0: the draw source counts per lane.
// PART (surface programs, the step at the queue point of trace_queue_kernel): 0 = the whole step;
// 1 = up to the point where the reference knows whether the ray goes on at this surface (moved to
// the surface, hit, inside the aperture): a lane that ends takes its status, the others keep st < 0
// with pos on the surface; 2 = the rest for those lanes (normal, Fresnel draw, new direction).  The
// rays that miss the aperture stop — a third of the point rays at the doublet's first face — then
// leave the wavefront BEFORE the normalisation and the Fresnel arithmetic, not after.  1 then 2 is
// the whole step, operation for operation.
// NISK >= 0 (surface programs): this is step NISK - 1 of the program, so a ray that ends here has evaluated
// exactly NISK intersections: instead of counting per lane and step (`nis` is left alone), the status of a
// ray that ends carries the count, st = ORT_ST_* | NISK << 8 (status_code / status_isect split it).
// OPT: what the step may assume (OPT_*).
template <bool FILT, class T, bool EXT, bool KEEP = true, int KIND = -1, int FLAGS = -1, int HASAP = -1, int DK = -1, bool FRESH = false,
          int PART = 0, int NISK = -1, int OPT = 0, class Sys, class Surf, class D>
__device__ inline void surface_step(const Sys &S, const Surf &s, const SurfAuxT<T> &ax, RayT<T> &r, D &draws,
                                    int &nis, int &st, int &xp, int &yp, bool &rare)
{
    constexpr int tag = (NISK >= 0 ? NISK : 0) << 8;
    constexpr bool DIES = !KEEP && FLAGS >= 0 && (FLAGS & ORT_F_SKIP_ON_REFLECT) != 0;
    static_assert(PART == 0 || (!EXT && !KEEP && KIND >= 0 && KIND != ORT_SURF_IMAGE && KIND != ORT_SURF_IRIS),
                  "half steps exist for the refracting steps of the surface programs");
    const bool live = st < 0;
    const int kind = KIND >= 0 ? KIND : __builtin_amdgcn_readfirstlane(s.kind);
    const unsigned flags = FLAGS >= 0 ? (unsigned)FLAGS : (unsigned)__builtin_amdgcn_readfirstlane((int)s.flags);
    const bool has_ap = HASAP >= 0 ? (HASAP != 0) : aperture_present<T>(s.aperture);   // aperture >= 0
    const int lost = ((flags & ORT_F_BOTTLE) ? ORT_ST_LOST_BOTTLE : ORT_ST_LOST_TELESCOPE) | tag;
    if constexpr (PART == 2) {
        // second half: every lane with st < 0 is on the surface (pos = the crossing point) and goes on
        VecT<T> N2;
        const bool cyl = kind != ORT_SURF_SPHERE;
        if (kind == ORT_SURF_SPHERE || kind == ORT_SURF_CYLINDER || kind == ORT_SURF_ELLIPSE) {
            const VecT<T> Nraw = {cyl ? T(0.0) : s.cx - r.pos.x, s.cy - r.pos.y, s.cz - r.pos.z};
            if (kind == ORT_SURF_ELLIPSE) N2 = vnormalise_f<FILT, T>(Nraw, live, rare, cyl);
            else N2 = vnormalise_est<FILT, T>(Nraw, T(s.radius), ax.rh, ax.rk, ax.r2_tol, live, rare, cyl);
        } else {
            N2 = {T(0.), T(0.), T(-1.)};
        }
        T u2;
        if constexpr (DK >= 0) {
            u2 = draws.template at<T, DK, FRESH>();
        } else {
            u2 = draws.template peek_as<T>();
            draws.advance(live);
        }
        const bool reflected2 = reflect_refract<FILT, KEEP, T, DIES>(r.dir, N2, s.n1, s.n2, s.eta, ax.eta2, u2, live, rare);
        const bool dies2 = reflected2 && (flags & ORT_F_SKIP_ON_REFLECT);
        st = (live & dies2) ? lost : st;
        return;
    }
    if constexpr (NISK < 0) nis += live ? 1 : 0;
    VecT<T> N;
    bool proceed;                    // lanes that reach the Fresnel decision at this surface
    int code = lost;                 // status of a lane that ends at this surface
    if (kind == ORT_SURF_SPHERE || kind == ORT_SURF_CYLINDER || kind == ORT_SURF_ELLIPSE) {
        T t;
        bool hit;
        const bool cyl = kind != ORT_SURF_SPHERE;
        if (kind == ORT_SURF_ELLIPSE) intersect_ellipse<FILT, T>(r, s.cy, s.cz, s.radius, s.radius_b, ax.ell_sa, ax.ell_sb, live, t, hit, rare);
        else intersect_quadric<FILT, T, (KIND == ORT_SURF_SPHERE ? (OPT & ~4) : (OPT & ~2))>(r, s.cx, s.cy, s.cz, s.radius, ax.r2, cyl, live, t, hit, rare,
                                                                                             ax.ax_ly, ax.ax_lz, ax.ax_c);
        int walk_end = -1;
        if (EXT && (flags & ORT_F_SCATTER)) {               // wave-uniform
            scatter_walk<T>(s, S.twopi, r, t, live && hit && !rare, draws, nis, walk_end);
            hit = hit && walk_end < 0;
        }
        const VecT<T> moved = vmad(r.dir, t, r.pos);
        r.pos = KEEP ? vselect(live && hit, moved, r.pos) : moved;
        bool out = false;
        if (has_ap) out = outside_aperture<FILT, T>(moved.x, moved.y, s.aperture, ax.ap2, ax.ap_tol, ax.ap_lo, ax.ap_hi, live && hit, rare);
        // normal = centre - pos, with orig%x = centre%x for the bottle (lens.f90:288-290)
        if constexpr (PART != 1) {
            const VecT<T> Nraw = {cyl ? T(0.0) : s.cx - moved.x, s.cy - moved.y, s.cz - moved.z};
            if (kind == ORT_SURF_ELLIPSE) N = vnormalise_f<FILT, T>(Nraw, live && hit, rare, cyl);
            else N = vnormalise_est<FILT, T>(Nraw, T(s.radius), ax.rh, ax.rk, ax.r2_tol, live && hit, rare, cyl);   // |N| = radius
        }
        // a lane that ends here: missed (Help3 where the reference aborts), outside the aperture,
        // reflected (all: `lost`), or ended inside the scattering walk
        if (flags & ORT_F_MISS_IS_HELP3) code = hit ? lost : (ORT_ST_HELP3 | tag);
        if (EXT) code = walk_end >= 0 ? walk_end : code;
        proceed = live && hit && !out;
    } else {
        // plane kinds: d = (z_plane - pos%z) / dir%z ; pos = pos + dir*d
        const T d = ORT_DIV(s.cz - r.pos.z, r.dir.z);
        const VecT<T> moved = vmad(r.dir, d, r.pos);
        if (kind == ORT_SURF_IMAGE) {
            r.pos = KEEP ? vselect(live, moved, r.pos) : moved;
            const int ist = make_image<FILT, T, (OPT & OPT_UNIT_DIR) != 0>(S, r, live, xp, yp, rare) | tag;
            st = live ? ist : st;
            return;
        }
        bool out = false;
        if (has_ap) out = outside_aperture<FILT, T>(moved.x, moved.y, s.aperture, ax.ap2, ax.ap_tol, ax.ap_lo, ax.ap_hi, live, rare);
        if (kind == ORT_SURF_IRIS) {
            if (KEEP) r.pos = vselect(live && out, moved, r.pos);   // pos = origpos unless lost (lens.f90:564, :643)
            st = (live && out) ? lost : st;
            return;
        }
        r.pos = KEEP ? vselect(live, moved, r.pos) : moved;
        N = {T(0.), T(0.), T(-1.)};                         // flatNormal, lens.f90:165
        proceed = live && !out;
    }
    if constexpr (PART == 1) {
        st = live ? (proceed ? -1 : code) : st;
        return;
    }
    T u;
    if constexpr (DK >= 0) {
        u = draws.template at<T, DK, FRESH>();
    } else {
        u = draws.template peek_as<T>();
        draws.advance(proceed);
    }
    const bool reflected = reflect_refract<FILT, KEEP, T, DIES>(r.dir, N, s.n1, s.n2, s.eta, ax.eta2, u, proceed, rare);
    const bool dies = reflected && (flags & ORT_F_SKIP_ON_REFLECT);
    st = live ? ((proceed & !dies) ? -1 : code) : st;
}

}  // namespace ort
