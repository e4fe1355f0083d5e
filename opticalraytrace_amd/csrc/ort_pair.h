// ort_pair.h — the fp32 path (BASELINE configs[4]) with TWO RAYS PER LANE: an experiment kept as evidence
// (kernel variant bit 5), NOT the default — it is bit-identical to the one-ray kernels and 4 % SLOWER.
//
// The idea.  The 157 TFLOP/s fp32 vector peak of the MI355X exists only for the PACKED forms v_pk_add_f32 /
// v_pk_mul_f32 / v_pk_fma_f32, which work on a pair of values held in a 64-bit register pair.  Pairing unrelated
// scalars of ONE ray does not pay (the operands must first be moved next to each other: the compiler's SLP
// vectoriser spends a v_mov per packed operand).  So a lane carries two independent rays, A and B, component by
// component in register pairs (x_A, x_B), (y_A, y_B), ...: every arithmetic instruction of the walk is then packed
// by construction and serves two rays; what has no packed form — compares, selects, v_rcp_f32 / v_sqrt_f32,
// conversions, the integer hash — is issued once per ray.  Static count of the point program: 703 packed
// arithmetic instructions + 1 147 others per PAIR of rays against 2 x 1 279 for two single rays: 28 % fewer.
//
// Why it loses (tools/ubench3.hip on the MI355X, issue cost per instruction per SIMD, saturated, in s_memtime ticks):
//   v_mul_f32 / v_add_f32 1.82     v_fma_f32 2.43     v_fma_f64 2.80     v_pk_{mul,add,fma}_f32 2.88
//   v_rcp_f32 / v_sqrt_f32 5.46    v_cmp_f32 6.28
// A packed instruction costs 1.58 two-operand fp32 instructions: it is worth two of them only for fma.  This path is
// 583 mul/add + 119 fma per ray, so packing buys 1.26 x on 55 % of the instructions, and the pair kernel pays for it
// with the second ray's selects and compares staying un-packed, more hazard nops (203 vs 56: v_cmp -> v_cndmask
// through SGPR masks), 84 VGPRs and 29-33 KB of LDS (5 waves per SIMD instead of 8).  Measured, 1e7 point rays:
// 0.2261 ms (two rays per lane) vs 0.2170 ms (one ray per lane); ring loop 0.0830 vs 0.0735 ms.
//
// The arithmetic per ray is, operation for operation and in the same order, that of the one-ray fp32
// instantiations of ort_device.h (literal predicates; hardware reciprocal + one correction step for '/', hardware
// square root) — packed IEEE fp32 operations round like the scalar ones — so a ray ends where the fp32 lockstep
// kernel puts it, bit for bit (tests/test_gpu_fp32.py: pair kernel == lockstep kernel, images and counters).
// Each function cites the reference lines through the ort_device.h function it mirrors.
#pragma once
#include "ort_device.h"

namespace ort {
namespace pk {

typedef float f2_t __attribute__((ext_vector_type(2)));

// a predicate per ray of the pair
struct pb { bool a, b; };
__device__ inline pb operator&(pb x, pb y) { return {(bool)(x.a & y.a), (bool)(x.b & y.b)}; }
__device__ inline pb operator|(pb x, pb y) { return {(bool)(x.a | y.a), (bool)(x.b | y.b)}; }
__device__ inline pb operator!(pb x) { return {!x.a, !x.b}; }
__device__ inline pb both(bool v) { return {v, v}; }
__device__ inline bool any_lane(pb p) { return wave_any(p.a | p.b); }

// a value per ray of the pair
struct pf {
    f2_t v;
    __device__ pf() = default;
    __device__ pf(float s) : v{s, s} {}
    __device__ pf(float a, float b) : v{a, b} {}
    __device__ explicit pf(f2_t w) : v(w) {}
};
__device__ inline pf operator+(pf a, pf b) { return pf(a.v + b.v); }
__device__ inline pf operator-(pf a, pf b) { return pf(a.v - b.v); }
__device__ inline pf operator*(pf a, pf b) { return pf(a.v * b.v); }
__device__ inline pf operator-(pf a) { return pf(-a.v); }
__device__ inline pb operator<(pf a, pf b) { return {a.v.x < b.v.x, a.v.y < b.v.y}; }
__device__ inline pb operator>(pf a, pf b) { return {a.v.x > b.v.x, a.v.y > b.v.y}; }
__device__ inline pb operator<=(pf a, pf b) { return {a.v.x <= b.v.x, a.v.y <= b.v.y}; }
__device__ inline pb operator>=(pf a, pf b) { return {a.v.x >= b.v.x, a.v.y >= b.v.y}; }
__device__ inline pb operator==(pf a, pf b) { return {a.v.x == b.v.x, a.v.y == b.v.y}; }
__device__ inline pb operator!=(pf a, pf b) { return {a.v.x != b.v.x, a.v.y != b.v.y}; }
__device__ inline pf sel(pb c, pf x, pf y) { return pf(c.a ? x.v.x : y.v.x, c.b ? x.v.y : y.v.y); }
__device__ inline pf pfma(pf a, pf b, pf c) { return pf(__builtin_elementwise_fma(a.v, b.v, c.v)); }
__device__ inline pf pabs(pf a) { return pf(__builtin_fabsf(a.v.x), __builtin_fabsf(a.v.y)); }
__device__ inline pf pfloor(pf a) { return pf(__builtin_floorf(a.v.x), __builtin_floorf(a.v.y)); }
__device__ inline pf prcp(pf a) { return pf(__builtin_amdgcn_rcpf(a.v.x), __builtin_amdgcn_rcpf(a.v.y)); }
// div_t(float, float), sqrt_t(float) of ort_device.h
__device__ inline pf pdiv(pf a, pf b)
{
    const pf r = prcp(b);
    const pf q = a * r;
    return pfma(pfma(-b, q, a), r, q);
}
__device__ inline pf psqrt(pf a) { return pf(__builtin_amdgcn_sqrtf(a.v.x), __builtin_amdgcn_sqrtf(a.v.y)); }

struct pvec { pf x, y, z; };
__device__ inline pvec vadd(pvec a, pvec b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline pvec vscale(pvec a, pf s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ inline pf vdot(pvec a, pvec b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
__device__ inline pvec vsel(pb c, pvec a, pvec b) { return {sel(c, a.x, b.x), sel(c, a.y, b.y), sel(c, a.z, b.z)}; }
// vnormalise<float> (magnitude_fn, src/vector_class.f90:175-186): sqrt, then three divisions
__device__ inline pvec vnormalise(pvec a)
{
    const pf t = psqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return {pdiv(a.x, t), pdiv(a.y, t), pdiv(a.z, t)};
}
struct pray { pvec pos, dir; };

// status / integer per ray of the pair
struct pi { int a, b; };
__device__ inline pi sel(pb c, pi x, pi y) { return {c.a ? x.a : y.a, c.b ? x.b : y.b}; }
__device__ inline pb live_of(pi st) { return {st.a < 0, st.b < 0}; }

// the draws of the pair: ProgDraws of ort_device.h, one per ray (the hash is integer work: issued per ray)
struct PairDraws {
    ProgDraws a, b;
    __device__ inline void init_index(uint64_t z0, uint32_t ia, uint32_t ib)
    {
        a.init_index(z0, ia, 0);
        b.init_index(z0, ib, 0);
    }
    template <int K, bool FRESH> __device__ inline pf at()
    {
        return pf(a.template at<float, K, FRESH>(), b.template at<float, K, FRESH>());
    }
};

// sincos_small_f32 of ort_device.h for both rays
struct psincos { pf s, c; };
__device__ inline psincos sincos_pair(pf x)
{
    const pf xk = x * pf(6.36619772367581382433e-01f);
    const pf k = pf(__builtin_rintf(xk.v.x), __builtin_rintf(xk.v.y));
    pf r = pfma(-k, pf(1.57079637050628662109375f), x);
    r = pfma(-k, pf(-4.37113900018624283e-8f), r);
    const pf z = r * r;
    pf ps = pfma(z, pf(2.7557314297e-06f), pf(-1.9841270114e-04f));
    ps = pfma(z, ps, pf(8.3333337680e-03f));
    ps = pfma(z, ps, pf(-1.6666667163e-01f));
    const pf sr = pfma(r * z, ps, r);
    pf pc = pfma(z, pf(-2.7557314297e-07f), pf(2.4801587642e-05f));
    pc = pfma(z, pc, pf(-1.3888889225e-03f));
    pc = pfma(z, pc, pf(4.1666667908e-02f));
    const pf hz = pf(0.5f) * z;
    const pf w = pf(1.0f) - hz;
    const pf cr = w + (((pf(1.0f) - w) - hz) + z * z * pc);
    const int na = (int)k.v.x, nb = (int)k.v.y;
    const pb swap = {(na & 1) != 0, (nb & 1) != 0};
    const pf ss = sel(swap, cr, sr), cc = sel(swap, sr, cr);
    const pb negs = {(na & 2) != 0, (nb & 2) != 0}, negc = {((na + 1) & 2) != 0, ((nb + 1) & 2) != 0};
    return {sel(negs, -ss, ss), sel(negc, -cc, cc)};
}

// emit_point<float> (src/sourceMod.f90:12-47)
template <class Sys>
__device__ inline void emit_point(const Sys &S, pray &r, PairDraws &d)
{
    const pf phi = pf(S.twopi) * d.template at<0, false>();
    const psincos sc = sincos_pair(phi);
    const pf ran = d.template at<1, false>();
    const pf cost = (pf(1.0f) - ran) + ran * pf(S.cos_theta_max);
    const pf sint = psqrt(pf(1.0f) - cost * cost);
    r.pos = {pf(0.0f), pf(0.0f), pf(0.0f) + pf(S.point_offset)};
    r.dir = {sint * sc.c, sint * sc.s, cost};
}

// emit_ring<float> (src/sourceMod.f90:250-300)
template <class Sys>
__device__ inline void emit_ring(const Sys &S, pray &r, PairDraws &d)
{
    pf rr = pf(S.ring_r1) + d.template at<0, false>() * pf(S.ring_r2 - S.ring_r1);
    pf theta = d.template at<1, false>() * pf(S.twopi);
    psincos sc = sincos_pair(theta);
    pf sq = psqrt(rr);
    const pf posx = sq * sc.c;
    const pf posy = sq * sc.s;
    const pf Ra = pf(S.ring_bottle_ra);
    const pf q = S.ring_ellipse ? pdiv(posy * Ra, pf(S.ring_bottle_rb)) : posy;
    const pf posz = pf(S.ring_bottle_z) + psqrt(Ra * Ra - q * q);
    rr = pf(0.f) + d.template at<2, false>() * pf(S.ring_lens_r2 - 0.f);
    theta = d.template at<3, false>() * pf(S.twopi);
    sc = sincos_pair(theta);
    sq = psqrt(rr);
    const pf ex = sq * sc.c - posx;
    const pf ey = sq * sc.s - posy;
    const pf ez = pf(S.ring_lens_z) - posz;
    const pf dist = psqrt(ex * ex + ey * ey + ez * ez);
    r.pos = {posx, posy, posz};
    r.dir = vnormalise(pvec{pdiv(ex, dist), pdiv(ey, dist), pdiv(ez, dist)});
}

// solve_and_pick<false, float> (solveQuadratic + root choice, src/surfaces.f90:227-260, :75-86)
__device__ inline void solve_and_pick(pf a, pf hb, pf c, pf &t, pb &hit)
{
    const pf b = pf(2.0f) * hb;
    const pf discrim = b * b - pf(4.0f) * a * c;
    const pb neg = discrim < pf(0.0f);
    const pf sq = psqrt(discrim);
    const pf q = sel(b > pf(0.0f), pf(-0.5f) * (b + sq), pf(-0.5f) * (b - sq));
    const pb dz = discrim == pf(0.0f);
    const pf xd = pdiv(pf(-0.5f) * b, a);
    const pf t0 = sel(dz, xd, pdiv(q, a));
    const pf t1 = sel(dz, xd, pdiv(c, q));
    const pb sw = t0 > t1;
    const pf lo = sel(sw, t1, t0), hi = sel(sw, t0, t1);
    const pb lneg = lo < pf(0.0f);
    t = sel(lneg, hi, lo);
    hit = !(lneg & (hi < pf(0.0f))) & !neg;
}

// intersect_quadric<false, float, OPT> (intersect_sphere / intersect_cylinder, src/surfaces.f90:52-130)
template <bool CYL, bool AXIS>
__device__ inline void intersect_quadric(const pray &r, float cx, float cy, float cz, float radius, pf &t, pb &hit)
{
    const pf Lx = CYL ? pf(0.0f) : (AXIS ? r.pos.x : r.pos.x - pf(cx));
    const pf Ly = AXIS ? r.pos.y : r.pos.y - pf(cy);
    const pf Lz = r.pos.z - pf(cz);
    const pf dx = CYL ? pf(0.0f) : r.dir.x;
    const pf a = (dx * dx) + (r.dir.y * r.dir.y) + (r.dir.z * r.dir.z);
    const pf hb = (dx * Lx) + (r.dir.y * Ly) + (r.dir.z * Lz);
    const pf c = ((Lx * Lx) + (Ly * Ly) + (Lz * Lz)) - pf(radius * radius);
    solve_and_pick(a, hb, c, t, hit);
}

// intersect_ellipse<false, float> (src/surfaces.f90:133-176)
__device__ inline void intersect_ellipse(const pray &r, float cy, float cz, float semia, float semib, pf &t, pb &hit)
{
    const float sa1 = div_t(1.f, semia * semia), sb1 = div_t(1.f, semib * semib);
    const pf sa = pf(sa1), sb = pf(sb1);
    const pf Ly = r.pos.y - pf(cy);
    const pf Lz = r.pos.z - pf(cz);
    const pf a = sa * (r.dir.z * r.dir.z) + sb * (r.dir.y * r.dir.y);
    const pf hb = sa * r.dir.z * Lz + sb * r.dir.y * Ly;
    const pf c = sa * (Lz * Lz) + sb * (Ly * Ly) - pf(1.f);
    solve_and_pick(a, hb, c, t, hit);
}

// fresnel<float> (src/surfaces.f90:336-372)
__device__ inline pf fresnel(pf costt, float n1, float n2, float eta)
{
    const pf sintt = psqrt(pf(1.f) - costt * costt);
    const pf sint2 = pf(eta) * sintt;
    const pf cost2 = psqrt(pf(1.f) - sint2 * sint2);
    const pf r1 = pabs(pdiv(pf(n1) * costt - pf(n2) * cost2, pf(n1) * costt + pf(n2) * cost2));
    const pf r2 = pabs(pdiv(pf(n1) * cost2 - pf(n2) * costt, pf(n1) * cost2 + pf(n2) * costt));
    pf tir = pf(0.5f) * (r1 * r1 + r2 * r2);
    tir = sel((tir != tir) | (tir > pf(1.f)) | (tir < pf(0.f)), pf(1.f), tir);
    return sel(sint2 > pf(1.f), pf(1.0f), sel(costt == pf(1.f), pf(0.f), tir));
}

// reflect_refract<false, false, float, DIES> (src/surfaces.f90:262-333); returns where the ray reflected
template <bool DIES>
__device__ inline pb reflect_refract(pvec &I, pvec N, float n1, float n2, float eta, pf u)
{
    const pf c1s = vdot(N, I);
    const pf c1 = pabs(c1s);
    const pf k = pf(1.0f) - pf(eta * eta) * (pf(1.0f) - c1 * c1);
    const pf c2 = psqrt(k);
    const pf ec1 = pf(eta) * c1;
    const pf m = ec1 - c2;
    const pb reflected = u <= fresnel(c1, n1, n2, eta);
    const pf mm = sel(c1s < pf(0.f), m, -m);
    const pf alpha = DIES ? pf(eta) : sel(reflected, pf(1.f), pf(eta));
    const pf beta = DIES ? mm : sel(reflected, -(pf(2.f) * c1s), mm);
    I = vadd(vscale(I, alpha), vscale(N, beta));
    return reflected;
}

// make_image<false, float> (makeImage2D, src/imageMod.f90:19-58); returns the ORT_ST_* status of each ray
template <class Sys>
__device__ inline pi make_image(const Sys &S, const pray &r, pb live, pi &xp, pi &yp)
{
    pvec d = vnormalise(r.dir);
    d = vscale(d, pf(-1.f));
    const pf top = (pf(0.f) * d.x) + (pf(0.f) * d.y) + (pf(-1.f) * d.z);
    const pf bottom = psqrt(vdot(d, d)) * pf(1.0f);
    const pb reject = pdiv(top, bottom) < pf(S.na_cos_min);
    const pf fx = pfloor(pdiv(r.pos.x, pf(S.bin_width)));
    const pf fy = pfloor(pdiv(r.pos.y, pf(S.bin_width)));
    const pb off = ((r.pos.x > pf(1000.f)) | (r.pos.y > pf(1000.f))) | !(pabs(fx) <= pf(200.f)) | !(pabs(fy) <= pf(200.f));
    const pb binned = live & !reject & !off;
    xp = sel(binned, pi{(int)fx.v.x, (int)fx.v.y}, xp);
    yp = sel(binned, pi{(int)fy.v.x, (int)fy.v.y}, yp);
    return sel(reject, pi{ORT_ST_NA_REJECT, ORT_ST_NA_REJECT}, sel(off, pi{ORT_ST_OFF_GRID, ORT_ST_OFF_GRID}, pi{ORT_ST_BINNED, ORT_ST_BINNED}));
}

// surface_step<false, float, false, false, KIND, FLAGS, HASAP, DK, FRESH, PART, NISK, OPT> for the pair: one step of a
// surface program (bottle src/lens.f90:230-350, plano :425-481, doublet :531-645, image src/optics_system.f90:48-49)
template <int KIND, int FLAGS, int HASAP, int DK, bool FRESH, int PART, int NISK, bool AXIS, class Sys, class Surf>
__device__ inline void surface_step(const Sys &S, const Surf &s, pray &r, PairDraws &d, pi &st, pi &xp, pi &yp)
{
    constexpr int tag = NISK << 8;
    constexpr bool DIES = kDietDiesOnReflect && (FLAGS & ORT_F_SKIP_ON_REFLECT) != 0;
    constexpr int lost1 = ((FLAGS & ORT_F_BOTTLE) ? ORT_ST_LOST_BOTTLE : ORT_ST_LOST_TELESCOPE) | tag;
    const pi lost = {lost1, lost1};
    const pb live = live_of(st);
    constexpr bool quadric = KIND == ORT_SURF_SPHERE || KIND == ORT_SURF_CYLINDER || KIND == ORT_SURF_ELLIPSE;
    constexpr bool cyl = KIND != ORT_SURF_SPHERE;
    if constexpr (PART == 2) {
        pvec N2;
        if constexpr (quadric) N2 = vnormalise(pvec{cyl ? pf(0.0f) : pf(s.cx) - r.pos.x, pf(s.cy) - r.pos.y, pf(s.cz) - r.pos.z});
        else N2 = {pf(0.f), pf(0.f), pf(-1.f)};
        const pf u2 = d.template at<DK, FRESH>();
        const pb reflected2 = reflect_refract<DIES>(r.dir, N2, s.n1, s.n2, s.eta, u2);
        const pb dies2 = (FLAGS & ORT_F_SKIP_ON_REFLECT) ? reflected2 : both(false);
        st = sel(live & dies2, lost, st);
        return;
    }
    pvec N;
    pb proceed;
    pi code = lost;
    if constexpr (quadric) {
        pf t;
        pb hit;
        if constexpr (KIND == ORT_SURF_ELLIPSE) intersect_ellipse(r, s.cy, s.cz, s.radius, s.radius_b, t, hit);
        else intersect_quadric<cyl, AXIS && !cyl>(r, s.cx, s.cy, s.cz, s.radius, t, hit);
        const pvec moved = vadd(r.pos, vscale(r.dir, t));
        r.pos = moved;
        pb out = both(false);
        if constexpr (HASAP != 0) out = psqrt(moved.x * moved.x + moved.y * moved.y) > pf(s.aperture);
        if constexpr (PART != 1) N = vnormalise(pvec{cyl ? pf(0.0f) : pf(s.cx) - moved.x, pf(s.cy) - moved.y, pf(s.cz) - moved.z});
        if constexpr ((FLAGS & ORT_F_MISS_IS_HELP3) != 0) code = sel(hit, lost, pi{ORT_ST_HELP3 | tag, ORT_ST_HELP3 | tag});
        proceed = live & hit & !out;
    } else {
        const pf dd = pdiv(pf(s.cz) - r.pos.z, r.dir.z);
        const pvec moved = vadd(r.pos, vscale(r.dir, dd));
        if constexpr (KIND == ORT_SURF_IMAGE) {
            r.pos = moved;
            pi ist = make_image(S, r, live, xp, yp);
            ist = {ist.a | tag, ist.b | tag};
            st = sel(live, ist, st);
            return;
        }
        pb out = both(false);
        if constexpr (HASAP != 0) out = psqrt(moved.x * moved.x + moved.y * moved.y) > pf(s.aperture);
        if constexpr (KIND == ORT_SURF_IRIS) {
            st = sel(live & out, lost, st);              // pos = origpos unless lost (lens.f90:564, :643)
            return;
        }
        r.pos = moved;
        N = {pf(0.f), pf(0.f), pf(-1.f)};
        proceed = live & !out;
    }
    if constexpr (PART == 1) {
        st = sel(live, sel(proceed, pi{-1, -1}, code), st);
        return;
    }
    const pf u = d.template at<DK, FRESH>();
    const pb reflected = reflect_refract<DIES>(r.dir, N, s.n1, s.n2, s.eta, u);
    const pb dies = (FLAGS & ORT_F_SKIP_ON_REFLECT) ? reflected : both(false);
    st = sel(live, sel(proceed & !dies, pi{-1, -1}, code), st);
}

}  // namespace pk
}  // namespace ort
