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
// (surface kind via readfirstlane, "does any lane need the literal formula" via
// ballot), plus the exec-masked side effects (image atomic, LDS queue traffic).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ort.h"

namespace ort {

// ----------------------------------------------------------------------------
// ORT-RNG-v1: counter-based uniforms keyed on (seed, phase, global ray, draw).
// Replaces the semantics of ran2() = one U[0,1) per call (src/random_mod.f90:39-46);
// the reference's own generator is the Fortran runtime's and is not reproducible
// across compilers or thread counts (SURVEY §7 "hard parts").
//   base = mix64(seed ^ (GOLDEN*phase)); z = base + GOLDEN*((ray<<24) + k + 1)
//   u = (mix64(z) >> 11) * 2^-53
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

// top 53 bits of x as a double in [0,1): both halves convert exactly and their
// sum has <= 53 significant bits, so this equals (double)(x >> 11) * 2^-53.
__device__ inline double bits_to_unit(uint64_t x)
{
    uint32_t hi = (uint32_t)(x >> 32);
    uint32_t lo = (uint32_t)(x >> 11) & 0x1FFFFFu;
    return (double)hi * 0x1.0p-32 + (double)lo * 0x1.0p-53;
}

// Per-ray draw source.  peek() is the next uniform, advance(c) consumes it where c.
// Keyed stream, or (parity entry) an explicit table: draw k at table[k*stride].
struct Draws {
    uint64_t z;            // base + GOLDEN*((ray<<24) + k)
    const double *table;
    int64_t stride;
    int len;
    int k;                 // draws consumed so far

    __device__ inline void init_keyed(uint64_t base, uint64_t ray, int first_draw)
    {
        table = nullptr; stride = 0; len = 0; k = first_draw;
        z = base + kGolden * ((ray << 24) + (uint64_t)first_draw);
    }
    __device__ inline void init_table(const double *t, int64_t s, int l, int first_draw)
    {
        table = t; stride = s; len = l; k = first_draw; z = 0;
    }
    __device__ inline double peek() const
    {
        if (table) return k < len ? table[(int64_t)k * stride] : 0.5;
        return bits_to_unit(mix64(z + kGolden));
    }
    __device__ inline void advance(bool c)
    {
        k += c ? 1 : 0;
        z += c ? kGolden : 0ull;
    }
    __device__ inline double next() { double u = peek(); advance(true); return u; }
};

// keyed-only variant used by the production kernels (one 64-bit counter per lane)
struct KeyedDraws {
    uint64_t z;
    __device__ inline void init_keyed(uint64_t base, uint64_t ray, int first_draw)
    {
        z = base + kGolden * ((ray << 24) + (uint64_t)first_draw);
    }
#ifdef ORT_ABL_NORNG
    __device__ inline double peek() const { return 0.73; }
#else
    __device__ inline double peek() const { return bits_to_unit(mix64(z + kGolden)); }
#endif
    __device__ inline void advance(bool c) { z += c ? kGolden : 0ull; }
    __device__ inline double next() { z += kGolden; return bits_to_unit(mix64(z)); }
};

// Development-only ablation switches (tools/ablate.sh): they BREAK the numerics contract
// and exist to price the IEEE divide / square-root expansions.  Never defined in the build.
#ifdef ORT_ABL_FASTSQRT
#define ORT_SQRT(x) __builtin_amdgcn_sqrt(x)
#else
#define ORT_SQRT(x) sqrt(x)
#endif
#ifdef ORT_ABL_FASTDIV
#define ORT_DIV(a, b) ((a) * __builtin_amdgcn_rcp(b))
#else
#define ORT_DIV(a, b) ((a) / (b))
#endif

// ----------------------------------------------------------------------------
// 3-vector algebra, src/vector_class.f90:48-186
// ----------------------------------------------------------------------------
struct Vec { double x, y, z; };

__device__ inline Vec vsub(Vec a, Vec b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline Vec vadd(Vec a, Vec b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline Vec vscale(Vec a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ inline double vdot(Vec a, Vec b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
__device__ inline Vec vselect(bool c, Vec a, Vec b) { return {c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }
// magnitude_fn (:175-186): NORMALISES, by three divisions
__device__ inline Vec vnormalise(Vec a)
{
    double tmp = ORT_SQRT(a.x * a.x + a.y * a.y + a.z * a.z);
    return {ORT_DIV(a.x, tmp), ORT_DIV(a.y, tmp), ORT_DIV(a.z, tmp)};
}

struct Ray { Vec pos, dir; };

__device__ inline bool wave_any(bool p) { return __ballot(p) != 0ull; }

// ----------------------------------------------------------------------------
// Filtered predicates (FILT = true, the production setting).
//
// What only feeds a DECISION — reflect or refract (u <= R), inside the aperture,
// which root of the quadratic, NA acceptance, which image bin — is first evaluated
// with a cheap approximation whose error bound is orders of magnitude smaller than
// the margin it is tested against; only when some lane of the wave lands inside
// the margin (probability ~1e-10 per test, and always for the special cases: total
// internal reflection, costt >= 1 at normal incidence, tangent rays, NaN) is the
// reference's literal formula evaluated (wave-uniform branch) and used for those
// lanes.  The decision taken is therefore always the reference's, and outcomes stay
// bit-identical, while ~4 of the ~13 fp64 divide / square-root expansions per
// surface leave the common path.  FILT = false evaluates every predicate literally
// (kept for A/B and for tests).
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

// ----------------------------------------------------------------------------
// solveQuadratic (src/surfaces.f90:227-260) + root choice (:75-86), predicated.
// Away from tangency the order of the two roots follows from signs alone
// (q^2 - a*c = (|b| sqrt(D) + D)/2 >= 1e-5 q^2 under the filter, a > 0, and
// rounding is monotone), so only the quotient the reference ends up returning is
// divided out, with its operands chosen by selects:
//   q > 0 : roots c/q <= q/a ; c/q < 0 <=> c < 0  -> t = c < 0 ? q/a : c/q   (always a hit)
//   q < 0 : q/a < 0 ; hit <=> c/q >= 0 <=> c <= 0 -> t = c/q
// ----------------------------------------------------------------------------
template <bool FILT>
__device__ inline void solve_and_pick(double a, double b, double c, bool live, double &t, bool &hit)
{
    const double discrim = b * b - 4.0 * a * c;
    const bool neg = discrim < 0.0;                   // :243 — no real root
    const double sq = ORT_SQRT(discrim);                  // NaN when neg: those lanes are misses
    const double q = (b > 0.0) ? -0.5 * (b + sq) : -0.5 * (b - sq);
    bool ok = false;
    t = 0.0;
    hit = false;
    if (FILT) {
        const double bb = b * b;
        ok = discrim > 1e-10 * bb && a > 1e-10 && a < 1e10 && bb < 1e200 &&
             fabs(q) > 1e-100 && (c == 0.0 || fabs(c) > 1e-200);
        const bool qpos = q > 0.0;
        const bool use_qa = qpos && (c < 0.0);
        t = ORT_DIV(use_qa ? q : c, use_qa ? a : q);
        hit = qpos || !(c > 0.0);
    }
    const bool slow = !neg && !ok;                    // tangent, degenerate or NaN: literal formula
    if (wave_any(live && slow)) {
        const bool dz = discrim == 0.0;               // :245-247
        const double xd = -0.5 * b / a;
        double t0 = dz ? xd : q / a;
        double t1 = dz ? xd : c / q;
        const bool sw = t0 > t1;                      // :75-79
        const double lo = sw ? t1 : t0, hi = sw ? t0 : t1;
        const bool lneg = lo < 0.0;                   // :80-83
        t = slow ? (lneg ? hi : lo) : t;
        hit = slow ? !(lneg && hi < 0.0) : hit;
    }
    hit = hit && !neg;
}

// intersect_sphere (src/surfaces.f90:52-89) and intersect_cylinder (:91-130) in
// one body: the x-axis cylinder is the sphere with the x terms removed
// (a = dz^2+dy^2 etc. — fp addition commutes, so the sums are bit-identical).
template <bool FILT>
__device__ inline void intersect_quadric(const Ray &r, double cx, double cy, double cz, double radius,
                                         bool cylinder, bool live, double &t, bool &hit)
{
    double Lx = cylinder ? 0.0 : r.pos.x - cx;
    double Ly = r.pos.y - cy;
    double Lz = r.pos.z - cz;
    double dx = cylinder ? 0.0 : r.dir.x;
    double a = (dx * dx) + (r.dir.y * r.dir.y) + (r.dir.z * r.dir.z);
    double b = 2.0 * ((dx * Lx) + (r.dir.y * Ly) + (r.dir.z * Lz));
    double c = ((Lx * Lx) + (Ly * Ly) + (Lz * Lz)) - radius * radius;
    solve_and_pick<FILT>(a, b, c, live, t, hit);
}

// intersect_ellipse, src/surfaces.f90:133-176
template <bool FILT>
__device__ inline void intersect_ellipse(const Ray &r, double cy, double cz, double semia, double semib,
                                         bool live, double &t, bool &hit)
{
    double sa = 1. / (semia * semia);
    double sb = 1. / (semib * semib);
    double Ly = r.pos.y - cy;
    double Lz = r.pos.z - cz;
    double a = sa * (r.dir.z * r.dir.z) + sb * (r.dir.y * r.dir.y);
    double b = 2 * (sa * r.dir.z * Lz + sb * r.dir.y * Ly);
    double c = sa * (Lz * Lz) + sb * (Ly * Ly) - 1;
    solve_and_pick<FILT>(a, b, c, live, t, hit);
}

// fresnel, src/surfaces.f90:336-372, as one expression (eta = n1/n2 rounded once on
// the host).  costt > 1 (rounding at normal incidence) makes sintt NaN, every
// comparison false, tir NaN -> 1: the reference then ALWAYS reflects; kept.
__device__ inline double fresnel(double costt, double n1, double n2, double eta)
{
    double sintt = ORT_SQRT(1. - costt * costt);
    double sint2 = eta * sintt;
    double cost2 = ORT_SQRT(1. - sint2 * sint2);
    double r1 = fabs((n1 * costt - n2 * cost2) / (n1 * costt + n2 * cost2));
    double r2 = fabs((n1 * cost2 - n2 * costt) / (n1 * cost2 + n2 * costt));
    double tir = 0.5 * (r1 * r1 + r2 * r2);
    tir = (tir != tir || tir > 1. || tir < 0.) ? 1. : tir;     // :366-369
    return (sint2 > 1.) ? 1.0 : ((costt == 1.) ? 0. : tir);    // :353-358
}

// reflect_refract (src/surfaces.f90:262-282) with reflect (:285-300) and refract
// (:303-333), predicated: the direction is committed where `live`.  The caller
// supplies the uniform u and consumes the draw.  Returns true where the ray reflected.
// FILT: R is only ever compared with u, so it is first formed with two approximate
// reciprocals and with refract's own c2 standing in for fresnel's cost2 (the same
// quantity, rounded along another path; they differ by < 1e-13 once k > 1e-6).
// |R' - R| < 1e-12, the margin is 1e-10.
template <bool FILT>
__device__ inline bool reflect_refract(Vec &I, Vec N, double n1, double n2, double eta, double u, bool live)
{
    const double c1s = vdot(N, I);                       // == vdot(I, N): the products commute
    const double c1 = fabs(c1s);                         // costt (fresnel) and |c1| (refract)
    const double k = 1.0 - eta * eta * (1.0 - c1 * c1);  // refract's radicand, refract's order (:327)
    const double c2 = ORT_SQRT(k);                           // NaN beyond total reflection: unused there
    bool reflected = false, decided = false;
    if (FILT) {
        double a1 = n1 * c1, b1 = n2 * c2, a2 = n1 * c2, b2 = n2 * c1;
        double f1 = (a1 - b1) * rcp_approx(a1 + b1);
        double f2 = (a2 - b2) * rcp_approx(a2 + b2);
        double R = 0.5 * (f1 * f1 + f2 * f2);
        // NaN anywhere -> undecided; c1 >= 1, k ~ 0 or < 0 (total reflection) -> literal path
        decided = (k > 1e-6) && (c1 < 1.0) && (fabs(u - R) > 1e-10);
        reflected = u < R;
    }
    if (wave_any(live && !decided)) {
        bool rl = u <= fresnel(c1, n1, n2, eta);         // :275
        reflected = decided ? reflected : rl;
    }
    const Vec refl = vsub(I, vscale(N, 2. * c1s));       // :297
    const Vec Nt = (c1s < 0.) ? N : vscale(N, -1.);      // :320-325
    const Vec refr = vadd(vscale(I, eta), vscale(Nt, eta * c1 - c2));   // :329
    I = vselect(live, vselect(reflected, refl, refr), I);
    return reflected;
}

// aperture test `sqrt(x^2+y^2) > A` (lens.f90:450-454, :576-580, :559-563): decided on
// the squares unless they agree to 1e-12 (then the reference's square root is taken)
template <bool FILT>
__device__ inline bool outside_aperture(double x, double y, double A, bool live)
{
    const double s2 = x * x + y * y;
    const double A2 = A * A;
    bool out = s2 > A2;
    const bool near = FILT ? !(fabs(s2 - A2) > 1e-12 * A2) : true;
    if (wave_any(live && near)) out = near ? (ORT_SQRT(s2) > A) : out;
    return out;
}

// ----------------------------------------------------------------------------
// emitters (straight-line)
// ----------------------------------------------------------------------------
// point, src/sourceMod.f90:12-47 (called without offset, src/main.f90:136)
template <class D>
__device__ inline void emit_point(const ort_system &S, Ray &r, D &draws)
{
    double phi = S.twopi * draws.next();
    double sinp, cosp;
    sincos(phi, &sinp, &cosp);
    double ran = draws.next();
    double cost = (1.0 - ran) + ran * S.cos_theta_max;
    double sint = ORT_SQRT(1.0 - cost * cost);
    r.dir = {sint * cosp, sint * sinp, cost};
    r.pos = {0.0, 0.0, 0.0};
}

// ring, src/sourceMod.f90:250-300
template <class D>
__device__ inline void emit_ring(const ort_system &S, Ray &r, D &draws)
{
    double rr = S.ring_r1 + draws.next() * (S.ring_r2 - S.ring_r1);       // ranu(r1, r2)
    double theta = draws.next() * S.twopi;
    double st, ct;
    sincos(theta, &st, &ct);
    double sq = ORT_SQRT(rr);
    double posx = sq * ct;
    double posy = sq * st;
    double Ra = S.ring_bottle_ra;
    double q = S.ring_ellipse ? posy * Ra / S.ring_bottle_rb : posy;       // :277 vs :279
    double posz = S.ring_bottle_z + ORT_SQRT(Ra * Ra - q * q);
    r.pos = {posx, posy, posz};
    rr = 0. + draws.next() * (S.ring_lens_r2 - 0.);                        // ranu(0., (radius+10e-3)**2)
    theta = draws.next() * S.twopi;
    sincos(theta, &st, &ct);
    sq = ORT_SQRT(rr);
    double ex = sq * ct - r.pos.x;
    double ey = sq * st - r.pos.y;
    double ez = S.ring_lens_z - r.pos.z;
    double dist = ORT_SQRT(ex * ex + ey * ey + ez * ez);
    r.dir = vnormalise({ORT_DIV(ex, dist), ORT_DIV(ey, dist), ORT_DIV(ez, dist)});
}

// ----------------------------------------------------------------------------
// makeImage2D, src/imageMod.f90:19-58, predicated.  The acceptance test
// acos(x) <= asin(0.22) is evaluated as x >= na_cos_min, where na_cos_min is the
// smallest double whose libm acos is <= asin(0.22), found on the host (no
// transcendental per ray, and the decision is the host libm's, i.e. the
// reference's).  NaN / x > 1 fall through as accepted, exactly as
// `if(angle > na) return` does with a NaN angle.  Returns the ORT_ST_* status.
// ----------------------------------------------------------------------------
template <bool FILT>
__device__ inline int make_image(const ort_system &S, const Ray &r, bool live, int &xp, int &yp)
{
    bool na_decided = false, reject = false;
    if (FILT) {
        // x = dir_z / |dir| up to 1e-13; the literal form below rounds it five more times
        double xa = r.dir.z * rsq_approx(vdot(r.dir, r.dir));
        na_decided = fabs(xa - S.na_cos_min) > 1e-10;
        reject = xa < S.na_cos_min;
    }
    if (wave_any(live && !na_decided)) {
        Vec d = vnormalise(r.dir);
        d = vscale(d, -1.);
        double top = (0. * d.x) + (0. * d.y) + (-1. * d.z);
        double bottom = ORT_SQRT(vdot(d, d)) * 1.0;
        bool rl = (top / bottom) < S.na_cos_min;
        reject = na_decided ? reject : rl;
    }
    double fx = 0., fy = 0.;
    bool bin_decided = false;
    if (FILT) {
        // floor(x / binwid) from one multiply unless the quotient is within 1e-6 of an integer
        // (|q * 2.3e-16| < 1e-9 for |q| < 4e6)
        double qx = r.pos.x * S.inv_bin_width, qy = r.pos.y * S.inv_bin_width;
        fx = floor(qx); fy = floor(qy);
        double gx = qx - fx, gy = qy - fy;
        bin_decided = gx > 1e-6 && gx < 1. - 1e-6 && gy > 1e-6 && gy < 1. - 1e-6 &&
                      fabs(qx) < 1e6 && fabs(qy) < 1e6;
    }
    if (wave_any(live && !reject && !bin_decided)) {
        double lx = floor(r.pos.x / S.bin_width), ly = floor(r.pos.y / S.bin_width);
        fx = bin_decided ? fx : lx;
        fy = bin_decided ? fy : ly;
    }
    const bool off = (r.pos.x > 1000 || r.pos.y > 1000) ||            // :48
                     !(fabs(fx) <= 200.) || !(fabs(fy) <= 200.);       // :52
    const bool binned = live && !reject && !off;
    xp = binned ? (int)fx : xp;
    yp = binned ? (int)fy : yp;
    return reject ? ORT_ST_NA_REJECT : (off ? ORT_ST_OFF_GRID : ORT_ST_BINNED);
}

// ----------------------------------------------------------------------------
// One surface of the staged list for every lane of the wave.  `st` < 0 marks a
// live ray; a ray that ends here gets its final ORT_ST_* status.  nis counts
// surface solves (the metric's unit of work).  The surface record is
// the same for all lanes (wave-uniform index), so kind/flags are branched on as
// scalars.
//   bottle   src/lens.f90:230-350      plano   :425-481     doublet :531-645
//   image    src/optics_system.f90:48-49 + imageMod
// ----------------------------------------------------------------------------
template <bool FILT, class D>
__device__ inline void surface_step(const ort_system &S, const ort_surface &s, Ray &r, D &draws,
                                    int &nis, int &st, int &xp, int &yp)
{
    const bool live = st < 0;
    const int kind = __builtin_amdgcn_readfirstlane(s.kind);
    const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane((int)s.flags);
    const bool has_ap = __builtin_amdgcn_readfirstlane(__double2hiint(s.aperture)) >= 0;   // aperture >= 0
    const int lost = (flags & ORT_F_BOTTLE) ? ORT_ST_LOST_BOTTLE : ORT_ST_LOST_TELESCOPE;
    nis += live ? 1 : 0;
    Vec N;
    bool proceed;                    // lanes that reach the Fresnel decision at this surface
    int ended = -1;                  // status of lanes that end before it
    if (kind == ORT_SURF_SPHERE || kind == ORT_SURF_CYLINDER || kind == ORT_SURF_ELLIPSE) {
        double t;
        bool hit;
        const bool cyl = kind != ORT_SURF_SPHERE;
        if (kind == ORT_SURF_ELLIPSE) intersect_ellipse<FILT>(r, s.cy, s.cz, s.radius, s.radius_b, live, t, hit);
        else intersect_quadric<FILT>(r, s.cx, s.cy, s.cz, s.radius, cyl, live, t, hit);
        const Vec moved = vadd(r.pos, vscale(r.dir, t));
        r.pos = vselect(live && hit, moved, r.pos);
        bool out = false;
        if (has_ap) out = outside_aperture<FILT>(moved.x, moved.y, s.aperture, live && hit);
        // normal = centre - pos, with orig%x = centre%x for the bottle (lens.f90:288-290)
        N = vnormalise({cyl ? 0.0 : s.cx - moved.x, s.cy - moved.y, s.cz - moved.z});
        ended = !hit ? ((flags & ORT_F_MISS_IS_HELP3) ? ORT_ST_HELP3 : lost) : (out ? lost : -1);
        proceed = live && hit && !out;
    } else {
        // plane kinds: d = (z_plane - pos%z) / dir%z ; pos = pos + dir*d
        const double d = ORT_DIV(s.cz - r.pos.z, r.dir.z);
        const Vec moved = vadd(r.pos, vscale(r.dir, d));
        if (kind == ORT_SURF_IMAGE) {
            r.pos = vselect(live, moved, r.pos);
            const int ist = make_image<FILT>(S, r, live, xp, yp);
            st = live ? ist : st;
            return;
        }
        bool out = false;
        if (has_ap) out = outside_aperture<FILT>(moved.x, moved.y, s.aperture, live);
        if (kind == ORT_SURF_IRIS) {
            r.pos = vselect(live && out, moved, r.pos);     // pos = origpos unless lost (lens.f90:564, :643)
            st = (live && out) ? lost : st;
            return;
        }
        r.pos = vselect(live, moved, r.pos);
        N = {0., 0., -1.};                                  // flatNormal, lens.f90:165
        ended = out ? lost : -1;
        proceed = live && !out;
    }
    const double u = draws.peek();
    draws.advance(proceed);
    const bool reflected = reflect_refract<FILT>(r.dir, N, s.n1, s.n2, s.eta, u, proceed);
    const bool dies = reflected && (flags & ORT_F_SKIP_ON_REFLECT);
    st = live ? (proceed ? (dies ? lost : -1) : ended) : st;
}

}  // namespace ort
