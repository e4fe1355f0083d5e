// ort_device.h — per-ray device functions of the MI355X trace path (gfx950, wave64).
//
// Numerics contract: every + - * / sqrt below is a separately rounded IEEE-754
// fp64 operation in the order the reference writes it (this file is compiled with
// -ffp-contract=off; fp64 '/' and sqrt lower to correctly rounded sequences), so
// that for identical input rays and identical uniforms the result equals the
// reference's bit for bit.  Reference = lewisfish/OpticalRayTrace, all `real` fp64
// (src/Makefile:2).  Each function cites the reference lines it implements.
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

// Per-ray draw source: keyed stream, or an explicit table (parity entry).
struct Draws {
    uint64_t z;            // base + GOLDEN*((ray<<24) + k), advanced by GOLDEN per draw
    const double *table;   // table mode when non-null: draw k at table[k*stride]
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
    __device__ inline double next()
    {
        int kk = k++;
        if (table) return kk < len ? table[(int64_t)kk * stride] : 0.5;
        z += kGolden;
        return bits_to_unit(mix64(z));
    }
};

// keyed-only variant used by the production kernels (no table pointer in registers)
struct KeyedDraws {
    uint64_t z;
    int k;
    __device__ inline void init_keyed(uint64_t base, uint64_t ray, int first_draw)
    {
        k = first_draw;
        z = base + kGolden * ((ray << 24) + (uint64_t)first_draw);
    }
    __device__ inline double next()
    {
        ++k;
        z += kGolden;
        return bits_to_unit(mix64(z));
    }
};

// ----------------------------------------------------------------------------
// 3-vector algebra, src/vector_class.f90:48-186
// ----------------------------------------------------------------------------
struct Vec { double x, y, z; };

__device__ inline Vec vsub(Vec a, Vec b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline Vec vadd(Vec a, Vec b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline Vec vscale(Vec a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ inline double vdot(Vec a, Vec b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
// magnitude_fn (:175-186): NORMALISES, by three divisions
__device__ inline Vec vnormalise(Vec a)
{
    double tmp = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return {a.x / tmp, a.y / tmp, a.z / tmp};
}

struct Ray { Vec pos, dir; };

// ----------------------------------------------------------------------------
// Filtered predicates (FILT = true, the production setting).
//
// Everything that feeds the ray STATE (t, pos, normal, refracted / reflected dir) is
// computed with the reference's own operations, correctly rounded, in its order.
// What only feeds a DECISION — reflect or refract (u <= R), inside the aperture,
// which root of the quadratic, NA acceptance, which image bin — is first evaluated
// with a cheap approximation whose error bound is orders of magnitude smaller than
// the margin it is tested against; only when the approximation lands inside the
// margin (probability ~1e-10 per test, and always for the special cases: total
// internal reflection, normal incidence, tangent rays, NaN) is the reference's
// literal formula evaluated.  The decision taken is therefore always the
// reference's, and outcomes stay bit-identical, while ~4 of the ~13 fp64 divide /
// square-root expansions per surface disappear from the common path.
// FILT = false evaluates every predicate literally (kept for A/B and for tests).
// ----------------------------------------------------------------------------
// 1/y with relative error < 2^-40 for finite normal y: hardware seed (v_rcp_f64,
// >= 22 good bits) + one Newton step.
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
// solveQuadratic, src/surfaces.f90:227-260 + root choice :75-86
// ----------------------------------------------------------------------------
template <bool FILT>
__device__ inline bool solve_and_pick(double a, double b, double c, double &t)
{
    double discrim = b * b - 4.0 * a * c;
    if (discrim < 0.0) return false;
    if (FILT) {
        // Away from tangency the order of the two roots follows from signs alone:
        // q^2 - a*c = (|b| sqrt(D) + D)/2 >= 1e-5 q^2 here, so with a > 0
        //   q > 0:  q/a >= c/q  (RN is monotone: the rounded quotients keep the order)
        //   q < 0:  q/a < 0, and the answer is c/q when that is >= 0.
        // Only the quotient the reference ends up returning is divided out.
        double bb = b * b;
        if (discrim > 1e-10 * bb && a > 1e-10 && a < 1e10 && bb < 1e200) {
            double sq = sqrt(discrim);
            double q = (b > 0.0) ? -0.5 * (b + sq) : -0.5 * (b - sq);
            if (fabs(q) > 1e-100) {
                double x1 = c / q;
                if (q > 0.0) {
                    if (x1 < 0.0) x1 = q / a;        // smaller root behind the ray: take the larger (> 0)
                } else if (x1 < 0.0) {
                    return false;                   // both roots behind the ray
                }
                t = x1;
                return true;
            }
        }
    }
    double t0, t1;
    if (discrim == 0.0) {
        t0 = -0.5 * b / a;
        t1 = t0;
    } else {
        double sq = sqrt(discrim);
        double q = (b > 0.0) ? -0.5 * (b + sq) : -0.5 * (b - sq);
        t0 = q / a;
        t1 = c / q;
    }
    if (t0 > t1) { double tmp = t1; t1 = t0; t0 = tmp; }
    if (t0 < 0.0) {
        t0 = t1;
        if (t0 < 0.0) return false;
    }
    t = t0;
    return true;
}

// intersect_sphere (src/surfaces.f90:52-89) and intersect_cylinder (:91-130) in
// one body: the x-axis cylinder is the sphere with the x terms removed
// (a = dz^2+dy^2 etc. — fp addition commutes, so the sums are bit-identical).
template <bool FILT>
__device__ inline bool intersect_quadric(const Ray &r, double cx, double cy, double cz,
                                         double radius, bool cylinder, double &t)
{
    double Lx = cylinder ? 0.0 : r.pos.x - cx;
    double Ly = r.pos.y - cy;
    double Lz = r.pos.z - cz;
    double dx = cylinder ? 0.0 : r.dir.x;
    double a = (dx * dx) + (r.dir.y * r.dir.y) + (r.dir.z * r.dir.z);
    double b = 2.0 * ((dx * Lx) + (r.dir.y * Ly) + (r.dir.z * Lz));
    double c = ((Lx * Lx) + (Ly * Ly) + (Lz * Lz)) - radius * radius;
    return solve_and_pick<FILT>(a, b, c, t);
}

// intersect_ellipse, src/surfaces.f90:133-176
template <bool FILT>
__device__ inline bool intersect_ellipse(const Ray &r, double cy, double cz,
                                         double semia, double semib, double &t)
{
    double sa = 1. / (semia * semia);
    double sb = 1. / (semib * semib);
    double Ly = r.pos.y - cy;
    double Lz = r.pos.z - cz;
    double a = sa * (r.dir.z * r.dir.z) + sb * (r.dir.y * r.dir.y);
    double b = 2 * (sa * r.dir.z * Lz + sb * r.dir.y * Ly);
    double c = sa * (Lz * Lz) + sb * (Ly * Ly) - 1;
    return solve_and_pick<FILT>(a, b, c, t);
}

// fresnel, src/surfaces.f90:336-372 (eta = n1/n2 rounded once on the host)
__device__ inline double fresnel(Vec I, Vec N, double n1, double n2, double eta)
{
    double costt = fabs(vdot(I, N));
    double sintt = sqrt(1. - costt * costt);
    double sint2 = eta * sintt;
    if (sint2 > 1.) return 1.0;
    if (costt == 1.) return 0.;
    double cost2 = sqrt(1. - sint2 * sint2);
    double r1 = fabs((n1 * costt - n2 * cost2) / (n1 * costt + n2 * cost2));
    double r2 = fabs((n1 * cost2 - n2 * costt) / (n1 * cost2 + n2 * costt));
    double tir = 0.5 * (r1 * r1 + r2 * r2);
    if (tir != tir || tir > 1. || tir < 0.) tir = 1.;
    return tir;
}

// reflect_refract (src/surfaces.f90:262-282) with reflect (:285-300) and refract
// (:303-333).  Consumes exactly one draw.  Returns true when the ray reflected.
// FILT: R is only ever compared with u, so it is first formed with two approximate
// reciprocals and with refract's own c2 standing in for fresnel's cost2 (the same
// quantity, rounded along another path; they differ by < 1e-13 once k > 1e-6).
// |R' - R| < 1e-12, the margin is 1e-10.
template <bool FILT, class D>
__device__ inline bool reflect_refract(Vec &I, Vec N, double n1, double n2, double eta, D &draws)
{
    double u = draws.next();
    double c1s = vdot(N, I);                       // == vdot(I, N): the products commute
    double c1 = fabs(c1s);                         // costt (fresnel) and |c1| (refract)
    double k = 1.0 - eta * eta * (1.0 - c1 * c1);  // refract's radicand, refract's order (:327)
    double c2 = sqrt(k);                           // NaN beyond total reflection: unused there
    bool reflected = false;
    bool decided = false;
    if (FILT) {
        double a1 = n1 * c1, b1 = n2 * c2, a2 = n1 * c2, b2 = n2 * c1;
        double f1 = (a1 - b1) * rcp_approx(a1 + b1);
        double f2 = (a2 - b2) * rcp_approx(a2 + b2);
        double R = 0.5 * (f1 * f1 + f2 * f2);
        decided = (k > 1e-6) && (c1 < 1.0) && (fabs(u - R) > 1e-10);   // NaN anywhere -> false;
                                                                      // c1 >= 1 (rounding at normal incidence): literal path (R = 0, or NaN -> 1)
        reflected = u < R;
    }
    if (!decided) reflected = u <= fresnel(I, N, n1, n2, eta);
    if (reflected) {
        double s = 2. * c1s;
        I = vsub(I, vscale(N, s));
        return true;
    }
    Vec Nt = N;
    if (!(c1s < 0.)) Nt = vscale(N, -1.);
    I = vadd(vscale(I, eta), vscale(Nt, eta * c1 - c2));
    return false;
}

// ----------------------------------------------------------------------------
// emitters
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
    double sint = sqrt(1.0 - cost * cost);
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
    double sq = sqrt(rr);
    double posx = sq * ct;
    double posy = sq * st;
    double Ra = S.ring_bottle_ra;
    double posz;
    if (S.ring_ellipse) {
        double q = posy * Ra / S.ring_bottle_rb;
        posz = S.ring_bottle_z + sqrt(Ra * Ra - q * q);
    } else {
        posz = S.ring_bottle_z + sqrt(Ra * Ra - posy * posy);
    }
    r.pos = {posx, posy, posz};
    rr = 0. + draws.next() * (S.ring_lens_r2 - 0.);                        // ranu(0., (radius+10e-3)**2)
    theta = draws.next() * S.twopi;
    sincos(theta, &st, &ct);
    sq = sqrt(rr);
    double ex = sq * ct - r.pos.x;
    double ey = sq * st - r.pos.y;
    double ez = S.ring_lens_z - r.pos.z;
    double dist = sqrt(ex * ex + ey * ey + ez * ez);
    r.dir = vnormalise({ex / dist, ey / dist, ez / dist});
}

// ----------------------------------------------------------------------------
// makeImage2D, src/imageMod.f90:19-58.  The acceptance test acos(x) <= asin(0.22)
// is evaluated as x >= na_cos_min, where na_cos_min is the smallest double whose
// libm acos is <= asin(0.22), found on the host (no transcendental per ray, and
// the decision is the host libm's, i.e. the reference's).  NaN / x > 1 fall
// through as accepted, exactly as `if(angle > na) return` does with a NaN angle.
// ----------------------------------------------------------------------------
template <bool FILT>
__device__ inline int make_image(const ort_system &S, const Ray &r, int &xp, int &yp)
{
    bool decided = false, reject = false;
    if (FILT) {
        // x = dir_z / |dir| up to 1e-13; the literal form below rounds it five more times
        double xa = r.dir.z * rsq_approx(vdot(r.dir, r.dir));
        decided = fabs(xa - S.na_cos_min) > 1e-10;
        reject = xa < S.na_cos_min;
    }
    if (!decided) {
        Vec d = vnormalise(r.dir);
        d = vscale(d, -1.);
        double top = (0. * d.x) + (0. * d.y) + (-1. * d.z);
        double bottom = sqrt(vdot(d, d)) * 1.0;
        reject = (top / bottom) < S.na_cos_min;
    }
    if (reject) return ORT_ST_NA_REJECT;
    if (r.pos.x > 1000 || r.pos.y > 1000) return ORT_ST_OFF_GRID;
    double fx = 0., fy = 0.;
    bool binned = false;
    if (FILT) {
        // floor(x / binwid) from one multiply unless the quotient is within 1e-6 of an integer
        // (|q * 2.3e-16| < 1e-9 for |q| < 4e6)
        double qx = r.pos.x * S.inv_bin_width, qy = r.pos.y * S.inv_bin_width;
        fx = floor(qx); fy = floor(qy);
        double gx = qx - fx, gy = qy - fy;
        binned = gx > 1e-6 && gx < 1. - 1e-6 && gy > 1e-6 && gy < 1. - 1e-6 &&
                 fabs(qx) < 1e6 && fabs(qy) < 1e6;
    }
    if (!binned) {
        fx = floor(r.pos.x / S.bin_width);
        fy = floor(r.pos.y / S.bin_width);
    }
    if (!(fabs(fx) <= 200.) || !(fabs(fy) <= 200.)) return ORT_ST_OFF_GRID;
    xp = (int)fx;
    yp = (int)fy;
    return ORT_ST_BINNED;
}

// ----------------------------------------------------------------------------
// One surface of the staged list.  Returns -1 to continue with the next surface,
// otherwise the ray's final ORT_ST_* status.  nis counts surface solves (the
// metric's unit of work).
//   bottle   src/lens.f90:230-350      plano   :425-481     doublet :531-645
//   image    src/optics_system.f90:48-49 + imageMod
// ----------------------------------------------------------------------------
// aperture test `sqrt(x^2+y^2) > A` (lens.f90:450-454, :576-580, :559-563): decided on the
// squares unless they agree to 1e-12 (then the reference's square root is taken)
template <bool FILT>
__device__ inline bool outside_aperture(double x, double y, double A)
{
    double s2 = x * x + y * y;
    if (FILT) {
        double A2 = A * A;
        if (fabs(s2 - A2) > 1e-12 * A2) return s2 > A2;
    }
    return sqrt(s2) > A;
}

template <bool FILT, class D>
__device__ inline int surface_step(const ort_system &S, const ort_surface &s, Ray &r, D &draws,
                                   int &nis, int &xp, int &yp)
{
    const int kind = s.kind;
    const int lost = (s.flags & ORT_F_BOTTLE) ? ORT_ST_LOST_BOTTLE : ORT_ST_LOST_TELESCOPE;
    Vec N;
    if (kind == ORT_SURF_SPHERE || kind == ORT_SURF_CYLINDER || kind == ORT_SURF_ELLIPSE) {
        double t;
        bool hit;
        const bool cyl = kind != ORT_SURF_SPHERE;
        if (kind == ORT_SURF_ELLIPSE) hit = intersect_ellipse<FILT>(r, s.cy, s.cz, s.radius, s.radius_b, t);
        else hit = intersect_quadric<FILT>(r, s.cx, s.cy, s.cz, s.radius, cyl, t);
        nis++;
        if (!hit) return (s.flags & ORT_F_MISS_IS_HELP3) ? ORT_ST_HELP3 : lost;
        r.pos = vadd(r.pos, vscale(r.dir, t));
        if (s.aperture >= 0. && outside_aperture<FILT>(r.pos.x, r.pos.y, s.aperture)) return lost;
        // normal = centre - pos, with orig%x = centre%x for the bottle (lens.f90:288-290)
        N = {cyl ? 0.0 : s.cx - r.pos.x, s.cy - r.pos.y, s.cz - r.pos.z};
        N = vnormalise(N);
    } else {
        // plane kinds: d = (z_plane - pos%z) / dir%z ; pos = pos + dir*d
        double d = (s.cz - r.pos.z) / r.dir.z;
        Vec moved = vadd(r.pos, vscale(r.dir, d));
        nis++;
        if (kind == ORT_SURF_IMAGE) {
            r.pos = moved;
            return make_image<FILT>(S, r, xp, yp);
        }
        if (s.aperture >= 0. && outside_aperture<FILT>(moved.x, moved.y, s.aperture)) {
            r.pos = moved;
            return lost;
        }
        if (kind == ORT_SURF_IRIS) return -1;          // pos = origpos (lens.f90:564, :643)
        r.pos = moved;
        N = {0., 0., -1.};                             // flatNormal, lens.f90:165
    }
    bool reflected = reflect_refract<FILT>(r.dir, N, s.n1, s.n2, s.eta, draws);
    if (reflected && (s.flags & ORT_F_SKIP_ON_REFLECT)) return lost;
    return -1;
}

}  // namespace ort
