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
// solveQuadratic, src/surfaces.f90:227-260 + root choice :75-86
// ----------------------------------------------------------------------------
__device__ inline bool solve_and_pick(double a, double b, double c, double &t)
{
    double discrim = b * b - 4.0 * a * c;
    if (discrim < 0.0) return false;
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
    return solve_and_pick(a, b, c, t);
}

// intersect_ellipse, src/surfaces.f90:133-176
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
    return solve_and_pick(a, b, c, t);
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
template <class D>
__device__ inline bool reflect_refract(Vec &I, Vec N, double n1, double n2, double eta, D &draws)
{
    double u = draws.next();
    if (u <= fresnel(I, N, n1, n2, eta)) {
        double s = 2. * vdot(N, I);
        I = vsub(I, vscale(N, s));
        return true;
    }
    double c1 = vdot(N, I);
    Vec Nt = N;
    if (c1 < 0.) c1 = -c1;
    else Nt = vscale(N, -1.);
    double c2 = sqrt(1.0 - eta * eta * (1.0 - c1 * c1));
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
__device__ inline int make_image(const ort_system &S, const Ray &r, int &xp, int &yp)
{
    Vec d = vnormalise(r.dir);
    d = vscale(d, -1.);
    double top = (0. * d.x) + (0. * d.y) + (-1. * d.z);
    double bottom = sqrt(vdot(d, d)) * 1.0;
    double x = top / bottom;
    if (x < S.na_cos_min) return ORT_ST_NA_REJECT;
    if (r.pos.x > 1000 || r.pos.y > 1000) return ORT_ST_OFF_GRID;
    double fx = floor(r.pos.x / S.bin_width);
    double fy = floor(r.pos.y / S.bin_width);
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
template <class D>
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
        if (kind == ORT_SURF_ELLIPSE) hit = intersect_ellipse(r, s.cy, s.cz, s.radius, s.radius_b, t);
        else hit = intersect_quadric(r, s.cx, s.cy, s.cz, s.radius, cyl, t);
        nis++;
        if (!hit) return (s.flags & ORT_F_MISS_IS_HELP3) ? ORT_ST_HELP3 : lost;
        r.pos = vadd(r.pos, vscale(r.dir, t));
        if (s.aperture >= 0.) {
            double rad = sqrt(r.pos.x * r.pos.x + r.pos.y * r.pos.y);
            if (rad > s.aperture) return lost;
        }
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
            return make_image(S, r, xp, yp);
        }
        if (s.aperture >= 0.) {
            double rad = sqrt(moved.x * moved.x + moved.y * moved.y);
            if (rad > s.aperture) { r.pos = moved; return lost; }
        }
        if (kind == ORT_SURF_IRIS) return -1;          // pos = origpos (lens.f90:564, :643)
        r.pos = moved;
        N = {0., 0., -1.};                             // flatNormal, lens.f90:165
    }
    bool reflected = reflect_refract(r.dir, N, s.n1, s.n2, s.eta, draws);
    if (reflected && (s.flags & ORT_F_SKIP_ON_REFLECT)) return lost;
    return -1;
}

}  // namespace ort
