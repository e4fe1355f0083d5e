/* TEST INFRASTRUCTURE — CPU oracle (plain C restatement) of the per-ray hot path
 * of lewisfish/OpticalRayTrace.  See ort_oracle.h for its status and who may use
 * it.  Every function cites the reference file:line it follows.  Operation order
 * is kept exactly as written in the Fortran (build with -ffp-contract=off): all
 * `real` are fp64 there (reference src/Makefile:2, -freal-4-real-8).
 */
#define _GNU_SOURCE
#include "ort_oracle.h"
#include <math.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#ifdef ORC_PINNED_LIBM
/* libort_oracle_pinned.so (oracle/Makefile `pinned`): the six libm entries of the trace path come from pinned_libm.cpp —
 * csrc/ort_libm.h, glibc 2.35's algorithms restated and pinned to committed known answers, compiled for the host —
 * instead of the machine's libm, for a machine whose libm is another one (oracle/binding.py decides; the default build
 * calls the host's libm, which is what the reference compiled here calls).  The set-up constants' atan / tan / asin
 * (src/sourceMod.f90:177-180, src/imageMod.f90:41) stay the host's in either build: the product's host side forms the
 * same constants with the same library. */
double ortp_sin(double), ortp_cos(double), ortp_log(double), ortp_atan2(double, double), ortp_acos(double);
void ortp_sincos(double, double *, double *);
#define sin ortp_sin
#define cos ortp_cos
#define sincos ortp_sincos
#define log ortp_log
#define atan2 ortp_atan2
#define acos ortp_acos
int orc_pinned_libm(void) { return 1; }
#else
int orc_pinned_libm(void) { return 0; }
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- RNG ---- */
/* Semantics of ran2(): one U[0,1) per call (src/random_mod.f90:39-46).  The
 * generator itself is the compiler runtime's and is replaced by ORT-RNG-v2:
 *   c = (ray<<24) + k;  h = mix64(base + GOLDEN*((c>>1) + 1)),  base = mix64(seed ^ (GOLDEN*phase));
 *   u = (k even ? h >> 32 : h & 0xffffffff) * 2^-32      (one hash serves two consecutive draws) */
#define GOLDEN 0x9E3779B97F4A7C15ull

static inline uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

/* ORT-RNG-v2w (the product's kernel variant bit 5): one hash per draw, u = (h >> 11) * 2^-53 — 53 random bits as the
 * runtime's random_number gives ran2's real(8) (src/random_mod.f90:44).  Process-wide switch of the keyed mode. */
static int g_wide_draws = 0;
void orc_set_wide_draws(int32_t on) { g_wide_draws = on != 0; }

double orc_uniform(uint64_t seed, int32_t phase, uint64_t ray, int32_t draw)
{
    uint64_t base = mix64(seed ^ (GOLDEN * (uint64_t)phase));
    uint64_t c = (ray << 24) + (uint64_t)draw;
    if (g_wide_draws) return (double)(mix64(base + GOLDEN * (c + 1ull)) >> 11) * 0x1.0p-53;
    uint64_t h = mix64(base + GOLDEN * ((c >> 1) + 1ull));
    uint32_t w = (c & 1ull) ? (uint32_t)h : (uint32_t)(h >> 32);
    return (double)w * 0x1.0p-32;
}

typedef struct {
    const double *table;   /* NULL => keyed */
    int64_t stride;
    int32_t len;
    uint64_t seed, ray;
    int32_t phase, k;
} draws_t;

static inline double ran2(draws_t *d)
{
    int32_t k = d->k++;
    if (d->table) return k < d->len ? d->table[(int64_t)k * d->stride] : 0.5;
    return orc_uniform(d->seed, d->phase, d->ray, k);
}

/* ranu, src/random_mod.f90:48-57 */
static inline double ranu(draws_t *d, double a, double b) { return a + ran2(d) * (b - a); }

/* rang, src/random_mod.f90:59-85: polar Box-Muller, a variable number of draws */
static void rang(draws_t *d, double *x, double *y, double avg, double sigma)
{
    double s = 1.;
    while (s >= 1.) {
        *x = ranu(d, -1., 1.);
        *y = ranu(d, -1., 1.);
        s = (*y) * (*y) + (*x) * (*x);
    }
    double cst = sqrt(-2. * log(s) / s);
    double tmp = (*x) * cst;
    *x = avg + sigma * tmp;
    tmp = (*y) * cst;
    *y = avg + sigma * tmp;
}

static const double PI_F = 3.14159265358979323846;   /* 4.*atan(1.), src/constants.f90:5 */
/* WHICH libm entry a sin/cos pair goes through decides its last bit: glibc 2.35's sincos() is another build of the
 * algorithm than its sin() and cos() (no fused multiply-adds, another split of the middle range), and
 * sincos(x) != (sin(x), cos(x)) for ~0.1 % of x.  The reference's compiled code (flang -O2) merges every sin/cos pair
 * of one argument into ONE sincos() call except the first pair of `ring` — sourceMod.o imports sin, cos, sincos;
 * stokes.o imports atan2, acos, sincos only — and gcc -O2 happened to do the same to the plain C that used to stand
 * here.  So that no optimiser's mood decides a bit, every site says which entry it means (tests/test_oracle_vs_ref.py
 * holds the oracle against the compiled reference bit for bit; the device follows the same table, ort_device.h). */
static double (*volatile libm_sin)(double) = sin;
static double (*volatile libm_cos)(double) = cos;
#define SIN_COS_TWO_CALLS(x, s, c) do { *(c) = libm_cos(x); *(s) = libm_sin(x); } while (0)
#define SINCOS_ONE_CALL(x, s, c) sincos(x, s, c)
#define ISORS_SINCOS(x, s, c) SINCOS_ONE_CALL(x, s, c)          /* src/sourceMod.f90:229-230 */

/* ------------------------------------------------------- vector_class ---- */
static inline orc_vec v(double x, double y, double z) { orc_vec r = {x, y, z}; return r; }
/* vec_minus_vec :48-57, vec_add_vec :84-93 */
static inline orc_vec vsub(orc_vec a, orc_vec b) { return v(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline orc_vec vadd(orc_vec a, orc_vec b) { return v(a.x + b.x, a.y + b.y, a.z + b.z); }
/* vec_mult_scal / scal_mult_vec :139-160 */
static inline orc_vec vscale(orc_vec a, double s) { return v(a.x * s, a.y * s, a.z * s); }
/* vec_dot :96-106 */
static inline double vdot(orc_vec a, orc_vec b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
/* magnitude_fn :175-186 — returns the NORMALISED vector, three divisions */
static inline orc_vec vmagnitude(orc_vec a)
{
    double tmp = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return v(a.x / tmp, a.y / tmp, a.z / tmp);
}

/* ----------------------------------------------------------- surfaces ---- */
/* solveQuadratic, src/surfaces.f90:227-260 */
static int solveQuadratic(double a, double b, double c, double *x0, double *x1)
{
    double discrim = b * b - 4.0 * a * c;
    if (discrim < 0.0) return 0;
    else if (discrim == 0.0) {
        *x0 = -0.5 * b / a;
        *x1 = *x0;
    } else {
        double q;
        if (b > 0.0) q = -0.5 * (b + sqrt(discrim));
        else         q = -0.5 * (b - sqrt(discrim));
        *x0 = q / a;
        *x1 = c / q;
    }
    return 1;
}

/* root choice shared by every intersect_*, src/surfaces.f90:75-86 */
static int pick_root(double t0, double t1, double *t)
{
    if (t0 > t1) { double tmp = t1; t1 = t0; t0 = tmp; }
    if (t0 < 0.0) {
        t0 = t1;
        if (t0 < 0.0) return 0;
    }
    *t = t0;
    return 1;
}

/* intersect_sphere, src/surfaces.f90:52-89 */
static int intersect_sphere(orc_vec orig, orc_vec dir, double *t, orc_vec centre, double radius)
{
    orc_vec L = vsub(orig, centre);
    double a = vdot(dir, dir);
    double b = 2.0 * vdot(dir, L);
    double c = vdot(L, L) - radius * radius;
    double t0, t1;
    if (!solveQuadratic(a, b, c, &t0, &t1)) return 0;
    return pick_root(t0, t1, t);
}

/* intersect_cylinder, src/surfaces.f90:91-130 — axis along x, uses y,z only */
static int intersect_cylinder(orc_vec orig, orc_vec dir, double *t, orc_vec centre, double radius)
{
    orc_vec L = vsub(orig, centre);
    double a = dir.z * dir.z + dir.y * dir.y;
    double b = 2 * (dir.z * L.z + dir.y * L.y);
    double c = L.z * L.z + L.y * L.y - radius * radius;
    double t0, t1;
    if (!solveQuadratic(a, b, c, &t0, &t1)) return 0;
    return pick_root(t0, t1, t);
}

/* intersect_ellipse, src/surfaces.f90:133-176 */
static int intersect_ellipse(orc_vec orig, orc_vec dir, double *t, orc_vec centre,
                             double semia, double semib)
{
    double semia2div = 1. / (semia * semia);
    double semib2div = 1. / (semib * semib);
    orc_vec L = vsub(orig, centre);
    double a = semia2div * (dir.z * dir.z) + semib2div * (dir.y * dir.y);
    double b = 2 * (semia2div * dir.z * L.z + semib2div * dir.y * L.y);
    double c = semia2div * (L.z * L.z) + semib2div * (L.y * L.y) - 1;
    double t0, t1;
    if (!solveQuadratic(a, b, c, &t0, &t1)) return 0;
    return pick_root(t0, t1, t);
}

/* intersect_cone, src/surfaces.f90:179-224 — apex up, axis along z, `centre` under the apex at the base */
static int intersect_cone(orc_vec orig, orc_vec dir, double *t, orc_vec centre, double radius, double height)
{
    double k = radius / height;
    k = k * k;
    orc_vec L = vsub(orig, centre);
    double a = dir.x * dir.x + dir.y * dir.y - (k * (dir.z * dir.z));
    double b = 2. * ((dir.x * L.x) + (dir.y * L.y) - (k * dir.z * (L.z - height)));
    double c = L.x * L.x + L.y * L.y - (k * ((L.z - height) * (L.z - height)));
    double t0, t1;
    if (!solveQuadratic(a, b, c, &t0, &t1)) return 0;
    return pick_root(t0, t1, t);
}

/* fresnel, src/surfaces.f90:336-372 */
static double fresnel(orc_vec I, orc_vec N, double n1, double n2)
{
    double costt = fabs(vdot(I, N));
    double sintt = sqrt(1. - costt * costt);
    double sint2 = n1 / n2 * sintt;
    double tir;
    if (sint2 > 1.) return 1.0;
    else if (costt == 1.) return 0.;
    else {
        sint2 = (n1 / n2) * sintt;
        double cost2 = sqrt(1. - sint2 * sint2);
        double r1 = fabs((n1 * costt - n2 * cost2) / (n1 * costt + n2 * cost2));
        double r2 = fabs((n1 * cost2 - n2 * costt) / (n1 * cost2 + n2 * costt));
        double f1 = r1 * r1;
        double f2 = r2 * r2;
        tir = 0.5 * (f1 + f2);
        if (isnan(tir) || tir > 1. || tir < 0.) tir = 1.;
        return tir;
    }
}

/* reflect, src/surfaces.f90:285-300 */
static orc_vec reflect(orc_vec I, orc_vec N)
{
    double s = 2. * vdot(N, I);
    return vsub(I, vscale(N, s));
}

/* refract, src/surfaces.f90:303-333 */
static orc_vec refract(orc_vec I, orc_vec N, double eta)
{
    orc_vec Ntmp = N;
    double c1 = vdot(Ntmp, I);
    if (c1 < 0.) c1 = -c1;
    else Ntmp = vscale(N, -1.);
    double c2 = sqrt(1.0 - eta * eta * (1.0 - c1 * c1));
    return vadd(vscale(I, eta), vscale(Ntmp, eta * c1 - c2));
}

/* reflect_refract, src/surfaces.f90:262-282 — consumes exactly one draw */
static void reflect_refract(orc_vec *I, orc_vec N, double n1, double n2, int *rflag, draws_t *d)
{
    *rflag = 0;
    if (ran2(d) <= fresnel(*I, N, n1, n2)) {
        *I = reflect(*I, N);
        *rflag = 1;
    } else {
        *I = refract(*I, N, n1 / n2);
    }
}

/* --------------------------------------------------------------- lens ---- */
/* Sellmeier / cauchy / dispersion, src/lens.f90:647-695 */
double orc_sellmeier(double wave, double b1, double b2, double b3, double c1, double c2, double c3)
{
    double w = wave * 1e6;
    double wave2 = w * w;
    double a = (b1 * wave2) / (wave2 - c1);
    double b = (b2 * wave2) / (wave2 - c2);
    double c = (b3 * wave2) / (wave2 - c3);
    return sqrt(1.0 + (a + b + c));
}

double orc_cauchy(double wave, double a, double b, double c)
{
    double w = wave * 1e6;
    /* wavetmp**(-2), **(-4): integer powers, evaluated as reciprocals of products */
    return a + b * (1.0 / (w * w)) + c * (1.0 / ((w * w) * (w * w)));
}

double orc_dispersion(double wave, double a, double b, double c)
{
    double w = wave * 1e6;
    double wave2 = w * w;
    return a - b * wave2 + (c / wave2);
}

/* libm entries of the scattering walk (tauint, stokes).  Development build -DORC_PERTURB
 * (tools/scatter_sensitivity.py): each can be made to return its value moved by one ulp in half of
 * the calls, which is what another libm (the GPU's) does to it — to measure which function's last
 * bit the walk amplifies.  The default build calls libm directly. */
#ifdef ORC_PERTURB
int orc_perturb_mask = 0;       /* 1 log, 2 atan2, 4 acos, 8 sin/cos of ri1/ri3, 16 sin/cos of phi */
static double nudge(double y, int bit)
{
    if (!(orc_perturb_mask & bit)) return y;
    uint64_t b; memcpy(&b, &y, 8);
    uint64_t h = mix64(b);
    if (h & 1) return y;
    return nextafter(y, (h & 2) ? INFINITY : -INFINITY);
}
#define M_LOG(x) nudge(log(x), 1)
#define M_ATAN2(y, x) nudge(atan2(y, x), 2)
#define M_ACOS(x) nudge(acos(x), 4)
#define M_SINCOS1(x, s, c) do { sincos(x, s, c); *(s) = nudge(*(s), 8); *(c) = nudge(*(c), 8); } while (0)
#define M_SINCOS2(x, s, c) do { sincos(x, s, c); *(s) = nudge(*(s), 16); *(c) = nudge(*(c), 16); } while (0)
#else
#define M_LOG(x) log(x)
#define M_ATAN2(y, x) atan2(y, x)
#define M_ACOS(x) acos(x)
#define M_SINCOS1(x, s, c) SINCOS_ONE_CALL(x, s, c)
#define M_SINCOS2(x, s, c) SINCOS_ONE_CALL(x, s, c)
#endif

/* tauint, src/surfaces.f90:13-50: optical depth to the next event inside the cylinder.
 * Returns 0 ok (dist = distance to the event, *tflag = 1 when that is the cylinder wall),
 * 1 for the reference's `error stop "no intersection"` (:33-39). */
static int tauint(orc_vec pos, orc_vec dir, double mua, double mus, orc_vec centre, double radius,
                  double *dist, int *tflag, draws_t *d, int *nis)
{
    double mu_tot = mua + mus;
    double tau = -M_LOG(ran2(d));
    *tflag = 0;
    int flag = intersect_cylinder(pos, dir, dist, centre, radius);
    (*nis)++;
    if (!flag) return 1;
    double tauradius = *dist * mu_tot;
    if (tau < tauradius) *dist = tau / mu_tot;
    else *tflag = 1;
    return 0;
}

/* stokes, src/stokes.f90:7-166: Henyey-Greenstein direction update (hgg /= 0 at both call
 * sites, src/lens.f90:269,320; the isotropic branch :31-48 is restated for completeness) */
static void stokes(orc_vec *dir, double hgg, draws_t *d)
{
    const double TWOPI = 2. * PI_F, PI = PI_F;
    double nxp = dir->x, nyp = dir->y, nzp = dir->z;
    double cost = dir->z;
    double sint = sqrt(1. - cost * cost);
    double g2 = hgg * hgg;
    double phi = M_ATAN2(dir->y, dir->x);
    double cosp, sinp;
    if (hgg == 0.0) {
        cost = 2. * ran2(d) - 1.;
        sint = (1. - cost * cost);
        if (sint <= 0.) sint = 0.; else sint = sqrt(sint);
        phi = TWOPI * ran2(d);
        SINCOS_ONE_CALL(phi, &sinp, &cosp);
        nxp = sint * cosp; nyp = sint * sinp; nzp = cost;
    } else {
        double costp = cost, sintp = sint, phip = phi;
        double w = (1. - g2) / (1. - hgg + 2. * hgg * ran2(d));
        double bmu = ((1. + g2) - w * w) / (2. * hgg);
        double cosb2 = bmu * bmu;
        if (fabs(bmu) > 1.) {
            if (bmu > 1.) { bmu = 1.; cosb2 = 1.; }
            else { bmu = -1.; cosb2 = 1.; }
        }
        double sinbt = sqrt(1. - cosb2);
        double ri1 = TWOPI * ran2(d);
        double cosi2 = 0., sini2 = 0., cosdph;
        if (ri1 > PI) {
            double ri3 = TWOPI - ri1;
            double cosi3, sini3;
            M_SINCOS1(ri3, &sini3, &cosi3);
            if (bmu == 1. || bmu == -1.) goto done;
            cost = costp * bmu + sintp * sinbt * cosi3;
            if (fabs(cost) < 1.) {
                sint = fabs(sqrt(1. - cost * cost));
                sini2 = sini3 * sintp / sint;
                double bott = sint * sinbt;
                cosi2 = costp / bott - cost * bmu / bott;
            } else {
                sint = 0.; sini2 = 0.;
                if (cost >= 1.) cosi2 = -1.;
                if (cost <= -1.) cosi2 = 1.;
            }
            cosdph = -cosi2 * cosi3 + sini2 * sini3 * bmu;
            if (fabs(cosdph) > 1.) cosdph = (cosdph > 1.) ? 1. : -1.;
            phi = phip + M_ACOS(cosdph);
            if (phi > TWOPI) phi = phi - TWOPI;
            if (phi < 0.) phi = phi + TWOPI;
        } else {
            double cosi1, sini1;
            M_SINCOS1(ri1, &sini1, &cosi1);
            if (bmu == 1. || bmu == -1.) goto done;
            cost = costp * bmu + sintp * sinbt * cosi1;
            if (fabs(cost) < 1.) {
                sint = fabs(sqrt(1. - cost * cost));
                sini2 = sini1 * sintp / sint;
                double bott = sint * sinbt;
                cosi2 = costp / bott - cost * bmu / bott;
            } else {
                sint = 0.; sini2 = 0.;
                if (cost >= 1.) cosi2 = -1.;
                if (cost <= -1.) cosi2 = 1.;
            }
            cosdph = -cosi1 * cosi2 + sini1 * sini2 * bmu;
            if (fabs(cosdph) > 1.) cosdph = (cosdph > 1.) ? 1. : -1.;
            phi = phip - M_ACOS(cosdph);
            if (phi > TWOPI) phi = phi - TWOPI;
            if (phi < 0.) phi = phi + TWOPI;
        }
        M_SINCOS2(phi, &sinp, &cosp);
        nxp = sint * cosp; nyp = sint * sinp; nzp = cost;
    }
done:
    *dir = v(nxp, nyp, nzp);
}

/* the scattering walk of src/lens.f90:262-282 (contents) / :312-333 (wall): t enters as the
 * distance to the wall and leaves as the distance of the last leg.  Returns 0 go on, 1 skip
 * (absorbed, or heading back), 2 "no intersection". */
static int scatter_walk(orc_vec *pos, orc_vec *dir, double *t, double mua, double mus, double hgg,
                        orc_vec centre, double radius, draws_t *d, int *nis)
{
    int tflag = 0;
    if (tauint(*pos, *dir, mua, mus, centre, radius, t, &tflag, d, nis)) return 2;
    while (!tflag) {
        *pos = vadd(*pos, vscale(*dir, *t));
        if (ran2(d) < mus / (mus + mua)) stokes(dir, hgg, d);
        else return 1;                                             /* absorbed */
        if (tauint(*pos, *dir, mua, mus, centre, radius, t, &tflag, d, nis)) return 2;
        if (sqrt(pos->x * pos->x + pos->z * pos->z) >= radius) break;   /* sic: x and z (:276, :327) */
    }
    if (dir->z < 0.) return 1;
    return 0;
}

/* bottle_forward_sub, src/lens.f90:230-350.  Returns 0 ok, 1 skip, 2 "no intersection". */
static int bottle_forward(const orc_bottle *B, orc_vec *pos, orc_vec *dir, draws_t *d, int *nis)
{
    double t;
    int flag;
    orc_vec orig, normal;

    if (B->ellipse) {
        double rad1 = B->radiusa - B->thickness;
        double rad2 = B->radiusb - B->thickness;
        flag = intersect_ellipse(*pos, *dir, &t, B->centre, rad1, rad2);
    } else {
        flag = intersect_cylinder(*pos, *dir, &t, B->centre, B->radiusa - B->thickness);
    }
    (*nis)++;
    if (!flag) return 1;

    if (B->mua_c + B->mus_c != 0.0) {                                   /* scatter_c, :218-219, :262-282 */
        int rc = scatter_walk(pos, dir, &t, B->mua_c, B->mus_c, .65, B->centre,
                              B->radiusa - B->thickness, d, nis);
        if (rc) return rc;
    }
    *pos = vadd(*pos, vscale(*dir, t));
    orig = *pos;
    orig.x = B->centre.x;
    normal = vsub(B->centre, orig);
    normal = vmagnitude(normal);
    reflect_refract(dir, normal, B->ncontents, B->nbottle, &flag, d);
    if (flag) return 1;

    if (B->ellipse) flag = intersect_ellipse(*pos, *dir, &t, B->centre, B->radiusa / 2., B->radiusb / 2.);
    else            flag = intersect_cylinder(*pos, *dir, &t, B->centre, B->radiusa);
    (*nis)++;
    if (!flag) return 1;

    if (B->mua_b + B->mus_b != 0.0) {                                   /* scatter_b, :312-333 */
        int rc = scatter_walk(pos, dir, &t, B->mua_b, B->mus_b, 0.9, B->centre, B->radiusa, d, nis);
        if (rc) return rc;
    }
    *pos = vadd(*pos, vscale(*dir, t));
    orig = *pos;
    orig.x = B->centre.x;
    normal = vsub(B->centre, orig);
    normal = vmagnitude(normal);
    reflect_refract(dir, normal, B->nbottle, 1.0, &flag, d);
    if (flag) return 1;
    return 0;
}

/* plano_forward_sub, src/lens.f90:425-481 */
static int plano_forward(const orc_plano *P, orc_vec *pos, orc_vec *dir, draws_t *d, int *nis)
{
    double a = P->centre.z + P->curve_radius - P->thickness;
    double dd = (a - pos->z) / dir->z;
    *pos = vadd(*pos, vscale(*dir, dd));
    (*nis)++;
    double r = sqrt(pos->x * pos->x + pos->y * pos->y);
    if (r > P->radius) return 1;

    int flag;
    reflect_refract(dir, P->flatNormal, P->n1, P->n2, &flag, d);   /* flag ignored, :458-459 */

    double t;
    flag = intersect_sphere(*pos, *dir, &t, P->centre, P->curve_radius);
    (*nis)++;
    if (!flag) return 1;
    *pos = vadd(*pos, vscale(*dir, t));

    orc_vec curvedNormal = vsub(P->centre, *pos);
    curvedNormal = vmagnitude(curvedNormal);
    reflect_refract(dir, curvedNormal, P->n2, P->n1, &flag, d);
    if (flag) return 1;
    return 0;
}

/* doublet_forward_sub, src/lens.f90:531-645.  Returns 0 ok, 1 skip, 2 "Help3". */
static int doublet_forward(const orc_doublet *D, orc_vec *pos, orc_vec *dir, draws_t *d, int *nis,
                           int iris_before, int iris_after, double iris_radius)
{
    orc_vec normal, origpos;
    int flag;
    double t, r;

    if (iris_before) {
        origpos = *pos;
        t = ((D->centre1.z - D->R1) - pos->z) / dir->z;
        *pos = vadd(*pos, vscale(*dir, t));
        (*nis)++;
        r = sqrt(pos->x * pos->x + pos->y * pos->y);
        if (r > D->radius * iris_radius) return 1;
        *pos = origpos;
    }

    flag = intersect_sphere(*pos, *dir, &t, D->centre1, D->R1);
    (*nis)++;
    if (!flag) return 1;
    *pos = vadd(*pos, vscale(*dir, t));
    r = sqrt(pos->x * pos->x + pos->y * pos->y);
    if (r > (D->radius * 1.0)) return 1;

    normal = vsub(*pos, D->centre1);
    normal = vmagnitude(normal);
    reflect_refract(dir, normal, D->n1, D->n2, &flag, d);
    if (flag) return 1;

    flag = intersect_sphere(*pos, *dir, &t, D->centre2, D->R2);
    (*nis)++;
    if (!flag) return 1;
    *pos = vadd(*pos, vscale(*dir, t));
    normal = vsub(D->centre2, *pos);
    normal = vmagnitude(normal);
    reflect_refract(dir, normal, D->n2, D->n3, &flag, d);
    if (flag) return 1;

    flag = intersect_sphere(*pos, *dir, &t, D->centre3, D->R3);
    (*nis)++;
    if (!flag) return 2;                                  /* error stop "Help3", :617 */
    *pos = vadd(*pos, vscale(*dir, t));
    normal = vsub(D->centre3, *pos);
    normal = vmagnitude(normal);
    reflect_refract(dir, normal, D->n3, D->n1, &flag, d);
    if (flag) return 1;

    if (iris_after) {
        origpos = *pos;
        t = ((D->centre3.z + D->R3) - pos->z) / dir->z;
        *pos = vadd(*pos, vscale(*dir, t));
        (*nis)++;
        r = sqrt(pos->x * pos->x + pos->y * pos->y);
        if (r > D->radius * iris_radius) return 1;
        *pos = origpos;
    }
    return 0;
}

/* telescope, src/optics_system.f90:6-52.  Returns 0 ok, 1 skip, 2 Help3. */
static int telescope(const orc_system *S, int ph, orc_vec *pos, orc_vec *dir, draws_t *d, int *nis)
{
    int rc = plano_forward(&S->L2[ph], pos, dir, d, nis);
    if (rc) return rc;
    rc = doublet_forward(&S->L3[ph], pos, dir, d, nis, S->iris_before, S->iris_after, S->iris_radius);
    if (rc) return rc;
    double dd = ((S->img_plane + S->fibre_offset) - pos->z) / dir->z;
    *pos = vadd(*pos, vscale(*dir, dd));
    (*nis)++;
    return 0;
}

/* ------------------------------------------------------------ sources ---- */

/* point, src/sourceMod.f90:12-47 (offset absent at the call site main.f90:136, bottle%centre%z
 * at :140 for the isors source) */
static void emit_point(double cosThetaMax, double offset, orc_vec *pos, orc_vec *dir, draws_t *d)
{
    const double twopi = 2. * PI_F;
    double phi = twopi * ran2(d);
    double cosp, sinp;
    SINCOS_ONE_CALL(phi, &sinp, &cosp);
    double ran = ran2(d);
    double cost = (1.0 - ran) + ran * cosThetaMax;
    double sint = sqrt(1.0 - cost * cost);
    *dir = v(sint * cosp, sint * sinp, cost);
    *pos = v(0.0, 0.0, 0.0 + offset);
}

/* create_spot, src/sourceMod.f90:122-159: deterministic fan of rays, no draws */
static void emit_spot(double cosThetaMax, int nrays, int n, orc_vec *pos, orc_vec *dir)
{
    const double twopi = 2. * PI_F;
    double nrays_sqrt = sqrt((double)nrays);
    double thetaMax = acos(cosThetaMax);
    double deltaPhi = twopi / nrays_sqrt;
    double deltaTheta = thetaMax / nrays_sqrt;
    double phi = deltaPhi * (double)(n % 10);
    double theta = deltaTheta * (double)(n / 10);
    double sinp, cosp;
    SINCOS_ONE_CALL(phi, &sinp, &cosp);
    double cost = libm_cos(theta);                      /* cos alone: the cos() entry */
    double sint = sqrt(1. - cost * cost);
    *dir = v(sint * cosp, sint * sinp, cost);
    *pos = v(0., 0., 0.);
}

/* point_on_bottle, src/sourceMod.f90:50-89 (the "crs" source of phase 1) */
static void emit_crs(const orc_system *S, orc_vec *pos, orc_vec *dir, draws_t *d)
{
    const double twopi = 2. * PI_F;
    double phi = twopi * ran2(d);
    double cosp, sinp;
    SINCOS_ONE_CALL(phi, &sinp, &cosp);
    double ran = ran2(d);
    double cost = (1.0 - ran) + ran * S->cosThetaMax;
    double sint = sqrt(1.0 - cost * cost);
    double tmp1, tmp2, t = 0.;
    rang(d, &tmp1, &tmp2, 0., S->spot_size);
    *pos = v(tmp1, tmp2, 1.0);
    *dir = v(0., 0., -1.);
    (void)intersect_cylinder(*pos, *dir, &t, S->bottle.centre, S->bottle.radiusa + S->bottle.thickness);
    *pos = vadd(*pos, vscale(*dir, t));
    *dir = v(sint * cosp, sint * sinp, cost);
}

/* bottle_backward_sub, src/lens.f90:352-423: air -> glass -> contents, from outside.  Reached only
 * from iSORS(ring = .false.), which no call site of the reference uses (src/main.f90:97 passes
 * .true., :141 is commented out); restated and pinned for completeness.  Returns skip. */
static int bottle_backward(const orc_bottle *B, orc_vec *pos, orc_vec *dir, draws_t *d)
{
    double t;
    int flag;
    orc_vec orig, normal;
    if (B->ellipse) flag = intersect_ellipse(*pos, *dir, &t, B->centre, B->radiusa, B->radiusb);
    else flag = intersect_cylinder(*pos, *dir, &t, B->centre, B->radiusa);
    if (!flag) return 1;
    *pos = vadd(*pos, vscale(*dir, t));
    orig = *pos;
    orig.x = B->centre.x;
    normal = vmagnitude(vsub(orig, B->centre));
    reflect_refract(dir, normal, 1., B->nbottle, &flag, d);
    if (flag) return 1;
    if (B->ellipse) flag = intersect_ellipse(*pos, *dir, &t, B->centre, B->radiusa - B->thickness, B->radiusb - B->thickness);
    else flag = intersect_cylinder(*pos, *dir, &t, B->centre, B->radiusa - B->thickness);
    if (!flag) return 1;
    *pos = vadd(*pos, vscale(*dir, t));
    orig = *pos;
    orig.x = B->centre.x;
    normal = vmagnitude(vsub(orig, B->centre));
    reflect_refract(dir, normal, B->nbottle, B->ncontents, &flag, d);
    return flag ? 1 : 0;
}

/* iSORS, src/sourceMod.f90:162-247: Gaussian beam through an axicon (a cone of glass, n = 1.4),
 * carried to the bottle, then aimed at a random point of the lens.  ring = 1 is what src/main.f90:97
 * calls.  Returns 0 where the reference aborts — `error stop "no intersection with bottle!"`
 * (:216-218): the beam reflected at the axicon (2.8 % of the rays) and flies away from the bottle —
 * having drawn nothing further; the caller counts the ray as lost (ORC_NO_INTERSECTION). */
static int emit_isors(const orc_system *S, orc_vec *pos, orc_vec *dir, draws_t *d, int ring)
{
    const double twopi = 2. * PI_F;
    const orc_bottle *B = &S->bottle;
    const orc_plano *L1 = &S->L2[0];
    const double axicon_n = 1.4, radius = 12.7e-3, height = 1.1e-3;
    double alpha = atan(height / radius);
    double k = (radius / height) * (radius / height);
    double base_pos = (S->isors_offset + S->ring_width) / tan(alpha * (axicon_n - 1.));
    orc_vec centre = v(0., 0., 0.);
    double posx, posy, t;
    int flag;
    rang(d, &posx, &posy, 0., S->ring_width);
    *pos = vadd(centre, v(posx, posy, 2 * height));
    *dir = v(0., 0., -1.);
    flag = intersect_cone(*pos, *dir, &t, centre, radius, height);
    if (flag) {
        *pos = vadd(*pos, vscale(*dir, t));
        orc_vec normal = v(2 * (pos->x - centre.x) / k, 2 * (pos->y - centre.y) / k,
                           -(2 * (pos->z - centre.z)) + 2 * height);
        normal = vscale(normal, -1.);
        normal = vmagnitude(normal);
        reflect_refract(dir, normal, axicon_n, 1., &flag, d);
        t = base_pos / dir->z;
        *pos = vadd(*pos, vscale(*dir, t));
        pos->z = B->radiusa + B->centre.z + 0x1p-52;          /* epsilon(1.) of a real*8 */
        if (ring) {
            if (B->ellipse) flag = intersect_ellipse(*pos, *dir, &t, B->centre, B->radiusa - B->thickness, B->radiusb - B->thickness);
            else flag = intersect_cylinder(*pos, *dir, &t, B->centre, B->radiusa - B->thickness);
            if (!flag) return 0;
            *pos = vadd(*pos, vscale(*dir, t));
        } else {
            (void)bottle_backward(B, pos, dir, d);            /* skip is not looked at */
            t = (B->centre.z - pos->z) / dir->z;
            *pos = vadd(*pos, vscale(*dir, t));
        }
    }
    double r = ring ? ranu(d, 0., L1->radius * L1->radius) : ranu(d, 0., (L1->radius + 10e-3) * (L1->radius + 10e-3));
    double theta = ran2(d) * twopi;
    double st, ct;
    ISORS_SINCOS(theta, &st, &ct);
    posx = sqrt(r) * ct;
    posy = sqrt(r) * st;
    orc_vec lp = v(posx, posy, L1->fb);
    double ex = lp.x - pos->x, ey = lp.y - pos->y, ez = lp.z - pos->z;
    double dist = sqrt(ex * ex + ey * ey + ez * ez);
    *dir = v((lp.x - pos->x) / dist, (lp.y - pos->y) / dist, (lp.z - pos->z) / dist);
    *dir = vmagnitude(*dir);
    return 1;
}

/* init_emit_image, src/sourceMod.f90:363-408 */
void orc_init_emit_image(const double *img, int32_t nphotons, uint64_t seed, int32_t *counts_scan)
{
    /* read(u) imgout fills imgout(a,b) = img[a + 512 b]; imgout = transpose(imgout) (:386) */
    double tot = 0.;
    for (int b = 0; b < 512; b++)                       /* sum(imgout): array element order */
        for (int a = 0; a < 512; a++) tot += img[(size_t)b + 512u * (size_t)a];
    draws_t d;
    memset(&d, 0, sizeof d);
    d.seed = seed; d.ray = 0; d.phase = 0; d.k = 0;
    for (int i = 0; i < 512; i++) {                     /* do i = 1, 512 ; do j = 1, 512 (:396-397) */
        for (int j = 0; j < 512; j++) {
            double pix = img[(size_t)j + 512u * (size_t)i];        /* transposed imgout(i,j) */
            double tmp = ((double)nphotons * pix) / tot;
            double diff = tmp - (double)(int)tmp;
            int c;
            if (ran2(&d) < diff && diff > 0) c = (int)tmp + 1;
            else c = (int)tmp;
            /* imgin(i,j); emit_image scans `do i2 ; do j2 : img(j2,i2)`: first index inner */
            counts_scan[(size_t)i + 512u * (size_t)j] = c;
        }
    }
}

/* emit, src/sourceMod.f90:325-361, for histogram cell (i = first index, j = second index), 1-based */
static void emit_cell(const orc_plano *lens, int i, int j, orc_vec *pos, orc_vec *dir, draws_t *d)
{
    const double twopi = 2. * PI_F;
    double dx = 5000e-6 / 512.;
    double x = ranu(d, ((double)i - 1.) * dx, (double)i * dx) - 2500e-6;
    double y = ranu(d, ((double)j - 1.) * dx, (double)j * dx) - 2500e-6;
    *pos = v(x, y, 0.0);
    double r = ranu(d, 0., lens->radius * lens->radius);
    double theta = ran2(d) * twopi;
    double st, ct;
    SINCOS_ONE_CALL(theta, &st, &ct);
    orc_vec lp = v(sqrt(r) * ct, sqrt(r) * st, lens->fb);
    double ex = lp.x - pos->x, ey = lp.y - pos->y, ez = lp.z - pos->z;
    double dist = sqrt(ex * ex + ey * ey + ez * ez);
    *dir = v((lp.x - pos->x) / dist, (lp.y - pos->y) / dist, (lp.z - pos->z) / dist);
    *dir = vmagnitude(*dir);
}

/* emit_image, src/sourceMod.f90:303-323, for the ray with serial index `iray`: the first cell in
 * scan order whose count is not yet used up = the cell whose cumulative count exceeds iray.
 * Returns 0 when the histogram is exhausted (the reference then re-uses stale pos/dir). */
static int emit_image_ray(const orc_system *S, const uint64_t *cdf, uint64_t iray, orc_vec *pos,
                          orc_vec *dir, draws_t *d)
{
    /* cdf[s] = rays emitted by cells 0..s-1; find the last s with cdf[s] <= iray < cdf[s+1] */
    if (iray >= cdf[512 * 512]) return 0;
    int lo = 0, hi = 512 * 512;                         /* cdf[lo] <= iray < cdf[hi] */
    while (hi - lo > 1) {
        int mid = (lo + hi) / 2;
        if (cdf[mid] <= iray) lo = mid; else hi = mid;
    }
    emit_cell(&S->L2[1], lo % 512 + 1, lo / 512 + 1, pos, dir, d);
    return 1;
}

static uint64_t *build_cdf(const orc_system *S)
{
    if (!S->img_counts) return NULL;
    uint64_t *cdf = malloc((512 * 512 + 1) * sizeof(uint64_t));
    cdf[0] = 0;
    for (int s = 0; s < 512 * 512; s++)
        cdf[s + 1] = cdf[s] + (uint64_t)(S->img_counts[s] > 0 ? S->img_counts[s] : 0);
    return cdf;
}

/* ring, src/sourceMod.f90:250-300 */
static void emit_ring(const orc_system *S, orc_vec *pos, orc_vec *dir, draws_t *d)
{
    const double twopi = 2. * PI_F;
    const orc_plano *lens = &S->L2[0];
    double Ra = S->bottle.radiusa, Rb = S->bottle.radiusb, off = S->bottle.centre.z;
    double r = ranu(d, S->r1, S->r2);
    double theta = ran2(d) * twopi;
    double ct1, st1;
    SIN_COS_TWO_CALLS(theta, &st1, &ct1);               /* :272-273: the one pair flang leaves as sin() and cos() */
    double posx = sqrt(r) * ct1;
    double posy = sqrt(r) * st1;
    double posz;
    if (S->bottle.ellipse) {
        double q = posy * Ra / Rb;
        posz = off + sqrt(Ra * Ra - q * q);
    } else {
        posz = off + sqrt(Ra * Ra - posy * posy);
    }
    *pos = v(posx, posy, posz);

    double rl = lens->radius + 10e-3;
    r = ranu(d, 0., rl * rl);
    theta = ran2(d) * twopi;
    double st, ct;
    SINCOS_ONE_CALL(theta, &st, &ct);                   /* :287-288 */
    posx = sqrt(r) * ct;
    posy = sqrt(r) * st;
    orc_vec lp = v(posx, posy, lens->fb);
    double ex = lp.x - pos->x, ey = lp.y - pos->y, ez = lp.z - pos->z;
    double dist = sqrt(ex * ex + ey * ey + ez * ez);
    *dir = v((lp.x - pos->x) / dist, (lp.y - pos->y) / dist, (lp.z - pos->z) / dist);
    *dir = vmagnitude(*dir);
}

/* -------------------------------------------------------------- image ---- */
/* makeImage2D, src/imageMod.f90:19-58.  Returns status, writes the bin. */
static int make_image(orc_vec dir, orc_vec pos, double diameter, int *xp_out, int *yp_out)
{
    orc_vec n = v(0., 0., -1.);
    n = vmagnitude(n);
    orc_vec d = vmagnitude(dir);
    d = vscale(d, -1.);
    double top = vdot(n, d);
    double bottom = sqrt(vdot(d, d)) * sqrt(vdot(n, n));
    double angle = acos(top / bottom);
    double na = asin(0.22);
    if (angle > na) return ORC_NA_REJECT;
    double binwid = diameter / 401.;
    if (pos.x > 1000 || pos.y > 1000) return ORC_OFF_GRID;
    double fx = floor(pos.x / binwid), fy = floor(pos.y / binwid);
    if (!(fabs(fx) <= 200.) || !(fabs(fy) <= 200.)) return ORC_OFF_GRID;
    *xp_out = (int)fx;
    *yp_out = (int)fy;
    return ORC_BINNED;
}

/* ---------------------------------------------------------- loop body ---- */
/* one iteration of src/main.f90:90-109 (phase 1) or :127-162 (phase 2) */
static int one_ray(const orc_system *S, int phase, int have_in, orc_vec *pos, orc_vec *dir,
                   draws_t *d, int *nis, int *xp, int *yp, orc_vec *epos, orc_vec *edir, uint64_t iray,
                   const uint64_t *cdf)
{
    int ph = phase - 1, rc;
    *nis = 0;
    if (!have_in) {
        if (phase == 1) {                                   /* main.f90:95-101 */
            if (S->source == 2) emit_crs(S, pos, dir, d);
            else if (S->source == 3 || S->source == 5) {       /* 5: iSORS(ring = .false.), test-only */
                if (!emit_isors(S, pos, dir, d, S->source == 3)) { *epos = *pos; *edir = *dir; return ORC_NO_INTERSECTION; }
            }
            else emit_ring(S, pos, dir, d);
        } else {                                            /* main.f90:132-142 */
            if (S->source == 4) {
                if (!emit_image_ray(S, cdf, iray, pos, dir, d)) { *epos = *pos; *edir = *dir; return ORC_LOST_TELESCOPE; }
            } else if (S->source == 1) emit_spot(S->cosThetaMax, S->nphotons, (int)(iray + 1), pos, dir);
            else if (S->source == 3) emit_point(S->cosThetaMax, S->bottle.centre.z, pos, dir, d);   /* main.f90:140 */
            else emit_point(S->cosThetaMax, 0.0, pos, dir, d);
        }
    }
    *epos = *pos; *edir = *dir;
    if (phase == 2 && S->use_bottle) {
        rc = bottle_forward(&S->bottle, pos, dir, d, nis);
        if (rc == 2) return ORC_NO_INTERSECTION;
        if (rc) return ORC_LOST_BOTTLE;
    }
    rc = telescope(S, ph, pos, dir, d, nis);
    if (rc == 1) return ORC_LOST_TELESCOPE;
    if (rc == 2) return ORC_HELP3;
    return make_image(*dir, *pos, S->image_diameter, xp, yp);
}

int orc_trace_rays(const orc_system *sys, int phase, int64_t n,
                   const double *pos_dir_in, int nu, const double *u, int draw_base,
                   uint64_t seed, uint64_t first_ray,
                   double *pos_dir_out, double *emitted_out, int32_t *status,
                   int32_t *bin_xy, int32_t *n_isect, int32_t *n_draws)
{
    if (!sys || (phase != 1 && phase != 2) || n < 0) return -1;
    uint64_t *cdf = build_cdf(sys);
    for (int64_t i = 0; i < n; i++) {
        draws_t d;
        memset(&d, 0, sizeof d);
        if (u) { d.table = u + i; d.stride = n; d.len = nu; }
        d.seed = seed; d.ray = first_ray + (uint64_t)i; d.phase = phase; d.k = draw_base;
        orc_vec pos = {0, 0, 0}, dir = {0, 0, 0}, ep, ed;
        if (pos_dir_in) {
            pos = v(pos_dir_in[0 * n + i], pos_dir_in[1 * n + i], pos_dir_in[2 * n + i]);
            dir = v(pos_dir_in[3 * n + i], pos_dir_in[4 * n + i], pos_dir_in[5 * n + i]);
        }
        int nis, xp = -9999, yp = -9999;
        int st = one_ray(sys, phase, pos_dir_in != NULL, &pos, &dir, &d, &nis, &xp, &yp, &ep, &ed,
                         first_ray + (uint64_t)i, cdf);
        if (pos_dir_out) {
            pos_dir_out[0 * n + i] = pos.x; pos_dir_out[1 * n + i] = pos.y; pos_dir_out[2 * n + i] = pos.z;
            pos_dir_out[3 * n + i] = dir.x; pos_dir_out[4 * n + i] = dir.y; pos_dir_out[5 * n + i] = dir.z;
        }
        if (emitted_out) {
            emitted_out[0 * n + i] = ep.x; emitted_out[1 * n + i] = ep.y; emitted_out[2 * n + i] = ep.z;
            emitted_out[3 * n + i] = ed.x; emitted_out[4 * n + i] = ed.y; emitted_out[5 * n + i] = ed.z;
        }
        if (status) status[i] = st;
        if (bin_xy) { bin_xy[i] = xp; bin_xy[n + i] = yp; }
        if (n_isect) n_isect[i] = nis;
        if (n_draws) n_draws[i] = d.k;
    }
    free(cdf);
    return 0;
}

int orc_trace(const orc_system *sys, int phase, uint64_t first, uint64_t n, uint64_t seed,
              int32_t *image, uint64_t *counters, int nthreads)
{
    if (!sys || (phase != 1 && phase != 2) || !image || !counters) return -1;
    uint64_t lost = 0, isect = 0, binned = 0, help3 = 0;
    int32_t *layer = image + (size_t)(phase - 1) * 401 * 401;
    uint64_t *cdf = build_cdf(sys);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(static) reduction(+:lost, isect, binned, help3)
    for (uint64_t i = 0; i < n; i++) {
        draws_t d;
        memset(&d, 0, sizeof d);
        d.seed = seed; d.ray = first + i; d.phase = phase; d.k = 0;
        orc_vec pos, dir, ep, ed;
        int nis, xp = 0, yp = 0;
        int st = one_ray(sys, phase, 0, &pos, &dir, &d, &nis, &xp, &yp, &ep, &ed, first + i, cdf);
        isect += (uint64_t)nis;
        if (st == ORC_LOST_BOTTLE || st == ORC_LOST_TELESCOPE || st == ORC_HELP3 || st == ORC_NO_INTERSECTION) lost++;
        if (st == ORC_HELP3) help3++;
        if (st == ORC_BINNED) {
            binned++;
#pragma omp atomic
            layer[(xp + 200) + 401 * (yp + 200)]++;
        }
    }
    free(cdf);
    counters[phase - 1] += lost;
    counters[2 + phase - 1] += isect;
    counters[4 + phase - 1] += binned;
    counters[6 + phase - 1] += help3;
    return 0;
}
