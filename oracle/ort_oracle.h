/* TEST INFRASTRUCTURE — CPU oracle for the per-ray hot path of
 * lewisfish/OpticalRayTrace.  NOT part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Parity status: PINNED.  This restatement is checked ray by ray (bit-exact on
 * pos/dir/status/bin for explicit-input rays) against oracle/_ref/libort_ref.so,
 * i.e. the reference's own Fortran sources compiled with flang (tests/
 * test_oracle_vs_ref.py, needs /root/reference or the prebuilt _ref), and against
 * the committed fixtures tests/golden/*.npz generated from that library by
 * tests/golden/make_golden.py.
 *
 * The structs mirror the reference's derived types (src/lens.f90:8-52), not the
 * product's surface table, so the oracle and the product share no data layout.
 */
#ifndef ORT_ORACLE_H
#define ORT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double x, y, z; } orc_vec;

/* type plano_convex (src/lens.f90:8-20) */
typedef struct {
    double thickness, curve_radius, radius, fb, f, n1, n2;
    orc_vec centre, flatNormal;
} orc_plano;

/* type achromatic_doublet (src/lens.f90:27-33) */
typedef struct {
    double R1, R2, R3, radius, fb, f, n1, n2, n3, thickness;
    orc_vec centre1, centre2, centre3;
} orc_doublet;

/* type glass_bottle (src/lens.f90:40-48).  Every shipped bottle has mua = mus = 0
 * (src/lens.f90:195-219); the scattering members drive src/lens.f90:262-282, :312-333. */
typedef struct {
    double nbottle, ncontents, thickness, radiusa, radiusb;
    orc_vec centre;
    int32_t ellipse, pad;
    double mua_b, mus_b, mua_c, mus_c;   /* wall (_b) and contents (_c) absorption / scattering [1/m] */
} orc_bottle;

/* run state of src/main.f90 + module setup globals read by telescope
 * (src/optics_system.f90:11).  Index 0 = phase 1 lenses (settings wavelength),
 * index 1 = phase 2 lenses (843 nm, src/main.f90:113-116). */
typedef struct {
    orc_plano   L2[2];
    orc_doublet L3[2];
    orc_bottle  bottle;
    double cosThetaMax, r1, r2, img_plane, fibre_offset, image_diameter, iris_radius;
    int32_t iris_before, iris_after, use_bottle;
    /* light source (src/setupMod.f90:85-99): 0 point, 1 spot, 2 crs, 3 isors, 4 image.  isors: the
     * reference aborts — `error stop "no intersection with bottle!"`, src/sourceMod.f90:217 — as
     * soon as a ray reflects at the axicon (2.8 % per ray); here that ray ends with
     * ORC_NO_INTERSECTION and is counted as lost.  5 = iSORS(ring = .false.) in phase 1: a variant
     * no call site of the reference uses, restated to pin bottle_backward_sub (tests only). */
    int32_t source;
    int32_t nphotons, pad2;        /* create_spot needs the loop length (src/main.f90:138) */
    double isors_offset, ring_width, spot_size;   /* spot_size after src/setupMod.f90:136 */
    /* image source (source == 4): imgin of src/sourceMod.f90:363-408 in the order emit_image scans
     * it (:313-321: second index outer, first index inner), 512*512 counts; NULL otherwise */
    const int32_t *img_counts;
} orc_system;

/* per-ray status (same numbering as include/ort.h, restated) */
enum {
    ORC_BINNED = 0,          /* counted in the image */
    ORC_NA_REJECT = 1,       /* reached the image plane, angle > asin(0.22) */
    ORC_OFF_GRID = 2,        /* reached the image plane, |bin| > 200 or pos > 1000 */
    ORC_LOST_BOTTLE = 3,     /* skip in bottle%forward (counted, phase 2) */
    ORC_LOST_TELESCOPE = 4,  /* skip in telescope (counted) */
    ORC_HELP3 = 5,           /* doublet face 3 missed: reference aborts (lens.f90:617) */
    ORC_NO_INTERSECTION = 6  /* tauint (surfaces.f90:38) or iSORS (sourceMod.f90:217) found no bottle crossing: reference aborts */
};

/* counters[8]: 0 lost ring (rcount), 1 lost point (pcount), 2 intersections ring,
 * 3 intersections point, 4 binned ring, 5 binned point, 6 help3 ring, 7 help3 point */

double orc_sellmeier(double wave, double b1, double b2, double b3, double c1, double c2, double c3);
double orc_cauchy(double wave, double a, double b, double c);
double orc_dispersion(double wave, double a, double b, double c);
double orc_uniform(uint64_t seed, int32_t phase, uint64_t ray, int32_t draw);
/* keyed draws from now on: 0 (default) ORT-RNG-v2 (32-bit draws, two per hash), 1 ORT-RNG-v2w (53-bit draws,
 * one per hash) — the streams of the product's kernel variant bit 5 */
void orc_set_wide_draws(int32_t on);

/* init_emit_image, src/sourceMod.f90:363-408, serial semantics (nphotonsLocal = nphotons):
 * img is the 512x512 float64 file content as read (first index fastest); rounding draw k of
 * the (i, j) loop is ORT-RNG-v2(seed, phase 0, ray 0, k).  counts_scan receives imgin in the
 * order emit_image scans it. */
void orc_init_emit_image(const double *img, int32_t nphotons, uint64_t seed, int32_t *counts_scan);

/* Parity entry.  SoA [6][n] (x,y,z,dx,dy,dz).  pos_dir_in NULL => emit with the
 * phase's source.  u NULL => ORT-RNG-v2 keyed on (seed, phase, first_ray+i);
 * else u is [nu][n] and draw k of ray i is u[k*n+i], k starting at draw_base.
 * Any output pointer may be NULL. */
int orc_trace_rays(const orc_system *sys, int phase, int64_t n,
                   const double *pos_dir_in, int nu, const double *u, int draw_base,
                   uint64_t seed, uint64_t first_ray,
                   double *pos_dir_out, double *emitted_out, int32_t *status,
                   int32_t *bin_xy, int32_t *n_isect, int32_t *n_draws);

/* Bulk entry: loop bodies of src/main.f90:90-109 / :127-162 over global ray
 * indices [first, first+n), image int32[2][401][401] (xp fastest) accumulated. */
int orc_trace(const orc_system *sys, int phase, uint64_t first, uint64_t n, uint64_t seed,
              int32_t *image, uint64_t *counters, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
