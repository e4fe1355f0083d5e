/* TEST INFRASTRUCTURE — not part of the product.
 *
 * Uniform-draw source injected into the *reference's own* Fortran modules when
 * they are compiled into oracle/_ref/libort_ref.so (see oracle/Makefile).
 *
 * Why: every optical surface in the reference consumes one U[0,1) from
 * `ran2()` (reference src/surfaces.f90:275, src/random_mod.f90:39-46), and
 * `ran2()` is the compiler runtime's `random_number`, whose stream depends on
 * the compiler and the thread count.  Two implementations can only be compared
 * ray by ray when they are fed the same draws, so the harness defines the ONE
 * runtime entry the reference's ran2() reaches — flang lowers
 * `call random_number(ran2)` (src/random_mod.f90:44) to _FortranARandomNumber —
 * here, in front of the Fortran runtime in the library's symbol order.  Module
 * `random` itself (ran2, ranu, rang: the Box-Muller sampling of the crs and isors
 * sources) is the reference's own source, compiled unmodified.  Two modes:
 *
 *   table mode  — the k-th draw of the current ray is u[k] from a caller table
 *   keyed mode  — the k-th draw of ray i of phase p is ORT-RNG-v2(seed,p,i,k)
 *
 * ORT-RNG-v2 (the same definition is restated, independently, in
 * oracle/ort_oracle.c and in the HIP kernels):
 *   base = mix64(seed ^ (GOLDEN * phase)),   c = (ray << 24) + k
 *   h    = mix64(base + GOLDEN * ((c >> 1) + 1)),   mix64 = SplitMix64 finaliser
 *   u    = (k even ? h >> 32 : h & 0xffffffff) * 2^-32
 */
#include <stdint.h>
#include <stddef.h>

#define GOLDEN 0x9E3779B97F4A7C15ull

static inline uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

static _Thread_local const double *tl_table;   /* table mode when non-NULL */
static _Thread_local int64_t  tl_table_stride;
static _Thread_local int32_t  tl_table_len;
static _Thread_local uint64_t tl_base;         /* keyed mode: mix64(seed ^ GOLDEN*phase) */
static _Thread_local uint64_t tl_ray;
static _Thread_local int32_t  tl_draw;         /* draws consumed by the current ray */

void ortref_rng_table(const double *u, int64_t stride, int32_t len, int32_t first_draw)
{
    tl_table = u; tl_table_stride = stride; tl_table_len = len; tl_draw = first_draw;
}

void ortref_rng_key(uint64_t seed, int32_t phase, uint64_t ray, int32_t first_draw)
{
    tl_base = mix64(seed ^ (GOLDEN * (uint64_t)phase));
    tl_table = NULL;
    tl_ray = ray;
    tl_draw = first_draw;
}

int32_t ortref_rng_draws(void) { return tl_draw; }

double ortref_draw(void)
{
    int32_t k = tl_draw++;
    if (tl_table) {
        if (k >= tl_table_len) return 0.5;   /* never reached on the checked path */
        return tl_table[(int64_t)k * tl_table_stride];
    }
    uint64_t c = (tl_ray << 24) + (uint64_t)k;
    uint64_t h = mix64(tl_base + GOLDEN * ((c >> 1) + 1ull));
    uint32_t w = (c & 1ull) ? (uint32_t)h : (uint32_t)(h >> 32);
    return (double)w * 0x1.0p-32;
}

/* flang's lowering of `call random_number(x)`: the Fortran runtime's RandomNumber(harvest descriptor, source file, line).
 * A descriptor (ISO_Fortran_binding CFI_cdesc_t) begins with the base address; the reference only ever draws one
 * default real at a time (a real*8 under -fdefault-real-8): src/random_mod.f90:44. */
void _FortranARandomNumber(void *harvest, const char *source, int line)
{
    (void)source; (void)line;
    **(double **)harvest = ortref_draw();
}
/* init_rng (src/random_mod.f90:10-37; not called by the harness, whose draws are keyed per ray) reaches random_seed(size=)
 * and random_seed(put=): the other two entries of the runtime's random.cpp object.  Defined here so that the static
 * Fortran runtime's copy of that object — which would define RandomNumber a second time — is not linked at all. */
void _FortranARandomSeedSize(void *size, const char *source, int line)
{
    (void)source; (void)line;
    if (size && *(int **)size) **(int **)size = 1;
}
void _FortranARandomSeedPut(void *put, const char *source, int line) { (void)put; (void)source; (void)line; }
