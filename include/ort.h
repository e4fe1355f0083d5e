/* ort.h — C ABI of libort_hip.so: the MI355X (gfx950) per-ray hot path of
 * lewisfish/OpticalRayTrace (emit -> bottle -> plano-convex -> doublet -> image
 * plane -> NA-filtered binning), behind plain C types.
 *
 * The reference has no FFI for this path: it sits inside `program raytrace`
 * behind Fortran module procedures.  Each entry point below names the reference
 * interface it replaces; INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *  - every function returns 0 on success or a negative ORT_E_* code; nothing
 *    throws or aborts across the ABI; ort_last_error() gives a message;
 *  - the caller owns every host buffer; device memory is owned by the context;
 *  - one context per device; calls on one context are serialised by the caller;
 *  - ray bundles are structure-of-arrays fp64 [6][n]: x, y, z, dx, dy, dz
 *    (component c of ray i at c*n + i) — the coalesced HBM layout;
 *  - the image is int32 [2][401][401], layer 0 = ring (phase 1), layer 1 =
 *    point (phase 2), bin (xp, yp) of a layer at (xp+200) + 401*(yp+200):
 *    the storage order of `image(-200:200,-200:200,2)` (src/main.f90:35);
 *  - there is NO CPU fallback: without a HIP device every call fails.
 */
#ifndef ORT_H
#define ORT_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORT_ABI_VERSION 2
#define ORT_MAX_SURFACES 12
#define ORT_IMAGE_N 401
#define ORT_IMAGE_BINS (2 * 401 * 401)
#define ORT_NUM_COUNTERS 8
#define ORT_MAX_RAYS_PER_LAUNCH (1u << 27)  /* ort_trace / ort_trace_resident cut a call into kernel launches of at
                                               most this many rays (bounds the re-run list, the fp32 hit log and the
                                               32-bit ray keys; every launch costs ~25 us of ramp and drain) */
#define ORT_MAX_RAY_INDEX (1ull << 40)      /* global ray indices are below this (the keyed draw counter is
                                               (ray << 24) + draw); a call reaching beyond it is ORT_E_INVALID */

/* error codes */
#define ORT_OK 0
#define ORT_E_INVALID (-1)   /* bad argument */
#define ORT_E_NODEVICE (-2)  /* no HIP device / device index out of range */
#define ORT_E_HIP (-3)       /* a HIP runtime call failed */
#define ORT_E_NOMEM (-4)
#define ORT_E_NOCOMM (-5)    /* RCCL missing or a collective failed (ort_allreduce) */
#define ORT_MAX_DEVICES 16   /* contexts one ort_allreduce call can span */

/* surface kinds of the staged surface list */
#define ORT_SURF_PLANE 0     /* move to z = cz, aperture test, Fresnel at N=(0,0,-1)  (src/lens.f90:446-459) */
#define ORT_SURF_SPHERE 1    /* intersect_sphere + move + normal + Fresnel            (src/lens.f90:462-479, :568-628) */
#define ORT_SURF_CYLINDER 2  /* x-axis cylinder in y-z                                (src/lens.f90:255-297, :303-348) */
#define ORT_SURF_ELLIPSE 3   /* x-axis elliptic cylinder, semi-axes radius (z), radius_b (y) (src/surfaces.f90:133-176) */
#define ORT_SURF_IRIS 4      /* test-only plane: r at z = cz must be <= aperture; position restored (src/lens.f90:551-565, :632-644) */
#define ORT_SURF_IMAGE 5     /* move to z = cz, then makeImage2D                      (src/optics_system.f90:48-49, src/imageMod.f90:19-58) */

/* emitters */
#define ORT_EMIT_RING 0      /* ring            src/sourceMod.f90:250-300 (phase 1 default) */
#define ORT_EMIT_POINT 1     /* point           src/sourceMod.f90:12-47   (phase 2: point and crs sources) */
#define ORT_EMIT_SPOT 2      /* create_spot     src/sourceMod.f90:122-159 (phase 2: spot source, no draws) */
#define ORT_EMIT_CRS 3       /* point_on_bottle src/sourceMod.f90:50-89   (phase 1: crs source) */
#define ORT_EMIT_IMAGE 4     /* emit_image/emit src/sourceMod.f90:303-361 (phase 2: image source; needs ort_set_image_source) */
#define ORT_EMIT_ISORS 5     /* iSORS(ring=.true.) src/sourceMod.f90:162-247 (phase 1: isors source; its phase 2 is
                                `point` started at bottle%centre%z, src/main.f90:140: point_offset below) */
#define ORT_EMIT_ISORS_NORING 6 /* iSORS(ring=.false.) + bottle_backward_sub, src/sourceMod.f90:162-247, src/lens.f90:352-423: through
                                the bottle from outside to the plane z = bottle%centre%z.  No call site of the reference reaches
                                it (src/main.f90:141 is commented out); generic kernels only.  Needs the point loop's list to
                                start with the bottle (surfaces[1][0], [1][1]: its glass and contents) */
#define ORT_IMAGE_SOURCE_CELLS (512 * 512)

/* surface flags */
#define ORT_F_SKIP_ON_REFLECT 1u /* a Fresnel reflection ends the ray (every surface but the plano flat face, src/lens.f90:458-459) */
#define ORT_F_MISS_IS_HELP3 2u   /* a miss here is the reference's `error stop "Help3"` (src/lens.f90:617): counted, not fatal */
#define ORT_F_BOTTLE 4u          /* losses at this surface are bottle losses (src/main.f90:150-151) */
#define ORT_F_TRACK 8u           /* the ray-path tracker pushes the position after this surface
                                    (src/main.f90:147, src/optics_system.f90:29,39,50) */
#define ORT_F_SCATTER 16u        /* random walk in the medium before this surface (tauint + stokes) */
#define ORT_MAX_PATH 6           /* emission + bottle + L1 + L2 + image plane (+ spare) */

/* One optical surface as staged into LDS (112 B). */
typedef struct ort_surface {
    double cx, cy, cz;   /* centre (sphere / cylinder / ellipse) or plane z in cz */
    double radius;       /* sphere / cylinder radius; ellipse semi-axis along z */
    double radius_b;     /* ellipse semi-axis along y (unused otherwise) */
    double n1, n2;       /* refractive index before / after the surface */
    double eta;          /* n1 / n2, rounded once on the host (IEEE division: same bits as on the device) */
    double aperture;     /* reject when sqrt(x^2+y^2) > aperture after the move; < 0: no test */
    /* medium BEFORE the surface when ORT_F_SCATTER (bottle contents / wall, src/lens.f90:262-282,
     * :312-333): absorption, scattering [1/m], Henyey-Greenstein g, cylinder radius tauint uses */
    double mua, mus, hgg, scat_radius;
    int32_t kind;        /* ORT_SURF_* */
    uint32_t flags;      /* ORT_F_* */
} ort_surface;

/* Everything the kernels need for both phases; built by the host from the
 * reference's settings/.params files (src/setupMod.f90:57-133, src/lens.f90:73-227,
 * src/main.f90:51-81, :113-116). */
typedef struct ort_system {
    int32_t abi_version;          /* ORT_ABI_VERSION */
    int32_t n_surfaces[2];        /* [0] phase 1 (ring), [1] phase 2 (point) */
    int32_t ring_ellipse;         /* bottle%ellipse for the ring emitter (src/sourceMod.f90:276) */
    int32_t split[2];             /* per phase: index of the first surface of segment 2 of the queued
                                     kernel (rays that survive surfaces [0, split) are compacted through
                                     a wave-private LDS queue); 0 or n_surfaces = single segment.
                                     Scheduling only: results do not depend on it. */
    int32_t emitter[2];           /* per phase: ORT_EMIT_* (src/main.f90:95-101, :132-142) */
    ort_surface surfaces[2][ORT_MAX_SURFACES];
    /* point emitter, src/sourceMod.f90:12-47 */
    double cos_theta_max;
    /* ring emitter, src/sourceMod.f90:250-300 */
    double ring_r1, ring_r2;      /* squared annulus radii (src/main.f90:68-70) */
    double ring_lens_r2;          /* (L2%radius + 10e-3)^2 */
    double ring_lens_z;           /* L2%fb */
    double ring_bottle_ra, ring_bottle_rb, ring_bottle_z;
    /* image, src/imageMod.f90:19-58 */
    double bin_width;             /* image_diameter / 401. */
    double inv_bin_width;         /* 1 / bin_width, for the filtered bin lookup (ort_device.h) */
    double na_angle;              /* asin(0.22) */
    double na_cos_min;            /* smallest x with acos(x) <= na_angle under the host libm: the
                                     NA test of src/imageMod.f90:39-44 as one compare per ray */
    double twopi;                 /* 2.*4.*atan(1.) (src/constants.f90:5) */
    /* spot source, create_spot (src/sourceMod.f90:122-159): twopi/sqrt(nphotons), acos(cosThetaMax)/sqrt(nphotons) */
    double spot_dphi, spot_dtheta;
    /* crs source, point_on_bottle (src/sourceMod.f90:50-89): Gaussian sigma (spot_size after
     * src/setupMod.f90:136), cylinder radius radiusa + thickness, bottle centre y, z */
    double crs_sigma, crs_radius, crs_cy, crs_cz;
    /* image source, emit (src/sourceMod.f90:325-361): lens%radius**2 and lens%fb of the 843 nm L2 */
    double img_lens_r2, img_lens_z;
    /* point emitter's start height: 0, or bottle%centre%z for the isors source (src/main.f90:140) */
    double point_offset;
    /* isors source, iSORS (src/sourceMod.f90:162-247): Gaussian sigma of the beam (ring_width), the
     * axicon cone (k = (radius/height)**2, height; n = 1.4), base_pos = (seperation + beam_width) /
     * tan(alpha (n - 1)), the z the ray is put at beside the bottle (radiusa + centre%z +
     * epsilon(1.)), the bottle's inner wall it is carried to (circular: rad1 = rad2 = radiusa -
     * thickness; elliptical: semi-axes minus thickness; centre y, z), lens%radius**2, lens%fb */
    double isors_sigma, isors_k, isors_height, isors_base_pos, isors_z;
    double isors_rad1, isors_rad2, isors_cy, isors_cz, isors_lens_r2, isors_lens_z;
} ort_system;

/* per-ray status written by ort_trace_rays */
#define ORT_ST_BINNED 0
#define ORT_ST_NA_REJECT 1
#define ORT_ST_OFF_GRID 2
#define ORT_ST_LOST_BOTTLE 3
#define ORT_ST_LOST_TELESCOPE 4
#define ORT_ST_HELP3 5
#define ORT_ST_NO_INTERSECTION 6  /* tauint found no wall, or the isors source no bottle: the reference aborts
                                     (src/surfaces.f90:33-39, src/sourceMod.f90:216-218); counted as lost */

/* counters[ORT_NUM_COUNTERS] */
#define ORT_C_LOST_RING 0     /* rcount, src/main.f90:40, optics_system.f90:32,42 */
#define ORT_C_LOST_POINT 1    /* pcount, src/main.f90:41,151 */
#define ORT_C_ISECT_RING 2    /* ray-surface intersections evaluated (the metric's unit of work) */
#define ORT_C_ISECT_POINT 3
#define ORT_C_BINNED_RING 4
#define ORT_C_BINNED_POINT 5
#define ORT_C_HELP3_RING 6
#define ORT_C_HELP3_POINT 7

typedef struct ort_ctx ort_ctx;

/* Library / device probing. */
int ort_abi_version(void);
/* Hash (first 16 hex digits of SHA-256) of the kernel sources the library was built from
 * (csrc/Makefile SOURCES: every ort_*.hip unit, then the headers, include/ort.h among them), so a host
 * — and the tests — can tell a stale binary from the sources next to it. */
const char *ort_build_id(void);
const char *ort_last_error(void);
int ort_device_count(int *count);

/* Context = one device's image accumulator + counters + staged system.
 * Replaces the allocation and zeroing in src/main.f90:35-41 and the lens
 * objects handed to the loop (src/main.f90:43). `stream` is a hipStream_t
 * passed as void*; NULL = the device's default (null) stream.  Every launch, copy and
 * memset of the context is issued on that stream, in call order. */
int ort_create(const ort_system *sys, int device, void *stream, ort_ctx **out);
/* Waits for the context's stream and releases everything the context owns.  Attached buffers
 * (ort_attach_buffers) are the caller's and are left as they are — in particular NOT completed: hits
 * still held in the per-XCD copies, and rays a group of launches deferred to its literal re-run, are
 * dropped.  A host that wants them calls ort_flush (or any of the calls listed there) first. */
int ort_destroy(ort_ctx *ctx);
/* Stage another system for the launches that FOLLOW (a sweep: runner.py re-runs the whole program per settings
 * file, here ~8 KB move).  Asynchronous on the context's stream and without waiting for it: traces already
 * queued keep the system they were launched with (a ring of 16 staged systems), so a host can queue
 * set_system / attach_buffers / trace for simulation after simulation and synchronise once at the end. */
int ort_set_system(ort_ctx *ctx, const ort_system *sys);
/* Histogram of the `image` light source (reference imgin, src/sourceMod.f90:363-408) as its
 * cumulative sum in the order emit_image scans the cells (:313-321): cdf[0] = 0,
 * cdf[s+1] = cdf[s] + count of cell s, ORT_IMAGE_SOURCE_CELLS + 1 host values.  Ray i of the
 * point loop starts in the cell with cdf[s] <= i < cdf[s+1].
 * SYNCHRONOUS, unlike ort_set_system: the context holds ONE table (2 MB), so the call waits for the traces already
 * queued on the stream (they read the old table) before it returns — a batched sweep of `image`-source systems drains
 * the queue once per simulation that brings a new table. */
int ort_set_image_source(ort_ctx *ctx, const int64_t *cdf);
int ort_reset(ort_ctx *ctx);                       /* image = 0, counters = 0 (src/main.f90:39-41) */
/* ort_trace bins into per-XCD private copies of the image and adds them into the image when the
 * image is next needed, and the few rays a group of launches deferred to the literal re-run are counted
 * (image AND counters) when the group closes (ort_read, ort_reset, ort_allreduce, ort_synchronize,
 * ort_device_image, ort_device_counters, ort_attach_buffers do both themselves).  A host that reads ATTACHED device buffers on its own
 * (torch tensors, its own RCCL communicator) calls ort_flush first; asynchronous, on the context's
 * stream. */
int ort_flush(ort_ctx *ctx);

/* The hot loop.  Replaces one OpenMP `do i = 1, nphotons` loop of
 * src/main.f90:90-109 (phase 1, ring) or :127-162 (phase 2, point) over the
 * global ray indices [first_ray, first_ray + n_rays): emit, trace, bin into the
 * context's device image and counters.  Draws are ORT-RNG-v2 keyed on
 * (seed, phase, global ray index, draw index), so any partition of the index
 * range over calls, contexts or GPUs accumulates the same image.
 * Asynchronous on the context's stream. */
int ort_trace(ort_ctx *ctx, int phase, uint64_t first_ray, uint64_t n_rays, uint64_t seed);

/* A BATCH of simulations in few launches — runner.py's experiment loops (:113-133 -s, :136-155 -p, :158-186 -i, :189-208 -o,
 * :232-261 -l: 75 systems), which the reference runs as one `./install.sh -n 32 -f <settings>` process per settings file
 * (runner.py:26-47).  Loop `phase` over the rays [first_ray, first_ray + n_rays) of EACH of the n systems, simulation i into
 * its own accumulators: d_images[i] (device, int32 [ORT_IMAGE_BINS]; NULL — or d_images itself NULL — = this simulation's
 * image is not wanted, `make_images = .false.`, src/main.f90:183: its hits are counted, not binned) and d_counters[i]
 * (device, uint64 [ORT_NUM_COUNTERS], required).  The accumulators are added to, not zeroed.  Means exactly, for every i:
 *     ort_set_system(ctx, &systems[i]); ort_attach_buffers(ctx, d_images[i], d_counters[i]); ort_trace(ctx, phase, ...)
 * — and that is bit for bit what it accumulates — but the simulations whose list is a surface program (every set-up of
 * runner.py's loops with the ring, point, crs or isors source, clear media) share MULTI-SYSTEM LAUNCHES: one kernel launch
 * per program over all of them (workgroup -> simulation, ray range), their systems and arguments staged in one copy, and
 * one literal re-run launch for what they deferred; at 1e6 rays a simulation alone is ~40 us of work inside ~25 us of ramp
 * and drain.  Everything else (a list no program matches, scattering media, another precision or kernel variant, more than
 * 2^24 rays — such a simulation fills the chip by itself) is traced one by one inside the call.  The context's staged
 * system, attached buffers and own accumulators are as before when it returns.  The `image` light source (one table per
 * context, ort_set_image_source) is refused with ORT_E_INVALID.  Asynchronous on the context's stream; complete results
 * (deferred rays included) need no further call than waiting for the stream. */
int ort_trace_batch(ort_ctx *ctx, int n, const ort_system *systems, int phase, uint64_t first_ray, uint64_t n_rays,
                    uint64_t seed, void *const *d_images, void *const *d_counters);

/* Same loop with the ray bundle resident in HBM (device pointer, SoA fp64
 * [6][n]) instead of emitted in-kernel: replaces the loop body after the emitter
 * call (src/main.f90:104-108, :145-161).  Ray i consumes draws
 * draw_base, draw_base+1, ... of key (seed, phase, first_ray + i).
 * ort_emit fills such a bundle with the phase's source
 * (ring src/sourceMod.f90:250, point :12), using draws 0..draw_base-1. */
int ort_emit(ort_ctx *ctx, int phase, uint64_t first_ray, uint64_t n_rays, uint64_t seed,
             double *d_pos_dir);
int ort_trace_resident(ort_ctx *ctx, int phase, uint64_t first_ray, uint64_t n_rays,
                       uint64_t seed, int draw_base, const double *d_pos_dir);

/* Parity / debug entry on HOST buffers, no side effect on the image or the
 * counters.  pos_dir_in NULL => emit in-kernel.  u NULL => keyed draws as in
 * ort_trace; else u is [nu][n] (draw k of ray i at k*n + i, first used draw is
 * draw_base).  Any output may be NULL.  Outputs: final pos/dir, the emitted
 * pos/dir, status (ORT_ST_*), bin (xp at i, yp at n+i; -9999 when not binned),
 * intersections evaluated, draws consumed.  Synchronous. */
int ort_trace_rays(ort_ctx *ctx, int phase, int64_t n,
                   const double *pos_dir_in, int nu, const double *u, int draw_base,
                   uint64_t seed, uint64_t first_ray,
                   double *pos_dir_out, double *emitted_out, int32_t *status,
                   int32_t *bin_xy, int32_t *n_isect, int32_t *n_draws);

/* Ray-path tracker (reference src/stackMod.f90, `use_tracker`): for rays [first_ray,
 * first_ray+n) of `phase` with keyed draws, the positions the reference pushes on its stack —
 * after emission (src/main.f90:103,144), after bottle%forward (:147), after each lens of
 * telescope and at the image plane (src/optics_system.f90:29,39,50) — plus, for a ray that is
 * lost, the position where it ended.  path is [n][ORT_MAX_PATH][3] (host), npath[n] the number
 * of points, status[n] the ORT_ST_* outcome.  No side effect on the image.  Synchronous. */
int ort_trace_paths(ort_ctx *ctx, int phase, int64_t n, uint64_t seed, uint64_t first_ray,
                    double *path, int32_t *npath, int32_t *status);

/* Accumulator access.  ort_read replaces reading `image`, `rcount`, `pcount`
 * after the loops (src/main.f90:175-185); synchronises the stream.
 * ort_device_image / ort_device_counters expose the device buffers (int32
 * [ORT_IMAGE_BINS], uint64 [ORT_NUM_COUNTERS]) so the host can sum them across
 * GPUs with RCCL in place (torch.distributed all_reduce) — the multi-GPU
 * equivalent of the OpenMP atomic image + reduction (src/main.f90:88,
 * src/imageMod.f90:55). */
int ort_read(ort_ctx *ctx, int32_t *image, uint64_t *counters);
/* Use caller-owned DEVICE buffers (e.g. torch tensors: int32[ORT_IMAGE_BINS],
 * int64[ORT_NUM_COUNTERS]) as the accumulators from now on; they are not zeroed
 * and not freed by the context.  NULL, NULL returns to the context's own.  Asynchronous: what was traced before
 * the call is completed in the OLD buffers by work queued on the stream (they must stay valid until the stream
 * has passed this point); what is traced afterwards lands in the new ones. */
int ort_attach_buffers(ort_ctx *ctx, void *d_image, void *d_counters);
/* Sum image and counters over the n contexts of ONE process (one context per device) in place,
 * over RCCL / xGMI: afterwards every context holds the global sums.  The multi-GPU equivalent
 * of the OpenMP atomic image + `reduction(+:rcount,pcount)` (src/imageMod.f90:55,
 * src/main.f90:88) for a host that drives all GPUs of a node itself (e.g. a Fortran main
 * program): trace shard [N g/n, N (g+1)/n) on context g, then one
 * ort_allreduce.  Asynchronous on each context's stream; ort_read synchronises.  RCCL is loaded
 * on first use (ORT_E_NOCOMM if absent).  A one-process-per-GPU host reduces the attached
 * buffers with its own communicator instead (torch.distributed: tracer.py). */
int ort_allreduce(ort_ctx **ctxs, int n);
/* The number of ranks of the communicator the last ort_allreduce of this process used, as RCCL reports it
 * (ncclCommCount); 0 before the first one.  A multi-GPU run states it next to its figures (bench.py `ranks_seen`). */
int ort_allreduce_ranks(int *n_ranks);
/* Give back the RCCL communicators ort_allreduce keeps for this process (they are created by its first call and re-used
 * while the device list stays the same; a call that fails drops them by itself, so the next one starts afresh).  For a host
 * that is done reducing before it is done with the devices — the counterpart of leaving the OpenMP parallel region
 * (src/main.f90:163).  Safe to call at any time, also before the first ort_allreduce. */
int ort_comm_destroy(void);
int ort_device_image(ort_ctx *ctx, void **d_image);
int ort_device_counters(ort_ctx *ctx, void **d_counters);
int ort_synchronize(ort_ctx *ctx);
/* Executed-work counters since the last ort_reset — bookkeeping for measurements, NOT part of the
 * result (the image and counters above are what the reference computes, however the work was done):
 *   [ORT_W_CULLED]   ring rays that segment 0 of the ring programs counted as lost at the first aperture
 *                    (one intersection each in ORT_C_ISECT_RING, as the reference executes them) WITHOUT
 *                    emitting them or solving a surface — see ort_set_kernel_variant bit 3
 *   [ORT_W_DEFERRED] rays the filtered kernels handed to the literal re-run (traced twice in part)
 * Synchronous. */
#define ORT_NUM_WORK 2
#define ORT_W_CULLED 0
#define ORT_W_DEFERRED 1
int ort_work_counters(ort_ctx *ctx, uint64_t *work);
/* Optional: allocate now the per-launch scratch a trace of up to n_rays rays needs (otherwise the
 * first such ort_trace allocates it, synchronising the stream once).  The reference allocates its
 * image up front as well (src/main.f90:35). */
int ort_reserve(ort_ctx *ctx, uint64_t n_rays);

/* Timing of the last launch of each kernel kind on the context's own stream
 * (HIP events recorded around the launch): ms, or <0 if none.  kind: 0 fused
 * trace, 1 resident trace, 2 emit. */
int ort_last_kernel_ms(ort_ctx *ctx, int kind, float *ms);
int ort_set_timing(ort_ctx *ctx, int enable);
/* Which kernel instantiation the context's last trace launch ran — e.g. "trace_queue_kernel<MODE_FUSED, double, PROG_POINT,
 * strict=1, wide=0>" — so that a measurement, or a test, can say what it measured (no counterpart in the reference). */
int ort_last_kernel_name(ort_ctx *ctx, char *buf, int capacity);
/* Durations (ms) of the most recent fused-trace launches (ort_trace), oldest first, at most 64:
 * every launch records its own event pair on the context's stream, so a host loop can issue
 * many steps without synchronising and read the per-launch times afterwards. */
int ort_kernel_times(ort_ctx *ctx, float *ms, int capacity, int *count);
/* Arithmetic of the traced path from now on: 0 (default) = fp64, the reference's arithmetic
 * (all `real` are fp64, src/Makefile:2), bit-exact; 1 = fp32 path (BASELINE configs[4]): the same
 * path in single precision, uniforms = top 24 bits of the same draws, without the exact path's corset
 * (hardware reciprocal / square root, fused multiply-adds, the cheap decision forms without margins).  It has
 * no reference to be exact against; tests/test_gpu_fp32.py measures its deviation from fp64;
 * 2 = fast fp64 (csrc/ort_fastd.h): fused multiply-adds, Newton-refined reciprocal / rsqrt
 * instead of IEEE divide / sqrt — ~1e-13 relative from the exact path (inside the 1e-10 of the
 * north star) but not bit-identical; tests/test_gpu_fastd.py measures it. */
int ort_set_precision(ort_ctx *ctx, int precision);
/* Tuning / A-B knob, a bit mask: bit 0 set (default) = queued kernel (LDS ray queue between
 * segments), clear = plain lockstep kernel; bit 1 set = every predicate evaluated literally
 * (no filtered predicates, see csrc/ort_device.h), clear (default) = filtered; bit 2 set =
 * bin straight into the image, clear (default) = bin into 8 private replicas folded into the
 * image after the launch; bit 3 set = the ring loop emits every ray, clear (default) = ring rays
 * whose third draw already puts them outside the first aperture are counted without being emitted
 * (queued surface-program kernels); bit 4 set = scattering bottles run the monolithic kernel (the random walk
 * compiled into the surface walk), clear (default) = the scattering pipeline (walk stages on full wavefronts in
 * front of the lean walk).  All combinations of bits 0-4 produce bit-identical rays, images and counters.
 * Bit 5 set = 53-BIT DRAWS (stream ORT-RNG-v2w, csrc/ort_device.h): every uniform is (h >> 11) * 2^-53 of its own hash,
 * as the reference's ran2() fills a real(8) (src/random_mod.f90:39-46), where the default stream ORT-RNG-v2 hands out
 * 32-bit draws, two per hash.  A different stream, therefore different rays (same statistics: tests/test_gpu_wide_draws.py
 * against the unmodified program).  In exact fp64 the production kernels have instantiations on this stream (the surface
 * programs, the generic filtered walk in clear media, the scattering pipeline: one hash per draw instead of one per two);
 * fp32 / fast fp64 and the A/B variants of bits 1 and 4 trace it with the lockstep kernel.  Every entry that draws honours
 * it (ort_trace, ort_emit, ort_trace_resident, ort_trace_rays with u == NULL, ort_trace_paths); the checker's keyed mode has
 * the same switch.
 * Bit 6 set = STRICT LIBM EMITTERS (exact fp64 only; ORT_E_INVALID with another precision): the light sources evaluate
 * sin / cos through glibc 2.35's own algorithms (csrc/ort_libm.h), entry by entry as the reference's compiled code calls
 * them, so an emitted ray — and with it every ray state, image and counter — equals the CPU checker's bit for bit; the
 * surface programs have their own instantiations with these emitters (csrc/ort_k_strict.hip).  Clear (default): the
 * emitters' own sin / cos (within an ulp of glibc's: emitted rays agree to 1e-12, discrete outcomes are identical).  The
 * scattering walk, rang's log and everything behind the emitters are glibc-exact / IEEE-exact in either setting.
 * Default 1. */
int ort_set_kernel_variant(ort_ctx *ctx, int variant);

#ifdef __cplusplus
}
#endif
#endif
