/*
 * rt_amd.h -- C ABI of the MI355X path-tracing kernel library
 * (libraytrace_amd.so).  Plain C, plain pointers and sizes.
 *
 * The reference (inrick/raytrace, C renderer) has no plugin or FFI layer: all
 * of it is one translation unit of static functions.  Its one seam is the
 * pthread entry point `static void* render(void*)` with its `render_arg`
 * (c/main.c:437-483), called once per row band by `raytrace()`
 * (c/main.c:510-545).  Every entry point below replaces a piece of that
 * seam; each comment names the reference lines it stands in for.
 *
 * Threading: one rt_ctx per device (or several); calls on one ctx must not
 * overlap, calls on different contexts may run from different host threads.
 * Errors: every function returns 0 on success or an RT_ERR_* code and never
 * aborts; rt_last_error() gives the text for the calling thread.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: rt_job.sample0 and RT_FLAG_PROGRESSIVE; rt_counters.tests split into sphere_tests and box_tests;
 *    rt_render_device_static; the unit-level rt_debug_filter / rt_debug_box_pass / rt_debug_arith */
#define RT_AMD_ABI_VERSION 2

enum {
  RT_OK = 0,
  RT_ERR_ARG = 1,      /* bad argument (null pointer, zero size, row range) */
  RT_ERR_HIP = 2,      /* a HIP runtime call failed; see rt_last_error()    */
  RT_ERR_NO_SCENE = 3, /* render before rt_scene_upload                     */
  RT_ERR_NOMEM = 4,
  RT_ERR_LIMIT = 5     /* a size is beyond what the kernels support         */
};

/* material kinds, numbered as material_t of c/structs.c:10-14 */
enum { RT_MATTE = 0, RT_METAL = 1, RT_DIELECTRIC = 2 };

/* One sphere: the fields of `sphere` + `material` (c/structs.c:16-36) with
 * the union flattened.  param = fuzz (metal) or ref_idx (dielectric). 36 B. */
typedef struct {
  float center[3];
  float radius;
  int32_t type;
  float albedo[3];
  float param;
} rt_sphere;

/* The eight fields of `camera` (c/structs.c:43-47), same order, 88 B. */
typedef struct {
  float llc[3], horiz[3], vert[3], origin[3], u[3], v[3], w[3];
  float lens_radius;
} rt_camera;

enum {
  RT_KERNEL_MEGA = 0,      /* one persistent lane per pixel, whole bounce loop in one kernel */
  RT_KERNEL_WAVEFRONT = 1  /* material-sorted wavefront stages (comparison variant)          */
};

/* What one call renders: the fields of render_arg (c/main.c:437-444) other
 * than the pointers, generalised from one contiguous band to a set of
 * equally spaced row tiles so that GPUs can interleave.  Tile t covers image
 * rows [row0 + t*tile_stride, +tile_rows) clipped to ny; rows are counted
 * from the top as in the output buffer (c/main.c:458-479).  Outputs hold the
 * tiles back to back, top row first, no padding. */
typedef struct {
  uint32_t nx, ny;          /* full image size (pixel coordinates are global)   */
  uint32_t row0;            /* first row of tile 0                              */
  uint32_t tile_rows;       /* rows per tile                                    */
  uint32_t tile_stride;     /* rows between the starts of consecutive tiles     */
  uint32_t n_tiles;
  uint32_t nsamples;        /* render_arg.nsamples                              */
  uint32_t max_depth;       /* reference: 50 (c/main.c:262)                     */
  float tmin;               /* reference: (float)0.001 (c/main.c:264)           */
  uint32_t seed;            /* seed of the counter generator                    */
  uint32_t kernel;          /* RT_KERNEL_*                                      */
  uint32_t flags;           /* RT_FLAG_* or 0                                   */
  uint32_t sample0;         /* index of the first sample this call renders: 0, or, with
                               RT_FLAG_PROGRESSIVE, the number of samples the context's
                               running sums already hold                        */
} rt_job;

/* also fill the cycle and lane-occupancy counters of rt_counters (slower) */
#define RT_FLAG_PROFILE 1u
/* filter every sphere for every ray instead of skipping groups of spheres
 * whose bounding sphere no ray of the wave can touch; same image either way */
#define RT_FLAG_FULL_SCAN 2u

/* Keep the per-pixel running sums of this render in the context, and, when sample0 > 0, continue
 * them: the call renders samples [sample0, sample0 + nsamples) of every pixel and its outputs are the
 * mean over all sample0 + nsamples samples, added in sample order (c/main.c:462-469) -- so k calls of m
 * samples give the bytes of one call of k*m samples.  A call with sample0 > 0 must follow, on the same
 * context, a progressive call with the same scene, camera, image geometry, seed, depth and tmin whose
 * samples ended at sample0; anything else is RT_ERR_ARG.  This is what a viewer that re-renders on a
 * camera change and refines while the camera rests needs (rs/src/gui.rs:164-172, 340-357). */
#define RT_FLAG_PROGRESSIVE 4u

/* Renders on one context are pipelined: the trace kernels run on two internal streams in turn, each sample
 * batch's resolve waits for its trace on the context's stream, so the next batch's (or the next render's) trace
 * kernel starts while the last waves of this one are still finishing and while this one's resolve runs.  Outputs
 * are complete in stream order on the context's stream as always.  RT_FLAG_SERIAL runs every kernel of the call
 * on the context's stream instead, one after the other: the form to time a kernel alone (rt_ctx_trace_ms brackets
 * the trace kernel with events on the stream it runs on, and a pipelined launch spends part of that time waiting
 * for compute units behind the launch before it). */
#define RT_FLAG_SERIAL 8u

/* Device-side event counts accumulated by the kernels since the last reset. */
typedef struct {
  uint64_t paths;           /* camera paths finished                            */
  uint64_t segments;        /* closest-hit scans (calls of hit_scene, c/main.c:264) */
  uint64_t sphere_tests;    /* ray-sphere tests executed in those scans: one per (ray, real sphere) put through the
                               discriminant filter (the test of c/main.c:216-221); a full scan runs segments * n */
  uint64_t box_tests;       /* ray-box tests of the culled scan: super-group, group and batch boxes, one per (ray, box) */
  uint64_t candidates;      /* sphere tests that went on to the exact evaluation */
  /* the rest only with RT_FLAG_PROFILE, summed over waves */
  uint64_t wave_iterations;      /* bounce-loop iterations                      */
  uint64_t live_lane_iterations; /* lanes with a live path in those iterations  */
  uint64_t cycles_scan, cycles_drain, cycles_shade, cycles_total; /* shader clocks */
  uint64_t realtime_total;       /* the waves' lifetimes in 100 MHz ticks: shader clock = cycles_total / realtime_total * 100 MHz */
  uint64_t own_groups, wave_groups; /* summed over live lanes and segments: sphere groups the lane's own ray needed / its wave scanned */
  /* when the first and the last wave of the trace kernels since the reset started and ended, in 100 MHz ticks of the
   * device's constant clock: how long the launch ramps up and how long its tail is */
  uint64_t first_wave_start, last_wave_start, first_wave_end, last_wave_end;
} rt_counters;

typedef struct rt_ctx rt_ctx;

/* number of rows the job covers (sum over tiles after clipping) */
uint32_t rt_job_rows(const rt_job *job);

const char *rt_last_error(void);
int rt_abi_version(void);
int rt_device_count(int *count);

/* One context per device; replaces the per-thread state of c/main.c:531-539. */
int rt_ctx_create(int device, rt_ctx **out);
int rt_ctx_destroy(rt_ctx *ctx);
/* Run the context's work on a caller-owned hipStream_t (NULL = the context's
 * own stream).  Lets a host framework order renders with its own work. */
int rt_ctx_set_stream(rt_ctx *ctx, void *hip_stream);
int rt_ctx_synchronize(rt_ctx *ctx);

/* Copies the sphere list to the device: stands in for the `scene*` that
 * raytrace() hands every render thread (c/main.c:499-501, 520).  Host
 * pointer; the list is immutable until the next upload.  At most 65,535
 * spheres, and at most 65,536 scan positions once the list is padded to whole
 * groups and the much larger spheres have groups of their own (a scene of
 * 65,535 similar spheres fits; RT_ERR_LIMIT otherwise).  Centres and radii
 * must be finite. */
int rt_scene_upload(rt_ctx *ctx, const rt_sphere *spheres, size_t n);

/* render() of c/main.c:446-483 for the rows of `job`, asynchronous on the
 * context's stream.  All pointers are DEVICE pointers sized for
 * rt_job_rows(job)*nx pixels: rgb8 3 bytes per pixel (required); the others
 * may be NULL: linear = 3 floats per pixel, mean radiance before the square
 * root; first_hit = sphere index hit by sample 0's camera ray or -1;
 * hit_sum = sum over samples of (first-hit index + 1), wrapping. */
int rt_render_device(rt_ctx *ctx, const rt_camera *cam, const rt_job *job,
                     uint8_t *d_rgb8, float *d_linear, int32_t *d_first_hit,
                     uint32_t *d_hit_sum);

/* rt_render_device for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid it).  No counterpart in the reference.  Same arguments, same
 * bytes; the kernels' scratch (per-path radiance, per-pixel running sums, the work-queue head, the event
 * counts) lives in one array of RT_STATIC_SCRATCH_BYTES that the library holds statically on every GPU
 * it runs on, so the call makes no hipMalloc/hipFree and creates nothing.  Limits: the megakernel only
 * (RT_KERNEL_MEGA), no RT_FLAG_PROGRESSIVE, no RT_FLAG_PROFILE, the scene uploaded beforehand, and
 * rt_job_rows(job) * nx <= RT_STATIC_MAX_PIXELS (render larger frames in row bands); beyond that it
 * returns RT_ERR_LIMIT.  Samples beyond what the scratch holds at once are rendered in batches.  Every
 * call on a GPU uses that GPU's one array: calls that overlap on one GPU (two contexts, two streams)
 * race, and rt_ctx_counters does not see this entry point's work. */
#define RT_STATIC_SCRATCH_BYTES (32u << 20)
#define RT_STATIC_MAX_PIXELS 700000u
int rt_render_device_static(rt_ctx *ctx, const rt_camera *cam, const rt_job *job,
                            uint8_t *d_rgb8, float *d_linear, int32_t *d_first_hit,
                            uint32_t *d_hit_sum);

/* The same through context-owned device buffers, then copied to HOST
 * pointers; returns when the copies are done.  This is the form the
 * reference's caller would bind: buf/len of render_arg (c/main.c:438-439). */
int rt_render_host(rt_ctx *ctx, const rt_camera *cam, const rt_job *job,
                   uint8_t *h_rgb8, float *h_linear, int32_t *h_first_hit,
                   uint32_t *h_hit_sum);

/* Duration of the last render on this context (trace and resolve kernels of
 * every sample batch), measured with HIP events on the stream they ran on
 * (milliseconds).  Waits for that render. */
int rt_ctx_last_kernel_ms(rt_ctx *ctx, float *ms);
/* Duration of the trace kernel(s) alone of the render `back` renders ago
 * (0 = the last; the context keeps 32), summed over its sample batches. */
int rt_ctx_trace_ms(rt_ctx *ctx, int back, float *ms);
int rt_ctx_counters(rt_ctx *ctx, rt_counters *out, int reset);

/* ---- unit-level entry points (used by the parity tests) ---- */

/* hit_scene (c/main.c:239-256) for n_rays independent rays (host pointers,
 * 6 floats each: origin, direction) against the uploaded scene. */
int rt_debug_closest_hit(rt_ctx *ctx, const float *rays, size_t n_rays,
                         float tmin, float tmax, int32_t *index, float *t);
/* first n uniforms of the counter stream of (seed, pixel, sample) as the
 * device computes them (host pointer, doubles) */
int rt_debug_uniforms(rt_ctx *ctx, uint32_t seed, uint32_t pixel,
                      uint32_t sample, size_t n, double *out);


/* The kernels evaluate the conservative discriminant filter, the group box test and a few divisions and
 * square roots with hand-scheduled packed instructions.  These entry points run exactly those device
 * routines on caller-supplied operands (host pointers) so that tests can compare them value for value
 * with the plain statement of the same arithmetic (csrc/rt_core.h compiled for the host), which follows
 * c/main.c:216-222.
 *
 * rt_debug_filter: the filter discriminant of every (ray, sphere) pair.  rays: 6 floats each (A, B);
 * spheres: 4 floats each (centre, radius), n_spheres a multiple of 8.  df_four / df_two: n_rays *
 * n_spheres floats each (ray-major), from the eight-sphere and the four-sphere routine. */
int rt_debug_filter(rt_ctx *ctx, const float *rays, size_t n_rays, const float *spheres, size_t n_spheres,
                    float *df_four, float *df_two);
/* rt_debug_box_pass: boxes: 6 floats each (centre, half extents), n_boxes even.  pass_vec / pass_sca:
 * n_rays * n_boxes bytes (0/1) from the routine that reads boxes from vector and from scalar registers;
 * box_ray: 9 floats per ray, the per-ray coefficients as the device formed them (its reciprocal is not
 * the host's). */
int rt_debug_box_pass(rt_ctx *ctx, const float *rays, size_t n_rays, const float *boxes, size_t n_boxes,
                      float sigma0, float sigma1, uint8_t *pass_vec, uint8_t *pass_sca, float *box_ray);
/* rt_debug_arith: out[i] = op(a[i], b[i]) in doubles (floats are widened exactly):
 *   RT_ARITH_SQRT_WIDENED   sqrt((double)(float)a)                        (c/main.c:222)
 *   RT_ARITH_DIV_WIDENED    a / (double)(float)b through the reciprocal   (c/main.c:223, 231)
 *   RT_ARITH_DIV_F32_BY     (float)a / (float)b through the reciprocal    (c/main.c:229, 465-466)
 *   RT_ARITH_RCP_NARROWED   (float)(1.0 / (double)(float)a)               (c/vec.c:45)
 *   RT_ARITH_UNIT_FLOAT     (float) of the draw with 31-bit integer a     (c/rand.c:15, c/vec.c:68)
 *   RT_ARITH_UNIT_DOUBLE    the same draw as a double                     (c/rand.c:15) */
enum { RT_ARITH_SQRT_WIDENED = 0, RT_ARITH_DIV_WIDENED = 1, RT_ARITH_DIV_F32_BY = 2, RT_ARITH_RCP_NARROWED = 3,
       RT_ARITH_UNIT_FLOAT = 4, RT_ARITH_UNIT_DOUBLE = 5 };
int rt_debug_arith(rt_ctx *ctx, int op, const double *a, const double *b, size_t n, double *out);
/* After a render with RT_FLAG_PROFILE: one word per wave of the trace kernel's grid (the first n, at most 16384), its
 * lifetime in 100 MHz ticks << 24 | its trips of the bounce loop -- for diagnosing the tail of a launch. */
int rt_debug_wave_log(rt_ctx *ctx, uint64_t *out, size_t n);

#ifdef __cplusplus
}
#endif
#endif
