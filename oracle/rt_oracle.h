/*
 * rt_oracle.h -- CPU restatement of the per-pixel path-tracing loop of the
 * reference C renderer (/root/reference/c/main.c, vec.c, rand.c, structs.c).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (raytrace_amd/,
 * include/) may include, link or call this.  Only tests/, the smoke check and
 * the cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * Parity pin: the sequential mode below reproduces the reference's PPM bytes
 * for every (samples, threads) pair the golden table in tests/golden/ holds;
 * the per-function entry points are compared with the real reference's static
 * functions through oracle/_ref/libref_hooked.so (see ref_hook.c).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* material kinds, numbering of c/structs.c:10-14 */
enum { ORC_MATTE = 0, ORC_METAL = 1, ORC_GLASS = 2 };

/* One sphere, flat 36-byte record: c/structs.c:16-36 without the union.
 * param = fuzz for ORC_METAL, refraction index for ORC_GLASS, unused else. */
typedef struct {
  float center[3];
  float radius;
  int32_t kind;
  float albedo[3];
  float param;
} orc_sphere;

/* c/structs.c:43-47 */
typedef struct {
  float llc[3], horiz[3], vert[3], origin[3], u[3], v[3], w[3];
  float lens_radius;
} orc_camera;

typedef struct {
  uint64_t paths;      /* camera paths traced                               */
  uint64_t segments;   /* calls of the closest-hit scan (c/main.c:264)      */
  uint64_t tests;      /* ray-sphere tests executed (c/main.c:249)          */
  uint64_t rng_draws;  /* uniform draws                                     */
  uint64_t absorbed;   /* metal scatters that returned false (c/main.c:179) */
  uint64_t capped;     /* paths that ran all max_depth bounces              */
} orc_stats;

/* ---- reference RNG (c/rand.c:3-16) ---- */
uint32_t orc_ref_hash(uint32_t index);
double   orc_ref_next(uint32_t *state);     /* state starts at 123456789 */

/* ---- counter RNG of the deterministic mode (build-defined) ---- */
uint32_t orc_mix32(uint32_t x);
uint32_t orc_draw_hash(uint32_t x);
uint32_t orc_path_key(uint32_t seed, uint32_t pixel_index, uint32_t sample);
uint32_t orc_ctr_bits(uint32_t key, uint32_t k);   /* 31-bit integer draw (23 random bits, low 8 zero) */
double   orc_ctr_uniform(uint32_t key, uint32_t k);

/* ---- host-side inputs ---- */
void orc_camera_make(const float lookfrom[3], const float lookat[3],
                     const float vup[3], float vfov, float aspect,
                     float aperture, float focus_dist, orc_camera *out);
/* camera of c/main.c:490-497 for an nx-by-ny image */
void orc_camera_default(size_t nx, size_t ny, orc_camera *out);

size_t orc_scene_small(orc_sphere *out, size_t cap);                /* c/main.c:369-408 */
size_t orc_scene_trivial(orc_sphere *out, size_t cap);              /* c/main.c:348-363 */
/* c/main.c:286-344 generalised to a half-extent (11 in the reference);
 * *rng_state is the building thread's stream (123456789 when fresh). */
size_t orc_scene_random(orc_sphere *out, size_t cap, int half_extent,
                        uint32_t *rng_state);
/* in-place material remap: matte->glass(1.5), metal->fuzz 0, sphere 0 kept */
void orc_scene_glassify(orc_sphere *s, size_t n);

/* ---- per-function entry points (for unit parity with the reference) ---- */
int orc_hit_sphere(const orc_sphere *s, const float A[3], const float B[3],
                   float tmin, float tmax, float *t, float p[3], float nrm[3]);
int orc_hit_scene(const orc_sphere *s, size_t n, const float A[3],
                  const float B[3], float tmin, float tmax, float *t,
                  float p[3], float nrm[3]);
/* the same scan for m rays (6 floats each), on `threads` pthreads; index -1 = no hit */
int orc_hit_scene_many(const orc_sphere *s, size_t n, const float *rays, size_t m, float tmin, float tmax,
                       int32_t *index, float *t, size_t threads);
float orc_schlick(float cosine, float ref_idx);
int orc_refract(const float u[3], const float n[3], float ni_over_nt,
                float out[3]);

/* ---- whole renders ---- */

/* Sequential mode: byte-for-byte the reference's image for "-n nsamples
 * -t threads" (c/main.c:485-548, 446-483): contiguous row bands, every band's
 * RNG stream restarted at 123456789 (c/rand.c:9).  rgb8 holds 3*nx*ny bytes. */
int orc_render_sequential(const orc_camera *cam, const orc_sphere *s, size_t n,
                          size_t nx, size_t ny, size_t nsamples, size_t threads,
                          uint8_t *rgb8, orc_stats *stats);

/* Deterministic mode: rows [row0,row0+rows) (0 = top) of an nx-by-ny_total
 * image; every (pixel, sample) path draws from its own counter stream
 * orc_path_key(seed, row*nx+col, sample).  Pixel coordinates are the
 * single-band form of c/main.c:465-466 (ymin 0, ymax 1).  Optional outputs
 * (may be NULL): linear = mean radiance before gamma, 3 floats per pixel;
 * first_hit = sphere index hit by sample 0's camera ray, -1 for none;
 * hit_sum = sum over samples of (first-hit index + 1), wrapping uint32.
 * host_threads splits the rows over pthreads (result independent of it). */
int orc_render_det(const orc_camera *cam, const orc_sphere *s, size_t n,
                   size_t nx, size_t ny_total, size_t row0, size_t rows,
                   size_t nsamples, size_t max_depth, uint32_t seed,
                   uint8_t *rgb8, float *linear, int32_t *first_hit,
                   uint32_t *hit_sum, orc_stats *stats, size_t host_threads);

#ifdef __cplusplus
}
#endif
#endif
