/*
 * rt_host.h -- host side of the renderer, plain C: the scene builders, the
 * camera set-up, the multi-GPU driver and the PPM writer.  It is what stays
 * on the CPU of the reference's c/main.c and c/structs.c; everything per
 * pixel goes through the C ABI of include/rt_amd.h.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the reference's generator (c/rand.c:3-16) with the seed made explicit;
 * only the scene builders use it */
typedef struct { uint32_t state; } rth_rng;
void rth_rng_init(rth_rng *g);                 /* state = 123456789 (c/rand.c:9) */
uint32_t rth_rng_hash(uint32_t index);
double rth_rng_next(rth_rng *g);

/* camera_new, c/structs.c:49-79 */
void rth_camera_new(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov, float aspect,
                    float aperture, float focus_dist, rt_camera *out);
/* the view raytrace() sets up, c/main.c:490-497, for an nx-by-ny image */
void rth_camera_default(uint32_t nx, uint32_t ny, rt_camera *out);

/* Scene builders.  Each returns the sphere count; called with out == NULL it
 * returns the count needed (the protocol of small_scene, c/main.c:366-372). */
size_t rth_scene_small(rt_sphere *out, size_t cap);                 /* c/main.c:369-408 */
size_t rth_scene_trivial(rt_sphere *out, size_t cap);               /* c/main.c:348-363 */
/* random_scene, c/main.c:286-344, on a (2*half_extent)^2 grid; the reference
 * has half_extent 11 (488 spheres); 100 gives the 40,004-sphere field */
size_t rth_scene_random(rt_sphere *out, size_t cap, int half_extent, rth_rng *g);
/* stress scene: random_scene's geometry, matte -> glass 1.5, metal -> mirror */
size_t rth_scene_glass(rt_sphere *out, size_t cap, int half_extent, rth_rng *g);

enum { RTH_SCENE_SMALL = 0, RTH_SCENE_FINAL = 1, RTH_SCENE_DENSE = 2, RTH_SCENE_GLASS = 3, RTH_SCENE_TRIVIAL = 4 };
/* allocates with malloc; caller frees */
int rth_scene_build(int which, rt_sphere **out, size_t *n);

/* Scene dump: a scene as a file, so that a fixture or a caller's scene can be rendered without its builder.
 * Little-endian: the 8 bytes "RTSCENE1", uint32 record size (36), uint32 count, then count records laid out as
 * rt_sphere (centre, radius, type, albedo, param: the sphere + material of c/structs.c:16-36).  rth_scene_load
 * allocates with malloc (caller frees) and rejects a file whose size, record size or material types are off. */
int rth_scene_save(const char *path, const rt_sphere *spheres, size_t n);
int rth_scene_load(const char *path, rt_sphere **out, size_t *n);

/* ppm_write, c/main.c:410-415 */
int rth_write_ppm(FILE *fp, const uint8_t *rgb8, size_t w, size_t h);
/* c/main.c:427-433: 8-bit RGB PNG (stored, i.e. uncompressed, deflate blocks) */
int rth_write_png(FILE *fp, const uint8_t *rgb8, size_t w, size_t h);
/* c/main.c:430-435: baseline JPEG (JFIF, Y Cb Cr 4:4:4, the standard's example tables scaled for `quality` 1..100;
 * the reference passes 90).  The build's own encoder: a conforming file of the same picture, not the bytes of the
 * reference's third-party encoder. */
int rth_write_jpg(FILE *fp, const uint8_t *rgb8, size_t w, size_t h, int quality);

/* raytrace(), c/main.c:485-548, with GPUs in place of threads: row tiles of
 * tile_rows rows are dealt round-robin over n_devices contexts (one host
 * thread each), every device renders its tiles in one launch, and the tiles
 * are copied into their places in rgb8 (3*nx*ny bytes, caller's).  Because
 * the generator is keyed by (pixel, sample) the image does not depend on
 * n_devices or tile_rows.  device_ids may be NULL (0..n_devices-1, wrapped
 * onto the devices present).  kernel_ms_max, if not NULL, receives the
 * longest per-device kernel time. */
typedef struct {
  uint32_t nx, ny, nsamples, max_depth, seed, tile_rows, kernel;
  int n_devices;
  const int *device_ids;
} rth_render_opts;

int rth_raytrace(const rt_camera *cam, const rt_sphere *spheres, size_t n, const rth_render_opts *opts, uint8_t *rgb8,
                 float *kernel_ms_max);
/* the same; devices_used, if not NULL, receives the number of distinct GPUs the ranks ran on */
int rth_raytrace_ex(const rt_camera *cam, const rt_sphere *spheres, size_t n, const rth_render_opts *opts, uint8_t *rgb8,
                    float *kernel_ms_max, int *devices_used);
/* text of the first failure of the last rth_* call on this thread (the failing library call may have run on
 * another thread, whose rt_last_error() the caller cannot see) */
const char *rth_last_error(void);
void rth_set_error(const char *text);

/* ---- render service: one GPU context and scene kept resident by a worker thread that re-renders when
 * a new camera arrives and refines the picture while the camera rests -- what the reference's
 * interactive front end does with its worker thread and job channel (rs/src/gui.rs:164-172, 340-357),
 * with progressive passes instead of one fixed-size render per request.
 *
 * rth_service_start uploads the scene once.  rth_service_submit replaces the pending request (the newest
 * camera wins); the worker renders it in passes of pass_samples samples, each pass continuing the sums
 * of the one before (RT_FLAG_PROGRESSIVE), until target_samples are in the picture or a newer request
 * arrives.  After every pass the picture so far is published; rth_service_latest copies it out with its
 * request id and sample count.  rth_service_wait blocks until request `id` has reached min_samples (or
 * its target_samples, if that is less; or failed).  rth_service_stop joins the worker and frees everything. */
typedef struct rth_service rth_service;
typedef struct {
  uint32_t nx, ny, pass_samples, target_samples, max_depth, seed;
} rth_service_request;
int rth_service_start(int device, const rt_sphere *spheres, size_t n, rth_service **out);
/* returns the id of the request (1, 2, ...) or 0 on a bad argument */
uint64_t rth_service_submit(rth_service *s, const rt_camera *cam, const rth_service_request *req);
/* rgb8: 3*nx*ny bytes of the request the picture belongs to (cap = its size in bytes); returns RT_OK,
 * RT_ERR_ARG when cap is too small or nothing has been published yet */
int rth_service_latest(rth_service *s, uint8_t *rgb8, size_t cap, uint64_t *id, uint32_t *samples, uint32_t *nx, uint32_t *ny);
int rth_service_wait(rth_service *s, uint64_t id, uint32_t min_samples);
/* scene uploads and render passes the worker has made so far */
void rth_service_stats(rth_service *s, uint64_t *uploads, uint64_t *passes);
int rth_service_stop(rth_service *s);

#ifdef __cplusplus
}
#endif
#endif
