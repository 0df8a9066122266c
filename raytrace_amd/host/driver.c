/*
 * driver.c -- raytrace() of c/main.c:485-548 with GPUs in place of threads.
 *
 * The reference cuts the image into `threads` contiguous row bands
 * (c/main.c:510-529), spawns one pthread per band (531-539) and joins (541-545).
 * Here the image is cut into tiles of tile_rows rows dealt round-robin to the
 * devices (sky rows are cheap and ground rows dear, so contiguous bands would
 * leave the top GPUs idle); one host thread per device creates its context,
 * uploads the scene, renders all its tiles in one launch and copies them to
 * their places in the caller's buffer.  That copy is the only communication.
 */
#include "rt_host.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* text of the first failure of the last rth_raytrace / service call on this thread (rt_last_error() is
 * per thread, and the failing call ran on a device thread) */
static _Thread_local char g_err[256];
const char *rth_last_error(void) { return g_err; }
void rth_set_error(const char *text) {
  strncpy(g_err, text ? text : "", sizeof g_err - 1);
  g_err[sizeof g_err - 1] = 0;
}

typedef struct {
  const rt_camera *cam;
  const rt_sphere *spheres;
  size_t n;
  const rth_render_opts *opts;
  int device, rank, world;
  uint8_t *image;
  float kernel_ms;
  int rc;
  char err[256];
} dev_job;

static void *device_thread(void *arg) {
  dev_job *j = (dev_job *)arg;
  const rth_render_opts *o = j->opts;
  rt_ctx *ctx = NULL;
  uint8_t *tiles = NULL;
  j->kernel_ms = 0.0f;
  j->rc = rt_ctx_create(j->device, &ctx);
  if (j->rc == RT_OK) j->rc = rt_scene_upload(ctx, j->spheres, j->n);
  if (j->rc == RT_OK) {
    rt_job job;
    memset(&job, 0, sizeof job);
    job.nx = o->nx;
    job.ny = o->ny;
    job.tile_rows = o->tile_rows;
    job.row0 = (uint32_t)j->rank * o->tile_rows;
    job.tile_stride = (uint32_t)j->world * o->tile_rows;
    uint32_t total_tiles = (o->ny + o->tile_rows - 1) / o->tile_rows;
    job.n_tiles = total_tiles > (uint32_t)j->rank ? (total_tiles - (uint32_t)j->rank + (uint32_t)j->world - 1) / (uint32_t)j->world : 0;
    job.nsamples = o->nsamples;
    job.max_depth = o->max_depth;
    job.tmin = (float)0.001;                                    /* c/main.c:264 */
    job.seed = o->seed;
    job.kernel = o->kernel;
    uint32_t rows = job.n_tiles ? rt_job_rows(&job) : 0;
    if (rows > 0) {
      size_t row_bytes = (size_t)3 * o->nx;
      tiles = (uint8_t *)malloc(row_bytes * rows);
      if (!tiles) {
        j->rc = RT_ERR_NOMEM;
      } else {
        j->rc = rt_render_host(ctx, j->cam, &job, tiles, NULL, NULL, NULL);
        if (j->rc == RT_OK) (void)rt_ctx_last_kernel_ms(ctx, &j->kernel_ms);
      }
      if (j->rc == RT_OK) {
        /* gather: tile t of this device -> rows row0 + t*stride of the image */
        size_t src_row = 0;
        for (uint32_t t = 0; t < job.n_tiles; t++) {
          uint32_t first = job.row0 + t * job.tile_stride;
          if (first >= o->ny) break;
          uint32_t cnt = o->ny - first < job.tile_rows ? o->ny - first : job.tile_rows;
          memcpy(j->image + row_bytes * first, tiles + row_bytes * src_row, row_bytes * cnt);
          src_row += cnt;
        }
      }
    }
  }
  if (j->rc != RT_OK) {
    strncpy(j->err, rt_last_error(), sizeof j->err - 1);
    j->err[sizeof j->err - 1] = 0;
  }
  free(tiles);
  if (ctx) rt_ctx_destroy(ctx);
  return NULL;
}

int rth_raytrace_ex(const rt_camera *cam, const rt_sphere *spheres, size_t n, const rth_render_opts *opts, uint8_t *rgb8,
                    float *kernel_ms_max, int *devices_used) {
  g_err[0] = 0;
  if (devices_used) *devices_used = 0;
  if (!cam || !spheres || !opts || !rgb8 || n == 0) { rth_set_error("null or empty argument"); return RT_ERR_ARG; }
  if (opts->n_devices <= 0 || opts->tile_rows == 0 || opts->nx == 0 || opts->ny == 0) { rth_set_error("zero size or device count"); return RT_ERR_ARG; }
  int present = 0;
  int rc = rt_device_count(&present);
  if (rc != RT_OK) { rth_set_error(rt_last_error()); return rc; }
  if (present <= 0) { rth_set_error("no GPU present"); return RT_ERR_HIP; }
  int world = opts->n_devices;
  dev_job *jobs = (dev_job *)calloc((size_t)world, sizeof *jobs);
  pthread_t *tid = (pthread_t *)calloc((size_t)world, sizeof *tid);
  if (!jobs || !tid) { free(jobs); free(tid); return RT_ERR_NOMEM; }
  for (int r = 0; r < world; r++) {
    jobs[r].cam = cam; jobs[r].spheres = spheres; jobs[r].n = n; jobs[r].opts = opts;
    jobs[r].rank = r; jobs[r].world = world;
    jobs[r].device = (opts->device_ids ? opts->device_ids[r] : r) % present;
    jobs[r].image = rgb8;
  }
  int started = 0;
  for (int r = 0; r < world; r++) {
    if (pthread_create(&tid[r], NULL, device_thread, &jobs[r]) != 0) { rc = RT_ERR_NOMEM; break; }
    started++;
  }
  if (rc != RT_OK) rth_set_error("could not start a device thread");
  float worst = 0.0f;
  for (int r = 0; r < started; r++) {
    pthread_join(tid[r], NULL);
    if (jobs[r].rc != RT_OK && rc == RT_OK) {
      rc = jobs[r].rc;
      rth_set_error(jobs[r].err);       /* the cause, from the thread that saw it */
    }
    if (jobs[r].kernel_ms > worst) worst = jobs[r].kernel_ms;
  }
  if (kernel_ms_max) *kernel_ms_max = worst;
  if (devices_used) {                   /* distinct devices: ranks beyond the devices present wrap onto them */
    int distinct = 0;
    for (int r = 0; r < world; r++) {
      int seen = 0;
      for (int q = 0; q < r; q++) seen |= jobs[q].device == jobs[r].device;
      distinct += !seen;
    }
    *devices_used = distinct;
  }
  free(jobs); free(tid);
  return rc;
}

int rth_raytrace(const rt_camera *cam, const rt_sphere *spheres, size_t n, const rth_render_opts *opts, uint8_t *rgb8,
                 float *kernel_ms_max) {
  return rth_raytrace_ex(cam, spheres, n, opts, rgb8, kernel_ms_max, NULL);
}
