/*
 * service.c -- a resident render worker: the scene stays on the GPU, a new
 * camera re-renders without another upload, and the picture refines while the
 * camera rests.
 *
 * The reference's interactive front end keeps one worker thread that receives
 * (arguments, reply channel) jobs and answers each with a finished image
 * (rs/src/gui.rs:164-172); the window sends a job when the user asks for a
 * render and shows the image when it arrives (rs/src/gui.rs:340-357).  Here
 * the job slot holds one pending request (a newer camera replaces an older
 * one that has not started), and a request is rendered as a sequence of
 * progressive passes (RT_FLAG_PROGRESSIVE of include/rt_amd.h): pass k renders
 * samples [k*m, (k+1)*m) and its output is the mean over all samples so far,
 * bit for bit what one render of (k+1)*m samples gives.
 */
#include "rt_host.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

struct rth_service {
  pthread_t thread;
  pthread_mutex_t lock;
  pthread_cond_t wake;       /* a request arrived, or stop */
  pthread_cond_t published;  /* a pass was published, or a request failed */
  int device;
  rt_ctx *ctx;
  int stop;
  /* pending request (id 0: none) */
  uint64_t next_id, pending_id;
  rt_camera pending_cam;
  rth_service_request pending;
  /* published picture */
  uint8_t *image;
  size_t image_cap;
  uint64_t image_id;
  uint32_t image_samples, image_nx, image_ny;
  /* outcome of the newest request the worker has taken up */
  uint64_t active_id;
  uint32_t active_samples;
  uint32_t active_target;    /* samples that request ends at */
  int active_rc;
  char err[256];
  uint64_t uploads, passes;
};

static void *service_worker(void *arg) {
  rth_service *s = (rth_service *)arg;
  uint8_t *tile = NULL;
  size_t tile_cap = 0;
  pthread_mutex_lock(&s->lock);
  for (;;) {
    while (!s->stop && s->pending_id == 0) pthread_cond_wait(&s->wake, &s->lock);
    if (s->stop) break;
    /* take the request up */
    const uint64_t id = s->pending_id;
    const rt_camera cam = s->pending_cam;
    const rth_service_request rq = s->pending;
    s->pending_id = 0;
    s->active_id = id;
    s->active_samples = 0;
    s->active_target = rq.target_samples;
    s->active_rc = RT_OK;
    pthread_mutex_unlock(&s->lock);

    const size_t bytes = (size_t)3 * rq.nx * rq.ny;
    int rc = RT_OK;
    if (bytes > tile_cap) {
      free(tile);
      tile = (uint8_t *)malloc(bytes);
      tile_cap = tile ? bytes : 0;
      if (!tile) rc = RT_ERR_NOMEM;
    }
    uint32_t done = 0;
    while (rc == RT_OK && done < rq.target_samples) {
      rt_job job;
      memset(&job, 0, sizeof job);
      job.nx = rq.nx;
      job.ny = rq.ny;
      job.tile_rows = rq.ny;
      job.tile_stride = rq.ny;
      job.n_tiles = 1;
      job.nsamples = rq.target_samples - done < rq.pass_samples ? rq.target_samples - done : rq.pass_samples;
      job.max_depth = rq.max_depth;
      job.tmin = (float)0.001;                                   /* c/main.c:264 */
      job.seed = rq.seed;
      job.kernel = RT_KERNEL_MEGA;
      job.flags = RT_FLAG_PROGRESSIVE;
      job.sample0 = done;
      rc = rt_render_host(s->ctx, &cam, &job, tile, NULL, NULL, NULL);
      pthread_mutex_lock(&s->lock);
      if (rc == RT_OK) {
        done += job.nsamples;
        s->passes++;
        if (bytes > s->image_cap) {
          free(s->image);
          s->image = (uint8_t *)malloc(bytes);
          s->image_cap = s->image ? bytes : 0;
        }
        if (s->image) {
          memcpy(s->image, tile, bytes);
          s->image_id = id;
          s->image_samples = done;
          s->image_nx = rq.nx;
          s->image_ny = rq.ny;
        } else {
          rc = RT_ERR_NOMEM;
        }
        s->active_samples = done;
      }
      const int superseded = s->pending_id != 0 || s->stop;
      if (rc != RT_OK) {
        s->active_rc = rc;
        strncpy(s->err, rc == RT_ERR_NOMEM ? "out of host memory" : rt_last_error(), sizeof s->err - 1);
        s->err[sizeof s->err - 1] = 0;
      }
      pthread_cond_broadcast(&s->published);
      pthread_mutex_unlock(&s->lock);
      if (superseded) break;                                     /* the newest camera wins */
    }
    pthread_mutex_lock(&s->lock);
    if (rc != RT_OK && s->active_rc == RT_OK) {                  /* (failed before the first pass) */
      s->active_rc = rc;
      strncpy(s->err, "out of host memory", sizeof s->err - 1);
      pthread_cond_broadcast(&s->published);
    }
  }
  pthread_mutex_unlock(&s->lock);
  free(tile);
  return NULL;
}

int rth_service_start(int device, const rt_sphere *spheres, size_t n, rth_service **out) {
  if (!out || !spheres || n == 0) { rth_set_error("null or empty argument"); return RT_ERR_ARG; }
  *out = NULL;
  rth_service *s = (rth_service *)calloc(1, sizeof *s);
  if (!s) { rth_set_error("out of host memory"); return RT_ERR_NOMEM; }
  s->device = device;
  int rc = rt_ctx_create(device, &s->ctx);
  if (rc == RT_OK) rc = rt_scene_upload(s->ctx, spheres, n);       /* once: the scene stays resident */
  if (rc != RT_OK) {
    rth_set_error(rt_last_error());
    if (s->ctx) rt_ctx_destroy(s->ctx);
    free(s);
    return rc;
  }
  s->uploads = 1;
  s->next_id = 1;
  pthread_mutex_init(&s->lock, NULL);
  pthread_cond_init(&s->wake, NULL);
  pthread_cond_init(&s->published, NULL);
  if (pthread_create(&s->thread, NULL, service_worker, s) != 0) {
    rth_set_error("could not start the worker thread");
    rt_ctx_destroy(s->ctx);
    free(s);
    return RT_ERR_NOMEM;
  }
  *out = s;
  return RT_OK;
}

uint64_t rth_service_submit(rth_service *s, const rt_camera *cam, const rth_service_request *rq) {
  if (!s || !cam || !rq || rq->nx == 0 || rq->ny == 0 || rq->pass_samples == 0 || rq->target_samples == 0 || rq->max_depth == 0)
    return 0;
  pthread_mutex_lock(&s->lock);
  const uint64_t id = s->next_id++;
  s->pending_id = id;                    /* replaces a request that has not started */
  s->pending_cam = *cam;
  s->pending = *rq;
  pthread_cond_signal(&s->wake);
  pthread_mutex_unlock(&s->lock);
  return id;
}

int rth_service_latest(rth_service *s, uint8_t *rgb8, size_t cap, uint64_t *id, uint32_t *samples, uint32_t *nx, uint32_t *ny) {
  if (!s || !rgb8) return RT_ERR_ARG;
  pthread_mutex_lock(&s->lock);
  int rc = RT_OK;
  const size_t bytes = (size_t)3 * s->image_nx * s->image_ny;
  if (s->image_id == 0 || cap < bytes) {
    rc = RT_ERR_ARG;
  } else {
    memcpy(rgb8, s->image, bytes);
    if (id) *id = s->image_id;
    if (samples) *samples = s->image_samples;
    if (nx) *nx = s->image_nx;
    if (ny) *ny = s->image_ny;
  }
  pthread_mutex_unlock(&s->lock);
  return rc;
}

int rth_service_wait(rth_service *s, uint64_t id, uint32_t min_samples) {
  if (!s || id == 0) return RT_ERR_ARG;
  pthread_mutex_lock(&s->lock);
  int rc = RT_OK;
  for (;;) {
    if (s->active_id == id && s->active_rc != RT_OK) { rc = s->active_rc; rth_set_error(s->err); break; }
    /* (a request ends at its target: asking for more samples than that waits for its last pass) */
    if (s->active_id == id && s->active_samples >= (min_samples < s->active_target ? min_samples : s->active_target)) break;
    if (s->active_id > id || (s->pending_id > id)) { rc = RT_ERR_ARG; rth_set_error("the request was superseded by a newer one"); break; }
    if (s->stop) { rc = RT_ERR_ARG; rth_set_error("the service is stopping"); break; }
    pthread_cond_wait(&s->published, &s->lock);
  }
  pthread_mutex_unlock(&s->lock);
  return rc;
}

void rth_service_stats(rth_service *s, uint64_t *uploads, uint64_t *passes) {
  if (!s) return;
  pthread_mutex_lock(&s->lock);
  if (uploads) *uploads = s->uploads;
  if (passes) *passes = s->passes;
  pthread_mutex_unlock(&s->lock);
}

int rth_service_stop(rth_service *s) {
  if (!s) return RT_OK;
  pthread_mutex_lock(&s->lock);
  s->stop = 1;
  pthread_cond_broadcast(&s->wake);
  pthread_cond_broadcast(&s->published);
  pthread_mutex_unlock(&s->lock);
  pthread_join(s->thread, NULL);
  rt_ctx_destroy(s->ctx);
  free(s->image);
  pthread_mutex_destroy(&s->lock);
  pthread_cond_destroy(&s->wake);
  pthread_cond_destroy(&s->published);
  free(s);
  return RT_OK;
}
