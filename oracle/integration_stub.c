/*
 * integration_stub.c -- the binding a maintainer of the reference would add to c/main.c to render on
 * GPUs: render_gpu() has the signature of render() (c/main.c:446) and is handed to pthread_create in its
 * place.  oracle/Makefile (target `integration`) splices this text into the reference translation unit
 * in front of raytrace() (c/main.c:485), adds the three new render_arg fields (after c/main.c:443), sets
 * them where raytrace() fills its render_arg (after c/main.c:525) and swaps &render for &render_gpu
 * (c/main.c:534); the result, oracle/_ref/ray_ref_gpu, is the reference's own command line, scene builder,
 * camera and PPM writer around libraytrace_amd.so.  TEST INFRASTRUCTURE: it proves the drop-in compiles
 * and runs; nothing in the product path uses it.  The integration guide shows this same text (a test compares them).
 */
#include "rt_amd.h"

static int rt_gpu_count(void) {
  int n = 0;
  return (rt_device_count(&n) == RT_OK && n > 0) ? n : 1;
}

static void* render_gpu(void* arg0) {
  render_arg* arg = (render_arg*) arg0;
  if (arg->ny == 0) return (void*)0;      /* (-t above the row count gives empty bands, c/main.c:512-513) */
  rt_ctx* ctx = NULL;
  if (rt_ctx_create(arg->device, &ctx) != RT_OK) {
    fprintf(stderr, "GPU context failed: %s\n", rt_last_error());
    return (void*)1;
  }

  /* scene -> flat records (sphere + material, c/structs.c:16-36) */
  size_t n = arg->sc->nspheres;
  rt_sphere* flat = calloc(n, sizeof *flat);
  for (size_t i = 0; i < n; i++) {
    sphere* s = &arg->sc->spheres[i];
    flat[i].center[0] = s->center.x; flat[i].center[1] = s->center.y; flat[i].center[2] = s->center.z;
    flat[i].radius = s->radius;
    flat[i].type = (int32_t)s->mat.type;
    if (s->mat.type == MATTE)      { v3 a = s->mat.matte.albedo; memcpy(flat[i].albedo, &a, 12); }
    else if (s->mat.type == METAL) { v3 a = s->mat.metal.albedo; memcpy(flat[i].albedo, &a, 12);
                                     flat[i].param = s->mat.metal.fuzz; }
    else                           { flat[i].param = s->mat.dielectric.ref_idx; }
  }
  int rc = rt_scene_upload(ctx, flat, n);
  free(flat);

  _Static_assert(sizeof(rt_camera) == sizeof(camera), "camera: same 22 floats in the same order (c/structs.c:43-47)");
  _Static_assert(sizeof(rt_sphere) == sizeof(sphere), "sphere record: the same 36 bytes (c/structs.c:16-36)");
  rt_camera cam;
  memcpy(&cam, arg->cam, sizeof cam);
  rt_job job = { .nx = (uint32_t)arg->nx, .ny = (uint32_t)arg->ny_total, .row0 = (uint32_t)arg->row0,
                 .tile_rows = (uint32_t)arg->ny, .tile_stride = (uint32_t)arg->ny, .n_tiles = 1,
                 .nsamples = (uint32_t)arg->nsamples, .max_depth = 50, .tmin = (float)0.001,
                 .seed = 0, .kernel = RT_KERNEL_MEGA };
  assert(arg->len == 3 * arg->nx * arg->ny);                     /* c/main.c:456 */
  if (rc == RT_OK) rc = rt_render_host(ctx, &cam, &job, arg->buf, NULL, NULL, NULL);
  if (rc != RT_OK) fprintf(stderr, "GPU render failed: %s\n", rt_last_error());
  rt_ctx_destroy(ctx);
  return (void*)(intptr_t)rc;            /* raytrace() asserts 0 after join, c/main.c:544 */
}
