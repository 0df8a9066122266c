/*
 * scene.c -- scene builders, camera set-up and the scene-building generator.
 * Host-only inputs of the GPU path.  Build with -ffp-contract=off: the
 * values must be the ones the reference's own expressions give, and several
 * of those are double expressions narrowed to float (noted line by line).
 */
#include "rt_host.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- generator, c/rand.c ---- */

void rth_rng_init(rth_rng *g) { g->state = 123456789u; }

uint32_t rth_rng_hash(uint32_t index) {
  index = (index << 13) ^ index;                                              /* c/rand.c:4 */
  uint32_t poly = index * index * 15731u + 789221u;
  return (index * poly + 1376312589u) & 0x7fffffffu;                          /* c/rand.c:5 */
}

double rth_rng_next(rth_rng *g) {
  uint32_t r = rth_rng_hash(g->state);
  g->state = r + 1u;                                                          /* c/rand.c:14 */
  return (double)r / 2147483647.0;                                            /* c/rand.c:15 */
}

/* ---- camera, c/structs.c:49-79 ---- */

static void sub3(const float a[3], const float b[3], float o[3]) { for (int i = 0; i < 3; i++) o[i] = a[i] - b[i]; }
static void cross3(const float a[3], const float b[3], float o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static float norm3(const float a[3]) { return sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
/* c/vec.c:41-46: scale by (float)(1.0/norm) */
static void unit3(const float a[3], float o[3]) {
  float k = (float)(1.0 / (double)norm3(a));
  for (int i = 0; i < 3; i++) o[i] = k * a[i];
}

void rth_camera_new(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov, float aspect,
                    float aperture, float focus_dist, rt_camera *out) {
  float theta = (float)((double)vfov * M_PI / 180.0);    /* double product and quotient, c/structs.c:56 */
  float hh = (float)tan((double)(theta / 2.0f));          /* c/structs.c:57 */
  float hw = aspect * hh;
  float d[3], w[3], uc[3], u[3], v[3];
  sub3(lookfrom, lookat, d);
  unit3(d, w);
  cross3(vup, w, uc);
  unit3(uc, u);
  cross3(w, u, v);
  float ku = hw * focus_dist, kv = hh * focus_dist;
  for (int i = 0; i < 3; i++) {
    out->llc[i] = ((lookfrom[i] - ku * u[i]) - kv * v[i]) - focus_dist * w[i];     /* c/structs.c:62-68 */
    out->horiz[i] = (2.0f * hw * focus_dist) * u[i];
    out->vert[i] = (2.0f * hh * focus_dist) * v[i];
    out->origin[i] = lookfrom[i];
    out->u[i] = u[i];
    out->v[i] = v[i];
    out->w[i] = w[i];
  }
  out->lens_radius = aperture / 2.0f;
}

void rth_camera_default(uint32_t nx, uint32_t ny, rt_camera *out) {
  const float from[3] = {10.0f, 2.5f, 5.0f}, at[3] = {-4.0f, 0.0f, -2.0f}, up[3] = {0.0f, 1.0f, 0.0f};
  float d[3];
  sub3(from, at, d);
  rth_camera_new(from, at, up, 20.0f, (float)nx / (float)ny, (float)0.05, norm3(d), out);
}

/* ---- scenes ---- */

static rt_sphere mk(float x, float y, float z, float r, int type, float ar, float ag, float ab, float param) {
  rt_sphere s;
  s.center[0] = x; s.center[1] = y; s.center[2] = z;
  s.radius = r;
  s.type = type;
  s.albedo[0] = ar; s.albedo[1] = ag; s.albedo[2] = ab;
  s.param = param;
  return s;
}

size_t rth_scene_small(rt_sphere *out, size_t cap) {
  const size_t need = 3 + 360 / 15;                                           /* c/main.c:371 */
  if (!out) return need;
  if (cap < need) return 0;
  size_t k = 0;
  out[k++] = mk(0, -1000, 0, 1000, RT_MATTE, (float)0.88, (float)0.96, (float)0.7, 0);
  out[k++] = mk(1.5f, 1, 0, 1, RT_DIELECTRIC, 0, 0, 0, 1.5f);
  out[k++] = mk(-1.5f, 1, 0, 1, RT_METAL, (float)0.8, (float)0.9, (float)0.8, 0.0f);
  for (int deg = 0; deg < 360; deg += 15) {
    float x = (float)sin(deg * M_PI / 180.0);                                 /* c/main.c:395 */
    float z = (float)cos(deg * M_PI / 180.0);
    float prod = x * z;
    float r1 = (float)(0.33 + (double)(prod / 9.0f));                         /* c/main.c:398 */
    float g = (float)(0.5 + (double)(prod / 2.0f));                           /* c/main.c:402 */
    out[k++] = mk(3.0f * x, r1, 3.0f * z, r1, RT_MATTE, x, g, z, 0);
  }
  return k;
}

size_t rth_scene_trivial(rt_sphere *out, size_t cap) {
  if (!out) return 2;
  if (cap < 2) return 0;
  out[0] = mk(0, -1000, 0, 1000, RT_MATTE, (float)0.88, (float)0.96, (float)0.7, 0);
  out[1] = mk(0, 1, 0, 1, RT_METAL, (float)0.8, (float)0.9, (float)0.8, (float)0.4);
  return 2;
}

size_t rth_scene_random(rt_sphere *out, size_t cap, int half_extent, rth_rng *g) {
  const size_t side = (size_t)(2 * half_extent);
  const size_t need = side * side + 4;
  if (!out) return need;
  if (cap < need || !g) return 0;
  size_t k = 0;
  out[k++] = mk(0, -1000, 0, 1000, RT_MATTE, (float)0.7, (float)0.7, (float)0.7, 0);
  for (int a = -half_extent; a < half_extent; a++) {
    for (int b = -half_extent; b < half_extent; b++) {
      /* draws in the order of c/main.c:298-317 */
      float pick = (float)rth_rng_next(g);
      float x = (float)(a + 0.9 * rth_rng_next(g));
      float z = (float)(b + 0.9 * rth_rng_next(g));
      const float y = (float)0.2, r = (float)0.2;
      if (pick < 0.8) {
        double p[6];
        for (int i = 0; i < 6; i++) p[i] = rth_rng_next(g);
        out[k++] = mk(x, y, z, r, RT_MATTE, (float)(p[0] * p[1]), (float)(p[2] * p[3]), (float)(p[4] * p[5]), 0);
      } else if (pick < 0.95) {
        double p[4];
        for (int i = 0; i < 4; i++) p[i] = rth_rng_next(g);
        out[k++] = mk(x, y, z, r, RT_METAL, (float)(0.5 * (1 + p[0])), (float)(0.5 * (1 + p[1])),
                      (float)(0.5 * (1 + p[2])), (float)(0.5 * p[3]));
      } else {
        out[k++] = mk(x, y, z, r, RT_DIELECTRIC, 0, 0, 0, 1.5f);
      }
    }
  }
  out[k++] = mk(0, 1, 0, 1, RT_DIELECTRIC, 0, 0, 0, 1.5f);                                   /* c/main.c:328-332 */
  out[k++] = mk(-4, 1, 0, 1, RT_MATTE, (float)0.2, (float)0.4, (float)0.7, 0);               /* c/main.c:333-337 */
  out[k++] = mk(4, 1, 0, 1, RT_METAL, (float)0.6, (float)0.6, (float)0.5, 0);                /* c/main.c:338-342 */
  return k;
}

size_t rth_scene_glass(rt_sphere *out, size_t cap, int half_extent, rth_rng *g) {
  size_t n = rth_scene_random(out, cap, half_extent, g);
  if (!out) return n;
  for (size_t i = 1; i < n; i++) {       /* sphere 0, the ground, stays matte so paths end */
    if (out[i].type == RT_MATTE) {
      out[i] = mk(out[i].center[0], out[i].center[1], out[i].center[2], out[i].radius, RT_DIELECTRIC, 0, 0, 0, 1.5f);
    } else if (out[i].type == RT_METAL) {
      out[i].param = 0.0f;
    }
  }
  return n;
}

int rth_scene_build(int which, rt_sphere **out, size_t *n) {
  if (!out || !n) return RT_ERR_ARG;
  rth_rng g;
  rth_rng_init(&g);
  size_t need;
  switch (which) {
  case RTH_SCENE_SMALL: need = rth_scene_small(NULL, 0); break;
  case RTH_SCENE_TRIVIAL: need = rth_scene_trivial(NULL, 0); break;
  case RTH_SCENE_FINAL: need = rth_scene_random(NULL, 0, 11, &g); break;
  case RTH_SCENE_GLASS: need = rth_scene_glass(NULL, 0, 11, &g); break;
  case RTH_SCENE_DENSE: need = rth_scene_random(NULL, 0, 100, &g); break;
  default: return RT_ERR_ARG;
  }
  rt_sphere *s = (rt_sphere *)malloc(need * sizeof *s);
  if (!s) return RT_ERR_NOMEM;
  size_t got;
  switch (which) {
  case RTH_SCENE_SMALL: got = rth_scene_small(s, need); break;
  case RTH_SCENE_TRIVIAL: got = rth_scene_trivial(s, need); break;
  case RTH_SCENE_FINAL: got = rth_scene_random(s, need, 11, &g); break;
  case RTH_SCENE_GLASS: got = rth_scene_glass(s, need, 11, &g); break;
  default: got = rth_scene_random(s, need, 100, &g); break;
  }
  if (got == 0) { free(s); return RT_ERR_ARG; }
  *out = s;
  *n = got;
  return RT_OK;
}

int rth_write_ppm(FILE *fp, const uint8_t *rgb8, size_t w, size_t h) {
  if (!fp || !rgb8) return RT_ERR_ARG;
  if (fprintf(fp, "P6\n%zu %zu 255\n", w, h) < 0) return RT_ERR_ARG;          /* c/main.c:412 */
  fflush(fp);
  if (w != 0 && h != 0 && fwrite(rgb8, 3 * w * h, 1, fp) != 1) return RT_ERR_ARG;
  return RT_OK;
}
