/*
 * rt_oracle.c -- CPU restatement of the reference C renderer's hot path.
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  Build with -ffp-contract=off:
 * every float/double operation below is written in the order and at the
 * precision C's usual arithmetic conversions give the reference expression
 * it cites, and must not be fused.
 *
 * Parity pin: tests/test_oracle_golden.py (sha256 of whole images against the
 * reference binary's, RNG known answers, camera constants, scene dump) and
 * tests/test_oracle_vs_ref.py (function by function against the compiled
 * reference, when oracle/_ref exists).
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* vectors: c/vec.c:1-62.  Plain float triples, all ops single float. */

typedef struct { float x, y, z; } f3;

static inline f3 f3_make(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 f3_ld(const float *p) { return f3_make(p[0], p[1], p[2]); }
static inline void f3_st(float *p, f3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline f3 f3_add(f3 a, f3 b) { return f3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 f3_sub(f3 a, f3 b) { return f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 f3_mul(f3 a, f3 b) { return f3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 f3_scale(float k, f3 a) { return f3_make(k * a.x, k * a.y, k * a.z); }
static inline f3 f3_div(f3 a, float k) { return f3_make(a.x / k, a.y / k, a.z / k); }
static inline f3 f3_negate(f3 a) { return f3_make(-a.x, -a.y, -a.z); }
/* c/vec.c:48-50: (x*x + y*y) + z*z, three products and two sums in float */
static inline float f3_dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* c/vec.c:37-39 */
static inline float f3_len(f3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
/* c/vec.c:41-46: the reciprocal is a double divide narrowed to float, then
 * three float products (the assert on a zero norm is not restated). */
static inline f3 f3_unit(f3 a) {
  float len = f3_len(a);
  float k = (float)(1.0 / (double)len);
  return f3_scale(k, a);
}
/* c/vec.c:52-58 */
static inline f3 f3_cross(f3 a, f3 b) {
  return f3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* c/vec.c:60-62: u - (2*dot(u,n))*n */
static inline f3 f3_reflect(f3 u, f3 n) {
  float k = 2.0f * f3_dot(u, n);
  return f3_sub(u, f3_scale(k, n));
}

/* ------------------------------------------------------------------ */
/* random numbers */

/* c/rand.c:3-6 (wrapping uint32 arithmetic) */
uint32_t orc_ref_hash(uint32_t index) {
  uint32_t i = (index << 13) ^ index;
  uint32_t inner = i * i * 15731u + 789221u;
  return (i * inner + 1376312589u) & 0x7fffffffu;
}

/* c/rand.c:8-16 with the thread-local seed made explicit */
double orc_ref_next(uint32_t *state) {
  uint32_t r = orc_ref_hash(*state);
  *state = r + 1u;
  return (double)r / (double)0x7fffffff;
}

/* Counter generator of the deterministic mode.  Build-defined (the
 * reference has only the sequential chain above): the per-path key is a
 * 32-bit avalanche hash of (seed, pixel, sample); draw k is the top 23 bits of
 * a two-round multiply-xorshift hash of key + k*golden, taken as the 31-bit
 * integer r with its low 8 bits zero and mapped to [0,1) by the same double
 * division as c/rand.c:15. */
uint32_t orc_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
uint32_t orc_path_key(uint32_t seed, uint32_t pixel_index, uint32_t sample) {
  uint32_t a = orc_mix32(pixel_index + 0x9e3779b9u * seed);
  return orc_mix32(a ^ (sample * 0x85ebca6bu + 0x165667b1u));
}
uint32_t orc_draw_hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 16; x *= 0x846ca68bu;
  return x;
}
uint32_t orc_ctr_bits(uint32_t key, uint32_t k) {
  return (orc_draw_hash(key + k * 0x9e3779b9u) >> 9) << 8;
}
double orc_ctr_uniform(uint32_t key, uint32_t k) {
  return (double)orc_ctr_bits(key, k) / (double)0x7fffffff;
}

typedef struct {
  int counter_mode;
  uint32_t state;       /* sequential chain */
  uint32_t key, k;      /* counter stream   */
  uint64_t draws;
} rng_t;

static inline double rng_next(rng_t *g) {
  g->draws++;
  if (g->counter_mode) return orc_ctr_uniform(g->key, g->k++);
  return orc_ref_next(&g->state);
}

/* c/vec.c:64-72.  Three draws per attempt in x, y, z order, each narrowed to
 * float before 2*u-1; accept when FLT_EPSILON <= |u| < 1. */
static f3 unit_ball(rng_t *g) {
  for (;;) {
    float ux = (float)rng_next(g);
    float uy = (float)rng_next(g);
    float uz = (float)rng_next(g);
    f3 u = f3_sub(f3_scale(2.0f, f3_make(ux, uy, uz)), f3_make(1.0f, 1.0f, 1.0f));
    float len = f3_len(u);
    if (!((double)len >= 1.0 || fabsf(len) < FLT_EPSILON)) return u;
  }
}

/* ------------------------------------------------------------------ */
/* camera: c/structs.c:49-92 */

void orc_camera_make(const float lookfrom[3], const float lookat[3],
                     const float vup[3], float vfov, float aspect,
                     float aperture, float focus_dist, orc_camera *out) {
  f3 from = f3_ld(lookfrom), at = f3_ld(lookat), up = f3_ld(vup);
  float theta = (float)((double)vfov * M_PI / 180.0);          /* structs.c:56 */
  float half_h = (float)tan((double)(theta / 2.0f));           /* structs.c:57 */
  float half_w = aspect * half_h;                              /* structs.c:58 */
  f3 w = f3_unit(f3_sub(from, at));
  f3 u = f3_unit(f3_cross(up, w));
  f3 v = f3_cross(w, u);
  f3 llc = f3_sub(f3_sub(f3_sub(from, f3_scale(half_w * focus_dist, u)),
                         f3_scale(half_h * focus_dist, v)),
                  f3_scale(focus_dist, w));                     /* structs.c:62-68 */
  f3_st(out->llc, llc);
  f3_st(out->horiz, f3_scale(2.0f * half_w * focus_dist, u));   /* structs.c:71 */
  f3_st(out->vert, f3_scale(2.0f * half_h * focus_dist, v));    /* structs.c:72 */
  f3_st(out->origin, from);
  f3_st(out->u, u);
  f3_st(out->v, v);
  f3_st(out->w, w);
  out->lens_radius = aperture / 2.0f;                           /* structs.c:77 */
}

/* c/main.c:490-497 */
void orc_camera_default(size_t nx, size_t ny, orc_camera *out) {
  float from[3] = {10.0f, 2.5f, 5.0f}, at[3] = {-4.0f, 0.0f, -2.0f}, up[3] = {0.0f, 1.0f, 0.0f};
  float focus = f3_len(f3_sub(f3_ld(from), f3_ld(at)));
  float aperture = (float)0.05;
  orc_camera_make(from, at, up, 20.0f, (float)nx / (float)ny, aperture, focus, out);
}

typedef struct { f3 A, B; } ray_t;

/* c/structs.c:81-92 */
static ray_t camera_ray(const orc_camera *c, float x, float y, rng_t *g) {
  f3 rd = f3_scale(c->lens_radius, unit_ball(g));
  f3 off = f3_add(f3_scale(rd.x, f3_ld(c->u)), f3_scale(rd.y, f3_ld(c->v)));
  f3 on_plane = f3_add(f3_ld(c->llc),
                       f3_add(f3_scale(x, f3_ld(c->horiz)), f3_scale(y, f3_ld(c->vert))));
  ray_t r;
  r.B = f3_sub(f3_sub(on_plane, f3_ld(c->origin)), off);
  r.A = f3_add(f3_ld(c->origin), off);
  return r;
}

/* ------------------------------------------------------------------ */
/* scenes */

static void put_sphere(orc_sphere *s, float cx, float cy, float cz, float r,
                       int kind, float ar, float ag, float ab, float param) {
  s->center[0] = cx; s->center[1] = cy; s->center[2] = cz;
  s->radius = r; s->kind = kind;
  s->albedo[0] = ar; s->albedo[1] = ag; s->albedo[2] = ab;
  s->param = param;
}

/* c/main.c:369-408 */
size_t orc_scene_small(orc_sphere *out, size_t cap) {
  size_t need = 3 + 360 / 15;
  if (!out) return need;
  if (cap < need) return 0;
  put_sphere(&out[0], 0.0f, -1000.0f, 0.0f, 1000.0f, ORC_MATTE, (float)0.88, (float)0.96, (float)0.7, 0.0f);
  put_sphere(&out[1], 1.5f, 1.0f, 0.0f, 1.0f, ORC_GLASS, 0.0f, 0.0f, 0.0f, 1.5f);
  put_sphere(&out[2], -1.5f, 1.0f, 0.0f, 1.0f, ORC_METAL, (float)0.8, (float)0.9, (float)0.8, 0.0f);
  size_t i = 3;
  for (int deg = 0; deg < 360; deg += 15) {
    float x = (float)sin(deg * M_PI / 180.0);
    float z = (float)cos(deg * M_PI / 180.0);
    float R0 = 3.0f;
    float R1 = (float)(0.33 + (double)(x * z / 9.0f));          /* main.c:398 */
    float green = (float)(0.5 + (double)(x * z / 2.0f));         /* main.c:402 */
    put_sphere(&out[i++], R0 * x, R1, R0 * z, R1, ORC_MATTE, x, green, z, 0.0f);
  }
  return i;
}

/* c/main.c:348-363 */
size_t orc_scene_trivial(orc_sphere *out, size_t cap) {
  if (!out) return 2;
  if (cap < 2) return 0;
  put_sphere(&out[0], 0.0f, -1000.0f, 0.0f, 1000.0f, ORC_MATTE, (float)0.88, (float)0.96, (float)0.7, 0.0f);
  put_sphere(&out[1], 0.0f, 1.0f, 0.0f, 1.0f, ORC_METAL, (float)0.8, (float)0.9, (float)0.8, (float)0.4);
  return 2;
}

/* c/main.c:286-344.  Draw order is the left-to-right order gcc and clang
 * both give the reference's initialiser lists: material choice, centre x,
 * centre z, then the material's own draws. */
size_t orc_scene_random(orc_sphere *out, size_t cap, int half_extent,
                        uint32_t *rng_state) {
  size_t side = (size_t)(2 * half_extent);
  size_t need = 1 + side * side + 3;
  if (!out) return need;
  if (cap < need) return 0;
  size_t i = 0;
  put_sphere(&out[i++], 0.0f, -1000.0f, 0.0f, 1000.0f, ORC_MATTE, (float)0.7, (float)0.7, (float)0.7, 0.0f);
  for (int a = -half_extent; a < half_extent; a++) {
    for (int b = -half_extent; b < half_extent; b++) {
      float choose = (float)orc_ref_next(rng_state);                       /* main.c:298 */
      float cx = (float)((double)a + 0.9 * orc_ref_next(rng_state));       /* main.c:299 */
      float cz = (float)((double)b + 0.9 * orc_ref_next(rng_state));
      float cy = (float)0.2, rad = (float)0.2;
      if ((double)choose < 0.8) {
        double r0 = orc_ref_next(rng_state), r1 = orc_ref_next(rng_state);
        double g0 = orc_ref_next(rng_state), g1 = orc_ref_next(rng_state);
        double b0 = orc_ref_next(rng_state), b1 = orc_ref_next(rng_state);
        put_sphere(&out[i++], cx, cy, cz, rad, ORC_MATTE,
                   (float)(r0 * r1), (float)(g0 * g1), (float)(b0 * b1), 0.0f);
      } else if ((double)choose < 0.95) {
        double r = orc_ref_next(rng_state), g = orc_ref_next(rng_state);
        double bl = orc_ref_next(rng_state), fz = orc_ref_next(rng_state);
        put_sphere(&out[i++], cx, cy, cz, rad, ORC_METAL,
                   (float)(0.5 * (1 + r)), (float)(0.5 * (1 + g)), (float)(0.5 * (1 + bl)),
                   (float)(0.5 * fz));
      } else {
        put_sphere(&out[i++], cx, cy, cz, rad, ORC_GLASS, 0.0f, 0.0f, 0.0f, 1.5f);
      }
    }
  }
  put_sphere(&out[i++], 0.0f, 1.0f, 0.0f, 1.0f, ORC_GLASS, 0.0f, 0.0f, 0.0f, 1.5f);
  put_sphere(&out[i++], -4.0f, 1.0f, 0.0f, 1.0f, ORC_MATTE, (float)0.2, (float)0.4, (float)0.7, 0.0f);
  put_sphere(&out[i++], 4.0f, 1.0f, 0.0f, 1.0f, ORC_METAL, (float)0.6, (float)0.6, (float)0.5, 0.0f);
  return i;
}

void orc_scene_glassify(orc_sphere *s, size_t n) {
  for (size_t i = 1; i < n; i++) {
    if (s[i].kind == ORC_MATTE) {
      s[i].kind = ORC_GLASS; s[i].param = 1.5f;
      s[i].albedo[0] = s[i].albedo[1] = s[i].albedo[2] = 0.0f;
    } else if (s[i].kind == ORC_METAL) {
      s[i].param = 0.0f;
    }
  }
}

/* ------------------------------------------------------------------ */
/* intersection: c/main.c:214-256 */

typedef struct { float t; f3 p, n; const orc_sphere *obj; } hit_t;

/* c/main.c:214-236.  The two roots are double expressions narrowed to float
 * (sqrt is the double libm call; -b and a are promoted). */
static int sphere_hit(const orc_sphere *s, const ray_t *r, float tmin, float tmax, hit_t *h) {
  f3 ctr = f3_ld(s->center);
  f3 oc = f3_sub(r->A, ctr);
  float a = f3_dot(r->B, r->B);
  float b = f3_dot(oc, r->B);
  float c = f3_dot(oc, oc) - s->radius * s->radius;
  float D = b * b - a * c;
  if (D > 0.0f) {
    double root = sqrt((double)D);
    float cand[2];
    cand[0] = (float)(((double)(-b) - root) / (double)a);
    cand[1] = (float)(((double)(-b) + root) / (double)a);
    for (int i = 0; i < 2; i++) {
      float t = cand[i];
      if (tmin < t && t < tmax) {
        h->t = t;
        h->p = f3_add(r->A, f3_scale(t, r->B));                 /* structs.c:6-8 */
        h->n = f3_div(f3_sub(h->p, ctr), s->radius);
        h->obj = s;
        return 1;
      }
    }
  }
  return 0;
}

/* c/main.c:239-256: in index order, tmax shrinking to the closest so far */
static int scene_hit(const orc_sphere *s, size_t n, const ray_t *r, float tmin,
                     float tmax, hit_t *h, uint64_t *tests) {
  int any = 0;
  float closest = tmax;
  hit_t tmp;
  for (size_t i = 0; i < n; i++) {
    if (sphere_hit(&s[i], r, tmin, closest, &tmp)) {
      any = 1;
      closest = tmp.t;
      *h = tmp;
    }
  }
  *tests += n;
  return any;
}

int orc_hit_sphere(const orc_sphere *s, const float A[3], const float B[3],
                   float tmin, float tmax, float *t, float p[3], float nrm[3]) {
  ray_t r = {f3_ld(A), f3_ld(B)};
  hit_t h;
  if (!sphere_hit(s, &r, tmin, tmax, &h)) return 0;
  *t = h.t; f3_st(p, h.p); f3_st(nrm, h.n);
  return 1;
}

int orc_hit_scene(const orc_sphere *s, size_t n, const float A[3],
                  const float B[3], float tmin, float tmax, float *t,
                  float p[3], float nrm[3]) {
  ray_t r = {f3_ld(A), f3_ld(B)};
  hit_t h;
  uint64_t tests = 0;
  if (!scene_hit(s, n, &r, tmin, tmax, &h, &tests)) return -1;
  *t = h.t; f3_st(p, h.p); f3_st(nrm, h.n);
  return (int)(h.obj - s);
}

/* orc_hit_scene for m rays (6 floats each: A, B), rays dealt to `threads` pthreads; index[i] = -1 for no hit,
 * t[i] written only on a hit.  The scan of each ray is scene_hit, c/main.c:239-258, unchanged. */
typedef struct {
  const orc_sphere *s; size_t n; const float *rays; size_t first, step, m; float tmin, tmax; int32_t *index; float *t;
} many_job;

static void *many_run(void *arg) {
  many_job *j = (many_job *)arg;
  for (size_t i = j->first; i < j->m; i += j->step) {
    ray_t r = {f3_ld(j->rays + 6 * i), f3_ld(j->rays + 6 * i + 3)};
    hit_t h;
    uint64_t tests = 0;
    if (scene_hit(j->s, j->n, &r, j->tmin, j->tmax, &h, &tests)) {
      j->index[i] = (int32_t)(h.obj - j->s);
      j->t[i] = h.t;
    } else {
      j->index[i] = -1;
    }
  }
  return NULL;
}

int orc_hit_scene_many(const orc_sphere *s, size_t n, const float *rays, size_t m, float tmin, float tmax,
                       int32_t *index, float *t, size_t threads) {
  if (threads == 0) threads = 1;
  pthread_t *tid = (pthread_t *)calloc(threads, sizeof *tid);
  many_job *jobs = (many_job *)calloc(threads, sizeof *jobs);
  if (!tid || !jobs) { free(tid); free(jobs); return -1; }
  for (size_t k = 0; k < threads; k++) {
    jobs[k] = (many_job){s, n, rays, k, threads, m, tmin, tmax, index, t};
    pthread_create(&tid[k], NULL, many_run, &jobs[k]);
  }
  for (size_t k = 0; k < threads; k++) pthread_join(tid[k], NULL);
  free(tid); free(jobs);
  return 0;
}

/* ------------------------------------------------------------------ */
/* materials: c/main.c:140-211 */

/* c/main.c:140-144: r0 in float; the polynomial in double, narrowed on return */
float orc_schlick(float cosine, float ref_idx) {
  float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
  r0 = r0 * r0;
  return (float)((double)r0 + (double)(1.0f - r0) * pow((double)(1.0f - cosine), 5.0));
}

/* c/main.c:146-158 */
static int refract_dir(f3 u, f3 n, float ni_over_nt, f3 *out) {
  f3 un = f3_unit(u);
  float dt = f3_dot(un, n);
  float D = (float)(1.0 - (double)(ni_over_nt * ni_over_nt * (1.0f - dt * dt)));
  if (D > 0.0f) {
    float rootD = (float)sqrt((double)D);
    *out = f3_sub(f3_scale(ni_over_nt, f3_sub(un, f3_scale(dt, n))), f3_scale(rootD, n));
    return 1;
  }
  return 0;
}

int orc_refract(const float u[3], const float n[3], float ni_over_nt, float out[3]) {
  f3 o;
  if (!refract_dir(f3_ld(u), f3_ld(n), ni_over_nt, &o)) return 0;
  f3_st(out, o);
  return 1;
}

/* c/main.c:163-211.  Returns 0 only for an absorbed metal ray. */
static int scatter(const ray_t *in, const hit_t *h, rng_t *g, f3 *atten, ray_t *out) {
  const orc_sphere *s = h->obj;
  switch (s->kind) {
  case ORC_MATTE: {
    f3 target = f3_add(f3_add(h->p, h->n), unit_ball(g));       /* main.c:169 */
    *atten = f3_ld(s->albedo);
    out->A = h->p;
    out->B = f3_sub(target, h->p);
    return 1;
  }
  case ORC_METAL: {
    f3 mirrored = f3_reflect(f3_unit(in->B), h->n);             /* main.c:175 */
    f3 dir = f3_add(mirrored, f3_scale(s->param, unit_ball(g)));
    *atten = f3_ld(s->albedo);
    out->A = h->p;
    out->B = dir;
    return f3_dot(dir, h->n) > 0.0f;
  }
  default: { /* ORC_GLASS, main.c:181-206 */
    float idx = s->param;
    f3 outward;
    float ni_over_nt, cosine;
    if (f3_dot(in->B, h->n) > 0.0f) {
      outward = f3_negate(h->n);
      ni_over_nt = idx;
      cosine = idx * f3_dot(in->B, h->n) / f3_len(in->B);
    } else {
      outward = h->n;
      ni_over_nt = (float)(1.0 / (double)idx);
      cosine = -f3_dot(in->B, h->n) / f3_len(in->B);
    }
    f3 bent;
    out->A = h->p;
    /* the uniform is drawn only when refraction is possible (&& short-circuit) */
    if (refract_dir(in->B, outward, ni_over_nt, &bent) &&
        rng_next(g) >= (double)orc_schlick(cosine, idx)) {
      out->B = bent;
    } else {
      out->B = f3_reflect(in->B, h->n);
    }
    *atten = f3_make(1.0f, 1.0f, 1.0f);
    return 1;
  }
  }
}

/* c/main.c:258-283.  An absorbed ray zeroes the colour and the loop goes on
 * with the unchanged ray, as the reference does. */
static f3 path_radiance(const ray_t *r0, const orc_sphere *s, size_t n, size_t max_depth,
                        rng_t *g, orc_stats *st, int *first_hit) {
  f3 color = f3_make(1.0f, 1.0f, 1.0f);
  ray_t r = *r0;
  const float tmin = (float)0.001;
  size_t depth;
  *first_hit = -1;
  for (depth = 0; depth < max_depth; depth++) {
    hit_t h;
    st->segments++;
    if (scene_hit(s, n, &r, tmin, FLT_MAX, &h, &st->tests)) {
      if (depth == 0) *first_hit = (int)(h.obj - s);
      ray_t next;
      f3 atten;
      if (scatter(&r, &h, g, &atten, &next)) {
        r = next;
        color = f3_mul(atten, color);
      } else {
        color = f3_make(0.0f, 0.0f, 0.0f);
        st->absorbed++;
      }
    } else {
      float t = (float)(0.5 * ((double)f3_unit(r.B).y + 1.0));   /* main.c:274 */
      float k = (float)(1.0 - (double)t);
      f3 white = f3_scale(k, f3_make(1.0f, 1.0f, 1.0f));
      f3 blue = f3_scale(t, f3_make((float)0.75, (float)0.95, (float)1.0));
      color = f3_mul(color, f3_add(white, blue));
      break;
    }
  }
  if (depth == max_depth) st->capped++;
  st->paths++;
  return color;
}

/* c/main.c:470-477 */
static void finish_pixel(f3 sum, size_t nsamples, uint8_t *rgb, float *linear) {
  f3 c = f3_div(sum, (float)nsamples);
  if (linear) f3_st(linear, c);
  c = f3_make((float)sqrt((double)c.x), (float)sqrt((double)c.y), (float)sqrt((double)c.z));
  if (isnan(c.x)) c.x = 0.0f;
  if (isnan(c.y)) c.y = 0.0f;
  if (isnan(c.z)) c.z = 0.0f;
  if (rgb) {
    rgb[0] = (uint8_t)(int)(255.0f * c.x);
    rgb[1] = (uint8_t)(int)(255.0f * c.y);
    rgb[2] = (uint8_t)(int)(255.0f * c.z);
  }
}

static void stats_add(orc_stats *dst, const orc_stats *src) {
  dst->paths += src->paths; dst->segments += src->segments; dst->tests += src->tests;
  dst->rng_draws += src->rng_draws; dst->absorbed += src->absorbed; dst->capped += src->capped;
}

/* ------------------------------------------------------------------ */
/* sequential mode: c/main.c:446-483 per band, c/main.c:504-545 partition */

typedef struct {
  const orc_camera *cam; const orc_sphere *s; size_t n;
  size_t nx, ny_band, nsamples;
  float ymin, ymax;
  uint8_t *out;
  orc_stats st;
} band_job;

static void *band_run(void *arg) {
  band_job *j = (band_job *)arg;
  rng_t g; memset(&g, 0, sizeof g);
  g.state = 123456789u;                                          /* rand.c:9 */
  float yheight = j->ymax - j->ymin;
  size_t bi = 0;
  memset(&j->st, 0, sizeof j->st);
  for (size_t row = j->ny_band; row > 0; row--) {
    for (size_t col = 0; col < j->nx; col++) {
      f3 sum = f3_make(0.0f, 0.0f, 0.0f);
      for (size_t sidx = 0; sidx < j->nsamples; sidx++) {
        float x = (float)((double)col + rng_next(&g)) / (float)j->nx;              /* main.c:465 */
        float y = j->ymin + yheight * (float)((double)(row - 1) + rng_next(&g)) / (float)j->ny_band;
        ray_t r = camera_ray(j->cam, x, y, &g);
        int fh;
        sum = f3_add(sum, path_radiance(&r, j->s, j->n, 50, &g, &j->st, &fh));
      }
      finish_pixel(sum, j->nsamples, j->out + bi, NULL);
      bi += 3;
    }
  }
  j->st.rng_draws = g.draws;
  return NULL;
}

int orc_render_sequential(const orc_camera *cam, const orc_sphere *s, size_t n,
                          size_t nx, size_t ny, size_t nsamples, size_t threads,
                          uint8_t *rgb8, orc_stats *stats) {
  if (!cam || !s || !rgb8 || threads == 0 || nsamples == 0) return 1;
  band_job *jobs = (band_job *)calloc(threads, sizeof *jobs);
  pthread_t *tid = (pthread_t *)calloc(threads, sizeof *tid);
  if (!jobs || !tid) { free(jobs); free(tid); return 2; }
  size_t done_rows = 0, off = 0;
  for (size_t i = 0; i < threads; i++) {                         /* main.c:510-529 */
    size_t left = ny - done_rows;
    size_t mine = left / (threads - i);
    jobs[i].cam = cam; jobs[i].s = s; jobs[i].n = n;
    jobs[i].nx = nx; jobs[i].ny_band = mine; jobs[i].nsamples = nsamples;
    jobs[i].ymax = (float)left / (float)ny;
    jobs[i].ymin = (float)(left - mine) / (float)ny;
    jobs[i].out = rgb8 + off;
    done_rows += mine;
    off += 3 * nx * mine;
  }
  for (size_t i = 0; i < threads; i++) pthread_create(&tid[i], NULL, band_run, &jobs[i]);
  if (stats) memset(stats, 0, sizeof *stats);
  for (size_t i = 0; i < threads; i++) {
    pthread_join(tid[i], NULL);
    if (stats) stats_add(stats, &jobs[i].st);
  }
  free(jobs); free(tid);
  return 0;
}

/* ------------------------------------------------------------------ */
/* deterministic mode */

typedef struct {
  const orc_camera *cam; const orc_sphere *s; size_t n;
  size_t nx, ny_total, row_begin, row_end, row_step, row0, nsamples, max_depth;
  uint32_t seed;
  uint8_t *rgb8; float *linear; int32_t *first_hit; uint32_t *hit_sum;
  orc_stats st;
} det_job;

static void *det_run(void *arg) {
  det_job *j = (det_job *)arg;
  memset(&j->st, 0, sizeof j->st);
  for (size_t row = j->row_begin; row < j->row_end; row += j->row_step) { /* 0 = top row */
    size_t up = j->ny_total - 1 - row;                          /* = (j-1) of main.c:466 */
    for (size_t col = 0; col < j->nx; col++) {
      size_t local = (row - j->row0) * j->nx + col;
      uint32_t pixel = (uint32_t)(row * j->nx + col);
      f3 sum = f3_make(0.0f, 0.0f, 0.0f);
      uint32_t hsum = 0; int fh0 = -1;
      for (size_t sidx = 0; sidx < j->nsamples; sidx++) {
        rng_t g; memset(&g, 0, sizeof g);
        g.counter_mode = 1;
        g.key = orc_path_key(j->seed, pixel, (uint32_t)sidx);
        float x = (float)((double)col + rng_next(&g)) / (float)j->nx;
        float y = (float)((double)up + rng_next(&g)) / (float)j->ny_total;
        ray_t r = camera_ray(j->cam, x, y, &g);
        int fh;
        sum = f3_add(sum, path_radiance(&r, j->s, j->n, j->max_depth, &g, &j->st, &fh));
        j->st.rng_draws += g.draws;
        hsum += (uint32_t)(fh + 1);
        if (sidx == 0) fh0 = fh;
      }
      finish_pixel(sum, j->nsamples, j->rgb8 ? j->rgb8 + 3 * local : NULL,
                   j->linear ? j->linear + 3 * local : NULL);
      if (j->first_hit) j->first_hit[local] = fh0;
      if (j->hit_sum) j->hit_sum[local] = hsum;
    }
  }
  return NULL;
}

int orc_render_det(const orc_camera *cam, const orc_sphere *s, size_t n,
                   size_t nx, size_t ny_total, size_t row0, size_t rows,
                   size_t nsamples, size_t max_depth, uint32_t seed,
                   uint8_t *rgb8, float *linear, int32_t *first_hit,
                   uint32_t *hit_sum, orc_stats *stats, size_t host_threads) {
  if (!cam || !s || nsamples == 0 || row0 + rows > ny_total) return 1;
  if (host_threads == 0) host_threads = 1;
  if (host_threads > rows && rows > 0) host_threads = rows;
  if (stats) memset(stats, 0, sizeof *stats);
  if (rows == 0) return 0;
  det_job *jobs = (det_job *)calloc(host_threads, sizeof *jobs);
  pthread_t *tid = (pthread_t *)calloc(host_threads, sizeof *tid);
  if (!jobs || !tid) { free(jobs); free(tid); return 2; }
  /* rows are dealt round-robin so cheap sky rows spread over the threads */
  for (size_t i = 0; i < host_threads; i++) {
    det_job *j = &jobs[i];
    j->cam = cam; j->s = s; j->n = n; j->nx = nx; j->ny_total = ny_total;
    j->row0 = row0; j->nsamples = nsamples; j->max_depth = max_depth; j->seed = seed;
    j->rgb8 = rgb8; j->linear = linear; j->first_hit = first_hit; j->hit_sum = hit_sum;
    j->row_begin = row0 + i;
    j->row_end = row0 + rows;
    j->row_step = host_threads;
  }
  for (size_t i = 0; i < host_threads; i++) pthread_create(&tid[i], NULL, det_run, &jobs[i]);
  for (size_t i = 0; i < host_threads; i++) {
    pthread_join(tid[i], NULL);
    if (stats) stats_add(stats, &jobs[i].st);
  }
  free(jobs); free(tid);
  return 0;
}
