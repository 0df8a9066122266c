/*
 * ref_hook_post.c -- appended behind the reference translation unit when
 * oracle/Makefile builds oracle/_ref/libref_hooked.so; it can therefore see
 * the reference's static functions and types (v3, ray, sphere, scene, camera,
 * hit_record of /root/reference/c) and exports thin callers for them.
 * TEST INFRASTRUCTURE ONLY.  Nothing here is an algorithm of its own except
 * the per-pixel sample loop of ref_det_render, which has to be re-stated
 * because the reference's render() cannot reseed per (pixel, sample).
 */

#include <stddef.h>

typedef struct {
  float center[3];
  float radius;
  int32_t kind;
  float albedo[3];
  float param;
} flat_sphere;

static void to_flat(const sphere *s, flat_sphere *f) {
  f->center[0] = s->center.x; f->center[1] = s->center.y; f->center[2] = s->center.z;
  f->radius = s->radius;
  f->kind = (int32_t)s->mat.type;
  f->albedo[0] = f->albedo[1] = f->albedo[2] = 0; f->param = 0;
  if (s->mat.type == MATTE) {
    f->albedo[0] = s->mat.matte.albedo.x; f->albedo[1] = s->mat.matte.albedo.y; f->albedo[2] = s->mat.matte.albedo.z;
  } else if (s->mat.type == METAL) {
    f->albedo[0] = s->mat.metal.albedo.x; f->albedo[1] = s->mat.metal.albedo.y; f->albedo[2] = s->mat.metal.albedo.z;
    f->param = s->mat.metal.fuzz;
  } else {
    f->param = s->mat.dielectric.ref_idx;
  }
}

static void from_flat(const flat_sphere *f, sphere *s) {
  memset(s, 0, sizeof *s);
  s->center = (v3){f->center[0], f->center[1], f->center[2]};
  s->radius = f->radius;
  s->mat.type = (material_t)f->kind;
  if (f->kind == MATTE) {
    s->mat.matte.albedo = (v3){f->albedo[0], f->albedo[1], f->albedo[2]};
  } else if (f->kind == METAL) {
    s->mat.metal.albedo = (v3){f->albedo[0], f->albedo[1], f->albedo[2]};
    s->mat.metal.fuzz = f->param;
  } else {
    s->mat.dielectric.ref_idx = f->param;
  }
}

/* ---- generator ---- */

uint32_t ref_f_randi(uint32_t x) { return f_randi(x); }

typedef struct { size_t n; double *out; } seq_job;
static void *seq_run(void *p) {
  seq_job *j = (seq_job *)p;
  hook_counter_mode = 0;
  for (size_t i = 0; i < j->n; i++) j->out[i] = my_rand();
  return NULL;
}
/* first n values of the sequential stream of a FRESH thread (seed 123456789) */
void ref_seq_stream(size_t n, double *out) {
  seq_job j = {n, out};
  pthread_t t;
  pthread_create(&t, NULL, seq_run, &j);
  pthread_join(t, NULL);
}

/* ---- camera ---- */

void ref_camera_new(const float from[3], const float at[3], const float up[3],
                    float vfov, float aspect, float aperture, float focus, float out[22]) {
  camera c = camera_new((v3){from[0], from[1], from[2]}, (v3){at[0], at[1], at[2]},
                        (v3){up[0], up[1], up[2]}, vfov, aspect, aperture, focus);
  memcpy(out, &c, sizeof c);
}

size_t ref_sizeof_camera(void) { return sizeof(camera); }
size_t ref_sizeof_sphere(void) { return sizeof(sphere); }

/* ---- scenes (each built on a fresh thread so the stream starts at the seed) ---- */

typedef struct { int which; flat_sphere *out; size_t cap; size_t n; } scene_job;
static void *scene_run(void *p) {
  scene_job *j = (scene_job *)p;
  hook_counter_mode = 0;
  if (j->which == 0) {
    size_t n = small_scene(NULL);
    sphere *tmp = calloc(n, sizeof *tmp);
    scene sc = {.nspheres = n, .spheres = tmp};
    n = small_scene(&sc);
    for (size_t i = 0; i < n && i < j->cap; i++) to_flat(&tmp[i], &j->out[i]);
    j->n = n;
    free(tmp);
  } else {
    scene sc = random_scene();
    for (size_t i = 0; i < sc.nspheres && i < j->cap; i++) to_flat(&sc.spheres[i], &j->out[i]);
    j->n = sc.nspheres;
    free(sc.spheres);
  }
  return NULL;
}
size_t ref_scene(int which, flat_sphere *out, size_t cap) {
  scene_job j = {which, out, cap, 0};
  pthread_t t;
  pthread_create(&t, NULL, scene_run, &j);
  pthread_join(t, NULL);
  return j.n;
}

/* ---- per-function callers ---- */

int ref_hit_sphere(const flat_sphere *f, const float A[3], const float B[3],
                   float tmin, float tmax, float *t, float p[3], float nrm[3]) {
  sphere s; from_flat(f, &s);
  ray r = {.A = {A[0], A[1], A[2]}, .B = {B[0], B[1], B[2]}};
  hit_record rec;
  if (!hit_sphere(&s, &r, tmin, tmax, &rec)) return 0;
  *t = rec.t;
  p[0] = rec.p.x; p[1] = rec.p.y; p[2] = rec.p.z;
  nrm[0] = rec.normal.x; nrm[1] = rec.normal.y; nrm[2] = rec.normal.z;
  return 1;
}

float ref_schlick(float cosine, float ref_idx) { return schlick(cosine, ref_idx); }

int ref_refract(const float u[3], const float n[3], float ni_over_nt, float out[3]) {
  v3 o;
  if (!refract((v3){u[0], u[1], u[2]}, (v3){n[0], n[1], n[2]}, ni_over_nt, &o)) return 0;
  out[0] = o.x; out[1] = o.y; out[2] = o.z;
  return 1;
}

/* scatter() on sphere f hit at (p, nrm) by ray (A,B), drawing from counter
 * stream key.  Returns the reference's bool; fills attenuation and the new ray. */
int ref_scatter(const flat_sphere *f, const float A[3], const float B[3],
                const float p[3], const float nrm[3], uint32_t key,
                float atten[3], float oA[3], float oB[3], uint32_t *draws_used) {
  sphere s; from_flat(f, &s);
  ray rin = {.A = {A[0], A[1], A[2]}, .B = {B[0], B[1], B[2]}};
  hit_record rec = {.t = 0, .p = {p[0], p[1], p[2]}, .normal = {nrm[0], nrm[1], nrm[2]}, .mat = &s.mat};
  v3 att = {0, 0, 0};
  ray out = rin;
  hook_counter_mode = 1; hook_key = key; hook_k = 0;
  int ok = scatter(&rin, &rec, &att, &out);
  hook_counter_mode = 0;
  *draws_used = hook_k;
  atten[0] = att.x; atten[1] = att.y; atten[2] = att.z;
  oA[0] = out.A.x; oA[1] = out.A.y; oA[2] = out.A.z;
  oB[0] = out.B.x; oB[1] = out.B.y; oB[2] = out.B.z;
  return ok;
}

/* ---- deterministic whole render through the reference's own
 *      camera_ray_at_xy / ray_color / hit_scene (max depth 50 is built in) ---- */

static uint32_t hook_path_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
  uint32_t a = hook_mix32(pixel + 0x9e3779b9u * seed);
  return hook_mix32(a ^ (sample * 0x85ebca6bu + 0x165667b1u));
}

int ref_det_render(const float cam22[22], const flat_sphere *fs, size_t n,
                   size_t nx, size_t ny_total, size_t row0, size_t rows,
                   size_t nsamples, uint32_t seed, uint8_t *rgb8, float *linear,
                   int32_t *first_hit, uint32_t *hit_sum) {
  camera cam; memcpy(&cam, cam22, sizeof cam);
  sphere *sp = calloc(n, sizeof *sp);
  if (!sp) return 2;
  for (size_t i = 0; i < n; i++) from_flat(&fs[i], &sp[i]);
  scene sc = {.nspheres = n, .spheres = sp};
  for (size_t row = row0; row < row0 + rows; row++) {
    size_t up = ny_total - 1 - row;
    for (size_t col = 0; col < nx; col++) {
      size_t local = (row - row0) * nx + col;
      v3 color = v3_zero;
      uint32_t hs = 0; int32_t fh0 = -1;
      for (size_t s = 0; s < nsamples; s++) {
        hook_counter_mode = 1;
        hook_key = hook_path_key(seed, (uint32_t)(row * nx + col), (uint32_t)s);
        hook_k = 0;
        /* the two lines of c/main.c:465-466 with ymin 0, ymax 1 */
        float x = (float)(col + 0 + my_rand()) / (float)nx;
        float y = 0.0f + 1.0f * (float)(up + my_rand()) / (float)ny_total;
        ray r = camera_ray_at_xy(&cam, x, y);
        hit_record rec;
        int32_t fh = -1;
        if (hit_scene(&sc, &r, 0.001, FLT_MAX, &rec))
          fh = (int32_t)((sphere *)((char *)rec.mat - offsetof(sphere, mat)) - sp);
        color = v3_add(color, ray_color(&r, &sc));
        hook_counter_mode = 0;
        hs += (uint32_t)(fh + 1);
        if (s == 0) fh0 = fh;
      }
      color = v3_kdiv(color, (float)nsamples);
      if (linear) { linear[3 * local] = color.x; linear[3 * local + 1] = color.y; linear[3 * local + 2] = color.z; }
      color = v3_sqrt(color);
      if (isnanf(color.x)) color.x = 0;
      if (isnanf(color.y)) color.y = 0;
      if (isnanf(color.z)) color.z = 0;
      if (rgb8) {
        rgb8[3 * local + 0] = (255.0f * color.x);
        rgb8[3 * local + 1] = (255.0f * color.y);
        rgb8[3 * local + 2] = (255.0f * color.z);
      }
      if (first_hit) first_hit[local] = fh0;
      if (hit_sum) hit_sum[local] = hs;
    }
  }
  free(sp);
  return 0;
}

/* whole image through the reference's render() itself, one band, sequential
 * stream of a fresh thread: the "-t 1" picture for any scene and size */
typedef struct { render_arg a; } band1_job;
int ref_render_one_band(const float cam22[22], const flat_sphere *fs, size_t n,
                        size_t nx, size_t ny, size_t nsamples, uint8_t *rgb8) {
  camera cam; memcpy(&cam, cam22, sizeof cam);
  sphere *sp = calloc(n, sizeof *sp);
  if (!sp) return 2;
  for (size_t i = 0; i < n; i++) from_flat(&fs[i], &sp[i]);
  scene sc = {.nspheres = n, .spheres = sp};
  render_arg a = {.buf = rgb8, .len = 3 * nx * ny, .cam = &cam, .sc = &sc,
                  .nsamples = nsamples, .nx = nx, .ny = ny, .ymin = 0.0f, .ymax = 1.0f};
  pthread_t t;
  pthread_create(&t, NULL, &render, &a);
  pthread_join(t, NULL);
  free(sp);
  return 0;
}
