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

static const char kSceneMagic[8] = {'R', 'T', 'S', 'C', 'E', 'N', 'E', '1'};

int rth_scene_save(const char *path, const rt_sphere *spheres, size_t n) {
  if (!path || !spheres || n == 0 || n > 0xffffffffu) { rth_set_error("null or empty argument"); return RT_ERR_ARG; }
  FILE *fp = fopen(path, "wb");
  if (!fp) { rth_set_error("could not open the scene file for writing"); return RT_ERR_ARG; }
  const uint32_t head[2] = {(uint32_t)sizeof(rt_sphere), (uint32_t)n};
  int ok = fwrite(kSceneMagic, 1, sizeof kSceneMagic, fp) == sizeof kSceneMagic && fwrite(head, sizeof head[0], 2, fp) == 2 &&
           fwrite(spheres, sizeof *spheres, n, fp) == n;
  if (fclose(fp) != 0) ok = 0;
  if (!ok) { rth_set_error("could not write the scene file"); return RT_ERR_ARG; }
  return RT_OK;
}

int rth_scene_load(const char *path, rt_sphere **out, size_t *n) {
  if (!path || !out || !n) { rth_set_error("null argument"); return RT_ERR_ARG; }
  *out = NULL;
  *n = 0;
  FILE *fp = fopen(path, "rb");
  if (!fp) { rth_set_error("could not open the scene file"); return RT_ERR_ARG; }
  char magic[8];
  uint32_t head[2];
  int rc = RT_OK;
  rt_sphere *s = NULL;
  if (fread(magic, 1, sizeof magic, fp) != sizeof magic || memcmp(magic, kSceneMagic, sizeof magic) != 0 ||
      fread(head, sizeof head[0], 2, fp) != 2 || head[0] != sizeof(rt_sphere) || head[1] == 0) {
    rth_set_error("not a scene file of this format");
    rc = RT_ERR_ARG;
  }
  if (rc == RT_OK) {
    s = (rt_sphere *)malloc((size_t)head[1] * sizeof *s);
    if (!s) { rth_set_error("out of host memory"); rc = RT_ERR_NOMEM; }
  }
  if (rc == RT_OK && (fread(s, sizeof *s, head[1], fp) != head[1] || fgetc(fp) != EOF)) {
    rth_set_error("the scene file's length does not match its count");
    rc = RT_ERR_ARG;
  }
  for (uint32_t i = 0; rc == RT_OK && i < head[1]; ++i)
    if (s[i].type < 0 || s[i].type > 2) { rth_set_error("the scene file holds an unknown material type"); rc = RT_ERR_ARG; }
  fclose(fp);
  if (rc != RT_OK) { free(s); return rc; }
  *out = s;
  *n = head[1];
  return RT_OK;
}

int rth_write_ppm(FILE *fp, const uint8_t *rgb8, size_t w, size_t h) {
  if (!fp || !rgb8) return RT_ERR_ARG;
  if (fprintf(fp, "P6\n%zu %zu 255\n", w, h) < 0) return RT_ERR_ARG;          /* c/main.c:412 */
  fflush(fp);
  if (w != 0 && h != 0 && fwrite(rgb8, 3 * w * h, 1, fp) != 1) return RT_ERR_ARG;
  return RT_OK;
}

/* ---- PNG (c/main.c:427-433 writes one through a third-party encoder; this is a writer of its own:
 * 8-bit RGB, filter 0 on every row, the pixel stream in stored deflate blocks -- valid, uncompressed) ---- */

static uint32_t crc_table[256];
static int crc_ready = 0;

static uint32_t crc32_update(uint32_t c, const uint8_t *p, size_t n) {
  if (!crc_ready) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t r = i;
      for (int k = 0; k < 8; k++) r = (r & 1u) ? 0xedb88320u ^ (r >> 1) : r >> 1;
      crc_table[i] = r;
    }
    crc_ready = 1;
  }
  for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xffu] ^ (c >> 8);
  return c;
}

static void be32(uint8_t *o, uint32_t v) { o[0] = (uint8_t)(v >> 24); o[1] = (uint8_t)(v >> 16); o[2] = (uint8_t)(v >> 8); o[3] = (uint8_t)v; }

static int png_chunk(FILE *fp, const char type[4], const uint8_t *data, size_t n) {
  uint8_t head[8], tail[4];
  be32(head, (uint32_t)n);
  memcpy(head + 4, type, 4);
  uint32_t c = crc32_update(0xffffffffu, head + 4, 4);
  c = crc32_update(c, data, n) ^ 0xffffffffu;
  be32(tail, c);
  return fwrite(head, 8, 1, fp) == 1 && (n == 0 || fwrite(data, n, 1, fp) == 1) && fwrite(tail, 4, 1, fp) == 1;
}

int rth_write_png(FILE *fp, const uint8_t *rgb8, size_t w, size_t h) {
  if (!fp || !rgb8 || w == 0 || h == 0 || w > 0x7fffffffu || h > 0x7fffffffu) return RT_ERR_ARG;
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  uint8_t ihdr[13];
  be32(ihdr, (uint32_t)w);
  be32(ihdr + 4, (uint32_t)h);
  ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;   /* 8 bits, RGB, deflate, no filter method, no interlace */
  const size_t row = 3 * w + 1, raw = row * h;
  const size_t blocks = (raw + 65534) / 65535;
  const size_t zlen = 2 + raw + 5 * blocks + 4;
  uint8_t *z = (uint8_t *)malloc(zlen);
  if (!z) return RT_ERR_NOMEM;
  size_t o = 0;
  z[o++] = 0x78; z[o++] = 0x01;                       /* zlib header: deflate, 32 KiB window, no preset dictionary */
  uint32_t a = 1, b = 0;                              /* Adler-32 of the raw stream */
  size_t in_block = 0, left = raw;
  for (size_t y = 0; y < h; y++) {
    for (size_t k = 0; k < row; k++) {
      if (in_block == 0) {                            /* a stored block: final flag, LEN, ~LEN */
        size_t len = left < 65535 ? left : 65535;
        z[o++] = (uint8_t)(left <= 65535);
        z[o++] = (uint8_t)len; z[o++] = (uint8_t)(len >> 8);
        z[o++] = (uint8_t)~len; z[o++] = (uint8_t)(~len >> 8);
        in_block = len;
      }
      uint8_t v = k == 0 ? 0 : rgb8[y * 3 * w + (k - 1)];   /* first byte of a row: filter type 0 */
      z[o++] = v;
      a = (a + v) % 65521u;
      b = (b + a) % 65521u;
      in_block--;
      left--;
    }
  }
  be32(z + o, (b << 16) | a);
  o += 4;
  int ok = fwrite(sig, 8, 1, fp) == 1 && png_chunk(fp, "IHDR", ihdr, 13) && png_chunk(fp, "IDAT", z, o) && png_chunk(fp, "IEND", NULL, 0);
  free(z);
  return ok ? RT_OK : RT_ERR_ARG;
}
