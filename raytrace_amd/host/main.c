/*
 * main.c -- command-line renderer: the reference's CLI (c/main.c:32-138) in
 * front of the GPU path.  Same options with the same meaning and messages
 * where they still apply:
 *     -o <file>.<ppm|png|jpg> | -   output file; '-' writes the PPM to stdout
 *     -n <samples>          samples per pixel (default 10, c/main.c:61)
 *     -t <devices>          number of GPUs to shard over (the reference's
 *                           thread count, c/main.c:62; default 1)
 * and, because the reference hard-codes them (c/main.c:486, 499):
 *     -x <width> -y <height>   image size (default 600x300)
 *     -s small|final|dense|glass|trivial   scene (default small, as the reference renders)
 *     -d <max depth>  -S <seed>
 *     -i <scene file>       render a scene dump (rth_scene_load) instead of a built-in scene
 *     -D <scene file>       also write the scene that is rendered as a scene dump
 * PPM, PNG and JPG (quality 90, c/main.c:434) are written by the host library's
 * own encoders: conforming files of the same picture, not the bytes of the
 * reference's vendored third-party encoders.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>

#include "rt_host.h"

static const char *argv0;

static void usage(void) {
  fprintf(stderr,
          "Usage: %s [-n <number of samples>] [-t <number of GPUs>] [-x <width>] [-y <height>]\n"
          "          [-s small|final|dense|glass|trivial | -i <scene file>] [-D <scene file>] [-d <max depth>] [-S <seed>]\n"
          "          -o <output file>.<ppm/png/jpg>\n"
          "\n"
          "If <output file> is '-' then a ppm image is written on stdout.\n"
          "\n"
          "Examples:\n"
          "\n",
          argv0);
  fprintf(stderr, "	%s -n 100 -t 8 -s final -x 1920 -y 1080 -o out.ppm\n", argv0);
  fprintf(stderr, "	%s -n 100 -t 4 -o out.jpg\n", argv0);
  fprintf(stderr, "	%s -o out.ppm\n", argv0);
  fprintf(stderr, "	%s -o - > out.ppm\n", argv0);
  exit(1);
}

static long positive(const char *what, const char *arg) {
  char *end = NULL;
  long n = strtol(arg, &end, 10);
  if (n <= 0 || strcmp("", end) != 0) {
    fprintf(stderr, "invalid %s (%s), must be a positive integer\n", what, arg);   /* c/main.c:100-105 */
    usage();
  }
  return n;
}

int main(int argc, char **argv) {
  argv0 = argv[0];
  FILE *fp = NULL;
  int write_png = 0, write_jpg = 0;
  rth_render_opts o;
  memset(&o, 0, sizeof o);
  o.nx = 600; o.ny = 300;          /* c/main.c:486 */
  o.nsamples = 10;                 /* c/main.c:61  */
  o.n_devices = 1;
  o.max_depth = 50;                /* c/main.c:262 */
  o.tile_rows = 8;
  int which = RTH_SCENE_SMALL;     /* what c/main.c:499 renders */
  const char *scene_in = NULL, *scene_out = NULL;
  int opt;
  while ((opt = getopt(argc, argv, "o:n:t:x:y:s:d:S:i:D:")) != -1) {
    switch (opt) {
    case 'o': {
      if (strcmp(optarg, "-") == 0) { fp = stdout; break; }
      const char *ext = strrchr(optarg, '.');
      if (ext == NULL) {
        fprintf(stderr, "missing file name extension, support ppm/png/jpg\n");   /* c/main.c:73-77 */
        usage();
      } else if (strcasecmp(ext, ".png") == 0) {
        write_png = 1;
      } else if (strcasecmp(ext, ".jpg") == 0 || strcasecmp(ext, ".jpeg") == 0) {   /* c/main.c:80-81 */
        write_jpg = 1;
      } else if (strcasecmp(ext, ".ppm") != 0) {
        fprintf(stderr, "unsupported file extension: %s (only support ppm/png/jpg)\n", ext);
        usage();
      }
      fp = fopen(optarg, "wb");
      if (fp == NULL) { perror("could not open output file"); exit(1); }
    } break;
    case 'n': o.nsamples = (uint32_t)positive("number of samples", optarg); break;
    case 't': o.n_devices = (int)positive("number of GPUs", optarg); break;
    case 'x': o.nx = (uint32_t)positive("width", optarg); break;
    case 'y': o.ny = (uint32_t)positive("height", optarg); break;
    case 'd': o.max_depth = (uint32_t)positive("max depth", optarg); break;
    case 'i': scene_in = optarg; break;
    case 'D': scene_out = optarg; break;
    case 'S': o.seed = (uint32_t)strtoul(optarg, NULL, 10); break;
    case 's':
      if (!strcmp(optarg, "small")) which = RTH_SCENE_SMALL;
      else if (!strcmp(optarg, "final")) which = RTH_SCENE_FINAL;
      else if (!strcmp(optarg, "dense")) which = RTH_SCENE_DENSE;
      else if (!strcmp(optarg, "glass")) which = RTH_SCENE_GLASS;
      else if (!strcmp(optarg, "trivial")) which = RTH_SCENE_TRIVIAL;
      else { fprintf(stderr, "unknown scene: %s\n", optarg); usage(); }
      break;
    default: usage();
    }
  }
  if (fp == NULL) usage();                                     /* c/main.c:126-128 */

  rt_sphere *spheres = NULL;
  size_t n = 0;
  if (scene_in) {
    if (rth_scene_load(scene_in, &spheres, &n) != RT_OK) { fprintf(stderr, "could not load %s: %s\n", scene_in, rth_last_error()); return 1; }
  } else if (rth_scene_build(which, &spheres, &n) != RT_OK) {
    fprintf(stderr, "could not build the scene\n");
    return 1;
  }
  if (scene_out && rth_scene_save(scene_out, spheres, n) != RT_OK) {
    fprintf(stderr, "could not write %s: %s\n", scene_out, rth_last_error());
    return 1;
  }
  rt_camera cam;
  rth_camera_default(o.nx, o.ny, &cam);
  uint8_t *img = (uint8_t *)malloc((size_t)3 * o.nx * o.ny);  /* heap, not the reference's stack VLA (c/main.c:504-505) */
  if (!img) { fprintf(stderr, "out of memory\n"); return 1; }
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  float kernel_ms = 0;
  int devices_used = 0;
  int rc = rth_raytrace_ex(&cam, spheres, n, &o, img, &kernel_ms, &devices_used);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (rc != RT_OK) { fprintf(stderr, "render failed (%d): %s\n", rc, rth_last_error()); return 1; }
  double wall = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  double paths = (double)o.nx * o.ny * o.nsamples;
  fprintf(stderr, "raytracing took %.3f seconds (%zu spheres, %ux%u, %u spp, %d rank(s) on %d GPU(s); kernels %.3f ms, %.3g paths/s)\n",
          wall, n, o.nx, o.ny, o.nsamples, o.n_devices, devices_used, kernel_ms, kernel_ms > 0 ? paths / (kernel_ms * 1e-3) : 0.0);
  if ((write_png ? rth_write_png(fp, img, o.nx, o.ny) : write_jpg ? rth_write_jpg(fp, img, o.nx, o.ny, 90) : rth_write_ppm(fp, img, o.nx, o.ny)) != RT_OK) {
    perror("could not write image");
    return 1;
  }
  if (fp != stdout && 0 != fclose(fp)) { perror("could not close output file"); exit(1); }   /* c/main.c:132-135 */
  free(img);
  free(spheres);
  return 0;
}
