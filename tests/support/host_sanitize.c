/* Exercises the host library's CPU-only code (scene builders, generator, camera, image writers, scene dumps, argument
 * errors) in one program that the test suite builds with -fsanitize=address,undefined: any out-of-bounds access,
 * leak or undefined operation ends it with a non-zero status.  TEST HARNESS ONLY.  argv[1] = a scratch directory. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_host.h"

#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) { fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

int main(int argc, char **argv) {
  CHECK(argc == 2);
  char path[1024];
  /* generator and camera */
  rth_rng g;
  rth_rng_init(&g);
  double sum = 0;
  for (int i = 0; i < 1000; ++i) sum += rth_rng_next(&g);
  CHECK(sum > 400 && sum < 600);
  rt_camera cam;
  rth_camera_default(1920, 1080, &cam);
  /* every scene builder, sized by its own first pass, then once more into a buffer that is too small */
  for (int which = RTH_SCENE_SMALL; which <= RTH_SCENE_TRIVIAL; ++which) {
    rt_sphere *s = NULL;
    size_t n = 0;
    CHECK(rth_scene_build(which, &s, &n) == RT_OK && n > 0);
    for (size_t i = 0; i < n; ++i) CHECK(s[i].type >= 0 && s[i].type <= 2);
    snprintf(path, sizeof path, "%s/scene_%d.bin", argv[1], which);
    CHECK(rth_scene_save(path, s, n) == RT_OK);
    rt_sphere *back = NULL;
    size_t nb = 0;
    CHECK(rth_scene_load(path, &back, &nb) == RT_OK && nb == n && memcmp(back, s, n * sizeof *s) == 0);
    free(back);
    free(s);
  }
  rt_sphere few[4];
  rth_rng_init(&g);
  CHECK(rth_scene_small(few, 4) == 0 || rth_scene_small(few, 4) <= 4);   /* never writes past cap */
  CHECK(rth_scene_random(few, 4, 11, &g) <= 4);
  /* bad scene files */
  snprintf(path, sizeof path, "%s/not_a_scene.bin", argv[1]);
  FILE *fp = fopen(path, "wb");
  CHECK(fp);
  fputs("RTSCENE1", fp);
  fclose(fp);
  rt_sphere *none = NULL;
  size_t nn = 0;
  CHECK(rth_scene_load(path, &none, &nn) != RT_OK && none == NULL);
  CHECK(rth_scene_load("/nonexistent/dir/x.bin", &none, &nn) != RT_OK);
  CHECK(rth_scene_save(NULL, NULL, 0) != RT_OK);
  /* image writers on sizes around the block and the stored-deflate limits */
  const size_t dims[][2] = {{1, 1}, {7, 9}, {8, 8}, {17, 3}, {300, 75}, {1, 300}};
  for (size_t d = 0; d < sizeof dims / sizeof dims[0]; ++d) {
    const size_t w = dims[d][0], h = dims[d][1];
    unsigned char *img = (unsigned char *)malloc(3 * w * h);
    CHECK(img);
    for (size_t i = 0; i < 3 * w * h; ++i) img[i] = (unsigned char)((i * 37u + d * 11u) ^ (i >> 3));
    const char *ext[3] = {"ppm", "png", "jpg"};
    for (int e = 0; e < 3; ++e) {
      snprintf(path, sizeof path, "%s/img_%zu.%s", argv[1], d, ext[e]);
      fp = fopen(path, "wb");
      CHECK(fp);
      int rc = e == 0 ? rth_write_ppm(fp, img, w, h) : e == 1 ? rth_write_png(fp, img, w, h) : rth_write_jpg(fp, img, w, h, d % 2 ? 90 : 100);
      CHECK(rc == RT_OK);
      CHECK(fclose(fp) == 0);
    }
    free(img);
  }
  CHECK(rth_write_jpg(NULL, NULL, 0, 0, 90) != RT_OK);
  CHECK(rth_write_png(NULL, NULL, 0, 0) != RT_OK);
  /* the service and the driver refuse null arguments without touching a GPU */
  CHECK(rth_service_start(0, NULL, 0, NULL) != RT_OK);
  CHECK(rth_service_submit(NULL, NULL, NULL) == 0);
  CHECK(rth_raytrace(NULL, NULL, 0, NULL, NULL, NULL) != RT_OK);
  CHECK(rth_last_error() != NULL);
  puts("host sanitize ok");
  return 0;
}
