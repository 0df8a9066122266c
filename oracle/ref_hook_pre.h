/*
 * ref_hook_pre.h -- force-included in front of the reference translation unit
 * when oracle/Makefile builds oracle/_ref/libref_hooked.so.  TEST
 * INFRASTRUCTURE ONLY.
 *
 * The reference's generator (c/rand.c) is compiled from where it lies, under
 * the name ref_seq_rand; my_rand() then becomes a switch between that
 * sequential chain and the build's counter stream, so the reference's own
 * camera_ray_at_xy / ray_color / scatter can be driven with per-(pixel,
 * sample) seeds.  The Makefile drops the TU's own '#include "rand.c"' line from
 * the stream it compiles; no reference source is copied or kept.
 */
#include <stdint.h>

#define my_rand ref_seq_rand
#include "rand.c" /* resolved through -I to /root/reference/c/rand.c */
#undef my_rand

static __thread int hook_counter_mode;
static __thread uint32_t hook_key, hook_k;
static __thread uint64_t hook_draws;

static uint32_t hook_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

static uint32_t hook_draw_hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 16; x *= 0x846ca68bu;
  return x;
}

static double my_rand(void) {
  hook_draws++;
  if (hook_counter_mode) {
    /* the build's counter draw: 23 random bits as a 31-bit integer, then the reference's own conversion (c/rand.c:15) */
    uint32_t bits = (hook_draw_hash(hook_key + (hook_k++) * 0x9e3779b9u) >> 9) << 8;
    return (double)bits / (double)0x7fffffff;
  }
  return ref_seq_rand();
}
