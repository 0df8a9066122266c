// core_host.cpp -- TEST HARNESS ONLY.  Compiles raytrace_amd/csrc/rt_core.h
// (the per-lane arithmetic the HIP kernels are built from) for the host, so
// that CPU-only tests can check that source bit for bit against the oracle:
// the filter's no-false-negative property, the exact hit test, the scatter
// functions, the generator and a per-pixel walk with the kernel's control
// flow.  The product library never runs this; there is no CPU render path.
#include <cstddef>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/rt_amd.h"
#include "../../raytrace_amd/csrc/rt_core.h"
#include "../../raytrace_amd/csrc/rt_cluster.h"

using namespace rtcore;

extern "C" {

void core_uniforms(uint32_t seed, uint32_t pixel, uint32_t sample, size_t n, double *out) {
  Rng g = rng_at(path_key(seed, pixel, sample));
  for (size_t i = 0; i < n; i++) out[i] = next_uniform(g);
}

// For every (ray, sphere) pair: reference discriminant sign vs filter sign.
// Returns the number of pairs with D_ref > 0 but filter <= 0 (must be 0);
// *passed = pairs the filter lets through, *ref_pos = pairs with D_ref > 0.
uint64_t core_filter_check(const rt_sphere *s, size_t n, const float *rays, size_t n_rays, uint64_t *passed,
                           uint64_t *ref_pos) {
  std::vector<FilterSphere> fs(n);
  for (size_t i = 0; i < n; i++) fs[i] = make_filter_sphere(s[i].center[0], s[i].center[1], s[i].center[2], s[i].radius);
  uint64_t bad = 0, pass = 0, pos = 0;
  for (size_t r = 0; r < n_rays; r++) {
    V3 A = v3(rays[6 * r], rays[6 * r + 1], rays[6 * r + 2]);
    V3 B = v3(rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]);
    FilterRay fr = make_filter_ray(A, B);
    float a = dot(B, B);
    for (size_t i = 0; i < n; i++) {
      V3 oc = A - v3(s[i].center[0], s[i].center[1], s[i].center[2]);
      float b = dot(oc, B);
      float c = dot(oc, oc) - s[i].radius * s[i].radius;
      float D = b * b - a * c;
      float Df = filter_discriminant(fr, fs[i].cx, fs[i].cy, fs[i].cz, fs[i].kprime);
      if (Df > 0.0f) pass++;
      if (D > 0.0f) {
        pos++;
        if (!(Df > 0.0f)) bad++;
      }
    }
  }
  if (passed) *passed = pass;
  if (ref_pos) *ref_pos = pos;
  return bad;
}

// packed two-sphere filter against the one-sphere form: number of halves whose bits differ (must be 0)
uint64_t core_filter_pair_mismatches(const rt_sphere *s, size_t n, const float *rays, size_t n_rays) {
  std::vector<FilterSphere> fs(n + 1);
  for (size_t i = 0; i < n; i++) fs[i] = make_filter_sphere(s[i].center[0], s[i].center[1], s[i].center[2], s[i].radius);
  fs[n] = FilterSphere{0.0f, 0.0f, 0.0f, 1.0e30f};
  uint64_t bad = 0;
  for (size_t r = 0; r < n_rays; r++) {
    V3 A = v3(rays[6 * r], rays[6 * r + 1], rays[6 * r + 2]);
    V3 B = v3(rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]);
    FilterRay fr = make_filter_ray(A, B);
    FilterRay2 fw = widen(fr);
    for (size_t i = 0; i + 1 <= n; i += 2) {
      f32x2 cx = {fs[i].cx, fs[i + 1].cx}, cy = {fs[i].cy, fs[i + 1].cy}, cz = {fs[i].cz, fs[i + 1].cz};
      f32x2 kp = {fs[i].kprime, fs[i + 1].kprime};
      f32x2 d = filter_discriminant2(fw, cx, cy, cz, kp);
      for (int h = 0; h < 2; h++) {
        float one = filter_discriminant(fr, fs[i + h].cx, fs[i + h].cy, fs[i + h].cz, fs[i + h].kprime);
        float two = d[h];
        if (std::memcmp(&one, &two, 4) != 0) bad++;
      }
    }
  }
  return bad;
}

// Group culling (rt_cluster.h): for every ray the kernel would cull with and
// every group whose box the ray's half line misses, the reference's hit test
// (c/main.c:214-236, any tmin >= 0) must report no hit for any member.
// Returns the number of violations (must be 0); *culled / *groups = how much
// a single ray skips.
uint64_t core_cull_check(const rt_sphere *s, size_t n, const float *rays, size_t n_rays, uint64_t *culled,
                         uint64_t *groups) {
  const size_t batch = 512;   // (the kernels' streaming batch)
  const rtcluster::Layout L = rtcluster::build_layout(s, n, batch);
  uint64_t bad = 0, cul = 0, tot = 0;
  for (size_t r = 0; r < n_rays; r++) {
    V3 A = v3(rays[6 * r], rays[6 * r + 1], rays[6 * r + 2]);
    V3 B = v3(rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]);
    if (!box_ray_in_range(A, B)) continue;   // the kernel scans every group for such rays
    const BoxRay br = make_box_ray(A, B, L.sigma0, L.sigma1);
    const float a = dot(B, B);
    const double inv_a = 1.0 / (double)a;
    for (size_t g = L.n_always_groups; g < L.boxes.size(); g++) {   // (the first groups are scanned for every ray)
      tot++;
      // culled when the ray misses the group's box, the box of its super-group or the box of its batch
      if (box_pass(br, L.boxes[g]) && box_pass(br, L.supers[g / rtcluster::kSuper]) &&
          box_pass(br, L.batches[g / (batch / rtcluster::kGroup)]))
        continue;
      cul++;
      for (int k = 0; k < rtcluster::kGroup; k++) {
        const uint32_t i = L.order[g * rtcluster::kGroup + k];
        if (i == 0xffffffffu) continue;
        float t;
        if (exact_hit(A, B, a, inv_a, s[i].center[0], s[i].center[1], s[i].center[2], s[i].radius * s[i].radius, 0.0f,
                      kFltMax, t))
          bad++;
      }
    }
  }
  if (culled) *culled = cul;
  if (groups) *groups = tot;
  return bad;
}

// the scan order is a permutation of the list plus pads; returns 0 if so
int core_layout_check(const rt_sphere *s, size_t n, uint32_t *n_scan, uint32_t *n_groups) {
  const rtcluster::Layout L = rtcluster::build_layout(s, n, 512);
  std::vector<int> seen(n, 0);
  for (uint32_t i : L.order)
    if (i != 0xffffffffu) {
      if (i >= n) return 1;
      seen[i]++;
    }
  for (size_t i = 0; i < n; i++)
    if (seen[i] != 1) return 2;
  if (L.n_scan % 64 || L.boxes.size() * 8 != L.n_scan || L.real_in_group.size() != L.boxes.size() ||
      L.supers.size() * rtcluster::kSuper != L.boxes.size() || L.batches.size() % 2 ||
      L.batches.size() * 512 < L.n_scan)
    return 3;
  *n_scan = L.n_scan;
  *n_groups = (uint32_t)L.boxes.size();
  return 0;
}

static int closest(const rt_sphere *s, const FilterSphere *fs, size_t n, V3 A, V3 B, float tmin, float tmax,
                   float &t_out, uint64_t *cand) {
  FilterRay fr = make_filter_ray(A, B);
  float a = dot(B, B);
  const double inv_a = 1.0 / (double)a;
  float best = tmax;
  int idx = -1;
  for (size_t i = 0; i < n; i++) {
    if (filter_discriminant(fr, fs[i].cx, fs[i].cy, fs[i].cz, fs[i].kprime) > 0.0f) {
      if (cand) (*cand)++;
      float t;
      if (exact_hit(A, B, a, inv_a, s[i].center[0], s[i].center[1], s[i].center[2], s[i].radius * s[i].radius, tmin, best, t)) {
        best = t;
        idx = (int)i;
      }
    }
  }
  t_out = best;
  return idx;
}

void core_closest_hit(const rt_sphere *s, size_t n, const float *rays, size_t n_rays, float tmin, float tmax,
                      int32_t *index, float *t) {
  std::vector<FilterSphere> fs(n);
  for (size_t i = 0; i < n; i++) fs[i] = make_filter_sphere(s[i].center[0], s[i].center[1], s[i].center[2], s[i].radius);
  for (size_t r = 0; r < n_rays; r++) {
    V3 A = v3(rays[6 * r], rays[6 * r + 1], rays[6 * r + 2]);
    V3 B = v3(rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]);
    float tt;
    index[r] = closest(s, fs.data(), n, A, B, tmin, tmax, tt, nullptr);
    t[r] = index[r] >= 0 ? tt : 0.0f;
  }
}

// The kernel's per-lane walk (mega_kernel of rt_device.hip) for the pixels of
// rows [row0, row0+rows): same functions, same order.
void core_render(const rt_camera *cam_in, const rt_sphere *s, size_t n, uint32_t nx, uint32_t ny, uint32_t row0,
                 uint32_t rows, uint32_t nsamples, uint32_t max_depth, float tmin, uint32_t seed, uint8_t *rgb8,
                 float *linear, int32_t *first_hit, uint32_t *hit_sum, uint64_t *stats /* paths, segments, cand */) {
  Camera cam;
  static_assert(sizeof(Camera) == sizeof(rt_camera), "layout");
  std::memcpy(&cam, cam_in, sizeof cam);
  std::vector<FilterSphere> fs(n);
  for (size_t i = 0; i < n; i++) fs[i] = make_filter_sphere(s[i].center[0], s[i].center[1], s[i].center[2], s[i].radius);
  uint64_t paths = 0, segs = 0, cands = 0;
  for (uint32_t row = row0; row < row0 + rows; row++) {
    for (uint32_t col = 0; col < nx; col++) {
      uint32_t pixel = row * nx + col, up = ny - 1 - row;
      V3 sum = v3(0, 0, 0);
      int32_t fh0 = -1;
      uint32_t hsum = 0;
      for (uint32_t sample = 0; sample < nsamples; sample++) {
        Rng g = rng_at(path_key(seed, pixel, sample));
        float x, y;
        pixel_xy(col, up, nx, ny, g, x, y);
        V3 A, B;
        camera_ray(cam, x, y, g, A, B);
        V3 color = v3(1, 1, 1);
        for (uint32_t depth = 0;;) {
          float t;
          segs++;
          int idx = closest(s, fs.data(), n, A, B, tmin, kFltMax, t, &cands);
          if (depth == 0) {
            hsum += (uint32_t)(idx + 1);
            if (sample == 0) fh0 = idx;
          }
          if (idx < 0) {
            color = color * sky(B);
            break;
          }
          V3 ctr = v3(s[idx].center[0], s[idx].center[1], s[idx].center[2]);
          V3 p = A + t * B;
          V3 nrm = (p - ctr) / s[idx].radius;
          Material m{s[idx].type, v3(s[idx].albedo[0], s[idx].albedo[1], s[idx].albedo[2]), s[idx].param};
          V3 atten, nA, nB;
          if (!scatter(m, A, B, p, nrm, g, atten, nA, nB)) {
            color = v3(0, 0, 0);
            break;
          }
          A = nA;
          B = nB;
          color = atten * color;
          if (++depth >= max_depth) break;
        }
        sum = sum + color;
        paths++;
      }
      size_t o = (size_t)(row - row0) * nx + col;
      V3 mean;
      uint8_t px[3];
      finish_pixel(sum, nsamples, mean, px);
      if (rgb8) std::memcpy(rgb8 + 3 * o, px, 3);
      if (linear) { linear[3 * o] = mean.x; linear[3 * o + 1] = mean.y; linear[3 * o + 2] = mean.z; }
      if (first_hit) first_hit[o] = fh0;
      if (hit_sum) hit_sum[o] = hsum;
    }
  }
  if (stats) { stats[0] = paths; stats[1] = segs; stats[2] = cands; }
}

// div_by_widened_float against the plain division: random quotients and the
// hard ones (x placed half an ulp of the quotient away from q * da, i.e. as
// close to a rounding boundary as a double x can be).  Returns mismatches.
uint64_t core_division_mismatches(uint64_t n, uint64_t seed) {
  uint64_t s = seed * 0x9e3779b97f4a7c15ull + 88172645463325252ull, bad = 0;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (uint64_t i = 0; i < n; i++) {
    uint32_t fb = (uint32_t)rnd();
    if (i & 1) fb = (fb & 0xff800000u) | (0x7fffffu - (uint32_t)(rnd() & 7));   // significands of all ones
    float a;
    memcpy(&a, &fb, 4);
    a = std::fabs(a);
    if (!(a > 1e-30f && a < 1e30f)) continue;
    const double da = a, r = 1.0 / da;
    uint64_t xb = (rnd() & 0x800fffffffffffffull) | ((uint64_t)(1023 - 100 + (rnd() % 200)) << 52);
    double x;
    memcpy(&x, &xb, 8);
    if (i % 7 == 0) {
      const double q = x / da, qn = std::nextafter(q, INFINITY);
      const long double mid = ((long double)q + (long double)qn) / 2;
      x = (double)(mid * (long double)da);
    }
    const double want = x / da, got = div_by_widened_float(x, da, r);
    if (memcmp(&want, &got, 8) != 0) bad++;
  }
  return bad;
}

// narrowed_double_reciprocal against the expression it replaces, for every float of one sign bit
// (first .. last bit patterns); and div_f32_by against the float divide on random operands.
uint64_t core_reciprocal_mismatches(uint32_t first, uint32_t last) {
  uint64_t bad = 0;
  for (uint64_t b = first; b <= last; b++) {
    const uint32_t u = (uint32_t)b;
    float x;
    memcpy(&x, &u, 4);
    if (!(x == x)) continue;
    const float want = (float)(1.0 / (double)x), got = narrowed_double_reciprocal(x);
    if (memcmp(&want, &got, 4) != 0) bad++;
  }
  return bad;
}

// draws first..last: the double and the float form of r / (2^31 - 1) against the division (c/rand.c:15)
uint64_t core_unit_mismatches(uint32_t first, uint32_t last) {
  uint64_t bad = 0;
  for (uint64_t b = first; b <= last; b++) {
    const uint32_t r = (uint32_t)b;
    const double want = (double)r / 2147483647.0, got = unit_from_bits(r);
    const float wantf = (float)want, gotf = unit_float_from_bits(r);
    if (memcmp(&want, &got, 8) != 0 || memcmp(&wantf, &gotf, 4) != 0) bad++;
  }
  return bad;
}

uint64_t core_float_division_mismatches(uint64_t n, uint64_t seed) {
  uint64_t s = seed * 0x9e3779b97f4a7c15ull + 88172645463325252ull, bad = 0;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (uint64_t i = 0; i < n; i++) {
    uint32_t kb = (uint32_t)rnd(), xb = (uint32_t)rnd();
    if (i & 1) kb = (kb & 0xff800000u) | (0x7fffffu - (uint32_t)(rnd() & 7));
    float k, x;
    memcpy(&k, &kb, 4);
    memcpy(&x, &xb, 4);
    if (!(std::fabs(k) > 1e-30f && std::fabs(k) < 1e30f) || !(std::fabs(x) < 1e30f)) continue;
    if (i % 5 == 0) {   // quotients next to a float midpoint
      const float q = x / k, qn = std::nextafter(q, INFINITY);
      x = (float)(((double)q + (double)qn) / 2 * (double)k);
      if (!std::isfinite(x)) continue;   // (an infinite dividend gives NaN instead of infinity; a hit point is finite)
    }
    const float want = x / k, got = div_f32_by(x, k, 1.0 / (double)k);
    if (memcmp(&want, &got, 4) != 0 && !(want == 0.0f && got == 0.0f)) bad++;
  }
  return bad;
}


// filter_discriminant for every (ray, sphere) pair, ray-major: the plain statement that the device's
// packed routines are compared with (spheres: centre and radius, 4 floats each)
void core_filter_values(const float *rays, size_t n_rays, const float *spheres, size_t n_spheres, float *out) {
  std::vector<FilterSphere> fs(n_spheres);
  for (size_t i = 0; i < n_spheres; i++) fs[i] = make_filter_sphere(spheres[4 * i], spheres[4 * i + 1], spheres[4 * i + 2], spheres[4 * i + 3]);
  for (size_t r = 0; r < n_rays; r++) {
    const FilterRay fr = make_filter_ray(v3(rays[6 * r], rays[6 * r + 1], rays[6 * r + 2]), v3(rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]));
    for (size_t i = 0; i < n_spheres; i++) out[r * n_spheres + i] = filter_discriminant(fr, fs[i].cx, fs[i].cy, fs[i].cz, fs[i].kprime);
  }
}

// box_pass for every (ray, box) pair given the per-ray coefficients (9 floats: ix iy iz nx ny nz ax ay az)
// and boxes (centre, half extents: 6 floats each)
void core_box_pass(const float *box_ray, size_t n_rays, const float *boxes, size_t n_boxes, uint8_t *out) {
  for (size_t r = 0; r < n_rays; r++) {
    const float *v = box_ray + 9 * r;
    const BoxRay br{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]};
    for (size_t b = 0; b < n_boxes; b++) {
      const float *q = boxes + 6 * b;
      out[r * n_boxes + b] = box_pass(br, GroupBox{q[0], q[1], q[2], q[3], q[4], q[5]}) ? 1 : 0;
    }
  }
}

// make_box_ray as the host forms it (its reciprocal is the correctly rounded one, the device's is within one ulp)
void core_make_box_ray(const float *rays, size_t n_rays, float sigma0, float sigma1, float *out) {
  for (size_t r = 0; r < n_rays; r++) {
    const BoxRay b = make_box_ray(v3(rays[6 * r], rays[6 * r + 1], rays[6 * r + 2]), v3(rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]), sigma0, sigma1);
    const float v[9] = {b.ix, b.iy, b.iz, b.nx, b.ny, b.nz, b.ax, b.ay, b.az};
    std::memcpy(out + 9 * r, v, sizeof v);
  }
}

// the scan layout's boxes for a scene (group boxes, then super-group boxes), 6 floats each; returns the count
size_t core_layout_boxes(const rt_sphere *s, size_t n, float *out, size_t cap, float *sigma0, float *sigma1) {
  const rtcluster::Layout L = rtcluster::build_layout(s, n, 512);
  size_t k = 0;
  for (const auto *list : {&L.boxes, &L.supers, &L.batches})
    for (const GroupBox &b : *list) {
      if (b.hx > 1.0e29f || b.hx < 0.0f) continue;   // "always" and "never" markers are not boxes
      if (out && k < cap) {
        const float v[6] = {b.cx, b.cy, b.cz, b.hx, b.hy, b.hz};
        std::memcpy(out + 6 * k, v, sizeof v);
      }
      k++;
    }
  if (sigma0) *sigma0 = L.sigma0;
  if (sigma1) *sigma1 = L.sigma1;
  return k;
}


// A draw has 23 random bits: r = m << 8.  For every m: the reference's conversion narrowed to float,
// (float)((double)r / 2147483647.0) (c/rand.c:15 then c/vec.c:68), is exactly m * 2^-23, and the unit-ball
// coordinate 2*u - 1 formed in two float roundings equals the kernels' single 2*(1 + m * 2^-23) - 3.
uint64_t core_draw23_mismatches(void) {
  uint64_t bad = 0;
  for (uint32_t m = 0; m < (1u << 23); m++) {
    const uint32_t r = m << 8;
    const float ref_u = (float)((double)r / 2147483647.0);
    const float mine = (float)m * 1.1920928955078125e-07f;   // m * 2^-23, exact
    if (memcmp(&ref_u, &mine, 4) != 0) bad++;
    const float ref_c = 2.0f * ref_u - 1.0f;
    const float f = __builtin_bit_cast(float, 0x3f800000u | m);
    const float got_c = rt_fma(2.0f, f, -3.0f);
    if (memcmp(&ref_c, &got_c, 4) != 0) bad++;
    if (unit_float_from_bits(r) != ref_u) bad++;
  }
  return bad;
}

}  // extern "C"
