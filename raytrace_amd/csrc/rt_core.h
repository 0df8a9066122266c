// rt_core.h -- per-lane arithmetic of the path tracer, written once for the
// HIP kernels.  Every function is __host__ __device__ so that the unit tests
// can also run the very same source on the CPU and compare it bit for bit
// with the oracle; the product library only ever runs it on the GPU.
//
// Rounding contract: this file is compiled with -ffp-contract=off.  A plain
// `a*b + c` is two roundings, as in the reference built the same way; every
// fused multiply-add is spelled rt_fma().  float division and square root are
// the correctly rounded ones.  Where C's promotion rules make a reference
// expression a double expression, it is a double expression here too; each
// such place cites its line.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rtcore {

constexpr float kFltEpsilon = 1.1920928955078125e-07f;
constexpr float kFltMax = 3.4028234663852886e+38f;

RT_HD float rt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

struct V3 {
  float x, y, z;
};

RT_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RT_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_HD V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_HD V3 operator*(float k, V3 a) { return V3{k * a.x, k * a.y, k * a.z}; }
RT_HD V3 operator/(V3 a, float k) { return V3{a.x / k, a.y / k, a.z / k}; }
RT_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
// c/vec.c:48-50
RT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// c/vec.c:37-39
RT_HD float length(V3 a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
// c/vec.c:41-46 forms the reciprocal as a double divide narrowed to float, (float)(1.0 / (double)len).
// That is the correctly rounded float reciprocal: 1/len for a 24-bit len is at least 2^-49 (relative)
// away from every midpoint of two floats, so rounding it to double first cannot move it across one
// (checked for every float by the test suite).  A float divide is half the instructions.
RT_HD float narrowed_double_reciprocal(float x) { return 1.0f / x; }

RT_HD V3 normalize(V3 a) {
  float k = narrowed_double_reciprocal(length(a));
  return k * a;
}
// c/vec.c:60-62
RT_HD V3 reflect(V3 u, V3 n) {
  float k = 2.0f * dot(u, n);
  return u - k * n;
}

// ---------------------------------------------------------------- generator
//
// Counter generator: draw k of path (seed, pixel, sample) is the 31-bit
// integer r = (draw_hash(key + k*0x9e3779b9) >> 9) << 8 -- 23 random bits, the
// low 8 zero -- returned as the double r / 2147483647 exactly as c/rand.c:15
// converts its integer.  23 bits because most draws are narrowed to float at
// once (c/vec.c:68), where r / 2147483647 for such an r IS m * 2^-23 (m = r >>
// 8): the quotient exceeds m * 2^-23 by a relative 2^-31, far inside half a
// float ulp of a 23-bit significand, so the narrowing needs no double
// arithmetic at all (checked for every m by the host tests).
//
// draw_hash is two multiply-xorshift rounds; its top 23 bits are what the
// draw uses, so the closing xorshift of a full avalanche hash (which only
// touches the low half) is left out, and both xorshifts are by 16, one
// instruction each on the device.  On triples of consecutive draws (the
// unit-ball points) its statistics are those of the full-avalanche mix32
// below, which still makes the per-path keys.

RT_HD uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

RT_HD uint32_t path_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
  uint32_t a = mix32(pixel + 0x9e3779b9u * seed);
  return mix32(a ^ (sample * 0x85ebca6bu + 0x165667b1u));
}

RT_HD uint32_t draw_hash(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 16;
  x *= 0x846ca68bu;
  return x;
}

// r / (2^31 - 1) correctly rounded to double without a division:
// r/(2^31-1) = r*2^-31 * (1 + 2^-31 + 2^-62 + ...).  x = r*2^-31 is exact; one
// fused multiply-add x*(2^-31 + 2^-62) + x rounds a value that differs from
// the quotient by less than x*2^-92, far inside the gap a 31-bit r leaves
// around any rounding boundary.  The host tests check all 2^31 values against
// the division.
RT_HD double unit_from_bits(uint32_t r31) {
  double x = (double)r31 * 4.656612873077392578125e-10;  // 2^-31
  const double tail = 4.656612873077392578125e-10 + 2.1684043449710088680149056017398834228515625e-19;
  return __builtin_fma(x, tail, x);
}

// (float)(r / (2^31 - 1)) where the draw is narrowed at once (c/vec.c:67-68):
// one product by 2^-31 (1 + 2^-31), the double nearest 1/(2^31 - 1).  The
// product is not always the correctly rounded double quotient, but it lies on
// the same side of every float rounding boundary; the host tests check all
// 2^31 values.
RT_HD float unit_float_from_bits(uint32_t r31) { return (float)((double)r31 * 0x1.00000002p-31); }

// the stream of one path: the hash input of its next draw, key + k * 0x9e3779b9 after k draws
struct Rng {
  uint32_t x;
};
RT_HD Rng rng_at(uint32_t key, uint32_t k = 0u) { return Rng{key + k * 0x9e3779b9u}; }

// the 31-bit integer of the draw whose hash input is x = key + k * 0x9e3779b9
RT_HD uint32_t bits_at(uint32_t x) { return (draw_hash(x) >> 9) << 8; }

// (float)(r / (2^31 - 1)) + 1 for that draw, i.e. 1 + m * 2^-23: the 23 bits are the significand of a float
// in [1, 2) (one funnel shift on the device)
RT_HD float one_plus_unit_at(uint32_t x) {
  const uint32_t h = draw_hash(x);
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_bit_cast(float, __builtin_amdgcn_alignbit(0x7fu, h, 9u));
#else
  return __builtin_bit_cast(float, 0x3f800000u | (h >> 9));
#endif
}

RT_HD uint32_t next_bits(Rng &g) {
  const uint32_t r = bits_at(g.x);
  g.x += 0x9e3779b9u;
  return r;
}

RT_HD double next_uniform(Rng &g) { return unit_from_bits(next_bits(g)); }


// c/vec.c:64-72.  The reference tests norm = sqrtf(s) against 1 and
// FLT_EPSILON; for a correctly rounded square root sqrtf(s) < 1 exactly when
// s < 1 (the float below 1 has a root that rounds below 1) and sqrtf(s) <
// 2^-23 exactly when s < 2^-46, so the loop tests s itself.
// one attempt from the three draws whose hash inputs are x, x + w, x + 2w (w = 0x9e3779b9): the
// candidate, and whether the loop would keep it
RT_HD bool unit_ball_try_at(uint32_t x, V3 &u) {
  const float fx = one_plus_unit_at(x), fy = one_plus_unit_at(x + 0x9e3779b9u), fz = one_plus_unit_at(x + 2u * 0x9e3779b9u);
  // 2*u - 1 (c/vec.c:67-68) with u = f - 1 a multiple of 2^-23: 2f - 3, exact (a multiple of 2^-22 below 1 in
  // magnitude), hence the value the reference's two roundings give
  u = v3(rt_fma(2.0f, fx, -3.0f), rt_fma(2.0f, fy, -3.0f), rt_fma(2.0f, fz, -3.0f));
  float s = u.x * u.x + u.y * u.y + u.z * u.z;
  return s < 1.0f && !(s < 1.4210854715202004e-14f);   // 2^-46
}

RT_HD bool unit_ball_try(Rng &g, V3 &u) {
  const bool keep = unit_ball_try_at(g.x, u);
  g.x += 3u * 0x9e3779b9u;
  return keep;
}

RT_HD V3 unit_ball(Rng &g) {
  V3 u;
  while (!unit_ball_try(g, u)) {
  }
  return u;
}

// ------------------------------------------------------------------ camera

struct Camera {  // same layout as rt_camera
  V3 llc, horiz, vert, origin, u, v, w;
  float lens_radius;
};

// c/structs.c:81-92, given the point `ball` drawn by random_in_unit_ball (c/structs.c:83)
RT_HD void camera_ray_from(const Camera &c, float x, float y, V3 ball, V3 &A, V3 &B) {
  V3 rd = c.lens_radius * ball;
  V3 off = rd.x * c.u + rd.y * c.v;
  V3 on_plane = c.llc + (x * c.horiz + y * c.vert);
  B = (on_plane - c.origin) - off;
  A = c.origin + off;
}

RT_HD void camera_ray(const Camera &c, float x, float y, Rng &g, V3 &A, V3 &B) {
  camera_ray_from(c, x, y, unit_ball(g), A, B);
}

// pixel -> (x, y) of c/main.c:465-466 with one band (ymin 0, ymax 1);
// `up` = rows below this one = the reference's j-1
RT_HD void pixel_xy(uint32_t col, uint32_t up, uint32_t nx, uint32_t ny, Rng &g, float &x, float &y) {
  x = (float)((double)col + next_uniform(g)) / (float)nx;
  y = (float)((double)up + next_uniform(g)) / (float)ny;
}

// pixel_xy with the two divisions done by div_f32_by (below); inv_nx = 1.0 / (double)(float)nx, same for ny
RT_HD float div_f32_by(float x, float k, double inv_k);
RT_HD void pixel_xy_by(uint32_t col, uint32_t up, uint32_t nx, uint32_t ny, double inv_nx, double inv_ny, Rng &g, float &x,
                       float &y) {
  x = div_f32_by((float)((double)col + next_uniform(g)), (float)nx, inv_nx);
  y = div_f32_by((float)((double)up + next_uniform(g)), (float)ny, inv_ny);
}

// ------------------------------------------------------- conservative filter
//
// The scan does not evaluate the reference's discriminant for every sphere.
// It evaluates a cheaper quantity Df with fused arithmetic in expanded form,
//   b  = A.B - C.B
//   c0 = (|C|^2 - r^2)' - 2 A.C            (' = lowered by the margin below)
//   Df = b*b - a*(|A|^2)(1-kappa) - a*c0
// such that  D_reference > 0  implies  Df > 0  (no false negatives), and runs
// the reference's exact arithmetic only for the spheres that pass.  The
// margin kappa*a*(|A|^2 + |C|^2 + r^2) with kappa = 2^-17 = 128 u covers both
// the rounding of this expansion (< 24 u a (M^2 + r^2), M = |A|+|C|) and the
// rounding of the reference's own discriminant (< 20 u a (|A-C|^2 + r^2)),
// using M^2 <= 2(|A|^2+|C|^2) and u = 2^-24.

constexpr double kFilterKappa = 1.0 / 131072.0;  // 2^-17

struct FilterSphere {  // one LDS entry, 16 B
  float cx, cy, cz, kprime;
};

// host side only, at scene upload (double arithmetic, result rounded down)
inline FilterSphere make_filter_sphere(float cx, float cy, float cz, float r) {
  double cc = (double)cx * cx + (double)cy * cy + (double)cz * cz;
  double rr = (double)r * r;
  double k = cc - rr - kFilterKappa * (cc + rr);
  float kf = (float)k;
  if ((double)kf > k) kf = __builtin_nextafterf(kf, -kFltMax);
  return FilterSphere{cx, cy, cz, kf};
}

struct FilterRay {  // per segment, per lane
  float nbx, nby, nbz;  // -B
  float ab;             // A.B
  float m2ax, m2ay, m2az;  // -2A
  float na;             // -(B.B)
  float naaa;           // -(B.B)(A.A)(1-kappa), rounded up in magnitude-safe direction
};

RT_HD FilterRay make_filter_ray(V3 A, V3 B) {
  double a = (double)B.x * B.x + (double)B.y * B.y + (double)B.z * B.z;
  double aa = (double)A.x * A.x + (double)A.y * A.y + (double)A.z * A.z;
  double ab = (double)A.x * B.x + (double)A.y * B.y + (double)A.z * B.z;
  FilterRay f;
  f.nbx = -B.x; f.nby = -B.y; f.nbz = -B.z;
  f.m2ax = -2.0f * A.x; f.m2ay = -2.0f * A.y; f.m2az = -2.0f * A.z;
  f.ab = (float)ab;
  f.na = -(float)a;
  f.naaa = (float)(-(a * aa) * (1.0 - kFilterKappa));
  return f;
}

// eight fused multiply-adds per ray-sphere pair
RT_HD float filter_discriminant(const FilterRay &f, float cx, float cy, float cz, float kprime) {
  float b = rt_fma(f.nbx, cx, rt_fma(f.nby, cy, rt_fma(f.nbz, cz, f.ab)));
  float c0 = rt_fma(f.m2ax, cx, rt_fma(f.m2ay, cy, rt_fma(f.m2az, cz, kprime)));
  return rt_fma(f.na, c0, rt_fma(b, b, f.naaa));
}

// The same for two spheres at once on packed FP32 (v_pk_fma_f32: two floats
// per lane per instruction, the form that reaches the chip's FP32 peak).  The
// arithmetic per half is exactly filter_discriminant's, so the two give the
// same bits.
#if defined(__clang__)
typedef float f32x2 __attribute__((ext_vector_type(2)));
#else
typedef float f32x2 __attribute__((vector_size(8)));
#endif

RT_HD f32x2 splat2(float v) {
  f32x2 r;
  r[0] = v;
  r[1] = v;
  return r;
}

RT_HD f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_elementwise_fma(a, b, c);
#else
  f32x2 r;
  r[0] = __builtin_fmaf(a[0], b[0], c[0]);
  r[1] = __builtin_fmaf(a[1], b[1], c[1]);
  return r;
#endif
}

struct FilterRay2 {  // FilterRay with every coefficient in both halves
  f32x2 nbx, nby, nbz, ab, m2ax, m2ay, m2az, na, naaa;
};

RT_HD FilterRay2 widen(const FilterRay &f) {
  FilterRay2 w;
  w.nbx = splat2(f.nbx); w.nby = splat2(f.nby); w.nbz = splat2(f.nbz); w.ab = splat2(f.ab);
  w.m2ax = splat2(f.m2ax); w.m2ay = splat2(f.m2ay); w.m2az = splat2(f.m2az);
  w.na = splat2(f.na); w.naaa = splat2(f.naaa);
  return w;
}

// cx = {cx of sphere 2p, cx of sphere 2p+1}, ...
RT_HD f32x2 filter_discriminant2(const FilterRay2 &f, f32x2 cx, f32x2 cy, f32x2 cz, f32x2 kprime) {
  f32x2 b = pk_fma(f.nbx, cx, pk_fma(f.nby, cy, pk_fma(f.nbz, cz, f.ab)));
  f32x2 c0 = pk_fma(f.m2ax, cx, pk_fma(f.m2ay, cy, pk_fma(f.m2az, cz, kprime)));
  return pk_fma(f.na, c0, pk_fma(b, b, f.naaa));
}

// ------------------------------------------------------------ exact hit test

// ----------------------------------------------------------- group box test
//
// Spheres are scanned in groups (rt_cluster.h); a ray skips a group when it
// misses the group's axis-aligned box, tested on the half line t >= 0 with the
// slab method in centre / half-extent form:
//   m_k = (c_k - A_k) / B_k,  near_k = m_k - H_k / |B_k|,  far_k = m_k + H_k / |B_k|
//   pass  <=>  min_k far_k >= max(max_k near_k, 0)
// with H_k = h_k (1 + sigma), h_k the stored half extent and sigma = sigma0 +
// sigma1 |A|^2 a per-ray factor (folded into 1/|B_k| once per ray, so the test
// costs the same as with fixed boxes).  The test only has to be conservative:
// a sphere the reference reports as hit (c/main.c:214-236 with t > tmin >= 0)
// must lie in a group that passes.
//  * If the reference's discriminant is positive, the sphere passes the
//    conservative filter above, whose margin puts the true line within rho_i
//    of the centre, rho_i^2 = r_i^2 + 1.375 kappa (|A|^2 + |C_i|^2 + r_i^2), hence
//    rho_i <= r_i + a_i + b_i |A|^2,  a_i = 1.375 kappa (|C_i|^2 + r_i^2) / (2 r_i),
//    b_i = 1.375 kappa / (2 r_i).  The closest approach is at t* = -b/a.
//  * If the reference also reports a root t > 0 then either its computed b is
//    negative, so the true t* is above -4u |A-C| / |B| and the origin is within
//    2^-22 (|A|+|C|) of a point of that ball; or b >= 0 and the larger root is
//    positive, which needs sqrt(D) > b, i.e. a*c <= 2u b^2, i.e. |A-C| <= r (1 +
//    8u): the origin is inside the ball.  Either way the half line t >= 0 meets
//    the ball of radius rho_i + 2^-21 (|A| + |C_i| + r_i).
//  * The stored box holds the part of that radius that does not depend on the
//    ray, r_i + a_i + 2^-21 (|C_i| + r_i); the rest, b_i |A|^2 + 2^-21 |A|, is
//    below h_min sigma when sigma0 = 2^-19 / h_min, sigma1 = (max b_i + 2^-19) /
//    h_min (h_min: the smallest stored half extent of the scene).
//  * The float evaluation below (one-ulp reciprocal, three roundings per slab)
//    is off by less than 0.75 * 2^-21 (H_k + |c_k| + |A_k|) / |B_k| per bound;
//    the stored half extents are widened for the first two terms
//    (rt_cluster.h) and sigma has room for the third.
// Rays with a direction component below 2^-30 in magnitude (its reciprocal
// would leave the range), |B|^2 >= 2^40 or |A|^2 >= 2^40 are not culled at all
// (box_ray_in_range).

// Distance cut (streamed scans of large scenes).  Once a ray has a hit at t_best, a box whose computed
// near bound max_k near_k exceeds
//   cut = t_best (1 + 2^-8) + 2^-7 (reach_max + b_max (|A|^2 + 1)) / |B|
// holds no sphere the reference would report nearer, and is skipped.  Why: let the reference report the
// root t_c for a sphere behind that box.  The quantities b, c, D of c/main.c:216-219 are float
// evaluations of the true b*, c*, D* with |b - b*| <= 4u |oc| |B| and |D - D*| <= (19u |oc|^2 + 7u r^2) |B|^2
// (u = 2^-24), so the root differs from a parameter t' of the TRUE line -- the true root where D* >= 0,
// else the closest approach, which the filter's margin keeps within rho_i of the centre -- by at most
// (2^-9.9 |oc| + 2^-10.6 r) / |B| (from |sqrt(D) - sqrt(D*)| <= sqrt|D - D*|); the double-precision root and
// its narrowing add 2^-24 relative.  The point at t' lies in the ball of radius rho_i <= reach_i + b_i |A|^2 +
// 2^-21 |A| the box was built around, so the computed near bound is at most t' (the box test's own
// argument above: computed bounds bracket every point of that ball), and |oc| <= |B| t' + 2 rho_i.
// Together: near <= t' <= t_c (1 + 2^-9.7) + 2^-8.5 rho_i / |B|.  A box with near > cut therefore has
// t_c > t_best for every sphere in it -- strictly, so the lowest-index rule for equal t is not touched.
// t' < 0 gives near < 0 <= cut: never skipped.  NaN bounds compare false and are not skipped.
constexpr float kCutRel = 1.00390625f;      // 1 + 2^-8
constexpr float kCutAbs = 0.0078125f;       // 2^-7
// the part of `cut` that does not depend on the hit: inv_len >= 1 / |B|
RT_HD float cut_slack(float reach_max, float b_max, float a2, float inv_len) {
  return kCutAbs * (reach_max + b_max * (a2 + 1.0f)) * inv_len * 1.001f;
}

struct GroupBox {  // centre and half extents; h = +1e30: always scanned, h = -1e30: never (padding)
  float cx, cy, cz, hx, hy, hz;
};

struct BoxRay {
  float ix, iy, iz;  // 1 / B_k (within one ulp)
  float nx, ny, nz;  // -A_k * (1 / B_k)
  float ax, ay, az;  // (1 + sigma) / |B_k|, rounded up
};

constexpr float kBoxMinDir = 9.313225746154785e-10f;       // 2^-30
constexpr float kBoxMaxNorm2 = 1099511627776.0f;           // 2^40

RT_HD bool box_ray_in_range(V3 A, V3 B) {
  float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(B.x), __builtin_fabsf(B.y)), __builtin_fabsf(B.z));
  return dot(A, A) < kBoxMaxNorm2 && lo >= kBoxMinDir && dot(B, B) < kBoxMaxNorm2;
}

RT_HD float rt_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}

// sigma0, sigma1: the scene's (rt_cluster.h).  The factor 1 + 2^-19 covers the roundings of this
// function (|A|^2, sigma, the products) and the one-ulp reciprocal.
RT_HD BoxRay make_box_ray(V3 A, V3 B, float sigma0, float sigma1) {
  BoxRay r;
  r.ix = rt_rcp(B.x); r.iy = rt_rcp(B.y); r.iz = rt_rcp(B.z);
  r.nx = -A.x * r.ix; r.ny = -A.y * r.iy; r.nz = -A.z * r.iz;
  const float grow = (1.0f + rt_fma(sigma1, dot(A, A), sigma0)) * 1.0000019073486328f;   // (1 + sigma)(1 + 2^-19)
  r.ax = __builtin_fabsf(r.ix) * grow; r.ay = __builtin_fabsf(r.iy) * grow; r.az = __builtin_fabsf(r.iz) * grow;
  return r;
}

// nine fused multiply-adds, two three-way min/max, one max, one compare
RT_HD bool box_pass(const BoxRay &r, const GroupBox &b) {
  float mx = rt_fma(r.ix, b.cx, r.nx), my = rt_fma(r.iy, b.cy, r.ny), mz = rt_fma(r.iz, b.cz, r.nz);
  float nx = rt_fma(-r.ax, b.hx, mx), ny = rt_fma(-r.ay, b.hy, my), nz = rt_fma(-r.az, b.hz, mz);
  float fx = rt_fma(r.ax, b.hx, mx), fy = rt_fma(r.ay, b.hy, my), fz = rt_fma(r.az, b.hz, mz);
  float tn = __builtin_fmaxf(__builtin_fmaxf(nx, ny), nz);
  float tf = __builtin_fminf(__builtin_fminf(fx, fy), fz);
  return tf >= __builtin_fmaxf(tn, 0.0f);
}

// x / da, correctly rounded, for a divisor that is a float widened to double,
// given r = 1.0 / da (correctly rounded).  q0 = RN(x r) is within 2^-52 of the
// quotient; its residual e0 = x - q0 da is exact in double because da has only
// 24 significant bits; q0 + e0 r is then within 2^-105 (relative) of x / da,
// and a quotient of a double by a 24-bit divisor is either exact or at least
// 2^-78 (relative) away from every rounding boundary, so rounding that sum
// gives RN(x / da).  Three operations instead of a division per root
// (c/main.c:220, 231 divide by the same `a` for every sphere a ray meets).
// Infinite or NaN operands give NaN where the division gives an infinity; the
// range test that follows fails for both.
// (The argument above needs r only to within a few ulp: with r = (1 + eps) / da the residual e0 is still exact
// -- it is below 2^-50 |x| and a multiple of ulp(q0) ulp(da) -- and the corrected sum is within |eps| 2^-51 of
// the quotient, far inside the 2^-78 gap.)
//
// The reciprocal itself, for a positive finite float widened to double: the hardware estimate and two
// Newton steps (within an ulp of 1 / da) instead of a full division.
RT_HD double reciprocal_of_widened_float(double da) {
#ifdef __HIP_DEVICE_COMPILE__
  double r = __builtin_amdgcn_rcp(da);
  double e = __builtin_fma(-da, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-da, r, 1.0);
  return __builtin_fma(r, e, r);
#else
  return 1.0 / da;
#endif
}

RT_HD double div_by_widened_float(double x, double da, double r) {
  double q0 = x * r;
  double e0 = __builtin_fma(-q0, da, x);
  return __builtin_fma(e0, r, q0);
}

// The float quotient x / k, given inv_k = 1.0 / (double)k (correctly rounded), for divisors that are
// reused: a sphere's radius (c/main.c:229), the image size (c/main.c:465-466).  The double quotient
// of two floats is at least 2^-49 (relative) away from every midpoint of two floats unless it is a
// float itself, so narrowing the correctly rounded double quotient gives the correctly rounded float
// quotient: five operations instead of the ten of a float divide.
RT_HD float div_f32_by(float x, float k, double inv_k) {
  return (float)div_by_widened_float((double)x, (double)k, inv_k);
}

// sqrt((double)D) for a float D > 0 (c/main.c:220 promotes the discriminant).  On the device the
// compiler's correctly rounded square root first rescales arguments below 2^-767 and patches zero and
// infinity afterwards; a widened positive float is never that small, so those eight instructions are
// dropped here and the rest -- v_rsq_f64 and the compiler's own correction steps, operation for
// operation -- is kept: the same bits for every finite argument.  An infinite D (overflowed products)
// gives NaN here and infinity there; exact_hit's range tests fail for both roots either way.
RT_HD double sqrt_of_widened_positive_float(double x) {
#ifdef __HIP_DEVICE_COMPILE__
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
#else
  return __builtin_sqrt(x);
#endif
}

// c/main.c:214-236 for one sphere.  r2 = radius*radius (a float product, as
// the reference forms it); inv_a = 1.0 / (double)a.  On a hit inside
// (tmin, tmax) returns true and t.
RT_HD bool exact_hit(V3 A, V3 B, float a, double inv_a, float cx, float cy, float cz, float r2, float tmin, float tmax,
                     float &t) {
  V3 oc = A - v3(cx, cy, cz);
  float b = dot(oc, B);
  float c = dot(oc, oc) - r2;
  float D = b * b - a * c;
  if (D > 0.0f) {
    double root = sqrt_of_widened_positive_float((double)D);
    double nb = (double)(-b), da = (double)a;
    float t0 = (float)div_by_widened_float(nb - root, da, inv_a);
    if (tmin < t0 && t0 < tmax) { t = t0; return true; }
    float t1 = (float)div_by_widened_float(nb + root, da, inv_a);
    if (tmin < t1 && t1 < tmax) { t = t1; return true; }
  }
  return false;
}

// --------------------------------------------------------------- materials

// c/main.c:140-144.  (1-cosine)^5 by three double products instead of libm
// pow: agrees with it to a few units of the last double place, which the
// narrowing to float hides except about once in 2^27 calls.
// The two lines of c/main.c:141-142 depend on the sphere alone: r0 = ((1 - n) / (1 + n))^2 in float
// arithmetic.  schlick_r0 computes them (on the host at scene upload, or here), schlick_with the rest.
RT_HD float schlick_r0(float ref_idx) {
  float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
  return r0 * r0;
}

RT_HD float schlick_with(float cosine, float r0) {
  double x = (double)(1.0f - cosine);
  double x2 = x * x;
  double x5 = x2 * x2 * x;
  return (float)((double)r0 + (double)(1.0f - r0) * x5);
}

RT_HD float schlick(float cosine, float ref_idx) {
  float r0 = schlick_r0(ref_idx);
  double x = (double)(1.0f - cosine);
  double x2 = x * x;
  double x5 = x2 * x2 * x;
  return (float)((double)r0 + (double)(1.0f - r0) * x5);
}

// c/main.c:146-158, given un = normalize(u)
RT_HD bool refract_unit(V3 un, V3 n, float ni_over_nt, V3 &out) {
  float dt = dot(un, n);
  float D = (float)(1.0 - (double)(ni_over_nt * ni_over_nt * (1.0f - dt * dt)));
  if (D > 0.0f) {
    float rootD = __builtin_sqrtf(D);
    out = ni_over_nt * (un - dt * n) - rootD * n;
    return true;
  }
  return false;
}

RT_HD bool refract(V3 u, V3 n, float ni_over_nt, V3 &out) { return refract_unit(normalize(u), n, ni_over_nt, out); }

struct Material {
  int type;
  V3 albedo;
  float param;
};

// The three materials of c/main.c:163-211, split so that a caller can draw the
// unit-ball point of the matte and the metal case (their only use of the
// generator, and the first thing either does) in one place.
//
// matte, c/main.c:168-173: the new direction
RT_HD V3 scatter_matte(V3 p, V3 n, V3 ball) {
  V3 target = (p + n) + ball;
  return target - p;
}

// The sky, the metal and the dielectric case all start from normalize(B) (c/main.c:176, 148, 275) and
// the dielectric also from |B| (c/main.c:189, 194); the *_unit forms take them as arguments so that a
// caller whose lanes take different cases computes them once.
//
// metal, c/main.c:174-180: the new direction; false when the ray is absorbed (c/main.c:179)
RT_HD bool scatter_metal_unit(V3 unitB, V3 n, float fuzz, V3 ball, V3 &dir) {
  V3 mirrored = reflect(unitB, n);
  dir = mirrored + fuzz * ball;
  return dot(dir, n) > 0.0f;
}

RT_HD bool scatter_metal(V3 B, V3 n, float fuzz, V3 ball, V3 &dir) { return scatter_metal_unit(normalize(B), n, fuzz, ball, dir); }

// dielectric, c/main.c:181-206: the new direction
// r0 = schlick_r0(idx) and inv_idx = narrowed_double_reciprocal(idx), (float)(1.0 / (double)idx) of c/main.c:193,
// are per-sphere constants that a caller may have at hand
RT_HD V3 scatter_dielectric_unit(V3 B, V3 unitB, float lenB, V3 n, float idx, float r0, float inv_idx, Rng &g) {
  float bn = dot(B, n);
  V3 outward;
  float ni_over_nt, cosine;
  if (bn > 0.0f) {
    outward = -n;
    ni_over_nt = idx;
    cosine = idx * bn / lenB;
  } else {
    outward = n;
    ni_over_nt = inv_idx;
    cosine = -bn / lenB;
  }
  V3 bent;
  bool go_through = false;
  if (refract_unit(unitB, outward, ni_over_nt, bent)) {
    go_through = next_uniform(g) >= (double)schlick_with(cosine, r0);
  }
  return go_through ? bent : reflect(B, n);
}

RT_HD V3 scatter_dielectric(V3 B, V3 n, float idx, Rng &g) {
  return scatter_dielectric_unit(B, normalize(B), length(B), n, idx, schlick_r0(idx), narrowed_double_reciprocal(idx), g);
}

// c/main.c:163-211.  p, n: hit point and normal.  Returns false for an
// absorbed metal ray (c/main.c:179); otherwise atten and the new ray.
RT_HD bool scatter(const Material &m, V3 A, V3 B, V3 p, V3 n, Rng &g, V3 &atten, V3 &oA, V3 &oB) {
  (void)A;
  oA = p;
  if (m.type == 0) {
    atten = m.albedo;
    oB = scatter_matte(p, n, unit_ball(g));
    return true;
  }
  if (m.type == 1) {
    atten = m.albedo;
    return scatter_metal(B, n, m.param, unit_ball(g), oB);
  }
  atten = v3(1.0f, 1.0f, 1.0f);
  oB = scatter_dielectric(B, n, m.param, g);
  return true;
}

// sky of c/main.c:274-278, given unit_y = normalize(B).y
RT_HD V3 sky_unit(float unit_y) {
  float t = (float)(0.5 * ((double)unit_y + 1.0));
  float k = (float)(1.0 - (double)t);
  V3 white = k * v3(1.0f, 1.0f, 1.0f);
  V3 blue = t * v3(0.75f, (float)0.95, 1.0f);
  return white + blue;
}

RT_HD V3 sky(V3 B) { return sky_unit(normalize(B).y); }

// c/main.c:470-477: mean, square root, NaN scrub, truncation to 8 bits
RT_HD void finish_pixel(V3 sum, uint32_t nsamples, V3 &mean, uint8_t rgb[3]) {
  mean = sum / (float)nsamples;
  V3 c = v3(__builtin_sqrtf(mean.x), __builtin_sqrtf(mean.y), __builtin_sqrtf(mean.z));
  if (c.x != c.x) c.x = 0.0f;
  if (c.y != c.y) c.y = 0.0f;
  if (c.z != c.z) c.z = 0.0f;
  rgb[0] = (uint8_t)(int)(255.0f * c.x);
  rgb[1] = (uint8_t)(int)(255.0f * c.y);
  rgb[2] = (uint8_t)(int)(255.0f * c.z);
}

}  // namespace rtcore
