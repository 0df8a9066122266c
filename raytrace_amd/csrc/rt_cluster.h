// rt_cluster.h -- host side, at scene upload: the order in which the kernels
// scan the sphere list and the boxes that let a wave skip whole groups of it.
//
// The reference scans the list in index order (c/main.c:248-254).  Its result
// does not depend on that order except for ties: the hit is the smallest
// "first root inside (tmin, FLT_MAX)" over all spheres, and among equal t the
// lowest index wins (strict `<`, c/main.c:225, 249-251).  The kernels
// therefore scan in an order of their own -- large spheres first, the rest
// split recursively into groups of kGroup neighbours (every split at an
// aligned group boundary, along the axis that leaves the smaller boxes, so
// that aligned runs of groups are compact regions) -- and break ties by the
// original index.
//
// Each group of regular spheres has an axis-aligned box; a ray that misses the
// box on its half line t >= 0 skips the group (test and proof obligations in
// rt_core.h, "group box test").  The stored box of a group contains, for each
// member, the ball of radius
//   r_i + a_i + 2^-21 (|C_i| + r_i),   a_i = 1.375 kappa (|C_i|^2 + r_i^2) / (2 r_i)
// (kappa = 2^-17 is the margin of the conservative sphere filter), its half
// extents widened by 2^-20 (|c_k| + 1) + 2^-19 h_k for the rounding of the slab
// arithmetic.  What the test needs beyond that grows with the ray's distance
// from the world origin and is applied per ray as a factor 1 + sigma0 + sigma1
// |A|^2 on the half extents; this file computes sigma0 and sigma1.  Four
// consecutive groups form a super-group whose box (the box around the four,
// widened once more for its own slab arithmetic) is tested first; a wave skips
// the four box tests of a super-group no lane's ray can touch.  Above the
// super-groups the same again: four of them form a quarter, and in scenes too
// large to stay in LDS, which are scanned in batches of `batch_spheres`
// consecutive scan positions, each batch and each region of kRegionBatches batches has
// a box (around the boxes below it, widened the same way); a wave skips what
// none of its rays can touch.
//
// Scenes like the reference's are flat fields of small spheres on one huge
// one; boxes around neighbours are flat too, so most rays that fly over a
// group miss its box.  Spheres much larger than the median (and degenerate
// ones, radius below 2^-16) go to groups that every ray scans.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_core.h"

namespace rtcluster {

constexpr int kGroup = 8;                      // spheres per group = one chunk of the scan
constexpr int kSuper = 4;                      // groups per super-group (a box of its own, tested first)
constexpr int kQuarterSupers = 4;               // super-groups per quarter (128 scan positions: the level between super-group and batch)
#ifndef RT_REGION_BATCHES
#define RT_REGION_BATCHES 4
#endif
constexpr int kRegionBatches = RT_REGION_BATCHES;              // batches per region (the level above the batches of a streamed scene)
constexpr double kTinyRadius = 1.0 / 65536.0;   // radius below this: scanned by every ray (keeps sigma1 bounded)
constexpr double kLargeFactor = 3.0;           // radius > this * median radius: scanned by every ray

struct Layout {
  std::vector<uint32_t> order;        // scan position -> original index, 0xffffffff for a pad
  std::vector<rtcore::GroupBox> boxes;  // one per group; h = +1e30 means "always scan", -1e30 "never"
  std::vector<rtcore::GroupBox> supers; // one per kSuper consecutive groups: the box around their boxes
  std::vector<rtcore::GroupBox> batches; // one per batch_spheres scan positions (an even number of them): the box around its super-groups
  std::vector<rtcore::GroupBox> quarters; // one per kQuarterSupers super-groups (an even number of them): the box around their boxes
  std::vector<rtcore::GroupBox> regions; // one per kRegionBatches batches (an even number of them): the box around their boxes
  std::vector<uint8_t> real_in_group; // real (non-pad) spheres per group
  uint32_t n_always_groups = 0;       // the first so many groups are the "always scan" ones; no box in `supers` covers them
  uint32_t n_scan = 0;                // = order.size(), a multiple of 2*kSuper*kGroup (a pair of super-groups)
  float sigma0 = 0.0f, sigma1 = 0.0f; // per-ray growth of the half extents: 1 + sigma0 + sigma1 |A|^2 (rt_core.h)
  // Of the spheres behind boxes: the largest `reach` (radius plus the filter's margin) and the largest factor of |A|^2
  // in that margin, both rounded up -- what the distance cut of a streamed scan adds to a hit distance
  // before it lets that hit rule out a box (rtcore::cut_slack)
  float cut_reach = 0.0f, cut_b = 0.0f;
};

inline float round_up_to_float(double v) {
  float f = (float)v;
  if ((double)f < v) f = __builtin_nextafterf(f, rtcore::kFltMax);
  return f;
}

// batch_spheres: a multiple of 2 * kSuper * kGroup
inline Layout build_layout(const rt_sphere *s, size_t n, size_t batch_spheres) {
  Layout L;
  std::vector<float> radii(n);
  for (size_t i = 0; i < n; ++i) radii[i] = s[i].radius;
  std::vector<float> sorted = radii;
  std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
  const double median = sorted[n / 2];
  std::vector<uint32_t> large, regular;
  for (size_t i = 0; i < n; ++i)
    ((double)radii[i] > kLargeFactor * median || !((double)radii[i] >= kTinyRadius) ? large : regular).push_back((uint32_t)i);

  // Order of the regular spheres: recursive split until a part is one group.  The split point is not the median but the most aligned group boundary inside the part, counted
  // in scan positions of the whole list (the large spheres' groups come first): the multiple of the
  // largest power of two that falls strictly inside.  Every aligned run of 2, 4, 64 groups -- a round, a
  // super-group, a batch of the kernels -- is then a subtree, i.e. one compact region, instead of the end
  // of one region and the start of the next.  (Cut at the median, a wave of the 40,004-sphere field wanted
  // 8.9 batches per bounce; on the reference's final scene the median split already gave boxes a wave's
  // rays touch 26 % less often than groups cut from a Morton curve.)
  {
    std::vector<uint32_t> ordered;
    ordered.reserve(regular.size());
    struct Part { size_t b0, e0, g0; };              // [b0, e0) of `regular`, first group = absolute group g0
    std::vector<Part> stack;
    stack.push_back(Part{0, regular.size(), (large.size() + kGroup - 1) / kGroup});
    while (!stack.empty()) {
      const Part part = stack.back();
      stack.pop_back();
      const size_t b0 = part.b0, e0 = part.e0, cnt = e0 - b0;
      if (cnt <= (size_t)kGroup) {
        for (size_t k = b0; k < e0; ++k) ordered.push_back(regular[k]);
        continue;
      }
      const size_t groups = (cnt + kGroup - 1) / kGroup, g1 = part.g0 + groups;   // (groups >= 2)
      size_t mid = part.g0 + 1;
      for (size_t pw = (size_t)1 << 30; pw >= 1; pw >>= 1) {
        const size_t m = (part.g0 / pw + 1) * pw;
        if (m < g1) {
          mid = m;
          break;
        }
      }
      const size_t left = (mid - part.g0) * kGroup;
      // Along which axis: the one that leaves the two parts with the smallest boxes, each box's surface weighted by
      // the number of spheres behind it (the chance that a ray meets a box goes with its surface).
      auto by_axis = [&](int ax) {
        std::sort(regular.begin() + b0, regular.begin() + e0, [&](uint32_t a, uint32_t c) {
          return s[a].center[ax] != s[c].center[ax] ? s[a].center[ax] < s[c].center[ax] : a < c;
        });
      };
      auto surface = [&](size_t first, size_t last) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (size_t k = first; k < last; ++k)
          for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo[d], (double)s[regular[k]].center[d] - (double)s[regular[k]].radius);
            hi[d] = std::max(hi[d], (double)s[regular[k]].center[d] + (double)s[regular[k]].radius);
          }
        const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x * y + y * z + z * x;
      };
      int best_ax = 0;
      double best_cost = 1e300;
      for (int ax = 0; ax < 3; ++ax) {
        by_axis(ax);
        const double cost = surface(b0, b0 + left) * (double)left + surface(b0 + left, e0) * (double)(cnt - left);
        if (cost < best_cost) {
          best_cost = cost;
          best_ax = ax;
        }
      }
      if (best_ax != 2) by_axis(best_ax);
      stack.push_back(Part{b0 + left, e0, mid});   // popped second: the left part comes first in the order
      stack.push_back(Part{b0, b0 + left, part.g0});
    }
    regular.swap(ordered);
  }

  // centre (as the kernel will see it) and widened half extents of the box [lo, hi]
  auto widened_box = [&](const double lo[3], const double hi[3]) {
    float c[3], h[3];
    for (int d = 0; d < 3; ++d) {
      c[d] = (float)(0.5 * (lo[d] + hi[d]));
      const double h0 = std::max(hi[d] - (double)c[d], (double)c[d] - lo[d]);
      h[d] = round_up_to_float((h0 + std::ldexp(std::fabs((double)c[d]) + 1.0, -20)) * (1.0 + std::ldexp(1.0, -19)));
    }
    return rtcore::GroupBox{c[0], c[1], c[2], h[0], h[1], h[2]};
  };
  double b_max = 0.0, h_min = 1e300, reach_max = 0.0;
  auto emit_group = [&](const uint32_t *idx, size_t cnt, bool always) {
    for (size_t k = 0; k < (size_t)kGroup; ++k) L.order.push_back(k < cnt ? idx[k] : 0xffffffffu);
    L.real_in_group.push_back((uint8_t)cnt);
    if (always || cnt == 0) {
      const float h = always ? 1.0e30f : -1.0e30f;
      L.boxes.push_back(rtcore::GroupBox{0.0f, 0.0f, 0.0f, h, h, h});
      return;
    }
    const double kap = rtcore::kFilterKappa;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t k = 0; k < cnt; ++k) {
      const rt_sphere &m = s[idx[k]];
      double ci2 = 0;
      for (int d = 0; d < 3; ++d) ci2 += (double)m.center[d] * m.center[d];
      const double r = (double)m.radius, r2 = r * r;
      const double a_i = 1.375 * kap * (ci2 + r2) / (2.0 * r);
      b_max = std::max(b_max, 1.375 * kap / (2.0 * r));
      const double reach = r + a_i + std::ldexp(std::sqrt(ci2) + r, -21);
      reach_max = std::max(reach_max, reach);
      for (int d = 0; d < 3; ++d) {
        lo[d] = std::min(lo[d], (double)m.center[d] - reach);
        hi[d] = std::max(hi[d], (double)m.center[d] + reach);
      }
    }
    const rtcore::GroupBox box = widened_box(lo, hi);
    h_min = std::min(h_min, (double)std::min(box.hx, std::min(box.hy, box.hz)));
    L.boxes.push_back(box);
  };

  for (size_t k = 0; k < large.size(); k += kGroup) emit_group(&large[k], std::min<size_t>(kGroup, large.size() - k), true);
  for (size_t k = 0; k < regular.size(); k += kGroup)
    emit_group(&regular[k], std::min<size_t>(kGroup, regular.size() - k), false);
  while (L.boxes.size() % (2 * kSuper)) emit_group(nullptr, 0, false);   // whole pairs of super-groups
  // the box around boxes [first, last) of `from`, those below index `skip_below` left out: "never" if
  // none of them is a real box, "always" if one of them is
  auto box_around = [&](const std::vector<rtcore::GroupBox> &from, size_t first, size_t last, size_t skip_below) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    bool always = false, any = false;
    for (size_t g = std::max(first, skip_below); g < last && g < from.size(); ++g) {
      const rtcore::GroupBox &b = from[g];
      if (b.hx > 1.0e29f) always = true;
      if (b.hx > 1.0e29f || b.hx < 0.0f) continue;
      any = true;
      const float c[3] = {b.cx, b.cy, b.cz}, h[3] = {b.hx, b.hy, b.hz};
      for (int d = 0; d < 3; ++d) {
        lo[d] = std::min(lo[d], (double)c[d] - (double)h[d]);
        hi[d] = std::max(hi[d], (double)c[d] + (double)h[d]);
      }
    }
    if (always || !any) {
      const float h = always ? 1.0e30f : -1.0e30f;
      return rtcore::GroupBox{0.0f, 0.0f, 0.0f, h, h, h};
    }
    return widened_box(lo, hi);
  };
  // The kernels put the first "always" groups (up to kSuper of them: one list's worth) on every ray's
  // list without a test, so the box of their super-group is the box around its other groups; a batch
  // that holds such a group is staged for every ray all the same.  Further "always" groups, if a scene
  // has that many large spheres, make their super-group an "always" one.
  L.n_always_groups = (uint32_t)std::min<size_t>((large.size() + kGroup - 1) / kGroup, (size_t)kSuper);
  std::vector<rtcore::GroupBox> supers_or_always;
  for (size_t g0 = 0; g0 < L.boxes.size(); g0 += kSuper) {
    L.supers.push_back(box_around(L.boxes, g0, g0 + kSuper, L.n_always_groups));
    supers_or_always.push_back(box_around(L.boxes, g0, g0 + kSuper, 0));
  }
  for (size_t q0 = 0; q0 < L.supers.size(); q0 += kQuarterSupers) L.quarters.push_back(box_around(L.supers, q0, q0 + kQuarterSupers, 0));
  if (L.quarters.size() % 2) L.quarters.push_back(rtcore::GroupBox{0.0f, 0.0f, 0.0f, -1.0e30f, -1.0e30f, -1.0e30f});
  const size_t supers_per_batch = batch_spheres / (kSuper * kGroup);
  for (size_t q0 = 0; q0 < L.supers.size(); q0 += supers_per_batch)
    L.batches.push_back(box_around(supers_or_always, q0, q0 + supers_per_batch, 0));
  if (L.batches.size() % 2) L.batches.push_back(rtcore::GroupBox{0.0f, 0.0f, 0.0f, -1.0e30f, -1.0e30f, -1.0e30f});
  for (size_t b0 = 0; b0 < L.batches.size(); b0 += kRegionBatches) L.regions.push_back(box_around(L.batches, b0, b0 + kRegionBatches, 0));
  if (L.regions.size() % 2) L.regions.push_back(rtcore::GroupBox{0.0f, 0.0f, 0.0f, -1.0e30f, -1.0e30f, -1.0e30f});
  L.n_scan = (uint32_t)L.order.size();
  if (h_min < 1e299) {   // (else there is nothing to cull)
    L.sigma0 = round_up_to_float(std::ldexp(1.0, -19) / h_min);
    L.sigma1 = round_up_to_float((b_max + std::ldexp(1.0, -19)) / h_min);
    L.cut_reach = round_up_to_float(reach_max);
    L.cut_b = round_up_to_float(b_max + std::ldexp(1.0, -21));   // (+ 2^-21 |A| <= 2^-21 (|A|^2 + 1))
  }
  return L;
}

}  // namespace rtcluster
