// rt_cluster.h -- host side, at scene upload: the order in which the kernels
// scan the sphere list and the bounding spheres that let a wave skip whole
// groups of it.
//
// The reference scans the list in index order (c/main.c:248-254).  Its result
// does not depend on that order except for ties: the hit is the smallest
// "first root inside (tmin, FLT_MAX)" over all spheres, and among equal t the
// lowest index wins (strict `<`, c/main.c:225, 249-251).  The kernels
// therefore scan in an order of their own -- large spheres first, the rest
// sorted along a Morton curve and cut into groups of kGroup neighbours -- and
// break ties by the original index.
//
// Each group has a bounding sphere tested with the same conservative filter as
// the spheres themselves (rt_core.h).  Its radius is chosen so that, for every
// ray origin with |A| <= kCullOriginLimit, a ray that passes the filter of any
// member also passes the filter of the bound; rays that start farther out skip
// the test and scan every group.  Derivation (u = 2^-24, kappa = 128u):
//   member passes  =>  d_i^2 < r_i^2 + 1.375 kappa (|A|^2 + |C_i|^2 + r_i^2)
//                      (its margin kappa plus its rounding error < 0.375 kappa)
//   bound passes  <=   d_b^2 <= R^2 + 0.625 kappa (|A|^2 + |C_b|^2 + R^2)
//                      (its margin kappa minus its rounding error)
//   d_b <= d_i + |C_i - C_b|
// so R >= max_i (rho_i + e_i) with rho_i taken at |A| = kCullOriginLimit (the
// worst case) is enough (the bound's own margin is simply not relied on).
//
// Scenes like the reference's are flat: hundreds of small spheres resting on
// one plane.  A bounding SPHERE of eight neighbours then reaches several
// sphere diameters above the plane and most rays that merely fly over a group
// "touch" it.  The bounds are therefore spheres in an axis-scaled space,
// x' = (sx x, sy y, sz z) with power-of-two factors (exact in float), i.e.
// ellipsoids flattened along the axis in which the groups are thin.  A line
// that comes within rho of C_i has a point whose offset from C_i is at most
// rho in every coordinate, i.e. at most s_k rho along axis k after scaling, so
// R'^2 >= max_i sum_k (|C_ik' - C_bk'| + s_k rho_i)^2 is enough.
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
constexpr double kCullOriginLimit = 32.0;      // |A| above this: no culling for that ray
constexpr double kLargeFactor = 3.0;           // radius > this * median radius: scanned by every ray

struct Layout {
  std::vector<uint32_t> order;        // scan position -> original index, 0xffffffff for a pad
  std::vector<rtcore::FilterSphere> bounds;   // one per group; K' = -1e30 means "always scan"
  std::vector<uint8_t> real_in_group; // real (non-pad) spheres per group
  uint32_t n_scan = 0;                // = order.size(), a multiple of 2*kGroup
  float scale[3] = {1.0f, 1.0f, 1.0f}; // axis factors of the space the bounds live in (powers of two)
};

inline uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

inline Layout build_layout(const rt_sphere *s, size_t n) {
  Layout L;
  std::vector<float> radii(n);
  for (size_t i = 0; i < n; ++i) radii[i] = s[i].radius;
  std::vector<float> sorted = radii;
  std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
  const double median = sorted[n / 2];
  std::vector<uint32_t> large, regular;
  for (size_t i = 0; i < n; ++i) ((double)radii[i] > kLargeFactor * median ? large : regular).push_back((uint32_t)i);

  // Morton order of the regular spheres' centres
  if (!regular.empty()) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t i : regular)
      for (int k = 0; k < 3; ++k) {
        lo[k] = std::min(lo[k], (double)s[i].center[k]);
        hi[k] = std::max(hi[k], (double)s[i].center[k]);
      }
    double extent = 1e-30;
    for (int k = 0; k < 3; ++k) extent = std::max(extent, hi[k] - lo[k]);
    std::vector<uint32_t> code(n, 0);
    for (uint32_t i : regular) {
      uint32_t q[3];
      for (int k = 0; k < 3; ++k) q[k] = (uint32_t)std::min(1023.0, ((double)s[i].center[k] - lo[k]) / extent * 1023.0);
      code[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    }
    std::stable_sort(regular.begin(), regular.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });
  }

  // axis factors: make the typical group as round as possible in the scaled space
  {
    double ext[3] = {0, 0, 0};
    size_t groups = 0;
    for (size_t k = 0; k + 1 <= regular.size(); k += kGroup) {
      size_t cnt = std::min<size_t>(kGroup, regular.size() - k);
      double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
      for (size_t j = 0; j < cnt; ++j) {
        const rt_sphere &m = s[regular[k + j]];
        for (int d = 0; d < 3; ++d) {
          lo[d] = std::min(lo[d], (double)m.center[d] - m.radius);
          hi[d] = std::max(hi[d], (double)m.center[d] + m.radius);
        }
      }
      for (int d = 0; d < 3; ++d) ext[d] += hi[d] - lo[d];
      groups++;
    }
    if (groups) {
      double emax = std::max(ext[0], std::max(ext[1], ext[2]));
      for (int d = 0; d < 3; ++d) {
        double want = ext[d] > 0 ? emax / ext[d] : 16.0;
        float f = 1.0f;
        while (f < 16.0f && (double)f * 1.5 < want) f *= 2.0f;
        L.scale[d] = f;
      }
    }
  }

  auto emit_group = [&](const uint32_t *idx, size_t cnt, bool always) {
    for (size_t k = 0; k < (size_t)kGroup; ++k) L.order.push_back(k < cnt ? idx[k] : 0xffffffffu);
    L.real_in_group.push_back((uint8_t)cnt);
    if (always || cnt == 0) {
      // pad groups never pass (K' = +1e30); "always" groups pass for every live ray
      L.bounds.push_back(rtcore::FilterSphere{0.0f, 0.0f, 0.0f, always ? -1.0e30f : 1.0e30f});
      return;
    }
    double c[3] = {0, 0, 0};
    for (size_t k = 0; k < cnt; ++k)
      for (int d = 0; d < 3; ++d) c[d] += (double)s[idx[k]].center[d] * L.scale[d];
    for (int d = 0; d < 3; ++d) c[d] /= (double)cnt;
    const float cf[3] = {(float)c[0], (float)c[1], (float)c[2]};   // the (scaled) centre the kernel will see
    const double kap = rtcore::kFilterKappa;
    const double A2 = kCullOriginLimit * kCullOriginLimit;
    double R = 0.0;
    for (size_t k = 0; k < cnt; ++k) {
      const rt_sphere &m = s[idx[k]];
      double ci2 = 0;
      for (int d = 0; d < 3; ++d) ci2 += (double)m.center[d] * m.center[d];
      const double r2 = (double)m.radius * m.radius;
      const double rho = std::sqrt(r2 + 1.375 * kap * (A2 + ci2 + r2));
      double far2 = 0;
      for (int d = 0; d < 3; ++d) {
        double reach = std::fabs((double)m.center[d] * L.scale[d] - (double)cf[d]) + (double)L.scale[d] * rho;
        far2 += reach * reach;
      }
      R = std::max(R, std::sqrt(far2));
    }
    const float Rf = (float)(R * (1.0 + 1e-6)) + 1e-6f;
    L.bounds.push_back(rtcore::make_filter_sphere(cf[0], cf[1], cf[2], Rf));
  };

  for (size_t k = 0; k < large.size(); k += kGroup) emit_group(&large[k], std::min<size_t>(kGroup, large.size() - k), true);
  for (size_t k = 0; k < regular.size(); k += kGroup)
    emit_group(&regular[k], std::min<size_t>(kGroup, regular.size() - k), false);
  if (L.bounds.size() % 2) emit_group(nullptr, 0, false);   // rounds are two groups
  L.n_scan = (uint32_t)L.order.size();
  return L;
}

}  // namespace rtcluster
