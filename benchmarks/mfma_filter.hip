// mfma_filter.hip -- the conservative discriminant filter of the full scan (rt_core.h: filter_discriminant, eight
// fused multiply-adds per ray-sphere pair) in two forms on the same data:
//
//   valu   the product's form: every lane holds one ray, the spheres come as wave-wide broadcast reads of the pair
//          image in LDS, 32 v_pk_fma_f32 per 8 spheres and wave, one compare per sphere;
//   mfma   b = A.B - C.B, four of the eight multiply-adds, on the matrix pipe: v_mfma_f32_32x32x2_f32 computes
//          32 spheres x 32 rays in K = 2 steps, two instructions make the K = 4 chain {1*ab, cz*(-Bz), cy*(-By),
//          cx*(-Bx)} in exactly the order of the fused chain (the instruction is a k-ordered fmaf chain, one rounding
//          per product), and the vector ALU does the other four multiply-adds and the compare on the accumulator
//          registers: lane l then holds ray (l & 31) of a half of the wave's rays and 16 of the 32 spheres.
//
// Prints pair tests per second of both forms, the number of pairs that pass and a checksum of all discriminant bits
// (both must agree between the forms: the same bits, not close ones).
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -o _out/mfma_filter benchmarks/mfma_filter.hip
//   _out/mfma_filter [spheres=512] [trips=200]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../raytrace_amd/csrc/rt_core.h"

using namespace rtcore;

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(e)                                                                      \
  do {                                                                                \
    hipError_t err_ = (e);                                                            \
    if (err_ != hipSuccess) {                                                         \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(err_));                  \
      std::exit(1);                                                                   \
    }                                                                                 \
  } while (0)

constexpr int kThreads = 256;

// the LDS pair image with one unused 16-byte entry after every 16 (as the product stages it)
__device__ __host__ inline uint32_t pair_slot(uint32_t e) { return e + (e >> 4); }

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float unit(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

// a ray of the kind the renderer traces: origin near the camera or in the sphere field, direction of any length
__device__ __forceinline__ void make_ray(uint32_t id, V3 &A, V3 &B) {
  const uint32_t h = hash32(id * 0x9e3779b9u + 12345u);
  A = v3(24.0f * unit(hash32(h + 1u)) - 12.0f, 3.0f * unit(hash32(h + 2u)), 24.0f * unit(hash32(h + 3u)) - 12.0f);
  B = v3(2.0f * unit(hash32(h + 4u)) - 1.0f, 2.0f * unit(hash32(h + 5u)) - 1.0f, 2.0f * unit(hash32(h + 6u)) - 1.0f);
  const float k = (h & 3u) == 0u ? 15.0f : (h & 3u) == 1u ? 0.05f : 1.0f;
  B = k * B;
}

struct Result {
  unsigned long long passed;     // pairs with a positive filter discriminant
  unsigned long long checksum;   // sum over all pairs of the discriminant's bits
  unsigned long long shader_cycles, realtime_ticks;   // summed over the waves: shader clock = 100 MHz * cycles / ticks
};

#define BCAST_LO "op_sel:[0,0,0] op_sel_hi:[0,1,1]"
#define BCAST_HI "op_sel:[1,0,0] op_sel_hi:[1,1,1]"

// ---- the product's form: 8 spheres (4 pairs = 8 entries of the pair image) for the lane's own ray
struct RayP {
  f32x2 p0, p1, p2, p3, p4;   // {-Bx,-By} {-Bz,A.B} {-2Ax,-2Ay} {-2Az,-(B.B)} {naaa,-}
};
__device__ __forceinline__ void filter_four_pairs(const RayP &f, const float4 (&q)[8], f32x2 (&d)[4]) {
  f32x2 t0, t1, t2, t3;
  const f32x2 cx0 = {q[0].x, q[0].y}, cy0 = {q[0].z, q[0].w}, cz0 = {q[1].x, q[1].y}, kp0 = {q[1].z, q[1].w};
  const f32x2 cx1 = {q[2].x, q[2].y}, cy1 = {q[2].z, q[2].w}, cz1 = {q[3].x, q[3].y}, kp1 = {q[3].z, q[3].w};
  const f32x2 cx2 = {q[4].x, q[4].y}, cy2 = {q[4].z, q[4].w}, cz2 = {q[5].x, q[5].y}, kp2 = {q[5].z, q[5].w};
  const f32x2 cx3 = {q[6].x, q[6].y}, cy3 = {q[6].z, q[6].w}, cz3 = {q[7].x, q[7].y}, kp3 = {q[7].z, q[7].w};
  asm("v_pk_fma_f32 %0, %8, %15, %8 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %1, %8, %19, %8 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %2, %8, %23, %8 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %3, %8, %27, %8 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %4, %10, %15, %16 " BCAST_LO "\n\t"
      "v_pk_fma_f32 %5, %10, %19, %20 " BCAST_LO "\n\t"
      "v_pk_fma_f32 %6, %10, %23, %24 " BCAST_LO "\n\t"
      "v_pk_fma_f32 %7, %10, %27, %28 " BCAST_LO "\n\t"
      "v_pk_fma_f32 %0, %9, %14, %0 " BCAST_HI "\n\t"
      "v_pk_fma_f32 %1, %9, %18, %1 " BCAST_HI "\n\t"
      "v_pk_fma_f32 %2, %9, %22, %2 " BCAST_HI "\n\t"
      "v_pk_fma_f32 %3, %9, %26, %3 " BCAST_HI "\n\t"
      "v_pk_fma_f32 %4, %11, %14, %4 " BCAST_HI "\n\t"
      "v_pk_fma_f32 %5, %11, %18, %5 " BCAST_HI "\n\t"
      "v_pk_fma_f32 %6, %11, %22, %6 " BCAST_HI "\n\t"
      "v_pk_fma_f32 %7, %11, %26, %7 " BCAST_HI "\n\t"
      "v_pk_fma_f32 %0, %9, %13, %0 " BCAST_LO "\n\t"
      "v_pk_fma_f32 %1, %9, %17, %1 " BCAST_LO "\n\t"
      "v_pk_fma_f32 %2, %9, %21, %2 " BCAST_LO "\n\t"
      "v_pk_fma_f32 %3, %9, %25, %3 " BCAST_LO "\n\t"
      "v_pk_fma_f32 %4, %11, %13, %4 " BCAST_LO "\n\t"
      "v_pk_fma_f32 %5, %11, %17, %5 " BCAST_LO "\n\t"
      "v_pk_fma_f32 %6, %11, %21, %6 " BCAST_LO "\n\t"
      "v_pk_fma_f32 %7, %11, %25, %7 " BCAST_LO "\n\t"
      "v_pk_fma_f32 %0, %0, %0, %12 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %1, %1, %1, %12 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %2, %2, %2, %12 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %3, %3, %3, %12 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %0, %10, %4, %0 " BCAST_HI "\n\t"
      "v_pk_fma_f32 %1, %10, %5, %1 " BCAST_HI "\n\t"
      "v_pk_fma_f32 %2, %10, %6, %2 " BCAST_HI "\n\t"
      "v_pk_fma_f32 %3, %10, %7, %3 " BCAST_HI
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(f.p1), "v"(f.p0), "v"(f.p3), "v"(f.p2), "v"(f.p4), "v"(cx0), "v"(cy0), "v"(cz0), "v"(kp0), "v"(cx1), "v"(cy1),
        "v"(cz1), "v"(kp1), "v"(cx2), "v"(cy2), "v"(cz2), "v"(kp2), "v"(cx3), "v"(cy3), "v"(cz3), "v"(kp3));
}

// ---- the matrix form's vector half: for one sphere pair {s, s+1} and one ray, with b = {b_s, b_s+1} from the
// accumulators: c0 = -2Az*cz + K', c0 += -2Ay*cy, c0 += -2Ax*cx, d = b*b + naaa, d += -(B.B)*c0 (the order of
// filter_discriminant).  rp2 = {-2Ax,-2Ay}, rp3 = {-2Az,-(B.B)}, rp4 = {naaa,-}
__device__ __forceinline__ f32x2 finish_pair(f32x2 b, f32x2 rp2, f32x2 rp3, f32x2 rp4, float4 e0, float4 e1) {
  const f32x2 cx = {e0.x, e0.y}, cy = {e0.z, e0.w}, cz = {e1.x, e1.y}, kp = {e1.z, e1.w};
  f32x2 c0, d;
  asm("v_pk_fma_f32 %1, %3, %7, %8 " BCAST_LO "\n\t"                       // c0 = -2Az*cz + K'
      "v_pk_fma_f32 %0, %9, %9, %4 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"   // d  = b*b + naaa
      "v_pk_fma_f32 %1, %2, %6, %1 " BCAST_HI "\n\t"                       // c0 += -2Ay*cy
      "s_nop 0\n\t"
      "v_pk_fma_f32 %1, %2, %5, %1 " BCAST_LO "\n\t"                       // c0 += -2Ax*cx
      "s_nop 0\n\t"
      "v_pk_fma_f32 %0, %3, %1, %0 " BCAST_HI                              // d += -(B.B)*c0
      : "=&v"(d), "=&v"(c0)
      : "v"(rp2), "v"(rp3), "v"(rp4), "v"(cx), "v"(cy), "v"(cz), "v"(kp), "v"(b));
  return d;
}

// kVerify: also sum the bits of every discriminant (the timed launches leave that out)
template <int kMode, bool kVerify>
__global__ __launch_bounds__(kThreads, 4) void filter_bench(const float4 *pairs, uint32_t n_spheres, uint32_t trips, Result *out) {
  extern __shared__ __align__(16) float4 lds_pairs[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  for (uint32_t i = tid; i < n_spheres; i += kThreads) lds_pairs[pair_slot(i)] = pairs[i];
  __syncthreads();
  unsigned long long passed = 0, checksum = 0;   // (passed: wave-uniform, counted from ballots; checksum: per lane)
  const unsigned long long c_begin = __builtin_amdgcn_s_memtime(), r_begin = __builtin_amdgcn_s_memrealtime();
  const uint32_t wave_id = blockIdx.x * (kThreads / 64) + (tid >> 6);
  for (uint32_t trip = 0; trip < trips; ++trip) {
    V3 A, B;
    make_ray((wave_id * trips + trip) * 64u + lane, A, B);
    const FilterRay f = make_filter_ray(A, B);
    if (kMode == 0) {
      const RayP r{f32x2{f.nbx, f.nby}, f32x2{f.nbz, f.ab}, f32x2{f.m2ax, f.m2ay}, f32x2{f.m2az, f.na}, f32x2{f.naaa, 0.0f}};
      for (uint32_t g = 0; g < n_spheres; g += 8u) {
        const float4 *src = lds_pairs + pair_slot(g);
        const float4 q[8] = {src[0], src[1], src[2], src[3], src[4], src[5], src[6], src[7]};
        f32x2 d[4];
        filter_four_pairs(r, q, d);
        // as in the product: eight compares into eight scalar masks, OR-ed, one branch for the chunk
        unsigned long long m[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) m[u] = __ballot(d[u >> 1][u & 1] > 0.0f);
        if ((((m[0] | m[1]) | (m[2] | m[3])) | ((m[4] | m[5]) | (m[6] | m[7]))) != 0ull) {
#pragma unroll
          for (int u = 0; u < 8; ++u) passed += (unsigned long long)__popcll(m[u]);
        }
        if (kVerify) {
#pragma unroll
          for (int u = 0; u < 8; ++u) checksum += __float_as_uint(d[u >> 1][u & 1]);
        }
      }
    } else {
      // the other half's ray: lane l and lane l ^ 32 exchange their coefficients
      auto other = [&](float v) { return __shfl_xor(v, 32, 64); };
      const bool upper = lane >= 32u;
      const float o_nbx = other(f.nbx), o_nby = other(f.nby), o_nbz = other(f.nbz), o_ab = other(f.ab);
      // B operands (lane l: B[k = l >> 5][ray l & 31]) of the two instructions of the chain, for the rays of the lower
      // half of the wave (set 0) and of the upper half (set 1): step 1 {k0: A.B, k1: -Bz}, step 2 {k0: -By, k1: -Bx}
      const float b1_0 = upper ? o_nbz : f.ab, b1_1 = upper ? f.nbz : o_ab;
      const float b2_0 = upper ? o_nbx : f.nby, b2_1 = upper ? f.nbx : o_nby;
      // the vector half's coefficients of the ray this lane finishes in set 0 and in set 1
      const f32x2 own2 = {f.m2ax, f.m2ay}, own3 = {f.m2az, f.na}, own4 = {f.naaa, 0.0f};
      const f32x2 oth2 = {other(f.m2ax), other(f.m2ay)}, oth3 = {other(f.m2az), other(f.na)}, oth4 = {other(f.naaa), 0.0f};
      const f32x2 s0p2 = upper ? oth2 : own2, s0p3 = upper ? oth3 : own3, s0p4 = upper ? oth4 : own4;
      const f32x2 s1p2 = upper ? own2 : oth2, s1p3 = upper ? own3 : oth3, s1p4 = upper ? own4 : oth4;
      const uint32_t j = lane & 31u;
      if (kMode == 2) {
        // the same with the matrix instructions of the next block issued before this block's accumulators are
        // finished on the vector ALU (two accumulator sets)
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto products = [&](uint32_t g, f32x16 &acc0, f32x16 &acc1) {
          const float *flat = reinterpret_cast<const float *>(lds_pairs + pair_slot(g + (j & ~1u)));
          const float a1 = upper ? flat[4 + (j & 1u)] : 1.0f;
          const float a2 = upper ? flat[0 + (j & 1u)] : flat[2 + (j & 1u)];
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1_0, zero, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1_1, zero, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2_0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2_1, acc1, 0, 0, 0);
        };
        auto finish = [&](uint32_t g, const f32x16 &acc0, const f32x16 &acc1) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const uint32_t first = (uint32_t)((2 * q & 3) + 8 * (2 * q >> 2)) + (upper ? 4u : 0u);
            const float4 *e = lds_pairs + pair_slot(g + first);
            const float4 e0 = e[0], e1 = e[1];
            const f32x2 d0 = finish_pair(f32x2{acc0[2 * q], acc0[2 * q + 1]}, s0p2, s0p3, s0p4, e0, e1);
            const f32x2 d1 = finish_pair(f32x2{acc1[2 * q], acc1[2 * q + 1]}, s1p2, s1p3, s1p4, e0, e1);
            const unsigned long long m0 = __ballot(d0[0] > 0.0f), m1 = __ballot(d0[1] > 0.0f), m2 = __ballot(d1[0] > 0.0f), m3 = __ballot(d1[1] > 0.0f);
            if (((m0 | m1) | (m2 | m3)) != 0ull) passed += (unsigned long long)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
            if (kVerify) checksum += (unsigned long long)__float_as_uint(d0[0]) + __float_as_uint(d0[1]) + __float_as_uint(d1[0]) + __float_as_uint(d1[1]);
          }
        };
        f32x16 a0, a1s, b0, b1s;
        products(0u, a0, a1s);
        for (uint32_t g = 0; g < n_spheres; g += 64u) {     // (n_spheres is a multiple of 64 in this mode)
          products(g + 32u, b0, b1s);
          finish(g, a0, a1s);
          if (g + 64u < n_spheres) products(g + 64u, a0, a1s);
          finish(g + 32u, b0, b1s);
        }
        continue;
      }
      for (uint32_t g = 0; g < n_spheres; g += 32u) {
        // A operands (lane l: A[sphere l & 31][k = l >> 5]): step 1 {k0: 1, k1: cz}, step 2 {k0: cy, k1: cx}, read from
        // the pair image: sphere s is half (s & 1) of entries 2 (s >> 1) (cx, cy) and 2 (s >> 1) + 1 (cz, K')
        const float *flat = reinterpret_cast<const float *>(lds_pairs + pair_slot(g + (j & ~1u)));
        const float a1 = upper ? flat[4 + (j & 1u)] : 1.0f;                 // cz | 1
        const float a2 = upper ? flat[0 + (j & 1u)] : flat[2 + (j & 1u)];   // cx | cy
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1_0, zero, 0, 0, 0);
        f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1_1, zero, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2_0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2_1, acc1, 0, 0, 0);
        // accumulator register r of lane l: ray l & 31, sphere (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the block
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const uint32_t first = (uint32_t)((2 * q & 3) + 8 * (2 * q >> 2)) + (upper ? 4u : 0u);   // even: a pair of the image
          const float4 *e = lds_pairs + pair_slot(g + first);
          const float4 e0 = e[0], e1 = e[1];
          const f32x2 d0 = finish_pair(f32x2{acc0[2 * q], acc0[2 * q + 1]}, s0p2, s0p3, s0p4, e0, e1);
          const f32x2 d1 = finish_pair(f32x2{acc1[2 * q], acc1[2 * q + 1]}, s1p2, s1p3, s1p4, e0, e1);
          const unsigned long long m0 = __ballot(d0[0] > 0.0f), m1 = __ballot(d0[1] > 0.0f), m2 = __ballot(d1[0] > 0.0f), m3 = __ballot(d1[1] > 0.0f);
          if (((m0 | m1) | (m2 | m3)) != 0ull) passed += (unsigned long long)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
          if (kVerify) checksum += (unsigned long long)__float_as_uint(d0[0]) + __float_as_uint(d0[1]) + __float_as_uint(d1[0]) + __float_as_uint(d1[1]);
        }
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) checksum += __shfl_down(checksum, off, 64);
  if (lane == 0) {
    atomicAdd(&out->passed, passed);
    atomicAdd(&out->checksum, checksum);
    atomicAdd(&out->shader_cycles, __builtin_amdgcn_s_memtime() - c_begin);
    atomicAdd(&out->realtime_ticks, __builtin_amdgcn_s_memrealtime() - r_begin);
  }
}

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 512u;
  const uint32_t trips = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 200u;
  if (n == 0 || n % 64u) {
    std::fprintf(stderr, "spheres must be a positive multiple of 64\n");
    return 1;
  }
  // spheres like the reference's final scene: small ones on a plane, a few large, as the pair image of the product
  std::vector<FilterSphere> fs(n);
  uint32_t state = 1u;
  auto rnd = [&]() {
    state = state * 1664525u + 1013904223u;
    return (float)(state >> 8) * (1.0f / 16777216.0f);
  };
  for (uint32_t i = 0; i < n; ++i)
    fs[i] = i < 4 ? make_filter_sphere(8.0f * rnd() - 4.0f, 1.0f, 2.0f * rnd(), 1.0f)
                  : make_filter_sphere(22.0f * rnd() - 11.0f, 0.2f, 22.0f * rnd() - 11.0f, 0.2f);
  std::vector<float4> pairs(n);
  for (uint32_t p = 0; p < n; p += 2) {
    pairs[p] = make_float4(fs[p].cx, fs[p + 1].cx, fs[p].cy, fs[p + 1].cy);
    pairs[p + 1] = make_float4(fs[p].cz, fs[p + 1].cz, fs[p].kprime, fs[p + 1].kprime);
  }
  float4 *d_pairs;
  Result *d_out;
  CHECK(hipMalloc(&d_pairs, n * sizeof(float4)));
  CHECK(hipMalloc(&d_out, 3 * sizeof(Result)));
  CHECK(hipMemcpy(d_pairs, pairs.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const uint32_t blocks = (uint32_t)prop.multiProcessorCount * 4u;
  const size_t lds = (size_t)pair_slot(n) * sizeof(float4) + 16;
  const double pair_tests = (double)blocks * (kThreads / 64) * trips * 64.0 * n;
  Result res[3];
  const char *names[3] = {"valu", "mfma", "mfma, two accumulator sets"};
  for (int mode = 0; mode < 3; ++mode) {
    double best = 1e30;
    Result timed{};
    for (int rep = 0; rep < 5; ++rep) {
      const bool verify = rep == 4;     // the last launch is the one whose checksum is compared; it is not timed
      CHECK(hipMemset(d_out + mode, 0, sizeof(Result)));
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      CHECK(hipEventRecord(e0, nullptr));
#define LAUNCH(M, V) hipLaunchKernelGGL((filter_bench<M, V>), dim3(blocks), dim3(kThreads), lds, nullptr, d_pairs, n, trips, d_out + mode)
      if (mode == 0 && !verify) LAUNCH(0, false);
      if (mode == 0 && verify) LAUNCH(0, true);
      if (mode == 1 && !verify) LAUNCH(1, false);
      if (mode == 1 && verify) LAUNCH(1, true);
      if (mode == 2 && !verify) LAUNCH(2, false);
      if (mode == 2 && verify) LAUNCH(2, true);
#undef LAUNCH
      CHECK(hipGetLastError());
      CHECK(hipEventRecord(e1, nullptr));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (!verify && ms < best) {
        best = ms;
        CHECK(hipMemcpy(&timed, d_out + mode, sizeof(Result), hipMemcpyDeviceToHost));
      }
      CHECK(hipEventDestroy(e0));
      CHECK(hipEventDestroy(e1));
    }
    CHECK(hipMemcpy(&res[mode], d_out + mode, sizeof(Result), hipMemcpyDeviceToHost));
    std::printf("{\"form\": \"%s\", \"spheres\": %u, \"trips\": %u, \"workgroups\": %u, \"ms\": %.4f, \"pair_tests_per_s\": %.4g, "
                "\"tflops_at_17_flop_per_test\": %.2f, \"shader_clock_ghz\": %.3f, \"passed\": %llu, \"checksum\": %llu}\n",
                names[mode], n, trips, blocks, best, pair_tests / (best * 1e-3), 17.0 * pair_tests / (best * 1e-3) / 1e12,
                0.1 * (double)timed.shader_cycles / (double)timed.realtime_ticks, res[mode].passed, res[mode].checksum);
  }
  const bool same = res[0].passed == res[1].passed && res[0].checksum == res[1].checksum && res[0].passed == res[2].passed &&
                    res[0].checksum == res[2].checksum;
  std::printf("{\"bits_agree\": %s}\n", same ? "true" : "false");
  return same ? 0 : 2;
}
