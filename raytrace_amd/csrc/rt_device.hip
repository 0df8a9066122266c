// rt_device.hip -- HIP kernels (gfx950) and the C ABI of include/rt_amd.h.
//
// Kernel layout.  One lane owns one pixel of an 8x8 pixel block per wave
// (four waves = one 16x16 block per workgroup) and walks that pixel's samples
// in order, so the per-pixel float sum has the reference's order
// (c/main.c:462-469).  A lane whose path ends starts its next sample at once
// ("persistent lane"); a wave leaves the bounce loop when a ballot shows no
// lane has work left.
//
// Each bounce is (1) a scan of the whole sphere list, staged as 16-byte
// {cx,cy,cz,K'} entries in LDS and read as wave-wide broadcasts, with the
// eight-FMA conservative filter of rt_core.h; spheres that pass are appended
// to a per-lane list in LDS; (2) a drain that runs the reference's exact
// arithmetic (c/main.c:214-236) on the listed spheres in index order, which
// gives exactly the hit the reference's linear scan gives because a sphere
// the filter rejects can never satisfy D > 0; (3) shading (c/main.c:163-211,
// 258-283).
//
// Compiled with -ffp-contract=off (see rt_core.h).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_core.h"

using namespace rtcore;

namespace {

// ----------------------------------------------------------------- limits
constexpr int kBlockThreads = 256;
constexpr int kWave = 64;
constexpr int kListCap = 32;          // pending candidates per lane
constexpr int kChunk = 8;             // spheres per unrolled filter group
constexpr int kDrainEvery = 16;       // spheres between overflow checks (<= kListCap/2)
constexpr uint32_t kMaxLdsSpheres = 2048;  // 32 KiB of filter entries per staged batch
constexpr uint32_t kMaxSpheres = 65535;    // list entries are 16-bit

struct DeviceScene {
  const float4 *filter;   // n_pad entries {cx, cy, cz, K'}; pads never pass
  const float4 *exact;    // n entries {cx, cy, cz, r*r}
  const float4 *mat;      // n entries {albedo rgb, param}
  const float2 *rad_type; // n entries {radius, type as float bits}
  uint32_t n, n_pad;
};

struct KernelArgs {
  Camera cam;
  rt_job job;
  uint32_t rows;          // rt_job_rows(job)
  DeviceScene sc;
  uint8_t *rgb8;
  float *linear;
  int32_t *first_hit;
  uint32_t *hit_sum;
  unsigned long long *counters;  // paths, segments, tests, candidates
};

__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// One closest-hit query for every lane of the workgroup, against filter
// entries [0, count) in LDS that stand for global sphere indices base+i.
// Lanes without a ray pass a FilterRay that can never pass (see dead_ray()).
struct HitState {
  float closest;
  int index;
  uint32_t cand;
};

__device__ __forceinline__ FilterRay dead_ray() {
  FilterRay f;
  f.nbx = f.nby = f.nbz = 0.0f;
  f.ab = 0.0f;
  f.m2ax = f.m2ay = f.m2az = 0.0f;
  f.na = 0.0f;
  f.naaa = -1.0f;
  return f;
}

__device__ __forceinline__ void drain_list(const DeviceScene &sc, const uint16_t *list, int &cnt, V3 A, V3 B, float a,
                                           float tmin, HitState &h) {
  for (int j = 0; wave_any(j < cnt); ++j) {
    if (j < cnt) {
      uint32_t idx = list[j * kBlockThreads];
      float4 e = sc.exact[idx];
      float t;
      if (exact_hit(A, B, a, e.x, e.y, e.z, e.w, tmin, h.closest, t)) {
        h.closest = t;
        h.index = (int)idx;
      }
    }
  }
  h.cand += (uint32_t)cnt;
  cnt = 0;
}

// scan LDS entries [0,count): count is a multiple of kDrainEvery
__device__ __forceinline__ void scan_batch(const DeviceScene &sc, const float4 *lds_filter, uint32_t base,
                                           uint32_t count, const FilterRay &fr, uint16_t *list, int &cnt, V3 A, V3 B,
                                           float a, float tmin, HitState &h) {
  for (uint32_t i0 = 0; i0 < count; i0 += kDrainEvery) {
#pragma unroll
    for (int g = 0; g < kDrainEvery; g += kChunk) {
      float d[kChunk];
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        float4 s = lds_filter[i0 + g + u];
        d[u] = filter_discriminant(fr, s.x, s.y, s.z, s.w);
      }
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        if (d[u] > 0.0f) {
          list[cnt * kBlockThreads] = (uint16_t)(base + i0 + g + u);
          ++cnt;
        }
      }
    }
    if (wave_any(cnt > kListCap - kDrainEvery)) drain_list(sc, list, cnt, A, B, a, tmin, h);
  }
}

// global row of compact (tile-major) local row lr
__device__ __forceinline__ uint32_t global_row(const rt_job &job, uint32_t lr) {
  uint32_t t = lr / job.tile_rows;
  return job.row0 + t * job.tile_stride + (lr - t * job.tile_rows);
}

template <bool kBatched>
__global__ __launch_bounds__(kBlockThreads) void mega_kernel(const KernelArgs args) {
  extern __shared__ __align__(16) unsigned char smem[];
  const DeviceScene &sc = args.sc;
  const rt_job &job = args.job;
  const uint32_t lds_entries = kBatched ? kMaxLdsSpheres : sc.n_pad;
  float4 *lds_filter = reinterpret_cast<float4 *>(smem);
  uint16_t *lds_list = reinterpret_cast<uint16_t *>(smem + (size_t)lds_entries * sizeof(float4));

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  // 8x8 pixels per wave, waves 2x2 in the workgroup
  const uint32_t col = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
  const uint32_t lrow = blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
  const bool valid = col < job.nx && lrow < args.rows;
  const uint32_t grow = valid ? global_row(job, lrow) : 0u;
  const uint32_t pixel = grow * job.nx + col;      // key of the counter streams
  const uint32_t up = job.ny - 1u - grow;          // rows below: j-1 of c/main.c:466
  uint16_t *my_list = lds_list + tid;

  if (!kBatched) {
    for (uint32_t i = tid; i < sc.n_pad; i += kBlockThreads) lds_filter[i] = sc.filter[i];
    __syncthreads();
  }

  bool alive = valid;
  uint32_t sample = 0, depth = 0;
  V3 A = v3(0, 0, 0), B = v3(0, 0, 1), color = v3(1, 1, 1), sum = v3(0, 0, 0);
  Rng g{0u, 0u};
  int32_t fh0 = -1;
  uint32_t hsum = 0;
  unsigned long long n_seg = 0, n_cand = 0, n_paths = 0;

  auto start_sample = [&]() {
    g.key = path_key(job.seed, pixel, sample);
    g.k = 0;
    float x, y;
    pixel_xy(col, up, job.nx, job.ny, g, x, y);
    camera_ray(args.cam, x, y, g, A, B);
    color = v3(1.0f, 1.0f, 1.0f);
    depth = 0;
  };
  if (alive) start_sample();

  for (;;) {
    if (kBatched) {
      if (!__syncthreads_or(alive ? 1 : 0)) break;
    } else {
      if (!wave_any(alive)) break;
    }
    // ---- (1)+(2): closest hit
    FilterRay fr = alive ? make_filter_ray(A, B) : dead_ray();
    const float a = dot(B, B);
    HitState h{kFltMax, -1, 0u};
    int cnt = 0;
    if (kBatched) {
      for (uint32_t base = 0; base < sc.n_pad; base += kMaxLdsSpheres) {
        uint32_t count = min(kMaxLdsSpheres, sc.n_pad - base);
        __syncthreads();
        for (uint32_t i = tid; i < count; i += kBlockThreads) lds_filter[i] = sc.filter[base + i];
        __syncthreads();
        scan_batch(sc, lds_filter, base, count, fr, my_list, cnt, A, B, a, job.tmin, h);
      }
    } else {
      scan_batch(sc, lds_filter, 0u, sc.n_pad, fr, my_list, cnt, A, B, a, job.tmin, h);
    }
    drain_list(sc, my_list, cnt, A, B, a, job.tmin, h);

    // ---- (3): shade
    if (alive) {
      n_seg++;
      n_cand += h.cand;
      bool done = false;
      if (depth == 0) {
        hsum += (uint32_t)(h.index + 1);
        if (sample == 0) fh0 = h.index;
      }
      if (h.index >= 0) {
        float4 e = sc.exact[h.index];
        float4 m4 = sc.mat[h.index];
        float2 rt = sc.rad_type[h.index];
        V3 ctr = v3(e.x, e.y, e.z);
        V3 p = A + h.closest * B;              // c/structs.c:6-8
        V3 n = (p - ctr) / rt.x;               // c/main.c:229
        Material m{__float_as_int(rt.y), v3(m4.x, m4.y, m4.z), m4.w};
        V3 atten, nA, nB;
        if (scatter(m, A, B, p, n, g, atten, nA, nB)) {
          A = nA;
          B = nB;
          color = atten * color;               // c/main.c:269
          depth++;
          done = depth >= job.max_depth;       // c/main.c:262, 282
        } else {
          // c/main.c:270-272: the colour becomes 0 and stays 0 whatever the
          // rest of the reference's loop does with the ray, so the path ends
          color = v3(0.0f, 0.0f, 0.0f);
          done = true;
        }
      } else {
        color = color * sky(B);                // c/main.c:274-279
        done = true;
      }
      if (done) {
        sum = sum + color;                     // c/main.c:468
        n_paths++;
        sample++;
        if (sample < job.nsamples) {
          start_sample();
        } else {
          alive = false;
          V3 mean;
          uint8_t rgb[3];
          finish_pixel(sum, job.nsamples, mean, rgb);
          size_t o = (size_t)lrow * job.nx + col;
          args.rgb8[3 * o + 0] = rgb[0];
          args.rgb8[3 * o + 1] = rgb[1];
          args.rgb8[3 * o + 2] = rgb[2];
          if (args.linear) {
            args.linear[3 * o + 0] = mean.x;
            args.linear[3 * o + 1] = mean.y;
            args.linear[3 * o + 2] = mean.z;
          }
          if (args.first_hit) args.first_hit[o] = fh0;
          if (args.hit_sum) args.hit_sum[o] = hsum;
        }
      }
    }
  }

  // event counts: one atomic per wave and counter
  unsigned long long w_paths = wave_sum(n_paths), w_seg = wave_sum(n_seg), w_cand = wave_sum(n_cand);
  if (lane == 0 && args.counters) {
    atomicAdd(&args.counters[0], w_paths);
    atomicAdd(&args.counters[1], w_seg);
    atomicAdd(&args.counters[2], w_seg * (unsigned long long)sc.n);
    atomicAdd(&args.counters[3], w_cand);
  }
}

// ---- unit-level kernels

__global__ __launch_bounds__(kBlockThreads) void closest_hit_kernel(DeviceScene sc, const float *rays, uint32_t n_rays,
                                                                     float tmin, float tmax, int32_t *index, float *t) {
  extern __shared__ __align__(16) unsigned char smem[];
  float4 *lds_filter = reinterpret_cast<float4 *>(smem);
  uint16_t *lds_list = reinterpret_cast<uint16_t *>(smem + (size_t)kMaxLdsSpheres * sizeof(float4));
  const uint32_t tid = threadIdx.x;
  const uint32_t r = blockIdx.x * kBlockThreads + tid;
  const bool alive = r < n_rays;
  V3 A = v3(0, 0, 0), B = v3(0, 0, 1);
  if (alive) {
    A = v3(rays[6 * r + 0], rays[6 * r + 1], rays[6 * r + 2]);
    B = v3(rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]);
  }
  FilterRay fr = alive ? make_filter_ray(A, B) : dead_ray();
  const float a = dot(B, B);
  HitState h{tmax, -1, 0u};
  int cnt = 0;
  uint16_t *my_list = lds_list + tid;
  for (uint32_t base = 0; base < sc.n_pad; base += kMaxLdsSpheres) {
    uint32_t count = min(kMaxLdsSpheres, sc.n_pad - base);
    __syncthreads();
    for (uint32_t i = tid; i < count; i += kBlockThreads) lds_filter[i] = sc.filter[base + i];
    __syncthreads();
    scan_batch(sc, lds_filter, base, count, fr, my_list, cnt, A, B, a, tmin, h);
  }
  drain_list(sc, my_list, cnt, A, B, a, tmin, h);
  if (alive) {
    index[r] = h.index;
    t[r] = h.index >= 0 ? h.closest : 0.0f;
  }
}

__global__ void uniforms_kernel(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t n, double *out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Rng g{path_key(seed, pixel, sample), i};
  out[i] = next_uniform(g);
}

// ------------------------------------------------------------------ host side

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

#define RT_HIP(expr)                                                                                    \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                       \
  } while (0)

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  int reserve(size_t n) {
    if (n <= cap) return RT_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T));
    if (e != hipSuccess) return fail(RT_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    cap = n;
    return RT_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

struct rt_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  bool timed = false;
  DevBuf<float4> filter, exact, mat;
  DevBuf<float2> rad_type;
  uint32_t n = 0, n_pad = 0;
  DevBuf<unsigned long long> counters;
  DevBuf<uint8_t> out_rgb8;
  DevBuf<float> out_linear;
  DevBuf<int32_t> out_first_hit;
  DevBuf<uint32_t> out_hit_sum;
};

namespace {

int check_job(const rt_job *job) {
  if (!job) return fail(RT_ERR_ARG, "job is null");
  if (job->nx == 0 || job->ny == 0) return fail(RT_ERR_ARG, "image size is zero");
  if (job->tile_rows == 0 || job->n_tiles == 0) return fail(RT_ERR_ARG, "no rows in job");
  if (job->n_tiles > 1 && job->tile_stride < job->tile_rows) return fail(RT_ERR_ARG, "tiles overlap");
  if (job->row0 >= job->ny) return fail(RT_ERR_ARG, "row0 outside the image");
  if (job->nsamples == 0) return fail(RT_ERR_ARG, "nsamples is zero");
  if (job->max_depth == 0) return fail(RT_ERR_ARG, "max_depth is zero");
  if ((uint64_t)job->nx * job->ny > 0xffffffffull) return fail(RT_ERR_LIMIT, "more than 2^32 pixels");
  if (job->kernel != RT_KERNEL_MEGA && job->kernel != RT_KERNEL_WAVEFRONT)
    return fail(RT_ERR_ARG, "unknown kernel variant");
  return RT_OK;
}

DeviceScene scene_view(const rt_ctx *c) {
  return DeviceScene{c->filter.p, c->exact.p, c->mat.p, c->rad_type.p, c->n, c->n_pad};
}

size_t mega_lds_bytes(uint32_t n_pad) {
  uint32_t entries = n_pad <= kMaxLdsSpheres ? n_pad : kMaxLdsSpheres;
  return (size_t)entries * sizeof(float4) + (size_t)kListCap * kBlockThreads * sizeof(uint16_t);
}

int launch_mega(rt_ctx *c, const rt_camera *cam, const rt_job *job, uint8_t *rgb8, float *linear, int32_t *first_hit,
                uint32_t *hit_sum) {
  KernelArgs a;
  static_assert(sizeof(Camera) == sizeof(rt_camera), "camera layouts differ");
  std::memcpy(&a.cam, cam, sizeof a.cam);
  a.job = *job;
  a.rows = rt_job_rows(job);
  a.sc = scene_view(c);
  a.rgb8 = rgb8;
  a.linear = linear;
  a.first_hit = first_hit;
  a.hit_sum = hit_sum;
  a.counters = c->counters.p;
  dim3 grid((job->nx + 15u) / 16u, (a.rows + 15u) / 16u);
  size_t lds = mega_lds_bytes(c->n_pad);
  RT_HIP(hipEventRecord(c->ev_begin, c->stream));
  if (c->n_pad <= kMaxLdsSpheres) {
    hipLaunchKernelGGL(mega_kernel<false>, grid, dim3(kBlockThreads), lds, c->stream, a);
  } else {
    hipLaunchKernelGGL(mega_kernel<true>, grid, dim3(kBlockThreads), lds, c->stream, a);
  }
  RT_HIP(hipGetLastError());
  RT_HIP(hipEventRecord(c->ev_end, c->stream));
  c->timed = true;
  return RT_OK;
}

}  // namespace

extern "C" {

uint32_t rt_job_rows(const rt_job *job) {
  if (!job || job->tile_rows == 0) return 0;
  uint64_t rows = 0;
  for (uint32_t t = 0; t < job->n_tiles; ++t) {
    uint64_t first = (uint64_t)job->row0 + (uint64_t)t * job->tile_stride;
    if (first >= job->ny) break;
    uint64_t last = first + job->tile_rows;
    if (last > job->ny) last = job->ny;
    rows += last - first;
  }
  return (uint32_t)rows;
}

const char *rt_last_error(void) { return g_last_error.c_str(); }
int rt_abi_version(void) { return RT_AMD_ABI_VERSION; }

int rt_device_count(int *count) {
  if (!count) return fail(RT_ERR_ARG, "count is null");
  RT_HIP(hipGetDeviceCount(count));
  return RT_OK;
}

int rt_ctx_create(int device, rt_ctx **out) {
  if (!out) return fail(RT_ERR_ARG, "out is null");
  *out = nullptr;
  int n = 0;
  RT_HIP(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(RT_ERR_ARG, "no such device");
  RT_HIP(hipSetDevice(device));
  rt_ctx *c = new (std::nothrow) rt_ctx();
  if (!c) return fail(RT_ERR_NOMEM, "out of host memory");
  c->device = device;
  hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_begin);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_end);
  if (e != hipSuccess) {
    rt_ctx_destroy(c);
    return fail(RT_ERR_HIP, std::string("context set-up: ") + hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  int rc = c->counters.reserve(4);
  if (rc != RT_OK) {
    rt_ctx_destroy(c);
    return rc;
  }
  e = hipMemsetAsync(c->counters.p, 0, 4 * sizeof(unsigned long long), c->stream);
  if (e != hipSuccess) {
    rt_ctx_destroy(c);
    return fail(RT_ERR_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
  }
  *out = c;
  return RT_OK;
}

int rt_ctx_destroy(rt_ctx *c) {
  if (!c) return RT_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->filter.release();
  c->exact.release();
  c->mat.release();
  c->rad_type.release();
  c->counters.release();
  c->out_rgb8.release();
  c->out_linear.release();
  c->out_first_hit.release();
  c->out_hit_sum.release();
  if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
  if (c->ev_end) (void)hipEventDestroy(c->ev_end);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return RT_OK;
}

int rt_ctx_set_stream(rt_ctx *c, void *hip_stream) {
  if (!c) return fail(RT_ERR_ARG, "ctx is null");
  RT_HIP(hipSetDevice(c->device));
  RT_HIP(hipStreamSynchronize(c->stream));
  c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
  c->timed = false;
  return RT_OK;
}

int rt_ctx_synchronize(rt_ctx *c) {
  if (!c) return fail(RT_ERR_ARG, "ctx is null");
  RT_HIP(hipSetDevice(c->device));
  RT_HIP(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_scene_upload(rt_ctx *c, const rt_sphere *spheres, size_t n) {
  if (!c || !spheres) return fail(RT_ERR_ARG, "null argument");
  if (n == 0) return fail(RT_ERR_ARG, "empty scene");
  if (n > kMaxSpheres) return fail(RT_ERR_LIMIT, "more than 65535 spheres");
  for (size_t i = 0; i < n; ++i) {
    if (spheres[i].type < RT_MATTE || spheres[i].type > RT_DIELECTRIC)
      return fail(RT_ERR_ARG, "unknown material type in sphere " + std::to_string(i));
  }
  RT_HIP(hipSetDevice(c->device));
  const uint32_t n_pad = (uint32_t)((n + kDrainEvery - 1) / kDrainEvery * kDrainEvery);
  std::vector<float4> filter(n_pad), exact(n), mat(n);
  std::vector<float2> rad_type(n);
  for (size_t i = 0; i < n; ++i) {
    const rt_sphere &s = spheres[i];
    FilterSphere f = make_filter_sphere(s.center[0], s.center[1], s.center[2], s.radius);
    filter[i] = make_float4(f.cx, f.cy, f.cz, f.kprime);
    exact[i] = make_float4(s.center[0], s.center[1], s.center[2], s.radius * s.radius);  // c/main.c:219
    mat[i] = make_float4(s.albedo[0], s.albedo[1], s.albedo[2], s.param);
    float type_bits;
    int32_t ty = s.type;
    std::memcpy(&type_bits, &ty, sizeof ty);
    rad_type[i] = make_float2(s.radius, type_bits);
  }
  // pads: K' so large that b*b - a*K' is negative for any ray
  for (size_t i = n; i < n_pad; ++i) filter[i] = make_float4(0.0f, 0.0f, 0.0f, 1.0e30f);
  int rc;
  if ((rc = c->filter.reserve(n_pad)) || (rc = c->exact.reserve(n)) || (rc = c->mat.reserve(n)) ||
      (rc = c->rad_type.reserve(n)))
    return rc;
  RT_HIP(hipStreamSynchronize(c->stream));
  RT_HIP(hipMemcpy(c->filter.p, filter.data(), n_pad * sizeof(float4), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(c->exact.p, exact.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(c->mat.p, mat.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(c->rad_type.p, rad_type.data(), n * sizeof(float2), hipMemcpyHostToDevice));
  c->n = (uint32_t)n;
  c->n_pad = n_pad;
  return RT_OK;
}

int rt_render_device(rt_ctx *c, const rt_camera *cam, const rt_job *job, uint8_t *d_rgb8, float *d_linear,
                     int32_t *d_first_hit, uint32_t *d_hit_sum) {
  if (!c || !cam || !d_rgb8) return fail(RT_ERR_ARG, "null argument");
  int rc = check_job(job);
  if (rc != RT_OK) return rc;
  if (c->n == 0) return fail(RT_ERR_NO_SCENE, "no scene uploaded");
  if (rt_job_rows(job) == 0) return fail(RT_ERR_ARG, "job covers no rows");
  RT_HIP(hipSetDevice(c->device));
  if (job->kernel == RT_KERNEL_WAVEFRONT) return fail(RT_ERR_ARG, "wavefront variant not built");
  return launch_mega(c, cam, job, d_rgb8, d_linear, d_first_hit, d_hit_sum);
}

int rt_render_host(rt_ctx *c, const rt_camera *cam, const rt_job *job, uint8_t *h_rgb8, float *h_linear,
                   int32_t *h_first_hit, uint32_t *h_hit_sum) {
  if (!c || !cam || !h_rgb8) return fail(RT_ERR_ARG, "null argument");
  int rc = check_job(job);
  if (rc != RT_OK) return rc;
  RT_HIP(hipSetDevice(c->device));
  const size_t px = (size_t)rt_job_rows(job) * job->nx;
  if (px == 0) return fail(RT_ERR_ARG, "job covers no rows");
  if ((rc = c->out_rgb8.reserve(3 * px))) return rc;
  if (h_linear && (rc = c->out_linear.reserve(3 * px))) return rc;
  if (h_first_hit && (rc = c->out_first_hit.reserve(px))) return rc;
  if (h_hit_sum && (rc = c->out_hit_sum.reserve(px))) return rc;
  rc = rt_render_device(c, cam, job, c->out_rgb8.p, h_linear ? c->out_linear.p : nullptr,
                        h_first_hit ? c->out_first_hit.p : nullptr, h_hit_sum ? c->out_hit_sum.p : nullptr);
  if (rc != RT_OK) return rc;
  RT_HIP(hipMemcpyAsync(h_rgb8, c->out_rgb8.p, 3 * px, hipMemcpyDeviceToHost, c->stream));
  if (h_linear)
    RT_HIP(hipMemcpyAsync(h_linear, c->out_linear.p, 3 * px * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (h_first_hit)
    RT_HIP(hipMemcpyAsync(h_first_hit, c->out_first_hit.p, px * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (h_hit_sum)
    RT_HIP(hipMemcpyAsync(h_hit_sum, c->out_hit_sum.p, px * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  RT_HIP(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_ctx_last_kernel_ms(rt_ctx *c, float *ms) {
  if (!c || !ms) return fail(RT_ERR_ARG, "null argument");
  if (!c->timed) return fail(RT_ERR_ARG, "no render has been timed on this context");
  RT_HIP(hipSetDevice(c->device));
  RT_HIP(hipEventSynchronize(c->ev_end));
  RT_HIP(hipEventElapsedTime(ms, c->ev_begin, c->ev_end));
  return RT_OK;
}

int rt_ctx_counters(rt_ctx *c, rt_counters *out, int reset) {
  if (!c) return fail(RT_ERR_ARG, "ctx is null");
  RT_HIP(hipSetDevice(c->device));
  RT_HIP(hipStreamSynchronize(c->stream));
  if (out) {
    unsigned long long v[4];
    RT_HIP(hipMemcpy(v, c->counters.p, sizeof v, hipMemcpyDeviceToHost));
    out->paths = v[0];
    out->segments = v[1];
    out->tests = v[2];
    out->candidates = v[3];
  }
  if (reset) RT_HIP(hipMemset(c->counters.p, 0, 4 * sizeof(unsigned long long)));
  return RT_OK;
}

int rt_debug_closest_hit(rt_ctx *c, const float *rays, size_t n_rays, float tmin, float tmax, int32_t *index,
                         float *t) {
  if (!c || !rays || !index || !t) return fail(RT_ERR_ARG, "null argument");
  if (c->n == 0) return fail(RT_ERR_NO_SCENE, "no scene uploaded");
  if (n_rays == 0) return RT_OK;
  if (n_rays > 0x7fffffffu) return fail(RT_ERR_LIMIT, "too many rays");
  RT_HIP(hipSetDevice(c->device));
  DevBuf<float> d_rays, d_t;
  DevBuf<int32_t> d_idx;
  int rc;
  if ((rc = d_rays.reserve(6 * n_rays)) || (rc = d_t.reserve(n_rays)) || (rc = d_idx.reserve(n_rays))) {
    d_rays.release(); d_t.release(); d_idx.release();
    return rc;
  }
  hipError_t e = hipMemcpy(d_rays.p, rays, 6 * n_rays * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    size_t lds = (size_t)kMaxLdsSpheres * sizeof(float4) + (size_t)kListCap * kBlockThreads * sizeof(uint16_t);
    uint32_t blocks = (uint32_t)((n_rays + kBlockThreads - 1) / kBlockThreads);
    hipLaunchKernelGGL(closest_hit_kernel, dim3(blocks), dim3(kBlockThreads), lds, c->stream, scene_view(c), d_rays.p,
                       (uint32_t)n_rays, tmin, tmax, d_idx.p, d_t.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(index, d_idx.p, n_rays * sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(t, d_t.p, n_rays * sizeof(float), hipMemcpyDeviceToHost);
  d_rays.release(); d_t.release(); d_idx.release();
  if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("closest_hit: ") + hipGetErrorString(e));
  return RT_OK;
}

int rt_debug_uniforms(rt_ctx *c, uint32_t seed, uint32_t pixel, uint32_t sample, size_t n, double *out) {
  if (!c || !out) return fail(RT_ERR_ARG, "null argument");
  if (n == 0) return RT_OK;
  if (n > (1u << 24)) return fail(RT_ERR_LIMIT, "too many draws");
  RT_HIP(hipSetDevice(c->device));
  DevBuf<double> d;
  int rc = d.reserve(n);
  if (rc) return rc;
  hipLaunchKernelGGL(uniforms_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, seed, pixel, sample,
                     (uint32_t)n, d.p);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(out, d.p, n * sizeof(double), hipMemcpyDeviceToHost);
  d.release();
  if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("uniforms: ") + hipGetErrorString(e));
  return RT_OK;
}

}  // extern "C"
