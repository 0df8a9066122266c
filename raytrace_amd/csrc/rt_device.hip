// rt_device.hip -- HIP kernels (gfx950) and the C ABI of include/rt_amd.h.
//
// trace_kernel: persistent workgroups take camera paths (pixel, sample) from a
// global queue; a lane whose path ends is handed the next one, so waves stay
// full.  One trip of its loop is one bounce of every live lane:
//   (1) closest hit.  The sphere list sits in LDS in scan order (rt_cluster.h)
//       as 32-byte sphere pairs.  Every lane's half line is tested against the
//       boxes of the quarters (128 positions), of the super-groups of a quarter
//       some lane can touch and, likewise one level down, of the super-groups'
//       groups of eight spheres; a lane whose ray can touch a group lists
//       (lane, group).  The listed pairs are then filtered 64 at a time by
//       whichever lanes are free: 32 packed FMAs per pair (the eight-FMA
//       conservative filter of rt_core.h on eight spheres).  (lane, sphere)
//       pairs that pass go to a second per-wave list, drained the same way,
//       each pair through the reference's exact arithmetic (c/main.c:214-236)
//       and an LDS 64-bit atomic min on {t, original index} -- the hit the
//       reference's linear scan returns, because a sphere the filter rejects
//       can never have D > 0 and the scan's result does not depend on its
//       order except through that tie-break.  RT_FLAG_FULL_SCAN filters every
//       sphere for every lane with wave-wide broadcast reads instead.
//   (2)-(5) shading (c/main.c:163-211, 258-283) and new camera rays
//       (c/structs.c:81-92) around ONE unit-ball rejection loop per trip.
// resolve_kernel adds each pixel's samples in the reference's order
// (c/main.c:462-477).  The wf_* kernels are a material-sorted wavefront
// variant of the same arithmetic, kept for comparison.
//
// Compiled with -ffp-contract=off (see rt_core.h).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_core.h"
#include "rt_cluster.h"

using namespace rtcore;

// RT_EXP: timing experiments that run one part of the scan twice (same results, the extra time is that
// part's cost); 0 in every real build
#ifndef RT_EXP
#define RT_EXP 0
#endif
#if RT_EXP
#define RT_ASM asm volatile   // so that the repeated block is not merged with the first
#else
#define RT_ASM asm
#endif
// RT_EXP 39: every part of a trip counts the lanes it had at work and the times it ran (thread efficiency by
// part); the sums land in the first words of the wave-log area (rt_debug_wave_log)
// RT_EXP 31..34: the profiling build's "drain" column shows one stage of the shading phase instead
// (31 rejection loop, 32 shading before it, 33 new paths, 34 shading after it)
#define RT_STAGE_BEGIN(N) const unsigned long long stage_t0_##N = (RT_EXP == N && kProfile) ? tick() : 0ull
#define RT_STAGE_END(N) \
  do {                  \
    if (RT_EXP == N && kProfile) c_drain += tick() - stage_t0_##N; \
  } while (0)

// RT_EXP 35..38: the profiling build's "drain" column shows one stage of the scan instead (35 ray set-up,
// 36 super-group boxes, 37 group-box rounds without the filter passes they trigger, 38 filter passes)
// RT_EXP 40: ... or, for a scene scanned in place, choosing the batches: region and batch boxes, near-to-far picks
#define RT_SCAN_STAGE ((RT_EXP >= 35 && RT_EXP <= 38) || RT_EXP == 40)
#define RT_HSTAGE_BEGIN(N) const unsigned long long hstage_t0_##N = (RT_EXP == N) ? tick() : 0ull
#define RT_HSTAGE_END(N) \
  do {                   \
    if (RT_EXP == N) h.drain_cycles += tick() - hstage_t0_##N; \
  } while (0)
#define RT_HSTAGE_EXCLUDE_BEGIN(N) const unsigned long long hstage_x0_##N = (RT_EXP == N) ? tick() : 0ull
#define RT_HSTAGE_EXCLUDE_END(N) \
  do {                           \
    if (RT_EXP == N) h.drain_cycles -= tick() - hstage_x0_##N; \
  } while (0)

// RT_EXP 61: the new-path stage reduced to a dozen instructions (other rays than the product's): the ceiling of what a
// cheaper stage could save

// RT_EXP 51..58: LDS bank-conflict attribution.  One kind of LDS access is issued a second time (same addresses,
// result thrown away); the growth of SQ_LDS_BANK_CONFLICT and SQ_LDS_IDX_ACTIVE over the plain build is that
// access kind's share.  51 ray table read by owner (pair filter), 52 pair image read by group (pair filter; scenes
// resident in LDS), 53 ray table read by owner (exact evaluation), 54 the 64-bit minimum on best[owner], 55 the
// (lane, group) list and group-size reads, 56 the (lane, group) list writes, 57 the (lane, sphere) list writes and
// reads, 58 the per-lane ray table and key writes.
#define RT_LDS_EXP (RT_EXP >= 51 && RT_EXP <= 58)
#if RT_LDS_EXP
typedef float rt_f32x4 __attribute__((ext_vector_type(4)));
typedef float rt_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const unsigned char lds_bytes;
__device__ __forceinline__ uint32_t lds_address(const void *p) { return (uint32_t)(uintptr_t)(lds_bytes *)p; }
__device__ __forceinline__ void lds_again_read_b128(uint32_t a) {
  rt_f32x4 x;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(a) : "memory");
}
__device__ __forceinline__ void lds_again_read_b64(uint32_t a) {
  rt_f32x2 x;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(a) : "memory");
}
__device__ __forceinline__ void lds_again_read_b32(uint32_t a) {
  uint32_t x;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(a) : "memory");
}
__device__ __forceinline__ void lds_again_read_u16(uint32_t a) {
  uint32_t x;
  asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(a) : "memory");
}
__device__ __forceinline__ void lds_again_read_u8(uint32_t a) {
  uint32_t x;
  asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(a) : "memory");
}
__device__ __forceinline__ void lds_again_write_b16(uint32_t a, uint32_t v) {
  asm volatile("ds_write_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_again_write_b32(uint32_t a, uint32_t v) {
  asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_again_write_b64(uint32_t a, unsigned long long v) {
  asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_again_write_b128(uint32_t a, float4 v) {
  const rt_f32x4 x = {v.x, v.y, v.z, v.w};
  asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(a), "v"(x) : "memory");
}
#endif

namespace {

// ----------------------------------------------------------------- limits
constexpr int kBlockThreads = 256;
constexpr int kChunk = 8;             // spheres per unrolled filter group
constexpr int kDrainEvery = 16;       // spheres per round of the scan (two groups)
#ifndef RT_MAX_RESIDENT
#define RT_MAX_RESIDENT 2048
#endif
constexpr uint32_t kMaxLdsSpheres = RT_MAX_RESIDENT;  // scan positions of a scene that stays in LDS for the whole kernel
#ifndef RT_BATCH_SPHERES
#define RT_BATCH_SPHERES 1024
#endif
constexpr uint32_t kBatchSpheres = RT_BATCH_SPHERES;    // larger scenes: scan positions per staged batch (the LDS footprint of
                                           // a batch is that of the reference's final scene)
constexpr uint32_t kMaxSpheres = 65535;    // list entries are 16-bit
constexpr int kNumCounters = 18;
constexpr int kWaveLogBase = 32, kWaveLogEntries = 16384;   // profiling builds: one word per wave behind the counters
constexpr int kBoxTestCounter = 13;   // (the others are numbered where rt_ctx_counters reads them)

// The scan's arrays are in SCAN order (rt_cluster.h): position -> sphere, pads included.
struct DeviceScene {
  const float4 *filter;   // n_pad entries, sphere pairs {cx0,cx1,cy0,cy1},{cz0,cz1,K'0,K'1}; pads never pass
  const float4 *boxes;    // 3 entries per round of two groups: {cx0,cx1,cy0,cy1},{cz0,cz1,hx0,hx1},{hy0,hy1,hz0,hz1}
  const float4 *sboxes;   // 3 entries per pair of super-groups (64 spheres), same layout as boxes; then the batch boxes
  const float4 *qboxes;   // 3 entries per pair of quarters (256 spheres: a quarter is four super-groups), same layout
  const float4 *exact;    // n_pad entries {cx, cy, cz, r*r}
  const uint32_t *orig;   // n_pad entries: index in the caller's list (ties go to the lowest), ~0 for pads
  // for shading, in the CALLER's order (n entries):
  const float4 *geom_o;   // {cx, cy, cz, radius}
  const float4 *mat_o;    // {albedo rgb, param}
  const int32_t *type_o;  // material type
  const double *inv_r_o;  // 1.0 / (double)radius, for the divisions of c/main.c:229 (div_f32_by)
  const float4 *shade_o;  // the same four per sphere as one 48-byte record: {cx, cy, cz, radius}, {albedo rgb, param}
                          // (a dielectric: {r0 of its Schlick term, 1 / ref_idx, 0, ref_idx}),
                          // {1.0 / (double)radius as two words, type, 0} -- what the megakernel's shading reads
  const uint8_t *group_real;   // n_pad/8 entries: real (non-pad) spheres per group
  uint32_t n, n_pad;
  float sigma0, sigma1;   // per-ray growth of the group boxes (rt_core.h, rt_cluster.h)
  float cut_reach, cut_b; // constants of the distance cut (rtcore::cut_slack)
  uint32_t n_full;        // scan positions up to the last real sphere, rounded up to a round (16): what a
                          // scan without culling has to visit (the rest is padding)
  uint32_t n_always_groups;   // the first so many groups go on every ray's list untested; their entries in
                              // `boxes` say "never", and `sboxes` are the boxes around the other groups
};

// The boxes of the batches of a streamed scene follow the super-group boxes in the same array: 3 entries
// per pair of batches (kBatchSpheres scan positions each), same layout.
__device__ __host__ inline uint32_t batch_pairs(uint32_t n_pad) { return ((n_pad + kBatchSpheres - 1u) / kBatchSpheres + 1u) / 2u; }
// pairs of quarters (a quarter = four super-groups = 128 scan positions) in `entries` scan positions
__device__ __host__ inline uint32_t quarter_pairs(uint32_t entries) { return (entries + 255u) / 256u; }
// ... and behind those the boxes of the regions (kRegionBatches batches each), again in pairs
#ifndef RT_REGION_BATCHES
#define RT_REGION_BATCHES 4
#endif
constexpr uint32_t kRegionBatches = RT_REGION_BATCHES;
__device__ __host__ inline uint32_t region_pairs(uint32_t n_pad) { return ((2u * batch_pairs(n_pad) + kRegionBatches - 1u) / kRegionBatches + 1u) / 2u; }

__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

// shader clock for the profiling builds, read once everything issued before it has completed
__device__ __forceinline__ unsigned long long tick() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t = __builtin_amdgcn_s_memtime();
  asm volatile("" ::: "memory");
  return t;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// smallest v of the wave's 64 lanes (all lanes active): four steps inside each row of 16 lanes, two across rows
template <int kCtrl, int kRowMask>
__device__ __forceinline__ float wave_min_step(float x) {
  const int y = __builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), kCtrl, kRowMask, 0xf, false);
  return __builtin_fminf(x, __int_as_float(y));
}
__device__ __forceinline__ float wave_min(float v) {
  v = wave_min_step<0xb1, 0xf>(v);    // quad_perm [1,0,3,2]
  v = wave_min_step<0x4e, 0xf>(v);    // quad_perm [2,3,0,1]
  v = wave_min_step<0x141, 0xf>(v);   // row_half_mirror
  v = wave_min_step<0x140, 0xf>(v);   // row_mirror: every lane of a row holds the row's minimum
  v = wave_min_step<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
  v = wave_min_step<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3: lane 63 holds the minimum
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Result of one closest-hit query of a lane.
struct HitState {
  float closest;    // smallest t (the query's tmax while there is no hit)
  int index;        // index of the sphere in the caller's list, -1 for none
  uint32_t cand;    // exact evaluations the wave performed (wave-uniform)
  uint32_t box_tests;    // box tests every ray of the wave had done for it (wave-uniform; culled scan)
  uint32_t scan_tests;   // sphere filter tests every ray of the wave had done for it (wave-uniform; scan without culling)
  uint32_t lane_tests;   // sphere filter tests this lane evaluated, for whichever ray (culled scan)
  uint32_t own_groups;   // groups this lane's own ray needed (diagnostic: what a per-ray cull would scan)
  uint32_t wave_groups;  // groups the wave scanned
  unsigned long long drain_cycles;   // shader cycles in the closing drain (profiling builds only)
#if RT_EXP == 39
  uint32_t fill[12];   // lanes at work and passes run: pair-filter passes, drain passes, always-group steps; [6] batch visits, [7] batches picked (streamed scenes)
#endif
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

// LDS carve-up shared by every kernel that scans.  Per workgroup: the pair
// image of the resident batch, its group boxes and the number of real spheres
// of each group; per lane: its ray (so that any lane of the wave can evaluate a
// sphere for it) and its best hit as one 64-bit key {t bits, sphere index}; per
// wave: the list of (lane, group) pairs whose box test passed and await the
// sphere filter, and the list of (lane, sphere) pairs that passed the filter
// and await exact evaluation.
// (lane, sphere) entries per wave.  The full scan adds up to 64 * kChunk per chunk and drains when fewer
// are free; the culled scan looks at what a pass produced before appending it (filter_listed_pairs) and
// gets by with a list that 32 lanes' worth of entries fit.
constexpr uint32_t kWaveListCapFull = 768, kWaveListCapCull = 288;
constexpr uint32_t wave_list_cap(bool cull) { return cull ? kWaveListCapCull : kWaveListCapFull; }
constexpr uint32_t kWaveListFlushFull = kWaveListCapFull - 64u * kChunk;   // drain above this before a chunk
static_assert(kWaveListCapFull >= 64u * kChunk && kWaveListCapCull >= 32u * kChunk, "list too short");
constexpr uint32_t kPairListCap = 280;              // (lane, group) entries per wave, 16 bits each (with this and a
                                                    // (lane, sphere) list of 288 the workgroup's LDS is 31,968 bytes for a
                                                    // 512-position scene: five fit a CU's 160 KiB, which is handed out
                                                    // in 1280-byte units)
constexpr uint32_t kPairListFlush = kPairListCap - 2u * 64u;       // a round adds at most two groups for 64 lanes
// A scene scanned in place, between two batches: the listed (lane, group) pairs are filtered once there are this many (a
// pass takes 64; dense field: 3.65e9 paths/s against 3.54e9 with a pass after every batch, 32 .. 64 within 1 %), the
// candidates are evaluated after every batch that produced any -- holding them back as well costs more in boxes the
// later cuts would have ruled out than the fuller passes save (40 / 40: 3.48e9).
#ifndef RT_MERGE_PAIRS
#define RT_MERGE_PAIRS 40
#endif
#ifndef RT_MERGE_CANDIDATES
#define RT_MERGE_CANDIDATES 1
#endif
constexpr uint32_t kMergePairs = RT_MERGE_PAIRS, kMergeCandidates = RT_MERGE_CANDIDATES;

// The per-lane ray table.  Any lane of a wave reads any other lane's ray from it (the lists name an owner), so
// what matters is how 32 or 16 arbitrary owners fall on the LDS banks.  Eight-byte entries, 64 owners to a row of
// 512 bytes, read with ds_read_b64: that instruction serves 32 lanes per cycle over 64 banks, so only owners o and
// o + 32 share banks and a read costs at most twice its conflict-free time whatever the owners are.  (As 48 bytes
// per lane read by the compiler's choice of ds_read2_b32 / ds_read_b96 -- 32 banks, twelve-dword stride: only eight
// distinct bank positions -- every read of 32 owners was a four-way conflict or worse.)
constexpr uint32_t kRayRows = 6, kRayRowBytes = 64u * 8u;
typedef float rt_v2f __attribute__((ext_vector_type(2)));

struct ScanLds {
  float4 *pairs;
  float4 *boxes;                 // 3 per round (16 spheres)
  float4 *sboxes;                // 3 per pair of super-groups (64 spheres)
  float4 *qboxes;                // 3 per pair of quarters (256 spheres)
  float2 *rays;                  // per wave six rows of 64 eight-byte entries, one entry per lane and row (ray_table_*):
                                 // {Ax, Ay}, {Az, a}, {Bx, By}, {Bz, -(B.B) of the filter}, {1/a as a double},
                                 // {A.B, naaa of the filter}
  unsigned long long *best;      // 1 per lane
  uint32_t *wlist;               // wave_list_cap() per wave: (owner lane << 16) | scan position
  uint16_t *plist;               // kPairListCap entries per wave: (group << 6) | owner lane -- 16 bits each for a scene
                                 // resident in LDS, 32 bits each (the group counted from the start of the scan order) for
                                 // one scanned in place
  uint8_t *real8;                // per group of the batch: its real (non-pad) spheres
};

// The pair image in LDS has one unused 16-byte entry after every two groups (16 entries, 256 bytes).  The pair
// filter reads entry k of a group per lane, each lane its own group; ds_read_b128 serves 16 lanes per cycle over
// the 16 sixteen-byte slots of a 256-byte bank row, and without the gap entry k of every group of the same parity
// lies in the same slot (a group is 128 bytes): different groups in one pass were served one after the other.
// With the gap, entry k of group g lies in slot (8 g + (g >> 1) + k) mod 16, distinct for 16 consecutive groups.
// An image read in place from global memory (scenes too large for LDS) has no gaps.
template <bool kInPlace>
__device__ __host__ __forceinline__ uint32_t pair_slot(uint32_t entry) { return kInPlace ? entry : entry + (entry >> 4); }
__device__ __host__ inline uint32_t pair_image_entries(uint32_t entries) { return entries + entries / 16u; }

template <bool kCull>
__device__ __forceinline__ ScanLds carve_lds(unsigned char *smem, uint32_t entries) {
  ScanLds l;
  l.pairs = reinterpret_cast<float4 *>(smem);
  l.boxes = l.pairs + pair_image_entries(entries);
  l.sboxes = l.boxes + 3u * (entries / 16u);
  l.qboxes = l.sboxes + 3u * (entries / 64u);
  l.rays = reinterpret_cast<float2 *>(l.qboxes + 3u * quarter_pairs(entries));
  l.best = reinterpret_cast<unsigned long long *>(l.rays + kRayRows * kBlockThreads);
  l.wlist = reinterpret_cast<uint32_t *>(l.best + kBlockThreads);
  l.plist = reinterpret_cast<uint16_t *>(l.wlist + (kBlockThreads / 64) * wave_list_cap(kCull));
  l.real8 = reinterpret_cast<uint8_t *>(l.plist + (kBlockThreads / 64) * kPairListCap * (entries == 0u ? 2u : 1u));
  return l;
}

__device__ __forceinline__ void stage_batch(const DeviceScene &sc, const ScanLds &lds, uint32_t base, uint32_t count,
                                            uint32_t tid) {
  for (uint32_t i = tid; i < count; i += kBlockThreads) lds.pairs[pair_slot<false>(i)] = sc.filter[base + i];
  for (uint32_t i = tid; i < 3u * (count / 16u); i += kBlockThreads) lds.boxes[i] = sc.boxes[3u * (base / 16u) + i];
  for (uint32_t i = tid; i < 3u * (count / 64u); i += kBlockThreads) lds.sboxes[i] = sc.sboxes[3u * (base / 64u) + i];
  for (uint32_t i = tid; i < 3u * quarter_pairs(count); i += kBlockThreads) lds.qboxes[i] = sc.qboxes[3u * (base / 256u) + i];
  for (uint32_t i = tid; i < count / 8u; i += kBlockThreads) lds.real8[i] = sc.group_real[base / 8u + i];
}

// What a wave carries through one scan.
typedef __attribute__((address_space(3))) uint32_t lds_u32;

typedef __attribute__((address_space(3))) uint16_t lds_u16;

struct WaveScan {
  uint32_t wlist_off;            // LDS byte address of this wave's (lane, sphere) list (wave-uniform, in a scalar register)
  uint32_t plist_off;            // the same for its (lane, group) list
  uint32_t rays_off;             // LDS byte address of this wave's ray table (wave-uniform, in a scalar register)
  unsigned long long *best;      // this wave's 64 keys
  uint32_t wcnt;                 // entries in the (lane, sphere) list (wave-uniform)
  uint32_t pcnt;                 // entries in the (lane, group) list (wave-uniform)
  uint32_t lane;
  uint32_t lane16;               // lane << 16, the owner field of a (lane, sphere) entry
  float tmin;                    // of the query (wave-uniform)
};

// Five rows of the owner's entry of the ray table (addr = table + 8 * owner), as five ds_read_b64 in one block that
// ends with the wait for them: written out because the compiler, left to itself, pairs such loads into
// ds_read2_b64 / ds_read2_b32, which are served over 32 banks at half the rate.
template <int kA, int kB, int kC, int kD, int kE>
__device__ __forceinline__ void ray_table_read5(uint32_t addr, rt_v2f &a, rt_v2f &b, rt_v2f &c, rt_v2f &d, rt_v2f &e) {
  asm volatile("ds_read_b64 %0, %5 offset:%6\n\t"
               "ds_read_b64 %1, %5 offset:%7\n\t"
               "ds_read_b64 %2, %5 offset:%8\n\t"
               "ds_read_b64 %3, %5 offset:%9\n\t"
               "ds_read_b64 %4, %5 offset:%10\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e)
               : "v"(addr), "n"(kA * (int)kRayRowBytes), "n"(kB * (int)kRayRowBytes), "n"(kC * (int)kRayRowBytes), "n"(kD * (int)kRayRowBytes),
                 "n"(kE * (int)kRayRowBytes)
               : "memory");
}

// Exact evaluation (c/main.c:214-236) of the listed (lane, sphere) pairs, 64
// pairs at a time whichever lanes they belong to: a lane's candidates are few
// (two on average) but unevenly spread, and this way the double-precision
// roots are computed with every lane busy.  The reference's shrinking-tmax
// scan returns the smallest first-root-in-(tmin, FLT_MAX) and, among equal t,
// the lowest index; an LDS 64-bit atomic min on {t bits, index} takes exactly
// that minimum in any order (t > tmin >= 0, so its bits order like its value).
__device__ __forceinline__ void drain_wave(const DeviceScene &sc, WaveScan &w, HitState &h) {
#if RT_EXP == 11
  for (int rep = 0; rep < 2; ++rep)
#endif
  for (uint32_t e0 = 0; e0 < w.wcnt; e0 += 64u) {
    const uint32_t e = e0 + w.lane;
#if RT_EXP == 39
    h.fill[2] += min(64u, w.wcnt - e0);
    h.fill[3] += 1u;
#endif
    if (e < w.wcnt) {
      const uint32_t entry = *(lds_u32 *)(uintptr_t)(w.wlist_off + (e << 2));
#if RT_EXP == 57
      lds_again_read_b32(w.wlist_off + (e << 2));
#endif
      const uint32_t owner = entry >> 16;
      // byte offsets of the sphere's entries: (entry & 0xffff) << 4 and << 2, one instruction each with
      // the shift count in a scalar register (the compiler's form of the same instruction keeps the
      // count in a vector register for the whole kernel)
      uint32_t off16, off4;
      asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(off16) : "s"(4u), "v"(entry));
      rt_v2f r0, r1, r2, r3, r4;
      ray_table_read5<0, 1, 2, 3, 4>(w.rays_off + (owner << 3), r0, r1, r2, r3, r4);
#if RT_EXP == 53
      {
        rt_v2f x0, x1, x2, x3, x4;
        ray_table_read5<0, 1, 2, 3, 4>(w.rays_off + (owner << 3), x0, x1, x2, x3, x4);
      }
#endif
      const float4 ra = make_float4(r0[0], r0[1], r1[0], r1[1]), rb = make_float4(r2[0], r2[1], r3[0], r3[1]);
      const double inv_a = __hiloint2double(__float_as_int(r4[1]), __float_as_int(r4[0]));
      const float4 ex = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sc.exact) + off16);
      // (the sphere's original index is fetched beside its geometry, not after the hit test: one memory latency per
      // pass instead of two)
      asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(off4) : "s"(2u), "v"(entry));
      const uint32_t original = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(sc.orig) + off4);
      float t;
      const bool in_range = exact_hit(v3(ra.x, ra.y, ra.z), v3(rb.x, rb.y, rb.z), ra.w, inv_a, ex.x, ex.y, ex.z, ex.w, w.tmin, kFltMax, t);
#if RT_EXP == 39
      h.fill[10] += (uint32_t)__popcll(__ballot(in_range));                      // candidates with a root in range
      h.fill[11] += (uint32_t)__popcll(__ballot((entry & 0xffffu) == 0u));         // candidates at scan position 0
#endif
      if (in_range) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | original;
        atomicMin(&w.best[owner], key);
#if RT_EXP == 54
        atomicMin(&w.best[owner], key);
#endif
      }
    }
  }
  h.cand += w.wcnt;
  w.wcnt = 0;
}

// The nine per-ray coefficients of the filter packed two to a register pair.
// v_pk_fma_f32 can take either half of a pair for both of its lanes (op_sel /
// op_sel_hi), so a coefficient that is the same for both spheres of a pair
// needs one register, not a splat pair: 10 VGPRs per ray instead of 18.
struct FilterRayP {
  f32x2 p0;  // {-Bx, -By}
  f32x2 p1;  // {-Bz, A.B}
  f32x2 p2;  // {-2Ax, -2Ay}
  f32x2 p3;  // {-2Az, -(B.B)}
  f32x2 p4;  // {-(B.B)(A.A)(1-kappa), unused}
};

__device__ __forceinline__ FilterRayP pack_ray(const FilterRay &f) {
  FilterRayP r;
  r.p0 = f32x2{f.nbx, f.nby};
  r.p1 = f32x2{f.nbz, f.ab};
  r.p2 = f32x2{f.m2ax, f.m2ay};
  r.p3 = f32x2{f.m2az, f.na};
  r.p4 = f32x2{f.naaa, 0.0f};
  return r;
}

// filter_discriminant2 (rt_core.h) for ONE sphere pair, same eight fused
// multiply-adds in the same order, hence the same bits; cx..kp are the pair's
// {sphere 0, sphere 1} values.
#define RT_BCAST_LO_S0 "op_sel:[0,0,0] op_sel_hi:[0,1,1]"
#define RT_BCAST_HI_S0 "op_sel:[1,0,0] op_sel_hi:[1,1,1]"
__device__ __forceinline__ f32x2 filter_pair(const FilterRayP &f, f32x2 cx, f32x2 cy, f32x2 cz, f32x2 kp) {
  f32x2 d, t;
  RT_ASM("v_pk_fma_f32 %0, %2, %8, %2 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"   // b  = -Bz*cz + A.B
      "v_pk_fma_f32 %1, %4, %8, %9 " RT_BCAST_LO_S0 "\n\t"                 // c0 = -2Az*cz + K'
      "v_pk_fma_f32 %0, %3, %7, %0 " RT_BCAST_HI_S0 "\n\t"                 // b += -By*cy
      "v_pk_fma_f32 %1, %5, %7, %1 " RT_BCAST_HI_S0 "\n\t"                 // c0 += -2Ay*cy
      "v_pk_fma_f32 %0, %3, %6, %0 " RT_BCAST_LO_S0 "\n\t"                 // b += -Bx*cx
      "v_pk_fma_f32 %1, %5, %6, %1 " RT_BCAST_LO_S0 "\n\t"                 // c0 += -2Ax*cx
      "s_nop 0\n\t"
      "v_pk_fma_f32 %0, %0, %0, %10 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"  // d = b*b + naaa
      "s_nop 0\n\t"
      "v_pk_fma_f32 %0, %4, %1, %0 " RT_BCAST_HI_S0                        // d += -(B.B)*c0
      : "=&v"(d), "=&v"(t)
      : "v"(f.p1), "v"(f.p0), "v"(f.p3), "v"(f.p2), "v"(cx), "v"(cy), "v"(cz), "v"(kp), "v"(f.p4));
  return d;
}

// The same for the four pairs of a chunk in one block, the eight chains
// interleaved so that no instruction waits on the one before it.
__device__ __forceinline__ void filter_four_pairs(const FilterRayP &f, const float4 (&q)[8], f32x2 (&d)[4]) {
  f32x2 t0, t1, t2, t3;
  const f32x2 cx0 = {q[0].x, q[0].y}, cy0 = {q[0].z, q[0].w}, cz0 = {q[1].x, q[1].y}, kp0 = {q[1].z, q[1].w};
  const f32x2 cx1 = {q[2].x, q[2].y}, cy1 = {q[2].z, q[2].w}, cz1 = {q[3].x, q[3].y}, kp1 = {q[3].z, q[3].w};
  const f32x2 cx2 = {q[4].x, q[4].y}, cy2 = {q[4].z, q[4].w}, cz2 = {q[5].x, q[5].y}, kp2 = {q[5].z, q[5].w};
  const f32x2 cx3 = {q[6].x, q[6].y}, cy3 = {q[6].z, q[6].w}, cz3 = {q[7].x, q[7].y}, kp3 = {q[7].z, q[7].w};
  RT_ASM("v_pk_fma_f32 %0, %8, %15, %8 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %1, %8, %19, %8 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %2, %8, %23, %8 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %3, %8, %27, %8 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %4, %10, %15, %16 " RT_BCAST_LO_S0 "\n\t"
      "v_pk_fma_f32 %5, %10, %19, %20 " RT_BCAST_LO_S0 "\n\t"
      "v_pk_fma_f32 %6, %10, %23, %24 " RT_BCAST_LO_S0 "\n\t"
      "v_pk_fma_f32 %7, %10, %27, %28 " RT_BCAST_LO_S0 "\n\t"
      "v_pk_fma_f32 %0, %9, %14, %0 " RT_BCAST_HI_S0 "\n\t"
      "v_pk_fma_f32 %1, %9, %18, %1 " RT_BCAST_HI_S0 "\n\t"
      "v_pk_fma_f32 %2, %9, %22, %2 " RT_BCAST_HI_S0 "\n\t"
      "v_pk_fma_f32 %3, %9, %26, %3 " RT_BCAST_HI_S0 "\n\t"
      "v_pk_fma_f32 %4, %11, %14, %4 " RT_BCAST_HI_S0 "\n\t"
      "v_pk_fma_f32 %5, %11, %18, %5 " RT_BCAST_HI_S0 "\n\t"
      "v_pk_fma_f32 %6, %11, %22, %6 " RT_BCAST_HI_S0 "\n\t"
      "v_pk_fma_f32 %7, %11, %26, %7 " RT_BCAST_HI_S0 "\n\t"
      "v_pk_fma_f32 %0, %9, %13, %0 " RT_BCAST_LO_S0 "\n\t"
      "v_pk_fma_f32 %1, %9, %17, %1 " RT_BCAST_LO_S0 "\n\t"
      "v_pk_fma_f32 %2, %9, %21, %2 " RT_BCAST_LO_S0 "\n\t"
      "v_pk_fma_f32 %3, %9, %25, %3 " RT_BCAST_LO_S0 "\n\t"
      "v_pk_fma_f32 %4, %11, %13, %4 " RT_BCAST_LO_S0 "\n\t"
      "v_pk_fma_f32 %5, %11, %17, %5 " RT_BCAST_LO_S0 "\n\t"
      "v_pk_fma_f32 %6, %11, %21, %6 " RT_BCAST_LO_S0 "\n\t"
      "v_pk_fma_f32 %7, %11, %25, %7 " RT_BCAST_LO_S0 "\n\t"
      "v_pk_fma_f32 %0, %0, %0, %12 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %1, %1, %1, %12 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %2, %2, %2, %12 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %3, %3, %3, %12 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %0, %10, %4, %0 " RT_BCAST_HI_S0 "\n\t"
      "v_pk_fma_f32 %1, %10, %5, %1 " RT_BCAST_HI_S0 "\n\t"
      "v_pk_fma_f32 %2, %10, %6, %2 " RT_BCAST_HI_S0 "\n\t"
      "v_pk_fma_f32 %3, %10, %7, %3 " RT_BCAST_HI_S0
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(f.p1), "v"(f.p0), "v"(f.p3), "v"(f.p2), "v"(f.p4),             // %8 .. %12
        "v"(cx0), "v"(cy0), "v"(cz0), "v"(kp0),                            // %13 .. %16
        "v"(cx1), "v"(cy1), "v"(cz1), "v"(kp1),                            // %17 .. %20
        "v"(cx2), "v"(cy2), "v"(cz2), "v"(kp2),                            // %21 .. %24
        "v"(cx3), "v"(cy3), "v"(cz3), "v"(kp3));                           // %25 .. %28
}

// The per-ray values of the group box test (rtcore::BoxRay) packed like
// FilterRayP, and box_pass for the two boxes of a round at once: the nine
// fused multiply-adds of rtcore::box_pass on both boxes (same operands, same
// order, hence the same bits), then per box min3 / max3 / max.
struct BoxRayP {
  f32x2 p0;  // {1/Bx, 1/By}
  f32x2 p1;  // {1/Bz, -Ax/Bx}
  f32x2 p2;  // {-Ay/By, -Az/Bz}
  f32x2 p3;  // {(1+sigma)/|Bx|, (1+sigma)/|By|}
  f32x2 p4;  // {(1+sigma)/|Bz|, unused}
};

__device__ __forceinline__ BoxRayP pack_box_ray(const BoxRay &r) {
  BoxRayP b;
  b.p0 = f32x2{r.ix, r.iy};
  b.p1 = f32x2{r.iz, r.nx};
  b.p2 = f32x2{r.ny, r.nz};
  b.p3 = f32x2{r.ax, r.ay};
  b.p4 = f32x2{r.az, 0.0f};
  return b;
}

// pass[i] = box i of the round is met by the lane's half line; q = the round's three LDS entries.
// kCut: ... and not farther along it than `cut` (rtcore: distance cut)
template <bool kCut = false>
__device__ __forceinline__ void box_pair(const BoxRayP &r, const float4 (&q)[3], bool (&pass)[2], float cut = 0.0f) {
  const f32x2 cx = {q[0].x, q[0].y}, cy = {q[0].z, q[0].w}, cz = {q[1].x, q[1].y};
  const f32x2 hx = {q[1].z, q[1].w}, hy = {q[2].x, q[2].y}, hz = {q[2].z, q[2].w};
  f32x2 nx, ny, nz, fx, fy, fz, mx, my, mz;
  RT_ASM("v_pk_fma_f32 %6, %9, %14, %10 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"   // mx = cx/Bx - Ax/Bx
         "v_pk_fma_f32 %7, %9, %15, %11 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\t"   // my
         "v_pk_fma_f32 %8, %10, %16, %11 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"  // mz
         "v_pk_fma_f32 %0, %12, %17, %6 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"  // near x
         "v_pk_fma_f32 %3, %12, %17, %6 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"                               // far x
         "v_pk_fma_f32 %1, %12, %18, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"  // near y
         "v_pk_fma_f32 %4, %12, %18, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"                               // far y
         "v_pk_fma_f32 %2, %13, %19, %8 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"  // near z
         "v_pk_fma_f32 %5, %13, %19, %8 op_sel:[0,0,0] op_sel_hi:[0,1,1]"                                    // far z
         : "=&v"(nx), "=&v"(ny), "=&v"(nz), "=&v"(fx), "=&v"(fy), "=&v"(fz), "=&v"(mx), "=&v"(my), "=&v"(mz)
         : "v"(r.p0), "v"(r.p1), "v"(r.p2), "v"(r.p3), "v"(r.p4),          // %9 .. %13
           "v"(cx), "v"(cy), "v"(cz), "v"(hx), "v"(hy), "v"(hz));          // %14 .. %19
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float tn, tf, lim;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn) : "v"(nx[i]), "v"(ny[i]), "v"(nz[i]));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf) : "v"(fx[i]), "v"(fy[i]), "v"(fz[i]));
    asm("v_max_f32 %0, 0, %1" : "=v"(lim) : "v"(tn));
    pass[i] = kCut ? (tf >= lim && !(tn > cut)) : tf >= lim;
  }
}

// box_pair for two boxes held in scalar registers (the batch boxes of a streamed scene, read with scalar
// loads): every instruction has one scalar operand, which is what the encoding allows.
struct BoxPairS {
  f32x2 cx, cy, cz, hx, hy, hz;
};

__device__ __forceinline__ void box_pair_scalar(const BoxRayP &r, const BoxPairS &b, bool (&pass)[2], float *near_bound = nullptr) {
  f32x2 nx, ny, nz, fx, fy, fz, mx, my, mz;
  RT_ASM("v_pk_fma_f32 %6, %9, %14, %10 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
         "v_pk_fma_f32 %7, %9, %15, %11 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\t"
         "v_pk_fma_f32 %8, %10, %16, %11 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
         "v_pk_fma_f32 %0, %12, %17, %6 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
         "v_pk_fma_f32 %3, %12, %17, %6 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
         "v_pk_fma_f32 %1, %12, %18, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
         "v_pk_fma_f32 %4, %12, %18, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
         "v_pk_fma_f32 %2, %13, %19, %8 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
         "v_pk_fma_f32 %5, %13, %19, %8 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
         : "=&v"(nx), "=&v"(ny), "=&v"(nz), "=&v"(fx), "=&v"(fy), "=&v"(fz), "=&v"(mx), "=&v"(my), "=&v"(mz)
         : "v"(r.p0), "v"(r.p1), "v"(r.p2), "v"(r.p3), "v"(r.p4),          // %9 .. %13
           "s"(b.cx), "s"(b.cy), "s"(b.cz), "s"(b.hx), "s"(b.hy), "s"(b.hz));   // %14 .. %19
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float tn, tf, lim;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn) : "v"(nx[i]), "v"(ny[i]), "v"(nz[i]));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf) : "v"(fx[i]), "v"(fy[i]), "v"(fz[i]));
    asm("v_max_f32 %0, 0, %1" : "=v"(lim) : "v"(tn));
    pass[i] = tf >= lim;
    if (near_bound) near_bound[i] = tn;
  }
}

// The same for two pairs (half a group): sixteen instructions, four chains.  The culled scan filters a
// group in two halves to stay inside the register budget of five waves per SIMD.
__device__ __forceinline__ void filter_two_pairs(const FilterRayP &f, const float4 (&q)[4], f32x2 &d0, f32x2 &d1) {
  f32x2 t0, t1;
  const f32x2 cx0 = {q[0].x, q[0].y}, cy0 = {q[0].z, q[0].w}, cz0 = {q[1].x, q[1].y}, kp0 = {q[1].z, q[1].w};
  const f32x2 cx1 = {q[2].x, q[2].y}, cy1 = {q[2].z, q[2].w}, cz1 = {q[3].x, q[3].y}, kp1 = {q[3].z, q[3].w};
  RT_ASM("v_pk_fma_f32 %0, %4, %11, %4 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
         "v_pk_fma_f32 %1, %4, %15, %4 op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
         "v_pk_fma_f32 %2, %6, %11, %12 " RT_BCAST_LO_S0 "\n\t"
         "v_pk_fma_f32 %3, %6, %15, %16 " RT_BCAST_LO_S0 "\n\t"
         "v_pk_fma_f32 %0, %5, %10, %0 " RT_BCAST_HI_S0 "\n\t"
         "v_pk_fma_f32 %1, %5, %14, %1 " RT_BCAST_HI_S0 "\n\t"
         "v_pk_fma_f32 %2, %7, %10, %2 " RT_BCAST_HI_S0 "\n\t"
         "v_pk_fma_f32 %3, %7, %14, %3 " RT_BCAST_HI_S0 "\n\t"
         "v_pk_fma_f32 %0, %5, %9, %0 " RT_BCAST_LO_S0 "\n\t"
         "v_pk_fma_f32 %1, %5, %13, %1 " RT_BCAST_LO_S0 "\n\t"
         "v_pk_fma_f32 %2, %7, %9, %2 " RT_BCAST_LO_S0 "\n\t"
         "v_pk_fma_f32 %3, %7, %13, %3 " RT_BCAST_LO_S0 "\n\t"
         "v_pk_fma_f32 %0, %0, %0, %8 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
         "v_pk_fma_f32 %1, %1, %1, %8 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
         "v_pk_fma_f32 %0, %6, %2, %0 " RT_BCAST_HI_S0 "\n\t"
         "v_pk_fma_f32 %1, %6, %3, %1 " RT_BCAST_HI_S0
         : "=&v"(d0), "=&v"(d1), "=&v"(t0), "=&v"(t1)
         : "v"(f.p1), "v"(f.p0), "v"(f.p3), "v"(f.p2), "v"(f.p4),             // %4 .. %8
           "v"(cx0), "v"(cy0), "v"(cz0), "v"(kp0),                            // %9 .. %12
           "v"(cx1), "v"(cy1), "v"(cz1), "v"(kp1));                           // %13 .. %16
}

// One group of kChunk spheres = kChunk float4 of the pair image (see scan_batch).
struct ChunkRegs {
  float4 q[kChunk];
};

template <bool kInPlace>
__device__ __forceinline__ void load_chunk(ChunkRegs &r, const float4 *lds_pairs, uint32_t first) {
  const float4 *src = lds_pairs + pair_slot<kInPlace>(first);   // (first is a multiple of kChunk: a group has no gap inside)
#pragma unroll
  for (int u = 0; u < kChunk; ++u) r.q[u] = src[u];
}

// filter the kChunk spheres held in r (scan positions first_index..+kChunk-1)
// and append the (lane, sphere) pairs that pass to the wave's list
__device__ __forceinline__ void filter_chunk(const ChunkRegs &r, uint32_t first_index, const FilterRayP &fr,
                                             WaveScan &w) {
  f32x2 d[kChunk / 2];
  filter_four_pairs(fr, r.q, d);
#if RT_EXP == 12
  {
    f32x2 d2[kChunk / 2];
    filter_four_pairs(fr, r.q, d2);
    asm volatile("" ::"v"(d2[0]), "v"(d2[1]), "v"(d2[2]), "v"(d2[3]));
  }
#endif
  // Eight compares into eight scalar masks, OR-ed on the scalar unit, one branch
  // for the whole chunk: no branch waits on a vector compare of its own.
  unsigned long long m[kChunk];
#pragma unroll
  for (int u = 0; u < kChunk; ++u) m[u] = __ballot(d[u >> 1][u & 1] > 0.0f);
  const unsigned long long any = ((m[0] | m[1]) | (m[2] | m[3])) | ((m[4] | m[5]) | (m[6] | m[7]));
  if (any != 0ull) {
    asm volatile("");  // (keeps the block a branch target; without it the nested ifs are merged)
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      if (m[u] != 0ull) {
        asm volatile("");
        // address = scalar (list base + 4 * count so far) + 4 * (passing lanes below this one),
        // value = lane field | scalar position: four vector operations per listed sphere
        const uint32_t rank =
            __builtin_amdgcn_mbcnt_hi((uint32_t)(m[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[u], 0u));
        uint32_t pos = first_index + u, tail = w.wlist_off + (w.wcnt << 2);
        asm("" : "+s"(pos), "+s"(tail));   // keep each one scalar operand
        lds_u32 *dst = (lds_u32 *)(uintptr_t)(tail + (rank << 2));
        if (d[u >> 1][u & 1] > 0.0f) *dst = w.lane16 | pos;
        w.wcnt += (uint32_t)__popcll(m[u]);
      }
    }
  }
}

// The sphere filter for the listed (lane, group) pairs, 64 pairs at a time whichever lanes they belong
// to: a ray needs two or three of a scene's groups, the 64 rays of a wave together three times as many,
// and filtering a group for the whole wave would spend two thirds of the lanes on rays that cannot touch
// it.  Each lane takes one pair, reads the owner's ray from the wave's table and rebuilds its filter
// coefficients (negations and doublings: exact), reads the group's eight spheres from the pair image
// (neighbouring list entries mostly name the same group, so most of these reads are broadcasts) and
// evaluates the same 32 packed FMAs as the full scan does, hence the same bits.  Spheres that pass go
// to the (lane, sphere) list under the owner's name.
template <bool kInPlace>
__device__ __forceinline__ void filter_listed_pairs(const DeviceScene &sc, const float4 *lds_pairs, const uint8_t *lds_real8,
                                                    uint32_t base, WaveScan &w, HitState &h) {
  RT_HSTAGE_BEGIN(38);
  RT_HSTAGE_EXCLUDE_BEGIN(37);
  uint32_t width = 64u;   // pairs per pass
  for (uint32_t e0 = 0; e0 < w.pcnt;) {
    const uint32_t e = e0 + w.lane;
    const bool valid = w.lane < width && e < w.pcnt;
    const uint32_t entry = !valid ? 0u
                           : kInPlace ? *(lds_u32 *)(uintptr_t)(w.plist_off + (e << 2))
                                      : (uint32_t)*(lds_u16 *)(uintptr_t)(w.plist_off + (e << 1));
    const uint32_t owner = entry & 63u, grp = entry >> 6;
    rt_v2f r0, r1, r2, r3, r5;
    ray_table_read5<0, 1, 2, 3, 5>(w.rays_off + (owner << 3), r0, r1, r2, r3, r5);
#if RT_EXP == 55
    if (valid && !kInPlace) lds_again_read_u16(w.plist_off + (e << 1));
#endif
#if RT_EXP == 51
    {
      rt_v2f x0, x1, x2, x3, x5;
      ray_table_read5<0, 1, 2, 3, 5>(w.rays_off + (owner << 3), x0, x1, x2, x3, x5);
    }
#endif
    FilterRayP f;
    f.p0 = f32x2{-r2[0], -r2[1]};
    f.p1 = f32x2{-r3[0], r5[0]};
    f.p2 = f32x2{-2.0f * r0[0], -2.0f * r0[1]};
    f.p3 = f32x2{-2.0f * r1[0], r3[1]};
    f.p4 = f32x2{r5[1], 0.0f};
    // A group's real spheres come first.  The second half of the groups is skipped when no pair of the
    // pass names a group with more than four -- as in the pass over the group of the few large
    // spheres, which every ray lists and which therefore fills a pass of its own.
    const uint32_t real = valid ? (uint32_t)lds_real8[grp] : 0u;
#if RT_EXP == 55
    if (valid && sc.n_pad <= kMaxLdsSpheres) lds_again_read_u8(lds_address(&lds_real8[grp]));
#endif
    const bool upper_half = __ballot(real > (uint32_t)(kChunk / 2)) != 0ull;
    f32x2 d[kChunk / 2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
    // first word of this lane's entries: owner | scan position of the group's first sphere
    const uint32_t word0 = (owner << 16) | (base + grp * kChunk);
    const unsigned long long valid_mask = __ballot(valid);   // (a scalar AND per sphere instead of a vector select)
#if RT_EXP == 39
    h.fill[0] += (uint32_t)__popcll(valid_mask);
    h.fill[1] += 1u;
#endif
    // Eight compares into eight scalar masks, OR-ed on the scalar unit, one branch
    // for the whole pass: no branch waits on a vector compare of its own.
    unsigned long long m[kChunk] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half == 0 || upper_half) {
        const float4 *src = lds_pairs + pair_slot<kInPlace>(grp * kChunk) + 4 * half;
        const float4 q[4] = {src[0], src[1], src[2], src[3]};
#if RT_EXP == 52
        if (sc.n_pad <= kMaxLdsSpheres)
          for (int j = 0; j < 4; ++j) lds_again_read_b128(lds_address(&src[j]));
#endif
        filter_two_pairs(f, q, d[2 * half], d[2 * half + 1]);
#pragma unroll
        for (int u = 4 * half; u < 4 * half + 4; ++u) m[u] = __ballot(d[u >> 1][u & 1] > 0.0f) & valid_mask;
      }
    }
    const unsigned long long any = ((m[0] | m[1]) | (m[2] | m[3])) | ((m[4] | m[5]) | (m[6] | m[7]));
    if (any != 0ull) {
      asm volatile("");  // (keeps the block a branch target; without it the nested ifs are merged)
      // A bounce usually lists about 120 spheres in all.  Should this pass's not fit, the list is
      // drained first; should they be more than the whole list, the pass is done again for 32 pairs,
      // which cannot list more than 32 * kChunk.
      uint32_t total = 0;
#pragma unroll
      for (int u = 0; u < kChunk; ++u) total += (uint32_t)__popcll(m[u]);
      if (w.wcnt + total > kWaveListCapCull) {
        if (total > kWaveListCapCull) {
          width = 32u;
          continue;
        }
        drain_wave(sc, w, h);
      }
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        if (m[u] != 0ull) {
          asm volatile("");
          // address = scalar (list base + 4 * count so far) + 4 * (passing lanes below this one):
          // four vector operations per listed sphere
          const uint32_t rank =
              __builtin_amdgcn_mbcnt_hi((uint32_t)(m[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[u], 0u));
          uint32_t tail = w.wlist_off + (w.wcnt << 2);
          asm("" : "+s"(tail));   // keep it one scalar operand
          lds_u32 *dst = (lds_u32 *)(uintptr_t)(tail + (rank << 2));
          if (d[u >> 1][u & 1] > 0.0f && valid) *dst = word0 + (uint32_t)u;
#if RT_EXP == 57
          if (d[u >> 1][u & 1] > 0.0f && valid) lds_again_write_b32(tail + (rank << 2), word0 + (uint32_t)u);
#endif
          w.wcnt += (uint32_t)__popcll(m[u]);
        }
      }
    }
    h.lane_tests += real;
    e0 += width;
    width = 64u;
  }
  w.pcnt = 0;
  RT_HSTAGE_EXCLUDE_END(37);
  RT_HSTAGE_END(38);
}

// Scan `count` spheres (a multiple of kDrainEvery = two groups) whose pair
// image is in LDS.  The LDS image holds sphere PAIRS: 32 bytes {cx0,cx1,
// cy0,cy1, cz0,cz1, K'0,K'1}, read as wave-wide broadcast ds_read_b128; one
// packed FMA then works on both spheres of a pair.
//
// kCull = false: every group is filtered; the reads run one chunk ahead of the
// arithmetic in a second register set, so their latency is covered by the 32
// packed FMAs of the chunk in hand rather than waited for.
//
// kCull = true: the boxes of the super-groups and, inside a super-group some
// lane's ray can touch, of the groups are tested against every lane's ray; a
// lane whose ray can touch a group puts (lane, group) on the wave's list, and
// filter_listed_pairs does the rest.  `uncullable` lanes (rays outside the
// range the boxes are valid for) list every group; lanes without a ray none.
template <bool kCull, bool kCut = false, bool kInPlace = false>
__device__ __forceinline__ void scan_batch(const DeviceScene &sc, const ScanLds &lds, uint32_t base, uint32_t count,
                                           const FilterRayP &fr, const BoxRayP &br, bool alive, bool uncullable,
                                           WaveScan &w, HitState &h, float cut = 0.0f, bool first_direct = false) {
  const float4 *lds_pairs = lds.pairs, *lds_boxes = lds.boxes, *lds_sboxes = lds.sboxes;
  static_assert(kDrainEvery == 2 * kChunk, "two register sets per round");
  if (!kCull) {
    ChunkRegs ra, rb;
    load_chunk<kInPlace>(ra, lds_pairs, 0);
    for (uint32_t i0 = 0; i0 < count; i0 += kDrainEvery) {
      load_chunk<kInPlace>(rb, lds_pairs, i0 + kChunk);
      if (w.wcnt > kWaveListFlushFull) drain_wave(sc, w, h);   // rare: the list usually lasts the whole scan
      filter_chunk(ra, base + i0, fr, w);
      // the read past the end of the last round stays inside the image and is never used
      load_chunk<kInPlace>(ra, lds_pairs, i0 + kDrainEvery < count ? i0 + kDrainEvery : 0u);
      if (w.wcnt > kWaveListFlushFull) drain_wave(sc, w, h);
      filter_chunk(rb, base + i0 + kChunk, fr, w);
    }
    return;
  }
  const unsigned long long all_mask = __ballot(uncullable), live_mask = __ballot(alive);
  uint32_t tests = 0;
  // one group the box test (or nothing) lets some lanes through to: list (lane, group) for those lanes
  auto list_group = [&](unsigned long long m, bool mine, uint32_t grp) {
    if (m != 0ull) {
      asm volatile("");
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (kInPlace) {
        // (the list outlives the batch: the group is counted from the start of the scan order)
        uint32_t tail = w.plist_off + (w.pcnt << 2), field = (base / (uint32_t)kChunk + grp) << 6;
        asm("" : "+s"(tail), "+s"(field));   // keep each one scalar operand
        lds_u32 *dst = (lds_u32 *)(uintptr_t)(tail + (rank << 2));
        if (mine) *dst = field | w.lane;
      } else {
        uint32_t tail = w.plist_off + (w.pcnt << 1), field = grp << 6;
        asm("" : "+s"(tail), "+s"(field));   // keep each one scalar operand
        lds_u16 *dst = (lds_u16 *)(uintptr_t)(tail + (rank << 1));
        if (mine) *dst = (uint16_t)(field | w.lane);
#if RT_EXP == 56
        if (mine) lds_again_write_b16(tail + (rank << 1), field | w.lane);
#endif
      }
      w.pcnt += (uint32_t)__popcll(m);
    }
  };
  // The groups every ray scans (the scene's few large spheres) are filtered here, by every lane for its own
  // ray from the coefficients it holds in registers, the spheres read as wave-wide broadcasts: no list entry,
  // no reading back of the ray.  (The lanes that list every group anyway get them from the rounds below.)
  // Four spheres at a time, so that what one step appends (at most 64 * 4 entries) fits the list.
  static_assert(kWaveListCapCull >= 64u * (kChunk / 2), "half a group's candidates must fit the list");
  if (base == 0u) {
    const unsigned long long own_mask = live_mask & ~all_mask;
    const bool own = alive && !uncullable;
    for (uint32_t g = 0; g < sc.n_always_groups; ++g) {
      const uint32_t real = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)lds.real8[g]);
#pragma unroll 1
      for (uint32_t half = 0; half < 2u; ++half) {
        if (half == 1u && real <= (uint32_t)(kChunk / 2)) break;
        if (w.wcnt + 64u * (kChunk / 2) > kWaveListCapCull) drain_wave(sc, w, h);
#if RT_EXP == 39
        h.fill[4] += (uint32_t)__popcll(own_mask);
        h.fill[5] += 1u;
#endif
        const float4 *src = lds_pairs + pair_slot<kInPlace>(g * kChunk) + 4u * half;
        const float4 q[4] = {src[0], src[1], src[2], src[3]};
        f32x2 d[2];
        filter_two_pairs(fr, q, d[0], d[1]);
        unsigned long long m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) m[u] = __ballot(d[u >> 1][u & 1] > 0.0f) & own_mask;
        // (closest_hit_scene has evaluated the sphere at scan position 0 for every lane's own ray already)
        if (first_direct && g == 0u && half == 0u) m[0] = 0ull;
        if (((m[0] | m[1]) | (m[2] | m[3])) != 0ull) {
          asm volatile("");
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (m[u] != 0ull) {
              asm volatile("");
              const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[u], 0u));
              uint32_t pos = g * kChunk + 4u * half + (uint32_t)u, tail = w.wlist_off + (w.wcnt << 2);
              asm("" : "+s"(pos), "+s"(tail));   // keep each one scalar operand
              lds_u32 *dst = (lds_u32 *)(uintptr_t)(tail + (rank << 2));
              if (d[u >> 1][u & 1] > 0.0f && own) *dst = w.lane16 | pos;
#if RT_EXP == 57
              if (d[u >> 1][u & 1] > 0.0f && own) lds_again_write_b32(tail + (rank << 2), w.lane16 | pos);
#endif
              w.wcnt += (uint32_t)__popcll(m[u]);
            }
          }
        }
      }
      h.lane_tests += own ? real : 0u;
    }
  }
  // Super-groups first, all of them in one tight loop (the boxes of a pair of super-groups, 64 spheres,
  // in one packed evaluation; two pairs per trip so that the reads of one overlap the arithmetic of the
  // other and no register is copied): a bit per super-group some lane's ray can touch.  A batch has at
  // most 2048 / 32 = 64 super-groups.
  constexpr uint32_t kPairSpan = 4u * kDrainEvery;   // spheres per pair of super-groups
  unsigned long long touched = 0ull;
  RT_HSTAGE_BEGIN(36);
  {
    auto test_pair = [&](const float4 (&sq)[3], uint32_t pair_index) {
      bool spass[2];
      box_pair<kCut>(br, sq, spass, cut);
      const unsigned long long s0 = (__ballot(spass[0]) & live_mask) | all_mask, s1 = (__ballot(spass[1]) & live_mask) | all_mask;
      // (two scalar selects; written as (s0 != 0 ? 1 : 0) the first one goes through a vector register)
      const unsigned long long bit = 1ull << (2u * pair_index);
      touched |= (s0 != 0ull ? bit : 0ull) | (s1 != 0ull ? bit << 1 : 0ull);
    };
    const uint32_t n_pairs = count / kPairSpan;
    if (n_pairs > 4u) {
      // more than four pairs of super-groups: the quarters first (four super-groups each, two boxes per packed
      // evaluation), then the two pairs of super-groups of every quarter some ray of the wave can touch
      const float4 *lds_qboxes = lds.qboxes;
      const uint32_t n_qpairs = quarter_pairs(count);
      uint32_t quarters = 0u;
      for (uint32_t qp = 0; qp < n_qpairs; ++qp) {
        const float4 qq[3] = {lds_qboxes[3u * qp], lds_qboxes[3u * qp + 1u], lds_qboxes[3u * qp + 2u]};
        bool qpass[2];
        box_pair<kCut>(br, qq, qpass, cut);
        const unsigned long long s0 = (__ballot(qpass[0]) & live_mask) | all_mask, s1 = (__ballot(qpass[1]) & live_mask) | all_mask;
        quarters |= ((s0 != 0ull ? 1u : 0u) | (s1 != 0ull ? 2u : 0u)) << (2u * qp);
      }
      tests += 2u * n_qpairs;
      quarters &= (1u << ((n_pairs + 1u) / 2u)) - 1u;   // (rays outside the boxes' range set every bit: not the padding's)
      while (quarters != 0u) {
        const uint32_t p0 = 2u * (uint32_t)__builtin_ctz(quarters);
        quarters &= quarters - 1u;
        const bool two = p0 + 1u < n_pairs;
        const uint32_t i0 = 3u * p0, i1 = 3u * (two ? p0 + 1u : p0);
        const float4 sa[3] = {lds_sboxes[i0], lds_sboxes[i0 + 1u], lds_sboxes[i0 + 2u]};
        const float4 sb[3] = {lds_sboxes[i1], lds_sboxes[i1 + 1u], lds_sboxes[i1 + 2u]};
        test_pair(sa, p0);
        if (two) test_pair(sb, p0 + 1u);
        tests += two ? 4u : 2u;
      }
    } else {
      float4 sa[3] = {lds_sboxes[0], lds_sboxes[1], lds_sboxes[2]}, sb[3];
      for (uint32_t pi = 0; pi < n_pairs; pi += 2u) {
        const uint32_t ib = 3u * (pi + 1u < n_pairs ? pi + 1u : 0u);   // (a read past the end is redirected and not used)
        sb[0] = lds_sboxes[ib];
        sb[1] = lds_sboxes[ib + 1];
        sb[2] = lds_sboxes[ib + 2];
        test_pair(sa, pi);
        const uint32_t ia = 3u * (pi + 2u < n_pairs ? pi + 2u : 0u);
        sa[0] = lds_sboxes[ia];
        sa[1] = lds_sboxes[ia + 1];
        sa[2] = lds_sboxes[ia + 2];
        if (pi + 1u < n_pairs) test_pair(sb, pi + 1u);
      }
      tests += 2u * n_pairs;
    }
  }
  RT_HSTAGE_END(36);
  RT_HSTAGE_BEGIN(37);
  // then the two rounds (four group boxes) of every super-group that was touched, the boxes of a round
  // read while the round before it is worked on (two register sets, no copies)
  if (touched != 0ull) {
    auto load_round = [&](float4 (&q)[3], uint32_t first) {
      q[0] = lds_boxes[first];
      q[1] = lds_boxes[first + 1u];
      q[2] = lds_boxes[first + 2u];
    };
    // one round: the two boxes in q against every lane's ray.  Should the pair list have to be filtered
    // first, the boxes read ahead (`ahead`, from entry `ahead_first`) are read again afterwards rather
    // than kept in registers across the filter pass.
    auto round2 = [&](const float4 (&q)[3], uint32_t i0, float4 (&ahead)[3], uint32_t ahead_first) {
      bool pass[2];
      box_pair<kCut>(br, q, pass, cut);
#if RT_EXP == 13
      {
        bool pass2[2];
        box_pair(br, q, pass2);
        asm volatile("" ::"s"(__ballot(pass2[0])), "s"(__ballot(pass2[1])));
      }
#endif
      const bool mine0 = (pass[0] && alive) || uncullable, mine1 = (pass[1] && alive) || uncullable;
      const unsigned long long m0 = (__ballot(pass[0]) & live_mask) | all_mask, m1 = (__ballot(pass[1]) & live_mask) | all_mask;
      tests += 2u;
      h.own_groups += (mine0 ? 1u : 0u) + (mine1 ? 1u : 0u);
      h.wave_groups += (m0 != 0ull ? 1u : 0u) + (m1 != 0ull ? 1u : 0u);
      if (w.pcnt > kPairListFlush) {
        filter_listed_pairs<kInPlace>(sc, lds_pairs, lds.real8, kInPlace ? 0u : base, w, h);
        load_round(ahead, ahead_first);
      }
      list_group(m0, mine0, i0 >> 3);
      list_group(m1, mine1, (i0 >> 3) + 1u);
    };
    uint32_t sg = (uint32_t)__builtin_ctzll(touched);
    touched &= touched - 1ull;
    float4 qa[3], qb[3];
    load_round(qa, 6u * sg);   // (a super-group's two rounds are six consecutive entries)
    for (;;) {
      const uint32_t j0 = sg * (2u * kDrainEvery);
      load_round(qb, 6u * sg + 3u);
      round2(qa, j0, qb, 6u * sg + 3u);
      const bool more = touched != 0ull;
      const uint32_t next_sg = more ? (uint32_t)__builtin_ctzll(touched) : sg;   // (at the end: a read for nothing)
      touched &= touched - 1ull;
      load_round(qa, 6u * next_sg);
      round2(qb, j0 + kDrainEvery, qa, 6u * next_sg);
      if (!more) break;
      sg = next_sg;
    }
  }
  RT_HSTAGE_END(37);
  // (a scene in LDS: the pairs name groups of this one batch.  Scanned in place, the list goes on to the next batch
  // and the caller decides when it is worth a pass: closest_hit_scene)
  if (!kInPlace) filter_listed_pairs<false>(sc, lds_pairs, lds.real8, base, w, h);
  h.box_tests += tests;
}

// Closest hit of the ray (A, B) of every lane of the workgroup against the
// whole scene, within (tmin, tmax).  Unbatched: the scene is already resident
// in LDS.  Batched: it is scanned in place from global memory in pieces of
// kBatchSpheres scan positions, every wave on its own (nothing is staged and
// there is no barrier).  Lanes without a ray pass dead_ray().
template <bool kBatched, bool kCull, bool kProfile = false>
__device__ __forceinline__ void closest_hit_scene(const DeviceScene &sc, const ScanLds &lds, uint32_t tid, bool alive,
                                                  V3 A, V3 B, float tmin, float tmax, HitState &h) {
  RT_HSTAGE_BEGIN(35);
  const FilterRay f = alive ? make_filter_ray(A, B) : dead_ray();
  const FilterRayP fr = pack_ray(f);   // (the culled scan rebuilds it from the wave's ray table instead)
  const BoxRayP br = pack_box_ray(make_box_ray(A, B, sc.sigma0, sc.sigma1));
  const float a = dot(B, B);
  // rays outside the range the group boxes were derived for scan every group
  const bool uncullable = kCull && alive && !box_ray_in_range(A, B);
  const uint32_t wave_first = tid & ~63u;
  const bool first_direct = kCull && sc.n_always_groups != 0u;
  static_assert(kRayRows * 64u * sizeof(float2) == 64u * 48u, "48 bytes of ray table per lane (scan_lds_bytes)");
  WaveScan w;
  w.wlist_off = (uint32_t)__builtin_amdgcn_readfirstlane(
      (int)(uint32_t)(uintptr_t)(lds_u32 *)(lds.wlist + (tid >> 6) * wave_list_cap(kCull)));
  w.plist_off = (uint32_t)__builtin_amdgcn_readfirstlane(
      (int)(uint32_t)(uintptr_t)(lds_u16 *)(lds.plist + (tid >> 6) * kPairListCap * (kBatched ? 2u : 1u)));
  w.rays_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u32 *)(lds.rays + kRayRows * wave_first));
  w.best = lds.best + wave_first;
  w.wcnt = 0;
  w.pcnt = 0;
  w.lane = tid & 63u;
  w.lane16 = w.lane << 16;
  w.tmin = tmin;
  // this lane's ray and "no hit yet" key where the other lanes of the wave can reach them; the
  // key {tmax bits, 0} can only be undercut by t < tmax
  const double inv_a = reciprocal_of_widened_float((double)a);   // one reciprocal per ray, no division per sphere
  {
    float2 *mine = lds.rays + kRayRows * wave_first + (tid & 63u);
    mine[0] = make_float2(A.x, A.y);
    mine[64] = make_float2(A.z, a);
    mine[128] = make_float2(B.x, B.y);
    mine[192] = make_float2(B.z, f.na);
    mine[256] = make_float2(__int_as_float(__double2loint(inv_a)), __int_as_float(__double2hiint(inv_a)));
    mine[320] = make_float2(f.ab, f.naaa);
#if RT_EXP == 58
    auto again = [&](uint32_t row, float lo, float hi) {
      lds_again_write_b64(lds_address(mine + 64u * row), ((unsigned long long)__float_as_uint(hi) << 32) | __float_as_uint(lo));
    };
    again(0, A.x, A.y);
    again(1, A.z, a);
    again(2, B.x, B.y);
    again(3, B.z, f.na);
    again(4, __int_as_float(__double2loint(inv_a)), __int_as_float(__double2hiint(inv_a)));
    again(5, f.ab, f.naaa);
#endif
  }
  {
    // (made here on every call: as a loop invariant the 64-bit constant would sit in two vector
    // registers for the whole kernel)
    uint32_t no_hit_hi = __float_as_uint(tmax), no_hit_lo = 0u;
    asm volatile("" : "+v"(no_hit_hi), "+v"(no_hit_lo));
    unsigned long long first_key = ((unsigned long long)no_hit_hi << 32) | no_hit_lo;
    if (first_direct) {
      // The sphere at scan position 0 -- the largest of the always-scanned ones, the ground of scenes like the
      // reference's -- is a candidate of nearly every ray (half of all exact evaluations of the final scene).  Every
      // lane evaluates it for its own ray here, from registers: no list entry, no reading back of the ray, no
      // atomic, and the passes over the list run with what is left.
      const float4 ex0 = sc.exact[0];
      float t0;
      if (alive && exact_hit(A, B, a, inv_a, ex0.x, ex0.y, ex0.z, ex0.w, tmin, kFltMax, t0)) {
        const unsigned long long key0 = ((unsigned long long)__float_as_uint(t0) << 32) | sc.orig[0];
        if (key0 < first_key) first_key = key0;
      }
    }
    lds.best[tid] = first_key;
#if RT_EXP == 58
    lds_again_write_b64(lds_address(&lds.best[tid]), first_key);
#endif
  }
  RT_HSTAGE_END(35);
  if (kBatched) {
    // A scene too large for LDS is scanned where it lies: its pair image, boxes and group sizes are read from
    // global memory (the 16 bytes per sphere of even a 65,535-sphere scene stay in the L2 of every XCD), batch
    // by batch -- a batch is kBatchSpheres scan positions with a box of its own, kRegionBatches batches a region with a
    // box above those.  Which batches: those whose box some ray of the WAVE can touch, found by every lane
    // testing its ray against the region boxes and then against the batch boxes of the regions that are
    // touched (two boxes per packed evaluation, read through the scalar cache).  Nothing is shared between the
    // waves of the workgroup, so there is no barrier in the scan and a wave visits only what its own 64 rays
    // need.  (Staging each batch into LDS for the whole workgroup, two barriers per batch, left the vector
    // ALUs idle 60 % of the time.)
    // (batches of 1024 and more: the largest scene has 64 of them, and the second mask and key are never used)
    constexpr bool kTwoMasks = (kMaxSpheres + kBatchSpheres) / kBatchSpheres > 64u;
    unsigned long long want_lo = ~0ull, want_hi = ~0ull;   // (without culling: every batch)
    typedef __attribute__((address_space(4))) const float *ScalarFloats;
    const ScalarFloats bb = (ScalarFloats)(uintptr_t)(sc.sboxes + 3u * (sc.n_pad / 64u));
    auto load_pair = [&](uint32_t q) {
      const ScalarFloats e = bb + 12u * q;
      return BoxPairS{f32x2{e[0], e[1]}, f32x2{e[2], e[3]}, f32x2{e[4], e[5]}, f32x2{e[6], e[7]}, f32x2{e[8], e[9]}, f32x2{e[10], e[11]}};
    };
    const unsigned long long all_mask = __ballot(uncullable), live_mask = __ballot(alive);
    RT_HSTAGE_BEGIN(40);
    if (kCull) {
      const uint32_t n_batch_pairs = batch_pairs(sc.n_pad);
      unsigned long long mine_lo = 0ull, mine_hi = 0ull;
      uint32_t tested = 0;
      if (all_mask == 0ull) {
        // regions first (kRegionBatches batches each, two boxes per packed evaluation), then the batch boxes of the regions
        // some ray of the wave can touch; the boxes come through the scalar cache
        const uint32_t n_region_pairs = region_pairs(sc.n_pad);
        uint32_t regions = 0;
        for (uint32_t q = 0; q < n_region_pairs; ++q) {
          bool pass[2];
          box_pair_scalar(br, load_pair(n_batch_pairs + q), pass);
          const unsigned long long m0 = __ballot(pass[0]) & live_mask, m1 = __ballot(pass[1]) & live_mask;
          regions |= ((m0 != 0ull ? 1u : 0u) | (m1 != 0ull ? 2u : 0u)) << (2u * q);
        }
        tested = 2u * n_region_pairs;
        while (regions != 0u) {
          const uint32_t r = (uint32_t)__builtin_ctz(regions);
          regions &= regions - 1u;
          const uint32_t q0 = r * (kRegionBatches / 2u), q1 = min(q0 + kRegionBatches / 2u, n_batch_pairs);
          BoxPairS cur = load_pair(q0);
          for (uint32_t q = q0; q < q1; ++q) {
            const BoxPairS nxt = load_pair(q + 1u < q1 ? q + 1u : q0);
            bool pass[2];
            box_pair_scalar(br, cur, pass);
            const unsigned long long m0 = __ballot(pass[0]) & live_mask, m1 = __ballot(pass[1]) & live_mask;
#if RT_EXP == 39
            h.fill[8] += (uint32_t)(__popcll(m0) + __popcll(m1));   // (lane, batch) pairs wanted
            h.fill[9] += (m0 != 0ull ? 1u : 0u) + (m1 != 0ull ? 1u : 0u);   // batches some lane wants
#endif
            const unsigned long long bits = (unsigned long long)((m0 != 0ull ? 1u : 0u) | (m1 != 0ull ? 2u : 0u));
            if (!kTwoMasks || q < 32u) mine_lo |= bits << (2u * q);
            else mine_hi |= bits << (2u * q - 64u);
            cur = nxt;
          }
          tested += 2u * (q1 - q0);
        }
      }
      if (all_mask != 0ull) mine_lo = mine_hi = ~0ull;
      h.box_tests += tested;
      want_lo = mine_lo;
      want_hi = mine_hi;
    }
    static_assert((kMaxSpheres + kBatchSpheres) / kBatchSpheres <= 128u, "two 64-bit masks");
    auto visit = [&](uint32_t base, float cut) {
      const uint32_t count = min(kBatchSpheres, sc.n_pad - base);
      // (without culling the trailing padding is not visited)
      const uint32_t span = kCull ? count : (sc.n_full > base ? min(count, sc.n_full - base) : 0u);
      ScanLds in_place = lds;   // (the per-wave lists, rays and keys stay in LDS; the scene's arrays are the global ones)
      // (the culled scan's pair list names groups counted from the start of the scan order, so its pair image and
      // group sizes are the whole arrays; the scan without culling walks the batch's own part)
      in_place.pairs = const_cast<float4 *>(kCull ? sc.filter : sc.filter + base);
      in_place.boxes = const_cast<float4 *>(sc.boxes + 3u * (base / 16u));
      in_place.sboxes = const_cast<float4 *>(sc.sboxes + 3u * (base / 64u));
      in_place.qboxes = const_cast<float4 *>(sc.qboxes + 3u * (base / 256u));
      in_place.real8 = const_cast<uint8_t *>(kCull ? sc.group_real : sc.group_real + base / 8u);
      scan_batch<kCull, kCull, true>(sc, in_place, base, span, fr, br, alive, uncullable, w, h, cut, first_direct);
    };
    if (kCull && live_mask != 0ull) {
      // Near batches first, and every hit found narrows what is still worth a visit (rtcore: distance cut).
      // Lane j holds the squared distance from the wave's first ray origin to the box of batch j (and of
      // batch j + 64): the rays of a wave start close together, so this is near-to-far along each of them
      // whichever way it points.  The wave takes the smallest key, tests that batch's box again -- now
      // against each ray's cut -- scans it if some ray still wants it, evaluates what the scan listed,
      // and moves each ray's cut up to its hit.  Batch 0 (the always-scanned large spheres live there:
      // the ground ends most rays) goes first.
      const uint32_t n_batches = 2u * batch_pairs(sc.n_pad);
      const int rep = (int)__builtin_ctzll(live_mask);
      const float ox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A.x), rep));
      const float oy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A.y), rep));
      const float oz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A.z), rep));
      const float *boxes_flat = reinterpret_cast<const float *>(sc.sboxes + 3u * (sc.n_pad / 64u));
      auto key_of = [&](uint32_t j, bool wanted) {
        float key = __builtin_inff();
        if (wanted && j < n_batches) {
          const float *e = boxes_flat + 12u * (j >> 1) + (j & 1u);
          const float dx = __builtin_fmaxf(__builtin_fabsf(ox - e[0]) - e[6], 0.0f);
          const float dy = __builtin_fmaxf(__builtin_fabsf(oy - e[2]) - e[8], 0.0f);
          const float dz = __builtin_fmaxf(__builtin_fabsf(oz - e[4]) - e[10], 0.0f);
          key = dx * dx + dy * dy + dz * dz;
          if (!(key < 1.0e30f)) key = 1.0e30f;   // (an always-scanned batch has no box: last, but visited)
        }
        return key;
      };
      float key0 = key_of(w.lane, ((want_lo >> w.lane) & 1ull) != 0ull);
      float key1 = (kTwoMasks && n_batches > 64u) ? key_of(w.lane + 64u, ((want_hi >> w.lane) & 1ull) != 0ull) : __builtin_inff();
      if (w.lane == 0u && (want_lo & 1ull)) key0 = -1.0f;
      // what a hit distance is raised by before it rules a box out; a ray outside the boxes' range is never cut
      const float slack = cut_slack(sc.cut_reach, sc.cut_b, dot(A, A), __builtin_amdgcn_rsqf(a) * 1.001f);
      float cut = __builtin_inff();
      bool cut_moved = false;   // some lane's cut is below infinity (wave-uniform)
      for (;;) {
        const float k = kTwoMasks ? __builtin_fminf(key0, key1) : key0;
        const float kmin = wave_min(k);
        if (!(kmin < __builtin_inff())) break;
        const int j = (int)__builtin_ctzll(__ballot(k == kmin));
        const bool second = kTwoMasks && __builtin_amdgcn_readlane((key0 == kmin) ? 0 : 1, j) != 0;
        if ((int)w.lane == j) {
          if (second) key1 = __builtin_inff();
          else key0 = __builtin_inff();
        }
#if RT_EXP == 39
        h.fill[7] += 1u;
#endif
        const uint32_t bi = (uint32_t)j + (second ? 64u : 0u);
        const uint32_t base = bi * kBatchSpheres;
        if (base >= sc.n_pad) continue;
        // (until a hit has moved some lane's cut, a batch that was wanted is wanted still)
        if (all_mask == 0ull && cut_moved) {
          bool pass[2];
          float near_bound[2];
          box_pair_scalar(br, load_pair(bi >> 1), pass, near_bound);
          const bool still = (bi & 1u) ? (pass[1] && !(near_bound[1] > cut)) : (pass[0] && !(near_bound[0] > cut));
          h.box_tests += 2u;
          if ((__ballot(still) & live_mask) == 0ull) continue;
        }
#if RT_EXP == 39
        h.fill[6] += 1u;
#endif
        RT_HSTAGE_EXCLUDE_BEGIN(40);
        visit(base, cut);
        // What the visit listed is filtered at once after batch 0 (the ground sphere of scenes like the reference's
        // lives there and its hit cuts most of what follows), afterwards only when a pass would run with enough of its
        // lanes at work; the rest waits for the next batch's, or for the end of the scan.  Candidates are evaluated
        // as soon as there are kMergeCandidates of them, and every evaluation moves the cuts up.
        const bool at_once = bi == 0u;
        if (w.pcnt != 0u && (at_once || w.pcnt >= kMergePairs)) filter_listed_pairs<true>(sc, sc.filter, sc.group_real, 0u, w, h);
        if (w.wcnt != 0u && (at_once || w.wcnt >= kMergeCandidates)) {
          drain_wave(sc, w, h);
          const float t_best = __uint_as_float((uint32_t)(lds.best[tid] >> 32));
          cut = rt_fma(t_best, kCutRel, slack);
          cut_moved = true;
        }
        RT_HSTAGE_EXCLUDE_END(40);
      }
      RT_HSTAGE_END(40);
      if (w.pcnt != 0u) filter_listed_pairs<true>(sc, sc.filter, sc.group_real, 0u, w, h);
    } else if (!kCull) {
#pragma unroll 1
      for (uint32_t half = 0; half < 2u; ++half) {
        unsigned long long want = half ? want_hi : want_lo;
        while (want != 0ull) {
          const uint32_t base = (half * 64u + (uint32_t)__builtin_ctzll(want)) * kBatchSpheres;
          want &= want - 1ull;
          if (base >= sc.n_pad) break;   // (a mask with every bit set names batches that do not exist)
          visit(base, 0.0f);
        }
      }
    }
  } else {
    scan_batch<kCull>(sc, lds, 0u, kCull ? sc.n_pad : sc.n_full, fr, br, alive, uncullable, w, h, 0.0f, first_direct);
  }
  if (!kCull) h.scan_tests += sc.n;
  const unsigned long long d0 = (kProfile && !RT_SCAN_STAGE) ? tick() : 0ull;
  drain_wave(sc, w, h);
  if (kProfile && !RT_SCAN_STAGE) h.drain_cycles = tick() - d0;
  const unsigned long long key = lds.best[tid];
  const float t = __uint_as_float((uint32_t)(key >> 32));
  if (alive && t < tmax) {
    h.closest = t;
    h.index = (int)(uint32_t)key;
  }
}

// global row of compact (tile-major) local row lr.  magic = ceil(2^32 / tile_rows) when the host found
// the multiply-high quotient exact for every row of the job (rows * tile_rows < 2^32, tile_rows > 1),
// else 0 and the division is done as such.
__device__ __forceinline__ uint32_t global_row(const rt_job &job, uint32_t lr, uint32_t magic = 0u) {
  uint32_t t = magic ? __umulhi(lr, magic) : lr / job.tile_rows;
  return job.row0 + t * job.tile_stride + (lr - t * job.tile_rows);
}

// ---- work decomposition of the trace kernel
//
// A work item is one camera path.  The job's compact pixel rows are cut into
// 8x8 pixel tiles; item id = ((tile * SB) + s) * 64 + l is sample s (of this
// launch's batch of SB samples) of pixel l of the tile (l & 7 across, l >> 3
// down).  Waves draw chunks of kChunkItems consecutive ids from one global
// counter and hand the ids of a chunk to whichever lanes are idle, so a wave
// starts on 64 neighbouring pixels and stays full until the queue is empty.
// Every path's radiance goes to rad[s][pixel] (with its first-hit index in
// the fourth word); resolve_kernel then adds each pixel's samples in sample
// order, which is the reference's summation order (c/main.c:462-469).
// Chunk sizes (multiples of 64).  Large chunks keep a wave on one tile for many samples (coherent rays: 1024 ids
// are 16 samples of a tile) and cost fewer queue operations, but the launch ends later: the last chunks handed
// out are still being worked on when most waves have left.  A launch that another one follows on the other
// stream (the next sample batch of the frame, or the next frame of a caller that queues them back to back) has that
// tail covered and takes the large size (1.21e10 against 1.185e10 paths/s on the reference's final scene); a
// launch on its own takes the small one (11.56 ms against 11.97).  Handing the last ids out in still smaller
// pieces, which an earlier build did, costs more than it evens out now that rows of tiles are taken bottom-up
// (the last paths are the sky's, the shortest).
#ifndef RT_CHUNK_ITEMS
#define RT_CHUNK_ITEMS 1024
#endif
#ifndef RT_CHUNK_ALONE
#define RT_CHUNK_ALONE 512
#endif
constexpr uint32_t kChunkItems = RT_CHUNK_ITEMS, kChunkAlone = RT_CHUNK_ALONE;

// image and queue geometry of a launch: what handing out new paths needs
struct PathGeom {
  rt_job job;
  uint32_t rows;          // compact rows of the job
  uint32_t tiles_x;       // ceil(nx / 8)
  uint32_t sample0;       // first sample of this batch
  uint32_t batch;         // SB: samples in this batch
  uint32_t total_items;   // tiles * SB * 64
  uint32_t last_tile_row; // ceil(rows / 8) - 1
  uint32_t chunk;         // ids a wave takes from the queue at a time
  uint32_t tile_rows_magic;  // see global_row
  double inv_nx, inv_ny;  // 1.0 / (double)(float)nx, ny (pixel_xy_by)
};


struct TraceArgs {
  Camera cam;
  PathGeom geo;
  DeviceScene sc;
  float4 *rad;            // [SB][rows*nx] {r, g, b, first-hit index bits}, or, when rad_words is 3, {r, g, b} packed
  unsigned int *queue;    // next unclaimed id (two words are kept per buffer; the second is unused)
  unsigned long long *counters;
  uint32_t rad_words;     // 4, or 3 when the caller wants neither first_hit nor hit_sum: the fourth word of a path's
                          // record is then neither written nor read (a quarter of the kernel's HBM traffic)
};

// A member of the kernel's argument struct, read from the kernel-argument segment at the point of use
// through a pointer the compiler cannot see through (so the scalar loads stay there).  The struct has
// some 80 words; held in scalar registers for the whole kernel they push other scalars out into
// vector-register lanes, and every access to those costs a vector instruction.  Used for what the loop
// touches once per trip outside the scan: the camera, the shading tables, the queue and image geometry.
template <typename T>
__device__ __forceinline__ T kernarg_now(size_t offset) {
  static_assert(sizeof(T) % 4 == 0, "whole words");
  const __attribute__((address_space(4))) unsigned char *ka =
      (const __attribute__((address_space(4))) unsigned char *)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  const __attribute__((address_space(4))) uint32_t *src = (const __attribute__((address_space(4))) uint32_t *)(ka + offset);
  T out;
  uint32_t *dst = reinterpret_cast<uint32_t *>(&out);
#pragma unroll
  for (int q = 0; q < (int)(sizeof(T) / 4); ++q) dst[q] = src[q];
  return out;
}

// A pointer read with kernarg_now has lost its address space: the compiler would reach through it with flat
// instructions, which also count as LDS traffic for the wave's wait counters.  These name global memory again.
typedef float rt_vec4f __attribute__((ext_vector_type(4)));
typedef float rt_vec3f __attribute__((ext_vector_type(3)));
typedef __attribute__((address_space(1))) float global_f32;
typedef __attribute__((address_space(1))) rt_vec4f global_vec4f;
typedef __attribute__((address_space(1))) rt_vec3f global_vec3f;
typedef __attribute__((address_space(1))) unsigned int global_u32;

// a finished path's record into the radiance buffer (the pointer and the record size come from the kernel-argument
// segment at this point; the size is the same for every wave of a launch)
__device__ __forceinline__ void store_radiance(uint32_t out_index, V3 color, int32_t first_hit) {
  global_f32 *rad = (global_f32 *)(uintptr_t)kernarg_now<float4 *>(offsetof(TraceArgs, rad));
  if (kernarg_now<uint32_t>(offsetof(TraceArgs, rad_words)) == 3u)
    *(global_vec3f *)(rad + (size_t)3 * out_index) = rt_vec3f{color.x, color.y, color.z};
  else
    *(global_vec4f *)(rad + (size_t)4 * out_index) = rt_vec4f{color.x, color.y, color.z, __int_as_float(first_hit)};
}

// kProfile: also accumulate, per wave, the shader cycles spent in the scan, the
// drain and the shading, and the number of live lanes per bounce iteration
// (counters 4..9); selected by RT_FLAG_PROFILE, for diagnosis only.
// (second launch-bound argument: waves per SIMD.  The culled scan of an LDS-resident scene fits 96 VGPRs
// and, with its short lists, 32,000 bytes of LDS, so five workgroups share a CU; the other variants keep
// four, i.e. at most 128 VGPRs.)
#ifndef RT_BATCHED_WAVES
#define RT_BATCHED_WAVES 4      // waves per SIMD the culled scan of a scene scanned in place is compiled for
#endif
template <bool kBatched, bool kCull, bool kProfile>
__global__ __launch_bounds__(kBlockThreads, kCull ? (kBatched ? RT_BATCHED_WAVES : 5) : 4) void trace_kernel(const TraceArgs args) {
  extern __shared__ __align__(16) unsigned char smem[];
  const DeviceScene &sc = args.sc;
  const rt_job &job = args.geo.job;
  const ScanLds lds = carve_lds<kCull>(smem, kBatched ? 0u : sc.n_pad);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t npix = args.geo.rows * job.nx;

  if (!kBatched) {
    stage_batch(sc, lds, 0u, sc.n_pad, tid);
    __syncthreads();
  }

  // wave-uniform queue state
  uint32_t pool_next = 0, pool_end = 0;
  // where the chunk in hand starts: its first (tile, sample) index and that index split into tile, sample
  // and the tile into row and column of tiles -- one set of divisions per chunk instead of per path
  uint32_t pool_ts0 = 0, pool_tile0 = 0, pool_s0 = 0, pool_ty0 = 0, pool_tx0 = 0;
  bool exhausted = false;

  bool alive = false;
  uint32_t depth = 0, out_index = 0;
  int32_t first_hit = -1;
  V3 A = v3(0, 0, 0), B = v3(0, 0, 1), color = v3(1, 1, 1);
  Rng g{0u};
  // event counts of the wave, kept on the scalar unit (ballot popcounts and wave-uniform values)
  // (a launch has fewer than 2^32 paths and a path at most max_depth segments, shared among thousands of waves: the
  // per-wave counts of paths, segments, candidates and trips fit 32 bits; the test counts may not)
  uint32_t n_seg = 0, n_cand = 0, n_paths = 0, n_iter = 0;
  unsigned long long n_tests = 0, n_box = 0;
  uint32_t lane_tests = 0;                    // sphere tests this lane evaluated (culled scan), summed over the wave at the end
  unsigned long long n_own = 0, n_wave = 0;   // per lane, profiling builds only
  unsigned long long c_scan = 0, c_drain = 0, c_shade = 0;
#if RT_EXP == 39
  // lanes at work / times executed, per part of the trip (wave-uniform counts)
  uint32_t fill[30] = {};
  uint32_t lane_tries = 0;
#define RT_FILL(K, MASK)                                   \
  do {                                                     \
    const unsigned long long fill_m = (MASK);              \
    fill[K] += (uint32_t)__popcll(fill_m);                 \
    fill[(K) + 1] += fill_m != 0ull ? 1u : 0u;             \
  } while (0)
#else
#define RT_FILL(K, MASK) do { } while (0)
#endif
  const unsigned long long c_begin = kProfile ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long r_begin = kProfile ? __builtin_amdgcn_s_memrealtime() : 0ull;

  // One trip = one bounce of every live lane: (1) closest hit, (2) shading up to the point where a
  // matte or metal hit needs its point in the unit ball, (3) new paths for the lanes whose path just
  // ended, up to the point where the camera needs its point in the unit ball (c/structs.c:83), (4) ONE
  // rejection loop (c/structs.c:39-47 / c/main.c:124-131) for all of those lanes together -- run
  // separately, each of the three loops would cost the wave its slowest lane's trip count -- and
  // (5) the rest of the scatter or of the camera ray.
  enum : int { kNone = 0, kMatte = 1, kMetal = 2, kNewPath = 3 };
  bool work = false;   // some lane of the wave (batched: of the workgroup) has a ray to trace
  for (;;) {
    unsigned long long c0 = 0, c1 = 0, c2 = 0;
    int pending = kNone;
    // (read only where `pending` says they were set; left without initial values the kernel saves three dozen moves
    // and then needs 22 registers more than its 96)
    V3 p = v3(0, 0, 0), n = v3(0, 0, 0), albedo = v3(0, 0, 0);   // (kNewPath: p.x, p.y hold the pixel's x, y)
    V3 unitB = v3(0, 0, 1);                                        // normalize(B) of the ray just traced
    float fuzz = 0.0f;
    bool finished = false;
    if (kProfile) c0 = c1 = c2 = tick();
    if (work) {
      // ---- (1): closest hit
      n_iter++;
      HitState h{0.0f, -1, 0u, 0u, 0u, 0u, 0u, 0u, 0ull};   // (closest is read only where index >= 0)
      // The scan runs at a higher wave priority than the shading: its instructions come with every lane at work
      // and from LDS, the shading's with a third of the lanes and behind loads from memory, so a SIMD that
      // holds waves of both kinds does better issuing the scan first and the shading in the gaps (final
      // scene: 11.2 against 11.5 ms for a kernel alone; any level above the shading's gives the same).
      __builtin_amdgcn_s_setprio(1);
      closest_hit_scene<kBatched, kCull, kProfile>(sc, lds, tid, alive, A, B, job.tmin, kFltMax, h);
      __builtin_amdgcn_s_setprio(0);
      if (kProfile) {
        c2 = tick();
        c1 = c2 - h.drain_cycles;
      }
      {
        const uint32_t live = (uint32_t)__popcll(__ballot(alive));
        n_seg += live;
        n_cand += h.cand;
        n_tests += (unsigned long long)h.scan_tests * live;
        n_box += (unsigned long long)h.box_tests * live;
        lane_tests += h.lane_tests;
#if RT_EXP == 39
        for (int k = 0; k < 6; ++k) fill[k] += h.fill[k];
        fill[24] += h.fill[6];
        fill[25] += h.fill[7];
        fill[26] += h.fill[8];
        fill[27] += h.fill[9];
        fill[28] += h.fill[10];
        fill[29] += h.fill[11];
        fill[6] += live;
        fill[7] += 1u;
#endif
      }
      // ---- (2): shade, first part
      RT_STAGE_BEGIN(32);
      RT_FILL(8, __ballot(alive && h.index >= 0));
      RT_FILL(12, __ballot(alive && h.index < 0));
      if (alive) {
        if (kProfile) {
          n_own += h.own_groups;
          n_wave += h.wave_groups;
        }
        // the sky, the metal and the dielectric case each start from normalize(B): once, for all lanes
        const float lenB = length(B);
        unitB = narrowed_double_reciprocal(lenB) * B;               // c/vec.c:41-46
        bool done = false;
        if (depth == 0) first_hit = h.index;
        if (h.index >= 0) {
          const global_vec4f *rec =
              (const global_vec4f *)(uintptr_t)kernarg_now<const float4 *>(offsetof(TraceArgs, sc) + offsetof(DeviceScene, shade_o)) + 3 * h.index;
          const rt_vec4f e = rec[0];
          const rt_vec4f m4 = rec[1];
          const rt_vec4f tail = rec[2];
          const int type = __float_as_int(tail.z);
          p = A + h.closest * B;                              // c/structs.c:6-8
          const double inv_r = __hiloint2double(__float_as_int(tail.y), __float_as_int(tail.x));
          const V3 pc = p - v3(e.x, e.y, e.z);
          n = v3(div_f32_by(pc.x, e.w, inv_r), div_f32_by(pc.y, e.w, inv_r), div_f32_by(pc.z, e.w, inv_r));   // c/main.c:229
          RT_FILL(10, __ballot(type == RT_DIELECTRIC));
          if (type == RT_DIELECTRIC) {
            // (a dielectric has no albedo -- attenuation 1, c/main.c:183 -- and its record carries r0 and 1 / idx there)
            B = scatter_dielectric_unit(B, unitB, lenB, n, m4.w, m4.x, m4.y, g);
            A = p;
            depth++;
            done = depth >= job.max_depth;                    // c/main.c:262, 282
          } else {
            pending = type == RT_MATTE ? kMatte : kMetal;
            albedo = v3(m4.x, m4.y, m4.z);
            fuzz = m4.w;
          }
        } else {
          color = color * sky_unit(unitB.y);                  // c/main.c:274-279
          done = true;
        }
        if (done) {
          store_radiance(out_index, color, first_hit);
          finished = true;
          alive = false;
        }
      }
      RT_STAGE_END(32);
    }
    // ---- (3): hand new paths to the lanes without one
    RT_STAGE_BEGIN(33);
    const unsigned long long idle_mask = __ballot(!alive);
    if (idle_mask != 0ull && !exhausted) {
      const PathGeom geo = kernarg_now<PathGeom>(offsetof(TraceArgs, geo));
      const rt_job &jb = geo.job;
      if (pool_next == pool_end) {
        uint32_t base = 0;
        const uint32_t limit = geo.total_items, take = geo.chunk;
        if (lane == 0)
          base = __hip_atomic_fetch_add((global_u32 *)(uintptr_t)kernarg_now<unsigned int *>(offsetof(TraceArgs, queue)), take, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= limit) {
          exhausted = true;
        } else {
          pool_next = base;
          pool_end = min(base + take, limit);
          pool_ts0 = base >> 6;
          pool_tile0 = pool_ts0 / geo.batch;
          pool_s0 = pool_ts0 - pool_tile0 * geo.batch;
          pool_ty0 = pool_tile0 / geo.tiles_x;
          pool_tx0 = pool_tile0 - pool_ty0 * geo.tiles_x;
        }
      }
      const uint32_t avail = pool_end - pool_next;
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
#if RT_EXP == 61
      // ceiling of what a cheaper new-path stage could save: the stage reduced to a dozen instructions (other rays
      // than the product's, spread over the picture in the same way; every path still has a record of its own)
      if (!alive && rank < avail) {
        const uint32_t id = pool_next + rank;
        g = Rng{id * 0x9e3779b9u};
        p.x = (float)(id & 2047u) * (1.0f / 2048.0f);
        p.y = (float)((id >> 11) & 1023u) * (1.0f / 1024.0f);
        color = v3(1.0f, 1.0f, 1.0f);
        depth = 0;
        first_hit = -1;
        out_index = id;
        alive = true;
        pending = kNewPath;
      }
#else
      if (!alive && rank < avail) {
        const uint32_t id = pool_next + rank;
        const uint32_t l = id & 63u, ts = id >> 6;
        // (tile, sample) and (tile row, tile column) by stepping from the chunk's start: a chunk spans
        // at most chunk / 64 + 1 consecutive ts values, so the loops run a few times at most
        // unless the batch or the image is only a few units wide
        uint32_t s_local = pool_s0 + (ts - pool_ts0), tx = pool_tx0, ty = pool_ty0;
        while (s_local >= geo.batch) {
          s_local -= geo.batch;
          tx++;
        }
        while (tx >= geo.tiles_x) {
          tx -= geo.tiles_x;
          ty++;
        }
        // Rows of tiles are taken from the bottom of the job upwards: the paths of a launch that are handed out
        // last decide how long its waves go on after the queue is empty, and in pictures with the sky on
        // top those are then the shortest paths instead of the longest.
        const uint32_t col = tx * 8u + (l & 7u), lrow = (geo.last_tile_row - ty) * 8u + (l >> 3);
        if (col < jb.nx && lrow < geo.rows) {
          const uint32_t grow = global_row(jb, lrow, geo.tile_rows_magic);
          const uint32_t sample = geo.sample0 + s_local;
          g = rng_at(path_key(jb.seed, grow * jb.nx + col, sample));   // keyed by the global pixel
          pixel_xy_by(col, jb.ny - 1u - grow, jb.nx, jb.ny, geo.inv_nx, geo.inv_ny, g, p.x, p.y);  // rows below = j-1 of c/main.c:466
          color = v3(1.0f, 1.0f, 1.0f);
          depth = 0;
          first_hit = -1;
          out_index = s_local * npix + lrow * jb.nx + col;
          alive = true;
          pending = kNewPath;
        }
      }
#endif
      pool_next += min((uint32_t)__popcll(idle_mask), avail);
    }
    RT_STAGE_END(33);
    RT_FILL(14, __ballot(pending == kNewPath));
    // ---- (4): the point in the unit ball
    V3 ball = v3(0, 0, 0);
    RT_STAGE_BEGIN(31);
#if RT_EXP == 39
    {
      // (lane-tries counted per lane: every lane still looking makes one try per trip of the loop)
      bool looking = pending != kNone;
      while (wave_any(looking)) {
        fill[17] += 1u;
        if (looking) {
          lane_tries++;
          looking = !unit_ball_try(g, ball);
        }
      }
    }
#else
    if (pending != kNone) ball = unit_ball(g);
#endif
#if RT_EXP == 14   // the rejection loop a second time, from a generator state that gives other trip counts
    if (pending != kNone) {
      Rng g2{g.x ^ 0x5bd1e995u};
      const V3 b2 = unit_ball(g2);
      asm volatile("" ::"v"(b2.x), "v"(b2.y), "v"(b2.z));
    }
#endif
    RT_STAGE_END(31);
    // ---- (5): shade, second part / camera ray
    RT_STAGE_BEGIN(34);
    RT_FILL(18, __ballot(pending == kNewPath));
    RT_FILL(20, __ballot(pending == kMatte));
    RT_FILL(22, __ballot(pending == kMetal));
    if (pending == kNewPath) {
      const Camera cam = kernarg_now<Camera>(offsetof(TraceArgs, cam));
      camera_ray_from(cam, p.x, p.y, ball, A, B);
    } else if (pending != kNone) {
      bool done;
      if (pending == kMatte) {
        B = scatter_matte(p, n, ball);
        done = false;
      } else {
        // c/main.c:270-272: after an absorbed ray the colour is 0 and stays 0 whatever the rest of the
        // reference's loop does with the ray, so the path ends
        done = !scatter_metal_unit(unitB, n, fuzz, ball, B);
        if (done) albedo = v3(0.0f, 0.0f, 0.0f);
      }
      A = p;
      color = albedo * color;                                 // c/main.c:269
      depth++;
      done = done || depth >= job.max_depth;                  // c/main.c:262, 282
      if (done) {
        store_radiance(out_index, color, first_hit);
        finished = true;
        alive = false;
      }
    }
    RT_STAGE_END(34);
    n_paths += (uint32_t)__popcll(__ballot(finished));
    if (kProfile) {
      unsigned long long c3 = tick();
      c_scan += c1 - c0;
#if RT_EXP < 31 || RT_EXP > 34
      c_drain += c2 - c1;
#endif
      c_shade += c3 - c2;
    }
    // ---- next trip, or the end: nothing to trace and nothing left to hand out
    work = wave_any(alive);
    if (!work && exhausted) break;
  }

  if (kProfile && lane == 0 && args.counters) {
    atomicAdd(&args.counters[6], c_scan);
    atomicAdd(&args.counters[7], c_drain);
    atomicAdd(&args.counters[8], c_shade);
    atomicAdd(&args.counters[9], __builtin_amdgcn_s_memtime() - c_begin);
    const unsigned long long r_end = __builtin_amdgcn_s_memrealtime();
    atomicAdd(&args.counters[10], r_end - r_begin);
    // when the first and the last wave started and ended (100 MHz ticks of the device's constant clock)
    atomicMin(&args.counters[14], r_begin);
    atomicMax(&args.counters[15], r_begin);
    atomicMin(&args.counters[16], r_end);
    atomicMax(&args.counters[17], r_end);
    // this wave's lifetime (ticks) and trips, for the tail diagnosis
    const uint32_t wave_id = blockIdx.x * (kBlockThreads / 64) + (tid >> 6);
    if (wave_id < (uint32_t)kWaveLogEntries) args.counters[kWaveLogBase + wave_id] = ((r_end - r_begin) << 24) | (unsigned long long)(n_iter & 0xffffffu);
  }
  if (kProfile) {
    unsigned long long w_own = wave_sum(n_own), w_wave = wave_sum(n_wave);
    if (lane == 0 && args.counters) {
      atomicAdd(&args.counters[11], w_own);
      atomicAdd(&args.counters[12], w_wave);
    }
  }
#if RT_EXP == 39
  fill[16] = (uint32_t)wave_sum((unsigned long long)lane_tries);
  if (lane == 0 && args.counters)
    for (int k = 0; k < 30; ++k) atomicAdd(&args.counters[kWaveLogBase + k], (unsigned long long)fill[k]);
#endif
#undef RT_FILL
  // event counts: one atomic per wave and counter
  n_tests += wave_sum((unsigned long long)lane_tests);
  if (lane == 0 && args.counters) {
    atomicAdd(&args.counters[0], (unsigned long long)n_paths);
    atomicAdd(&args.counters[1], (unsigned long long)n_seg);
    atomicAdd(&args.counters[2], n_tests);
    atomicAdd(&args.counters[3], (unsigned long long)n_cand);
    atomicAdd(&args.counters[4], (unsigned long long)n_iter);
    atomicAdd(&args.counters[5], (unsigned long long)n_seg);    // (live lanes summed over the trips = segments)
    atomicAdd(&args.counters[kBoxTestCounter], n_box);
  }
}

// Sample accumulation and gamma (c/main.c:462-477): one lane per pixel adds
// this batch's samples in sample order to the running sum carried in `acc`
// ({r, g, b sum, hit_sum bits} per pixel) and, on the last batch, divides,
// takes the square root, scrubs NaN and quantises.  A progressive render keeps
// the sums after its last batch too, for the call that continues it.
struct ResolveArgs {
  const float4 *rad;
  float4 *acc;
  int32_t *fh_acc;        // first-hit index of sample 0, carried across batches
  uint32_t rad_words;     // 4, or 3: records without the first-hit word (first_hit and hit_sum are then not produced)
  uint32_t npix, batch, nsamples;   // nsamples: all samples the sums will hold after this batch's launch sequence
  int first_batch, last_batch;
  int keep_sum;           // also store the sums on the last batch (RT_FLAG_PROGRESSIVE)
  uint8_t *rgb8;
  float *linear;
  int32_t *first_hit;
  uint32_t *hit_sum;
};

__device__ __forceinline__ void resolve_pixel(const ResolveArgs &args, uint32_t p) {
  V3 sum = v3(0.0f, 0.0f, 0.0f);
  uint32_t hsum = 0;
  int32_t fh0 = -1;
  if (!args.first_batch) {
    float4 a = args.acc[p];
    sum = v3(a.x, a.y, a.z);
    hsum = (uint32_t)__float_as_int(a.w);
    fh0 = args.fh_acc[p];
  }
  if (args.rad_words == 3u) {
    const float *rad3 = reinterpret_cast<const float *>(args.rad);
    for (uint32_t s = 0; s < args.batch; ++s) {
      const float3 v = *reinterpret_cast<const float3 *>(rad3 + 3 * ((size_t)s * args.npix + p));
      sum = sum + v3(v.x, v.y, v.z);             // c/main.c:468, in sample order
    }
  } else {
    for (uint32_t s = 0; s < args.batch; ++s) {
      float4 v = args.rad[(size_t)s * args.npix + p];
      sum = sum + v3(v.x, v.y, v.z);             // c/main.c:468, in sample order
      int32_t fh = __float_as_int(v.w);
      hsum += (uint32_t)(fh + 1);
      if (args.first_batch && s == 0) fh0 = fh;
    }
  }
  if (!args.last_batch || args.keep_sum) {
    args.acc[p] = make_float4(sum.x, sum.y, sum.z, __int_as_float((int)hsum));
    args.fh_acc[p] = fh0;
    if (!args.last_batch) return;
  }
  V3 mean;
  uint8_t rgb[3];
  finish_pixel(sum, args.nsamples, mean, rgb);
  args.rgb8[3 * (size_t)p + 0] = rgb[0];
  args.rgb8[3 * (size_t)p + 1] = rgb[1];
  args.rgb8[3 * (size_t)p + 2] = rgb[2];
  if (args.linear) {
    args.linear[3 * (size_t)p + 0] = mean.x;
    args.linear[3 * (size_t)p + 1] = mean.y;
    args.linear[3 * (size_t)p + 2] = mean.z;
  }
  if (args.first_hit) args.first_hit[p] = fh0;
  if (args.hit_sum) args.hit_sum[p] = hsum;
}

__global__ __launch_bounds__(256) void resolve_kernel(const ResolveArgs args) {
  for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < args.npix; p += gridDim.x * 256u) resolve_pixel(args, p);
}


// ---------------------------------------------------------------------------
// Material-sorted wavefront variant (RT_KERNEL_WAVEFRONT).  Same arithmetic,
// same work items and the same radiance buffer as trace_kernel, but the bounce
// loop is cut into stages that hand paths over through global memory:
//   wf_generate   dead slots of a fixed pool take the next unclaimed paths
//   wf_intersect  closest hit for every live slot, then the slot id is pushed
//                 on the queue of what it hit: sky, matte, metal or dielectric
//   wf_shade<Q>   one launch per queue: every lane runs the same material
// It exists to measure what sorting by material buys against the megakernel's
// divergent shading, and what the state traffic and the launches cost.

struct WfPool {
  float4 *org;     // {A.xyz, t of the hit}
  float4 *dir;     // {B.xyz, hit index bits}
  float4 *col;     // {colour rgb, first-hit index bits}
  uint4 *misc;     // {rng stream position, unused, depth, out_index};  out_index == kWfDead: free slot
  uint32_t *qlist; // 4 queues of up to `slots` slot ids
  uint32_t *ctrl;  // [0] next item, [1..4] queue counts, [5] paths finished
  uint32_t slots;
};
constexpr uint32_t kWfDead = 0xffffffffu;
constexpr int kWfNext = 0, kWfQueue = 1, kWfFinished = 5, kWfCtrlWords = 8;

// Every workgroup of the wavefront stages handles kWfPerBlock slots in
// kWfSub passes of 256 and talks to the global counters once per counter:
// a single word takes only ~90 atomics per microsecond, which per-wave
// reservations (65,536 per stage and round) would make the bottleneck.
constexpr int kWfSub = 4;
constexpr uint32_t kWfPerBlock = kWfSub * kBlockThreads;

__global__ __launch_bounds__(kBlockThreads) void wf_generate(const TraceArgs args, const WfPool pool) {
  __shared__ uint32_t n_dead, base_id;
  const rt_job &job = args.geo.job;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) n_dead = 0;
  __syncthreads();
  uint32_t rank[kWfSub];
  bool dead[kWfSub];
#pragma unroll
  for (int sub = 0; sub < kWfSub; ++sub) {
    const uint32_t slot = blockIdx.x * kWfPerBlock + sub * kBlockThreads + tid;
    dead[sub] = slot < pool.slots && pool.misc[slot].w == kWfDead;
    rank[sub] = dead[sub] ? atomicAdd(&n_dead, 1u) : 0u;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t b = args.geo.total_items;
    if (n_dead != 0 && pool.ctrl[kWfNext] < args.geo.total_items) b = atomicAdd(&pool.ctrl[kWfNext], n_dead);
    base_id = b;
  }
  __syncthreads();
#pragma unroll
  for (int sub = 0; sub < kWfSub; ++sub) {
    if (!dead[sub]) continue;
    const uint32_t slot = blockIdx.x * kWfPerBlock + sub * kBlockThreads + tid;
    const uint32_t id = base_id + rank[sub];
    if (id >= args.geo.total_items) continue;
    const uint32_t l = id & 63u, ts = id >> 6;
    const uint32_t tile = ts / args.geo.batch, s_local = ts - tile * args.geo.batch;
    const uint32_t ty = tile / args.geo.tiles_x, tx = tile - ty * args.geo.tiles_x;
    const uint32_t col = tx * 8u + (l & 7u), lrow = ty * 8u + (l >> 3);
    if (col >= job.nx || lrow >= args.geo.rows) continue;   // off-image id: the slot stays free
    const uint32_t grow = global_row(job, lrow);
    Rng g = rng_at(path_key(job.seed, grow * job.nx + col, args.geo.sample0 + s_local));
    float x, y;
    pixel_xy(col, job.ny - 1u - grow, job.nx, job.ny, g, x, y);
    V3 A, B;
    camera_ray(args.cam, x, y, g, A, B);
    pool.org[slot] = make_float4(A.x, A.y, A.z, 0.0f);
    pool.dir[slot] = make_float4(B.x, B.y, B.z, __int_as_float(-1));
    pool.col[slot] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(-1));
    pool.misc[slot] = make_uint4(g.x, 0u, 0u, s_local * (args.geo.rows * job.nx) + lrow * job.nx + col);
  }
}

template <bool kBatched, bool kCull>
__global__ __launch_bounds__(kBlockThreads) void wf_intersect(const TraceArgs args, const WfPool pool) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ uint32_t qcnt[4], qbase[4];
  const DeviceScene &sc = args.sc;
  const ScanLds lds = carve_lds<kCull>(smem, kBatched ? 0u : sc.n_pad);
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  if (tid < 4) qcnt[tid] = 0;
  if (!kBatched) stage_batch(sc, lds, 0u, sc.n_pad, tid);
  __syncthreads();
  int q[kWfSub];
  uint32_t pos[kWfSub];
  unsigned long long n_seg = 0, n_cand = 0, n_tests = 0, n_box = 0;   // of the wave (wave-uniform)
  uint32_t lane_tests = 0;
#pragma unroll 1
  for (int sub = 0; sub < kWfSub; ++sub) {
    const uint32_t slot = blockIdx.x * kWfPerBlock + sub * kBlockThreads + tid;
    const bool alive = slot < pool.slots && pool.misc[slot].w != kWfDead;
    q[sub] = -1;
    pos[sub] = 0;
    if (!wave_any(alive)) continue;
    V3 A = v3(0, 0, 0), B = v3(0, 0, 1);
    if (alive) {
      float4 o = pool.org[slot], d = pool.dir[slot];
      A = v3(o.x, o.y, o.z);
      B = v3(d.x, d.y, d.z);
    }
    HitState h{kFltMax, -1, 0u, 0u, 0u, 0u, 0u, 0u, 0ull};
    closest_hit_scene<kBatched, kCull>(sc, lds, tid, alive, A, B, args.geo.job.tmin, kFltMax, h);
    {
      const uint32_t live = (uint32_t)__popcll(__ballot(alive));
      n_seg += live;
      n_cand += h.cand;
      n_tests += (unsigned long long)h.scan_tests * live;
      n_box += (unsigned long long)h.box_tests * live;
      lane_tests += h.lane_tests;
    }
    if (alive) {
      // queue 0 = sky, 1 + material type otherwise
      q[sub] = h.index < 0 ? 0 : 1 + sc.type_o[h.index];
      pos[sub] = atomicAdd(&qcnt[q[sub]], 1u);
      pool.org[slot].w = h.closest;
      pool.dir[slot].w = __int_as_float(h.index);
    }
  }
  __syncthreads();
  if (tid < 4 && qcnt[tid] != 0) qbase[tid] = atomicAdd(&pool.ctrl[kWfQueue + tid], qcnt[tid]);
  __syncthreads();
#pragma unroll
  for (int sub = 0; sub < kWfSub; ++sub) {
    if (q[sub] >= 0) {
      const uint32_t slot = blockIdx.x * kWfPerBlock + sub * kBlockThreads + tid;
      pool.qlist[(size_t)q[sub] * pool.slots + qbase[q[sub]] + pos[sub]] = slot;
    }
  }
  n_tests += wave_sum((unsigned long long)lane_tests);
  if (lane == 0 && args.counters && n_seg) {
    atomicAdd(&args.counters[1], n_seg);
    atomicAdd(&args.counters[2], n_tests);
    atomicAdd(&args.counters[3], n_cand);
    atomicAdd(&args.counters[kBoxTestCounter], n_box);
  }
}

// Q: 0 sky, 1 matte, 2 metal, 3 dielectric
template <int Q>
__global__ __launch_bounds__(kBlockThreads) void wf_shade(const TraceArgs args, const WfPool pool) {
  const uint32_t count = pool.ctrl[kWfQueue + Q];
  if (blockIdx.x * kWfPerBlock >= count) return;
  uint32_t n_finished = 0;
#pragma unroll 1
  for (int sub = 0; sub < kWfSub; ++sub) {
    const uint32_t i = blockIdx.x * kWfPerBlock + sub * kBlockThreads + threadIdx.x;
    if (i >= count) continue;
    bool finished = false;
    const uint32_t slot = pool.qlist[(size_t)Q * pool.slots + i];
    const float4 o = pool.org[slot], d = pool.dir[slot], c4 = pool.col[slot];
    uint4 mi = pool.misc[slot];
    V3 A = v3(o.x, o.y, o.z), B = v3(d.x, d.y, d.z), color = v3(c4.x, c4.y, c4.z);
    const int index = __float_as_int(d.w);
    int first_hit = __float_as_int(c4.w);
    if (mi.z == 0u) first_hit = index;
    if (Q == 0) {
      color = color * sky(B);                    // c/main.c:274-279
      finished = true;
    } else {
      const float4 e = args.sc.geom_o[index];
      const float4 m4 = args.sc.mat_o[index];
      V3 p = A + o.w * B;
      V3 n = (p - v3(e.x, e.y, e.z)) / e.w;
      Material m{Q - 1, v3(m4.x, m4.y, m4.z), m4.w};
      Rng g{mi.x};
      V3 atten, nA, nB;
      if (scatter(m, A, B, p, n, g, atten, nA, nB)) {
        color = atten * color;
        mi.x = g.x;
        mi.z += 1u;
        finished = mi.z >= args.geo.job.max_depth;
        if (!finished) {
          pool.org[slot] = make_float4(nA.x, nA.y, nA.z, 0.0f);
          pool.dir[slot] = make_float4(nB.x, nB.y, nB.z, __int_as_float(-1));
          pool.col[slot] = make_float4(color.x, color.y, color.z, __int_as_float(first_hit));
          pool.misc[slot] = mi;
        }
      } else {
        color = v3(0.0f, 0.0f, 0.0f);            // c/main.c:270-272
        finished = true;
      }
    }
    if (finished) {
      args.rad[mi.w] = make_float4(color.x, color.y, color.z, __int_as_float(first_hit));
      pool.misc[slot].w = kWfDead;
      n_finished++;
    }
  }
  __shared__ uint32_t block_finished;
  if (threadIdx.x == 0) block_finished = 0;
  __syncthreads();
  const unsigned long long w = wave_sum((unsigned long long)n_finished);
  if ((threadIdx.x & 63u) == 0 && w) atomicAdd(&block_finished, (uint32_t)w);
  __syncthreads();
  if (threadIdx.x == 0 && block_finished) {
    atomicAdd(&pool.ctrl[kWfFinished], block_finished);
    if (args.counters) atomicAdd(&args.counters[0], (unsigned long long)block_finished);
  }
}

// ---- unit-level kernels

template <bool kCull>
__global__ __launch_bounds__(kBlockThreads) void closest_hit_kernel(DeviceScene sc, const float *rays, uint32_t n_rays,
                                                                     float tmin, float tmax, int32_t *index, float *t) {
  extern __shared__ __align__(16) unsigned char smem[];
  const ScanLds lds = carve_lds<kCull>(smem, 0u);
  const uint32_t tid = threadIdx.x;
  const uint32_t r = blockIdx.x * kBlockThreads + tid;
  const bool alive = r < n_rays;
  V3 A = v3(0, 0, 0), B = v3(0, 0, 1);
  if (alive) {
    A = v3(rays[6 * r + 0], rays[6 * r + 1], rays[6 * r + 2]);
    B = v3(rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]);
  }
  HitState h{tmax, -1, 0u, 0u, 0u, 0u, 0u, 0u, 0ull};
  closest_hit_scene<true, kCull>(sc, lds, tid, alive, A, B, tmin, tmax, h);
  if (alive) {
    index[r] = h.index;
    t[r] = h.index >= 0 ? h.closest : 0.0f;
  }
}

__global__ void uniforms_kernel(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t n, double *out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Rng g = rng_at(path_key(seed, pixel, sample), i);
  out[i] = next_uniform(g);
}


// The filter routines on caller-supplied operands: every lane takes one ray and runs the pair image
// `pairs` (n_spheres entries, groups of eight) through filter_four_pairs, with the coefficients packed as
// the full scan packs them, and through filter_two_pairs, with the coefficients rebuilt from the ray
// table entries as the culled scan rebuilds them.
__global__ __launch_bounds__(kBlockThreads) void debug_filter_kernel(const float *rays, uint32_t n_rays, const float4 *pairs,
                                                                     uint32_t n_spheres, float *df_four, float *df_two) {
  const uint32_t r = blockIdx.x * kBlockThreads + threadIdx.x;
  if (r >= n_rays) return;
  const V3 A = v3(rays[6 * r + 0], rays[6 * r + 1], rays[6 * r + 2]), B = v3(rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]);
  const FilterRay f = make_filter_ray(A, B);
  const FilterRayP fr = pack_ray(f);
  // what closest_hit_scene keeps in the ray table, and what filter_listed_pairs makes of it
  const rt_v2f r0 = {A.x, A.y}, r1 = {A.z, dot(B, B)}, r2 = {B.x, B.y}, r3 = {B.z, f.na}, r5 = {f.ab, f.naaa};
  FilterRayP f2;
  f2.p0 = f32x2{-r2[0], -r2[1]};
  f2.p1 = f32x2{-r3[0], r5[0]};
  f2.p2 = f32x2{-2.0f * r0[0], -2.0f * r0[1]};
  f2.p3 = f32x2{-2.0f * r1[0], r3[1]};
  f2.p4 = f32x2{r5[1], 0.0f};
  for (uint32_t g = 0; g < n_spheres; g += kChunk) {
    float4 q[kChunk];
#pragma unroll
    for (int u = 0; u < kChunk; ++u) q[u] = pairs[g + u];
    f32x2 d[kChunk / 2];
    filter_four_pairs(fr, q, d);
#pragma unroll
    for (int u = 0; u < kChunk; ++u) df_four[(size_t)r * n_spheres + g + u] = d[u >> 1][u & 1];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const float4 qq[4] = {q[4 * half], q[4 * half + 1], q[4 * half + 2], q[4 * half + 3]};
      f32x2 d0, d1;
      filter_two_pairs(f2, qq, d0, d1);
      float *dst = df_two + (size_t)r * n_spheres + g + 4 * half;
      dst[0] = d0[0];
      dst[1] = d0[1];
      dst[2] = d1[0];
      dst[3] = d1[1];
    }
  }
}

// The box routines on caller-supplied operands: `boxes` holds pairs of boxes in the layout of
// DeviceScene::boxes (3 entries per pair).  box_pair reads them from vector registers, box_pair_scalar
// through scalar loads.
__global__ __launch_bounds__(kBlockThreads) void debug_box_kernel(const float *rays, uint32_t n_rays, const float4 *boxes, uint32_t n_boxes,
                                                                  float sigma0, float sigma1, uint8_t *pass_vec, uint8_t *pass_sca, float *box_ray) {
  const uint32_t r = blockIdx.x * kBlockThreads + threadIdx.x;
  const bool mine = r < n_rays;
  V3 A = v3(0, 0, 0), B = v3(1, 1, 1);
  if (mine) {
    A = v3(rays[6 * r + 0], rays[6 * r + 1], rays[6 * r + 2]);
    B = v3(rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]);
  }
  const BoxRay b = make_box_ray(A, B, sigma0, sigma1);
  const BoxRayP br = pack_box_ray(b);
  if (mine) {
    const float v[9] = {b.ix, b.iy, b.iz, b.nx, b.ny, b.nz, b.ax, b.ay, b.az};
    for (int k = 0; k < 9; ++k) box_ray[9 * (size_t)r + k] = v[k];
  }
  typedef __attribute__((address_space(4))) const float *ScalarFloats;
  const ScalarFloats sb = (ScalarFloats)(uintptr_t)boxes;
  for (uint32_t q = 0; q < n_boxes / 2u; ++q) {   // (every lane of the wave takes every trip: the scalar loads are wave-uniform)
    const float4 vq[3] = {boxes[3 * q], boxes[3 * q + 1], boxes[3 * q + 2]};
    bool pv[2], ps[2];
    box_pair(br, vq, pv);
    const ScalarFloats e = sb + 12u * q;
    const BoxPairS sq{f32x2{e[0], e[1]}, f32x2{e[2], e[3]}, f32x2{e[4], e[5]}, f32x2{e[6], e[7]}, f32x2{e[8], e[9]}, f32x2{e[10], e[11]}};
    box_pair_scalar(br, sq, ps);
    if (mine) {
      pass_vec[(size_t)r * n_boxes + 2 * q] = pv[0];
      pass_vec[(size_t)r * n_boxes + 2 * q + 1] = pv[1];
      pass_sca[(size_t)r * n_boxes + 2 * q] = ps[0];
      pass_sca[(size_t)r * n_boxes + 2 * q + 1] = ps[1];
    }
  }
}

__global__ void debug_arith_kernel(int op, const double *a, const double *b, uint32_t n, double *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = a[i], y = b ? b[i] : 0.0;
  double r = 0.0;
  switch (op) {
    case RT_ARITH_SQRT_WIDENED: r = sqrt_of_widened_positive_float((double)(float)x); break;
    case RT_ARITH_DIV_WIDENED: {
      const double da = (double)(float)y;
      r = div_by_widened_float(x, da, reciprocal_of_widened_float(da));
    } break;
    case RT_ARITH_DIV_F32_BY: r = (double)div_f32_by((float)x, (float)y, 1.0 / (double)(float)y); break;
    case RT_ARITH_RCP_NARROWED: r = (double)narrowed_double_reciprocal((float)x); break;
    case RT_ARITH_UNIT_FLOAT: r = (double)unit_float_from_bits((uint32_t)x); break;
    case RT_ARITH_UNIT_DOUBLE: r = unit_from_bits((uint32_t)x); break;
    default: break;
  }
  out[i] = r;
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

// Scratch of rt_render_device_static: one array per GPU, allocated when the code object is loaded on
// it and held until the process ends.  On the host the name is a stand-in, not an address: rt_ctx_create
// asks the runtime for the array's address on the context's device (static_scratch_address) and every
// use below goes through that pointer.
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[kMaxDevices] = {};

// the address of `scratch` on `device`, which must be the current device; asked once per device
int static_scratch_address(int device, unsigned char **out) {
  if (device < 0 || device >= kMaxDevices) return fail(RT_ERR_LIMIT, "device index beyond the static scratch table");
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) return fail(RT_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    g_scratch_ptr[device] = static_cast<unsigned char *>(ptr);
  }
  *out = g_scratch_ptr[device];
  return RT_OK;
}
}  // namespace

// what a progressive render leaves behind for the call that continues it
struct ProgressiveState {
  bool valid = false;
  rt_camera cam;
  rt_job job;               // the geometry fields, seed, depth, tmin of the call that made the sums
  uint32_t samples = 0;     // samples the sums hold
  unsigned long scene_generation = 0;
};

struct rt_ctx {
  int device = 0;
  unsigned char *static_scratch = nullptr;   // this device's copy of `scratch`
  ProgressiveState progressive;
  unsigned long scene_generation = 0;        // bumped by every rt_scene_upload
  hipStream_t own_stream = nullptr;
  // The trace kernels run on a stream of their own and each sample batch's resolve waits for its trace on the
  // context's stream: the trace of the next batch (or of the next render) overlaps the resolve of this one, which
  // is bound by HBM where the trace is bound by the vector ALUs.  Two radiance buffers alternate.
  hipStream_t trace_stream[2] = {nullptr, nullptr};   // (two, alternating like the buffers: the next batch's trace
                                                      // kernel fills the compute units as this one's last waves leave)
  hipEvent_t ev_traced[2] = {nullptr, nullptr}, ev_resolved[2] = {nullptr, nullptr};
  bool resolved_recorded[2] = {false, false};
  bool last_render_piped = true;     // else the trace stream has to wait for the context's stream once
  hipEvent_t ev_handover = nullptr;
  unsigned long batches_launched = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;   // around the whole of the last render
  bool timed = false;
  // trace-kernel event pairs of the last kHistory renders (a render has one pair per sample batch)
  static constexpr int kHistory = 32;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_events[kHistory];
  int trace_used[kHistory] = {};
  long renders = 0;
  DevBuf<float4> filter, boxes, sboxes, qboxes, exact, geom_o, mat_o, shade_o;
  DevBuf<int32_t> type_o;
  DevBuf<uint32_t> orig;
  DevBuf<double> inv_r_o;
  DevBuf<uint8_t> group_real;
  uint32_t n = 0, n_pad = 0, n_full = 0, n_always_groups = 0;
  std::map<std::pair<const void *, size_t>, int> blocks_per_cu;   // resident workgroups per CU of a trace kernel variant
  float sigma0 = 0.0f, sigma1 = 0.0f, cut_reach = 0.0f, cut_b = 0.0f;
  DevBuf<unsigned long long> counters;
  DevBuf<unsigned int> queue;        // work-queue head of the trace kernel
  DevBuf<float4> rad[2];             // per-path radiance of one sample batch (two: see trace_stream)
  DevBuf<float4> acc;                // running per-pixel sums between batches
  DevBuf<int32_t> fh_acc;
  int num_cus = 256;
  // slot pool of the wavefront variant
  DevBuf<float4> wf_org, wf_dir, wf_col;
  DevBuf<uint4> wf_misc;
  DevBuf<uint32_t> wf_qlist, wf_ctrl;
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
  if (!(job->tmin >= 0.0f)) return fail(RT_ERR_ARG, "tmin must be >= 0");
  if ((uint64_t)job->nx * job->ny > 0xffffffffull) return fail(RT_ERR_LIMIT, "more than 2^32 pixels");
  if (job->kernel != RT_KERNEL_MEGA && job->kernel != RT_KERNEL_WAVEFRONT)
    return fail(RT_ERR_ARG, "unknown kernel variant");
  if ((uint64_t)job->sample0 + job->nsamples > 0xffffffffull) return fail(RT_ERR_LIMIT, "sample indices beyond 2^32");
  return RT_OK;
}

DeviceScene scene_view(const rt_ctx *c) {
  return DeviceScene{c->filter.p, c->boxes.p, c->sboxes.p, c->qboxes.p, c->exact.p, c->orig.p, c->geom_o.p, c->mat_o.p, c->type_o.p, c->inv_r_o.p, c->shade_o.p, c->group_real.p, c->n, c->n_pad, c->sigma0, c->sigma1, c->cut_reach, c->cut_b, c->n_full, c->n_always_groups};
}

size_t scan_lds_bytes(uint32_t entries, bool cull) {
  return (size_t)pair_image_entries(entries) * sizeof(float4) + (size_t)3 * (entries / 16u + entries / 64u + quarter_pairs(entries)) * sizeof(float4) +  // pair image, boxes
         (size_t)3 * kBlockThreads * sizeof(float4) + (size_t)kBlockThreads * sizeof(unsigned long long) +  // rays, keys
         (size_t)(kBlockThreads / 64) * (wave_list_cap(cull) * sizeof(uint32_t) + kPairListCap * (entries == 0u ? sizeof(uint32_t) : sizeof(uint16_t))) +  // wave lists
         (size_t)((entries / 8u + 15u) & ~15u);                                             // real spheres per group
}
size_t trace_lds_bytes(uint32_t n_pad, bool cull) { return scan_lds_bytes(n_pad <= kMaxLdsSpheres ? n_pad : 0u, cull); }

// samples per launch: the radiance buffer holds batch * pixels float4 and is
// kept under kRadBudgetBytes
constexpr size_t kRadBudgetBytes = (size_t)3 << 30;

uint32_t batch_size(uint32_t npix, uint32_t nsamples, uint32_t rad_words) {
  size_t per_sample = (size_t)npix * rad_words * sizeof(float);
  size_t fit = kRadBudgetBytes / per_sample;
  if (fit < 1) fit = 1;
  return (uint32_t)(fit < nsamples ? fit : nsamples);
}

// One sample batch through the wavefront stages.  The host loop needs the
// finished-path count to know when to stop; it reads it back every few rounds.
constexpr uint32_t kWfSlots = 1u << 22;

int run_wavefront_batch(rt_ctx *c, const TraceArgs &t, uint32_t valid_paths, bool batched, size_t lds) {
  int rc;
  const uint32_t slots = kWfSlots;
  if ((rc = c->wf_org.reserve(slots)) || (rc = c->wf_dir.reserve(slots)) || (rc = c->wf_col.reserve(slots)) ||
      (rc = c->wf_misc.reserve(slots)) || (rc = c->wf_qlist.reserve((size_t)4 * slots)) ||
      (rc = c->wf_ctrl.reserve(kWfCtrlWords)))
    return rc;
  WfPool pool{c->wf_org.p, c->wf_dir.p, c->wf_col.p, c->wf_misc.p, c->wf_qlist.p, c->wf_ctrl.p, slots};
  RT_HIP(hipMemsetAsync(pool.misc, 0xff, (size_t)slots * sizeof(uint4), c->stream));
  RT_HIP(hipMemsetAsync(pool.ctrl, 0, kWfCtrlWords * sizeof(uint32_t), c->stream));
  const dim3 grid((slots + kWfPerBlock - 1) / kWfPerBlock), block(kBlockThreads);
  const uint32_t min_rounds = (t.geo.total_items + slots - 1) / slots;
  for (uint32_t round = 0;; ++round) {
    RT_HIP(hipMemsetAsync(pool.ctrl + kWfQueue, 0, 4 * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(wf_generate, grid, block, 0, c->stream, t, pool);
    const bool cull = (t.geo.job.flags & RT_FLAG_FULL_SCAN) == 0;
    if (batched && cull) hipLaunchKernelGGL((wf_intersect<true, true>), grid, block, lds, c->stream, t, pool);
    if (batched && !cull) hipLaunchKernelGGL((wf_intersect<true, false>), grid, block, lds, c->stream, t, pool);
    if (!batched && cull) hipLaunchKernelGGL((wf_intersect<false, true>), grid, block, lds, c->stream, t, pool);
    if (!batched && !cull) hipLaunchKernelGGL((wf_intersect<false, false>), grid, block, lds, c->stream, t, pool);
    hipLaunchKernelGGL(wf_shade<0>, grid, block, 0, c->stream, t, pool);
    hipLaunchKernelGGL(wf_shade<1>, grid, block, 0, c->stream, t, pool);
    hipLaunchKernelGGL(wf_shade<2>, grid, block, 0, c->stream, t, pool);
    hipLaunchKernelGGL(wf_shade<3>, grid, block, 0, c->stream, t, pool);
    RT_HIP(hipGetLastError());
    if (round + 1 >= min_rounds && (round & 3u) == 3u) {
      uint32_t finished = 0;
      RT_HIP(hipMemcpyAsync(&finished, pool.ctrl + kWfFinished, sizeof finished, hipMemcpyDeviceToHost, c->stream));
      RT_HIP(hipStreamSynchronize(c->stream));
      if (finished >= valid_paths) break;
      if (round > min_rounds + 4u * t.geo.job.max_depth + 64u) return fail(RT_ERR_HIP, "wavefront variant did not drain");
    }
  }
  return RT_OK;
}

// Device memory one render works in, beyond its inputs and outputs.
struct Workspace {
  float4 *rad[2];            // per-path radiance of one sample batch: batch * npix entries each; batches alternate
                             // between the two when the render is pipelined, else only rad[0] is used
  bool pipelined;            // trace kernels on the context's trace stream (see rt_ctx::trace_stream)
  float4 *acc;               // running per-pixel sums between batches (needed when batch < nsamples or progressive)
  int32_t *fh_acc;
  unsigned int *queue;       // work-queue head of the trace kernel
  unsigned long long *counters;
  uint32_t batch;            // samples per launch
  uint32_t rad_words;        // words per path in rad: 4, or 3 without the first-hit word (TraceArgs::rad_words)
};

// the persistent grid of a trace-kernel variant: as many workgroups as the device holds at the
// variant's occupancy (registers and LDS decide: five per CU for the culled scan of a scene the size
// of the reference's, fewer otherwise); asked once per kernel and LDS size
const void *trace_kernel_fn(bool batched, bool cull_scan, bool profile) {
#define RT_TRACE_FN(B, C, P) reinterpret_cast<const void *>(&trace_kernel<B, C, P>)
  if (!batched && cull_scan && !profile) return RT_TRACE_FN(false, true, false);
  if (!batched && cull_scan && profile) return RT_TRACE_FN(false, true, true);
  if (!batched && !cull_scan && !profile) return RT_TRACE_FN(false, false, false);
  if (!batched && !cull_scan && profile) return RT_TRACE_FN(false, false, true);
  if (batched && cull_scan && !profile) return RT_TRACE_FN(true, true, false);
  if (batched && cull_scan && profile) return RT_TRACE_FN(true, true, true);
  if (batched && !cull_scan && !profile) return RT_TRACE_FN(true, false, false);
  return RT_TRACE_FN(true, false, true);
#undef RT_TRACE_FN
}

int trace_blocks_per_cu(rt_ctx *c, const void *kernel_fn, size_t lds, int *out) {
  int &per_cu = c->blocks_per_cu[std::make_pair(kernel_fn, lds)];
  if (per_cu == 0) RT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel_fn, kBlockThreads, lds));
  if (per_cu < 1) return fail(RT_ERR_HIP, "the trace kernel does not fit a compute unit");
  *out = per_cu;
  return RT_OK;
}

// `record`: time the trace kernels with the context's event pairs (created on demand; a caller that
// must not create anything passes false)
int launch_render(rt_ctx *c, const rt_camera *cam, const rt_job *job, const Workspace &ws, bool record, uint8_t *rgb8,
                  float *linear, int32_t *first_hit, uint32_t *hit_sum) {
  static_assert(sizeof(Camera) == sizeof(rt_camera), "camera layouts differ");
  const uint32_t rows = rt_job_rows(job);
  const uint32_t npix = rows * job->nx;
  const uint32_t tiles_x = (job->nx + 7u) / 8u, tiles_y = (rows + 7u) / 8u;
  const uint32_t batch = ws.batch;
  if ((uint64_t)tiles_x * tiles_y * batch * 64u > 0xfffffff0ull - 4ull * kWfSlots)
    return fail(RT_ERR_LIMIT, "too many paths in one launch");
  const bool progressive = (job->flags & RT_FLAG_PROGRESSIVE) != 0;
  const uint32_t nsamples_total = job->sample0 + job->nsamples;

  TraceArgs t;
  std::memcpy(&t.cam, cam, sizeof t.cam);
  t.geo.job = *job;
  t.geo.rows = rows;
  t.geo.tiles_x = tiles_x;
  t.geo.last_tile_row = tiles_y - 1u;
  t.sc = scene_view(c);
  t.rad = ws.rad[0];
  t.queue = ws.queue;
  t.counters = ws.counters;
  t.rad_words = ws.rad_words;
  const bool cull_scan = (job->flags & RT_FLAG_FULL_SCAN) == 0;
  size_t lds = trace_lds_bytes(c->n_pad, cull_scan);
#if RT_EXP >= 41 && RT_EXP <= 43   // occupancy experiment: an LDS request of which only three (41), four (42) or two (43) fit a CU
  lds = RT_EXP == 41 ? 53760 : RT_EXP == 42 ? 40960 : 65536;
#endif
  const bool batched = c->n_pad > kMaxLdsSpheres, profile = (job->flags & RT_FLAG_PROFILE) != 0;
  const void *kernel_fn = trace_kernel_fn(batched, cull_scan, profile);
  int per_cu = 0, rc;
  if ((rc = trace_blocks_per_cu(c, kernel_fn, lds, &per_cu))) return rc;
  const dim3 grid((unsigned)(c->num_cus * per_cu)), block(kBlockThreads);

  const int slot = (int)(c->renders % rt_ctx::kHistory);
  auto &pairs = c->trace_events[slot];
  int used = 0;
  const bool piped_render = ws.pipelined && job->kernel == RT_KERNEL_MEGA;
  const bool own_buffers = ws.rad[1] != nullptr;     // the context's buffers (not the static scratch)
  if (piped_render && !c->last_render_piped) {
    // the render before ran wholly on the context's stream and shared the queue head and the counters
    RT_HIP(hipEventRecord(c->ev_handover, c->stream));
    RT_HIP(hipStreamWaitEvent(c->trace_stream[0], c->ev_handover, 0));
    RT_HIP(hipStreamWaitEvent(c->trace_stream[1], c->ev_handover, 0));
  }
  if (own_buffers) c->last_render_piped = piped_render;
  if (record) RT_HIP(hipEventRecord(c->ev_begin, piped_render ? c->trace_stream[c->batches_launched & 1u] : c->stream));
  for (uint32_t s0 = 0; s0 < job->nsamples; s0 += batch) {
    if (record && (int)pairs.size() <= used) {
      hipEvent_t e0, e1;
      RT_HIP(hipEventCreate(&e0));
      RT_HIP(hipEventCreate(&e1));
      pairs.emplace_back(e0, e1);
    }
    const uint32_t sb = job->nsamples - s0 < batch ? job->nsamples - s0 : batch;
    t.geo.sample0 = job->sample0 + s0;
    t.geo.batch = sb;
    t.geo.inv_nx = 1.0 / (double)(float)job->nx;
    t.geo.inv_ny = 1.0 / (double)(float)job->ny;
    t.geo.tile_rows_magic = (job->tile_rows > 1 && (uint64_t)t.geo.rows * job->tile_rows < (1ull << 32))
                            ? (uint32_t)(((1ull << 32) + job->tile_rows - 1) / job->tile_rows)
                            : 0u;
    t.geo.total_items = tiles_x * tiles_y * sb * 64u;
    // which radiance buffer, and on which stream the trace runs
    const bool piped = piped_render;
    const int buf = piped ? (int)(c->batches_launched & 1u) : 0;
    hipStream_t ts = piped ? c->trace_stream[buf] : c->stream;
    t.rad = ws.rad[buf];
    t.queue = ws.queue + 2 * buf;
    {
      // will this launch's last waves have company?  Yes if another batch of this render follows, or if the launch
      // before it is still at work (a caller that queues renders back to back)
      const bool followed = s0 + sb < job->nsamples;
      const bool behind = piped && c->batches_launched > 0 && hipEventQuery(c->ev_traced[buf ^ 1]) == hipErrorNotReady;
      t.geo.chunk = piped && (followed || behind) ? kChunkItems : kChunkAlone;
    }
    // (the resolve that last read this buffer must be done before the trace overwrites it)
    if (piped && c->resolved_recorded[buf]) RT_HIP(hipStreamWaitEvent(ts, c->ev_resolved[buf], 0));
    RT_HIP(hipMemsetAsync(t.queue, 0, 2 * sizeof(unsigned int), ts));
    if (record) RT_HIP(hipEventRecord(pairs[used].first, ts));
    if (job->kernel == RT_KERNEL_WAVEFRONT) {
      if ((rc = run_wavefront_batch(c, t, npix * sb, batched, lds))) return rc;
    } else {
      void *kernel_args[] = {&t};
      RT_HIP(hipLaunchKernel(kernel_fn, grid, block, kernel_args, lds, ts));
    }
    if (record) {
      RT_HIP(hipEventRecord(pairs[used].second, ts));
      ++used;
    }
    if (piped) {
      RT_HIP(hipEventRecord(c->ev_traced[buf], ts));
      RT_HIP(hipStreamWaitEvent(c->stream, c->ev_traced[buf], 0));
    }
    ResolveArgs r;
    r.rad = ws.rad[buf];
    r.rad_words = ws.rad_words;
    r.acc = ws.acc;
    r.fh_acc = ws.fh_acc;
    r.npix = npix;
    r.batch = sb;
    r.nsamples = nsamples_total;
    r.first_batch = job->sample0 == 0 && s0 == 0;
    r.last_batch = s0 + sb >= job->nsamples;
    r.keep_sum = progressive;
    r.rgb8 = rgb8;
    r.linear = linear;
    r.first_hit = first_hit;
    r.hit_sum = hit_sum;
    // (beside a trace kernel the resolve runs as a few resident workgroups that walk the pixels, so that it
    // trickles through HBM instead of contending for every compute unit at once)
    const uint32_t resolve_blocks = (npix + 255u) / 256u;
    hipLaunchKernelGGL(resolve_kernel, dim3(piped && resolve_blocks > 512u ? 512u : resolve_blocks), dim3(256), 0, c->stream, r);
    RT_HIP(hipGetLastError());
    if (own_buffers) {
      RT_HIP(hipEventRecord(c->ev_resolved[buf], c->stream));
      c->resolved_recorded[buf] = true;
    }
    if (piped) c->batches_launched++;
  }
  if (record) {
    RT_HIP(hipEventRecord(c->ev_end, c->stream));
    c->trace_used[slot] = used;
    c->renders++;
    c->timed = true;
  }
  return RT_OK;
}

// the fields of a job that must not change between the calls of one progressive render
bool same_progressive_render(const ProgressiveState &p, const rt_camera *cam, const rt_job *job, unsigned long scene_generation) {
  const rt_job &q = p.job;
  return p.valid && p.scene_generation == scene_generation && std::memcmp(&p.cam, cam, sizeof *cam) == 0 && q.nx == job->nx &&
         q.ny == job->ny && q.row0 == job->row0 && q.tile_rows == job->tile_rows && q.tile_stride == job->tile_stride &&
         q.n_tiles == job->n_tiles && q.max_depth == job->max_depth && std::memcmp(&q.tmin, &job->tmin, sizeof q.tmin) == 0 &&
         q.seed == job->seed;
}

}  // namespace

namespace {
// host arrays to the device, a kernel, device arrays back: the plumbing of the unit-level entry points
struct DebugBuffers {
  std::vector<std::pair<void *, size_t>> bufs;
  ~DebugBuffers() {
    for (auto &b : bufs) (void)hipFree(b.first);
  }
  template <typename T>
  int get(size_t n, T **out, const void *init = nullptr) {
    void *p = nullptr;
    if (hipMalloc(&p, n * sizeof(T) + 16) != hipSuccess) return fail(RT_ERR_NOMEM, "hipMalloc failed");
    bufs.emplace_back(p, n * sizeof(T));
    if (init && hipMemcpy(p, init, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return fail(RT_ERR_HIP, "hipMemcpy to device failed");
    *out = static_cast<T *>(p);
    return RT_OK;
  }
};
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
  {
    // The two trace streams get the highest stream priority.  Not for the scheduling: the runtime keeps a pool of
    // hardware queues per priority, so streams of a priority nobody else uses never share a hardware queue with the
    // caller's streams.  Sharing one is fatal for the overlap of consecutive launches: the wait "resolve after its
    // trace" queued on the caller's stream sits in front of the NEXT trace launch in the same hardware queue, and
    // every launch then starts only when the one before it has ended (measured with the process limited to two
    // hardware queues: 12.3 ms per 1080p frame against 10.9; with the priority set, 10.9 either way; with the
    // default four queues the sharing happens in some runs and not in others).  The lowest priority gives the same
    // separation but must not be used: when two processes share a GPU their normal-priority queues, waiting on
    // events of the trace streams, keep the low-priority queues that would signal those events from ever running.
    int least = 0, greatest = 0, chosen = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess) chosen = greatest;
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->trace_stream[0], hipStreamNonBlocking, chosen);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->trace_stream[1], hipStreamNonBlocking, chosen);
  }
  for (int i = 0; i < 2 && e == hipSuccess; ++i) {
    e = hipEventCreateWithFlags(&c->ev_traced[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_resolved[i], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_handover, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_begin);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_end);
  if (e != hipSuccess) {
    rt_ctx_destroy(c);
    return fail(RT_ERR_HIP, std::string("context set-up: ") + hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  // this device's copy of the static scratch (rt_render_device_static), with the device current
  // (a context works without it: rt_render_device_static then returns an error, nothing else depends on it)
  if (static_scratch_address(device, &c->static_scratch) != RT_OK) c->static_scratch = nullptr;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
    c->num_cus = prop.multiProcessorCount;
  int rc = c->counters.reserve(kWaveLogBase + kWaveLogEntries);
  if (rc == RT_OK) rc = c->queue.reserve(4);
  if (rc != RT_OK) {
    rt_ctx_destroy(c);
    return rc;
  }
  {
    unsigned long long zero[kNumCounters] = {};
    zero[14] = zero[16] = ~0ull;   // (the two minima)
    e = hipMemcpy(c->counters.p, zero, sizeof zero, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    rt_ctx_destroy(c);
    return fail(RT_ERR_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
  }
  *out = c;
  return RT_OK;
}

int rt_ctx_destroy(rt_ctx *c) {
  if (!c) return RT_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (int i = 0; i < 2; ++i)
    if (c->trace_stream[i]) (void)hipStreamSynchronize(c->trace_stream[i]);
  c->filter.release();
  c->boxes.release();
  c->sboxes.release();
  c->qboxes.release();
  c->orig.release();
  c->inv_r_o.release();
  c->shade_o.release();
  c->group_real.release();
  c->exact.release();
  c->geom_o.release();
  c->mat_o.release();
  c->type_o.release();
  c->counters.release();
  c->queue.release();
  c->rad[0].release();
  c->rad[1].release();
  c->acc.release();
  c->fh_acc.release();
  c->wf_org.release();
  c->wf_dir.release();
  c->wf_col.release();
  c->wf_misc.release();
  c->wf_qlist.release();
  c->wf_ctrl.release();
  c->out_rgb8.release();
  c->out_linear.release();
  c->out_first_hit.release();
  c->out_hit_sum.release();
  for (auto &slot : c->trace_events)
    for (auto &pr : slot) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
  if (c->ev_end) (void)hipEventDestroy(c->ev_end);
  for (int i = 0; i < 2; ++i) {
    if (c->ev_traced[i]) (void)hipEventDestroy(c->ev_traced[i]);
    if (c->ev_resolved[i]) (void)hipEventDestroy(c->ev_resolved[i]);
  }
  if (c->ev_handover) (void)hipEventDestroy(c->ev_handover);
  for (int i = 0; i < 2; ++i)
    if (c->trace_stream[i]) (void)hipStreamDestroy(c->trace_stream[i]);
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
    // (the scan order is built by sorting on the centres)
    if (!std::isfinite(spheres[i].center[0]) || !std::isfinite(spheres[i].center[1]) || !std::isfinite(spheres[i].center[2]) ||
        !std::isfinite(spheres[i].radius))
      return fail(RT_ERR_ARG, "centre or radius of sphere " + std::to_string(i) + " is not finite");
  }
  RT_HIP(hipSetDevice(c->device));
  // scan order and group bounds (rt_cluster.h); everything below is in scan order
  const rtcluster::Layout L = rtcluster::build_layout(spheres, n, kBatchSpheres);
  const uint32_t n_pad = L.n_scan;
  static_assert(rtcluster::kGroup == kChunk, "one group per chunk");
  // (scan positions 0 .. 65535 fit the 16-bit fields; the padding that completes the last pair of super-groups, and
  // the separate groups of much larger spheres, can push a scene of just under 65,536 spheres over)
  if (n_pad > kMaxSpheres + 1u) return fail(RT_ERR_LIMIT, "more than 65536 scan positions after padding (" + std::to_string(n_pad) + ")");
  std::vector<float4> filter(n_pad), exact(n_pad), boxes(3 * (n_pad / 16)), sboxes(3 * (n_pad / 64)), geom_o(n), mat_o(n);
  std::vector<int32_t> type_o(n);
  std::vector<double> inv_r_o(n);
  std::vector<float4> shade_o(3 * n);
  std::vector<FilterSphere> fs(n_pad);
  for (size_t i = 0; i < n; ++i) {
    const rt_sphere &s = spheres[i];
    geom_o[i] = make_float4(s.center[0], s.center[1], s.center[2], s.radius);
    mat_o[i] = make_float4(s.albedo[0], s.albedo[1], s.albedo[2], s.param);
    type_o[i] = s.type;
    inv_r_o[i] = 1.0 / (double)s.radius;
    uint32_t w[2];
    std::memcpy(w, &inv_r_o[i], sizeof w);
    float lo, hi, ty;
    std::memcpy(&lo, &w[0], 4);
    std::memcpy(&hi, &w[1], 4);
    std::memcpy(&ty, &s.type, 4);
    shade_o[3 * i] = geom_o[i];
    shade_o[3 * i + 1] = s.type == RT_DIELECTRIC
                             ? make_float4(schlick_r0(s.param), narrowed_double_reciprocal(s.param), 0.0f, s.param)   // c/main.c:141-142, 193
                             : mat_o[i];
    shade_o[3 * i + 2] = make_float4(lo, hi, ty, 0.0f);
  }
  for (uint32_t p = 0; p < n_pad; ++p) {
    const uint32_t i = L.order[p];
    if (i == 0xffffffffu) {   // pad: K' so large that b*b - a*K' is negative for any ray
      fs[p] = FilterSphere{0.0f, 0.0f, 0.0f, 1.0e30f};
      exact[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      continue;
    }
    const rt_sphere &s = spheres[i];
    fs[p] = make_filter_sphere(s.center[0], s.center[1], s.center[2], s.radius);
    exact[p] = make_float4(s.center[0], s.center[1], s.center[2], s.radius * s.radius);  // c/main.c:219
  }
  // pair images: entries 2p, 2p+1 = {x0,x1,y0,y1}, {z0,z1,K'0,K'1} of items 2p, 2p+1
  for (uint32_t p2 = 0; p2 < n_pad; p2 += 2) {
    filter[p2] = make_float4(fs[p2].cx, fs[p2 + 1].cx, fs[p2].cy, fs[p2 + 1].cy);
    filter[p2 + 1] = make_float4(fs[p2].cz, fs[p2 + 1].cz, fs[p2].kprime, fs[p2 + 1].kprime);
  }
  const GroupBox never{0.0f, 0.0f, 0.0f, -1.0e30f, -1.0e30f, -1.0e30f};
  for (uint32_t g = 0; g < n_pad / 8; g += 2) {
    // (the groups every ray scans are listed without a test: the box tests must not list them again)
    const GroupBox &b0 = g < L.n_always_groups ? never : L.boxes[g], &b1 = g + 1 < L.n_always_groups ? never : L.boxes[g + 1];
    boxes[3 * (g / 2)] = make_float4(b0.cx, b1.cx, b0.cy, b1.cy);
    boxes[3 * (g / 2) + 1] = make_float4(b0.cz, b1.cz, b0.hx, b1.hx);
    boxes[3 * (g / 2) + 2] = make_float4(b0.hy, b1.hy, b0.hz, b1.hz);
  }
  for (uint32_t q = 0; q < n_pad / 64; ++q) {
    const GroupBox &b0 = L.supers[2 * q], &b1 = L.supers[2 * q + 1];
    sboxes[3 * q] = make_float4(b0.cx, b1.cx, b0.cy, b1.cy);
    sboxes[3 * q + 1] = make_float4(b0.cz, b1.cz, b0.hx, b1.hx);
    sboxes[3 * q + 2] = make_float4(b0.hy, b1.hy, b0.hz, b1.hz);
  }
  std::vector<float4> qboxes(3 * quarter_pairs(n_pad));
  if (L.quarters.size() != 2 * (size_t)quarter_pairs(n_pad)) return fail(RT_ERR_LIMIT, "quarter boxes do not match the scan length");
  for (uint32_t q = 0; q < quarter_pairs(n_pad); ++q) {
    const GroupBox &b0 = L.quarters[2 * q], &b1 = L.quarters[2 * q + 1];
    qboxes[3 * q] = make_float4(b0.cx, b1.cx, b0.cy, b1.cy);
    qboxes[3 * q + 1] = make_float4(b0.cz, b1.cz, b0.hx, b1.hx);
    qboxes[3 * q + 2] = make_float4(b0.hy, b1.hy, b0.hz, b1.hz);
  }
  const uint32_t n_batch_pairs = (uint32_t)(L.batches.size() / 2);
  if (n_batch_pairs != batch_pairs(n_pad)) return fail(RT_ERR_LIMIT, "batch boxes do not match the scan length");
  const uint32_t n_region_pairs = (uint32_t)(L.regions.size() / 2);
  static_assert(kRegionBatches == (uint32_t)rtcluster::kRegionBatches, "one region size");
  if (n_region_pairs != region_pairs(n_pad)) return fail(RT_ERR_LIMIT, "region boxes do not match the scan length");
  sboxes.resize(3 * (n_pad / 64) + 3 * n_batch_pairs + 3 * n_region_pairs);
  for (uint32_t q = 0; q < n_region_pairs; ++q) {
    const GroupBox &b0 = L.regions[2 * q], &b1 = L.regions[2 * q + 1];
    float4 *dst = &sboxes[3 * (n_pad / 64) + 3 * n_batch_pairs + 3 * q];
    dst[0] = make_float4(b0.cx, b1.cx, b0.cy, b1.cy);
    dst[1] = make_float4(b0.cz, b1.cz, b0.hx, b1.hx);
    dst[2] = make_float4(b0.hy, b1.hy, b0.hz, b1.hz);
  }
  for (uint32_t q = 0; q < n_batch_pairs; ++q) {
    const GroupBox &b0 = L.batches[2 * q], &b1 = L.batches[2 * q + 1];
    float4 *dst = &sboxes[3 * (n_pad / 64) + 3 * q];
    dst[0] = make_float4(b0.cx, b1.cx, b0.cy, b1.cy);
    dst[1] = make_float4(b0.cz, b1.cz, b0.hx, b1.hx);
    dst[2] = make_float4(b0.hy, b1.hy, b0.hz, b1.hz);
  }
  int rc;
  if ((rc = c->filter.reserve(n_pad)) || (rc = c->boxes.reserve(3 * (n_pad / 16))) || (rc = c->sboxes.reserve(sboxes.size())) || (rc = c->qboxes.reserve(qboxes.size())) || (rc = c->exact.reserve(n_pad)) ||
      (rc = c->geom_o.reserve(n)) || (rc = c->mat_o.reserve(n)) || (rc = c->type_o.reserve(n)) ||
      (rc = c->orig.reserve(n_pad)) || (rc = c->inv_r_o.reserve(n)) || (rc = c->shade_o.reserve(3 * n)) || (rc = c->group_real.reserve(n_pad / 8)))
    return rc;
  RT_HIP(hipStreamSynchronize(c->stream));
  RT_HIP(hipMemcpy(c->filter.p, filter.data(), n_pad * sizeof(float4), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(c->boxes.p, boxes.data(), 3 * (n_pad / 16) * sizeof(float4), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(c->sboxes.p, sboxes.data(), sboxes.size() * sizeof(float4), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(c->qboxes.p, qboxes.data(), qboxes.size() * sizeof(float4), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(c->exact.p, exact.data(), n_pad * sizeof(float4), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(c->geom_o.p, geom_o.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(c->mat_o.p, mat_o.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(c->type_o.p, type_o.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(c->inv_r_o.p, inv_r_o.data(), n * sizeof(double), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(c->shade_o.p, shade_o.data(), 3 * n * sizeof(float4), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(c->group_real.p, L.real_in_group.data(), n_pad / 8, hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(c->orig.p, L.order.data(), n_pad * sizeof(uint32_t), hipMemcpyHostToDevice));
  c->scene_generation++;
  c->progressive.valid = false;
  c->n = (uint32_t)n;
  c->n_pad = n_pad;
  c->n_always_groups = L.n_always_groups;
  c->sigma0 = L.sigma0;
  c->sigma1 = L.sigma1;
  c->cut_reach = L.cut_reach;
  c->cut_b = L.cut_b;
  uint32_t last_real = 0;
  for (uint32_t p = 0; p < n_pad; ++p)
    if (L.order[p] != 0xffffffffu) last_real = p;
  c->n_full = (last_real / 16u + 1u) * 16u;
  // the occupancy of the trace-kernel variants this scene can run, asked now so that a render never has to
  for (int variant = 0; variant < 4; ++variant) {
    const bool cull = (variant & 1) != 0, profile = (variant & 2) != 0, batched = n_pad > kMaxLdsSpheres;
    int per_cu = 0;
    if ((rc = trace_blocks_per_cu(c, trace_kernel_fn(batched, cull, profile), trace_lds_bytes(n_pad, cull), &per_cu))) return rc;
  }
  return RT_OK;
}

int rt_render_device(rt_ctx *c, const rt_camera *cam, const rt_job *job, uint8_t *d_rgb8, float *d_linear,
                     int32_t *d_first_hit, uint32_t *d_hit_sum) {
  if (!c || !cam || !d_rgb8) return fail(RT_ERR_ARG, "null argument");
  int rc = check_job(job);
  if (rc != RT_OK) return rc;
  if (c->n == 0) return fail(RT_ERR_NO_SCENE, "no scene uploaded");
  const uint32_t rows = rt_job_rows(job);
  if (rows == 0) return fail(RT_ERR_ARG, "job covers no rows");
  const bool progressive = (job->flags & RT_FLAG_PROGRESSIVE) != 0;
  if (job->sample0 != 0) {
    if (!progressive) return fail(RT_ERR_ARG, "sample0 > 0 needs RT_FLAG_PROGRESSIVE");
    if (!same_progressive_render(c->progressive, cam, job, c->scene_generation) || c->progressive.samples != job->sample0)
      return fail(RT_ERR_ARG, "no running sums of this render that end at sample0 (scene, camera, geometry, seed, depth and "
                              "tmin must be those of the progressive call before)");
  }
  RT_HIP(hipSetDevice(c->device));
  const uint32_t npix = rows * job->nx;
  Workspace ws;
  // (the megakernel leaves the first-hit word of a path's record out when nobody will read it; the sums of a
  // progressive render may be continued by a call that does)
  ws.rad_words = (job->kernel == RT_KERNEL_MEGA && !progressive && !d_first_hit && !d_hit_sum) ? 3u : 4u;
  ws.batch = batch_size(npix, job->nsamples, ws.rad_words);
  const size_t rad_entries = ((size_t)ws.batch * npix * ws.rad_words + 3u) / 4u;   // in float4 units
  ws.pipelined = job->kernel == RT_KERNEL_MEGA && (job->flags & RT_FLAG_SERIAL) == 0;
  if (ws.pipelined && (c->rad[0].cap < rad_entries || c->rad[1].cap < rad_entries))
    RT_HIP(hipStreamSynchronize(c->stream));   // (a buffer about to be replaced may still be in use)
  if ((rc = c->rad[0].reserve(rad_entries)) || (ws.pipelined && (rc = c->rad[1].reserve(rad_entries)))) return rc;
  // (reserve keeps what a buffer holds unless it has to grow, and the sums of a progressive render were
  // reserved for the same npix)
  if ((ws.batch < job->nsamples || progressive) && ((rc = c->acc.reserve(npix)) || (rc = c->fh_acc.reserve(npix)))) return rc;
  ws.rad[0] = c->rad[0].p;
  ws.rad[1] = c->rad[1].p;
  ws.acc = c->acc.p;
  ws.fh_acc = c->fh_acc.p;
  ws.queue = c->queue.p;
  ws.counters = c->counters.p;
  c->progressive.valid = false;   // (whatever happens below, the sums are no longer those of the render before)
  rc = launch_render(c, cam, job, ws, true, d_rgb8, d_linear, d_first_hit, d_hit_sum);
  if (rc == RT_OK && progressive) {
    c->progressive.valid = true;
    c->progressive.cam = *cam;
    c->progressive.job = *job;
    c->progressive.samples = job->sample0 + job->nsamples;
    c->progressive.scene_generation = c->scene_generation;
  }
  return rc;
}

int rt_render_device_static(rt_ctx *c, const rt_camera *cam, const rt_job *job, uint8_t *d_rgb8, float *d_linear,
                            int32_t *d_first_hit, uint32_t *d_hit_sum) {
  if (!c || !cam || !d_rgb8) return fail(RT_ERR_ARG, "null argument");
  int rc = check_job(job);
  if (rc != RT_OK) return rc;
  if (c->n == 0) return fail(RT_ERR_NO_SCENE, "no scene uploaded");
  if (job->kernel != RT_KERNEL_MEGA) return fail(RT_ERR_LIMIT, "rt_render_device_static runs the megakernel only");
  if ((job->flags & RT_FLAG_PROGRESSIVE) != 0 || job->sample0 != 0)
    return fail(RT_ERR_LIMIT, "rt_render_device_static does not continue progressive renders");
  // (the profiling kernel logs one word per wave behind the event counts; the scratch's header has no room for that)
  if ((job->flags & RT_FLAG_PROFILE) != 0) return fail(RT_ERR_LIMIT, "rt_render_device_static does not run the profiling kernel");
  const uint64_t npix64 = (uint64_t)rt_job_rows(job) * job->nx;
  if (npix64 == 0) return fail(RT_ERR_ARG, "job covers no rows");
  if (npix64 > RT_STATIC_MAX_PIXELS)
    return fail(RT_ERR_LIMIT, "rt_render_device_static: " + std::to_string(npix64) + " pixels in one call, the static scratch holds " +
                                  std::to_string(RT_STATIC_MAX_PIXELS) + " (render the frame in row bands)");
  if (!c->static_scratch) return fail(RT_ERR_HIP, "the static scratch was not found on this device");
  RT_HIP(hipSetDevice(c->device));
  // carve-up of the array: [0, 256) queue head and event counts; running sums; first-hit carry; radiance
  const uint32_t npix = (uint32_t)npix64;
  constexpr size_t kHeader = 256;
  static_assert(64 + kNumCounters * sizeof(unsigned long long) <= kHeader, "header too small");
  static_assert(kHeader + (size_t)RT_STATIC_MAX_PIXELS * (2 * sizeof(float4) + sizeof(int32_t)) + 16 <= RT_STATIC_SCRATCH_BYTES,
                "the largest job must fit with one sample per batch");
  unsigned char *base = c->static_scratch;
  const size_t acc_off = kHeader, fh_off = acc_off + (size_t)npix * sizeof(float4);
  const size_t rad_off = (fh_off + (size_t)npix * sizeof(int32_t) + 15u) & ~(size_t)15u;
  Workspace ws;
  ws.rad_words = (!d_first_hit && !d_hit_sum) ? 3u : 4u;
  const size_t fit = (RT_STATIC_SCRATCH_BYTES - rad_off) / ((size_t)npix * ws.rad_words * sizeof(float));
  ws.batch = (uint32_t)(fit < job->nsamples ? fit : job->nsamples);
  ws.queue = reinterpret_cast<unsigned int *>(base);
  ws.counters = reinterpret_cast<unsigned long long *>(base + 64);
  ws.acc = reinterpret_cast<float4 *>(base + acc_off);
  ws.fh_acc = reinterpret_cast<int32_t *>(base + fh_off);
  ws.rad[0] = reinterpret_cast<float4 *>(base + rad_off);
  ws.rad[1] = nullptr;
  ws.pipelined = false;                  // (one scratch array: nothing to alternate with)
  // the array keeps the last call's contents: the queue head is cleared before every launch
  // (launch_render), the counts here; sums and radiance are written before they are read
  RT_HIP(hipMemsetAsync(base, 0, kHeader, c->stream));
  // nothing below may create anything: the occupancy of the kernel variants was asked at scene upload
  const bool cull_scan = (job->flags & RT_FLAG_FULL_SCAN) == 0, batched = c->n_pad > kMaxLdsSpheres;
  const void *fn = trace_kernel_fn(batched, cull_scan, false);
  if (c->blocks_per_cu.find(std::make_pair(fn, trace_lds_bytes(c->n_pad, cull_scan))) == c->blocks_per_cu.end())
    return fail(RT_ERR_HIP, "kernel occupancy unknown (scene upload did not complete)");
  return launch_render(c, cam, job, ws, false, d_rgb8, d_linear, d_first_hit, d_hit_sum);
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

int rt_ctx_trace_ms(rt_ctx *c, int back, float *ms) {
  if (!c || !ms) return fail(RT_ERR_ARG, "null argument");
  if (back < 0 || back >= rt_ctx::kHistory || (long)back >= c->renders)
    return fail(RT_ERR_ARG, "no such render in the history");
  RT_HIP(hipSetDevice(c->device));
  const int slot = (int)((c->renders - 1 - back) % rt_ctx::kHistory);
  float total = 0.0f;
  for (int i = 0; i < c->trace_used[slot]; ++i) {
    float t = 0.0f;
    RT_HIP(hipEventSynchronize(c->trace_events[slot][i].second));
    RT_HIP(hipEventElapsedTime(&t, c->trace_events[slot][i].first, c->trace_events[slot][i].second));
    total += t;
  }
  *ms = total;
  return RT_OK;
}

int rt_ctx_counters(rt_ctx *c, rt_counters *out, int reset) {
  if (!c) return fail(RT_ERR_ARG, "ctx is null");
  RT_HIP(hipSetDevice(c->device));
  RT_HIP(hipStreamSynchronize(c->stream));
  if (out) {
    unsigned long long v[kNumCounters];
    RT_HIP(hipMemcpy(v, c->counters.p, sizeof v, hipMemcpyDeviceToHost));
    out->paths = v[0];
    out->segments = v[1];
    out->sphere_tests = v[2];
    out->box_tests = v[kBoxTestCounter];
    out->candidates = v[3];
    out->wave_iterations = v[4];
    out->live_lane_iterations = v[5];
    out->cycles_scan = v[6];
    out->cycles_drain = v[7];
    out->cycles_shade = v[8];
    out->cycles_total = v[9];
    out->realtime_total = v[10];
    out->own_groups = v[11];
    out->wave_groups = v[12];
    out->first_wave_start = v[14] == ~0ull ? 0 : v[14];
    out->last_wave_start = v[15];
    out->first_wave_end = v[16] == ~0ull ? 0 : v[16];
    out->last_wave_end = v[17];
  }
  if (reset) {
    unsigned long long zero[kNumCounters] = {};
    zero[14] = zero[16] = ~0ull;   // (the two minima)
    RT_HIP(hipMemcpy(c->counters.p, zero, sizeof zero, hipMemcpyHostToDevice));
  }
  return RT_OK;
}

int rt_debug_closest_hit(rt_ctx *c, const float *rays, size_t n_rays, float tmin, float tmax, int32_t *index,
                         float *t) {
  if (!c || !rays || !index || !t) return fail(RT_ERR_ARG, "null argument");
  if (c->n == 0) return fail(RT_ERR_NO_SCENE, "no scene uploaded");
  if (n_rays == 0) return RT_OK;
  if (n_rays > 0x7fffffffu) return fail(RT_ERR_LIMIT, "too many rays");
  if (!(tmin >= 0.0f) || !(tmax > tmin)) return fail(RT_ERR_ARG, "need 0 <= tmin < tmax");
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
    size_t lds = scan_lds_bytes(0u, true);
    uint32_t blocks = (uint32_t)((n_rays + kBlockThreads - 1) / kBlockThreads);
    hipLaunchKernelGGL(closest_hit_kernel<true>, dim3(blocks), dim3(kBlockThreads), lds, c->stream, scene_view(c), d_rays.p,
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



int rt_debug_filter(rt_ctx *c, const float *rays, size_t n_rays, const float *spheres, size_t n_spheres, float *df_four, float *df_two) {
  if (!c || !rays || !spheres || !df_four || !df_two) return fail(RT_ERR_ARG, "null argument");
  if (n_rays == 0 || n_spheres == 0) return RT_OK;
  if (n_spheres % kChunk) return fail(RT_ERR_ARG, "n_spheres must be a multiple of 8");
  if (n_rays * n_spheres > (1ull << 28)) return fail(RT_ERR_LIMIT, "too many pairs");
  RT_HIP(hipSetDevice(c->device));
  std::vector<float4> pairs(n_spheres);
  for (size_t p2 = 0; p2 < n_spheres; p2 += 2) {
    const FilterSphere s0 = make_filter_sphere(spheres[4 * p2], spheres[4 * p2 + 1], spheres[4 * p2 + 2], spheres[4 * p2 + 3]);
    const FilterSphere s1 = make_filter_sphere(spheres[4 * p2 + 4], spheres[4 * p2 + 5], spheres[4 * p2 + 6], spheres[4 * p2 + 7]);
    pairs[p2] = make_float4(s0.cx, s1.cx, s0.cy, s1.cy);
    pairs[p2 + 1] = make_float4(s0.cz, s1.cz, s0.kprime, s1.kprime);
  }
  DebugBuffers d;
  float *d_rays, *d_four, *d_two;
  float4 *d_pairs;
  int rc;
  if ((rc = d.get(6 * n_rays, &d_rays, rays)) || (rc = d.get(n_spheres, &d_pairs, pairs.data())) ||
      (rc = d.get(n_rays * n_spheres, &d_four)) || (rc = d.get(n_rays * n_spheres, &d_two)))
    return rc;
  hipLaunchKernelGGL(debug_filter_kernel, dim3((unsigned)((n_rays + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads), 0, c->stream,
                     d_rays, (uint32_t)n_rays, d_pairs, (uint32_t)n_spheres, d_four, d_two);
  RT_HIP(hipGetLastError());
  RT_HIP(hipStreamSynchronize(c->stream));
  RT_HIP(hipMemcpy(df_four, d_four, n_rays * n_spheres * sizeof(float), hipMemcpyDeviceToHost));
  RT_HIP(hipMemcpy(df_two, d_two, n_rays * n_spheres * sizeof(float), hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_debug_box_pass(rt_ctx *c, const float *rays, size_t n_rays, const float *boxes, size_t n_boxes, float sigma0, float sigma1,
                      uint8_t *pass_vec, uint8_t *pass_sca, float *box_ray) {
  if (!c || !rays || !boxes || !pass_vec || !pass_sca || !box_ray) return fail(RT_ERR_ARG, "null argument");
  if (n_rays == 0 || n_boxes == 0) return RT_OK;
  if (n_boxes % 2) return fail(RT_ERR_ARG, "n_boxes must be even");
  if (n_rays * n_boxes > (1ull << 28)) return fail(RT_ERR_LIMIT, "too many pairs");
  RT_HIP(hipSetDevice(c->device));
  std::vector<float4> packed(3 * (n_boxes / 2));
  for (size_t q = 0; q < n_boxes / 2; ++q) {
    const float *b0 = boxes + 12 * q, *b1 = b0 + 6;
    packed[3 * q] = make_float4(b0[0], b1[0], b0[1], b1[1]);
    packed[3 * q + 1] = make_float4(b0[2], b1[2], b0[3], b1[3]);
    packed[3 * q + 2] = make_float4(b0[4], b1[4], b0[5], b1[5]);
  }
  DebugBuffers d;
  float *d_rays, *d_box_ray;
  float4 *d_boxes;
  uint8_t *d_vec, *d_sca;
  int rc;
  if ((rc = d.get(6 * n_rays, &d_rays, rays)) || (rc = d.get(packed.size(), &d_boxes, packed.data())) ||
      (rc = d.get(n_rays * n_boxes, &d_vec)) || (rc = d.get(n_rays * n_boxes, &d_sca)) || (rc = d.get(9 * n_rays, &d_box_ray)))
    return rc;
  hipLaunchKernelGGL(debug_box_kernel, dim3((unsigned)((n_rays + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads), 0, c->stream,
                     d_rays, (uint32_t)n_rays, d_boxes, (uint32_t)n_boxes, sigma0, sigma1, d_vec, d_sca, d_box_ray);
  RT_HIP(hipGetLastError());
  RT_HIP(hipStreamSynchronize(c->stream));
  RT_HIP(hipMemcpy(pass_vec, d_vec, n_rays * n_boxes, hipMemcpyDeviceToHost));
  RT_HIP(hipMemcpy(pass_sca, d_sca, n_rays * n_boxes, hipMemcpyDeviceToHost));
  RT_HIP(hipMemcpy(box_ray, d_box_ray, 9 * n_rays * sizeof(float), hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_debug_wave_log(rt_ctx *c, uint64_t *out, size_t n) {
  if (!c || !out) return fail(RT_ERR_ARG, "null argument");
  if (n > (size_t)kWaveLogEntries) n = kWaveLogEntries;
  RT_HIP(hipSetDevice(c->device));
  RT_HIP(hipStreamSynchronize(c->stream));
  RT_HIP(hipMemcpy(out, c->counters.p + kWaveLogBase, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_debug_arith(rt_ctx *c, int op, const double *a, const double *b, size_t n, double *out) {
  if (!c || !a || !out) return fail(RT_ERR_ARG, "null argument");
  if (op < RT_ARITH_SQRT_WIDENED || op > RT_ARITH_UNIT_DOUBLE) return fail(RT_ERR_ARG, "unknown operation");
  if ((op == RT_ARITH_DIV_WIDENED || op == RT_ARITH_DIV_F32_BY) && !b) return fail(RT_ERR_ARG, "the operation needs a divisor array");
  if (n == 0) return RT_OK;
  if (n > (1u << 26)) return fail(RT_ERR_LIMIT, "too many operands");
  RT_HIP(hipSetDevice(c->device));
  DebugBuffers d;
  double *d_a, *d_b = nullptr, *d_out;
  int rc;
  if ((rc = d.get(n, &d_a, a)) || (b && (rc = d.get(n, &d_b, b))) || (rc = d.get(n, &d_out))) return rc;
  hipLaunchKernelGGL(debug_arith_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, op, d_a, d_b, (uint32_t)n, d_out);
  RT_HIP(hipGetLastError());
  RT_HIP(hipStreamSynchronize(c->stream));
  RT_HIP(hipMemcpy(out, d_out, n * sizeof(double), hipMemcpyDeviceToHost));
  return RT_OK;
}

}  // extern "C"
