// valu_rate.hip -- issue-rate microbenchmarks behind the kernel design:
//   fma      : independent v_fma_f32 streams
//   pkfma    : independent v_pk_fma_f32 streams (two floats per lane per op)
//   fma_lds  : 8 v_fma_f32 per broadcast ds_read_b128 (the filter's shape)
//   fma_sgpr : 8 v_fma_f32 per 16 bytes fetched with s_load (scalar path)
//   mfma     : v_mfma_f32_16x16x4_f32 alone, and interleaved with v_fma_f32
// Each is run at 1, 2, 4, 8 waves per SIMD (256-thread blocks, 1..8 blocks per
// CU on 256 CUs).  Output: wave-instructions per cycle per SIMD at the clock
// measured with s_memtime / s_memrealtime.
//
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      printf("%s: %s\n", #x, hipGetErrorString(e));                               \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

constexpr int kIters = 2000;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_fma(float *out, float x, unsigned long long *clk) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile(
          "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
          "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(x));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void k_pkfma(float *out, float x, unsigned long long *clk) {
  f32x2 a0 = {(float)threadIdx.x, 1}, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f32x2 xx = {x, x};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile(
          "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
          "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(xx));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  f32x2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// the filter's shape: one broadcast 16-byte LDS read feeds 8 FMAs
__global__ __launch_bounds__(256) void k_fma_lds(float *out, float x, unsigned long long *clk) {
  __shared__ f32x4 lds[512];
  for (int i = threadIdx.x; i < 512; i += 256) lds[i] = f32x4{x, x * 0.5f, x * 0.25f, (float)i};
  __syncthreads();
  float nb0 = threadIdx.x, nb1 = x, nb2 = x + 1, ab = 0.5f, ma0 = 2, ma1 = 3, ma2 = 4, na = -1, naaa = -3;
  float acc = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < kIters / 4; ++i) {
#pragma unroll 16
    for (int u = 0; u < 32; ++u) {
      f32x4 s = lds[(i * 32 + u) & 511];
      float b = __builtin_fmaf(nb0, s.x, __builtin_fmaf(nb1, s.y, __builtin_fmaf(nb2, s.z, ab)));
      float c0 = __builtin_fmaf(ma0, s.x, __builtin_fmaf(ma1, s.y, __builtin_fmaf(ma2, s.z, s.w)));
      float d = __builtin_fmaf(na, c0, __builtin_fmaf(b, b, naaa));
      acc = fmaxf(acc, d);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// the same with the 16 bytes fetched by scalar loads (uniform address -> SGPRs)
__global__ __launch_bounds__(256) void k_fma_sgpr(float *out, float x, const f32x4 *__restrict__ tab, unsigned long long *clk) {
  float nb0 = threadIdx.x, nb1 = x, nb2 = x + 1, ab = 0.5f, ma0 = 2, ma1 = 3, ma2 = 4, na = -1, naaa = -3;
  float acc = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < kIters / 4; ++i) {
#pragma unroll 16
    for (int u = 0; u < 32; ++u) {
      f32x4 s = tab[(i * 32 + u) & 511];   // uniform index: s_load_dwordx4
      float b = __builtin_fmaf(nb0, s.x, __builtin_fmaf(nb1, s.y, __builtin_fmaf(nb2, s.z, ab)));
      float c0 = __builtin_fmaf(ma0, s.x, __builtin_fmaf(ma1, s.y, __builtin_fmaf(ma2, s.z, s.w)));
      float d = __builtin_fmaf(na, c0, __builtin_fmaf(b, b, naaa));
      acc = fmaxf(acc, d);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// mfma_per: MFMAs per group; fma_per: v_fma per group
template <int kMfma, int kFma>
__global__ __launch_bounds__(256) void k_mfma(float *out, float x, unsigned long long *clk) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float a = threadIdx.x, b = x;
  float f0 = 1, f1 = 2, f2 = 3, f3 = 4, f4 = 5, f5 = 6, f6 = 7, f7 = 8;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (kMfma >= 1) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      if (kFma >= 1) asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n" : "+v"(f0), "+v"(f1) : "v"(x));
      if (kMfma >= 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      if (kFma >= 2) asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n" : "+v"(f2), "+v"(f3) : "v"(x));
      if (kMfma >= 3) c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
      if (kFma >= 3) asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n" : "+v"(f4), "+v"(f5) : "v"(x));
      if (kMfma >= 4) c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
      if (kFma >= 4) asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n" : "+v"(f6), "+v"(f7) : "v"(x));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  f32x4 s = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <typename F>
void run(const char *name, F launch, double wave_instr_per_thread_block_wave, float *d_out, unsigned long long *d_clk) {
  for (int blocks_per_cu : {1, 2, 4, 8}) {
    int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch(grid);  // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch(grid);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk[2];
    CHECK(hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost));
    double ghz = (double)clk[0] / ((double)clk[1] * 10.0);  // memrealtime ticks at 100 MHz
    // per SIMD: blocks_per_cu waves, each issuing wave_instr instructions, in clk[0] cycles (block 0's view)
    double per_simd = blocks_per_cu * wave_instr_per_thread_block_wave / (double)clk[0];
    double wall_per_simd = blocks_per_cu * wave_instr_per_thread_block_wave / (ms * 1e-3 * ghz * 1e9);
    printf("%-10s waves/SIMD %d: %.3f ms, clock %.2f GHz, wave-instr/cycle/SIMD %.3f (in-kernel) %.3f (wall)\n", name,
           blocks_per_cu, ms, ghz, per_simd, wall_per_simd);
  }
}

int main() {
  float *d_out;
  unsigned long long *d_clk;
  f32x4 *d_tab;
  CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float)));
  CHECK(hipMalloc(&d_clk, 2 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&d_tab, 512 * sizeof(f32x4)));
  std::vector<f32x4> tab(512);
  for (int i = 0; i < 512; ++i) tab[i] = f32x4{1.0f, 0.5f, 0.25f, (float)i};
  CHECK(hipMemcpy(d_tab, tab.data(), 512 * sizeof(f32x4), hipMemcpyHostToDevice));
  float x = 1.0000001f;
  run("fma", [&](int g) { hipLaunchKernelGGL(k_fma, dim3(g), dim3(256), 0, 0, d_out, x, d_clk); }, kIters * 32.0, d_out, d_clk);
  run("pkfma", [&](int g) { hipLaunchKernelGGL(k_pkfma, dim3(g), dim3(256), 0, 0, d_out, x, d_clk); }, kIters * 32.0, d_out, d_clk);
  // 9 VALU (8 fma + max) per sphere
  run("fma_lds", [&](int g) { hipLaunchKernelGGL(k_fma_lds, dim3(g), dim3(256), 0, 0, d_out, x, d_clk); }, (kIters / 4) * 32 * 9.0, d_out, d_clk);
  run("fma_sgpr", [&](int g) { hipLaunchKernelGGL(k_fma_sgpr, dim3(g), dim3(256), 0, 0, d_out, x, d_tab, d_clk); }, (kIters / 4) * 32 * 9.0, d_out, d_clk);
  run("mfma4", [&](int g) { hipLaunchKernelGGL((k_mfma<4, 0>), dim3(g), dim3(256), 0, 0, d_out, x, d_clk); }, kIters * 16.0, d_out, d_clk);
  run("mfma1", [&](int g) { hipLaunchKernelGGL((k_mfma<1, 0>), dim3(g), dim3(256), 0, 0, d_out, x, d_clk); }, kIters * 4.0, d_out, d_clk);
  run("mfma4+fma8", [&](int g) { hipLaunchKernelGGL((k_mfma<4, 4>), dim3(g), dim3(256), 0, 0, d_out, x, d_clk); }, kIters * (16.0 + 32.0), d_out, d_clk);
  run("mfma2+fma8", [&](int g) { hipLaunchKernelGGL((k_mfma<2, 4>), dim3(g), dim3(256), 0, 0, d_out, x, d_clk); }, kIters * (8.0 + 32.0), d_out, d_clk);
  run("mfma1+fma8", [&](int g) { hipLaunchKernelGGL((k_mfma<1, 4>), dim3(g), dim3(256), 0, 0, d_out, x, d_clk); }, kIters * (4.0 + 32.0), d_out, d_clk);
  run("mfma0+fma8", [&](int g) { hipLaunchKernelGGL((k_mfma<0, 4>), dim3(g), dim3(256), 0, 0, d_out, x, d_clk); }, kIters * 32.0, d_out, d_clk);
  return 0;
}
