// op_rate.hip -- issue cost of single vector instructions on gfx950: eight
// independent streams of one instruction, four waves per SIMD, cycles per
// wave-instruction per SIMD from s_memtime.  The numbers price the parts of
// the trace kernel that are not fused multiply-adds (generator, conversions,
// compares, double-precision roots).
//
// build: hipcc -O3 --offload-arch=gfx950 -o op_rate op_rate.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      printf("%s: %s\n", #x, hipGetErrorString(e));                               \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

constexpr int kIters = 2000;

// 32-bit destination, 32-bit sources: OP d, d, x [, x]
#define KERNEL32(NAME, TEMPLATE)                                                                        \
  __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned x, unsigned long long *clk) {      \
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,  \
             a7 = a0 + 7;                                                                                \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                \
    for (int i = 0; i < kIters; ++i) {                                                                   \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                    \
        asm volatile(TEMPLATE(0) TEMPLATE(1) TEMPLATE(2) TEMPLATE(3) TEMPLATE(4) TEMPLATE(5) TEMPLATE(6)  \
                         TEMPLATE(7)                                                                     \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)    \
                     : "v"(x));                                                                          \
      }                                                                                                  \
    }                                                                                                    \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                  \
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;                                           \
  }

// 64-bit destination and sources
#define KERNEL64(NAME, TEMPLATE)                                                                        \
  __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned x, unsigned long long *clk) {      \
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
           a7 = a0 + 7, xx = (double)x * 1e-9;                                                           \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                \
    for (int i = 0; i < kIters; ++i) {                                                                   \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                    \
        asm volatile(TEMPLATE(0) TEMPLATE(1) TEMPLATE(2) TEMPLATE(3) TEMPLATE(4) TEMPLATE(5) TEMPLATE(6)  \
                         TEMPLATE(7)                                                                     \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)    \
                     : "v"(xx));                                                                         \
      }                                                                                                  \
    }                                                                                                    \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);      \
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;                                           \
  }

#define T_MUL_LO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define T_MUL_HI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define T_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %8\n"
#define T_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define T_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define T_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define T_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %8\n"
#define T_RCP32(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define T_CMP(i) "v_cmp_lt_f32 s[20:21], %" #i ", %8\n"
#define T_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
#define T_CVTF32U32(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define T_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %8\n"
#define T_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define T_ADD64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define T_RSQ64(i) "v_rsq_f64 %" #i ", %" #i "\n"
#define T_RCP64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define T_LDEXP64(i) "v_ldexp_f64 %" #i ", %" #i ", 1\n"

KERNEL32(k_mul_lo, T_MUL_LO)
KERNEL32(k_mul_hi, T_MUL_HI)
KERNEL32(k_mad24, T_MAD24)
KERNEL32(k_xor, T_XOR)
KERNEL32(k_lshladd, T_LSHLADD)
KERNEL32(k_fma32, T_FMA32)
KERNEL32(k_max3, T_MAX3)
KERNEL32(k_rcp32, T_RCP32)
KERNEL32(k_mbcnt, T_MBCNT)
KERNEL32(k_cvtf32u32, T_CVTF32U32)
KERNEL64(k_fma64, T_FMA64)
KERNEL64(k_mul64, T_MUL64)
KERNEL64(k_add64, T_ADD64)
KERNEL64(k_rsq64, T_RSQ64)
KERNEL64(k_rcp64, T_RCP64)
KERNEL64(k_ldexp64, T_LDEXP64)

// compare into a scalar pair (clobbers s[20:21])
__global__ __launch_bounds__(256) void k_cmp(unsigned *out, unsigned x, unsigned long long *clk) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float xf = (float)x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile(T_CMP(0) T_CMP(1) T_CMP(2) T_CMP(3) T_CMP(4) T_CMP(5) T_CMP(6) T_CMP(7)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(xf)
                   : "s20", "s21");
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

// conversions between widths
__global__ __launch_bounds__(256) void k_cvt_f64_u32(unsigned *out, unsigned x, unsigned long long *clk) {
  unsigned a0 = threadIdx.x + x;
  double d0, d1, d2, d3, d4, d5, d6, d7;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile("v_cvt_f64_u32 %0, %8\n v_cvt_f64_u32 %1, %8\n v_cvt_f64_u32 %2, %8\n v_cvt_f64_u32 %3, %8\n"
                   "v_cvt_f64_u32 %4, %8\n v_cvt_f64_u32 %5, %8\n v_cvt_f64_u32 %6, %8\n v_cvt_f64_u32 %7, %8\n"
                   : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7)
                   : "v"(a0));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_cvt_f32_f64(unsigned *out, unsigned x, unsigned long long *clk) {
  double a0 = (double)(threadIdx.x + x);
  float d0, d1, d2, d3, d4, d5, d6, d7;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %8\n v_cvt_f32_f64 %2, %8\n v_cvt_f32_f64 %3, %8\n"
                   "v_cvt_f32_f64 %4, %8\n v_cvt_f32_f64 %5, %8\n v_cvt_f32_f64 %6, %8\n v_cvt_f32_f64 %7, %8\n"
                   : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7)
                   : "v"(a0));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <typename K>
void run(const char *name, K kernel, unsigned *d_out, unsigned long long *d_clk) {
  const int waves_per_simd = 4;
  const int grid = 256 * waves_per_simd;   // 256 CUs, one 4-wave block per SIMD slot
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d_out, 3u, d_clk);
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d_out, 3u, d_clk);
  CHECK(hipDeviceSynchronize());
  unsigned long long clk = 0;
  CHECK(hipMemcpy(&clk, d_clk, sizeof clk, hipMemcpyDeviceToHost));
  const double instr = (double)kIters * 32.0 * waves_per_simd;   // wave-instructions per SIMD in clk cycles
  printf("%-14s %.2f cycles per wave-instruction per SIMD\n", name, (double)clk / instr);
}

int main() {
  unsigned *d_out;
  unsigned long long *d_clk;
  CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(unsigned)));
  CHECK(hipMalloc(&d_clk, 2 * sizeof(unsigned long long)));
  run("v_fma_f32", k_fma32, d_out, d_clk);
  run("v_max3_f32", k_max3, d_out, d_clk);
  run("v_cmp_lt_f32", k_cmp, d_out, d_clk);
  run("v_xor_b32", k_xor, d_out, d_clk);
  run("v_lshl_add_u32", k_lshladd, d_out, d_clk);
  run("v_mbcnt_lo", k_mbcnt, d_out, d_clk);
  run("v_mad_u32_u24", k_mad24, d_out, d_clk);
  run("v_mul_lo_u32", k_mul_lo, d_out, d_clk);
  run("v_mul_hi_u32", k_mul_hi, d_out, d_clk);
  run("v_rcp_f32", k_rcp32, d_out, d_clk);
  run("v_cvt_f32_u32", k_cvtf32u32, d_out, d_clk);
  run("v_cvt_f64_u32", k_cvt_f64_u32, d_out, d_clk);
  run("v_cvt_f32_f64", k_cvt_f32_f64, d_out, d_clk);
  run("v_fma_f64", k_fma64, d_out, d_clk);
  run("v_mul_f64", k_mul64, d_out, d_clk);
  run("v_add_f64", k_add64, d_out, d_clk);
  run("v_ldexp_f64", k_ldexp64, d_out, d_clk);
  run("v_rsq_f64", k_rsq64, d_out, d_clk);
  run("v_rcp_f64", k_rcp64, d_out, d_clk);
  return 0;
}
