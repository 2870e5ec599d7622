// Microbenchmark: the pipelined forward's exact per-gap instruction mix and cheaper variants of it (2 waves per SIMD).
// gap = 1 MFMA + {fma stage} + {2 v_exp_f32} + {row-sum stage} + {cvt_pk} [+ one LDS read]
// build: hipcc --offload-arch=gfx950 -O3 tools/mb_gap2.hip -o tools/bin/mb_gap2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// FMA: 0 none, 1 two v_fma_f32, 2 one v_pk_fma_f32, 3 two v_mul_f32
// ADD: 0 none, 1 two v_add_f32, 2 one v_pk_add_f32
// NE: number of v_exp_f32;  CVT: 0/1;  DS: 0 none, 1 ds_read_b128, 2 ds_read_b64_tr_b16
template <int FMA, int NE, int ADD, int CVT, int DS, bool DOMFMA>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  extern __shared__ char smem[];
  const int l = threadIdx.x;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x0 = 0.001f * l, x1 = 0.002f * l, e0 = -1.f, e1 = -2.f, l0 = 0.f, l1 = 0.f;
  f32x2 xp = {0.001f * l, 0.002f * l}, lp = {0.f, 0.f}, cp = {0.999f, 0.999f}, dp = {1e-4f, 1e-4f};
  unsigned w = 0;
  u32x4 ld = {0, 0, 0, 0};
  const float c = 0.999f, d = 0.0001f;
  const unsigned addr = (unsigned)(uintptr_t)smem + (l & 63) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if constexpr (DOMFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
      if constexpr (DS == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(addr));
      if constexpr (DS == 2) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(*(unsigned long long*)&ld) : "v"(addr));
      if constexpr (ADD == 1) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(l0) : "v"(e0));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(l1) : "v"(e1));
      } else if constexpr (ADD == 2) {
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(lp) : "v"(xp));
      }
      if constexpr (CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(e0), "v"(e1));
      if constexpr (NE >= 1) asm volatile("v_exp_f32 %0, %1" : "=v"(e0) : "v"(x0));
      if constexpr (NE >= 2) asm volatile("v_exp_f32 %0, %1" : "=v"(e1) : "v"(x1));
      if constexpr (FMA == 1) {
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x0) : "v"(acc[(g + 2) & 3][g]), "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x1) : "v"(acc[(g + 2) & 3][(g + 1) & 15]), "v"(c), "v"(d));
      } else if constexpr (FMA == 2) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xp) : "v"(cp), "v"(dp));
      } else if constexpr (FMA == 3) {
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x0) : "v"(acc[(g + 2) & 3][g]), "v"(c));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x1) : "v"(acc[(g + 2) & 3][(g + 1) & 15]), "v"(c));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (DS != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float r = x0 + x1 + e0 + e1 + l0 + l1 + xp[0] + lp[1] + (float)w + (float)ld[0];
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  if (r == 123.456f) out[l] = r;
}

template <int FMA, int NE, int ADD, int CVT, int DS, bool DOMFMA>
double run(int iters = 4000) {
  float* d;
  (void)hipMalloc(&d, 4096);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto kern = k<FMA, NE, ADD, CVT, DS, DOMFMA>;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 4096, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(d);
  return ms * 1e6 / ((double)iters * 16 * 2);
}

int main() {
  for (int i = 0; i < 3; ++i) run<0, 0, 0, 0, 0, true>();  // warm the clocks
  const double base = run<0, 0, 0, 0, 0, true>();
  printf("pure MFMA gap %.2f ns = 32 cycles (%.2f GHz); everything below in cycles per gap at that clock, 2 waves per SIMD\n", base, 32 / base);
#define R(name, ...) printf("  %-44s %6.1f   (no MFMA: %5.1f)\n", name, run<__VA_ARGS__, true>() / base * 32, run<__VA_ARGS__, false>() / base * 32)
  R("current: 2fma 2exp 2add cvt", 1, 2, 1, 1, 0);
  R("current + ds_read_b128", 1, 2, 1, 1, 1);
  R("current + ds_read_b64_tr", 1, 2, 1, 1, 2);
  R("pk_fma 2exp 2add cvt", 2, 2, 1, 1, 0);
  R("2fma 2exp pk_add cvt", 1, 2, 2, 1, 0);
  R("pk_fma 2exp pk_add cvt", 2, 2, 2, 1, 0);
  R("2mul 2exp 2add cvt", 3, 2, 1, 1, 0);
  R("2fma 2exp cvt (no adds)", 1, 2, 0, 1, 0);
  R("2exp 2add cvt (no fma)", 0, 2, 1, 1, 0);
  R("2exp cvt", 0, 2, 0, 1, 0);
  R("2fma 2add cvt (no exp)", 1, 0, 1, 1, 0);
  R("2fma 1exp 2add cvt", 1, 1, 1, 1, 0);
  R("2exp", 0, 2, 0, 0, 0);
  R("1exp", 0, 1, 0, 0, 0);
  R("cvt only", 0, 0, 0, 1, 0);
  R("ds_read_b128 only", 0, 0, 0, 0, 1);
  return 0;
}
