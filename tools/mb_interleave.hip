// Microbenchmark: can one wave overlap MFMA and VALU (exp/fma/cvt) work, and how do 1 / 2 / 3 waves per SIMD behave
// when each wave's instruction stream is phase-separated vs interleaved?
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mb_interleave.hip -o /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>  // 0: MFMA phase then VALU phase; 1: interleaved by sched_group_barrier; 2: MFMA only; 3: VALU only
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ char smem[];
  const int l = threadIdx.x;
  u32x4 a = {0x3f803f80u + l, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u + l, 0x3c003c00u, 0x3c003c00u};
  f32x16 s = {0}, o0 = {0}, o1 = {0}, o2 = {0}, x = {0};
  for (int r = 0; r < 16; ++r) x[r] = 0.001f * (l + r);
  float c = 0.5f, m = 0.25f;
  for (int it = 0; it < iters; ++it) {
    // MFMA work: 4-chain + 6 independent-ish
    if (MODE == 4 || MODE == 6) __builtin_amdgcn_s_setprio(0);
    if (MODE == 5) __builtin_amdgcn_s_setprio(2);
    if (MODE == 7 || MODE == 8) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (MODE == 8) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(s) : "v"(a), "v"(b));
        else s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), s, 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o0) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o1) : "v"(b), "v"(a));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o2) : "v"(a), "v"(a));
      }
    } else if (MODE != 3) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), s, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, b), __builtin_bit_cast(bf16x8_t, a), o1, 0, 0, 0);
        o2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, a), o2, 0, 0, 0);
      }
    }
    // VALU work on x (independent of this iteration's MFMAs): 8 max3-ish, 16 fma, 16 exp, 8 cvt
    if (MODE == 4) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(2); }
    if (MODE == 5) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0); }
    if (MODE == 6) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(3); }
    if (MODE != 2) {
      float mx = x[0];
#pragma unroll
      for (int r = 1; r < 16; r += 2) mx = fmaxf(fmaxf(mx, x[r]), x[(r + 1) & 15]);
      m = fmaxf(m, mx * 1e-6f);
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_exp2f(fmaf(x[r], c, -m));
      u32x4 p;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
        bf2 v0 = {(__bf16)x[4 * r], (__bf16)x[4 * r + 1]};
        bf2 v1 = {(__bf16)x[4 * r + 2], (__bf16)x[4 * r + 3]};
        p[r] = __builtin_bit_cast(unsigned, v0) ^ __builtin_bit_cast(unsigned, v1);
      }
      b[0] ^= p[0] & 1; b[1] ^= p[1] & 1; b[2] ^= p[2] & 1; b[3] ^= p[3] & 1;
    }
    if (MODE == 1 || MODE == 8) {
#pragma unroll
      for (int g = 0; g < 10; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x2, 5, 0);   // 5 VALU
      }
    } else if (MODE == 0 || MODE >= 4) {
      __builtin_amdgcn_sched_group_barrier(0x8, 10, 0);
      __builtin_amdgcn_sched_group_barrier(0x2, 60, 0);
    }
  }
  float r = s[0] + o0[1] + o1[2] + o2[3] + x[4] + m;
  if (r == 123.456f) out[l] = r;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, int iters) {
  float* d; hipMalloc(&d, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  const size_t lds = blocks_per_cu == 1 ? 100 * 1024 : (blocks_per_cu == 2 ? 70 * 1024 : 50 * 1024);
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: blocks_per_cu waves, each iters * 10 MFMAs
  const double cyc_per_iter_per_wave = ms * 1e-3 * 2.1e9 / iters;  // assuming ~2.1 GHz
  printf("%-28s waves/SIMD %d: %.3f ms  -> %.0f cycles/iter/SIMD-slot (MFMA pipe needs %d/iter/wave)\n", name, blocks_per_cu, ms,
         cyc_per_iter_per_wave, 320);
}

int main() {
  const int iters = 20000;
  for (int occ = 1; occ <= 3; ++occ) {
    run<2>("mfma only", occ, iters);
    run<3>("valu only", occ, iters);
    run<0>("phase separated", occ, iters);
    run<1>("interleaved 1:5", occ, iters);
    run<7>("asm agpr acc (o only)", occ, iters);
    run<8>("asm agpr acc (all)", occ, iters);
  }
  return 0;
}
