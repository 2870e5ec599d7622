// Microbenchmark of the pipelined forward's issue model: a stream of "gaps" = one v_mfma_f32_32x32x16_bf16 (round-robin over
// NACC independent accumulators) + NV plain VALU ops + NE v_exp_f32, all independent inside a gap.
// Question: how many VALU ops hide under an MFMA, with the accumulators in arch VGPRs vs AGPRs, 1 vs 2 waves per SIMD?
// build: hipcc --offload-arch=gfx950 -O3 tools/mb_gap.hip -o /tmp/mb_gap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <bool AGPR>
__device__ __forceinline__ void mfma(f32x16& c, const u32x4& a, const u32x4& b) {
  if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <bool AGPR>
__device__ __forceinline__ void mfma_ab_agpr(f32x16& c, const u32x4& a, const u32x4& b) {  // operands in AGPRs too
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "a"(a), "a"(b));
}

// MODE: 0 acc in VGPR, 1 acc in AGPR (A/B in VGPR), 2 acc + A/B in AGPR
template <int MODE, int NV, int NE, int NACC, bool DOMFMA>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int l = threadIdx.x;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x[12], e[4];
  for (int j = 0; j < 12; ++j) x[j] = 0.001f * (l + j);
  for (int j = 0; j < 4; ++j) e[j] = -1.f - j;
  const float c = 0.999f, d = 0.0001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if constexpr (DOMFMA) {
        if constexpr (MODE == 2) mfma_ab_agpr<true>(acc[g % NACC], a, b);
        else mfma<MODE == 1>(acc[g % NACC], a, b);
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(c), "v"(d));
#pragma unroll
      for (int j = 0; j < NE; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0.f;
  for (int i = 0; i < NACC; ++i) r += acc[i][0];
  for (int j = 0; j < 12; ++j) r += x[j];
  for (int j = 0; j < 4; ++j) r += e[j];
  if (r == 123.456f) out[l] = r;
}

template <int MODE, int NV, int NE, int NACC, bool DOMFMA>
double run(int waves_per_simd, int iters) {
  float* d;
  hipMalloc(&d, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int threads = 256 * waves_per_simd;
  hipLaunchKernelGGL((k<MODE, NV, NE, NACC, DOMFMA>), dim3(256), dim3(threads), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NV, NE, NACC, DOMFMA>), dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  return ms * 1e6 / ((double)iters * 16 * waves_per_simd);  // ns per gap per SIMD (one wave's gap; with 2 waves: per issued gap)
}

template <int MODE, int NACC>
void sweep(const char* name, double ns_mfma) {
  const int iters = 4000;
  for (int w = 1; w <= 2; ++w) {
    printf("%-26s w/SIMD %d:", name, w);
    printf(" v0 %.2f", run<MODE, 0, 0, NACC, true>(w, iters) / ns_mfma);
    printf(" | v3 %.2f", run<MODE, 3, 0, NACC, true>(w, iters) / ns_mfma);
    printf(" | v5 %.2f", run<MODE, 5, 0, NACC, true>(w, iters) / ns_mfma);
    printf(" | v7 %.2f", run<MODE, 7, 0, NACC, true>(w, iters) / ns_mfma);
    printf(" | v9 %.2f", run<MODE, 9, 0, NACC, true>(w, iters) / ns_mfma);
    printf(" | v5e2 %.2f", run<MODE, 5, 2, NACC, true>(w, iters) / ns_mfma);
    printf(" | v3e2 %.2f", run<MODE, 3, 2, NACC, true>(w, iters) / ns_mfma);
    printf(" | v0e2 %.2f", run<MODE, 0, 2, NACC, true>(w, iters) / ns_mfma);
    printf("   (x 32 cycles)\n");
  }
}

int main() {
  const int iters = 4000;
  const double ns_mfma = run<1, 0, 0, 4, true>(1, iters);  // pure MFMA stream, 4 accumulators: 32 cycles per gap
  printf("pure MFMA gap: %.2f ns (=32 cycles -> %.2f GHz)\n", ns_mfma, 32.0 / ns_mfma);
  for (int w = 1; w <= 2; ++w) {
    printf("VALU only w/SIMD %d (in units of one MFMA = 32 cycles): v5 %.2f | v7 %.2f | v9 %.2f | v5e2 %.2f | v0e2 %.2f | v0e4 %.2f\n", w,
           run<0, 5, 0, 4, false>(w, iters) / ns_mfma, run<0, 7, 0, 4, false>(w, iters) / ns_mfma, run<0, 9, 0, 4, false>(w, iters) / ns_mfma,
           run<0, 5, 2, 4, false>(w, iters) / ns_mfma, run<0, 0, 2, 4, false>(w, iters) / ns_mfma, run<0, 0, 4, 4, false>(w, iters) / ns_mfma);
  }
  sweep<0, 4>("acc VGPR, 4 accs", ns_mfma);
  sweep<1, 4>("acc AGPR, 4 accs", ns_mfma);
  sweep<2, 4>("acc+AB AGPR, 4 accs", ns_mfma);
  sweep<0, 2>("acc VGPR, 2 accs", ns_mfma);
  sweep<1, 2>("acc AGPR, 2 accs", ns_mfma);
  sweep<0, 1>("acc VGPR, 1 acc (chain)", ns_mfma);
  sweep<1, 1>("acc AGPR, 1 acc (chain)", ns_mfma);
  return 0;
}
