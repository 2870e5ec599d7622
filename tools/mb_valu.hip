// Issue-cost microbenchmark of the VALU instructions the attention softmax uses (gfx950).
// One wave per SIMD (and two), long unrolled independent streams; reports cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = seed * (i + 1) + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      float& a = x[r & 15];
      float& b = x[(r + 5) & 15];
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
      if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
      if (OP == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a) : "v"(b));
      if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
      if (OP == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&x[(2 * r) & 14]) : "v"(*(double*)&x[(2 * r + 6) & 14]));
      if (OP == 5) asm volatile("v_exp_f16 %0, %0" : "+v"(a));
      if (OP == 6) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(a) : "v"(b));
      if (OP == 7) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a) : "v"(b));
      if (OP == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
      if (OP == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&x[(2 * r) & 14]) : "v"(*(double*)&x[(2 * r + 6) & 14]));
      if (OP == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
      if (OP == 11) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
      if (OP == 12) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a));
      if (OP == 13) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(a));
      if (OP == 14) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
      if (OP == 15) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a) : "v"(b));
      if (OP == 16) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a) : "v"(b));
      if (OP == 17) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b));
      if (OP == 18) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&x[(2 * r) & 14]) : "v"(*(double*)&x[(2 * r + 6) & 14]));
      if (OP == 19) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a) : "v"(b));
      if (OP == 20) asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(a));
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  if (s == 1.2345f) out[threadIdx.x] = s;
}
template <int OP>
void run(const char* name) {
  float* d; hipMalloc(&d, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int occ = 1; occ <= 2; ++occ) {
    const size_t lds = occ == 1 ? 100 * 1024 : 70 * 1024;
    hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<OP>, dim3(256 * occ), dim3(256), lds, 0, d, 10, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * occ), dim3(256), lds, 0, d, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s waves/SIMD %d: %7.3f ms  %6.2f ns per wave-instr-slot\n", name, occ, ms, ms * 1e6 / ((double)iters * REP * occ));
  }
}
int main() {
  run<8>("v_add_f32"); run<10>("v_mul_f32"); run<0>("v_fma_f32"); run<19>("v_fmac_f32"); run<3>("v_max3_f32"); run<17>("v_mov_b32");
  run<1>("v_exp_f32"); run<5>("v_exp_f16"); run<2>("v_cvt_pk_bf16_f32"); run<16>("v_cvt_pk_f16_f32"); run<7>("v_cvt_pkrtz_f16_f32");
  run<4>("v_pk_fma_f32"); run<9>("v_pk_mul_f32"); run<18>("v_pk_add_f32"); run<6>("v_pk_fma_f16"); run<11>("v_perm_b32");
  run<12>("v_cvt_f32_f16"); run<13>("v_lshlrev_b32"); run<14>("v_and_b32"); run<15>("v_ldexp_f32");
  return 0;
}
