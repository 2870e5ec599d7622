// Probe + issue-cost microbenchmark of the DPP diagonal sums (flasht5_amd/csrc/diag_sum.h), round 4.
//  A) correctness of the primitive against the host (random blocks, several consecutive steps, both forms: compiler-visible and pinned asm)
//  B) issue cost of its instructions inside an MFMA stream (the framework of mb_issue.hip) and of the dK/dV gap mix with them added
// build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -I include tools/mb_diag.hip -o tools/bin/mb_diag
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../flasht5_amd/csrc/diag_sum.h"
#include "../flasht5_amd/csrc/attn_fwd64.h"
using namespace fat5;

constexpr int STEPS = 5, OFF = 32 * (STEPS + 2) + 31;

template <bool ASM>
__global__ void probe(const float* X, float* out, int* cnt) {
  const int lane = threadIdx.x;
  DiagCarry c;
  diag_carry_zero(c);
  int base = 0;
  auto emit = [&](float F) {
    if (lane < 32) { out[base + lane + OFF] += F; cnt[base + lane + OFF] += 1; }
    base -= 32;
  };
  for (int st = 0; st < STEPS; ++st) {
    DiagStep s;
    diag_step_zero(s);
    float x[16];
    for (int r = 0; r < 16; ++r) x[r] = X[(st * 16 + r) * 64 + lane];
    if constexpr (!ASM) {
      static_for<16>([&](auto ri) { diag_elem<decltype(ri)::value>(s, x[decltype(ri)::value], lane & 15); });
    } else {
      float t[16];
      static_for<16>([&](auto ri) { constexpr int r = decltype(ri)::value; t[r] = diag_elem_mask<r>(x[r]); diag_elem_u<r>(s, x[r]); });
      asm volatile("s_nop 4" ::: "memory");
      static_for<16>([&](auto ri) { constexpr int r = decltype(ri)::value; diag_elem_b<r>(s, t[r]); });
      asm volatile("s_nop 4" : "+v"(s.u0), "+v"(s.u1), "+v"(s.b0), "+v"(s.b1));
    }
    emit(diag_finish(c, s, lane));
  }
  DiagStep z;
  diag_step_zero(z);
  emit(diag_finish(c, z, lane));
  emit(diag_finish(c, z, lane));
}

__global__ void lanes(int* o) {
  const int l = threadIdx.x;
  o[l] = __builtin_amdgcn_update_dpp(-1, l, 0x120 + 3, 0xf, 0xf, false);             // row_ror:3
  const auto s16 = __builtin_amdgcn_permlane16_swap((unsigned)l, (unsigned)(100 + l), false, false);
  o[64 + l] = s16[0]; o[128 + l] = s16[1];
  const auto s32 = __builtin_amdgcn_permlane32_swap((unsigned)l, (unsigned)(100 + l), false, false);
  o[192 + l] = s32[0]; o[256 + l] = s32[1];
}

// ---------------------------------------------------------------------------------------------------------------------------
enum { ADD, ADD_DPP, CND_S, FMAC_DPP, PERM16, MOV_DPP, NKIND };
static const char* kNames[NKIND] = {"v_add_f32", "v_add_f32_dpp row_ror", "v_cndmask_b32_e64 (sgpr mask)", "v_fmac_f32_dpp row_ror", "v_permlane16_swap_b32", "v_mov_b32_dpp row_ror"};
template <int KIND>
__device__ __forceinline__ void op(float& x, float& y, const float d) {
  const uint64_t m = 0x0007000700070007ull;
  if constexpr (KIND == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(d));
  else if constexpr (KIND == ADD_DPP) asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:5 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(d));
  else if constexpr (KIND == CND_S) asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(x) : "v"(d), "s"(m));
  else if constexpr (KIND == FMAC_DPP) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:5 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(d), "v"(y));
  else if constexpr (KIND == PERM16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  else if constexpr (KIND == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_ror:5 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(d));
}
template <int KIND, int N>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  const int l = threadIdx.x;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x[8], y[8];
  for (int j = 0; j < 8; ++j) { x[j] = 0.001f * (l + j); y[j] = 1.f + j; }
  float d = 0.0001f * l;
  asm volatile("" : "+v"(d));
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[g & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < N; ++j) op<KIND>(x[j], y[j], d);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  for (int j = 0; j < 8; ++j) r += x[j] + y[j];
  if (r == 123.456f) out[l] = r;
  if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND, int N>
double run() {
  static float* d = nullptr;
  static long long* c = nullptr;
  if (!d) { hipMalloc(&d, 4096); hipMalloc(&c, 64); }
  const int iters = 2000;
  k<KIND, N><<<256, 256>>>(d, c, iters);
  k<KIND, N><<<256, 256>>>(d, c, iters);
  hipDeviceSynchronize();
  long long h = 0;
  hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  return (double)h / (iters * 16.0);
}
template <int KIND>
void row() {
  const double g0 = run<KIND, 0>(), g2 = run<KIND, 2>(), g4 = run<KIND, 4>(), g8 = run<KIND, 8>();
  printf("%-32s gap %6.1f %6.1f %6.1f %6.1f (N = 0, 2, 4, 8) -> %5.2f ticks per instruction (4 -> 8)\n", kNames[KIND], g0, g2, g4, g8, (g8 - g4) / 4.0);
}

// the dK/dV gap (cvt mul exp fma + one LDS read) with the diagonal-sum ops of one element added
template <int SEQ>
__global__ __launch_bounds__(256) void kmix(float* out, long long* cyc, int iters) {
  const int l = threadIdx.x;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x[8], y[8];
  unsigned u[8];
  for (int j = 0; j < 8; ++j) { x[j] = 0.001f * (l + j); y[j] = -1.f - j; u[j] = l + j; }
  const float c = 0.999f, d = 0.0001f;
  __shared__ unsigned lds[8192];
  for (int i = l; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  u32x4 fr[4] = {a, a, a, a};
  const unsigned laddr = (unsigned)(size_t)(lds) + (l & 63) * 16;
  float U = 0.f, B = 0.f, T[2] = {0.f, 0.f};
  const uint64_t m = 0x0007000700070007ull;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[g & 3]) : "v"(a), "v"(b));
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[3]) : "v"(x[3]), "v"(y[3]));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[2]) : "v"(c));
      asm volatile("v_exp_f32 %0, %0" : "+v"(y[1]));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(c), "v"(d));
      if constexpr (SEQ >= 1) {
        asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:5 row_mask:0xf bank_mask:0xf" : "+v"(U) : "v"(x[4]));
        asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(T[g & 1]) : "v"(x[4]), "s"(m));
        asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:5 row_mask:0xf bank_mask:0xf" : "+v"(B) : "v"(T[(g + 1) & 1]));
      }
      if constexpr (SEQ == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(fr[g & 3]) : "v"(laddr));
      if constexpr (SEQ == 3) { asm volatile("ds_read_b128 %0, %1" : "=v"(fr[g & 3]) : "v"(laddr)); asm volatile("s_waitcnt lgkmcnt(2)"); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)");
  float r = U + B + T[0] + T[1];
  for (int i = 0; i < 4; ++i) r += acc[i][0] + __builtin_bit_cast(float, fr[i][0]);
  for (int j = 0; j < 8; ++j) r += x[j] + y[j] + (float)u[j];
  if (r == 123.456f) out[l] = r;
  if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int SEQ>
void mix(const char* what) {
  float* d; long long* c;
  hipMalloc(&d, 4096); hipMalloc(&c, 64);
  kmix<SEQ><<<256, 256>>>(d, c, 2000);
  kmix<SEQ><<<256, 256>>>(d, c, 2000);
  hipDeviceSynchronize();
  long long h = 0;
  hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("mix %-60s gap %6.1f (1 wave/SIMD)\n", what, (double)h / (2000 * 16.0));
  hipFree(d); hipFree(c);
}

int main() {
  {
    int* o; hipMalloc(&o, 320 * 4);
    lanes<<<1, 64>>>(o);
    std::vector<int> h(320);
    hipMemcpy(h.data(), o, 320 * 4, hipMemcpyDeviceToHost);
    printf("row_ror:3   lanes 0..19: "); for (int i = 0; i < 20; ++i) printf("%d ", h[i]); printf("\n");
    printf("perm16 r0   lanes 0,16,32,48: %d %d %d %d   r1: %d %d %d %d\n", h[64], h[64 + 16], h[64 + 32], h[64 + 48], h[128], h[128 + 16], h[128 + 32], h[128 + 48]);
    printf("perm32 r0   lanes 0,32: %d %d   r1: %d %d\n", h[192], h[192 + 32], h[256], h[256 + 32]);
  }
  const int NB = 2 * OFF + 64;
  std::vector<float> X(STEPS * 16 * 64);
  srand(7);
  for (auto& v : X) v = (float)(rand() % 2001 - 1000) / 256.f;
  std::vector<double> ref(NB, 0.0);
  for (int st = 0; st < STEPS; ++st)
    for (int r = 0; r < 16; ++r)
      for (int l = 0; l < 64; ++l) {
        const int hi = l >> 5, row = 32 * st + (r & 3) + 8 * (r >> 2) + 4 * hi, key = l & 31;
        ref[key - row + OFF] += X[(st * 16 + r) * 64 + l];
      }
  float *dX, *dO; int* dC;
  hipMalloc(&dX, X.size() * 4); hipMalloc(&dO, NB * 4); hipMalloc(&dC, NB * 4);
  hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  for (int variant = 0; variant < 2; ++variant) {
    hipMemset(dO, 0, NB * 4); hipMemset(dC, 0, NB * 4);
    if (variant == 0) probe<false><<<1, 64>>>(dX, dO, dC); else probe<true><<<1, 64>>>(dX, dO, dC);
    std::vector<float> o(NB); std::vector<int> cn(NB);
    hipMemcpy(o.data(), dO, NB * 4, hipMemcpyDeviceToHost); hipMemcpy(cn.data(), dC, NB * 4, hipMemcpyDeviceToHost);
    double worst = 0; int multi = 0, bad = 0;
    for (int i = 0; i < NB; ++i) {
      worst = std::max(worst, std::fabs(o[i] - ref[i]));
      if (cn[i] > 1) ++multi;
      if (std::fabs(o[i] - ref[i]) > 1e-3) { if (bad < 8) printf("  bin %d: got %f want %f (writes %d)\n", i - OFF, o[i], ref[i], cn[i]); ++bad; }
    }
    printf("diag probe (%s): max |err| %.3g, bins written more than once %d, wrong bins %d -> %s\n", variant ? "pinned asm" : "compiler-visible", worst, multi, bad,
           (bad == 0 && multi == 0) ? "OK" : "FAIL");
  }
  row<ADD>(); row<ADD_DPP>(); row<CND_S>(); row<FMAC_DPP>(); row<PERM16>(); row<MOV_DPP>();
  mix<0>("dK/dV gap: cvt mul exp fma");
  mix<1>("dK/dV gap + add_dpp cndmask add_dpp (one element's diagonal ops)");
  mix<2>("... + ds_read_b128");
  mix<3>("... + ds_read_b128 + lgkmcnt(2)");
  return 0;
}
