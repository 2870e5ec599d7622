// Issue cost of single VALU instructions inside an MFMA stream (gfx950), the number the hand-placed gap schedules of attn_fwd64.h /
// attn_bwd64.h are built on: gap = one v_mfma_f32_32x32x16_bf16 (AGPR accumulators, round-robin over 4) + N copies of instruction
// X on independent registers; cycles per gap for N = 0, 4, 8 -> slope = issue cycles per X.  1 and 2 waves per SIMD (the 2-wave
// column is wave 0's own time: the oldest wave of a SIMD has issue priority, so it is a latency, not a throughput, number).
// build: hipcc --offload-arch=gfx950 -O3 tools/mb_issue.hip -o /tmp/mb_issue && /tmp/mb_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

enum { FMA, FMAC, MUL, ADD, EXP, CVTPK, DOT2C, PKMUL, PKADD, MAX3, PERM, MFMA16, LSHL, NOP0, NKIND };
static const char* kNames[NKIND] = {"v_fma_f32 (VOP3)", "v_fmac_f32 (VOP2)", "v_mul_f32", "v_add_f32", "v_exp_f32", "v_cvt_pk_bf16_f32",
                                    "v_dot2c_f32_bf16", "v_pk_mul_f32", "v_pk_add_f32", "v_max3_f32", "v_perm_b32",
                                    "v_mfma_f32_16x16x32_bf16", "v_lshlrev_b32", "s_nop 0"};

template <int KIND>
__device__ __forceinline__ void op(float& x, float& y, f32x2& p, f32x4& q, unsigned& u, const float c, const float d, const u32x4& a, const u32x4& b) {
  if constexpr (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
  else if constexpr (KIND == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
  else if constexpr (KIND == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c));
  else if constexpr (KIND == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(d));
  else if constexpr (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(y));
  else if constexpr (KIND == CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(x), "v"(y));
  else if constexpr (KIND == DOT2C) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x) : "v"(u), "v"(a[0]));
  else if constexpr (KIND == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(p));
  else if constexpr (KIND == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(p));
  else if constexpr (KIND == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
  else if constexpr (KIND == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u) : "v"(a[1]), "v"(b[1]));
  else if constexpr (KIND == MFMA16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(q) : "v"(a), "v"(b));
  else if constexpr (KIND == LSHL) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u));
  else if constexpr (KIND == NOP0) asm volatile("s_nop 0");
}

template <int KIND, int N>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  const int l = threadIdx.x;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x[8], y[8];
  f32x2 p[8];
  f32x4 q[8];
  unsigned u[8];
  for (int j = 0; j < 8; ++j) { x[j] = 0.001f * (l + j); y[j] = -1.f - j; p[j] = f32x2{x[j], y[j]}; q[j] = f32x4{0.f, 0.f, 0.f, 0.f}; u[j] = l + j; }
  const float c = 0.999f, d = 0.0001f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[g & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < N; ++j) op<KIND>(x[j], y[j], p[j], q[j], u[j], c, d, a, b);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  for (int j = 0; j < 8; ++j) r += x[j] + y[j] + p[j][0] + q[j][0] + (float)u[j];
  if (r == 123.456f) out[l] = r;
  if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int N>
double run(int wps) {
  static float* d = nullptr;
  static long long* c = nullptr;
  if (!d) { hipMalloc(&d, 4096); hipMalloc(&c, 64); }
  const int iters = 2000;
  k<KIND, N><<<256, 256 * wps>>>(d, c, iters);
  k<KIND, N><<<256, 256 * wps>>>(d, c, iters);
  hipDeviceSynchronize();
  long long h = 0;
  hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  return (double)h / (iters * 16.0);
}


// gap = MFMA + a fixed sequence of ops on independent registers (the backward's per-gap mix, in different orders / encodings)
template <int SEQ>
__global__ __launch_bounds__(512) void kmix(float* out, long long* cyc, int iters) {
  const int l = threadIdx.x;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x[8], y[8];
  f32x2 p[8];
  f32x4 q[8];
  unsigned u[8];
  for (int j = 0; j < 8; ++j) { x[j] = 0.001f * (l + j); y[j] = -1.f - j; p[j] = f32x2{x[j], y[j]}; q[j] = f32x4{0.f, 0.f, 0.f, 0.f}; u[j] = l + j; }
  const float c = 0.999f, d = 0.0001f;
  __shared__ unsigned lds[8192];
  for (int i = l; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  u32x4 fr[4] = {a, a, a, a};
  u32x2 frh[4][2];
  for (int i = 0; i < 4; ++i) { frh[i][0] = u32x2{1u, 2u}; frh[i][1] = u32x2{3u, 4u}; }
  const unsigned laddr = (unsigned)(size_t)(lds) + (l & 63) * 16;
  const long long t0 = __builtin_readcyclecounter();
#define OP(K, j) op<K>(x[j], y[j], p[j], q[j], u[j], c, d, a, b)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[g & 3]) : "v"(a), "v"(b));
      if constexpr (SEQ == 0) { OP(FMA, 0); OP(EXP, 1); OP(MUL, 2); OP(CVTPK, 3); }
      else if constexpr (SEQ == 1) { OP(EXP, 1); OP(FMA, 0); OP(MUL, 2); OP(CVTPK, 3); }
      else if constexpr (SEQ == 2) { OP(CVTPK, 3); OP(MUL, 2); OP(EXP, 1); OP(FMA, 0); }
      else if constexpr (SEQ == 3) { OP(FMA, 0); OP(EXP, 1); OP(MUL, 2); OP(PERM, 3); }
      else if constexpr (SEQ == 4) { OP(FMAC, 0); OP(EXP, 1); OP(MUL, 2); OP(CVTPK, 3); }
      else if constexpr (SEQ == 5) { OP(FMA, 0); OP(EXP, 1); OP(MUL, 2); }
      else if constexpr (SEQ == 6) { OP(FMA, 0); OP(MUL, 2); OP(CVTPK, 3); }
      else if constexpr (SEQ == 7) { OP(CVTPK, 3); OP(EXP, 1); OP(MUL, 2); OP(FMA, 0); }
      else if constexpr (SEQ == 8) { OP(EXP, 1); OP(MUL, 2); OP(FMA, 0); OP(CVTPK, 3); }
      else if constexpr (SEQ == 9) { OP(FMA, 0); OP(FMA, 4); OP(EXP, 1); OP(EXP, 5); OP(MUL, 2); OP(MUL, 6); OP(CVTPK, 3); }   // q64-like: 2 elements, 1 pack
      else if constexpr (SEQ == 10) { OP(FMA, 0); OP(FMA, 4); OP(EXP, 1); OP(EXP, 5); OP(ADD, 2); OP(ADD, 6); OP(CVTPK, 3); }  // fwd-like
      else if constexpr (SEQ == 11) { OP(FMA, 0); OP(FMA, 4); OP(EXP, 1); OP(EXP, 5); OP(CVTPK, 3); }                         // fwd, sums on MFMA
      else if constexpr (SEQ >= 12 && SEQ <= 16) {
        // the dK/dV gap with its LDS traffic: one 16-byte read per gap into a 4-deep ring of fragments
        OP(CVTPK, 3); OP(MUL, 2); OP(EXP, 1); OP(FMA, 0);
        if constexpr (SEQ == 12 || SEQ == 14 || SEQ == 15) asm volatile("ds_read_b128 %0, %1" : "=v"(fr[g & 3]) : "v"(laddr));
        if constexpr (SEQ == 13 || SEQ == 16) asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:512" : "=&v"(frh[g & 3][0]), "=&v"(frh[g & 3][1]) : "v"(laddr));
        if constexpr (SEQ == 14 || SEQ == 16) asm volatile("s_waitcnt lgkmcnt(2)");
        if constexpr (SEQ == 15) asm volatile("s_waitcnt lgkmcnt(0)");
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)");
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + __builtin_bit_cast(float, fr[i][0]) + __builtin_bit_cast(float, frh[i][0][0]) + __builtin_bit_cast(float, frh[i][1][0]);
  for (int j = 0; j < 8; ++j) r += x[j] + y[j] + p[j][0] + q[j][0] + (float)u[j];
  if (r == 123.456f) out[l] = r;
  if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int SEQ>
void mix(const char* what) {
  float* d; long long* c;
  hipMalloc(&d, 4096); hipMalloc(&c, 64);
  double g[2];
  for (int wps = 1; wps <= 2; ++wps) {
    kmix<SEQ><<<256, 256 * wps>>>(d, c, 2000);
    kmix<SEQ><<<256, 256 * wps>>>(d, c, 2000);
    hipDeviceSynchronize();
    long long h = 0;
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    g[wps - 1] = (double)h / (2000 * 16.0);
  }
  printf("mix %-44s gap %6.1f (1 wave/SIMD)  %6.1f per wave (2 waves/SIMD)\n", what, g[0], g[1]);
  hipFree(d); hipFree(c);
}

// ---- throughput at TWO waves per SIMD (the forward's regime), wall clock over the whole launch: ns per MFMA gap and SIMD ----
template <int SEQ>
__global__ __launch_bounds__(512) void kfwd(float* out, int iters) {
  const int l = threadIdx.x;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x[8], y[8];
  f32x2 p[8];
  f32x4 q[8];
  unsigned u[8];
  for (int j = 0; j < 8; ++j) { x[j] = 0.001f * (l + j); y[j] = -1.f - j; p[j] = f32x2{x[j], y[j]}; q[j] = f32x4{0.f, 0.f, 0.f, 0.f}; u[j] = l + j; }
  const float c = 0.999f, d = 0.0001f;
  __shared__ unsigned lds[8192];
  for (int i = l; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  u32x4 fr[4] = {a, a, a, a};
  const unsigned laddr = (unsigned)(size_t)(lds) + (l & 63) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
      if constexpr (SEQ == 1) { OP(FMA, 0); OP(FMA, 4); OP(EXP, 1); OP(EXP, 5); OP(ADD, 2); OP(ADD, 6); OP(CVTPK, 3); }
      else if constexpr (SEQ >= 2 && SEQ <= 4) { OP(FMA, 0); OP(FMA, 4); OP(EXP, 1); OP(EXP, 5); OP(CVTPK, 3); if ((g & 3) == 3) OP(MFMA16, 7); }
      else if constexpr (SEQ == 5) { OP(MUL, 0); OP(MUL, 4); OP(EXP, 1); OP(EXP, 5); OP(CVTPK, 3); if ((g & 3) == 3) OP(MFMA16, 7); }
      else if constexpr (SEQ == 6) { OP(EXP, 1); OP(EXP, 5); }
      else if constexpr (SEQ == 7) { OP(FMA, 0); OP(FMA, 4); OP(EXP, 1); OP(EXP, 5); }
      else if constexpr (SEQ == 8) { OP(FMA, 0); OP(FMA, 4); OP(EXP, 1); OP(EXP, 5); OP(PERM, 3); if ((g & 3) == 3) OP(MFMA16, 7); }
      if constexpr (SEQ == 3 || SEQ == 4) { if ((g & 3) != 3) asm volatile("ds_read_b128 %0, %1" : "=v"(fr[g & 3]) : "v"(laddr)); }
      if constexpr (SEQ == 4) { if ((g & 1) == 0) asm volatile("s_waitcnt lgkmcnt(2)"); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + __builtin_bit_cast(float, fr[i][0]);
  for (int j = 0; j < 8; ++j) r += x[j] + y[j] + p[j][0] + q[j][0] + (float)u[j];
  if (r == 123.456f) out[l] = r;
}
template <int SEQ>
void fwdmix(const char* what) {
  float* d;
  hipMalloc(&d, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int wps = 1; wps <= 2; ++wps) {
    kfwd<SEQ><<<256, 256 * wps>>>(d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kfwd<SEQ><<<256, 256 * wps>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("fwd-mix %-52s %d wave/SIMD: %6.2f ns per MFMA and SIMD (%5.1f cycles at 2.4 GHz)\n", what, wps, ms * 1e6 / (iters * 16.0 * wps),
           ms * 1e6 / (iters * 16.0 * wps) * 2.4);
  }
  hipFree(d);
}

template <int KIND>
void row() {
  for (int wps = 1; wps <= 2; ++wps) {
    const double g0 = run<KIND, 0>(wps), g4 = run<KIND, 4>(wps), g8 = run<KIND, 8>(wps);
    printf("%-26s %d wave/SIMD: gap %6.1f %6.1f %6.1f (N = 0, 4, 8) -> %5.2f shader-clock ticks per instruction (4 -> 8)\n", kNames[KIND], wps, g0, g4, g8,
           (g8 - g4) / 4.0);
  }
}

int main() {
  printf("(ticks of s_memtime's 100 MHz-class counter scaled by the runtime; compare rows, not absolute values)\n");
  row<FMA>(); row<FMAC>(); row<MUL>(); row<ADD>(); row<EXP>(); row<CVTPK>(); row<DOT2C>(); row<PKMUL>(); row<PKADD>(); row<MAX3>(); row<PERM>();
  row<MFMA16>(); row<LSHL>(); row<NOP0>();
  mix<0>("fma exp mul cvt"); mix<1>("exp fma mul cvt"); mix<2>("cvt mul exp fma (dK/dV body order)"); mix<7>("cvt exp mul fma"); mix<8>("exp mul fma cvt");
  mix<3>("fma exp mul perm"); mix<4>("fmac exp mul cvt"); mix<5>("fma exp mul"); mix<6>("fma mul cvt");
  mix<9>("2fma 2exp 2mul cvt (dQ body)"); mix<10>("2fma 2exp 2add cvt (forward)"); mix<11>("2fma 2exp cvt (forward, sums on MFMA)");
  mix<12>("dK/dV gap + ds_read_b128"); mix<13>("dK/dV gap + 2 ds_read_b64"); mix<14>("dK/dV gap + ds_read_b128 + lgkmcnt(2)");
  mix<15>("dK/dV gap + ds_read_b128 + lgkmcnt(0)"); mix<16>("dK/dV gap + 2 ds_read_b64 + lgkmcnt(2)");
  fwdmix<0>("MFMA only"); fwdmix<6>("2 exp"); fwdmix<7>("2 fma 2 exp"); fwdmix<1>("2 fma 2 exp 2 add cvt (round-2 forward gap)");
  fwdmix<2>("2 fma 2 exp cvt + MFMA16 per 4 gaps"); fwdmix<3>("... + 3 ds_read_b128 per 4 gaps"); fwdmix<4>("... + s_waitcnt per 2 gaps");
  fwdmix<5>("2 mul 2 exp cvt + MFMA16 per 4 gaps"); fwdmix<8>("2 fma 2 exp perm + MFMA16 per 4 gaps");
  return 0;
}
