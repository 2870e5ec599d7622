// Microbenchmark: the forward's per-block dependency structure (S chain -> softmax VALU -> PV MFMAs), fully dependent
// inside a wave, 1 / 2 / 3 waves per SIMD, MFMA operands from registers or from LDS.  How much of the VALU time do
// co-resident waves hide under each other's MFMAs?
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mb_dep.hip -o tools/mb_dep.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;

__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pk(float x, float y) { bf2 v = {(__bf16)x, (__bf16)y}; return __builtin_bit_cast(unsigned, v); }

// WHAT: 0 full block, 1 MFMA only (VALU removed, pb constant), 2 VALU only (MFMAs removed)
template <int WHAT, bool LDSOPS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  u32x4* lds = reinterpret_cast<u32x4*>(smem) + w * 1024;
  for (int i = l; i < 1024; i += 64) lds[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  __syncthreads();
  u32x4 q[4], kf[4], vf[4];
  for (int i = 0; i < 4; ++i) { q[i] = u32x4{0x3c003c00u, 0x3c003c00u + l, 0x3c003c00u, 0x3c003c00u}; kf[i] = lds[l + 64 * i]; vf[i] = lds[l + 64 * (i + 4)]; }
  f32x16 o0 = {0}, o1 = {0};
  f32x2 lsum = {0.f, 0.f};
  const f32x16 z = {0};
  float m = 0.25f;
  for (int it = 0; it < iters; ++it) {
    if (LDSOPS) for (int i = 0; i < 4; ++i) kf[i] = lds[l + 64 * ((it + i) & 7)];
    f32x16 s = z;
    if (WHAT != 2) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) s = mf(kf[kk], q[kk], kk == 0 ? z : s);
    } else {
      for (int r = 0; r < 16; ++r) s[r] = o0[r] * 1e-3f + (float)it;
    }
    u32x4 pb[2];
    if (WHAT != 1) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 x = {s[r], s[r + 1]};
        x = x * 0.18f + (-m);
        x[0] = __builtin_amdgcn_exp2f(x[0]);
        x[1] = __builtin_amdgcn_exp2f(x[1]);
        lsum += x;
        s[r] = x[0]; s[r + 1] = x[1];
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) pb[t][j] = pk(s[8 * t + 2 * j], s[8 * t + 2 * j + 1]);
    } else {
      pb[0] = u32x4{__float_as_uint(s[0]), __float_as_uint(s[5]), 0x3c003c00u, 0x3c003c00u};
      pb[1] = u32x4{__float_as_uint(s[9]), __float_as_uint(s[13]), 0x3c003c00u, 0x3c003c00u};
    }
    if (LDSOPS) for (int i = 0; i < 4; ++i) vf[i] = lds[l + 64 * ((it + i + 3) & 7)];
    if (WHAT != 2) {
      o0 = mf(vf[0], pb[0], o0);
      o1 = mf(vf[1], pb[0], o1);
      o0 = mf(vf[2], pb[1], o0);
      o1 = mf(vf[3], pb[1], o1);
    } else {
      o0[0] += __uint_as_float(pb[0][0] ^ pb[1][3]); o1[1] += __uint_as_float(pb[0][2] ^ pb[1][1]);
      o0[2] += __uint_as_float(pb[0][1] ^ pb[1][2]); o1[3] += __uint_as_float(pb[0][3] ^ pb[1][0]);
    }
  }
  float r = o0[0] + o1[1] + lsum[0] + lsum[1] + o0[7] + o1[9];
  if (r == 123.456f) out[threadIdx.x] = r;
}


// Software-pipelined variant: S of block j+1 is formed while the VALU work of block j runs (independent streams inside
// one wave, interleaved by sched_group_barrier); the P.V MFMAs of block j follow.
template <int ILV>
__global__ __launch_bounds__(256) void kp(float* out, int iters) {
  const int l = threadIdx.x & 63;
  u32x4 q[4], kf[4], vf[4];
  for (int i = 0; i < 4; ++i) { q[i] = u32x4{0x3c003c00u, 0x3c003c00u + l, 0x3c003c00u, 0x3c003c00u}; kf[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u + l, 0x3c003c00u}; vf[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u + l + i, 0x3c003c00u}; }
  f32x16 o0 = {0}, o1 = {0};
  f32x2 lsum = {0.f, 0.f};
  const f32x16 z = {0};
  float m = 0.25f;
  f32x16 s = z;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) s = mf(kf[kk], q[kk], kk == 0 ? z : s);
  for (int it = 0; it < iters; ++it) {
    f32x16 sn = z;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) sn = mf(kf[kk], q[kk], kk == 0 ? z : sn);
    u32x4 pb[2];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      f32x2 x = {s[r], s[r + 1]};
      x = x * 0.18f + (-m);
      x[0] = __builtin_amdgcn_exp2f(x[0]);
      x[1] = __builtin_amdgcn_exp2f(x[1]);
      lsum += x;
      s[r] = x[0]; s[r + 1] = x[1];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) pb[t][j] = pk(s[8 * t + 2 * j], s[8 * t + 2 * j + 1]);
    o0 = mf(vf[0], pb[0], o0);
    o1 = mf(vf[1], pb[0], o1);
    o0 = mf(vf[2], pb[1], o0);
    o1 = mf(vf[3], pb[1], o1);
    if (ILV) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);   // 1 MFMA (S chain)
        __builtin_amdgcn_sched_group_barrier(0x2, 7, 0);   // 7 VALU
      }
      __builtin_amdgcn_sched_group_barrier(0x2, 12, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);     // P.V
    }
    s = sn;
    q[0][0] ^= (pb[0][0] & 1);  // keep the S chain from being hoisted out of the loop
  }
  float r = o0[0] + o1[1] + lsum[0] + lsum[1] + s[3];
  if (r == 123.456f) out[threadIdx.x] = r;
}
template <int ILV>
double runp(int occ, int iters) {
  float* d; (void)hipMalloc(&d, 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t lds = occ == 1 ? 100 * 1024 : (occ == 2 ? 70 * 1024 : 50 * 1024);
  (void)hipFuncSetAttribute((const void*)kp<ILV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((kp<ILV>), dim3(256 * occ), dim3(256), lds, 0, d, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((kp<ILV>), dim3(256 * occ), dim3(256), lds, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / iters / occ;
}

// Ping-pong variant: 8 waves per workgroup = two groups of 4 (one wave of each group per SIMD), forced into opposite
// phases with s_barrier: while group A runs its MFMA phase (P.V of the previous block + the score chain of the next),
// group B runs its softmax VALU phase, then they swap.  Does the SIMD overlap wave A's MFMAs with wave B's VALU work?
__global__ __launch_bounds__(512) void kpp(float* out, int iters) {
  const int l = threadIdx.x & 63, grp = (threadIdx.x >> 6) >> 2;
  u32x4 q[4], kf[4], vf[4];
  for (int i = 0; i < 4; ++i) { q[i] = u32x4{0x3c003c00u, 0x3c003c00u + l, 0x3c003c00u, 0x3c003c00u}; kf[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u + l, 0x3c003c00u}; vf[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u + l + i, 0x3c003c00u}; }
  f32x16 o0 = {0}, o1 = {0};
  f32x2 lsum = {0.f, 0.f};
  const f32x16 z = {0};
  float m = 0.25f;
  f32x16 s = z;
  u32x4 pb[2] = {u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}};
  auto mfma_phase = [&]() {
    o0 = mf(vf[0], pb[0], o0);
    o1 = mf(vf[1], pb[0], o1);
    o0 = mf(vf[2], pb[1], o0);
    o1 = mf(vf[3], pb[1], o1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) s = mf(kf[kk], q[kk], kk == 0 ? z : s);
  };
  auto valu_phase = [&]() {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      f32x2 x = {s[r], s[r + 1]};
      x = x * 0.18f + (-m);
      x[0] = __builtin_amdgcn_exp2f(x[0]);
      x[1] = __builtin_amdgcn_exp2f(x[1]);
      lsum += x;
      s[r] = x[0]; s[r + 1] = x[1];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) pb[t][j] = pk(s[8 * t + 2 * j], s[8 * t + 2 * j + 1]);
    q[0][0] ^= (pb[0][0] & 1);
  };
  if (grp == 1) { mfma_phase(); }   // group B starts one phase ahead
  __builtin_amdgcn_s_barrier();
  for (int it = 0; it < iters; ++it) {
    if (grp == 0) mfma_phase(); else valu_phase();
    __builtin_amdgcn_s_barrier();
    if (grp == 0) valu_phase(); else mfma_phase();
    __builtin_amdgcn_s_barrier();
  }
  float r = o0[0] + o1[1] + lsum[0] + lsum[1] + s[3];
  if (r == 123.456f) out[threadIdx.x] = r;
}
double runpp(int iters) {
  float* d; (void)hipMalloc(&d, 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t lds = 100 * 1024;  // one 8-wave workgroup per CU
  (void)hipFuncSetAttribute((const void*)kpp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kpp, dim3(256), dim3(512), lds, 0, d, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kpp, dim3(256), dim3(512), lds, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / iters / 2;  // two blocks (one per group) per iteration per SIMD
}

template <int WHAT, bool LDSOPS>
double run(int occ, int iters) {
  float* d; (void)hipMalloc(&d, 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t lds = occ == 1 ? 100 * 1024 : (occ == 2 ? 70 * 1024 : 50 * 1024);
  (void)hipFuncSetAttribute((const void*)k<WHAT, LDSOPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<WHAT, LDSOPS>), dim3(256 * occ), dim3(256), lds, 0, d, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<WHAT, LDSOPS>), dim3(256 * occ), dim3(256), lds, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / iters / occ;  // ns per block per wave-slot share of a SIMD
}

int main() {
  const int iters = 200000;   // long enough to reach the sustained clock
  printf("ns per 32x32-key block per wave (8 MFMAs = 256 cycles; at 1.95-2.4 GHz that is 107-131 ns)\n");
  for (int occ = 1; occ <= 3; ++occ) {
    printf("waves/SIMD %d | regs: full %.1f  mfma-only %.1f  valu-only %.1f | lds operands: full %.1f  mfma-only %.1f\n", occ,
           run<0, false>(occ, iters), run<1, false>(occ, iters), run<2, false>(occ, iters), run<0, true>(occ, iters), run<1, true>(occ, iters));
    printf("             | software-pipelined (regs): compiler order %.1f  forced 1:7 interleave %.1f\n", runp<0>(occ, iters), runp<1>(occ, iters));
  }
  printf("ping-pong (2 waves/SIMD in forced opposite phases, s_barrier): %.1f ns per block\n", runpp(iters));
  return 0;
}
