// Instantiations of the 64-rows-per-wave pipelined forward (attn_fwd64.h) for one head_dim (-DFAT5_INST_D=64).
#include "attn_fwd64.h"
#include <cstdlib>
#include <algorithm>
#include "attn_launch.h"

#ifndef FAT5_INST_D
#error "FAT5_INST_D must be defined"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

namespace fat5 {

#if FAT5_INST_D == 128
// head_dim 128 (round 5): 256-row workgroups, one wave per SIMD, bias none / rpe1d
template <bool BF16, int BIAS, bool SPREAD>
static hipError_t launch_w1s(const AttnArgs& a, int grid, hipStream_t s) {
  const size_t smem = Fwd64Cfg<128, false>::smem(a.R, BIAS);
  auto kern = attn_fwd64_w1_kernel<128, BF16, BIAS, SPREAD>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  AttnArgs am = a;
  fill_div_magic(am, grid);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, am);
  return hipGetLastError();
}
template <bool BF16, int BIAS>
static hipError_t launch_w1(const AttnArgs& a, int nw, int grid, hipStream_t s) {
  return nw == 5 ? launch_w1s<BF16, BIAS, true>(a, grid, s) : launch_w1s<BF16, BIAS, false>(a, grid, s);
}
// nw == 5: the ring requests spread over the MFMA gaps (grids of at most one round: see attn_fwd64_body)
hipError_t launch_fwd64_d128(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s) {
  if (bias == FAT5_BIAS_RPE1D) return bf16 ? launch_w1<true, FAT5_BIAS_RPE1D>(a, nw, grid, s) : launch_w1<false, FAT5_BIAS_RPE1D>(a, nw, grid, s);
  if (bias == FAT5_BIAS_NONE) return bf16 ? launch_w1<true, FAT5_BIAS_NONE>(a, nw, grid, s) : launch_w1<false, FAT5_BIAS_NONE>(a, nw, grid, s);
  if (bias == FAT5_BIAS_DENSE && bf16) {  // three ring slots + the two-tile bias ring: exactly 160 KB
    const size_t smem = Fwd64Cfg<128, false, true>::smem(0, FAT5_BIAS_DENSE);
    auto kern = attn_fwd64_w1_kernel<128, true, FAT5_BIAS_DENSE, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, a);  // (no division magic: the batch-inner decode of a shared bias divides by run-time values)
    return hipGetLastError();
  }
  return hipErrorInvalidValue;
}
#else
template <int D, bool BF16, int BIAS, bool KSPLIT>
static hipError_t launch64(const AttnArgs& a, int grid, hipStream_t s) {
  const size_t smem = Fwd64Cfg<D, KSPLIT>::smem(a.R, BIAS);
  auto kern = attn_fwd64_kernel<D, BF16, BIAS, KSPLIT>;
  if (smem > 48 * 1024) {  // (idempotent driver call; the library keeps no state of its own)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
  }
  // One workgroup per item (persistent workgroups walking the items were measured no faster at (4,12,8192,64): 1536 equal items
  // over 512 slots balance by themselves).
  AttnArgs am = a;
  fill_div_magic(am, grid);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, am);
  return hipGetLastError();
}

template <int D, bool BF16>
static hipError_t launch64_dense(const AttnArgs& a, int grid, hipStream_t s) {
  const size_t smem = Fwd64Cfg<D, false>::smem(a.R, FAT5_BIAS_DENSE);
  auto kern = attn_fwd64_dense_kernel<D, BF16>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, a);
  return hipGetLastError();
}

template <bool KSPLIT>
static hipError_t launch64_bias(const AttnArgs& a, int bf16, int bias, int grid, hipStream_t s) {
  if constexpr (!KSPLIT) {
    if (bias == FAT5_BIAS_DENSE) return bf16 ? launch64_dense<FAT5_INST_D, true>(a, grid, s) : launch64_dense<FAT5_INST_D, false>(a, grid, s);
  }
  if (bias == FAT5_BIAS_RPE1D)
    return bf16 ? launch64<FAT5_INST_D, true, FAT5_BIAS_RPE1D, KSPLIT>(a, grid, s) : launch64<FAT5_INST_D, false, FAT5_BIAS_RPE1D, KSPLIT>(a, grid, s);
  return bf16 ? launch64<FAT5_INST_D, true, FAT5_BIAS_NONE, KSPLIT>(a, grid, s) : launch64<FAT5_INST_D, false, FAT5_BIAS_NONE, KSPLIT>(a, grid, s);
}
// nw == 3: both workgroup forms in one launch (a.mix_*; grid = 256-row + 128-row workgroups)
template <bool BF16, int BIAS>
static hipError_t launch64_mixed(const AttnArgs& a, int grid, hipStream_t s) {
  const size_t smem = std::max(Fwd64Cfg<FAT5_INST_D, false>::smem(a.R, BIAS), Fwd64Cfg<FAT5_INST_D, true>::smem(a.R, BIAS));
  auto kern = attn_fwd64_mixed_kernel<FAT5_INST_D, BF16, BIAS>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  AttnArgs am = a;
  fill_div_magic(am, grid);
  {
    const long b_hi = (std::max<long>(a.M - 256L * (a.mix_a_lo + 1), 0) + 127) / 128, b_lo = (std::max<long>(a.M - 256L * a.mix_a_lo, 0) + 127) / 128;
    am.mg_mix[0] = div_magic(a.mix_a_lo + 1, grid); am.mg_mix[1] = div_magic(a.mix_a_lo, grid);
    am.mg_mix[2] = div_magic(b_hi, grid); am.mg_mix[3] = div_magic(b_lo, grid);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, am);
  return hipGetLastError();
}

// nw == 2: the key-split variant (two waves per 64 query rows, 128-row workgroups); otherwise 256-row workgroups
hipError_t CAT(launch_fwd64_d, FAT5_INST_D)(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s) {
  if (nw == 3) {
    if (bias == FAT5_BIAS_RPE1D) return bf16 ? launch64_mixed<true, FAT5_BIAS_RPE1D>(a, grid, s) : launch64_mixed<false, FAT5_BIAS_RPE1D>(a, grid, s);
    return bf16 ? launch64_mixed<true, FAT5_BIAS_NONE>(a, grid, s) : launch64_mixed<false, FAT5_BIAS_NONE>(a, grid, s);
  }
  return nw == 2 ? launch64_bias<true>(a, bf16, bias, grid, s) : launch64_bias<false>(a, bf16, bias, grid, s);
}

#endif

size_t CAT(smem_fwd64_d, FAT5_INST_D)(int R, int bias) {
#if FAT5_INST_D == 128
  if (bias == FAT5_BIAS_DENSE) return Fwd64Cfg<128, false, true>::smem(R, bias);
#endif
  return Fwd64Cfg<FAT5_INST_D>::smem(R, bias);
}

}  // namespace fat5
