// Instantiations of the 64-keys-per-wave pipelined dK/dV body (attn_bwd64.h) for one head_dim (-DFAT5_INST_D=64).
#include "attn_bwd64.h"
#include "attn_launch.h"
#include <algorithm>
#include <cstring>

#ifndef FAT5_INST_D
#error "FAT5_INST_D must be defined"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

namespace fat5 {

// 1 / scale representable in the 16-bit operand dtype: the one-term selector of the dense body
static inline bool inv_scale_is_16bit(float scale, bool bf16) { return bf16 ? is_one16<true>(1.f / scale) : is_one16<false>(1.f / scale); }
template <int D, bool BF16, int BIAS, bool HALF, bool ONE = false, bool NODIAG = false>
static hipError_t launch_kv64(const AttnArgs& a, int grid, hipStream_t s) {
  // (operands / outputs through LDS images whenever the workgroup's LDS allows: see BwdQ64Cfg)
  AttnArgs as = a;
  fill_div_magic(as, grid);
  using Cfg = Bwd64Cfg<D, HALF, false, BIAS == FAT5_BIAS_DENSE>;
  as.lds_stage = Cfg::smem(a.R, BIAS, true) <= 160 * 1024;
  const size_t smem = Cfg::smem(a.R, BIAS, as.lds_stage != 0);
  auto kern = attn_bwd_kv64_kernel<D, BF16, BIAS, HALF, ONE, NODIAG>;
  if (smem > 48 * 1024) {  // (idempotent driver call; the library keeps no state of its own)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, as);
  return hipGetLastError();
}

template <int D, bool BF16, int BIAS, bool QDG = false>
static hipError_t launch_q64(const AttnArgs& a, int grid, hipStream_t s) {
  // (operands / outputs through wave-private LDS images whenever the workgroup's LDS allows: a radius beyond ~500 does not)
  AttnArgs as = a;
  fill_div_magic(as, grid);
  as.lds_stage = BwdQ64Cfg<D>::smem(a.R, BIAS, true, QDG) <= 160 * 1024;
  const size_t smem = BwdQ64Cfg<D>::smem(a.R, BIAS, as.lds_stage != 0, QDG);
  auto kern = attn_bwd_q64_kernel<D, BF16, BIAS, QDG>;
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, as);
  return hipGetLastError();
}
hipError_t CAT(launch_bwd_q64_d, FAT5_INST_D)(const AttnArgs& a, int bf16, int bias, int /*nw*/, int grid, hipStream_t s) {
  if (bias == FAT5_BIAS_RPE1D) {
    // (a.diag_q: the layout of this call has the dQ side form the table gradient's diagonal sums -- the stage-by-stage form of the one-launch backward)
    if (a.diag_q) return bf16 ? launch_q64<FAT5_INST_D, true, FAT5_BIAS_RPE1D, true>(a, grid, s) : launch_q64<FAT5_INST_D, false, FAT5_BIAS_RPE1D, true>(a, grid, s);
    return bf16 ? launch_q64<FAT5_INST_D, true, FAT5_BIAS_RPE1D>(a, grid, s) : launch_q64<FAT5_INST_D, false, FAT5_BIAS_RPE1D>(a, grid, s);
  }
  return bf16 ? launch_q64<FAT5_INST_D, true, FAT5_BIAS_NONE>(a, grid, s) : launch_q64<FAT5_INST_D, false, FAT5_BIAS_NONE>(a, grid, s);
}

template <bool HALF>
static hipError_t launch_kv64_bias(const AttnArgs& a, int bf16, int bias, int grid, hipStream_t s) {
  if constexpr (!HALF) {  // (dense bias, round 5: 256-key workgroups; fp16 since the second half of the round)
    if (bias == FAT5_BIAS_DENSE) {
      const bool one = inv_scale_is_16bit(a.scale, bf16 != 0);
      if (bf16) return one ? launch_kv64<FAT5_INST_D, true, FAT5_BIAS_DENSE, false, true>(a, grid, s) : launch_kv64<FAT5_INST_D, true, FAT5_BIAS_DENSE, false, false>(a, grid, s);
      return one ? launch_kv64<FAT5_INST_D, false, FAT5_BIAS_DENSE, false, true>(a, grid, s) : launch_kv64<FAT5_INST_D, false, FAT5_BIAS_DENSE, false, false>(a, grid, s);
    }
  }
  if (bias == FAT5_BIAS_RPE1D) {
    if constexpr (!HALF) {
      if (a.diag_q)  // (... and the dK/dV side none)
        return bf16 ? launch_kv64<FAT5_INST_D, true, FAT5_BIAS_RPE1D, false, false, true>(a, grid, s) : launch_kv64<FAT5_INST_D, false, FAT5_BIAS_RPE1D, false, false, true>(a, grid, s);
    }
    return bf16 ? launch_kv64<FAT5_INST_D, true, FAT5_BIAS_RPE1D, HALF>(a, grid, s) : launch_kv64<FAT5_INST_D, false, FAT5_BIAS_RPE1D, HALF>(a, grid, s);
  }
  return bf16 ? launch_kv64<FAT5_INST_D, true, FAT5_BIAS_NONE, HALF>(a, grid, s) : launch_kv64<FAT5_INST_D, false, FAT5_BIAS_NONE, HALF>(a, grid, s);
}
template <int D, bool BF16, int BIAS>
static hipError_t launch_kv64_mixed(const AttnArgs& a, int grid, hipStream_t s) {
  AttnArgs as = a;
  fill_div_magic(as, grid);
  as.lds_stage = std::max(Bwd64Cfg<D, false>::smem(a.R, BIAS, true), Bwd64Cfg<D, true>::smem(a.R, BIAS, true)) <= 160 * 1024;
  const bool st = as.lds_stage != 0;
  const size_t smem = std::max(Bwd64Cfg<D, false>::smem(a.R, BIAS, st), Bwd64Cfg<D, true>::smem(a.R, BIAS, st));
  auto kern = attn_bwd_kv64_mixed_kernel<D, BF16, BIAS>;
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, as);
  return hipGetLastError();
}
// nw == 2: the half-length variant (128-key workgroups, two wave pairs each walking half of the query steps); nw == 3: both in one
// launch (a.mix_full pairs per XCD as 256-key workgroups, the others half-length); otherwise 256 keys
hipError_t CAT(launch_bwd_kv64_d, FAT5_INST_D)(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s) {
  if (nw == 3) {
    if (bias == FAT5_BIAS_RPE1D)
      return bf16 ? launch_kv64_mixed<FAT5_INST_D, true, FAT5_BIAS_RPE1D>(a, grid, s) : launch_kv64_mixed<FAT5_INST_D, false, FAT5_BIAS_RPE1D>(a, grid, s);
    return bf16 ? launch_kv64_mixed<FAT5_INST_D, true, FAT5_BIAS_NONE>(a, grid, s) : launch_kv64_mixed<FAT5_INST_D, false, FAT5_BIAS_NONE>(a, grid, s);
  }
  return nw == 2 ? launch_kv64_bias<true>(a, bf16, bias, grid, s) : launch_kv64_bias<false>(a, bf16, bias, grid, s);
}
// dK/dV (self-sufficient 256-key form) and dQ in one launch: a.n_kv_blocks workgroups of the former, then the dQ workgroups
template <int D, bool BF16, int BIAS, bool QDG = false>
static hipError_t launch_fused64(const AttnArgs& a, int grid, hipStream_t s) {
  AttnArgs as = a;
  fill_div_magic(as, grid);
  as.lds_stage = std::max(Bwd64Cfg<D, false, true>::smem(a.R, BIAS, true), BwdQ64Cfg<D>::smem(a.R, BIAS, true, QDG)) <= 160 * 1024;
  const bool st = as.lds_stage != 0;
  const size_t smem = std::max(Bwd64Cfg<D, false, true>::smem(a.R, BIAS, st), BwdQ64Cfg<D>::smem(a.R, BIAS, st, QDG));
  auto kern = attn_bwd_fused64_kernel<D, BF16, BIAS, QDG>;
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, as);
  return hipGetLastError();
}
hipError_t CAT(launch_bwd_fused64_d, FAT5_INST_D)(const AttnArgs& a, int bf16, int bias, int /*nw*/, int grid, hipStream_t s) {
  if (bias == FAT5_BIAS_RPE1D) {
    if (a.diag_q)  // (the dQ half forms the table gradient's diagonal sums: a.part_stride counts ITS row blocks)
      return bf16 ? launch_fused64<FAT5_INST_D, true, FAT5_BIAS_RPE1D, true>(a, grid, s) : launch_fused64<FAT5_INST_D, false, FAT5_BIAS_RPE1D, true>(a, grid, s);
    return bf16 ? launch_fused64<FAT5_INST_D, true, FAT5_BIAS_RPE1D>(a, grid, s) : launch_fused64<FAT5_INST_D, false, FAT5_BIAS_RPE1D>(a, grid, s);
  }
  return bf16 ? launch_fused64<FAT5_INST_D, true, FAT5_BIAS_NONE>(a, grid, s) : launch_fused64<FAT5_INST_D, false, FAT5_BIAS_NONE>(a, grid, s);
}
size_t CAT(smem_bwd_fused64_d, FAT5_INST_D)(int R, int bias) {
  return std::max(Bwd64Cfg<FAT5_INST_D, false, true>::smem(R, bias), BwdQ64Cfg<FAT5_INST_D>::smem(R, bias, false, true));
}
size_t CAT(smem_bwd_kv64_d, FAT5_INST_D)(int R, int bias) {
  return bias == FAT5_BIAS_DENSE ? Bwd64Cfg<FAT5_INST_D, false, false, true>::smem(R, bias) : Bwd64Cfg<FAT5_INST_D>::smem(R, bias);
}
size_t CAT(smem_bwd_kv64h_d, FAT5_INST_D)(int R, int bias) { return Bwd64Cfg<FAT5_INST_D, true>::smem(R, bias); }

}  // namespace fat5
