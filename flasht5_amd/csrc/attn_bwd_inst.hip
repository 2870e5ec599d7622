// Instantiations of the backward kernels for one head_dim (compile with -DFAT5_INST_D=32|64|128).
#include "attn_bwd.h"
#include "attn_bwd_dbias.h"
#include "attn_launch.h"
#include <algorithm>

#ifndef FAT5_INST_D
#error "FAT5_INST_D must be defined"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

namespace fat5 {

template <typename K>
static hipError_t set_smem(K kern, size_t smem, size_t& configured) {
  if (smem > 48 * 1024 && smem > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    configured = smem;  // per instantiation; benign race (idempotent call)
  }
  return hipSuccess;
}

template <int D, bool BF16, int BIAS, int NW>
static hipError_t launch_q(const AttnArgs& a, int grid, hipStream_t s) {
  const size_t smem = BwdQCfg<D, NW>::smem(a.R, BIAS);
  auto kern = attn_bwd_q_kernel<D, BF16, BIAS, NW>;
  static size_t configured = 0;
  hipError_t e = set_smem(kern, smem, configured);
  if (e != hipSuccess) return e;
  AttnArgs am = a;
  fill_div_magic(am, grid);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), smem, s, am);
  return hipGetLastError();
}
template <int D, bool BF16, int BIAS, int NW>
static hipError_t launch_kv(const AttnArgs& a, int grid, hipStream_t s) {
  const size_t smem = BwdKVCfg<D, NW>::smem(a.R, BIAS);
  auto kern = attn_bwd_kv_kernel<D, BF16, BIAS, NW>;
  static size_t configured = 0;
  hipError_t e = set_smem(kern, smem, configured);
  if (e != hipSuccess) return e;
  AttnArgs am = a;
  fill_div_magic(am, grid);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), smem, s, am);
  return hipGetLastError();
}

template <int D, bool BF16, int BIAS, int NW>
static hipError_t launch_fused(const AttnArgs& a, int grid, hipStream_t s) {
  const size_t smem = std::max(BwdKVCfg<D, NW>::smem(a.R, BIAS), BwdQCfg<D, NW>::smem(a.R, BIAS));
  auto kern = attn_bwd_fused_kernel<D, BF16, BIAS, NW>;
  static size_t configured = 0;
  hipError_t e = set_smem(kern, smem, configured);
  if (e != hipSuccess) return e;
  AttnArgs am = a;
  fill_div_magic(am, grid);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), smem, s, am);
  return hipGetLastError();
}

// dense bias gradient by in-kernel batch reduction (attn_bwd_dbias.h): grid = H x ceil(M / 128) x key splits; a.lds_stage != 0 (D <= 64) selects the form
// with two wave groups sharing the batch (512 threads, two waves per SIMD)
template <typename K>
static hipError_t launch_dbias_one(K kern, size_t smem, size_t& configured, int threads, const AttnArgs& a, void* dbias, float* scratch, int grid, hipStream_t s) {
  hipError_t e = set_smem(kern, smem, configured);
  if (e != hipSuccess) return e;
  AttnArgs am = a;
  fill_div_magic(am, grid);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), smem, s, am, (uint16_t*)dbias, scratch);
  return hipGetLastError();
}
hipError_t CAT(launch_bwd_dbias_d, FAT5_INST_D)(const AttnArgs& a, int bf16, void* dbias, float* scratch, int grid, hipStream_t s) {
  static size_t cfg[4] = {0, 0, 0, 0};
#if FAT5_INST_D <= 64
  if (a.lds_stage) {
    const size_t smem2 = BwdDbiasCfg<FAT5_INST_D, 4, true>::smem();
    if (bf16) return launch_dbias_one(attn_bwd_dbias_split_kernel<FAT5_INST_D, true, 4>, smem2, cfg[2], 512, a, dbias, scratch, grid, s);
    return launch_dbias_one(attn_bwd_dbias_split_kernel<FAT5_INST_D, false, 4>, smem2, cfg[3], 512, a, dbias, scratch, grid, s);
  }
#endif
  const size_t smem = BwdDbiasCfg<FAT5_INST_D, 4>::smem();
  if (bf16) return launch_dbias_one(attn_bwd_dbias_kernel<FAT5_INST_D, true, 4>, smem, cfg[0], 256, a, dbias, scratch, grid, s);
  return launch_dbias_one(attn_bwd_dbias_kernel<FAT5_INST_D, false, 4>, smem, cfg[1], 256, a, dbias, scratch, grid, s);
}

#define DISPATCH(FN, a, bf16, bias, nw, grid, s)                                                     \
  do {                                                                                               \
    if (bf16) {                                                                                      \
      if (bias == FAT5_BIAS_NONE) return nw == 2 ? FN<FAT5_INST_D, true, 0, 2>(a, grid, s) : FN<FAT5_INST_D, true, 0, 4>(a, grid, s); \
      if (bias == FAT5_BIAS_DENSE) return nw == 2 ? FN<FAT5_INST_D, true, 1, 2>(a, grid, s) : FN<FAT5_INST_D, true, 1, 4>(a, grid, s); \
      return nw == 2 ? FN<FAT5_INST_D, true, 2, 2>(a, grid, s) : FN<FAT5_INST_D, true, 2, 4>(a, grid, s); \
    } else {                                                                                         \
      if (bias == FAT5_BIAS_NONE) return nw == 2 ? FN<FAT5_INST_D, false, 0, 2>(a, grid, s) : FN<FAT5_INST_D, false, 0, 4>(a, grid, s); \
      if (bias == FAT5_BIAS_DENSE) return nw == 2 ? FN<FAT5_INST_D, false, 1, 2>(a, grid, s) : FN<FAT5_INST_D, false, 1, 4>(a, grid, s); \
      return nw == 2 ? FN<FAT5_INST_D, false, 2, 2>(a, grid, s) : FN<FAT5_INST_D, false, 2, 4>(a, grid, s); \
    }                                                                                                \
  } while (0)

hipError_t CAT(launch_bwd_q_d, FAT5_INST_D)(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s) {
  DISPATCH(launch_q, a, bf16, bias, nw, grid, s);
}
hipError_t CAT(launch_bwd_kv_d, FAT5_INST_D)(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s) {
  DISPATCH(launch_kv, a, bf16, bias, nw, grid, s);
}
// fused launch: 4-wave tiles only (the small-problem configuration)
hipError_t CAT(launch_bwd_fused_d, FAT5_INST_D)(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s) {
  (void)nw;
  if (bf16) {
    if (bias == FAT5_BIAS_NONE) return launch_fused<FAT5_INST_D, true, 0, 4>(a, grid, s);
    if (bias == FAT5_BIAS_DENSE) return launch_fused<FAT5_INST_D, true, 1, 4>(a, grid, s);
    return launch_fused<FAT5_INST_D, true, 2, 4>(a, grid, s);
  }
  if (bias == FAT5_BIAS_NONE) return launch_fused<FAT5_INST_D, false, 0, 4>(a, grid, s);
  if (bias == FAT5_BIAS_DENSE) return launch_fused<FAT5_INST_D, false, 1, 4>(a, grid, s);
  return launch_fused<FAT5_INST_D, false, 2, 4>(a, grid, s);
}
size_t CAT(smem_bwd_q_d, FAT5_INST_D)(int nw, int R, int bias) {
  return nw == 2 ? BwdQCfg<FAT5_INST_D, 2>::smem(R, bias) : BwdQCfg<FAT5_INST_D, 4>::smem(R, bias);
}
size_t CAT(smem_bwd_kv_d, FAT5_INST_D)(int nw, int R, int bias) {
  return nw == 2 ? BwdKVCfg<FAT5_INST_D, 2>::smem(R, bias) : BwdKVCfg<FAT5_INST_D, 4>::smem(R, bias);
}

}  // namespace fat5
