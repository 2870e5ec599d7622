// Instantiations of the forward kernel for one head_dim (compile with -DFAT5_INST_D=32|64|128).
#include "attn_fwd.h"
#include <cstdlib>
#include "attn_launch.h"

#ifndef FAT5_INST_D
#error "FAT5_INST_D must be defined"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

namespace fat5 {

template <int D, bool BF16, int BIAS, int NW, bool BDMA = false>
static hipError_t launch_one(const AttnArgs& a, int grid, hipStream_t s) {
  size_t smem = FwdCfg<D, NW>::smem(a.R, BIAS);
#ifdef FAT5_FWD_PAD_LDS  // developer experiment: the no-bias forward at the dense instantiations' LDS footprint (occupancy)
  if (BIAS == FAT5_BIAS_NONE) smem += FAT5_FWD_PAD_LDS;
#endif
  auto kern = attn_fwd_kernel<D, BF16, BIAS, NW, BDMA>;
  static size_t configured = 0;  // per instantiation; benign race (idempotent call)
  if (smem > 48 * 1024 && smem > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    configured = smem;
  }
  AttnArgs am = a;
  fill_div_magic(am, grid);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), smem, s, am);
  return hipGetLastError();
}

template <int D, bool BF16, int BIAS, bool BDMA = false>
static hipError_t launch_split(const AttnArgs& a, int grid, hipStream_t s) {
  size_t smem = FwdCfg<D, 4, true>::smem(a.R, BIAS);
  auto kern = attn_fwd_split_kernel<D, BF16, BIAS, 4, BDMA>;
  static size_t configured = 0;
  if (smem > 48 * 1024 && smem > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    configured = smem;
  }
  AttnArgs am = a;
  fill_div_magic(am, grid);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, am);
  return hipGetLastError();
}

template <int D, bool BF16, int BIAS, bool BDMA = false>
static hipError_t launch_nw(const AttnArgs& a, int nw, int grid, hipStream_t s) {
  if (nw == -4) return launch_split<D, BF16, BIAS, BDMA>(a, grid, s);  // two waves per 32 query rows (short sequences)
  if (nw == 2) return launch_one<D, BF16, BIAS, 2, BDMA>(a, grid, s);
  if (nw == 8) return launch_one<D, BF16, BIAS, 8, BDMA>(a, grid, s);
  return launch_one<D, BF16, BIAS, 4, BDMA>(a, grid, s);
}
template <int D, bool BF16>
static hipError_t launch_bias(const AttnArgs& a, int bias, int nw, int grid, hipStream_t s) {
  switch (bias) {
    case FAT5_BIAS_NONE: return launch_nw<D, BF16, FAT5_BIAS_NONE>(a, nw, grid, s);
    case FAT5_BIAS_DENSE:  // (bias tiles by LDS-DMA -- aligned, unit-stride rows, no packed batch -- as a compile-time fact: attn_fwd.h)
      return (a.bias_dma && a.cu_q == nullptr) ? launch_nw<D, BF16, FAT5_BIAS_DENSE, true>(a, nw, grid, s)
                                                : launch_nw<D, BF16, FAT5_BIAS_DENSE>(a, nw, grid, s);
    default: return launch_nw<D, BF16, FAT5_BIAS_RPE1D>(a, nw, grid, s);
  }
}

hipError_t CAT(launch_fwd_d, FAT5_INST_D)(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s) {
  return bf16 ? launch_bias<FAT5_INST_D, true>(a, bias, nw, grid, s) : launch_bias<FAT5_INST_D, false>(a, bias, nw, grid, s);
}
size_t CAT(smem_fwd_d, FAT5_INST_D)(int nw, int R, int bias) {
  return FwdCfg<FAT5_INST_D, 4>::smem(R, bias);  // independent of NW
}

}  // namespace fat5
