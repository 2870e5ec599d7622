// Instantiations of the dense-bias dQ + batch-reduced dBias body (attn_bwd_qdb64.h), D = 64, bf16 / fp16.
#include "attn_bwd_qdb64.h"
#include "attn_launch.h"
#include <cstring>

namespace fat5 {

template <bool BF16>
static hipError_t launch_qdb64_t(const AttnArgs& a, void* dbias_out, int partial, int grid, hipStream_t s) {
  AttnArgs as = a;
  as.n_mblk = (a.M + 63) / 64;
  as.mg_mblk = div_magic(as.n_mblk, grid);
  grid = (grid + 7) / 8 * 8;  // (eight contiguous chunks of work items, one per XCD: see the kernel)
  constexpr int smem = BwdQdb64Cfg<64>::SMEM;
  // (1 / scale a 16-bit value itself -- 1, 8, ...: the one-term selector, four bias MFMAs per step instead of eight)
  const bool one = is_one16<BF16>(1.f / a.scale);
  auto go = [&](auto kern) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, as, dbias_out);
    return hipGetLastError();
  };
  if (partial) return one ? go(attn_bwd_qdb64_kernel<64, BF16, true, true>) : go(attn_bwd_qdb64_kernel<64, BF16, true, false>);
  return one ? go(attn_bwd_qdb64_kernel<64, BF16, false, true>) : go(attn_bwd_qdb64_kernel<64, BF16, false, false>);
}
// grid = H * ceil(B / 4) * ceil(M / 64) workgroups; `dbias_out`: the (H, M, N) 16-bit dbias (B <= 4) or the (ceil(B / 4), H, M, N) fp32 slabs
hipError_t launch_bwd_qdb64_d64(const AttnArgs& a, int bf16, void* dbias_out, int partial, int grid, hipStream_t s) {
  return bf16 ? launch_qdb64_t<true>(a, dbias_out, partial, grid, s) : launch_qdb64_t<false>(a, dbias_out, partial, grid, s);
}

// The dense backward in one launch (attn_bwd_dfused64_kernel): the row statistics by bwd_stat2_kernel, then a.n_kv_blocks 256-key dK/dV workgroups (padded to a
// multiple of eight) followed by the `grid_qdb` dQ + dBias workgroups.  a.n_nblk / a.part_stride as for launch_bwd_kv64_d64, a.stat2 set.
template <bool BF16>
static hipError_t launch_dfused64_t(const AttnArgs& a, void* dbias_out, int partial, int grid_qdb, hipStream_t s) {
  if (!a.stat2) return hipErrorInvalidValue;
  const long nst = (a.M + 31) / 32, nsw = (long)a.B * a.H * nst;
  hipLaunchKernelGGL((bwd_stat2_kernel<64, BF16>), dim3((unsigned)((nsw + 3) / 4)), dim3(256), 0, s, a);
  hipError_t e0 = hipGetLastError();
  if (e0 != hipSuccess) return e0;
  AttnArgs as = a;
  as.n_mblk = (a.M + 63) / 64;
  const int nkv8 = (a.n_kv_blocks + 7) & ~7;
  const int grid = nkv8 + (grid_qdb + 7) / 8 * 8;
  as.mg_mblk = div_magic(as.n_mblk, grid);
  as.mg_nblk = div_magic(as.n_nblk, grid);
  as.mg_H = div_magic(as.H, grid);
  as.lds_stage = 0;  // (the dense dK/dV body's ring leaves no room for staged K / V images: Bwd64Cfg)
  constexpr int smem = BwdQdb64Cfg<64>::SMEM;
  static_assert(Bwd64Cfg<64, false, false, true>::RINGB <= smem, "the dK/dV half's ring fits the dQ half's LDS");
  const bool one = is_one16<BF16>(1.f / a.scale);
  auto go = [&](auto kern) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, as, dbias_out);
    return hipGetLastError();
  };
  if (partial) return one ? go(attn_bwd_dfused64_kernel<64, BF16, true, true>) : go(attn_bwd_dfused64_kernel<64, BF16, true, false>);
  return one ? go(attn_bwd_dfused64_kernel<64, BF16, false, true>) : go(attn_bwd_dfused64_kernel<64, BF16, false, false>);
}
hipError_t launch_bwd_dfused64_d64(const AttnArgs& a, int bf16, void* dbias_out, int partial, int grid_qdb, hipStream_t s) {
  return bf16 ? launch_dfused64_t<true>(a, dbias_out, partial, grid_qdb, s) : launch_dfused64_t<false>(a, dbias_out, partial, grid_qdb, s);
}

hipError_t launch_dbias_partial_reduce(const float* part, void* out, int bf16, int ngrp, int H, int M, int N, int causal, hipStream_t s) {
  const int64_t HMN = (int64_t)H * M * N;
  const int grid = (int)((HMN / 8 + 255) / 256);
  if (bf16) hipLaunchKernelGGL(dbias_partial_reduce_kernel<true>, dim3(grid), dim3(256), 0, s, part, (uint16_t*)out, ngrp, HMN, M, N, causal, N - M);
  else hipLaunchKernelGGL(dbias_partial_reduce_kernel<false>, dim3(grid), dim3(256), 0, s, part, (uint16_t*)out, ngrp, HMN, M, N, causal, N - M);
  return hipGetLastError();
}

}  // namespace fat5
