// Host-visible launchers of the templated attention kernels (one translation unit per head_dim).
#pragma once
#include <hip/hip_runtime.h>
#include "attn_common.h"

namespace fat5 {
// exact quotients by multiplication for the workgroup-index decode (attn_common.h: fast_div) wherever grid x divisor < 2^32
inline void fill_div_magic(AttnArgs& a, long grid) {
  a.mg_mblk = div_magic(a.n_mblk, grid);
  a.mg_nblk = div_magic(a.n_nblk, grid);
  a.mg_H = div_magic(a.H, grid);
}
// each returns hipError_t of the launch; nw in {2,4}
#define FAT5_DECL_LAUNCH(D)                                                                           \
  hipError_t launch_fwd_d##D(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s); \
  hipError_t launch_bwd_q_d##D(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s); \
  hipError_t launch_bwd_kv_d##D(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s); \
  hipError_t launch_bwd_fused_d##D(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s); \
  hipError_t launch_bwd_dbias_d##D(const AttnArgs& a, int bf16, void* dbias, float* scratch, int grid, hipStream_t s); \
  size_t smem_fwd_d##D(int nw, int R, int bias);                                                      \
  size_t smem_bwd_q_d##D(int nw, int R, int bias);                                                    \
  size_t smem_bwd_kv_d##D(int nw, int R, int bias);
FAT5_DECL_LAUNCH(32)
FAT5_DECL_LAUNCH(64)
FAT5_DECL_LAUNCH(128)
#undef FAT5_DECL_LAUNCH
// 64 query rows per wave, software-pipelined (attn_fwd64.h): bias none / rpe1d, no packed batches
hipError_t launch_fwd64_d64(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s);
size_t smem_fwd64_d64(int R, int bias);
hipError_t launch_fwd64_d128(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s);  // (bias none / rpe1d)
size_t smem_fwd64_d128(int R, int bias);  // dynamic LDS of one workgroup (two fit a CU up to 80 KB each)
// 64 keys per wave, software-pipelined dK/dV body (attn_bwd64.h): bf16, bias none / rpe1d, no packed batches
hipError_t launch_bwd_kv64_d64(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s);
size_t smem_bwd_kv64_d64(int R, int bias);
size_t smem_bwd_kv64h_d64(int R, int bias);  // (nw == 2: 128-key workgroups, half the query steps per wave pair)
// 64 query rows per wave, software-pipelined dQ body (attn_bwd64.h): same conditions
hipError_t launch_bwd_q64_d64(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s);
// dK/dV (256-key workgroups, row statistics formed in the body) + dQ in one launch: a.n_kv_blocks workgroups of the former first
hipError_t launch_bwd_fused64_d64(const AttnArgs& a, int bf16, int bias, int nw, int grid, hipStream_t s);
size_t smem_bwd_fused64_d64(int R, int bias);
// dense (1, H, M, N) bias: dQ + the batch-reduced dbias, four batch elements per workgroup (attn_bwd_qdb64.h): bf16, D = 64
hipError_t launch_bwd_qdb64_d64(const AttnArgs& a, int bf16, void* dbias_out, int partial, int grid, hipStream_t s);
// ... and the whole dense backward in one launch: row statistics ahead (bwd_stat2_kernel), then a.n_kv_blocks 256-key dense dK/dV workgroups + the dQ + dBias ones
hipError_t launch_bwd_dfused64_d64(const AttnArgs& a, int bf16, void* dbias_out, int partial, int grid_qdb, hipStream_t s);
hipError_t launch_dbias_partial_reduce(const float* part, void* out, int bf16, int ngrp, int H, int M, int N, int causal, hipStream_t s);
}  // namespace fat5
