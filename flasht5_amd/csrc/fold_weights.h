// Stacked projection weights of the T5 blocks (host side: flasht5_amd/fused_linear.py).  Up to three nn.Linear weights applied to the same
// normalised input -- Wq | Wk | Wv (reference src/model/modeling_flash_t5.py:304-318), wi_0 | wi_1 (:159-160) -- run as ONE library GEMM on their stack:
//  * fold_weights_kernel: the stack, optionally times diag(g) (the RMSNorm weight folded into the projection: the backward's dL/dxhat = dout (W diag g));
//  * fold_weights_bwd_kernel / fold_weights_dg_kernel: gradient of the folded stack -> per-weight gradients and the norm weight's gradient.
// (Rounds 3-5 also held a hand-written MFMA GEMM here -- the norm statistics in its prologue, the residual add in its epilogue: `fat5_linear_fused`.  It ran at
//  453-669 TF/s where hipBLASLt gives 690-1020 on the FAT5-base shapes and was removed in round 6; the GEMMs are library GEMMs now.)
#pragma once
#include "attn_common.h"

namespace fat5 {

// wfold[n][k] = w_i[n - first_i][k] * g[k]: up to three (rows_i, K) weights stacked along n with the RMSNorm weight folded in
// (g == nullptr: the plain stack) -- ONE launch instead of cat + float + mul + cast; 8 elements per thread
template <bool BF16>
__global__ __launch_bounds__(256) void fold_weights_kernel(const uint16_t* __restrict__ w0, const uint16_t* __restrict__ w1,
                                                           const uint16_t* __restrict__ w2, int n0, int n1, int n2, int64_t ld0,
                                                           int64_t ld1, int64_t ld2, const uint16_t* __restrict__ g,
                                                           uint16_t* __restrict__ out, int K) {
  const int kc = K / 8;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (int64_t)(n0 + n1 + n2) * kc) return;
  const int n = (int)(id / kc), c = (int)(id - (int64_t)n * kc);
  const uint16_t* src = n < n0 ? w0 + (int64_t)n * ld0 : (n < n0 + n1 ? w1 + (int64_t)(n - n0) * ld1 : w2 + (int64_t)(n - n0 - n1) * ld2);
  u32x4 v = *reinterpret_cast<const u32x4*>(src + 8 * c);
  if (g) {
    const u32x4 gv = *reinterpret_cast<const u32x4*>(g + 8 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = pack2<BF16>(cvt_lo<BF16>(v[j]) * cvt_lo<BF16>(gv[j]), cvt_hi<BF16>(v[j]) * cvt_hi<BF16>(gv[j]));
  }
  *reinterpret_cast<u32x4*>(out + (int64_t)n * K + 8 * c) = v;
}

// Backward of the folded weight: dwg = d(W_stack diag g) (N, K) as it comes out of the dout^T xhat GEMM ->
//   dW_i[n][k] = dwg[n][k] * g[k]   (written per source weight),   dg[k] = sum_n dwg[n][k] * W_stack[n][k]  (fp32 sum, fixed order).
// One workgroup = 64 columns x one slab of `rows_per` rows (8 column chunks x 32 row phases; the 32 partial sums of a column meet in
// LDS) and writes the slab's column sums to part[slab][k]; fold_weights_dg_kernel adds the slabs in order.  (Round 3, first version:
// one workgroup per 64 columns walking ALL rows -- 12 workgroups for K = 768: 58 us per call, 3.5 ms of the 21 ms config-5 step.)
template <bool BF16>
__global__ __launch_bounds__(256) void fold_weights_bwd_kernel(const uint16_t* __restrict__ dwg, const uint16_t* __restrict__ w0,
                                                               const uint16_t* __restrict__ w1, const uint16_t* __restrict__ w2, int n0,
                                                               int n1, int n2, int64_t ld0, int64_t ld1, int64_t ld2,
                                                               const uint16_t* __restrict__ g, uint16_t* __restrict__ dw0,
                                                               uint16_t* __restrict__ dw1, uint16_t* __restrict__ dw2,
                                                               float* __restrict__ part, int K, int rows_per) {
  __shared__ float red[32][65];
  const int c8 = threadIdx.x & 7, ph = threadIdx.x >> 3;
  const int col = blockIdx.x * 64 + 8 * c8;
  const u32x4 gv = *reinterpret_cast<const u32x4*>(g + col);
  float gf[8], acc[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { gf[2 * j] = cvt_lo<BF16>(gv[j]); gf[2 * j + 1] = cvt_hi<BF16>(gv[j]); }
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const int ntot = n0 + n1 + n2;
  const int r0 = blockIdx.y * rows_per, r1 = min(ntot, r0 + rows_per);
  for (int n = r0 + ph; n < r1; n += 32) {
    const uint16_t* src;
    uint16_t* dst;
    if (n < n0) { src = w0 + (int64_t)n * ld0; dst = dw0 ? dw0 + (int64_t)n * K : nullptr; }
    else if (n < n0 + n1) { src = w1 + (int64_t)(n - n0) * ld1; dst = dw1 ? dw1 + (int64_t)(n - n0) * K : nullptr; }
    else { src = w2 + (int64_t)(n - n0 - n1) * ld2; dst = dw2 ? dw2 + (int64_t)(n - n0 - n1) * K : nullptr; }
    const u32x4 dv = *reinterpret_cast<const u32x4*>(dwg + (int64_t)n * K + col);
    const u32x4 wv = *reinterpret_cast<const u32x4*>(src + col);
    u32x4 ov;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d0 = cvt_lo<BF16>(dv[j]), d1 = cvt_hi<BF16>(dv[j]);
      acc[2 * j] = fmaf(d0, cvt_lo<BF16>(wv[j]), acc[2 * j]);
      acc[2 * j + 1] = fmaf(d1, cvt_hi<BF16>(wv[j]), acc[2 * j + 1]);
      ov[j] = pack2<BF16>(d0 * gf[2 * j], d1 * gf[2 * j + 1]);
    }
    if (dst) *reinterpret_cast<u32x4*>(dst + col) = ov;
  }
  if (!part) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ph][8 * c8 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += red[r][threadIdx.x];
    part[(int64_t)blockIdx.y * K + blockIdx.x * 64 + threadIdx.x] = s;
  }
}

// dg[k] = sum over the slabs (in order) of part[slab][k]
template <bool BF16>
__global__ __launch_bounds__(256) void fold_weights_dg_kernel(const float* __restrict__ part, uint16_t* __restrict__ dg, int K, int nslab) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  float s = 0.f;
  for (int s0 = 0; s0 < nslab; s0 += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = (s0 + u < nslab) ? part[(int64_t)(s0 + u) * K + k] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  dg[k] = to16<BF16>(s);
}

}  // namespace fat5
