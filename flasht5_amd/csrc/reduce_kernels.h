// Small bandwidth-bound helpers of the attention backward.
#pragma once
#include "attn_common.h"

namespace fat5 {

// dbias[bb][hb][m][n] = sum over the broadcast batch / head dims of ds[b][h][m][n]
// (replaces the reference's `ds.sum(0, keepdim=True)`, flash_attention_v2_bias.py:214-215, and adds
//  the head reduction the reference gets wrong for (1,1,M,N) biases -- SURVEY Q4).
// ds is (B,H,MN) contiguous in the bias dtype (already rounded like the reference, :720); the sum
// runs in fp32 in a fixed order => deterministic.
template <bool BF16>
__global__ __launch_bounds__(256) void dbias_reduce_kernel(const uint16_t* __restrict__ ds, uint16_t* __restrict__ out,
                                                           int B, int H, int Bb, int Hb, int64_t MN) {
  // one thread = 8 consecutive elements of one (bb, hb) output slice
  const int64_t chunks = (MN + 7) / 8;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = chunks * Bb * Hb;
  if (gid >= total) return;
  const int64_t c = gid % chunks;
  const int slice = (int)(gid / chunks);
  const int bb = slice / Hb, hb = slice % Hb;
  const int b_lo = (Bb == 1) ? 0 : bb, b_hi = (Bb == 1) ? B : bb + 1;
  const int h_lo = (Hb == 1) ? 0 : hb, h_hi = (Hb == 1) ? H : hb + 1;
  const int64_t e0 = c * 8;
  const bool full = (e0 + 8 <= MN) && ((MN & 7) == 0);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int b = b_lo; b < b_hi; ++b)
    for (int h = h_lo; h < h_hi; ++h) {
      const uint16_t* src = ds + ((int64_t)b * H + h) * MN + e0;
      if (full) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[2 * j] += cvt_lo<BF16>(v[j]);
          acc[2 * j + 1] += cvt_hi<BF16>(v[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (e0 + j < MN) acc[j] += cvt16<BF16>(src[j]);
      }
    }
  uint16_t* dst = out + ((int64_t)bb * Hb + hb) * MN + e0;
  if (full) {
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = pack2<BF16>(acc[2 * j], acc[2 * j + 1]);
    *reinterpret_cast<u32x4*>(dst) = v;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (e0 + j < MN) dst[j] = to16<BF16>(acc[j]);
  }
}

// drpe1d[h][i] = sum_b sum_blk part[(b*H + h)*nblk + blk][i], fixed order.
__global__ __launch_bounds__(256) void drpe_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                          int B, int H, int nblk, int n1) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= H * n1) return;
  const int h = gid / n1, i = gid % n1;
  float acc = 0.f;
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < nblk; ++k) acc += part[(((int64_t)b * H + h) * nblk + k) * n1 + i];
  out[gid] = acc;
}

}  // namespace fat5
