// Small bandwidth-bound helpers of the attention backward.
#pragma once
#include "attn_common.h"

namespace fat5 {

// dbias[bb][hb][m][n] = sum over the broadcast batch / head dims of ds[b][h][m][n]
// (replaces the reference's `ds.sum(0, keepdim=True)`, flash_attention_v2_bias.py:214-215, and adds
//  the head reduction the reference gets wrong for (1,1,M,N) biases -- SURVEY Q4).
// ds is (B,H,MN) contiguous in the bias dtype (already rounded like the reference, :720); the sum
// runs in fp32 in a fixed order => deterministic.
// causal_n > 0: causal mask with N = causal_n keys per row and offset P = causal_p (key n is visible to row m iff n <= m + P).
// The dQ body never visits the tiles above the diagonal, so the staging tensor holds garbage there: masked elements are
// zeros by definition and are not read (a chunk whose first key is masked lies entirely above the diagonal; a chunk with a
// visible first key sits in a visited tile, whose masked elements the kernel wrote as zeros).
template <bool BF16>
__global__ __launch_bounds__(256) void dbias_reduce_kernel(const uint16_t* __restrict__ ds, uint16_t* __restrict__ out,
                                                           int B, int H, int Bb, int Hb, int64_t MN, int causal_n, int causal_p) {
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
  const bool full = (e0 + 8 <= MN) && ((MN & 7) == 0) && (causal_n <= 0 || (causal_n & 7) == 0);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  bool vis[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) vis[j] = true;
  if (causal_n > 0) {
    if (full) {  // (the chunk lies inside one row: ONE division instead of eight 64-bit ones)
      const int64_t m = e0 / causal_n;
      const int64_t n0 = e0 - m * causal_n;
#pragma unroll
      for (int j = 0; j < 8; ++j) vis[j] = n0 + j <= m + causal_p;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t idx = e0 + j, m = idx / causal_n;
        vis[j] = (idx - m * causal_n) <= m + causal_p;
      }
    }
  }
  const bool any_vis = vis[0] || !full;  // (full chunks lie inside one row: keys ascend, so the first one decides)
  if (full && any_vis) {
    // slices in groups of eight, every load of a group in flight before the first add (the sum keeps the order b-major, h-minor)
    const int nb = b_hi - b_lo, nh = h_hi - h_lo, ns = nb * nh;
    for (int s0 = 0; s0 < ns; s0 += 8) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int si = min(s0 + u, ns - 1);
        const int b = b_lo + si / nh, h = h_lo + si % nh;
        v[u] = *reinterpret_cast<const u32x4*>(ds + ((int64_t)b * H + h) * MN + e0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s0 + u < ns) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[2 * j] += cvt_lo<BF16>(v[u][j]);
            acc[2 * j + 1] += cvt_hi<BF16>(v[u][j]);
          }
        }
    }
  } else if (any_vis) {
    for (int b = b_lo; b < b_hi; ++b)
      for (int h = h_lo; h < h_hi; ++h) {
        const uint16_t* src = ds + ((int64_t)b * H + h) * MN + e0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (e0 + j < MN && vis[j]) acc[j] += cvt16<BF16>(src[j]);
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

// One 1024-thread workgroup per head: drpe1d[h][i] = sum over the (b, key-block) partials (fixed order: up to four
// interleaved partial chains per entry, then a fixed tree); optionally the T5 table gradient
// dtable[bucket][h] = sum_{i: bucket[i] == bucket} drpe1d[h][i] in the same launch (32 lanes per bucket, fixed order).
__global__ __launch_bounds__(1024) void drpe_reduce_kernel(const float* __restrict__ part, float* __restrict__ out1d,
                                                           const int32_t* __restrict__ bucket, float* __restrict__ dtable,
                                                           int B, int H, int nblk, int n1, int nbuckets, int unit_begin,
                                                           int unit_count, uint32_t mg_n1, uint32_t mg_nblk) {  // (mg_*: fast_div magics of n1 / nblk or 0)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sv4 = reinterpret_cast<float*>(smem);          // [4][n1] partial chains
  float* sv = sv4 + 4 * n1;                             // [n1] reduced diagonal sums of this head
  int* sb = reinterpret_cast<int*>(sv + n1);            // [n1] bucket ids
  const int h = blockIdx.x, tid = threadIdx.x;
  const int nparts = B * nblk;                          // partial rows of this head: (b, blk) -> b*H*nblk + h*nblk + blk
  // bucket ids: in flight together with the partial rows (loaded after the first barrier they cost a memory round trip of their own)
  int bkr[5];  // (n1 <= 2 * 2048 + 1)
#pragma unroll
  for (int u = 0; u < 5; ++u) bkr[u] = (bucket && tid + 1024 * u < n1) ? bucket[tid + 1024 * u] : 0;
  // NG interleaved partial chains per entry, as many as fit ONE pass of the 1024 threads (n1 = 257: three -- a fourth would send four
  // threads around the loop again, and every trip is a cross-XCD memory round trip the whole workgroup waits for)
  const int NG = n1 <= 256 ? 4 : (n1 <= 341 ? 3 : (n1 <= 512 ? 2 : 1));
  for (int wv = tid; wv < NG * n1; wv += 1024) {
    const int g = fast_div(wv, n1, mg_n1), i = wv - g * n1;
    float acc = 0.f;
    // 16 partial rows per round, all loads in flight before the first add (each is a cross-XCD round trip: walking
    // them one by one cost 39 us for the 256 partial rows per head of S = 8192); adds stay in the chain's fixed order
    for (int p0 = g; p0 < nparts; p0 += 16 * NG) {
      float vv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int pidx = p0 + NG * u;
        vv[u] = 0.f;
        if (pidx < nparts) {
          const int b = fast_div(pidx, nblk, mg_nblk), blk = pidx - b * nblk;
          // a unit-range call wrote the partial rows of its own units only (u = h * B + b): the others stay out of the sum
          const bool mine = unit_count <= 0 || (unsigned)(h * B + b - unit_begin) < (unsigned)unit_count;
          if (mine) vv[u] = part[(((int64_t)b * H + h) * nblk + blk) * n1 + i];
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (p0 + NG * u < nparts) acc += vv[u];
    }
    sv4[g * n1 + i] = acc;
  }
  for (int wv = NG * n1 + tid; wv < 4 * n1; wv += 1024) sv4[wv] = 0.f;  // (unused chains)
#pragma unroll
  for (int u = 0; u < 5; ++u)
    if (tid + 1024 * u < n1) sb[tid + 1024 * u] = bkr[u];
  __syncthreads();
  for (int i = tid; i < n1; i += 1024) {
    const float acc = (sv4[i] + sv4[n1 + i]) + (sv4[2 * n1 + i] + sv4[3 * n1 + i]);
    sv[i] = acc;
    if (out1d) out1d[(int64_t)h * n1 + i] = acc;
  }
  if (!dtable) return;
  __syncthreads();
  // 32 lanes per bucket, each scans every 32nd entry (n1 = 257: 9 LDS round trips instead of the 33 of an 8-lane scan), then an
  // in-group butterfly: fixed order
  const int sub = tid & 31;
  for (int bk = tid >> 5; bk < nbuckets; bk += 32) {
    float acc = 0.f;
    for (int i = sub; i < n1; i += 32) acc += (sb[i] == bk) ? sv[i] : 0.f;
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 8, 64);
    acc += __shfl_xor(acc, 16, 64);
    if (sub == 0) dtable[(int64_t)bk * H + h] = acc;
  }
}

// rpe1d[h][i] = table[bucket[i]][h] as fp32: the (H, 2R+1) Toeplitz generator of the T5 bias from the (num_buckets, H) table
// (reference: the embedding lookup + permute of RelativePositionalEncoding.compute_bias, src/utils/positional_encoding.py:100-101,
//  restricted to the 2R+1 distinct relative positions).  One launch per forward call: the generator is never cached across
// calls, so whatever updates the table in place (an optimizer writing through raw pointers) is seen by the next forward.
template <int DT>
__global__ __launch_bounds__(256) void rpe1d_gather_kernel(const void* __restrict__ table, const int32_t* __restrict__ bucket,
                                                           float* __restrict__ rpe1d, int H, int n1, int nbuckets) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * n1) return;
  const int h = i / n1, j = i - h * n1;
  const int bk = min(max(bucket[j], 0), nbuckets - 1);
  float v;
  if constexpr (DT == FAT5_F32) v = reinterpret_cast<const float*>(table)[(int64_t)bk * H + h];
  else v = cvt16<DT == FAT5_BF16>(reinterpret_cast<const uint16_t*>(table)[(int64_t)bk * H + h]);
  rpe1d[i] = v;
}

}  // namespace fat5
