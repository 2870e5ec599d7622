// Linear layers with the T5 pre-norm / residual fused in, for gfx950 ("RMSNorm -> QKV projection prologue,
// output-projection + residual epilogue", reference src/model/modeling_flash_t5.py:304-318 (layer_norm -> Wq / Wk / Wv),
// :159-164 (h + wo(act(layer_norm(h)))), :95-98 (FlashT5LayerNorm)).
//
//   out[m][n] = rowscale[m] * sum_k a[m][k] * w[n][k]   (+ res[m][n])
//
// a: (M, K) activations, w: (N, K) = an nn.Linear weight as stored, out / res: (M, N); 16-bit dtype, fp32 accumulation.
//  * NORM: rowscale[m] = rsqrt(mean_k a[m][k]^2 + eps), formed INSIDE the kernel from the A tiles on their way through LDS --
//    with w = W * diag(g) (the norm weight folded into the projection by the caller) this is Linear(RMSNorm(a; g)):
//    RMSNorm(a)[m][k] = a[m][k] * rstd[m] * g[k], so (RMSNorm(a) W^T)[m][n] = rstd[m] * sum_k a[m][k] (g[k] W[n][k]).
//    The normalised activation is never written (forward) nor kept for the backward.
//  * RES: the residual add of the sub-layer as the epilogue: out = res + round(a w^T), rounded twice like the two separate ops.
//
// One workgroup = 4 waves = a 128 x 128 output tile, each wave a 64 x 64 quadrant (four 32x32x16 MFMA accumulators); K in steps
// of 64: A and W tiles travel global -> LDS by DMA (double buffered, counted vmcnt) into the XOR-swizzled row-major images of
// attn_common.h; both operands are contracted along their rows, so every fragment is a ds_read_b128.  The accumulators hold
// the TRANSPOSED tile (lane = output row m, registers = 16 output columns): the row scale is one scalar per lane, and the
// epilogue stages the tile through LDS to leave as whole 256-byte rows.
#pragma once
#include "attn_common.h"
#include "attn_fwd64.h"  // dma16_asm, wait_dma_all

namespace fat5 {

// BK_ elements of K per stage, NS_ stages in the ring
template <int BK_, int NS_>
struct LinCfgT {
  static constexpr int BM = 128, BN = 128, BK = BK_, NS = NS_, NT = 256;
  static constexpr int IMG = rm_bytes<BK, BM>();      // one 128-row x BK-element image
  static constexpr int STAGE = 2 * IMG;               // A + W
  static constexpr int CROW = 2 * BN + 8;             // staged output row: 256 bytes + 8 (bank spread for the per-lane row writes)
  static constexpr int SMEM = NS * STAGE;
  static_assert(BM * CROW <= NS * STAGE, "output staging reuses the operand stages");
};
using LinCfg = LinCfgT<64, 2>;

template <bool BF16>
FAT5_DEV void dot2_acc(float& acc, uint32_t a) {
  if constexpr (BF16) asm("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(acc) : "v"(a));
  else asm("v_dot2c_f32_f16 %0, %1, %1" : "+v"(acc) : "v"(a));
}

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

struct LinArgs {
  const uint16_t *a, *w, *res;
  uint16_t* out;
  float* rstd_out;
  int64_t lda, ldw, ldr, ldo;
  int M, N, K;
  float eps;
};

template <bool BF16, bool NORM, bool RES, typename Cfg = LinCfg>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2 * Cfg::SMEM <= 160 * 1024 ? 2 : 1))) void linear_fused_kernel(const LinArgs p) {
  constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, NS = Cfg::NS, NT = Cfg::NT, IMG = Cfg::IMG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, lq = l & 31, hi = l >> 5;
  const int wm = w >> 1, wn = w & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int mt = blockIdx.x / tiles_n, nt = blockIdx.x - mt * tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk = p.K / BK;

  using Dma = DmaStage<BK, BM, NT>;
  static_assert(Dma::NV == 1, "one swizzle phase per piece");
  constexpr int PER = Dma::PER;  // 1-KiB pieces per wave, operand and stage
  Dma ast, wst;
  ast.init(p.lda, tid);
  wst.init(p.ldw, tid);
  const __amdgpu_buffer_rsrc_t ars = make_rows_rsrc(p.a + (int64_t)m0 * p.lda, p.lda, min(BM, p.M - m0), p.K);
  const __amdgpu_buffer_rsrc_t wrs = make_rows_rsrc(p.w + (int64_t)n0 * p.ldw, p.ldw, min(BN, p.N - n0), p.K);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)w * 1024u);
  auto dma_stage = [&](int kt, int slot) {
    const uint32_t koff = (uint32_t)(kt * BK * 2);
#pragma unroll
    for (int i = 0; i < PER; ++i) dma16_asm(ars, wave_lds + (uint32_t)(slot * Cfg::STAGE + NT * 16 * i), ast.voff[0], koff + ast.piece_step * i);
#pragma unroll
    for (int i = 0; i < PER; ++i) dma16_asm(wrs, wave_lds + (uint32_t)(slot * Cfg::STAGE + IMG + NT * 16 * i), wst.voff[0], koff + wst.piece_step * i);
  };
  // wait until all but the last `stages` requested stages of this wave have landed (LDS-DMA requests retire in order)
  auto wait_but = [&](int stages) {
    if constexpr (NS >= 4) if (stages >= 2) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER * 2) : "memory"); return; }
    if constexpr (NS >= 3) if (stages >= 1) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER * 1) : "memory"); return; }
    wait_dma_all();
  };

  FragAddr<BK> fa;
  fa.init(l);
  f32x16 acc[2][2];  // [nb][mb]: rows = 32 output columns n (C layout), lane column = output row m
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;
  // NORM: sum_k a[row][k]^2 of the lane's OWN two rows (64 wm + 32 mb + lq), from the A fragments on their way into the MFMAs (a lane
  // holds chunk 2 kk + hi of its row: the two halves of a wave meet in one cross-lane add at the end) -- the rows whose scale
  // the lane needs in the epilogue, so the statistics never go through LDS
  float ss[2] = {0.f, 0.f};

  // Ring: stage kt lives in slot kt % NS.  Iteration kt: wait for the own pieces of stage kt (everything but the NS - 2 younger
  // requests), barrier (all pieces of kt are in LDS, and every wave is done with the reads of kt - 1), request stage kt + NS - 1
  // into the slot of kt - 1, contract stage kt.  NS - 1 stages are in flight while one is contracted; ONE barrier per iteration.
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < nk) dma_stage(s0, s0);
  for (int kt = 0; kt < nk; ++kt) {
    const int slot = kt % NS;
    wait_but(min(NS - 2, nk - 1 - kt));
    __syncthreads();
    if (kt + NS - 1 < nk) dma_stage(kt + NS - 1, (kt + NS - 1) % NS);
    const char* imgA = smem + slot * Cfg::STAGE;
    const char* imgW = imgA + IMG;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      u32x4 af[2], wf[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) af[mb] = ld_rm<BK>(imgA, fa, 2 * wm + mb, kk);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) wf[nb] = ld_rm<BK>(imgW, fa, 2 * wn + nb, kk);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc[nb][mb] = mfma32<BF16>(wf[nb], af[mb], acc[nb][mb]);
      if constexpr (NORM) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int j = 0; j < 4; ++j) dot2_acc<BF16>(ss[mb], af[mb][j]);
      }
    }
  }
  __syncthreads();  // (the ring becomes the output staging area)
  float rs[2] = {1.f, 1.f};
  if constexpr (NORM) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const float tot = ss[mb] + __shfl_xor(ss[mb], 32, 64);
      rs[mb] = rsqrtf(tot / (float)p.K + p.eps);
      const int m = m0 + 64 * wm + 32 * mb + lq;
      if (nt == 0 && wn == 0 && hi == 0 && p.rstd_out && m < p.M) p.rstd_out[m] = rs[mb];
    }
  }
  // ---- epilogue: scale, round, stage the 128 x 128 tile in LDS as rows of m, leave as whole rows (+ residual) ----
  char* sC = smem;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int ml = 64 * wm + 32 * mb + lq;
    const float sc = rs[mb];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 wv;
        wv[0] = pack2<BF16>(acc[nb][mb][4 * g + 0] * sc, acc[nb][mb][4 * g + 1] * sc);
        wv[1] = pack2<BF16>(acc[nb][mb][4 * g + 2] * sc, acc[nb][mb][4 * g + 3] * sc);
        *reinterpret_cast<u32x2*>(sC + ml * Cfg::CROW + (64 * wn + 32 * nb + 8 * g + 4 * hi) * 2) = wv;
      }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < BM * (BN / 4) / NT; ++i) {
    const int id = tid + NT * i, row = id >> 5, c4 = id & 31;  // 32 eight-byte pieces per row
    const int m = m0 + row, n = n0 + 4 * c4;
    if (m < p.M && n < p.N) {
      u32x2 v = *reinterpret_cast<const u32x2*>(sC + row * Cfg::CROW + c4 * 8);
      if constexpr (RES) {
        const u32x2 r = *reinterpret_cast<const u32x2*>(p.res + (int64_t)m * p.ldr + n);
        v[0] = pack2<BF16>(cvt_lo<BF16>(v[0]) + cvt_lo<BF16>(r[0]), cvt_hi<BF16>(v[0]) + cvt_hi<BF16>(r[0]));
        v[1] = pack2<BF16>(cvt_lo<BF16>(v[1]) + cvt_lo<BF16>(r[1]), cvt_hi<BF16>(v[1]) + cvt_hi<BF16>(r[1]));
      }
      *reinterpret_cast<u32x2*>(p.out + (int64_t)m * p.ldo + n) = v;
    }
  }
}

}  // namespace fat5
