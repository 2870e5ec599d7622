// FlashAttention-2 backward with additive bias for gfx950 (CDNA4): two kernels.
//
//  attn_bwd_q_kernel   replaces `_bwd_preprocess` + `_bwd_q_kernel`
//                      (src/model/ops/flash_attention_v2_bias.py:516-556, :748-905):
//                      workgroup = (b, h, 32*NW query rows); computes delta = rowsum(o*do) for its
//                      rows (written to scratch for the kv kernel) and dQ.
//  attn_bwd_kv_kernel  replaces `_bwd_kv_kernel` (:559-745): workgroup = (b, h, 32*NW keys);
//                      computes dK, dV and the bias gradient (dense dS tile store, or per-diagonal
//                      sums in RPE mode -- never an (M,N) tensor in that mode).
//
// Orientation (see attn_common.h for the MFMA layouts):
//  q kernel : S^T[k][q] and dP^T[k][q] (keys in registers, q on lanes) so that dS^T feeds
//             dQ^T[d][q] += K^T . dS^T as the B operand without any cross-lane movement.
//  kv kernel: S[q][k] and dP[q][k] (q in registers, keys on lanes) so that P / dS feed
//             dV^T[d][k] += dO^T . P and dK^T[d][k] += Q^T . dS the same way.
// Every streamed operand lives in ONE row-major swizzled LDS image: fragments whose contraction runs
// along the row come from ds_read_b128, fragments whose contraction runs across rows (K^T, Q^T, dO^T)
// from ds_read_b64_tr_b16 of the same image.  Tiles are prefetched through buffer descriptors
// (rows past the end read as zero in hardware) one tile ahead, one barrier per tile.
// Probabilities are recomputed in the exp2 domain: p = exp2(s*c2 + (bias2 - L2)), with sm_scale*log2e
// and any tile-constant bias folded into the one FMA in front of v_exp_f32.
#pragma once
#include "attn_common.h"
#include "attn_fwd.h"  // load_bias_block
#include "diag_sum.h"  // per-diagonal sums of dS on the VALU (DPP row rotations)

namespace fat5 {

// =============================================================================================
// dQ kernel
// =============================================================================================
template <int D, int NW>
struct BwdQCfg {
  static constexpr int BM = 32 * NW;
  static constexpr int BN = 64;
  static constexpr int NT = 64 * NW;
  static constexpr int KRM = rm_bytes<D, BN>();
  static constexpr int VRM = rm_bytes<D, BN>();
  static constexpr int STAGE = KRM + VRM;
  static constexpr int BIASB = BM * BN * 2;  // dense mode: one (BM x 64) 16-bit bias tile per buffer
  static size_t smem(int R, int bias_mode) {
    return 2 * STAGE + (bias_mode == FAT5_BIAS_RPE1D ? rpe_table_bytes(R) + 16 : 0) +
           (bias_mode == FAT5_BIAS_DENSE ? 2 * (size_t)BIASB : 0);
  }
};

#ifndef FAT5_BWD_MINW
#define FAT5_BWD_MINW 2  // the register allocator must leave room for 2 waves per SIMD (<= 256 VGPR+AGPR)
#endif
template <int D, bool BF16, int BIAS, int NW>
FAT5_DEV void attn_bwd_q_body(const AttnArgs& a, const int bid) {
  using Cfg = BwdQCfg<D, NW>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT;
  constexpr int KK = D / 16, DB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sT = reinterpret_cast<float*>(smem + 2 * Cfg::STAGE) + kRpePad;  // (entry d of copy 0 at sT[d + R]; see attn_common.h)
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lq = l & 31, hi = l >> 5;
  int b, h, mblk;
  decode_unit(a, bid, a.n_mblk, b, h, mblk, (FAT5_CAUSAL_ORDER && a.causal) ? 1 : 0);
  int M = a.M, N = a.N;
  int64_t qoff = (int64_t)b * a.qs[0], koff = (int64_t)b * a.ks[0], voff = (int64_t)b * a.vs[0], ooff = (int64_t)b * a.os[0],
          dooff = (int64_t)b * a.dos[0], dqoff = (int64_t)b * a.dqs[0];
  int64_t stat_off = ((int64_t)b * a.H + h) * a.M;
  if (a.cu_q) {  // packed batch: (total, H, D) tensors, lse / delta (H, total_q)
    const int q0 = a.cu_q[b], k0 = a.cu_k[b];
    M = a.cu_q[b + 1] - q0;
    N = a.cu_k[b + 1] - k0;
    qoff = (int64_t)q0 * a.qs[2];
    ooff = (int64_t)q0 * a.os[2];
    dooff = (int64_t)q0 * a.dos[2];
    dqoff = (int64_t)q0 * a.dqs[2];
    koff = (int64_t)k0 * a.ks[2];
    voff = (int64_t)k0 * a.vs[2];
    stat_off = (int64_t)h * a.total_q + q0;
  }
  const int m0 = mblk * BM;
  if (m0 >= M) return;
  const uint16_t* qb = a.q + qoff + (int64_t)h * a.qs[1];
  const uint16_t* kb_ = a.k + koff + (int64_t)h * a.ks[1];
  const uint16_t* vb = a.v + voff + (int64_t)h * a.vs[1];
  const uint16_t* ob = a.o + ooff + (int64_t)h * a.os[1];
  const uint16_t* dob = a.dout + dooff + (int64_t)h * a.dos[1];
  uint16_t* dqb = a.dq + dqoff + (int64_t)h * a.dqs[1];

  const int P = N - M;
  int n_end = N;
  if (a.causal) n_end = min(N, m0 + BM + P);
  const int nt = n_end > 0 ? (n_end + BN - 1) / BN : 0;

  const int qrow0 = m0 + 32 * w;
  const int qrow = qrow0 + lq;
  const int qrow_c = min(qrow, M - 1);

  // Q and dO fragments (B operands), delta = rowsum(o * do)  (reference _bwd_preprocess, :516-556)
  u32x4 qf[KK], dof[KK];
  float dsum = 0.f;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    qf[kk] = load_frag16(qb + (int64_t)qrow_c * a.qs[2], kk, hi, a.dvalid);
    dof[kk] = load_frag16(dob + (int64_t)qrow_c * a.dos[2], kk, hi, a.dvalid);
    const u32x4 of = load_frag16(ob + (int64_t)qrow_c * a.os[2], kk, hi, a.dvalid);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dsum = fmaf(cvt_lo<BF16>(of[j]), cvt_lo<BF16>(dof[kk][j]), dsum);
      dsum = fmaf(cvt_hi<BF16>(of[j]), cvt_hi<BF16>(dof[kk][j]), dsum);
    }
  }
  const float delta = pair_sum(dsum);
  if (a.delta && qrow < M && hi == 0) a.delta[stat_off + qrow] = delta;  // (not needed by the fused launch)
  const float Lq = a.lse[stat_off + qrow_c];
  // p = exp2(x - L2); rows with L = -inf (fully masked) contribute nothing
  // rows without a visible key (lse = -inf) and rows whose every key is masked by a finfo.min bias (lse ~ -2e38: the
  // reference's `use_masking`; p = exp(s - L) has no digits left there, in the reference kernels neither) contribute nothing
  const float nL2 = (Lq < kDeadRowLse) ? -INFINITY : -Lq * kLog2e;
  if (a.stat2 && hi == 0 && qrow < (M + 31) / 32 * 32) {
    // accumulator initial values of the 64-key dK/dV body (attn_bwd64.h): S' = Q K^T - L/scale, dP' = dO V^T - delta; rows that
    // contribute nothing (past M, dead) start at -inf * sign(scale) -> p = 0
    float* st = a.stat2 + (((int64_t)b * a.H + h) * ((M + 31) / 32) + (qrow >> 5)) * 64 + (qrow & 31);
    const bool live = qrow < M && !(Lq < kDeadRowLse);
    st[0] = live ? -Lq / a.scale : (a.scale > 0.f ? -INFINITY : INFINITY);
    st[32] = qrow < M ? -delta : 0.f;
  }

  const float* sTa = sT;  // this lane's aligned copy of the table
  if constexpr (BIAS == FAT5_BIAS_RPE1D) sTa = sT + ((a.R - qrow) & 3) * rpe_n1p(a.R);
  const uint16_t* brow = nullptr;
  uint16_t* dsrow = nullptr;  // this lane's row of the rounded dS tile (dense bias gradient), or nullptr
  uint16_t* dstile = nullptr; // the (b, h) slice of the dS output
  // dense bias tiles: global -> LDS beside K / V (see the forward)
  using BDma = DmaStage<BN, BM, NT, true>;
  BDma bdm;
  BiasTileReader brd;
  char* sB = smem + 2 * Cfg::STAGE;  // [2][BM][64] 16-bit
  const bool bias_dma = (BIAS == FAT5_BIAS_DENSE) && a.bias_dma && a.cu_q == nullptr;
  __amdgpu_buffer_rsrc_t brs = make_rows_rsrc(qb, a.qs[2], 0, D);
  if constexpr (BIAS == FAT5_BIAS_DENSE) {
    if (bias_dma) {
      bdm.init(a.bs[2], tid);
      brd.init(32 * w + lq, hi);
      brs = make_rows_rsrc(a.bias + (int64_t)b * a.bs[0] + (int64_t)h * a.bs[1] + (int64_t)m0 * a.bs[2], a.bs[2], M - m0, N);
    }
    brow = a.bias + (int64_t)b * a.bs[0] + (int64_t)h * a.bs[1] + (int64_t)qrow_c * a.bs[2];
    if (a.ds_out) {
      dstile = a.ds_out + (int64_t)b * a.dss[0] + (int64_t)h * a.dss[1];
      dsrow = dstile + (int64_t)qrow_c * a.dss[2];
    }
  }

  FragAddr<D> fa;
  fa.init(l);
  f32x16 dqacc[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;

  DmaStage<D, BN, NT> kst, vst;  // K / V tiles global -> LDS directly (see the dK/dV body)
  kst.init(a.ks[2], tid, a.dvalid);
  vst.init(a.vs[2], tid, a.dvalid);
  const __amdgpu_buffer_rsrc_t krs = make_rows_rsrc(kb_, a.ks[2], N, a.dvalid);
  const __amdgpu_buffer_rsrc_t vrs = make_rows_rsrc(vb, a.vs[2], N, a.dvalid);
  const uint32_t kstride_b = (uint32_t)a.ks[2] * 2u, vstride_b = (uint32_t)a.vs[2] * 2u;
  if (nt > 0) {
    kst.issue(krs, 0, smem, tid);
    vst.issue(vrs, 0, smem + Cfg::KRM, tid);
    if constexpr (BIAS == FAT5_BIAS_DENSE)
      if (bias_dma) bdm.issue(brs, 0, sB, tid);
  }
  // (table after the first tile's DMA is in flight: one memory round trip for the prologue)
  if constexpr (BIAS == FAT5_BIAS_RPE1D) rpe_table_fill(sT - kRpePad, a.rpe1d + (int64_t)h * (2 * a.R + 1), a.R, tid, NT);
  __syncthreads();
  // see attn_fwd.h: keep the compiler's waitcnt model from chaining the loop's MFMAs to the tile prefetch
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) asm volatile("" ::"v"(qf[kk]), "v"(dof[kk]));
  asm volatile("" ::"v"(nL2), "v"(delta));
  // dP^T accumulators start at -delta (this lane's query row): dS = P * dP' needs no per-element subtract
  f32x16 ndelta16;
#pragma unroll
  for (int r = 0; r < 16; ++r) ndelta16[r] = -delta;

  const float c2 = a.scale * kLog2e;
  float cst_neg = 0.f, cst_pos = 0.f;
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    cst_neg = sT[0];
    cst_pos = sT[2 * a.R];
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  auto tile = [&]<bool FAST, int BUF>(int t, float cst) {
    const int n0 = t * BN;
    const char* sK = smem + BUF * Cfg::STAGE;
    const char* sV = sK + Cfg::KRM;
    // dense dS of a full tile goes through LDS (whole-row stores); tails and unaligned outputs use per-lane stores
    const bool ds_lds = (BIAS == FAT5_BIAS_DENSE) && bias_dma && dstile != nullptr && a.ds_vec8 && (n0 + BN <= N);
    const bool more = (t + 1 < nt);
    if (more) {
      char* nK = smem + (BUF ^ 1) * Cfg::STAGE;  // (its last readers passed the previous tile's barrier)
      kst.issue(krs, (uint32_t)(n0 + BN) * kstride_b, nK, tid);
      vst.issue(vrs, (uint32_t)(n0 + BN) * vstride_b, nK + Cfg::KRM, tid);
      if constexpr (BIAS == FAT5_BIAS_DENSE)
        if (bias_dma) bdm.issue(brs, (uint32_t)(n0 + BN) * 2u, sB + (BUF ^ 1) * Cfg::BIASB, tid);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int nb = n0 + 32 * kb;
      u32x4 kf[KK], vf[KK];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        kf[kk] = ld_rm<D>(sK, fa, kb, kk);
        vf[kk] = ld_rm<D>(sV, fa, kb, kk);
      }
      f32x16 s, dp;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        s = mfma32<BF16>(kf[kk], qf[kk], kk == 0 ? zero16 : s);      // S^T  = K Q^T
        dp = mfma32<BF16>(vf[kk], dof[kk], kk == 0 ? ndelta16 : dp);  // dP^T - delta = V dO^T - delta
      }
      if constexpr (FAST) {
        const float ad = cst + nL2;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fast_exp2(fmaf(s[r], c2, ad)) * dp[r];
      } else {
        if constexpr (BIAS == FAT5_BIAS_DENSE) {
          float bv[16];
          if (bias_dma) brd.template load<BF16>(sB + BUF * Cfg::BIASB, kb, bv);
          else load_bias_block<BF16>(brow, nb, hi, N, a.bias_vec4 && (nb + 32 <= N), bv);
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], c2, bias_log2(bv[r]) + nL2);
        } else if constexpr (BIAS == FAT5_BIAS_RPE1D) {
          // one straight-line path for far, edge and band blocks: four aligned 16-byte reads of this lane's padded table copy
          // (the window start is clamped into the copy -- attn_common.h)
          const int R = a.R;
          const float4* tp4 = reinterpret_cast<const float4*>(sTa + rpe_clamp_asc(R + nb + 4 * hi - qrow - ((R - qrow) & 3), R));
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 bq = tp4[2 * g];
            s[4 * g + 0] = fmaf(s[4 * g + 0], c2, bq.x + nL2);
            s[4 * g + 1] = fmaf(s[4 * g + 1], c2, bq.y + nL2);
            s[4 * g + 2] = fmaf(s[4 * g + 2], c2, bq.z + nL2);
            s[4 * g + 3] = fmaf(s[4 * g + 3], c2, bq.w + nL2);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], c2, nL2);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r]) * dp[r];  // dS = P (dP - delta)   (:713)
        const bool nmask = nb + 32 > N;
        const bool cmask = a.causal && (nb + 31 > qrow0 + P);
        if (nmask || cmask) {
          const int lim = a.causal ? min(N - 1, qrow + P) : N - 1;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (nb + crow(r, hi) > lim) s[r] = 0.f;
        }
      }
      // dQ^T[d][q] += K^T[d][key] dS^T[key][q]   (dS rounded to the input dtype like the reference, :720)
      u32x4 dsv[2];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) dsv[t2] = pack8<BF16>(s, t2);
      if constexpr (BIAS == FAT5_BIAS_DENSE) {
        // dense bias gradient: this orientation holds 4 consecutive keys of a query row per register group, so the
        // rounded dS tile leaves as 8-byte stores (the dK/dV body would need sixteen 2-byte stores per lane)
        if (ds_lds) {
          // full tile: park the rounded dS block in the (already consumed) bias tile, same positions; it leaves the
          // workgroup as whole 128-byte rows at the end of the tile
          brd.store(sB + BUF * Cfg::BIASB, kb, dsv);
        } else if (dsrow && qrow < M) {
          if (a.ds_vec4 && nb + 32 <= N) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const u32x2 w2 = {dsv[g >> 1][2 * (g & 1)], dsv[g >> 1][2 * (g & 1) + 1]};
              *reinterpret_cast<u32x2*>(dsrow + nb + 8 * g + 4 * hi) = w2;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int n = nb + crow(r, hi);
              if (n < N) dsrow[n] = to16<BF16>(s[r]);
            }
          }
        }
      }
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const u32x4 dsb = dsv[t2];
#pragma unroll
        for (int db = 0; db < DB; ++db) dqacc[db] = mfma32<BF16>(ld_tr<D>(sK, fa, kb, t2, db), dsb, dqacc[db]);
      }
    }
    if constexpr (BIAS == FAT5_BIAS_DENSE) {
      if (ds_lds) {
        // this wave's 32 rows x 128 bytes of dS (written by this wave only): 16-byte pieces, 8 lanes per row -> every
        // store instruction writes 8 whole rows of the tile
        const char* tl = sB + BUF * Cfg::BIASB + (32 * w) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int j = l + 64 * i, row = j >> 3, slot = j & 7;
          const u32x4 vv = *reinterpret_cast<const u32x4*>(tl + row * 128 + slot * 16);
          const int m = m0 + 32 * w + row;
          const int n = n0 + ((slot ^ swz<64>(row)) << 3);
          if (m < M) *reinterpret_cast<u32x4*>(dstile + (int64_t)m * a.dss[2] + n) = vv;
        }
      }
    }
    __syncthreads();
  };

  // tile classes as in the forward (boundaries rounded to even tile indices: one body per loop)
  int ta = 0, tb0 = 0, tb1 = 0;
  float cst_a = 0.f, cst_b = 0.f;
  if (BIAS != FAT5_BIAS_DENSE) {
    int t_full = N / BN;
    if (a.causal) t_full = min(t_full, max(0, (m0 + P + 1) / BN));
    t_full = min(t_full, nt);
    if constexpr (BIAS == FAT5_BIAS_RPE1D) {
      const int lim_a = m0 - a.R - (BN - 1);
      ta = lim_a >= 0 ? min(t_full, lim_a / BN + 1) : 0;
      const int lo = m0 + BM - 1 + a.R;
      tb0 = min(t_full, max(ta, (lo + BN - 1) / BN));
      tb1 = t_full;
      cst_a = cst_neg;
      cst_b = cst_pos;
    } else {
      ta = t_full;
    }
    ta &= ~1;
    tb0 = (tb0 + 1) & ~1;
    tb1 &= ~1;
    if (tb1 < tb0) tb0 = tb1 = ta;
  }
  int t = 0;
  for (; t < ta; t += 2) {
    tile.template operator()<true, 0>(t, cst_a);
    tile.template operator()<true, 1>(t + 1, cst_a);
  }
  const int g0 = min(max(tb0, ta), nt & ~1);
  for (; t < g0; t += 2) {
    tile.template operator()<false, 0>(t, 0.f);
    tile.template operator()<false, 1>(t + 1, 0.f);
  }
  for (; t < tb1; t += 2) {
    tile.template operator()<true, 0>(t, cst_b);
    tile.template operator()<true, 1>(t + 1, cst_b);
  }
  for (; t + 1 < nt; t += 2) {
    tile.template operator()<false, 0>(t, 0.f);
    tile.template operator()<false, 1>(t + 1, 0.f);
  }
  if (t < nt) tile.template operator()<false, 0>(t, 0.f);

  if (qrow < M) {
    const float scale = a.scale;
    uint16_t* drow = dqb + (int64_t)qrow * a.dqs[2];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 wv;
        wv[0] = pack2<BF16>(dqacc[db][4 * g + 0] * scale, dqacc[db][4 * g + 1] * scale);
        wv[1] = pack2<BF16>(dqacc[db][4 * g + 2] * scale, dqacc[db][4 * g + 3] * scale);
        if (32 * db + 8 * g + 4 * hi < a.dvalid) *reinterpret_cast<u32x2*>(drow + 32 * db + 8 * g + 4 * hi) = wv;
      }
  }
}

// =============================================================================================
// dK / dV / dBias kernel
// =============================================================================================
template <int D, int NW>
struct BwdKVCfg {
  static constexpr int BNK = 32 * NW;  // keys per workgroup
  static constexpr int BMQ = 64;       // query rows per loop step
  static constexpr int NT = 64 * NW;
  static constexpr int QRM = rm_bytes<D, BMQ>();
  static constexpr int STAT = BMQ * 4 * 2;  // -L*log2e and delta for the BMQ rows
  static constexpr int STAGE = 2 * QRM + STAT;
  static constexpr int BIASB = BMQ * BNK * 2;  // dense mode: one (64 query rows x BNK keys) 16-bit bias tile per buffer
  static size_t smem(int R, int bias_mode) {
    // rpe: table + one private diagonal accumulator per wave
    return 2 * STAGE + (bias_mode == FAT5_BIAS_RPE1D ? rpe_off(R) : 0) +
           (bias_mode == FAT5_BIAS_DENSE ? 2 * (size_t)BIASB : 0);
  }
  static __host__ __device__ size_t rpe_off(int R) { return (rpe_table_bytes(R) + (size_t)(2 * R + 1) * 4 * NW + 63) / 64 * 64; }
};

// SELFD: delta = rowsum(o * do) is formed here from the O tile (prefetched beside dO) instead of being read from the
// dQ kernel's scratch -- removes the only dependency between the two kernels so that they can share one launch.
template <int D, bool BF16, int BIAS, int NW, bool SELFD>
FAT5_DEV void attn_bwd_kv_body(const AttnArgs& a, const int bid) {
  using Cfg = BwdKVCfg<D, NW>;
  constexpr int BNK = Cfg::BNK, BMQ = Cfg::BMQ, NT = Cfg::NT;
  constexpr int KK = D / 16, DB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lq = l & 31, hi = l >> 5;
  int b, h, nblk;
  decode_unit(a, bid, a.n_nblk, b, h, nblk, (FAT5_CAUSAL_ORDER && a.causal) ? 2 : 0);
  const int bh = b * a.H + h;  // (row of this (batch, head) in the partial-sum workspace)
  int M = a.M, N = a.N;
  int64_t qoff = (int64_t)b * a.qs[0], koff = (int64_t)b * a.ks[0], voff = (int64_t)b * a.vs[0], ooff = (int64_t)b * a.os[0],
          dooff = (int64_t)b * a.dos[0], dkoff = (int64_t)b * a.dks[0], dvoff = (int64_t)b * a.dvs[0];
  int64_t stat_off = ((int64_t)b * a.H + h) * a.M;
  if (a.cu_q) {  // packed batch: (total, H, D) tensors, lse / delta (H, total_q)
    const int q0 = a.cu_q[b], k0 = a.cu_k[b];
    M = a.cu_q[b + 1] - q0;
    N = a.cu_k[b + 1] - k0;
    qoff = (int64_t)q0 * a.qs[2];
    ooff = (int64_t)q0 * a.os[2];
    dooff = (int64_t)q0 * a.dos[2];
    koff = (int64_t)k0 * a.ks[2];
    voff = (int64_t)k0 * a.vs[2];
    dkoff = (int64_t)k0 * a.dks[2];
    dvoff = (int64_t)k0 * a.dvs[2];
    stat_off = (int64_t)h * a.total_q + q0;
  }
  const int n0 = nblk * BNK;
  if (n0 >= N) {
    // (packed batches: key blocks past the end of a short sequence still own a partial row of the diagonal sums)
    if constexpr (BIAS == FAT5_BIAS_RPE1D) {
      if (a.drpe_part) {
        float* out = a.drpe_part + ((int64_t)bh * a.n_nblk + nblk) * (2 * a.R + 1);
        for (int i2 = tid; i2 < 2 * a.R + 1; i2 += NT) out[i2] = 0.f;
      }
    }
    return;
  }
  const uint16_t* qb = a.q + qoff + (int64_t)h * a.qs[1];
  const uint16_t* kb_ = a.k + koff + (int64_t)h * a.ks[1];
  const uint16_t* vb = a.v + voff + (int64_t)h * a.vs[1];
  const uint16_t* dob = a.dout + dooff + (int64_t)h * a.dos[1];
  uint16_t* dkb = a.dk + dkoff + (int64_t)h * a.dks[1];
  uint16_t* dvb = a.dv + dvoff + (int64_t)h * a.dvs[1];

  const int P = N - M;
  const int krow0 = n0 + 32 * w;
  const int krow = krow0 + lq;
  const int krow_c = min(krow, N - 1);

  // K and V fragments (B operands) for this lane's key
  u32x4 kf[KK], vf[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    kf[kk] = load_frag16(kb_ + (int64_t)krow_c * a.ks[2], kk, hi, a.dvalid);
    vf[kk] = load_frag16(vb + (int64_t)krow_c * a.vs[2], kk, hi, a.dvalid);
  }

  // RPE: table + per-wave private diagonal accumulators in LDS
  float* sT = reinterpret_cast<float*>(smem + 2 * Cfg::STAGE) + kRpePad;  // (entry d of copy 0 at sT[d + R]; see attn_common.h)
  const int n1 = 2 * a.R + 1;
  float* sD0 = sT - kRpePad + 4 * rpe_n1p(a.R);  // NW wave-private (2R+1) diagonal accumulators behind the four table copies
  float* sD = sD0 + w * n1;
  // Per-diagonal sums of dS (the gradient of the bias generator) on the VALU: diag_sum.h -- one DPP row rotation per element and a
  // few cross-lane operations per block, no LDS round trip (rounds 1-3: a skewed LDS tile + column sums on the matrix pipe, before
  // that 16 LDS float atomics per lane: ds_add_f32 costs ~600 cycles per wave instruction on gfx950).  `dcar` carries the partial
  // diagonals from one 32-row block to the next (query blocks ascend); a finished diagonal is handed over exactly once.
  DiagCarry dcar;
  diag_carry_zero(dcar);
  bool diag_run = false;  // (wave-uniform) dcar holds partial diagonals of the block at diag_mb
  int diag_mb = 0;
  float far_neg = 0.f, far_pos = 0.f;
  // lanes 0..31 of F: the finished sums of the diagonals base + lane; far bins in registers, near bins stored into the wave-private
  // (zero-initialised) array
  auto diag_emit = [&](const float F, const int base) {
    if (hi == 0) {
      const int d = base + lq;
      if (d <= -a.R) far_neg += F;
      else if (d >= a.R) far_pos += F;
      else sD[d + a.R] = F;  // every near diagonal of a wave is finished exactly once (query blocks ascend)
    }
  };
  auto diag_flush = [&]() {  // the end of a run: the two 32-diagonal windows below the last block's leave the carry
    if (diag_run) {
      DiagStep z;
      diag_step_zero(z);
      const int base = krow0 - diag_mb;
      diag_emit(diag_finish(dcar, z, l), base - 32);
      diag_emit(diag_finish(dcar, z, l), base - 64);
      diag_run = false;
    }
  };
  auto tree16 = [](const f32x16& x) {  // balanced tree: no long dependent chain, one live register afterwards
    const float t0 = (x[0] + x[1]) + (x[2] + x[3]), t1 = (x[4] + x[5]) + (x[6] + x[7]);
    const float t2 = (x[8] + x[9]) + (x[10] + x[11]), t3 = (x[12] + x[13]) + (x[14] + x[15]);
    return (t0 + t1) + (t2 + t3);
  };
  const bool want_drpe = (BIAS == FAT5_BIAS_RPE1D) && (a.drpe_part != nullptr);
  const uint16_t* bbase = nullptr;
  // Dense bias: lane = key, registers = 16 query rows -> a direct read is sixteen 2-byte gathers from 16 rows (0.7 TB/s
  // measured).  Instead the (64 x BNK) tile of this workgroup goes global -> LDS in 16-byte pieces beside Q / dO (row
  // major, unswizzled) and the lanes pick their elements with ds_read_u16 (base + immediate offsets).
  using BDma = DmaStage<BNK, BMQ, NT, false>;
  BDma bdm;
  char* sB = smem + 2 * Cfg::STAGE;  // [2][BMQ][BNK] 16-bit
  const bool bias_dma = (BIAS == FAT5_BIAS_DENSE) && a.bias_dma;
  if constexpr (BIAS == FAT5_BIAS_DENSE) {
    bbase = a.bias + (int64_t)b * a.bs[0] + (int64_t)h * a.bs[1];
    bdm.init(a.bs[2], tid);
  }

  FragAddr<D> fa;
  fa.init(l);
  f32x16 dkacc[DB], dvacc[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkacc[i][r] = 0.f; dvacc[i][r] = 0.f; }

  // query-tile range: causal => only rows with q + P >= n0 see this key block
  int m_lo = 0;
  if (a.causal) m_lo = max(0, n0 - P) / BMQ * BMQ;
  const int mt0 = m_lo / BMQ;
  const int mt1 = (M + BMQ - 1) / BMQ;
  const int ntile = mt1 - mt0;

  // Q / dO tiles go global -> LDS directly (no staging registers: this body sits at the 256-VGPR limit of two
  // waves per SIMD, and every spilled register is a scratch round trip on the critical path of a 1-2 waves/SIMD grid)
  using Dma = DmaStage<D, BMQ, NT>;
  Dma qdm, dodm;
  qdm.init(a.qs[2], tid, a.dvalid);
  dodm.init(a.dos[2], tid, a.dvalid);
  const __amdgpu_buffer_rsrc_t qrs = make_rows_rsrc(qb, a.qs[2], M, a.dvalid);
  const __amdgpu_buffer_rsrc_t dors = make_rows_rsrc(dob, a.dos[2], M, a.dvalid);
  const uint32_t qstride_b = (uint32_t)a.qs[2] * 2u, dostride_b = (uint32_t)a.dos[2] * 2u;
  const __amdgpu_buffer_rsrc_t brs = make_rows_rsrc(bias_dma ? bbase + n0 : dob, bias_dma ? a.bs[2] : a.dos[2], M,
                                                    bias_dma ? min(BNK, N - n0) : D);
  const uint32_t bstride_b = (uint32_t)a.bs[2] * 2u;
  // SELFD: the O pieces that pair with this thread's dO pieces (same row / same swizzled chunk), through registers;
  // rows >= M read as zero -> delta 0
  u32x4 ofr[SELFD ? Dma::PER : 1];
  const uint16_t* ob = SELFD ? a.o + ooff + (int64_t)h * a.os[1] : nullptr;
  const __amdgpu_buffer_rsrc_t ors = make_rows_rsrc(SELFD ? ob : dob, SELFD ? a.os[2] : a.dos[2], M, D);
  Dma odm;
  odm.init(a.os[2], tid, a.dvalid);
  const uint32_t ostride_b = (uint32_t)a.os[2] * 2u;
  // Row statistics of a tile, staged in the form the MFMA accumulators are INITIALISED with (C operand of the first
  // k-step instead of zero): S' = Q K^T - L/scale, so that p = exp2(S'*c2 + bias) needs no per-element "+ (-L)", and
  // dP' = dO V^T - delta.  Rows that contribute nothing (m >= M, or L = -inf) start at -inf*sign(scale): p = 0.
  const float inv_scale = 1.f / a.scale;  // scale != 0 (the API substitutes 1e-30 for an exact zero)
  const float ninf_c = a.scale > 0.f ? -INFINITY : INFINITY;
  float st_l = 0.f, st_d = 0.f;
  auto load_stats = [&](int mrow0) {
    if (tid < BMQ) {
      const int m = mrow0 + tid;
      st_l = ninf_c;
      st_d = 0.f;
      if (m < M) {
        const float L = a.lse[stat_off + m];
        st_l = (L < kDeadRowLse) ? ninf_c : -L * inv_scale;  // (see kDeadRowLse; -L / scale would overflow for a masked row)
        if constexpr (!SELFD) st_d = -a.delta[stat_off + m];
      }
    }
    if constexpr (SELFD) {
#pragma unroll
      for (int i = 0; i < Dma::PER; ++i)
        ofr[i] = odm.load_piece(ors, (uint32_t)mrow0 * ostride_b, i);
    }
  };
  // `st` = the LDS buffer the tile was DMA'd into.  SELFD reads this thread's OWN dO pieces back (visible to the issuing
  // wave after its vmcnt), dots them with the O pieces, and a butterfly over the C = D/8 consecutive lanes of a row
  // finishes delta.
  auto store_stats = [&](char* st) {
    float* sL = reinterpret_cast<float*>(st + 2 * Cfg::QRM);
    if (tid < BMQ) {
      sL[tid] = st_l;
      if constexpr (!SELFD) sL[BMQ + tid] = st_d;
    }
    if constexpr (SELFD) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < Dma::PER; ++i) {
        const u32x4 dv = *reinterpret_cast<const u32x4*>(st + Cfg::QRM + Dma::own_off(tid, i));
        float pd = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pd = fmaf(cvt_lo<BF16>(ofr[i][j]), cvt_lo<BF16>(dv[j]), pd);
          pd = fmaf(cvt_hi<BF16>(ofr[i][j]), cvt_hi<BF16>(dv[j]), pd);
        }
#pragma unroll
        for (int off = 1; off < Dma::C; off <<= 1) pd += __shfl_xor(pd, off, 64);
        const int id = tid + NT * i;
        if (id % Dma::C == 0) sL[BMQ + id / Dma::C] = -pd;
      }
    }
  };
  if (ntile > 0) {
    qdm.issue(qrs, (uint32_t)(mt0 * BMQ) * qstride_b, smem, tid);
    dodm.issue(dors, (uint32_t)(mt0 * BMQ) * dostride_b, smem + Cfg::QRM, tid);
    if constexpr (BIAS == FAT5_BIAS_DENSE)
      if (bias_dma) bdm.issue(brs, (uint32_t)(mt0 * BMQ) * bstride_b, sB, tid);
    load_stats(mt0 * BMQ);
  }
  // (after the first tile's loads are in flight: ONE memory round trip for the whole prologue instead of two)
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    rpe_table_fill(sT - kRpePad, a.rpe1d + (int64_t)h * n1, a.R, tid, NT);
    for (int i = tid; i < n1 * NW; i += NT) sD0[i] = 0.f;
  }
  if (ntile > 0) store_stats(smem);
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) asm volatile("" ::"v"(kf[kk]), "v"(vf[kk]));

  const float c2 = a.scale * kLog2e;
  float cst_neg = 0.f, cst_pos = 0.f;
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    cst_neg = sT[0];
    cst_pos = sT[2 * a.R];
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // One 64-query tile.  FAST: nothing of the tile is masked for any key of the workgroup and the bias is the
  // constant `cst` (none / all-far RPE); FARSIDE (RPE only): -1 far-negative, +1 far-positive, 0 not applicable.
  auto tile = [&]<bool FAST, int BUF, int FARSIDE>(int mt, float cst) {
    const int mrow0 = mt * BMQ;
    const char* sQ = smem + BUF * Cfg::STAGE;
    const char* sDO = sQ + Cfg::QRM;
    const float* sL = reinterpret_cast<const float*>(sDO + Cfg::QRM);
    const bool more = (mt + 1 < mt1);
    if (more) {  // next tile straight into the other buffer (its last readers passed the previous tile's barrier)
      char* nb_ = smem + (BUF ^ 1) * Cfg::STAGE;
      qdm.issue(qrs, (uint32_t)(mrow0 + BMQ) * qstride_b, nb_, tid);
      dodm.issue(dors, (uint32_t)(mrow0 + BMQ) * dostride_b, nb_ + Cfg::QRM, tid);
      if constexpr (BIAS == FAT5_BIAS_DENSE)
        if (bias_dma) bdm.issue(brs, (uint32_t)(mrow0 + BMQ) * bstride_b, sB + (BUF ^ 1) * Cfg::BIASB, tid);
      load_stats(mrow0 + BMQ);
    }
#pragma unroll
    for (int qbk = 0; qbk < 2; ++qbk) {
      const int mb = mrow0 + 32 * qbk;  // first query row of this 32-row block
      // S = Q K^T first (its own fragment registers die before dP = dO V^T is formed: the kernel runs at the
      // 256-register limit of 2 waves per SIMD)
      f32x16 s, dp;
      // accumulator initial values: rows mb + 8g + 4hi + (0..3)  <->  registers 4g .. 4g+3
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 nl4 = *reinterpret_cast<const float4*>(sL + 32 * qbk + 8 * g + 4 * hi);
        s[4 * g] = nl4.x; s[4 * g + 1] = nl4.y; s[4 * g + 2] = nl4.z; s[4 * g + 3] = nl4.w;
      }
      {
        u32x4 qa[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) qa[kk] = ld_rm<D>(sQ, fa, qbk, kk);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) s = mfma32<BF16>(qa[kk], kf[kk], s);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 dl4 = *reinterpret_cast<const float4*>(sL + BMQ + 32 * qbk + 8 * g + 4 * hi);
        dp[4 * g] = dl4.x; dp[4 * g + 1] = dl4.y; dp[4 * g + 2] = dl4.z; dp[4 * g + 3] = dl4.w;
      }
      {
        u32x4 da[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) da[kk] = ld_rm<D>(sDO, fa, qbk, kk);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) dp = mfma32<BF16>(da[kk], vf[kk], dp);
      }
      // C layout: lane (key = krow, hi), register r <-> query row mb + crow(r, hi)
      f32x16 p;
      if constexpr (FAST) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          p[r] = fast_exp2(fmaf(s[r], c2, cst));
          s[r] = p[r] * dp[r];
        }
      } else {
        if constexpr (BIAS == FAT5_BIAS_DENSE) {
          if (bias_dma) {
            // this lane's column of the staged tile: row 32*qbk + crow(r, hi), key 32*w + lq
            const uint16_t* tb = reinterpret_cast<const uint16_t*>(sB + BUF * Cfg::BIASB) + (32 * qbk + 4 * hi) * BNK + 32 * w + lq;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              s[r] = fmaf(s[r], c2, bias_log2(cvt16<BF16>(bias_clamp1<BF16>(tb[((r & 3) + 8 * (r >> 2)) * BNK]))));
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = mb + crow(r, hi);
              const float bvl = (m < M && krow < N) ? cvt16<BF16>(bias_clamp1<BF16>(bbase[(int64_t)m * a.bs[2] + krow])) : 0.f;
              s[r] = fmaf(s[r], c2, bias_log2(bvl));
            }
          }
        } else if constexpr (BIAS == FAT5_BIAS_RPE1D) {
          // one straight-line path for far, edge and band blocks: entries of r = 4g..4g+3 are 4 consecutive DEscending entries of
          // this lane's padded table copy (alignment (R + krow - 3) & 3 is constant): four aligned 16-byte reads, window clamped
          const int R = a.R;
          const int al = (R + krow - 3) & 3;
          const float* tb = sT + al * rpe_n1p(R) + rpe_clamp_desc(R + krow - mb - 4 * hi - 3 - al, R);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 bq = *reinterpret_cast<const float4*>(tb - 8 * g);
            s[4 * g + 0] = fmaf(s[4 * g + 0], c2, bq.w);
            s[4 * g + 1] = fmaf(s[4 * g + 1], c2, bq.z);
            s[4 * g + 2] = fmaf(s[4 * g + 2], c2, bq.y);
            s[4 * g + 3] = fmaf(s[4 * g + 3], c2, bq.x);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] *= c2;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          p[r] = fast_exp2(s[r]);
          s[r] = p[r] * dp[r];
        }
        const bool nmask = krow0 + 32 > N;
        const bool cmask = a.causal && (krow0 + 31 > mb + P);
        if (nmask || cmask) {
          // key visible to query m iff krow < N and (causal) krow <= m + P  <=>  crow(r,hi) >= thr  (per-lane thr)
          int thr = a.causal ? (krow - P - mb) : -(1 << 30);
          if (krow >= N) thr = 1 << 30;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool ok = crow(r, hi) >= thr;
            p[r] = ok ? p[r] : 0.f;
            s[r] = ok ? s[r] : 0.f;
          }
        }
      }
      // (dense bias gradient: the rounded dS tile is written by the dQ body, whose layout allows 8-byte stores)
      // operands of the two output GEMMs (P and dS rounded to the input dtype like the reference, :702 / :720)
      u32x4 pbv[2], dsv[2];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        pbv[t2] = pack8<BF16>(p, t2);
        dsv[t2] = pack8<BF16>(s, t2);
      }
      if constexpr (BIAS == FAT5_BIAS_RPE1D) {
        if (want_drpe) {
          if constexpr (FAST) {
            diag_flush();
            if constexpr (FARSIDE < 0) far_neg += tree16(s); else far_pos += tree16(s);
          } else {
            const int R = a.R;
            const int dmin = krow0 - (mb + 31), dmax = krow0 + 31 - mb;
            if (dmax <= -R || dmin >= R) {
              diag_flush();
              const float acc = tree16(s);
              if (dmax <= -R) far_neg += acc; else far_pos += acc;
            } else {
              // the block's dS (fp32, masked elements zero) onto its diagonals
              DiagStep st;
              diag_step_zero(st);
              static_for<16>([&](auto ri) { diag_elem<decltype(ri)::value>(st, s[decltype(ri)::value], l & 15); });
              diag_emit(diag_finish(dcar, st, l), krow0 - mb);
              diag_run = true;
              diag_mb = mb;
            }
          }
        }
      }
      // ---- dV^T += dO^T P ;  dK^T += Q^T dS   (A fragments: transposed reads of the dO / Q images) -----
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          dvacc[db] = mfma32<BF16>(ld_tr<D>(sDO, fa, qbk, t2, db), pbv[t2], dvacc[db]);
          dkacc[db] = mfma32<BF16>(ld_tr<D>(sQ, fa, qbk, t2, db), dsv[t2], dkacc[db]);
        }
      }
    }
    if (more) store_stats(smem + (BUF ^ 1) * Cfg::STAGE);
    __syncthreads();  // (carries the vmcnt(0) that retires this wave's DMA pieces)
  };

  // Tile classes over mt in [mt0, mt1) relative index i = mt - mt0 (even boundaries, see attn_fwd.h).
  // Increasing m means decreasing delta = k - q:  [far-positive FAST] [generic] [far-negative FAST] [generic tail].
  int ia = 0, ib0 = 0, ib1 = 0;  // FAST for i < ia (cst_a) and ib0 <= i < ib1 (cst_b)
  float cst_a = 0.f, cst_b = 0.f;
  const bool key_tail = n0 + BNK > N;
  if (BIAS != FAT5_BIAS_DENSE && !key_tail) {
    int mt_full = M / BMQ;  // tiles without an M tail
    int mt_first = mt0;     // causal: first tile where every key of the workgroup is visible to every row
    if (a.causal) mt_first = max(mt0, (max(0, n0 + BNK - 1 - P) + BMQ - 1) / BMQ);
    if constexpr (BIAS == FAT5_BIAS_RPE1D) {
      // far-positive: n0 - (mrow0 + BMQ - 1) >= R  <=>  mrow0 <= n0 - R - (BMQ - 1)
      const int lim_p = n0 - a.R - (BMQ - 1);
      const int mt_p_end = lim_p >= 0 ? lim_p / BMQ + 1 : 0;   // tiles [mt_first, mt_p_end) are far-positive
      // far-negative: n0 + BNK - 1 - mrow0 <= -R  <=>  mrow0 >= n0 + BNK - 1 + R
      const int mt_n_beg = (n0 + BNK - 1 + a.R + BMQ - 1) / BMQ;
      if (!a.causal) {  // (with a causal mask positive deltas are masked, never FAST)
        ia = max(0, min(mt_p_end, mt_full) - mt0);
        if (mt_first > mt0) ia = 0;
      }
      ib1 = max(ia, mt_full - mt0);
      ib0 = min(ib1, max(ia, max(mt_n_beg, mt_first) - mt0));
      cst_a = cst_pos;
      cst_b = cst_neg;
    } else {
      ib1 = max(0, mt_full - mt0);
      ib0 = min(ib1, max(0, mt_first - mt0));
    }
    ia &= ~1;
    ib0 = (ib0 + 1) & ~1;
    ib1 &= ~1;
    if (ib1 < ib0) ib0 = ib1 = ia;
  }
  int i = 0;
  for (; i < ia; i += 2) {
    tile.template operator()<true, 0, 1>(mt0 + i, cst_a);
    tile.template operator()<true, 1, 1>(mt0 + i + 1, cst_a);
  }
  const int g0 = min(max(ib0, ia), ntile & ~1);
  for (; i < g0; i += 2) {
    tile.template operator()<false, 0, 0>(mt0 + i, 0.f);
    tile.template operator()<false, 1, 0>(mt0 + i + 1, 0.f);
  }
  for (; i < ib1; i += 2) {
    tile.template operator()<true, 0, -1>(mt0 + i, cst_b);
    tile.template operator()<true, 1, -1>(mt0 + i + 1, cst_b);
  }
  for (; i + 1 < ntile; i += 2) {
    tile.template operator()<false, 0, 0>(mt0 + i, 0.f);
    tile.template operator()<false, 1, 0>(mt0 + i + 1, 0.f);
  }
  if (i < ntile) tile.template operator()<false, 0, 0>(mt0 + i, 0.f);

  // the partial diagonal sums first: their workgroup barrier would otherwise also wait for the dK / dV stores below
  // (stores count on vmcnt), i.e. for a full memory round trip on the critical path of the workgroup
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    if (want_drpe) {
      diag_flush();
      // wave-reduce the far sums (fixed order), fold into the wave's private array
      far_neg = wave_sum(far_neg);
      far_pos = wave_sum(far_pos);
      if (l == 0) {  // lane 0 has hi = 0: its sD is the wave's first array
        sD[0] += far_neg;
        sD[2 * a.R] += far_pos;
      }
      __syncthreads();
      float* out = a.drpe_part + ((int64_t)bh * a.n_nblk + nblk) * n1;
      for (int i2 = tid; i2 < n1; i2 += NT) {
        float acc = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) acc += sD0[ww * n1 + i2];
        out[i2] = acc;
      }
    }
  }
  if (krow < N) {
    const float scale = a.scale;
    uint16_t* dkrow = dkb + (int64_t)krow * a.dks[2];
    uint16_t* dvrow = dvb + (int64_t)krow * a.dvs[2];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 wk, wv;
        wk[0] = pack2<BF16>(dkacc[db][4 * g + 0] * scale, dkacc[db][4 * g + 1] * scale);
        wk[1] = pack2<BF16>(dkacc[db][4 * g + 2] * scale, dkacc[db][4 * g + 3] * scale);
        wv[0] = pack2<BF16>(dvacc[db][4 * g + 0], dvacc[db][4 * g + 1]);
        wv[1] = pack2<BF16>(dvacc[db][4 * g + 2], dvacc[db][4 * g + 3]);
        if (32 * db + 8 * g + 4 * hi < a.dvalid) {
          *reinterpret_cast<u32x2*>(dkrow + 32 * db + 8 * g + 4 * hi) = wk;
          *reinterpret_cast<u32x2*>(dvrow + 32 * db + 8 * g + 4 * hi) = wv;
        }
      }
  }
}

// ---- launchable kernels -------------------------------------------------------------------------------------
#ifndef FAT5_BWDQ_MINW
#define FAT5_BWDQ_MINW 2  // (3 fits at D <= 64 with a few spills but measured no faster: 82.7 vs 83.3 us at S=2048)
#endif
// head_dim 128 (round 6, profiles/r06_d128_dq_minw.log): the dense-bias instantiation needs ~100 bytes of scratch per lane at two waves per SIMD (bias tile reader, dS
// staging) -- one wave per SIMD with the whole register file is faster there ((16,12,1024,128) causal dense: dQ stage 275 -> 222 us, backward 582 -> 526); the T5-table
// and no-bias instantiations fit two waves with 0 / 32 bytes of scratch and lose at one ((4,12,1024,128) T5 table: 55 -> 71 us)
#ifndef FAT5_BWDQ128_MINW
#define FAT5_BWDQ128_MINW(BIAS) ((BIAS) == FAT5_BIAS_DENSE ? 1 : 2)
#endif
template <int D, bool BF16, int BIAS, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(D <= 64 ? FAT5_BWDQ_MINW : FAT5_BWDQ128_MINW(BIAS))))
void attn_bwd_q_kernel(const AttnArgs a) {
  attn_bwd_q_body<D, BF16, BIAS, NW>(a, blockIdx.x);
}
// D = 128: 128 accumulator + 64 K/V fragment registers per lane -- the body needs ~370 VGPRs, so it runs one wave per
// SIMD (512-register budget) instead of spilling 119 registers at two.
template <int D, bool BF16, int BIAS, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(D <= 64 ? FAT5_BWD_MINW : 1)))
void attn_bwd_kv_kernel(const AttnArgs a) {
  attn_bwd_kv_body<D, BF16, BIAS, NW, false>(a, blockIdx.x);
}
// Both backward kernels in ONE launch (horizontal fusion) for problems whose two grids together fit the chip at two
// workgroups per CU: workgroups [0, n_kv_blocks) run the dK/dV body (the longer one first), the rest the dQ body.
// Neither half depends on the other (the dK/dV half forms delta itself), so a short-sequence backward costs
// max(dQ, dK/dV) instead of their sum.
template <int D, bool BF16, int BIAS, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(D <= 64 ? FAT5_BWD_MINW : 1)))
void attn_bwd_fused_kernel(const AttnArgs a) {
  if ((int)blockIdx.x < a.n_kv_blocks) attn_bwd_kv_body<D, BF16, BIAS, NW, true>(a, blockIdx.x);
  else attn_bwd_q_body<D, BF16, BIAS, NW>(a, blockIdx.x - a.n_kv_blocks);
}

}  // namespace fat5
