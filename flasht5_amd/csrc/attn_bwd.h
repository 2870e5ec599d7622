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
#pragma once
#include "attn_common.h"
#include "attn_fwd.h"  // load_bias_block

namespace fat5 {

// =============================================================================================
// dQ kernel
// =============================================================================================
template <int D, int NW>
struct BwdQCfg {
  static constexpr int BM = 32 * NW;
  static constexpr int BN = 64;
  static constexpr int NT = 64 * NW;
  static constexpr int KRM = rm_bytes<D, BN>();  // K row-major
  static constexpr int VRM = rm_bytes<D, BN>();  // V row-major
  static constexpr int KTR = tr_bytes<D, BN>();  // K transposed
  static constexpr int STAGE = KRM + VRM + KTR;
  static size_t smem(int R, int bias_mode) {
    return 2 * STAGE + (bias_mode == FAT5_BIAS_RPE1D ? (size_t)(2 * R + 1) * 4 + 16 : 0);
  }
};

template <int D, bool BF16, int BIAS, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_q_kernel(const AttnArgs a) {
  using Cfg = BwdQCfg<D, NW>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT;
  constexpr int KK = D / 16, DB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sT = reinterpret_cast<float*>(smem + 2 * Cfg::STAGE);

  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lq = l & 31, hi = l >> 5;
  int bh, mblk;
  decode_block(blockIdx.x, a.B * a.H, a.n_mblk, bh, mblk);
  const int b = bh / a.H, h = bh % a.H;
  const int M = a.M, N = a.N;
  const int m0 = mblk * BM;
  if (m0 >= M) return;
  const uint16_t* qb = a.q + (int64_t)b * a.qs[0] + (int64_t)h * a.qs[1];
  const uint16_t* kb_ = a.k + (int64_t)b * a.ks[0] + (int64_t)h * a.ks[1];
  const uint16_t* vb = a.v + (int64_t)b * a.vs[0] + (int64_t)h * a.vs[1];
  const uint16_t* ob = a.o + (int64_t)b * a.os[0] + (int64_t)h * a.os[1];
  const uint16_t* dob = a.dout + (int64_t)b * a.dos[0] + (int64_t)h * a.dos[1];
  uint16_t* dqb = a.dq + (int64_t)b * a.dqs[0] + (int64_t)h * a.dqs[1];
  const int64_t stat_off = ((int64_t)b * a.H + h) * M;

  const int P = N - M;
  int n_end = N;
  if (a.causal) n_end = min(N, m0 + BM + P);
  const int nt = n_end > 0 ? (n_end + BN - 1) / BN : 0;

  const int qrow0 = m0 + 32 * w;
  const int qrow = qrow0 + lq;
  const int qrow_c = min(qrow, M - 1);

  // Q and dO fragments (B operands), delta = rowsum(o * do)
  u32x4 qf[KK], dof[KK];
  float dsum = 0.f;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    qf[kk] = *reinterpret_cast<const u32x4*>(qb + (int64_t)qrow_c * a.qs[2] + 16 * kk + 8 * hi);
    dof[kk] = *reinterpret_cast<const u32x4*>(dob + (int64_t)qrow_c * a.dos[2] + 16 * kk + 8 * hi);
    const u32x4 of = *reinterpret_cast<const u32x4*>(ob + (int64_t)qrow_c * a.os[2] + 16 * kk + 8 * hi);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dsum = fmaf(cvt_lo<BF16>(of[j]), cvt_lo<BF16>(dof[kk][j]), dsum);
      dsum = fmaf(cvt_hi<BF16>(of[j]), cvt_hi<BF16>(dof[kk][j]), dsum);
    }
  }
  const float delta = dsum + xchg32(dsum);
  if (qrow < M && hi == 0) a.delta[stat_off + qrow] = delta;
  const float Lq = a.lse[stat_off + qrow_c];
  // p = exp2(y*log2e - L*log2e); rows with L = -inf (fully masked) contribute nothing
  const float nL = (Lq == -INFINITY) ? -INFINITY : -Lq * kLog2e;

  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    const int n1 = 2 * a.R + 1;
    for (int i = tid; i < n1; i += NT) sT[i] = a.rpe1d[(int64_t)h * n1 + i];
  }
  const uint16_t* brow = nullptr;
  if constexpr (BIAS == FAT5_BIAS_DENSE)
    brow = a.bias + (int64_t)b * a.bs[0] + (int64_t)h * a.bs[1] + (int64_t)qrow_c * a.bs[2];

  f32x16 dqacc[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;

  PairStage<D, BN, NT> kst, vst;
  if (nt > 0) {
    kst.load(kb_, a.ks[2], 0, N, tid);
    vst.load(vb, a.vs[2], 0, N, tid);
    kst.store_rm(smem, tid);
    vst.store_rm(smem + Cfg::KRM, tid);
    kst.store_tr(smem + Cfg::KRM + Cfg::VRM, tid);
  }
  __syncthreads();

  const float scale = a.scale;
  for (int t = 0; t < nt; ++t) {
    const int n0 = t * BN;
    const char* sK = smem + (t & 1) * Cfg::STAGE;
    const char* sV = sK + Cfg::KRM;
    const char* sKt = sV + Cfg::VRM;
    const bool more = (t + 1 < nt);
    if (more) {
      kst.load(kb_, a.ks[2], n0 + BN, N, tid);
      vst.load(vb, a.vs[2], n0 + BN, N, tid);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int nb = n0 + 32 * kb;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        s = mfma32<BF16>(frag_rm<D>(sK, 32 * kb + lq, kk, hi), qf[kk], s);
        dp = mfma32<BF16>(frag_rm<D>(sV, 32 * kb + lq, kk, hi), dof[kk], dp);
      }
      if constexpr (BIAS == FAT5_BIAS_DENSE) {
        float bv[16];
        load_bias_block<BF16>(brow, nb, hi, N, a.bias_vec4 && (nb + 32 <= N), bv);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], scale, bv[r]);
      } else if constexpr (BIAS == FAT5_BIAS_RPE1D) {
        const int R = a.R;
        const int dmin = nb - (qrow0 + 31), dmax = nb + 31 - qrow0;
        if (dmax <= -R || dmin >= R) {
          const float c = (dmax <= -R) ? sT[0] : sT[2 * R];
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], scale, c);
        } else {
          const int dl = nb + 4 * hi - qrow;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int d = dl + (r & 3) + 8 * (r >> 2);
            s[r] = fmaf(s[r], scale, sT[min(max(d, -R), R) + R]);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] *= scale;
      }
      // p = exp(y - L), ds = p * (dp - delta)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(fmaf(s[r], kLog2e, nL));
        s[r] = p * (dp[r] - delta);
      }
      const bool nmask = nb + 32 > N;
      const bool cmask = a.causal && (nb + 31 > qrow0 + P);
      if (nmask || cmask) {
        const int lim = a.causal ? min(N - 1, qrow + P) : N - 1;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (nb + crow(r, hi) > lim) s[r] = 0.f;
      }
      // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const u32x4 dsb = pack8<BF16>(s, t2);
#pragma unroll
        for (int db = 0; db < DB; ++db)
          dqacc[db] = mfma32<BF16>(frag_tr<BN>(sKt, 32 * db + lq, 32 * kb + 16 * t2 + 4 * hi), dsb, dqacc[db]);
      }
    }
    if (more) {
      char* nK = smem + ((t + 1) & 1) * Cfg::STAGE;
      kst.store_rm(nK, tid);
      vst.store_rm(nK + Cfg::KRM, tid);
      kst.store_tr(nK + Cfg::KRM + Cfg::VRM, tid);
    }
    __syncthreads();
  }

  if (qrow < M) {
    uint16_t* drow = dqb + (int64_t)qrow * a.dqs[2];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 wv;
        wv[0] = pack2<BF16>(dqacc[db][4 * g + 0] * scale, dqacc[db][4 * g + 1] * scale);
        wv[1] = pack2<BF16>(dqacc[db][4 * g + 2] * scale, dqacc[db][4 * g + 3] * scale);
        *reinterpret_cast<u32x2*>(drow + 32 * db + 8 * g + 4 * hi) = wv;
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
  static constexpr int QTR = tr_bytes<D, BMQ>();
  static constexpr int STAT = BMQ * 4 * 2;  // -L*log2e and delta for the BMQ rows
  static constexpr int STAGE = 2 * QRM + 2 * QTR + STAT;
  static size_t smem(int R, int bias_mode) {
    // rpe: table + one private accumulator per wave
    return 2 * STAGE + (bias_mode == FAT5_BIAS_RPE1D ? (size_t)(2 * R + 1) * 4 * (1 + NW) + 64 : 0);
  }
};

template <int D, bool BF16, int BIAS, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_kv_kernel(const AttnArgs a) {
  using Cfg = BwdKVCfg<D, NW>;
  constexpr int BNK = Cfg::BNK, BMQ = Cfg::BMQ, NT = Cfg::NT;
  constexpr int KK = D / 16, DB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lq = l & 31, hi = l >> 5;
  int bh, nblk;
  decode_block(blockIdx.x, a.B * a.H, a.n_nblk, bh, nblk);
  const int b = bh / a.H, h = bh % a.H;
  const int M = a.M, N = a.N;
  const int n0 = nblk * BNK;
  if (n0 >= N) return;
  const uint16_t* qb = a.q + (int64_t)b * a.qs[0] + (int64_t)h * a.qs[1];
  const uint16_t* kb_ = a.k + (int64_t)b * a.ks[0] + (int64_t)h * a.ks[1];
  const uint16_t* vb = a.v + (int64_t)b * a.vs[0] + (int64_t)h * a.vs[1];
  const uint16_t* dob = a.dout + (int64_t)b * a.dos[0] + (int64_t)h * a.dos[1];
  uint16_t* dkb = a.dk + (int64_t)b * a.dks[0] + (int64_t)h * a.dks[1];
  uint16_t* dvb = a.dv + (int64_t)b * a.dvs[0] + (int64_t)h * a.dvs[1];
  const int64_t stat_off = ((int64_t)b * a.H + h) * M;

  const int P = N - M;
  const int krow0 = n0 + 32 * w;
  const int krow = krow0 + lq;
  const int krow_c = min(krow, N - 1);

  // K and V fragments (B operands) for this lane's key
  u32x4 kf[KK], vf[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    kf[kk] = *reinterpret_cast<const u32x4*>(kb_ + (int64_t)krow_c * a.ks[2] + 16 * kk + 8 * hi);
    vf[kk] = *reinterpret_cast<const u32x4*>(vb + (int64_t)krow_c * a.vs[2] + 16 * kk + 8 * hi);
  }

  // RPE: table + per-wave private diagonal accumulators in LDS
  float* sT = reinterpret_cast<float*>(smem + 2 * Cfg::STAGE);
  const int n1 = 2 * a.R + 1;
  float* sD = sT + n1 + w * n1;
  float far_neg = 0.f, far_pos = 0.f;
  const bool want_drpe = (BIAS == FAT5_BIAS_RPE1D) && (a.drpe_part != nullptr);
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    for (int i = tid; i < n1; i += NT) sT[i] = a.rpe1d[(int64_t)h * n1 + i];
    for (int i = tid; i < n1 * NW; i += NT) sT[n1 + i] = 0.f;
  }
  const uint16_t* bbase = nullptr;
  uint16_t* dsbase = nullptr;
  if constexpr (BIAS == FAT5_BIAS_DENSE) {
    bbase = a.bias + (int64_t)b * a.bs[0] + (int64_t)h * a.bs[1];
    if (a.ds_out) dsbase = a.ds_out + (int64_t)b * a.dss[0] + (int64_t)h * a.dss[1];
  }

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

  PairStage<D, BMQ, NT> qst, dost;
  auto stage_stats = [&](char* st, int mrow0) {
    float* sL = reinterpret_cast<float*>(st + 2 * Cfg::QRM + 2 * Cfg::QTR);
    for (int i = tid; i < BMQ; i += NT) {
      const int m = mrow0 + i;
      float nl = -INFINITY, dl = 0.f;
      if (m < M) {
        const float L = a.lse[stat_off + m];
        nl = (L == -INFINITY) ? -INFINITY : -L * kLog2e;
        dl = a.delta[stat_off + m];
      }
      sL[i] = nl;
      sL[BMQ + i] = dl;
    }
  };
  if (mt0 < mt1) {
    qst.load(qb, a.qs[2], mt0 * BMQ, M, tid);
    dost.load(dob, a.dos[2], mt0 * BMQ, M, tid);
    qst.store_rm(smem, tid);
    dost.store_rm(smem + Cfg::QRM, tid);
    qst.store_tr(smem + 2 * Cfg::QRM, tid);
    dost.store_tr(smem + 2 * Cfg::QRM + Cfg::QTR, tid);
    stage_stats(smem, mt0 * BMQ);
  }
  __syncthreads();

  const float scale = a.scale;
  for (int mt = mt0; mt < mt1; ++mt) {
    const int mrow0 = mt * BMQ;
    const int buf = (mt - mt0) & 1;
    const char* sQ = smem + buf * Cfg::STAGE;
    const char* sDO = sQ + Cfg::QRM;
    const char* sQt = sDO + Cfg::QRM;
    const char* sDOt = sQt + Cfg::QTR;
    const float* sL = reinterpret_cast<const float*>(sDOt + Cfg::QTR);
    const bool more = (mt + 1 < mt1);
    if (more) {
      qst.load(qb, a.qs[2], mrow0 + BMQ, M, tid);
      dost.load(dob, a.dos[2], mrow0 + BMQ, M, tid);
    }
#pragma unroll
    for (int qbk = 0; qbk < 2; ++qbk) {
      const int mb = mrow0 + 32 * qbk;  // first query row of this 32-row block
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        s = mfma32<BF16>(frag_rm<D>(sQ, 32 * qbk + lq, kk, hi), kf[kk], s);
        dp = mfma32<BF16>(frag_rm<D>(sDO, 32 * qbk + lq, kk, hi), vf[kk], dp);
      }
      // C layout: lane (key = krow, hi), register r <-> query row mb + crow(r, hi)
      if constexpr (BIAS == FAT5_BIAS_DENSE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + crow(r, hi);
          const float bvl = (m < M && krow < N) ? cvt16<BF16>(bbase[(int64_t)m * a.bs[2] + krow]) : 0.f;
          s[r] = fmaf(s[r], scale, bvl);
        }
      } else if constexpr (BIAS == FAT5_BIAS_RPE1D) {
        const int R = a.R;
        const int dmin = krow0 - (mb + 31), dmax = krow0 + 31 - mb;
        if (dmax <= -R || dmin >= R) {
          const float c = (dmax <= -R) ? sT[0] : sT[2 * R];
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], scale, c);
        } else {
          const int dl = krow - mb - 4 * hi;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int d = dl - ((r & 3) + 8 * (r >> 2));
            s[r] = fmaf(s[r], scale, sT[min(max(d, -R), R) + R]);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] *= scale;
      }
      // per-row statistics for rows mb + 8g + 4hi + (0..3)
      f32x16 p;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 nl4 = *reinterpret_cast<const float4*>(sL + 32 * qbk + 8 * g + 4 * hi);
        const float4 dl4 = *reinterpret_cast<const float4*>(sL + BMQ + 32 * qbk + 8 * g + 4 * hi);
        const float nl[4] = {nl4.x, nl4.y, nl4.z, nl4.w};
        const float dl[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          const float pv = fast_exp2(fmaf(s[r], kLog2e, nl[j]));
          p[r] = pv;
          s[r] = pv * (dp[r] - dl[j]);
        }
      }
      const bool nmask = krow0 + 32 > N;
      const bool cmask = a.causal && (krow0 + 31 > mb + P);
      if (nmask || cmask) {
        // key visible to query m iff krow <= m + P (and krow < N)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + crow(r, hi);
          const bool ok = (krow < N) && (!a.causal || krow <= m + P);
          if (!ok) { p[r] = 0.f; s[r] = 0.f; }
        }
      }
      // ---- bias gradient ------------------------------------------------------------------
      if constexpr (BIAS == FAT5_BIAS_DENSE) {
        if (dsbase) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mb + crow(r, hi);
            if (m < M && krow < N) dsbase[(int64_t)m * a.dss[2] + krow] = to16<BF16>(s[r]);
          }
        }
      } else if constexpr (BIAS == FAT5_BIAS_RPE1D) {
        if (want_drpe) {
          const int R = a.R;
          const int dmin = krow0 - (mb + 31), dmax = krow0 + 31 - mb;
          if (dmax <= -R || dmin >= R) {
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc += s[r];
            if (dmax <= -R) far_neg += acc; else far_pos += acc;
          } else {
            const int dl = krow - mb - 4 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int d = dl - ((r & 3) + 8 * (r >> 2));
              atomicAdd(&sD[min(max(d, -R), R) + R], s[r]);
            }
          }
        }
      }
      // ---- dV^T += dO^T P ;  dK^T += Q^T dS -------------------------------------------------
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const u32x4 pb = pack8<BF16>(p, t2);
        const u32x4 dsb = pack8<BF16>(s, t2);
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          dvacc[db] = mfma32<BF16>(frag_tr<BMQ>(sDOt, 32 * db + lq, 32 * qbk + 16 * t2 + 4 * hi), pb, dvacc[db]);
          dkacc[db] = mfma32<BF16>(frag_tr<BMQ>(sQt, 32 * db + lq, 32 * qbk + 16 * t2 + 4 * hi), dsb, dkacc[db]);
        }
      }
    }
    if (more) {
      char* nb_ = smem + (buf ^ 1) * Cfg::STAGE;
      qst.store_rm(nb_, tid);
      dost.store_rm(nb_ + Cfg::QRM, tid);
      qst.store_tr(nb_ + 2 * Cfg::QRM, tid);
      dost.store_tr(nb_ + 2 * Cfg::QRM + Cfg::QTR, tid);
      stage_stats(nb_, mrow0 + BMQ);
    }
    __syncthreads();
  }

  if (krow < N) {
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
        *reinterpret_cast<u32x2*>(dkrow + 32 * db + 8 * g + 4 * hi) = wk;
        *reinterpret_cast<u32x2*>(dvrow + 32 * db + 8 * g + 4 * hi) = wv;
      }
  }

  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    if (want_drpe) {
      // wave-reduce the far sums (fixed butterfly order), fold into the wave's private array
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        far_neg += __shfl_xor(far_neg, off, 64);
        far_pos += __shfl_xor(far_pos, off, 64);
      }
      if (l == 0) {
        sD[0] += far_neg;
        sD[2 * a.R] += far_pos;
      }
      __syncthreads();
      float* out = a.drpe_part + ((int64_t)bh * a.n_nblk + nblk) * n1;
      for (int i = tid; i < n1; i += NT) {
        float acc = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) acc += sT[n1 + ww * n1 + i];
        out[i] = acc;
      }
    }
  }
}

}  // namespace fat5
