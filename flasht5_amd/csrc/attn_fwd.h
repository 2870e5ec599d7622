// FlashAttention-2 forward with additive bias for gfx950 (CDNA4).
//
// Replaces the reference Triton `_fwd_kernel` (src/model/ops/flash_attention_v2_bias.py:327-483).
//
// Work decomposition: one workgroup = NW waves = one (b, h, 32*NW query rows) tile; a wave owns 32
// query rows for the whole K/V sweep.  Per 64-key K/V tile:
//   S^T[key][q]  = K . Q^T        v_mfma_f32_32x32x16 (A = K fragment from the row-major swizzled
//                                  LDS image, B = Q fragment kept in registers)
//   online softmax in registers   lane (q = lane&31, hi = lane>>5) holds 16 of the 32 keys of each
//                                  32-key block of its row: row max / row sum are 15 in-lane ops
//                                  plus ONE exchange with lane^32
//   O^T[d][q]   += V^T . P^T      A = V^T fragment from the transposed LDS image, B = P^T straight
//                                  from the S^T accumulator registers (the MFMA k-slot <-> key
//                                  mapping is chosen so that no cross-lane shuffle is needed)
// K/V tiles are prefetched global->registers one tile ahead and written to the other LDS buffer
// after the compute of the current tile (one barrier per tile).
#pragma once
#include "attn_common.h"

namespace fat5 {

template <int D, int NW>
struct FwdCfg {
  static constexpr int BM = 32 * NW;
  static constexpr int BN = 64;
  static constexpr int NT = 64 * NW;
  static constexpr int KBYTES = rm_bytes<D, BN>();
  static constexpr int VBYTES = tr_bytes<D, BN>();
  static constexpr int STAGE = KBYTES + VBYTES;
  static size_t smem(int R, int bias_mode) {
    return 2 * STAGE + (bias_mode == FAT5_BIAS_RPE1D ? (size_t)(2 * R + 1) * 4 + 16 : 0);
  }
};

// dense bias for one 32-key block, C layout of S^T: lane (q, hi) needs keys nb + crow(r, hi)
template <bool BF16>
FAT5_DEV void load_bias_block(const uint16_t* brow, int nb, int hi, int N, bool fast, float (&bv)[16]) {
  if (fast) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x2 w = *reinterpret_cast<const u32x2*>(brow + nb + 8 * g + 4 * hi);
      bv[4 * g + 0] = cvt_lo<BF16>(w[0]);
      bv[4 * g + 1] = cvt_hi<BF16>(w[0]);
      bv[4 * g + 2] = cvt_lo<BF16>(w[1]);
      bv[4 * g + 3] = cvt_hi<BF16>(w[1]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nb + crow(r, hi);
      bv[r] = (n < N) ? cvt16<BF16>(brow[n]) : 0.f;
    }
  }
}

template <int D, bool BF16, int BIAS, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const AttnArgs a) {
  using Cfg = FwdCfg<D, NW>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT;
  constexpr int KK = D / 16, DB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sT = reinterpret_cast<float*>(smem + 2 * Cfg::STAGE);

  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lq = l & 31, hi = l >> 5;
  int bh, mblk;
  decode_block(blockIdx.x, a.B * a.H, a.n_mblk, bh, mblk);
  const int b = bh / a.H, h = bh % a.H;

  int M = a.M, N = a.N;
  int64_t qoff = (int64_t)b * a.qs[0], koff = (int64_t)b * a.ks[0], voff = (int64_t)b * a.vs[0],
          ooff = (int64_t)b * a.os[0];
  int64_t lse_off = ((int64_t)b * a.H + h) * a.M;
  if (a.cu_q) {
    const int q0 = a.cu_q[b], k0 = a.cu_k[b];
    M = a.cu_q[b + 1] - q0;
    N = a.cu_k[b + 1] - k0;
    qoff = (int64_t)q0 * a.qs[2];
    ooff = (int64_t)q0 * a.os[2];
    koff = (int64_t)k0 * a.ks[2];
    voff = (int64_t)k0 * a.vs[2];
    lse_off = (int64_t)h * a.total_q + q0;
  }
  const int m0 = mblk * BM;
  if (m0 >= M) return;
  const uint16_t* qb = a.q + qoff + (int64_t)h * a.qs[1];
  const uint16_t* kb_ = a.k + koff + (int64_t)h * a.ks[1];
  const uint16_t* vb = a.v + voff + (int64_t)h * a.vs[1];
  uint16_t* ob = a.o + ooff + (int64_t)h * a.os[1];

  const int P = N - M;  // bottom-right causal offset
  int n_end = N;
  if (a.causal) n_end = min(N, m0 + BM + P);
  const int nt = n_end > 0 ? (n_end + BN - 1) / BN : 0;

  const int qrow0 = m0 + 32 * w;        // first query row of this wave
  const int qrow = qrow0 + lq;          // this lane's query row
  const int qrow_c = min(qrow, M - 1);  // clamped for loads

  // Q fragment (B operand of S^T = K Q^T): Q[q][16kk + 8hi + j]
  u32x4 qf[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
    qf[kk] = *reinterpret_cast<const u32x4*>(qb + (int64_t)qrow_c * a.qs[2] + 16 * kk + 8 * hi);

  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    const int n1 = 2 * a.R + 1;
    for (int i = tid; i < n1; i += NT) sT[i] = a.rpe1d[(int64_t)h * n1 + i];
  }
  const uint16_t* brow = nullptr;
  if constexpr (BIAS == FAT5_BIAS_DENSE)
    brow = a.bias + (int64_t)b * a.bs[0] + (int64_t)h * a.bs[1] + (int64_t)qrow_c * a.bs[2];

  f32x16 oacc[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  PairStage<D, BN, NT> kst, vst;
  if (nt > 0) {
    kst.load(kb_, a.ks[2], 0, N, tid);
    vst.load(vb, a.vs[2], 0, N, tid);
    kst.store_rm(smem, tid);
    vst.store_tr(smem + Cfg::KBYTES, tid);
  }
  __syncthreads();

  const float scale = a.scale;
  for (int t = 0; t < nt; ++t) {
    const int n0 = t * BN;
    const char* sK = smem + (t & 1) * Cfg::STAGE;
    const char* sV = sK + Cfg::KBYTES;
    const bool more = (t + 1 < nt);
    if (more) {
      kst.load(kb_, a.ks[2], n0 + BN, N, tid);
      vst.load(vb, a.vs[2], n0 + BN, N, tid);
    }

    // ---- S^T = K Q^T for the two 32-key blocks --------------------------------------------
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
        s[kb] = mfma32<BF16>(frag_rm<D>(sK, 32 * kb + lq, kk, hi), qf[kk], s[kb]);
    }

    // ---- y = s*scale + bias (natural-log units), masks -------------------------------------
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int nb = n0 + 32 * kb;
      if constexpr (BIAS == FAT5_BIAS_DENSE) {
        float bv[16];
        load_bias_block<BF16>(brow, nb, hi, N, a.bias_vec4 && (nb + 32 <= N), bv);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = fmaf(s[kb][r], scale, bv[r]);
      } else if constexpr (BIAS == FAT5_BIAS_RPE1D) {
        const int R = a.R;
        const int dmin = nb - (qrow0 + 31), dmax = nb + 31 - qrow0;  // wave-uniform
        if (dmax <= -R || dmin >= R) {
          const float c = (dmax <= -R) ? sT[0] : sT[2 * R];
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kb][r] = fmaf(s[kb][r], scale, c);
        } else {
          const int dl = nb + 4 * hi - qrow;  // delta of r = 0
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int d = dl + (r & 3) + 8 * (r >> 2);
            const int idx = min(max(d, -R), R) + R;
            s[kb][r] = fmaf(s[kb][r], scale, sT[idx]);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] *= scale;
      }
      const bool nmask = nb + 32 > N;
      const bool cmask = a.causal && (nb + 31 > qrow0 + P);
      if (nmask || cmask) {
        const int lim = a.causal ? min(N - 1, qrow + P) : N - 1;  // last visible key of this row
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (nb + crow(r, hi) > lim) s[kb][r] = -INFINITY;
      }
    }

    // ---- online softmax ---------------------------------------------------------------------
    float mx = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    mx = fmaxf(mx, xchg32(mx));
    const float m_new = fmaxf(m_run, mx);
    const float m_sub = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = fast_exp2((m_run - m_sub) * kLog2e);
    const float nm = -m_sub * kLog2e;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(fmaf(s[kb][r], kLog2e, nm));
        s[kb][r] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;

    // ---- O^T += V^T P^T ----------------------------------------------------------------------
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const u32x4 pb = pack8<BF16>(s[kb], t2);
#pragma unroll
        for (int db = 0; db < DB; ++db)
          oacc[db] = mfma32<BF16>(frag_tr<BN>(sV, 32 * db + lq, 32 * kb + 16 * t2 + 4 * hi), pb, oacc[db]);
      }

    if (more) {
      char* nK = smem + ((t + 1) & 1) * Cfg::STAGE;
      kst.store_rm(nK, tid);
      vst.store_tr(nK + Cfg::KBYTES, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: o = acc / l, L = m + ln(l) --------------------------------------------------
  const float l_tot = l_run + xchg32(l_run);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  if (qrow < M) {
    uint16_t* orow = ob + (int64_t)qrow * a.os[2];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 wv;
        wv[0] = pack2<BF16>(oacc[db][4 * g + 0] * inv, oacc[db][4 * g + 1] * inv);
        wv[1] = pack2<BF16>(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv);
        *reinterpret_cast<u32x2*>(orow + 32 * db + 8 * g + 4 * hi) = wv;
      }
    if (hi == 0) a.lse[lse_off + qrow] = l_tot > 0.f ? m_run + fast_log2(l_tot) * kLn2 : -INFINITY;
  }
}

}  // namespace fat5
