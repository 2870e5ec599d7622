// Software-pipelined FlashAttention-2 forward for gfx950 (the production forward; attn_fwd.h keeps the simpler,
// phase-separated version for A/B runs).
//
// Why: on gfx950 one SIMD overlaps matrix (MFMA) and vector (VALU) work only when both come from the SAME wave's
// instruction stream with no dependency between them -- two co-resident waves that sit in the same phase simply
// serialise (measured: tools/mb_interleave.hip, PMC SQ_VALU_MFMA_BUSY + SQ_ACTIVE_INST_VALU ~ 100 % of the kernel).
// So every wave software-pipelines its own 32-key blocks:
//
//   step j:   [MFMA] S(j+1) = K(j+1) Q^T          (independent of everything below -> runs under the VALU work)
//             [VALU] P(j)   = exp2(S(j)*c2 + add - m)   (S(j) was produced one step earlier)
//             [MFMA] O     += V(j)^T P(j)^T ,  l += 1 . P(j)^T
//             [VALU] bias / mask / row max of S(j+1)
//
// which needs block j+1's K tile while block j's V tile is still in use: a 3-slot LDS ring (prefetch distance two
// tiles, global -> registers -> LDS, one barrier per tile).  Everything else (layouts, exp2-domain folding, deferred
// rescale, tr-read V fragments, row sums on the matrix pipe, buffer-descriptor prefetch) is as in attn_fwd.h.
#pragma once
#include "attn_fwd.h"

#ifndef FAT5_PABL
#define FAT5_PABL 0  // developer ablation bits (results become garbage): 1 no global prefetch, 2 no barrier, 4 no fragment reads, 8 no exp, 16 no PV, 32 no rescale check, 64 no row max
#endif

namespace fat5 {

template <int D, int NW>
struct FwdPipeCfg {
  static constexpr int BM = 32 * NW;
  static constexpr int BN = 64;
  static constexpr int NT = 64 * NW;
  static constexpr int KBYTES = rm_bytes<D, BN>();
  static constexpr int VBYTES = rm_bytes<D, BN>();
  static constexpr int STAGE = KBYTES + VBYTES;
  static constexpr int NSLOT = 3;
  static size_t smem(int R, int bias_mode) {
    return NSLOT * STAGE + (bias_mode == FAT5_BIAS_RPE1D ? (size_t)(2 * R + 1) * 4 + 16 : 0);
  }
};

template <int D, bool BF16, int BIAS, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(FAT5_FWD_MINW)))
void attn_fwd_pipe_kernel(const AttnArgs a) {
  using Cfg = FwdPipeCfg<D, NW>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT;
  constexpr int KK = D / 16, DB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sT = reinterpret_cast<float*>(smem + Cfg::NSLOT * Cfg::STAGE);

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

  const int qrow0 = m0 + 32 * w;
  const int qrow = qrow0 + lq;
  const int qrow_c = min(qrow, M - 1);

  u32x4 qf[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
    qf[kk] = *reinterpret_cast<const u32x4*>(qb + (int64_t)qrow_c * a.qs[2] + 16 * kk + 8 * hi);

  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    const int n1 = 2 * a.R + 1;
    for (int i = tid; i < n1; i += NT) sT[i] = a.rpe1d[(int64_t)h * n1 + i] * kLog2e;  // log2 units
  }
  const uint16_t* brow = nullptr;
  if constexpr (BIAS == FAT5_BIAS_DENSE)
    brow = a.bias + (int64_t)b * a.bs[0] + (int64_t)h * a.bs[1] + (int64_t)qrow_c * a.bs[2];

  FragAddr<D> fa;
  fa.init(l);

  f32x16 oacc[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  f32x16 lacc;  // every register = running row sum of (rounded) P for this lane's query
#pragma unroll
  for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
  const uint32_t one2 = pack2<BF16>(1.f, 1.f);
  const u32x4 ones = {one2, one2, one2, one2};
  float m_run = -INFINITY;  // running row max, log2 units

  RowStage<D, BN, NT> kst, vst;
  kst.init(a.ks[2], tid);
  vst.init(a.vs[2], tid);
  const __amdgpu_buffer_rsrc_t krs = make_rows_rsrc(kb_, a.ks[2], N, D);
  const __amdgpu_buffer_rsrc_t vrs = make_rows_rsrc(vb, a.vs[2], N, D);
  const uint32_t kstride_b = (uint32_t)a.ks[2] * 2u, vstride_b = (uint32_t)a.vs[2] * 2u;
  // prologue: tiles 0 and 1 into slots 0 and 1
  for (int t = 0; t < ((FAT5_PABL & 1) ? 3 : 2) && t < nt; ++t) {
    kst.load_buf(krs, (uint32_t)(t * BN) * kstride_b, tid);
    vst.load_buf(vrs, (uint32_t)(t * BN) * vstride_b, tid);
    kst.store_rm(smem + t * Cfg::STAGE, tid);
    vst.store_rm(smem + t * Cfg::STAGE + Cfg::KBYTES, tid);
  }
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) asm volatile("" ::"v"(qf[kk]));  // see attn_fwd.h (waitcnt model)

  const float c2 = a.scale * kLog2e;
  const bool fold_ok = c2 > 0.f;
  float cst_neg = 0.f, cst_pos = 0.f;
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    cst_neg = sT[0];
    cst_pos = sT[2 * a.R];
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // ---- pipeline state: scores of the block whose softmax is due next --------------------------------------
  f32x16 s_cur;
  float mul_cur = c2, add_cur = 0.f, mcand_cur = -INFINITY;
  bool need_rescale = true;

  // S^T block = K(block) Q^T from the row-major image `sK`
  auto qk_block = [&](const char* sK, int kb) {
    u32x4 kf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) kf[kk] = ld_rm<D>(sK, fa, kb, kk);
    f32x16 s;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) s = mfma32<BF16>(kf[kk], qf[kk], kk == 0 ? zero16 : s);
    return s;
  };
  // bias / masks / row-max candidate of a freshly computed block (GEN = false: all-visible block, constant bias cst)
  auto post_block = [&]<bool GEN>(f32x16& s, int nb, float cst, float& mul, float& add, float& mcand) {
    if constexpr (!GEN) {
      mul = c2;
      add = cst;
      mcand = (FAT5_PABL & 64) ? cst : fmaf(max16(s), c2, cst);
    } else {
      bool folded = fold_ok;
      float cb = 0.f;
      if constexpr (BIAS == FAT5_BIAS_DENSE) {
        folded = false;
        float bv[16];
        load_bias_block<BF16>(brow, nb, hi, N, a.bias_vec4 && (nb + 32 <= N), bv);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], c2, bv[r] * kLog2e);
      } else if constexpr (BIAS == FAT5_BIAS_RPE1D) {
        const int R = a.R;
        const int dmin = nb - (qrow0 + 31), dmax = nb + 31 - qrow0;  // wave-uniform
        if (dmax <= -R || dmin >= R) {
          cb = (dmax <= -R) ? cst_neg : cst_pos;
          if (!folded) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], c2, cb);
          }
        } else if (dmin > -R && dmax < R) {
          folded = false;
          const float* tp = sT + (R + nb + 4 * hi - qrow);
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], c2, tp[(r & 3) + 8 * (r >> 2)]);
        } else {
          folded = false;
          const int dl = nb + 4 * hi - qrow;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int d = dl + (r & 3) + 8 * (r >> 2);
            s[r] = fmaf(s[r], c2, sT[min(max(d, -R), R) + R]);
          }
        }
      } else {
        if (!folded) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] *= c2;
        }
      }
      const bool nmask = nb + 32 > N;
      const bool cmask = a.causal && (nb + 31 > qrow0 + P);
      if (nmask || cmask) {
        const int lim = (a.causal ? min(N - 1, qrow + P) : N - 1) - nb - 4 * hi;  // last visible crow of this lane
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = ((r & 3) + 8 * (r >> 2) > lim) ? -INFINITY : s[r];
      }
      const float m16 = max16(s);
      mul = folded ? c2 : 1.f;
      add = folded ? cb : 0.f;
      mcand = folded ? fmaf(m16, c2, cb) : m16;
    }
    if (!(FAT5_PABL & 64)) mcand = pair_max(mcand);
  };

  // Operand fragments are double-buffered in registers: the LDS reads for step j+1 (K of block j+2, V of block j+1)
  // are issued at the top of step j, so no MFMA ever waits on an LDS read issued in its own step (a wave stalls as
  // a whole at s_waitcnt -- the independent VALU work behind it would stall too).
  u32x4 kf_next[KK];       // K fragments of block j+1 (consumed by this step's QK^T)
  u32x4 vf_cur[2][DB];     // V^T fragments of block j   (consumed by this step's PV)
  auto load_kf = [&](u32x4 (&kf)[KK], const char* sK, int kb) {
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) kf[kk] = ld_rm<D>(sK, fa, kb, kk);
  };
  auto load_vf = [&](u32x4 (&vf)[2][DB], const char* sV, int kb) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int db = 0; db < DB; ++db) vf[t2][db] = ld_tr<D>(sV, fa, kb, t2, db);
  };

  // One pipeline step: softmax + PV of the current block j while the next block's scores are formed.
  //   sKnn/kbnn: image + block of K(j+2);  sVn/kbn: image + block of V(j+1);  nbn: first key of block j+1.
  auto step = [&]<bool GEN_NEXT>(bool has_next, const char* sKnn, int kbnn, const char* sVn, int kbn, int nbn, float cst_next) {
    // deferred rescale: every P.V of earlier blocks has been issued (program order), so scaling O and l here is exact
    if (__builtin_expect(!(FAT5_PABL & 32) && need_rescale, 0)) {
      const float m_new = fmaxf(m_run, mcand_cur);
      const float alpha = fast_exp2(m_run - ((m_new == -INFINITY) ? 0.f : m_new));
#pragma unroll
      for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
      m_run = m_new;
    }
    const float ad = add_cur - ((m_run == -INFINITY) ? 0.f : m_run);
    const float mulc = mul_cur;
    // next step's operands
    u32x4 kf_nn[KK], vf_n[2][DB];
    if constexpr (FAT5_PABL & 4) {
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) kf_nn[kk] = kf_next[kk];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int db = 0; db < DB; ++db) vf_n[t2][db] = vf_cur[t2][db];
    } else {
      load_kf(kf_nn, sKnn, kbnn);
      load_vf(vf_n, sVn, kbn);
    }
    __builtin_amdgcn_sched_barrier(0);
    // next block's scores: independent MFMA chain that runs under the exponentials below
    f32x16 s_next = zero16;
    if (has_next) {
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) s_next = mfma32<BF16>(kf_next[kk], qf[kk], kk == 0 ? zero16 : s_next);
    }
    // P = exp2(S*mul + ad), packed to the input dtype
    u32x4 pb[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) s_cur[r] = (FAT5_PABL & 8) ? fmaf(s_cur[r], mulc, ad) : fast_exp2(fmaf(s_cur[r], mulc, ad));
    pb[0] = pack8<BF16>(s_cur, 0);
    pb[1] = pack8<BF16>(s_cur, 1);
    // O^T += V^T P^T ; l += 1 . P^T
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      if constexpr (FAT5_PABL & 16) {
        lacc[0] += __builtin_bit_cast(float, pb[t2][0] ^ pb[t2][1] ^ pb[t2][2] ^ pb[t2][3] ^ vf_cur[t2][0][0] ^ vf_cur[t2][DB - 1][1]);
        continue;
      }
#pragma unroll
      for (int db = 0; db < DB; ++db) oacc[db] = mfma32<BF16>(vf_cur[t2][db], pb[t2], oacc[db]);
      lacc = mfma32<BF16>(ones, pb[t2], lacc);
    }
    // bias / mask / row max of the next block
    if (has_next) {
      post_block.template operator()<GEN_NEXT>(s_next, nbn, cst_next, mul_cur, add_cur, mcand_cur);
      s_cur = s_next;
      need_rescale = __any(mcand_cur > m_run + FAT5_DEFER_THR);  // decided one step early: no VALU->branch stall
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) kf_next[kk] = kf_nn[kk];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int db = 0; db < DB; ++db) vf_cur[t2][db] = vf_n[t2][db];
  };

  // ---- tile classes (workgroup-uniform): FAST = every key visible to every row, bias one constant -------------
  int ta = 0, tb0 = 0, tb1 = 0;
  float cst_a = 0.f, cst_b = 0.f;
  if (fold_ok && BIAS != FAT5_BIAS_DENSE) {
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
      tb0 = tb1 = ta;
    }
  }
  auto is_fast = [&](int t) { return t < ta || (t >= tb0 && t < tb1); };

  // first block's scores; operands of the first step
  if (nt > 0) {
    load_kf(kf_next, smem, 1);
    load_vf(vf_cur, smem + Cfg::KBYTES, 0);
    s_cur = qk_block(smem, 0);
    if (is_fast(0)) post_block.template operator()<false>(s_cur, 0, 0 < ta ? cst_a : cst_b, mul_cur, add_cur, mcand_cur);
    else post_block.template operator()<true>(s_cur, 0, 0.f, mul_cur, add_cur, mcand_cur);
  }

  // One tile = two steps.  FF: this tile and the next are FAST with the same constant (the hot body); otherwise the
  // generic post-processing decides per block at run time (it also handles all-visible / constant-bias blocks).
  auto tile = [&]<bool FF>(int t, float cst) {
    const int slot = t % 3;
    const char* sK = smem + slot * Cfg::STAGE;
    const char* sV = sK + Cfg::KBYTES;
    const char* sKn = smem + ((slot + 1) % 3) * Cfg::STAGE;
    const bool pre = (FAT5_PABL & 1) ? false : (t + 2 < nt);
    if (pre) {
      kst.load_buf(krs, (uint32_t)((t + 2) * BN) * kstride_b, tid);
      vst.load_buf(vrs, (uint32_t)((t + 2) * BN) * vstride_b, tid);
    }
    const int n0 = t * BN;
    const char* sVn = sKn + Cfg::KBYTES;
    // step (t,0): block j = (t,0); K(j+2) = (t+1,0) ; V(j+1) = (t,1)      step (t,1): K(j+2) = (t+1,1) ; V(j+1) = (t+1,0)
    if constexpr (FF) {
      step.template operator()<false>(true, sKn, 0, sV, 1, n0 + 32, cst);
      step.template operator()<false>(true, sKn, 1, sVn, 0, n0 + BN, cst);
    } else {
      step.template operator()<true>(true, sKn, 0, sV, 1, n0 + 32, 0.f);
      step.template operator()<true>(t + 1 < nt, sKn, 1, sVn, 0, n0 + BN, 0.f);
    }
    if (pre) {
      char* dst = smem + ((slot + 2) % 3) * Cfg::STAGE;
      kst.store_rm(dst, tid);
      vst.store_rm(dst + Cfg::KBYTES, tid);
    }
    if constexpr (!(FAT5_PABL & 2)) __syncthreads();
  };

  int t = 0;
  for (; t + 1 < ta; ++t) tile.template operator()<true>(t, cst_a);          // FAST followed by FAST (same constant)
  for (; t < tb0 || (t + 1 >= tb1 && t < nt); ++t) {                           // generic stretch (incl. range borders)
    if (t >= tb0 && t + 1 < tb1) break;
    tile.template operator()<false>(t, 0.f);
  }
  for (; t + 1 < tb1; ++t) tile.template operator()<true>(t, cst_b);
  for (; t < nt; ++t) tile.template operator()<false>(t, 0.f);

  // ---- epilogue: o = acc / l, L = m + ln(l) --------------------------------------------------
  const float l_tot = lacc[0];
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
    if (hi == 0) a.lse[lse_off + qrow] = l_tot > 0.f ? (m_run + fast_log2(l_tot)) * kLn2 : -INFINITY;
  }
}

}  // namespace fat5
