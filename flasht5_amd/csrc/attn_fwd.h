// FlashAttention-2 forward with additive bias for gfx950 (CDNA4).
//
// Replaces the reference Triton `_fwd_kernel` (src/model/ops/flash_attention_v2_bias.py:327-483).
//
// Work decomposition: one workgroup = NW waves = one (b, h, 32*NW query rows) tile; a wave owns 32
// query rows for the whole K/V sweep.  Per 64-key K/V tile:
//   S^T[key][q]  = K . Q^T        v_mfma_f32_32x32x16 (A = K fragment, ds_read_b128 from the row-major
//                                  swizzled LDS image; B = Q fragment kept in registers)
//   online softmax in registers   lane (q = lane&31, hi = lane>>5) holds 16 of the 32 keys of each
//                                  32-key block of its row: row max / row sum are in-lane ops plus ONE
//                                  exchange with lane^32.  exp2 domain, sm_scale*log2e and tile-constant
//                                  bias folded into the exponent FMA; the O/l rescale is skipped while no
//                                  row maximum of the wave grows by more than DEFER_THR (exact: the same
//                                  stale maximum is used for P, l and the final normalisation).
//   O^T[d][q]   += V^T . P^T      A = V^T fragment read TRANSPOSED from the row-major V image with
//                                  ds_read_b64_tr_b16, B = P^T straight from the S^T accumulator
//                                  registers (the MFMA k-slot <-> key mapping is chosen so that no
//                                  cross-lane shuffle is needed)
// K/V tiles are prefetched global->registers one tile ahead and written to the other LDS buffer
// after the compute of the current tile (one barrier per tile).
#pragma once
#include "attn_common.h"

namespace fat5 {

#ifndef FAT5_DEFER_THR
#define FAT5_DEFER_THR 6.0f  // log2 units: P <= 2^6 while the running max is stale
#endif

// SPLIT: short sequences (grid smaller than the chip): TWO waves share 32 query rows, wave `half` takes the first / second
// 32-key block of every staged 64-key tile; their (m, l, O) are merged through LDS at the end.  Twice the workgroups,
// half the sequential blocks per wave, same staging traffic per key.
template <int D, int NW, bool SPLIT = false>
struct FwdCfg {
  static constexpr int BM = SPLIT ? 16 * NW : 32 * NW;
  static constexpr int BN = 64;
  static constexpr int NT = 64 * NW;
  static constexpr int KBYTES = rm_bytes<D, BN>();
  static constexpr int VBYTES = rm_bytes<D, BN>();
  static constexpr int STAGE = KBYTES + VBYTES;
  static constexpr int BIASB = BM * BN * 2;  // dense mode: one (BM x 64) 16-bit bias tile per buffer
  static size_t smem(int R, int bias_mode) {
    return 2 * STAGE + (bias_mode == FAT5_BIAS_RPE1D ? rpe_table_bytes(R) + 16 : 0) + 16 +
           (bias_mode == FAT5_BIAS_DENSE ? 2 * (size_t)BIASB : 0);
  }
};

#ifndef FAT5_OPTIMISTIC
#define FAT5_OPTIMISTIC 1  // bf16: all-visible constant-bias tiles skip the running row maximum (see attn_fwd_kernel)
#endif

// dense bias for one 32-key block, C layout of S^T: lane (q, hi) needs keys nb + crow(r, hi)
template <bool BF16>
FAT5_DEV void load_bias_block(const uint16_t* brow, int nb, int hi, int N, bool fast, float (&bv)[16]) {
  if (fast) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u32x2 w = *reinterpret_cast<const u32x2*>(brow + nb + 8 * g + 4 * hi);
      w[0] = bias_clamp2<BF16>(w[0]);
      w[1] = bias_clamp2<BF16>(w[1]);
      bv[4 * g + 0] = cvt_lo<BF16>(w[0]);
      bv[4 * g + 1] = cvt_hi<BF16>(w[0]);
      bv[4 * g + 2] = cvt_lo<BF16>(w[1]);
      bv[4 * g + 3] = cvt_hi<BF16>(w[1]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nb + crow(r, hi);
      bv[r] = (n < N) ? cvt16<BF16>(bias_clamp1<BF16>(brow[n])) : 0.f;
    }
  }
}

FAT5_DEV float max3f(float a, float b, float c) {
  return fmaxf(fmaxf(a, b), c);
}
FAT5_DEV float max16(const f32x16& s) {
  const float a = max3f(s[0], s[1], s[2]), b = max3f(s[3], s[4], s[5]), c = max3f(s[6], s[7], s[8]);
  const float d = max3f(s[9], s[10], s[11]), e = max3f(s[12], s[13], s[14]);
  return max3f(max3f(a, b, c), max3f(d, e, s[15]), s[15]);
}

#ifndef FAT5_FWD_MINW
#define FAT5_FWD_MINW 3  // waves per SIMD the register allocator must leave room for at D <= 64 (D = 128: always 2)
#endif
// BDMA: dense bias tiles provably travel by LDS-DMA (decided at launch) -- a compile-time fact removes the per-block branch between
// the two bias sources, which keeps hipcc from scheduling across it (dense forward -3..5 %)
template <int D, bool BF16, int BIAS, int NW, bool SPLIT = false, bool BDMA = false>
FAT5_DEV void attn_fwd_body(const AttnArgs& a) {
  static_assert(!SPLIT || NW % 2 == 0, "SPLIT pairs the waves of a workgroup");
  using Cfg = FwdCfg<D, NW, SPLIT>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT;
  constexpr int KK = D / 16, DB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* sFlag = reinterpret_cast<int*>(smem + 2 * Cfg::STAGE);  // optimistic pass overflowed -> redo exactly
  float* sT = reinterpret_cast<float*>(smem + 2 * Cfg::STAGE + 16) + kRpePad;  // (entry d of copy 0 at sT[d + R]; see attn_common.h)

  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lq = l & 31, hi = l >> 5;
  constexpr int QG = SPLIT ? NW / 2 : NW;              // query groups (32 rows each) per workgroup
  const int half = SPLIT ? w / QG : 0;                 // SPLIT: which 32-key block of every tile this wave owns
  const int qg = SPLIT ? w - half * QG : w;
  int b, h, mblk;
  decode_unit(a, blockIdx.x, a.n_mblk, b, h, mblk, (FAT5_CAUSAL_ORDER && a.causal) ? 1 : 0);

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

  const int qrow0 = m0 + 32 * qg;       // first query row of this wave
  const int qrow = qrow0 + lq;          // this lane's query row
  const int qrow_c = min(qrow, M - 1);  // clamped for loads

  // Q fragment (B operand of S^T = K Q^T): Q[q][16kk + 8hi + j]
  u32x4 qf[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
    qf[kk] = load_frag16(qb + (int64_t)qrow_c * a.qs[2], kk, hi, a.dvalid);

  const float* sTa = sT;  // this lane's aligned copy of the table
  if constexpr (BIAS == FAT5_BIAS_RPE1D) sTa = sT + ((a.R - qrow) & 3) * rpe_n1p(a.R);
  const uint16_t* brow = nullptr;
  // dense bias: the (BM x 64) tile of this workgroup goes global -> LDS beside K / V in 16-byte pieces (a direct read is
  // 8 bytes per lane from 32 different rows per instruction); fallback for rows that are not 16-byte aligned
  using BDma = DmaStage<BN, BM, NT, true>;
  BDma bdm;
  BiasTileReader brd;
  char* sB = smem + 2 * Cfg::STAGE + 16;  // [2][BM][64] 16-bit (the RPE table region is unused in dense mode)
  const bool bias_dma = BDMA || ((BIAS == FAT5_BIAS_DENSE) && a.bias_dma && a.cu_q == nullptr);
  __amdgpu_buffer_rsrc_t brs = make_rows_rsrc(qb, a.qs[2], 0, D);
  if constexpr (BIAS == FAT5_BIAS_DENSE) {
    brow = a.bias + (int64_t)b * a.bs[0] + (int64_t)h * a.bs[1] + (int64_t)qrow_c * a.bs[2];
    if (bias_dma) {
      bdm.init(a.bs[2], tid);
      brd.init(32 * qg + lq, hi);
      brs = make_rows_rsrc(a.bias + (int64_t)b * a.bs[0] + (int64_t)h * a.bs[1] + (int64_t)m0 * a.bs[2], a.bs[2], M - m0, N);
    }
  }

  FragAddr<D> fa;
  fa.init(l);

  // Optimistic softmax (bf16 only).  FlashAttention's reference point m need not be the running row maximum: ANY m
  // gives the exact result as long as exp2(x - m) neither overflows nor is flushed.  bf16 P and the fp32
  // accumulators share fp32's exponent range, so once a row has a baseline (exact first tile pair) the all-visible
  // constant-bias tiles skip the row maximum, the lane exchange and the rescale test (15 of ~70 VALU instructions per
  // 32-key block): P = exp2(s*c2 + cst - m) with the stale m, l via the matrix pipe, and once per tile pair one compare
  // renormalises O, l by an exact power of two when l >= 2^40.  A score more than ~87 nats above everything seen before
  // would overflow inside one pair: l becomes inf/NaN, the workgroup notices at the end and redoes its tile with the
  // exact algorithm (second pass).  fp16 P would overflow at 2^16, so fp16 always runs the exact pass.
  // (round 5: dense tiles without masked keys run the FAST / optimistic body too, their bias added per element in front of it -- the round-3 experiment
  //  that spilled at the three-wave register cap; the dense instantiations have two waves per SIMD now: their LDS allows no more anyway)
  constexpr bool OPT = FAT5_OPTIMISTIC && BF16;
  if (OPT && tid == 0) *sFlag = 0;

  f32x16 oacc[DB];
  float m_run;  // reference point of the exponentials (running row max, possibly stale), log2 units
  f32x2 l_run;  // per-lane partial row sum (two interleaved chains: v_pk_add_f32)

  DmaStage<D, BN, NT> kst, vst;
  kst.init(a.ks[2], tid, a.dvalid);
  vst.init(a.vs[2], tid, a.dvalid);
  const __amdgpu_buffer_rsrc_t krs = make_rows_rsrc(kb_, a.ks[2], N, a.dvalid);
  const __amdgpu_buffer_rsrc_t vrs = make_rows_rsrc(vb, a.vs[2], N, a.dvalid);
  const uint32_t kstride_b = (uint32_t)a.ks[2] * 2u, vstride_b = (uint32_t)a.vs[2] * 2u;

  // first K/V tile (+ dense bias tile) in flight BEFORE the RPE table is fetched: one memory round trip for the prologue
  auto stage_first = [&]() {
    if (nt > 0) {
      kst.issue(krs, 0, smem, tid);
      vst.issue(vrs, 0, smem + Cfg::KBYTES, tid);
      if constexpr (BIAS == FAT5_BIAS_DENSE)
        if (bias_dma) bdm.issue(brs, 0, sB, tid);
    }
  };
  stage_first();
  if constexpr (BIAS == FAT5_BIAS_RPE1D) rpe_table_fill(sT - kRpePad, a.rpe1d + (int64_t)h * (2 * a.R + 1), a.R, tid, NT);
  __syncthreads();  // tile 0, sT, sFlag visible
  // Touch the Q fragments here: their global loads are otherwise still "pending" in the compiler's waitcnt model at
  // the loop header, and every QK^T MFMA inside the loop then waits on vmcnt, i.e. on the K/V PREFETCH of its own tile.
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) asm volatile("" ::"v"(qf[kk]));
  const float c2 = a.scale * kLog2e;  // scores -> log2 units
  const bool fold_ok = c2 > 0.f;      // max(s*c) = c*max(s) only for c > 0
  float cst_neg = 0.f, cst_pos = 0.f;
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    cst_neg = sT[0];
    cst_pos = sT[2 * a.R];
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // One K/V tile.  FAST: no key of the tile is masked for any row of the workgroup and the bias is one
  // constant `cst` (none, or an all-far RPE tile): raw scores stay in registers, scale and constant are folded
  // into the exponent FMA.  Otherwise the generic body handles masks / per-element bias per 32-key block.
  auto tile = [&]<int MODE, int BUF>(int t, float cst) {  // MODE 0 generic, 1 FAST, 2 FAST optimistic
    constexpr bool FAST = MODE != 0;
    const int n0 = t * BN;
    const char* sK = smem + BUF * Cfg::STAGE;
    const char* sV = sK + Cfg::KBYTES;
    const bool more = (t + 1 < nt);
    // prefetch of the next tile through the buffer descriptors (zero-filled past row N-1 by the hardware)
    if (more) {
      char* nK = smem + (BUF ^ 1) * Cfg::STAGE;  // (its last readers passed the previous tile's barrier)
      kst.issue(krs, (uint32_t)(n0 + BN) * kstride_b, nK, tid);
      vst.issue(vrs, (uint32_t)(n0 + BN) * vstride_b, nK + Cfg::KBYTES, tid);
      if constexpr (BIAS == FAT5_BIAS_DENSE)
        if (bias_dma) bdm.issue(brs, (uint32_t)(n0 + BN) * 2u, sB + (BUF ^ 1) * Cfg::BIASB, tid);
    }
    // ---- per block: online softmax -> O^T += V^T P^T ----
    const char* sKb = sK + (SPLIT ? half * 64 * D : 0);  // (32 rows of 2*D bytes)
    const char* sVb = sV + (SPLIT ? half * 64 * D : 0);
#pragma unroll
    for (int kb = 0; kb < (SPLIT ? 1 : 2); ++kb) {
      const int kbr = SPLIT ? half : kb;  // block index inside the tile
      const int nb = n0 + 32 * kbr;
      // one block at a time: 32 fewer live registers (fits three waves per SIMD); overlap comes from the other waves
      f32x16 s;
      {
        u32x4 kf[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) kf[kk] = ld_rm<D>(sKb, fa, kb, kk);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) s = mfma32<BF16>(kf[kk], qf[kk], kk == 0 ? zero16 : s);
      }

      float mul, add, mcand;
      if constexpr (FAST && BIAS == FAT5_BIAS_DENSE) {
        // dense tile, every key visible: x = s * c2 + bias * log2(e) per element, then the constant-bias body (reference point: running maximum in the
        // exact tiles, the stale one in the optimistic tiles)
        float bv[16];
        brd.template load<BF16>(sB + BUF * Cfg::BIASB, kbr, bv);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], c2, bias_log2(bv[r]));
        mul = 1.f;
        add = 0.f;
        if constexpr (MODE != 2) mcand = max16(s);
      } else if constexpr (MODE == 2) {
        mul = c2;
        add = cst;
      } else if constexpr (FAST) {
        mul = c2;
        add = cst;
        mcand = fmaf(max16(s), c2, cst);
      } else {
        bool folded = fold_ok;
        float cb = 0.f;
        if constexpr (BIAS == FAT5_BIAS_DENSE) {
          folded = false;
          float bv[16];
          if (bias_dma) brd.template load<BF16>(sB + BUF * Cfg::BIASB, kbr, bv);
          else load_bias_block<BF16>(brow, nb, hi, N, a.bias_vec4 && (nb + 32 <= N), bv);
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], c2, bias_log2(bv[r]));
        } else if constexpr (BIAS == FAT5_BIAS_RPE1D) {
          if constexpr (SPLIT) {
            // one straight-line path for far, edge and band blocks: four aligned 16-byte reads of this lane's padded table copy
            // (entries of r = 4g .. 4g+3 are consecutive; the window start is clamped into the copy -- attn_common.h)
            const int R = a.R;
            folded = false;
            const float4* tp4 = reinterpret_cast<const float4*>(sTa + rpe_clamp_asc(R + nb + 4 * hi - qrow - ((R - qrow) & 3), R));
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 bq = tp4[2 * g];
              s[4 * g + 0] = fmaf(s[4 * g + 0], c2, bq.x);
              s[4 * g + 1] = fmaf(s[4 * g + 1], c2, bq.y);
              s[4 * g + 2] = fmaf(s[4 * g + 2], c2, bq.z);
              s[4 * g + 3] = fmaf(s[4 * g + 3], c2, bq.w);
            }
          } else {
            // (the unsplit instantiations sit at the 168-register limit of three waves per SIMD: left to itself hipcc hoists the table reads of
            // the straight-line form far up and spills 55 registers -- they are pinned behind a scheduling barrier here)
            const int R = a.R;
            folded = false;
            __builtin_amdgcn_sched_barrier(0);
            const float4* tp4 = reinterpret_cast<const float4*>(sTa + rpe_clamp_asc(R + nb + 4 * hi - qrow - ((R - qrow) & 3), R));
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 bq = tp4[2 * g];
              s[4 * g + 0] = fmaf(s[4 * g + 0], c2, bq.x);
              s[4 * g + 1] = fmaf(s[4 * g + 1], c2, bq.y);
              s[4 * g + 2] = fmaf(s[4 * g + 2], c2, bq.z);
              s[4 * g + 3] = fmaf(s[4 * g + 3], c2, bq.w);
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
          const int lim = a.causal ? min(N - 1, qrow + P) : N - 1;  // last visible key of this row
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (nb + crow(r, hi) > lim) s[r] = -INFINITY;
        }
        const float m16 = max16(s);
        mul = folded ? c2 : 1.f;
        add = folded ? cb : 0.f;
        mcand = folded ? fmaf(m16, c2, cb) : m16;  // -inf stays -inf (c2 > 0)
      }

      // online softmax (log2 domain), deferred rescale
      if constexpr (MODE != 2) mcand = pair_max(mcand);
      if (MODE != 2 && __any(mcand > m_run + FAT5_DEFER_THR)) {
        const float m_new = fmaxf(m_run, mcand);
        const float alpha = fast_exp2(m_run - ((m_new == -INFINITY) ? 0.f : m_new));
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        m_run = m_new;
      }
      const float ad = add - ((m_run == -INFINITY) ? 0.f : m_run);
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        // packed FMA (v_pk_fma_f32): two exponents per instruction
        f32x2 x = {s[r], s[r + 1]};
        x = x * mul + ad;
        x[0] = fast_exp2(x[0]);
        x[1] = fast_exp2(x[1]);
        s[r] = x[0];
        s[r + 1] = x[1];
        l_run += x;
      }
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const u32x4 pb = pack8<BF16>(s, t2);
#pragma unroll
        for (int db = 0; db < DB; ++db) oacc[db] = mfma32<BF16>(ld_tr<D>(sVb, fa, kb, t2, db), pb, oacc[db]);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the next block's fragment loads from being hoisted over this block
    }

    __syncthreads();  // (with DMA staging: carries the vmcnt(0) that retires this wave's pieces)
  };

  // Tile classes (workgroup-uniform), all boundaries rounded to EVEN tile indices so that every loop below runs
  // (buffer 0, buffer 1) pairs of ONE body: accumulators stay in place (no register shuffles between bodies).
  //   [0, ta)      FAST, constant cst_a  (no bias / all rows far-negative)
  //   [ta, tb0)    generic
  //   [tb0, tb1)   FAST, constant cst_b  (all rows far-positive)
  //   [tb1, nt)    generic (N tail, causal diagonal)
  int ta = 0, tb0 = 0, tb1 = 0;
  float cst_a = 0.f, cst_b = 0.f;
  if (fold_ok && (BIAS != FAT5_BIAS_DENSE || bias_dma)) {
    int t_full = N / BN;                                            // tiles without an N tail
    if (a.causal) t_full = min(t_full, max(0, (m0 + P + 1) / BN));  // n0 + BN - 1 <= m0 + P
    t_full = min(t_full, nt);
    if constexpr (BIAS == FAT5_BIAS_RPE1D) {
      const int lim_a = m0 - a.R - (BN - 1);                        // n0 + BN - 1 - m0 <= -R
      ta = lim_a >= 0 ? min(t_full, lim_a / BN + 1) : 0;
      const int lo = m0 + BM - 1 + a.R;                             // n0 - (m0 + BM - 1) >= R
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
  // once per optimistic tile pair: keep l (and O) below 2^40 by an exact power of two
  auto renorm = [&]() {
    const float lchk = l_run[0] + l_run[1];  // lane partial <= row sum: conservative trigger
    if (__builtin_expect(__any(!(lchk < 0x1p40f)), 0)) {
      const float lc = pair_sum(lchk);  // both lanes of a row must pick the same exponent
      const int e = (lc >= 0x1p40f) ? (int)((__float_as_uint(lc) >> 23) & 0xffu) - 127 : 0;
      const float alpha = __uint_as_float((uint32_t)(127 - min(e, 126)) << 23);  // 2^-e
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
      m_run += (float)e;
    }
  };

  for (int pass = 0;; ++pass) {
    const bool opt = OPT && pass == 0;
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    m_run = -INFINITY;
    l_run = f32x2{0.f, 0.f};
    if (pass > 0) {  // (pass 0: staged before the loop)
      stage_first();
      __syncthreads();
    }

    int t = 0;
    // range A: exact (all of it, or just the baseline pair of an optimistic pass)
    for (const int te = opt ? min(ta, 2) : ta; t < te; t += 2) {
      tile.template operator()<1, 0>(t, cst_a);
      tile.template operator()<1, 1>(t + 1, cst_a);
    }
    if constexpr (OPT) {
      if (opt && t < ta) {
        m_run = (m_run == -INFINITY) ? 0.f : m_run;  // rows without a visible key so far: l = 0, any finite m will do
        for (; t < ta; t += 2) {
          renorm();
          tile.template operator()<2, 0>(t, cst_a);
          tile.template operator()<2, 1>(t + 1, cst_a);
        }
      }
    }
    const int g0 = min(max(tb0, ta), nt & ~1);
    for (; t < g0; t += 2) {
      tile.template operator()<0, 0>(t, 0.f);
      tile.template operator()<0, 1>(t + 1, 0.f);
    }
    // range B
    for (const int te = opt ? (t == 0 ? min(tb1, 2) : t) : tb1; t < te; t += 2) {
      tile.template operator()<1, 0>(t, cst_b);
      tile.template operator()<1, 1>(t + 1, cst_b);
    }
    if constexpr (OPT) {
      if (opt && t < tb1) {
        m_run = (m_run == -INFINITY) ? 0.f : m_run;
        for (; t < tb1; t += 2) {
          renorm();
          tile.template operator()<2, 0>(t, cst_b);
          tile.template operator()<2, 1>(t + 1, cst_b);
        }
      }
    }
    for (; t + 1 < nt; t += 2) {
      tile.template operator()<0, 0>(t, 0.f);
      tile.template operator()<0, 1>(t + 1, 0.f);
    }
    if (t < nt) tile.template operator()<0, 0>(t, 0.f);

    if constexpr (!OPT) {
      break;
    } else {
      if (!opt) break;
      if (!(l_run[0] + l_run[1] < 0x1p120f)) *sFlag = 1;
      __syncthreads();
      if (*sFlag == 0) break;
    }
  }

  if constexpr (SPLIT) {
    // merge the two key halves of every query group: wave (half 1) parks its state in LDS (the tile buffers are dead:
    // every wave is past the last tile's barrier), wave (half 0) folds it in with the usual two-reference-point rule
    float* sx = reinterpret_cast<float*>(smem) + qg * (2 + 16 * DB) * 64 + l;  // [2 + 16*DB][64 lanes] per query group
    if (half == 1) {
      sx[0] = m_run;
      sx[64] = l_run[0] + l_run[1];
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) sx[(2 + 16 * db + r) * 64] = oacc[db][r];
    }
    __syncthreads();
    if (half == 1) return;
    const float m_b = sx[0], l_b = sx[64];
    const float m_n = fmaxf(m_run, m_b);
    const float sa = (m_run == -INFINITY) ? 0.f : fast_exp2(m_run - m_n);
    const float sb = (m_b == -INFINITY) ? 0.f : fast_exp2(m_b - m_n);
    l_run = f32x2{(l_run[0] + l_run[1]) * sa + l_b * sb, 0.f};
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] = oacc[db][r] * sa + sx[(2 + 16 * db + r) * 64] * sb;
    m_run = m_n;
  }
  // ---- epilogue: o = acc / l, L = m + ln(l) --------------------------------------------------
  const float l_tot = pair_sum(l_run[0] + l_run[1]);
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
        if (32 * db + 8 * g + 4 * hi < a.dvalid) *reinterpret_cast<u32x2*>(orow + 32 * db + 8 * g + 4 * hi) = wv;
      }
    if (hi == 0) a.lse[lse_off + qrow] = l_tot > 0.f ? (m_run + fast_log2(l_tot)) * kLn2 : -INFINITY;
  }
}

// (dense bias: the two tile buffers + two bias tiles are 48 .. 96 KB of LDS -- at most two waves per SIMD fit anyway, so the allocator gets their registers)
template <int D, bool BF16, int BIAS, int NW, bool BDMA = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(D <= 64 && BIAS != FAT5_BIAS_DENSE ? FAT5_FWD_MINW : 2)))
void attn_fwd_kernel(const AttnArgs a) {
  attn_fwd_body<D, BF16, BIAS, NW, false, BDMA>(a);
}
template <int D, bool BF16, int BIAS, int NW, bool BDMA = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(D <= 64 && BIAS != FAT5_BIAS_DENSE ? FAT5_FWD_MINW : 2)))
void attn_fwd_split_kernel(const AttnArgs a) {
  attn_fwd_body<D, BF16, BIAS, NW, true, BDMA>(a);
}

}  // namespace fat5
