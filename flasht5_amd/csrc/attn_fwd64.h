// FlashAttention-2 forward, long-sequence body for gfx950: 64 query rows per wave, software-pipelined tile loop.
//
// Same contract as attn_fwd.h (replaces the reference Triton `_fwd_kernel`, src/model/ops/flash_attention_v2_bias.py:327-483)
// for bias = none / in-kernel T5 RPE.  What is different, and why:
//
//  * A wave owns TWO 32-row query blocks (A, B).  Every K fragment read from LDS feeds two MFMAs and -- the point --
//    consecutive MFMAs always target DIFFERENT accumulators.  On gfx950 an instruction issued between two MFMAs that
//    chain on the same accumulator costs ~43 cycles (MI355X_MICROARCH.md, per-instruction constants): the 32-row body
//    (attn_fwd.h) has 4-long same-accumulator chains, so its softmax VALU work cannot be placed under its MFMAs at all
//    (measured there: time = 32 * #MFMA + sum of VALU issue).  Here VALU / LDS work sits in the gaps between MFMAs.
//  * Three-stage software pipeline over 32-key blocks i: while the VALU pipe turns the scores of block i into
//    probabilities (one FMA + v_exp_f32 + row-sum add per element, packed to 16 bit), the matrix pipe runs
//    O^T += V^T P^T of block i-1 and S^T = K Q^T of block i+1 -- 16 MFMAs that do not depend on this block's VALU work.
//  * K and V live in separate 3-slot LDS rings filled global -> LDS by DMA two tiles (K) / one tile (V) ahead;
//    one barrier per 64-key tile.
//  * The pipelined loop covers the all-visible constant-bias tiles with the optimistic softmax of attn_fwd.h (bf16; exact
//    second pass if a row sum overflows).  Band / masked / baseline tiles run an exact, unpipelined body on both blocks.
#pragma once
#include "attn_common.h"
#include <utility>
#include "attn_fwd.h"

namespace fat5 {

template <int N, typename F>
FAT5_DEV void static_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}
template <int NMF, int DB>
constexpr bool v0_ok() { return NMF / 2 - 2 >= 0 && NMF / 2 - 2 + 2 * DB <= NMF; }

template <int D>
struct Fwd64Cfg {
  static constexpr int NW = 4, BM = 64 * NW, BN = 64, NT = 64 * NW, NS = 4;  // NS ring slots per operand
  static constexpr int TILE = rm_bytes<D, BN>();
  static constexpr int KOFF = 0, VOFF = NS * TILE, FLAG = 2 * NS * TILE, TAB = FLAG + 16;
  static size_t smem(int R, int bias_mode) { return TAB + (bias_mode == FAT5_BIAS_RPE1D ? rpe_table_bytes(R) + 16 : 0); }
};

#ifndef FAT5_F64_PIN
#define FAT5_F64_PIN 1  // pin the hand-placed MFMA / VALU / LDS interleave of the pipelined block: a sched_barrier after every MFMA gap
#endif
#ifndef FAT5_F64_ORDER
#define FAT5_F64_ORDER 0  // 0: the 8 P.V MFMAs, then the 8 Q.K MFMAs of a block; 1: alternating
#endif

// LDS access by integer address (base VGPR + compile-time constant -> the constant lands in the instruction's offset field;
// pointer arithmetic on the dynamic-LDS symbol costs a v_add per access instead)
FAT5_DEV u32x4 lds_rd128(uint32_t addr) {
  typedef const u32x4 __attribute__((address_space(3))) * p_t;
  return *(p_t)(uintptr_t)addr;
}
FAT5_DEV u32x4 lds_rd_tr(uint32_t a0, uint32_t a1) {  // two ds_read_b64_tr_b16 -> one transposed operand fragment
  typedef s16x4_t __attribute__((address_space(3))) * p_t;
  const u32x2 x = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((p_t)(uintptr_t)a0));
  const u32x2 y = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((p_t)(uintptr_t)a1));
  return u32x4{x[0], x[1], y[0], y[1]};
}
// One LDS-DMA piece (64 lanes x 16 bytes -> 1 KiB at the wave-uniform LDS address `lds_dst`) issued from inline asm: hipcc
// does not count it, so it neither waits for it in front of unrelated ds_reads (the builtin form draws a conservative
// vmcnt(0) in front of every LDS read that might alias) nor at barriers -- the kernel places its own s_waitcnt vmcnt.
FAT5_DEV void dma16_asm(__amdgpu_buffer_rsrc_t rsrc, uint32_t lds_dst, uint32_t voff, uint32_t soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}
// single VALU instructions as volatile asm (see pipe_block)
FAT5_DEV float asm_fma(float a, float b, float c) {
  float r;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
FAT5_DEV float asm_exp2(float x) {
  float r;
  asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
FAT5_DEV void asm_add(float& l, float p) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(l) : "v"(p)); }
template <bool BF16>
FAT5_DEV uint32_t asm_cvt_pk(float a, float b) {
  uint32_t r;
  if constexpr (BF16) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));  // (gfx950: one instruction, round-to-nearest-even like the cast)
  return r;
}
#ifndef FAT5_F64_LMFMA
#define FAT5_F64_LMFMA 1  // row sums of the pipelined blocks on the matrix pipe (0: two v_add_f32 per element pair)
#endif
// acc += A(16x32) . B(32x16), in place (asm: a builtin may pick a fresh destination, and no hazard padding exists between asm ops)
template <bool BF16>
FAT5_DEV void mfma16_acc(f32x4& acc, const u32x4 A, const u32x4 B) {
  if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B));
}
FAT5_DEV void wait_dma_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#ifndef FAT5_F64_MINW
#define FAT5_F64_MINW 2  // waves per SIMD the register allocator leaves room for
#endif

template <int D, bool BF16, int BIAS>
FAT5_DEV void attn_fwd64_body(const AttnArgs& a, const int item) {
  static_assert(BIAS != FAT5_BIAS_DENSE, "dense bias runs the 32-row body");
  using Cfg = Fwd64Cfg<D>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT, TILE = Cfg::TILE, NS = Cfg::NS;
  constexpr int KK = D / 16, DB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* sFlag = reinterpret_cast<int*>(smem + Cfg::FLAG);
  float* sT = reinterpret_cast<float*>(smem + Cfg::TAB) + kRpePad;  // (entry d of copy 0 at sT[d + R]; see attn_common.h)

  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lq = l & 31, hi = l >> 5;
  int b, h, mblk;
  decode_unit(a, item, a.n_mblk, b, h, mblk);
  const int M = a.M, N = a.N;
  const int m0 = mblk * BM;
  if (m0 >= M) return;
  const uint16_t* qb_ = a.q + (int64_t)b * a.qs[0] + (int64_t)h * a.qs[1];
  const uint16_t* kb_ = a.k + (int64_t)b * a.ks[0] + (int64_t)h * a.ks[1];
  const uint16_t* vb_ = a.v + (int64_t)b * a.vs[0] + (int64_t)h * a.vs[1];
  uint16_t* ob_ = a.o + (int64_t)b * a.os[0] + (int64_t)h * a.os[1];
  const int64_t lse_off = ((int64_t)b * a.H + h) * a.M;

  const int P = N - M;  // bottom-right causal offset
  int n_end = N;
  if (a.causal) n_end = min(N, m0 + BM + P);
  const int nt = n_end > 0 ? (n_end + BN - 1) / BN : 0;

  const int qrow0 = m0 + 64 * w;  // first query row of this wave; block qb covers rows qrow0 + 32*qb ..+31
  // Q fragments (B operand of S^T = K Q^T): Q[q][16kk + 8hi + j]
  u32x4 qf[2][KK];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qr = min(qrow0 + 32 * qb + lq, M - 1);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
      qf[qb][kk] = *reinterpret_cast<const u32x4*>(qb_ + (int64_t)qr * a.qs[2] + 16 * kk + 8 * hi);
  }
  const float* sTa = sT;  // this lane's aligned copy of the RPE table (rows 32 apart share the alignment)
  if constexpr (BIAS == FAT5_BIAS_RPE1D) sTa = sT + ((a.R - (qrow0 + lq)) & 3) * rpe_n1p(a.R);

  FragAddr<D> fa;
  fa.init(l);

  constexpr bool OPT = FAT5_OPTIMISTIC && BF16;
  if (OPT && tid == 0) *sFlag = 0;

  f32x16 oacc[2][DB];
  float m_run[2];
  float l_run[2][2];
#if FAT5_F64_LMFMA
  // Row sums of the pipelined blocks: lacc[qb] += SEL . P^T words (the B operands of the P.V products, i.e. the ROUNDED
  // probabilities) with one 16x16x32 MFMA per four packed words instead of 16 v_add_f32. As a 32x16 B operand the words of lanes
  // n, n+16, n+32, n+48 form column n; SEL row 4G has ones for the k-groups of parity G%2, every other row is zero, so register 0 of
  // lane l ends up with the sum over both key halves (lanes l%32 and l%32 + 32) of ITS query column; registers 1..3 stay zero.
  f32x4 lacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  u32x4 sel;
  {
    const int li = (int)(threadIdx.x & 63), i16 = li & 15;
    const uint32_t one2 = pack2<BF16>(1.f, 1.f);
    const uint32_t wv = ((i16 & 3) == 0 && (((li >> 4) ^ (i16 >> 2)) & 1) == 0) ? one2 : 0u;
    sel = u32x4{wv, wv, wv, wv};
    asm volatile("" : "+v"(sel));  // (resident: never re-materialised by VALU moves right in front of an asm consumer)
  }
#endif

  DmaStage<D, BN, NT> kst, vst;
  kst.init(a.ks[2], tid);
  vst.init(a.vs[2], tid);
  const __amdgpu_buffer_rsrc_t krs = make_rows_rsrc(kb_, a.ks[2], N, D);
  const __amdgpu_buffer_rsrc_t vrs = make_rows_rsrc(vb_, a.vs[2], N, D);
  const uint32_t kstride_b = (uint32_t)a.ks[2] * 2u, vstride_b = (uint32_t)a.vs[2] * 2u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(tid >> 6) * 1024u);
  using Dma = DmaStage<D, BN, NT>;
  static_assert(Dma::NV == 1, "one voffset per lane");
  auto dma_tile = [&](const Dma& st, __amdgpu_buffer_rsrc_t rs, uint32_t tile_off, uint32_t lds_tile) {
#pragma unroll
    for (int i = 0; i < Dma::PER; ++i) dma16_asm(rs, wave_lds + lds_tile + (uint32_t)(NT * 16 * i), st.voff[0], tile_off + st.piece_step * i);
  };
  auto dma_k = [&](int t, int slot) { dma_tile(kst, krs, (uint32_t)(t * BN) * kstride_b, (uint32_t)(Cfg::KOFF + slot * TILE)); };
  auto dma_v = [&](int t, int slot) { dma_tile(vst, vrs, (uint32_t)(t * BN) * vstride_b, (uint32_t)(Cfg::VOFF + slot * TILE)); };
  // Ring protocol.  Tile t lives in slot t % NS of both rings.  Iteration t (between two barriers) may read K(t), K(t+1) and
  // V(t); it starts by issuing K(t+NS-1) and V(t+NS-2) into the slots of K(t-1) / V(t-2), whose last readers are behind the
  // barrier that ended iteration t-1.  Before its closing barrier every wave waits for its own pieces of K(t+2) and V(t+1)
  // -- everything but the requests issued in THIS iteration (counted vmcnt: LDS-DMA requests retire in order), so a
  // request has a full iteration more than it needs to land.
  auto stage_first = [&]() {
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
      if (i < nt) dma_k(i, i);
#pragma unroll
    for (int i = 0; i < NS - 2; ++i)
      if (i < nt) dma_v(i, i);
  };
  auto begin_iter = [&](int t, int slot) {
    const int sk = slot == 0 ? NS - 1 : slot - 1;                    // (t + NS - 1) % NS
    const int sv = slot <= 1 ? slot + NS - 2 : slot - 2;             // (t + NS - 2) % NS
    if (t + NS - 1 < nt) dma_k(t + NS - 1, sk);
    if (t + NS - 2 < nt) dma_v(t + NS - 2, sv);
  };
  auto end_iter = [&](int t) {
    constexpr int PER = Dma::PER;
    if (t + NS - 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else if (t + NS - 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else wait_dma_all();
    __syncthreads();
  };
  stage_first();
  if constexpr (BIAS == FAT5_BIAS_RPE1D) rpe_table_fill(sT - kRpePad, a.rpe1d + (int64_t)h * (2 * a.R + 1), a.R, tid, NT);
  wait_dma_all();
  __syncthreads();
  // per-lane LDS addresses of the fragment reads (ring base folded in; slot / block / step offsets are immediates)
  uint32_t rmA[KK], trA[2][DB];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    rmA[kk] = lds0 + (uint32_t)(Cfg::KOFF + fa.rm[kk]);
    asm volatile("" : "+v"(rmA[kk]));  // opaque: one register per address, constants go to the offset fields
  }
#pragma unroll
  for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      trA[j2][db] = lds0 + (uint32_t)(Cfg::VOFF + fa.tr[j2][db]);
      asm volatile("" : "+v"(trA[j2][db]));
    }
  auto rd_k = [&](uint32_t off, int blk, int kk) { return lds_rd128(rmA[kk] + off + (uint32_t)(blk * 32 * 2 * D)); };
  auto rd_v = [&](uint32_t off, int blk, int t2, int db) {
    const uint32_t o = off + (uint32_t)((32 * blk + 16 * t2) * 2 * D);
    return lds_rd_tr(trA[0][db] + o, trA[1][db] + o);
  };
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) asm volatile("" ::"v"(qf[qb][kk]));  // (see attn_fwd.h: retire the loads in the waitcnt model)
  const float c2 = a.scale * kLog2e;
  const bool fold_ok = c2 > 0.f;
  float cst_neg = 0.f, cst_pos = 0.f;
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    cst_neg = sT[0];
    cst_pos = sT[2 * a.R];
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // ------------------------------------------------------------------------------------------------------------------
  // Exact tile (unpipelined): MODE 0 generic (RPE band, N tail, causal diagonal), MODE 1 all-visible constant-bias tile.
  // Both query blocks per 32-key block: K / V fragments are read once and used twice.
  // ------------------------------------------------------------------------------------------------------------------
  auto tile_exact = [&]<int MODE>(int t, int slot, float cst) {
    const int n0 = t * BN;
    const uint32_t soff = (uint32_t)(slot * TILE);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int nb = n0 + 32 * kb;
      f32x16 s[2];
      {
        u32x4 kf[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) kf[kk] = rd_k(soff, kb, kk);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) s[qb] = mfma32<BF16>(kf[kk], qf[qb][kk], kk == 0 ? zero16 : s[qb]);
      }
      u32x4 vf[2][DB];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int db = 0; db < DB; ++db) vf[t2][db] = rd_v(soff, kb, t2, db);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int qr0 = qrow0 + 32 * qb, qrow = qr0 + lq;
        f32x16& sq = s[qb];
        float mul, add, mcand;
        if constexpr (MODE == 1) {
          mul = c2;
          add = cst;
          mcand = fmaf(max16(sq), c2, cst);
        } else {
          bool folded = fold_ok;
          float cb = 0.f;
          if constexpr (BIAS == FAT5_BIAS_RPE1D) {
            // far, edge and band blocks alike: four aligned 16-byte reads of this lane's padded table copy, window clamped (attn_common.h)
            const int R = a.R;
            folded = false;
            const float4* tp4 = reinterpret_cast<const float4*>(sTa + rpe_clamp_asc(R + nb + 4 * hi - qrow - ((R - qrow) & 3), R));
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 bq = tp4[2 * g];
              sq[4 * g + 0] = fmaf(sq[4 * g + 0], c2, bq.x);
              sq[4 * g + 1] = fmaf(sq[4 * g + 1], c2, bq.y);
              sq[4 * g + 2] = fmaf(sq[4 * g + 2], c2, bq.z);
              sq[4 * g + 3] = fmaf(sq[4 * g + 3], c2, bq.w);
            }
          } else {
            if (!folded) {
#pragma unroll
              for (int r = 0; r < 16; ++r) sq[r] *= c2;
            }
          }
          const bool nmask = nb + 32 > N;
          const bool cmask = a.causal && (nb + 31 > qr0 + P);
          if (nmask || cmask) {
            const int lim = a.causal ? min(N - 1, qrow + P) : N - 1;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (nb + crow(r, hi) > lim) sq[r] = -INFINITY;
          }
          const float m16 = max16(sq);
          mul = folded ? c2 : 1.f;
          add = folded ? cb : 0.f;
          mcand = folded ? fmaf(m16, c2, cb) : m16;
        }
        mcand = pair_max(mcand);
        if (__any(mcand > m_run[qb] + FAT5_DEFER_THR)) {
          const float m_new = fmaxf(m_run[qb], mcand);
          const float alpha = fast_exp2(m_run[qb] - ((m_new == -INFINITY) ? 0.f : m_new));
          l_run[qb][0] *= alpha;
          l_run[qb][1] *= alpha;
#pragma unroll
          for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][i][r] *= alpha;
          m_run[qb] = m_new;
        }
        const float ad = add - ((m_run[qb] == -INFINITY) ? 0.f : m_run[qb]);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float x0 = fast_exp2(fmaf(sq[r], mul, ad)), x1 = fast_exp2(fmaf(sq[r + 1], mul, ad));
          sq[r] = x0;
          sq[r + 1] = x1;
          l_run[qb][0] += x0;
          l_run[qb][1] += x1;
        }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          const u32x4 pb = pack8<BF16>(sq, t2);
#pragma unroll
          for (int db = 0; db < DB; ++db) oacc[qb][db] = mfma32<BF16>(vf[t2][db], pb, oacc[qb][db]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ------------------------------------------------------------------------------------------------------------------
  // Pipelined state (between 32-key blocks):
  //   S        raw scores of the block whose softmax is due (both query blocks)
  //   PB, VF   packed probabilities / V^T fragments of the block whose P.V product is due; the last two words of
  //            PB[1][1] are still on their way: xc = exponent arguments of that block's chunk 15, pc = probabilities of
  //            its chunk 14 (row sums not yet updated with either)
  // ------------------------------------------------------------------------------------------------------------------
  f32x16 S[2];
  u32x4 PB[2][2], VF[2][DB];
  float xc[2], pc[2];

  // One pipelined 32-key block i = 16 MFMA gaps.  Gap g holds, all mutually independent:
  //   MFMA g          g < 8: O^T += VF . PB (block i-1; VF was fetched during the previous block)   g >= 8: S' = K(KS, KB) . Q^T (block i+1)
  //   VALU            chunk g: x = s*c2 + ad (2 elements) | chunk g-1: p = exp2(x) | chunk g-2: l += p, pack to 16 bit
  //   LDS             gaps 0..3: one K fragment of (KS, KB); gaps 8..15: one ds_read_b64_tr_b16 of the V^T fragments of
  //                   (VS, VB) = block i
  // No instruction waits on another one of its own gap and consecutive MFMAs never share an accumulator.  The VALU ops
  // are volatile asm: hipcc neither reorders nor packs them (v_pk_*_f32 beside MFMAs is an anti-lever, MI355X_MICROARCH.md)
  // and pads no hazards for them -- the youngest MFMA result they read (S'[0], finished by MFMA 14) is two MFMA issue
  // periods old when chunk 0 of the next block reads it.
  auto pipe_block = [&]<int KS, int KB, int VS, int VB>(const float ad0, const float ad1) {
    static_assert(D == 64, "gap schedule written for D = 64 (16 MFMAs, 16 two-element chunks per block)");
    constexpr uint32_t koff = KS * TILE + KB * 32 * 2 * D, voff = VS * TILE + VB * 32 * 2 * D;
    u32x4 kf[KK];
    f32x16 Sn[2];
    u32x4 PBn[2][2];
    u32x2 vh[2][DB][2];  // halves of the transposed fragments
    float X[16][2], Pr[16][2];
    static_for<16>([&](auto gi) {
      constexpr int g = decltype(gi)::value;
      // ---- MFMA ----
      if constexpr (g < 8) {
        constexpr int qb = g & 1, db = (g >> 1) & 1, t2 = g >> 2;
        oacc[qb][db] = mfma32<BF16>(VF[t2][db], PB[qb][t2], oacc[qb][db]);
      } else {
        constexpr int idx = g - 8, qb = idx & 1, kk = idx >> 1;
        if constexpr (kk == 0) Sn[qb] = mfma32<BF16>(kf[kk], qf[qb][kk], zero16);
        else Sn[qb] = mfma32<BF16>(kf[kk], qf[qb][kk], Sn[qb]);
      }
      // ---- LDS ----
      if constexpr (g < KK) {
        kf[g] = lds_rd128(rmA[g] + koff);
      } else if constexpr (g >= 8) {
        constexpr int v = g - 8, t2 = v >> 2, db = (v >> 1) & 1, j2 = v & 1;
        typedef s16x4_t __attribute__((address_space(3))) * p_t;
        vh[t2][db][j2] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((p_t)(uintptr_t)(trA[j2][db] + voff + (uint32_t)(16 * t2 * 2 * D))));
      }
      // ---- VALU: add + pack of chunk g-2 ----
      {
        constexpr int c = g - 2;  // -2, -1: chunks 14, 15 of the previous block (query block 1, words 2, 3 of PB[1][1])
        constexpr int cq = c < 0 ? 1 : (c >> 3), cr = c < 0 ? 2 * (c + 8) : 2 * (c & 7);
        float p0, p1;
        if constexpr (c == -2) { p0 = pc[0]; p1 = pc[1]; }
        else if constexpr (c == -1) { p0 = Pr[15][0]; p1 = Pr[15][1]; }  // (slot 15 of this block's array is free until gap 15)
        else { p0 = Pr[c][0]; p1 = Pr[c][1]; }
#if !FAT5_F64_LMFMA
        asm_add(l_run[cq][0], p0);
        asm_add(l_run[cq][1], p1);
#endif
        const uint32_t wd = asm_cvt_pk<BF16>(p0, p1);
        if constexpr (c < 0) PB[1][1][(cr & 7) >> 1] = wd;
        else PBn[cq][cr >> 3][(cr & 7) >> 1] = wd;
      }
      // ---- VALU: exp of chunk g-1 ----
      if constexpr (g == 0) {
        Pr[15][0] = asm_exp2(xc[0]);
        Pr[15][1] = asm_exp2(xc[1]);
      } else {
        Pr[g - 1][0] = asm_exp2(X[g - 1][0]);
        Pr[g - 1][1] = asm_exp2(X[g - 1][1]);
      }
      // ---- VALU: exponent arguments of chunk g ----
      {
        constexpr int cq = g >> 3, cr = 2 * (g & 7);
        const float ad = cq == 0 ? ad0 : ad1;
        X[g][0] = asm_fma(S[cq][cr], c2, ad);
        X[g][1] = asm_fma(S[cq][cr + 1], c2, ad);
      }
#if FAT5_F64_LMFMA
      // chunk c is packed in gap c + 2: a group of four words is summed two gaps after its last one (the previous block's last group,
      // whose words 2, 3 were packed in gaps 0, 1, in gap 3); nothing reads lacc before the next block boundary, a full gap away
      if constexpr (g == 3) mfma16_acc<BF16>(lacc[1], sel, PB[1][1]);
      else if constexpr (g == 7) mfma16_acc<BF16>(lacc[0], sel, PBn[0][0]);
      else if constexpr (g == 11) mfma16_acc<BF16>(lacc[0], sel, PBn[0][1]);
      else if constexpr (g == 14) mfma16_acc<BF16>(lacc[1], sel, PBn[1][0]);
#endif
#if FAT5_F64_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
    });
    // hand over: chunk 14's probabilities and chunk 15's arguments finish inside the next block
    pc[0] = Pr[14][0]; pc[1] = Pr[14][1];
    xc[0] = X[15][0]; xc[1] = X[15][1];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      S[qb] = Sn[qb];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) PB[qb][t2] = PBn[qb][t2];
    }
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int db = 0; db < DB; ++db) VF[t2][db] = u32x4{vh[t2][db][0][0], vh[t2][db][0][1], vh[t2][db][1][0], vh[t2][db][1][1]};
  };
  // nothing pending: an all-zero product, chunk arguments of -inf (exp2 -> 0)
  auto pipe_reset = [&]() {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) PB[qb][t2] = zero4;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int db = 0; db < DB; ++db) VF[t2][db] = zero4;
    xc[0] = xc[1] = -INFINITY;
    pc[0] = pc[1] = 0.f;
  };
  // finish the pending block: its last two chunks, then its product (V fragments are in registers already)
  auto pipe_drain = [&]() {
    const float q0 = fast_exp2(xc[0]), q1 = fast_exp2(xc[1]);
#if !FAT5_F64_LMFMA
    l_run[1][0] += pc[0] + q0;
    l_run[1][1] += pc[1] + q1;
#endif
    PB[1][1][2] = pack2<BF16>(pc[0], pc[1]);
    PB[1][1][3] = pack2<BF16>(q0, q1);
#if FAT5_F64_LMFMA
    asm volatile("s_nop 1" : "+v"(PB[1][1]));  // (VALU write -> asm MFMA read: two wait states by hand)
    mfma16_acc<BF16>(lacc[1], sel, PB[1][1]);
#endif
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) oacc[qb][db] = mfma32<BF16>(VF[t2][db], PB[qb][t2], oacc[qb][db]);
    pipe_reset();
  };

  // keep l (and O) below 2^40 by an exact power of two (optimistic tiles never rescale otherwise).  A pending product was
  // formed against the old reference point: finish it first (T13 hazard: everything at the old scale is scaled exactly once).
#if FAT5_F64_LMFMA
  // fold the matrix-pipe row sums into l_run (both key halves hold the full sum and pair_sum adds the halves: half each, exact)
  auto merge_lacc = [&]() {
    asm volatile("s_nop 11" : "+v"(lacc[0]), "+v"(lacc[1]));  // (asm MFMA -> VALU read: no padding is generated; tied so no read moves above it)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      l_run[qb][0] += 0.5f * lacc[qb][0];
      lacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
#endif
  auto renorm = [&]() {
#if FAT5_F64_LMFMA
    const float lchk = fmaxf(l_run[0][0] + l_run[0][1] + lacc[0][0], l_run[1][0] + l_run[1][1] + lacc[1][0]);
#else
    const float lchk = fmaxf(l_run[0][0] + l_run[0][1], l_run[1][0] + l_run[1][1]);
#endif
    if (__builtin_expect(__any(!(lchk < 0x1p40f)), 0)) {
      pipe_drain();
#if FAT5_F64_LMFMA
      merge_lacc();
#endif
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const float lc = pair_sum(l_run[qb][0] + l_run[qb][1]);
        const int e = (lc >= 0x1p40f) ? (int)((__float_as_uint(lc) >> 23) & 0xffu) - 127 : 0;
        const float alpha = __uint_as_float((uint32_t)(127 - min(e, 126)) << 23);
        l_run[qb][0] *= alpha;
        l_run[qb][1] *= alpha;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[qb][i][r] *= alpha;
        m_run[qb] += (float)e;
      }
    }
  };

  // pipelined range of tiles [t, te), constant bias `cst`; `slot` = t % 3 on entry and te % 3 on exit.
  // Tile-iteration t runs blocks 2t, 2t+1: it reads K(t) block 1 and K(t+1) block 0 (scores of blocks 2t+1, 2t+2) and both
  // blocks of V(t) (fragments for the products of blocks 2t, 2t+1) -- inside the ring protocol's window.
  auto pipe_range = [&](int& t, const int te, int& slot, const float cst) {
    if (t >= te) return;  // (slot == 0 here: the callers run exact tiles up to a multiple of NS)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) m_run[qb] = (m_run[qb] == -INFINITY) ? 0.f : m_run[qb];  // rows without a visible key so far
    // fill: scores of the first block, nothing pending
    {
      const uint32_t soff = (uint32_t)(slot * TILE);
      u32x4 kf[KK];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) kf[kk] = rd_k(soff, 0, kk);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) S[qb] = mfma32<BF16>(kf[kk], qf[qb][kk], kk == 0 ? zero16 : S[qb]);
      pipe_reset();
    }
    // steady state: NS tiles (ring slots 0 .. NS-1) per trip, straight-line -- a slot switch inside the loop makes the
    // register allocator reconcile the bodies at every merge (tuple copies, spilled accumulators)
    auto one_tile = [&]<int SL>(int tt) {
      constexpr int S1 = (SL + 1) % NS;
      renorm();
      const float ad0 = cst - m_run[0], ad1 = cst - m_run[1];
      begin_iter(tt, SL);
      pipe_block.template operator()<SL, 1, SL, 0>(ad0, ad1);
      pipe_block.template operator()<S1, 0, SL, 1>(ad0, ad1);
      end_iter(tt);
    };
    for (; t + NS <= te; t += NS) {
      static_for<NS>([&](auto si) { one_tile.template operator()<decltype(si)::value>(t + decltype(si)::value); });
    }
    static_for<NS - 1>([&](auto si) {
      constexpr int sl = decltype(si)::value;
      if (t < te && slot == sl) {
        one_tile.template operator()<sl>(t);
        ++t;
        slot = sl + 1;
      }
    });
    pipe_drain();
#if FAT5_F64_LMFMA
    merge_lacc();
#endif
  };
  auto exact_range = [&]<int MODE>(int& t, const int te, int& slot, const float cst) {
    for (; t < te; ++t) {
      begin_iter(t, slot);
      tile_exact.template operator()<MODE>(t, slot, cst);
      slot = slot == NS - 1 ? 0 : slot + 1;
      end_iter(t);
    }
  };

  // Tile classes (workgroup-uniform):  [0, ta) FAST cst_a | [ta, tb0) generic | [tb0, tb1) FAST cst_b | [tb1, nt) generic
  int ta = 0, tb0 = 0, tb1 = 0;
  float cst_a = 0.f, cst_b = 0.f;
  if (fold_ok) {
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
    if (tb1 < tb0) tb0 = tb1 = ta;
  }

  for (int pass = 0;; ++pass) {
    const bool opt = OPT && pass == 0;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[qb][i][r] = 0.f;
      m_run[qb] = -INFINITY;
      l_run[qb][0] = l_run[qb][1] = 0.f;
    }
    if (pass > 0) {
      __syncthreads();  // every wave is done with the rings
      stage_first();
      wait_dma_all();
      __syncthreads();
    }
    int t = 0, slot = 0;
    // range A: exact baseline (two tiles), then pipelined optimistic
    exact_range.template operator()<1>(t, opt ? min(ta, NS) : ta, slot, cst_a);
    if constexpr (OPT) {
      if (opt) pipe_range(t, ta, slot, cst_a);
    }
    exact_range.template operator()<0>(t, min(max(tb0, ta), nt), slot, 0.f);
    // range B
    exact_range.template operator()<1>(t, opt ? min(tb1, max(NS, (t + NS - 1) / NS * NS)) : tb1, slot, cst_b);
    if constexpr (OPT) {
      if (opt) pipe_range(t, tb1, slot, cst_b);
    }
    exact_range.template operator()<0>(t, nt, slot, 0.f);

    if constexpr (!OPT) {
      break;
    } else {
      if (!opt) break;
      if (!(l_run[0][0] + l_run[0][1] < 0x1p120f) || !(l_run[1][0] + l_run[1][1] < 0x1p120f)) *sFlag = 1;
      __syncthreads();
      if (*sFlag == 0) break;
    }
  }

  // ---- epilogue: o = acc / l, L = m + ln(l) ----
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = qrow0 + 32 * qb + lq;
    const float l_tot = pair_sum(l_run[qb][0] + l_run[qb][1]);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (qrow < M) {
      uint16_t* orow = ob_ + (int64_t)qrow * a.os[2];
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 wv;
          wv[0] = pack2<BF16>(oacc[qb][db][4 * g + 0] * inv, oacc[qb][db][4 * g + 1] * inv);
          wv[1] = pack2<BF16>(oacc[qb][db][4 * g + 2] * inv, oacc[qb][db][4 * g + 3] * inv);
          *reinterpret_cast<u32x2*>(orow + 32 * db + 8 * g + 4 * hi) = wv;
        }
      if (hi == 0) a.lse[lse_off + qrow] = l_tot > 0.f ? (m_run[qb] + fast_log2(l_tot)) * kLn2 : -INFINITY;
    }
  }
}

template <int D, bool BF16, int BIAS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FAT5_F64_MINW)))
void attn_fwd64_kernel(const AttnArgs a, const int n_items) {
  // gridDim.x == n_items (one workgroup per item, the default) or a smaller multiple of 8 (persistent workgroups walking the
  // items with stride gridDim.x; an item keeps the XCD = index % 8 that decode_block() gave its (b, h)).
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    attn_fwd64_body<D, BF16, BIAS>(a, item);
    __syncthreads();  // this item's last LDS readers (table, flag) are done before the next item overwrites them
  }
}

}  // namespace fat5
