// FlashAttention-2 backward with a DENSE additive bias shared by the batch -- the reference's own operator, `flash_attention_v2_bias(q, k, v,
// bias(1,H,M,N))` (src/model/ops/flash_attention_v2_bias.py:228-288; caller modeling_flash_t5.py:280-285): dQ AND the batch-reduced bias gradient
// in ONE pipelined body on gfx950 (round 5).
//
// The reference writes the whole dS tensor (B,H,M,N) from its dK/dV kernel (:716-729) and sums it over the batch afterwards (`ds.sum(0)`,
// :149-162, :214-215): 2 * B*H*M*N*2 bytes of HBM traffic that decide its backward.  Rounds 1-4 of this library either staged the
// same tensor from the 32-row dQ body (B * H * M * N * 2 <= 64 MB) or recomputed S / dP a third time in a batch-inner kernel (attn_bwd_dbias.h:
// 2.3 ms of the 6.2 ms dense backward at (4,12,8192,64)).  Here:
//
//  * A workgroup owns 64 query rows of ONE head and FOUR batch elements: wave w runs the 64-rows-per-wave pipelined dQ body (attn_bwd64.h:
//    attn_bwd_q64_body -- S^T = K Q^T, dP^T = V dO^T - delta, dQ^T += K^T dS^T; 24 MFMA gaps per 32-key step) for batch element 4 c + w.  The four
//    waves see the SAME (64 rows x 32 keys) bias tile: it travels global -> LDS once per step and workgroup (one 1-KiB DMA piece per wave) and every
//    wave reads its elements as two 16-byte LDS reads per query block.
//  * The batch reduction of dS never leaves the CU: every wave writes its step's dS, rounded to the bias dtype exactly like the reference
//    (`ds.to(dtype)`, :720), into an LDS exchange buffer (4 x 16 bytes per lane); behind the step's barrier every wave sums one quarter of the tile
//    (16 rows x 32 keys) over the four waves ON THE MATRIX PIPE -- four v_mfma_f32_16x16x32 with a 0/1 selector as the A operand: each adds the
//    16-bit values of two waves in fp32, 64 pipe cycles per step instead of 60 VALU operations per lane -- rounds once and stores 32-byte row pieces
//    of dbias (fp32 partial sums per group of four batch elements when B > 4; `dbias_partial_reduce_kernel` adds the groups in a fixed order).
//    Deterministic: fixed summation order, no atomics.  The reference's sum (fp32 accumulation of bf16-rounded dS, one final rounding) is reproduced.
//  * The bias is added ON THE MATRIX PIPE (as in the dense form of the 64-key dK/dV body, attn_bwd64.h: Bwd64Cfg): S^T = K Q^T + B^T E with B^T the bias tile read
//    as a transposed operand fragment (lane = key slot, k-slot = query row; ds_read_b64_tr_b16 on the tile: its suppliers point at the 4-key groups of
//    the key permutation below) and E[k][row] = 1/scale where k-slot k is that row -- as two 16-bit terms (hi + lo) in two k-slots that see the same bias
//    row, so a fragment holds one 8-byte transposing read twice and one MFMA covers 8 rows: eight more MFMAs per step instead of 96 VALU instructions
//    per lane.  The causal mask rides in the C operand of the same MFMAs (-inf where the
//    key is masked: the steps on the diagonal run the pipelined iteration, and a causal sweep is padded to whole trips of four steps with fully masked ones).
//  * Keys are PERMUTED inside a step (LDS row rho of the K / V images holds key pi(rho), qdb_pi below): the MFMA k-slot <-> key mapping is free as
//    long as both operands agree, and with this one a lane's 16 score registers of a query block are the 16 keys 16 hi .. 16 hi + 15 of its row --
//    its bias values are 32 contiguous bytes -- in the order {0-3, 8-11 | 4-7, 12-15}: the two packed operand fragments of its rounded dS are, as they
//    stand, the two 16-byte chunks whose interleave leaves every lane of the reduction 8 consecutive keys (one 16-byte dbias store per lane).
//  * K / V rings are wave-private (every wave has its own batch element): four K slots and three V slots of 4 KiB per wave, filled by LDS-DMA three
//    steps ahead, ONE piece per MFMA gap (all 36 pieces of a workgroup issued at once queue up in front of the CU's one texture-address unit -- 16
//    cycles each -- and stall the in-order waves behind them: measured 480 us of a 2.1 ms launch at (4,12,8192)); the only workgroup barrier of a step
//    is the one the shared bias tile and the dS exchange need.  The ring area doubles as the staging area of the prologue (the wave's 64 rows of
//    Q | dO | O arrive as whole rows, attn_bwd64.h).
#pragma once
#include "attn_common.h"
#include "attn_bwd64.h"

#ifndef FAT5_QDB_ABL
#define FAT5_QDB_ABL 0  // developer ablations (timing only, wrong results): 1 no dbias stores, 2 no exchange (writes, reads, reduction MFMAs), 4 no bias reads / conversions, 8 no barrier, 16 no bias DMA, 32 no K / V DMA after the prologue, 64 dbias stores all to the same 4 KiB, 128 non-temporal dbias stores, 256 DMA pieces staggered by wave parity
#endif

namespace fat5 {

template <int D>
struct BwdQdb64Cfg {
  static constexpr int NW = 4, BM = 64, KT = 32, NT = 64 * NW;
  static constexpr int NSK = 4, NSV = 3, NSB = 4;  // ring slots: K, V (its image is dead one iteration after it lands in use: three slots give the same prefetch distance), bias
  static constexpr int IMG = rm_bytes<D, KT>();   // one 32-key image (K or V): 4 KiB
  static constexpr int VOFF = NSK * IMG;          // a wave's V ring behind its K ring
  static constexpr int WRING = (NSK + NSV) * IMG; // one wave's private rings (28 KiB) = its prologue staging area (Q | dO | O images of 64 rows: 24 KiB)
  static_assert(WRING >= 3 * 64 * 2 * D, "the private rings hold the wave's three staged 64-row images");
  static constexpr int BT = BM * KT * 2;          // the step's bias tile (64 rows x 32 keys, 16 bit): 4 KiB
  static constexpr int BOFF = NW * WRING;         // bias ring
  static constexpr int XOFF = BOFF + NSB * BT;    // two exchange buffers of four dS tiles each
  static constexpr int XB = NW * BT;
  static_assert((XB & (XB - 1)) == 0 && (XOFF & XB) == 0, "the exchange buffer of a step is selected by one XOR");
  static constexpr int SMEM = XOFF + 2 * XB;      // 160 KiB: the whole LDS of a CU
  static_assert(SMEM <= 160 * 1024, "LDS");
  static constexpr int PIECES = 9;                // LDS-DMA pieces per wave and step: 4 K + 4 V + 1 bias
};

// key (inside the step) of score register r of a lane with half `hi` (C-layout row rho = crow(r, hi) of the K / V images holds it; see the header)
FAT5_DEV constexpr int qdb_key(int r, int hi) { return 16 * hi + (r & 3) + 8 * ((r >> 2) & 1) + 4 * (r >> 3); }
// bias word (of the lane's eight: keys 16 hi + 2 j, + 1) and half that holds register r's key
FAT5_DEV constexpr int qdb_word(int r) { return qdb_key(r, 0) >> 1; }

// PARTIAL: the sums go out as fp32 (one (H, M, N) slab per group of four batch elements) instead of the final 16-bit dbias
// ONE: 1 / scale is itself a 16-bit value (1: T5, 8: the default) -- one selector term, 16 bias rows per MFMA: four bias MFMAs per step instead of eight
// bid: the workgroup's index among the body's own (the launch may hold other workgroups in front: attn_bwd_dfused64_kernel; a multiple of eight of them, so that
// bid % 8 stays the XCD); wstats: write the row statistics the dK/dV kernels read (a.stat2) -- off when bwd_stat2_kernel wrote them ahead of the launch
template <int D, bool BF16, bool PARTIAL, bool ONE>
FAT5_DEV void attn_bwd_qdb64_body(const AttnArgs& a, void* dbias_out, const int bid, const bool wstats) {
  static_assert(D == 64, "gap schedule written for D = 64");
  using Cfg = BwdQdb64Cfg<D>;
  constexpr int IMG = Cfg::IMG, BT = Cfg::BT;
  constexpr int KK = D / 16, DB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, lq = l & 31, hi = l >> 5;
  // workgroup -> (head, group of four batch elements, 64-row block): the row blocks of one (head, group) share an XCD (their K / V stay in its L2)
  // Work items (pair-major: (head, group) x row block) are cut into eight contiguous chunks, one per XCD (workgroup -> XCD blockIdx % 8): an XCD
  // walks the row blocks of ONE (head, group) at a time -- its 32 CUs stream that pair's K / V (4 x 2 x N x 128 bytes) through the XCD's L2 roughly in
  // step -- whatever the number of pairs (12 heads x 1 group: 1.5 pairs per XCD; the round-robin deal of decode_block needs a multiple of 8).
  const int ngrp = (a.B + 3) >> 2;
  int pair, mblk;
  {
    const int W = a.H * ngrp * a.n_mblk, x = bid & 7, idx = bid >> 3;
    const int base = W >> 3, rem = W & 7;
    if (idx >= base + (x < rem ? 1 : 0)) return;  // (grid = 8 ceil(W / 8))
    const int item = x * base + min(x, rem) + idx;
    const int npair = a.H * ngrp;
    if (FAT5_CAUSAL_ORDER && a.causal && (npair & 7) == 0) {
      // causal: row-block-major over the XCD's pairs (x, x + 8, ...), the blocks that see the most keys first -- longest-first list scheduling (decode_block, order 1)
      const int per = npair >> 3, tq = idx / per;
      pair = (idx - tq * per) * 8 + x;
      mblk = a.n_mblk - 1 - tq;
    } else {
      pair = fast_div(item, a.n_mblk, a.mg_mblk);
      mblk = item - pair * a.n_mblk;
      if (a.causal) mblk = a.n_mblk - 1 - mblk;  // (the row blocks that see the most keys first inside every pair)
    }
  }
  const int h = pair / ngrp, grp = pair - h * ngrp;
  const int b_ = 4 * grp + w;
  const bool bvalid = b_ < a.B;  // (a wave beyond the batch runs on the last element with every probability forced to zero: it adds nothing and stores nothing)
  const int b = bvalid ? b_ : a.B - 1;
  const int M = a.M, N = a.N;
  const int m0 = mblk * Cfg::BM;
  if (m0 >= M) return;
  const uint16_t* qb_ = a.q + (int64_t)b * a.qs[0] + (int64_t)h * a.qs[1];
  const uint16_t* kb_ = a.k + (int64_t)b * a.ks[0] + (int64_t)h * a.ks[1];
  const uint16_t* vb_ = a.v + (int64_t)b * a.vs[0] + (int64_t)h * a.vs[1];
  const uint16_t* ob_ = a.o + (int64_t)b * a.os[0] + (int64_t)h * a.os[1];
  const uint16_t* dob_ = a.dout + (int64_t)b * a.dos[0] + (int64_t)h * a.dos[1];
  uint16_t* dqb_ = a.dq + (int64_t)b * a.dqs[0] + (int64_t)h * a.dqs[1];
  const int64_t stat_off = ((int64_t)b * a.H + h) * a.M;
  const int P = N - M;
  int n_end = N;
  if (a.causal) n_end = min(N, m0 + Cfg::BM + P);
  int nt = n_end > 0 ? (n_end + 31) / 32 : 0;
  // causal: whole trips of the pipelined loop -- the steps added lie above the diagonal (every probability an exact zero through the mask in the score
  // MFMAs' C operand; their dbias tiles are written as the zeros they are) and cost a third of the general iterations they replace
  if (a.causal && nt > 0 && ((nt + 3) & ~3) * 32 <= N) nt = (nt + 3) & ~3;
  const int qw0 = m0;  // (every wave the same 64 rows)

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t ring_w = lds0 + (uint32_t)(w * Cfg::WRING);  // this wave's ring / staging area
  float Lq_[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) Lq_[qb] = a.lse[stat_off + min(qw0 + 32 * qb + lq, M - 1)];
  // ---- prologue: the wave's rows of Q | dO | O as three swizzled row-major images (rows past M arrive as zeros) ----
  {
    using SDma = DmaStage<D, 64, 64>;
    static_assert(SDma::PER == 8 && SDma::NV == 2, "eight 1-KiB pieces of 8 rows per tensor");
    SDma sq, sdo, so;
    sq.init(a.qs[2], l);
    sdo.init(a.dos[2], l);
    so.init(a.os[2], l);
    const __amdgpu_buffer_rsrc_t qrs = make_rows_rsrc(qb_, a.qs[2], M, D), dors = make_rows_rsrc(dob_, a.dos[2], M, D), ors = make_rows_rsrc(ob_, a.os[2], M, D);
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)ring_w);
    const uint32_t q0 = (uint32_t)qw0 * (uint32_t)a.qs[2] * 2u, do0 = (uint32_t)qw0 * (uint32_t)a.dos[2] * 2u, o0 = (uint32_t)qw0 * (uint32_t)a.os[2] * 2u;
#pragma unroll
    for (int i = 0; i < SDma::PER; ++i) {
      dma16_asm(qrs, dst + (uint32_t)(i * 1024), sq.voff[i % 2], q0 + sq.piece_step * (i / 2));
      dma16_asm(dors, dst + (uint32_t)(8192 + i * 1024), sdo.voff[i % 2], do0 + sdo.piece_step * (i / 2));
      dma16_asm(ors, dst + (uint32_t)(16384 + i * 1024), so.voff[i % 2], o0 + so.piece_step * (i / 2));
    }
  }
  FragAddr<D> fa;
  fa.init(l);
  wait_dma_all();  // (own pieces only: the area is private)
  u32x4 qf[2][KK], dof[2][KK];
  float nL2[2];
  f32x16 nd16[2];
  {
    u32x4 off_[2][KK];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const uint32_t ad = ring_w + (uint32_t)(fa.rm[kk] + qb * 32 * 2 * D);
        qf[qb][kk] = lds_rd128(ad);
        dof[qb][kk] = lds_rd128(ad + 8192u);
        off_[qb][kk] = lds_rd128(ad + 16384u);
      }
    // delta = rowsum(o * do) (reference _bwd_preprocess, :516-556) and the row statistics the dK/dV kernels read (attn_bwd64.h)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qrow = qw0 + 32 * qb + lq;
      float dsum = 0.f;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dsum = fmaf(cvt_lo<BF16>(off_[qb][kk][j]), cvt_lo<BF16>(dof[qb][kk][j]), dsum);
          dsum = fmaf(cvt_hi<BF16>(off_[qb][kk][j]), cvt_hi<BF16>(dof[qb][kk][j]), dsum);
        }
      const float delta = pair_sum(dsum);
      if (bvalid && a.delta && qrow < M && hi == 0) a.delta[stat_off + qrow] = delta;
      const float Lq = Lq_[qb];
      nL2[qb] = (Lq < kDeadRowLse || !bvalid) ? -INFINITY : -Lq * kLog2e;  // (dead rows: attn_bwd.h)
      if (wstats && bvalid && a.stat2 && hi == 0 && qrow < (M + 31) / 32 * 32) {
        float* st = a.stat2 + (((int64_t)b * a.H + h) * ((M + 31) / 32) + (qrow >> 5)) * 64 + (qrow & 31);
        const bool live = qrow < M && !(Lq < kDeadRowLse);
        st[0] = live ? -Lq / a.scale : (a.scale > 0.f ? -INFINITY : INFINITY);
        st[32] = qrow < M ? -delta : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) nd16[qb][r] = -delta;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the fragments are in registers before the ring's first requests overwrite the images)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) asm volatile("" : "+a"(qf[qb][kk]), "+a"(dof[qb][kk]));  // MFMA-only operands: AGPRs

  f32x16 dq[2][DB];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[qb][i][r] = 0.f;

  // ---- rings: key step t lives in slot t % 4 of the wave's K ring, t % 3 of its V ring (rows permuted by pi) and t % 4 of the shared bias ring ----
  // one K (V) image = four 1-KiB pieces of 8 LDS rows: piece i holds rows rho = 8 i + r8 = crow(r, hi) with r = (r8 & 3) + 4 i, hi = r8 >> 2, i.e. keys
  // (r8 & 3) + 16 (r8 >> 2) + 8 (i & 1) + 4 (i >> 1); the swizzle of row 8 i + r8 depends on i through (2 i + (r8 >> 2)) & 3: two phases (i even / odd)
  uint32_t kvo[2], vvo[2];
  {
    const int r8 = l >> 3, slot = l & 7;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int rho = 8 * ph + r8;  // (rows 8 i + r8 with the same parity of i share the swizzle)
      const int src = slot ^ swz<D>(rho);
      const int key = (r8 & 3) + 16 * (r8 >> 2);
      kvo[ph] = (uint32_t)(key * a.ks[2] * 2 + (src << 4));
      vvo[ph] = (uint32_t)(key * a.vs[2] * 2 + (src << 4));
    }
  }
  const __amdgpu_buffer_rsrc_t krs = make_rows_rsrc(kb_, a.ks[2], N, D);
  const __amdgpu_buffer_rsrc_t vrs = make_rows_rsrc(vb_, a.vs[2], N, D);
  const uint32_t kstride_b = (uint32_t)a.ks[2] * 2u, vstride_b = (uint32_t)a.vs[2] * 2u;
  // bias tile of a step: 64 rows x 64 bytes, row-major with the 16-byte chunks XOR-ed by (row >> 2) & 3 (conflict-free for the lanes' 32-byte reads,
  // the exchange buffers use the same layout); wave w fetches rows 16 w .. 16 w + 15: lane -> (row 16 w + l / 4, slot l % 4)
  const uint16_t* bias_h = a.bias + (int64_t)h * a.bs[1];
  const __amdgpu_buffer_rsrc_t brs = make_rows_rsrc(bias_h, a.bs[2], M, N);
  const uint32_t bvo = (uint32_t)((l >> 2) * a.bs[2] * 2 + ((((l & 3) ^ ((l >> 4) & 3))) << 4));
  const uint32_t brow0 = (uint32_t)(m0 + 16 * w) * (uint32_t)a.bs[2] * 2u;
  const uint32_t ring_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)ring_w);
  const uint32_t bias_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (uint32_t)(Cfg::BOFF + w * 1024)));
  // piece k of step t: 0..3 K, 4..7 V, 8 the bias rows of this wave.  Steps past the end are requested like any other (rows past N arrive as zeros,
  // nobody reads them): the counted wait below then holds for every iteration
  auto dma_piece = [&](const uint32_t tt, const int k) {
    if (k < 4) {
      if (!(FAT5_QDB_ABL & 32)) dma16_asm(krs, ring_s + (tt & 3u) * (uint32_t)IMG + (uint32_t)(k * 1024), kvo[k & 1], (tt * 32u + (uint32_t)(8 * (k & 1) + 4 * (k >> 1))) * kstride_b);
    } else if (k < 8) {
      const int i = k - 4;
      if (!(FAT5_QDB_ABL & 32)) dma16_asm(vrs, ring_s + (uint32_t)Cfg::VOFF + (tt % 3u) * (uint32_t)IMG + (uint32_t)(i * 1024), vvo[i & 1], (tt * 32u + (uint32_t)(8 * (i & 1) + 4 * (i >> 1))) * vstride_b);
    } else {
      if (!(FAT5_QDB_ABL & 16)) dma16_asm(brs, bias_s + (tt & 3u) * (uint32_t)BT, bvo, brow0 + tt * 64u);
    }
  };
  auto dma_step = [&](int t) {
    const uint32_t tt = (uint32_t)__builtin_amdgcn_readfirstlane(t);
#pragma unroll
    for (int k = 0; k < Cfg::PIECES; ++k) dma_piece(tt, k);
  };
  // E(t): step t+1 has landed (own K / V pieces; the bias pieces of all waves behind the barrier); every wave is done with step t-1 (its bias
  // tile, and the exchange buffer of step t-2); the K / bias slots of step t-1 and the V slot of step t take step t+3.  The wait is COUNTED: LDS-DMA
  // requests retire in order, so with at most PIECES requests pending the pending LOADS are among the nine of step t+2 -- whatever the dbias stores in between do
  // (stores share the counter and are not ordered against loads: a pending store only makes the wait wait for a piece of step t+2 as well).
  auto wait_step = [&]() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::PIECES) : "memory");
    if constexpr (!(FAT5_QDB_ABL & 8)) __syncthreads();
  };
  auto sync_step = [&](int t) {
    wait_step();
    dma_step(t + 3);
  };
#pragma unroll
  for (int i = 0; i < 3; ++i) dma_step(i);
  // K slot 3 is read (against an all-zero dS) before anything lands in it: finite contents
#pragma unroll
  for (int i = 0; i < IMG / (64 * 16); ++i) *reinterpret_cast<u32x4*>(smem + w * Cfg::WRING + 3 * IMG + (i * 64 + l) * 16) = u32x4{0u, 0u, 0u, 0u};
  wait_step();

  // per-lane LDS addresses (ring base folded in; slot / image offsets are immediates)
  uint32_t rmA[KK], trA[2][DB];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    rmA[kk] = ring_w + (uint32_t)fa.rm[kk];
    asm volatile("" : "+v"(rmA[kk]));
  }
#pragma unroll
  for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      trA[j2][db] = ring_w + (uint32_t)fa.tr[j2][db];
      asm volatile("" : "+v"(trA[j2][db]));
    }
  // bias / exchange tiles: this lane's row 32 qb + lq, chunks 2 hi and 2 hi + 1 (keys 16 hi .. 16 hi + 15)
  const int fsw = (lq >> 2) & 3;
  uint32_t tA[2];  // byte offset inside a tile of chunk 2 hi + i of row lq (query block qb: + 2048)
#pragma unroll
  for (int i = 0; i < 2; ++i) tA[i] = (uint32_t)(lq * 64 + (((2 * hi + i) ^ fsw) << 4));
  // bias fragments (A operands of the bias MFMAs): ds_read_b64_tr_b16 with this lane as SUPPLIER of row 16 t2 + 8 j2 + 4 hi + (i >> 2) (i = l & 15), keys
  // 16 (c & 1) + 8 (c >> 1) + 4 g .. + 3 (c = i & 3, g = (l >> 4) & 1): the 4-key group whose slots 16 g + 4 c .. + 3 the receiving lanes of its 16 hold
  uint32_t btA[2];  // [jj & 1]; query block qb: + 2048, jj >> 1: + 1024, ring slot: + BT s
  {
    const int i = l & 15, c = i & 3, g = (l >> 4) & 1;
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const int row = 8 * j2 + 4 * hi + (i >> 2);
      btA[j2] = lds0 + (uint32_t)(Cfg::BOFF + row * 64 + (((2 * (c & 1) + (c >> 1)) ^ ((2 * j2 + hi) & 3)) << 4) + 8 * g);
      asm volatile("" : "+v"(btA[j2]));
    }
  }
  // selector operands E(jj) (B: lane = query row lq; k-slot (hi, j) <-> bias row 8 jj + 4 hi + (j & 3), 1/scale's leading 16 bits for j < 4, the next 16 for j >= 4)
  u32x4 selB[4];
  const uint32_t blim2 = bias_mfma_limit<BF16>(a.scale);  // (packed clamp of the bias words ahead of their MFMAs)
  {
    const float invf = 1.f / a.scale;
    uint32_t ih, il;
    split16<BF16>(invf, ih, il);
#pragma unroll
    for (int jj = 0; jj < (ONE ? 2 : 4); ++jj) {
      uint32_t wv[4];
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2) {
        // (ONE: operand jj = t2 covers rows 16 t2 + 8 (j >> 2) + 4 hi + (j & 3), every slot 1 / scale)
        const int r0 = ONE ? 16 * jj + 8 * ((2 * j2) >> 2) + 4 * hi + ((2 * j2) & 3) : 8 * jj + 4 * hi + ((2 * j2) & 3);
        const uint32_t val = (ONE || j2 < 2) ? ih : il;
        wv[j2] = (r0 == lq ? val : 0u) | (r0 + 1 == lq ? val << 16 : 0u);
      }
      selB[jj] = u32x4{wv[0], wv[1], wv[2], wv[3]};
      asm volatile("" : "+v"(selB[jj]));
    }
  }
  const float ninf_c = a.scale > 0.f ? -INFINITY : INFINITY;  // a raw score this large is a zero probability
  // exchange: this wave's tile of buffer 0 (writes).  A lane's 16 rounded dS values of a query block (keys 16 hi + 0 .. 15) go out as chunk 2 hi = keys
  // {0-3, 8-11} and chunk 2 hi + 1 = keys {4-7, 12-15} of its row: with that interleave the reduction leaves every lane 8 CONSECUTIVE keys (one 16-byte
  // dbias store per lane and step, whole 64-byte row pieces per four lanes).  The reader's side: lane (n = l & 15, gq = l >> 4) sums row 16 w + n; for
  // accumulator `ac` and MFMA `m` it reads chunk 2 (gq & 1) + ac of that row in the tile of wave 2 m + (gq >> 1)
  uint32_t xw[2] = {lds0 + (uint32_t)(Cfg::XOFF + w * BT) + tA[0], lds0 + (uint32_t)(Cfg::XOFF + w * BT) + tA[1]};
  uint32_t xr[2];
  {
    const int n = l & 15, gq = l >> 4;
#pragma unroll
    for (int ac = 0; ac < 2; ++ac)
      xr[ac] = lds0 + (uint32_t)(Cfg::XOFF + (gq >> 1) * BT + (16 * w + n) * 64 + ((((2 * (gq & 1) + ac) ^ ((n >> 2) & 3))) << 4));
  }
  // selector of the reduction MFMAs: A (16 x 32), row i = l & 15, k = 8 (l >> 4) + j: one iff k mod 16 == i
  u32x4 sel16;
  {
    const int i16 = l & 15, kb8 = 8 * ((l >> 4) & 1);
    uint32_t wv[4];
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
      constexpr uint32_t one16 = BF16 ? 0x3F80u : 0x3C00u;  // 1.0
      const uint32_t lo = (kb8 + 2 * j2 == i16) ? one16 : 0u, hi16 = (kb8 + 2 * j2 + 1 == i16) ? one16 : 0u;
      wv[j2] = lo | (hi16 << 16);
    }
    sel16 = u32x4{wv[0], wv[1], wv[2], wv[3]};
    asm volatile("" : "+v"(sel16));
  }
  // dbias stores: lane (n, gq) holds keys 8 gq .. 8 gq + 3 (accumulator 0) and 8 gq + 4 .. 8 gq + 7 (accumulator 1) of row 16 w + n of the step's tile
  constexpr int ESZ = PARTIAL ? 4 : 2;
  char* dbh = reinterpret_cast<char*>(dbias_out) + ((int64_t)(PARTIAL ? grp * a.H + h : h) * M * N) * ESZ;
  const __amdgpu_buffer_rsrc_t dbrs = __builtin_amdgcn_make_buffer_rsrc(dbh, 0, __builtin_amdgcn_readfirstlane((int)((int64_t)M * N * ESZ)), 0x00020000);
  const uint32_t dbrec = (uint32_t)((int64_t)M * N * ESZ);
  const uint32_t dvo = (uint32_t)(((m0 + 16 * w + (l & 15)) * N + 8 * (l >> 4)) * ESZ);
  const bool row_ok = m0 + 16 * w + (l & 15) < M;

  const float c2 = a.scale * kLog2e;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // Pipeline state between two iterations (iteration i = key step i is in its softmax stage):
  //   S, DP     S^T = K Q^T and dP'^T = V dO^T - delta of step i (lane = query row, register r <-> key 32 i + 16 hi + r)
  //   DSB       dS^T of step i-1 rounded to the bias dtype (:720): B operands of the dQ products AND the words the batch sum is made of
  //   TRK       the K^T fragments (t2 = 0; db = 0, 1) of step i-1
  f32x16 S[2], DP[2];
  u32x4 DSB[2][2], TRK[2];

  auto rd_tr = [&](uint32_t off, int t2, int db) {
    const uint32_t o = off + (uint32_t)(16 * t2 * 2 * D);
    return lds_rd_tr(trA[0][db] + o, trA[1][db] + o);
  };
  // scores of the step whose K image sits at byte offset ko of the wave's K ring and whose V image at VOFF + vo
  // the causal mask as the score MFMAs' C operand: register r of query block qb holds key nb + qdb_key(r, hi) -- masked where that exceeds row + P
  const int lim0 = qw0 + lq + P - 16 * hi;  // (query block 1: + 32)
  auto mask_c = [&](const int nb, f32x16 (&out)[2]) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int lm = lim0 + 32 * qb - nb;
#pragma unroll
      for (int r = 0; r < 16; ++r) out[qb][r] = (qdb_key(r, 0) > lm) ? ninf_c : 0.f;
    }
  };
  // scores of the step at key nb: its K image at byte offset ko of the wave's K ring, its V image at VOFF + vo, its bias tile at BOFF + bo
  auto score_step = [&](const uint32_t ko, const uint32_t vo, const uint32_t bo, const int nb, f32x16 (&Sx)[2], f32x16 (&DPx)[2]) {
    u32x4 kf[KK], vf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      kf[kk] = lds_rd128(rmA[kk] + ko);
      vf[kk] = lds_rd128(rmA[kk] + (uint32_t)Cfg::VOFF + vo);
    }
    if (a.causal) mask_c(nb, Sx);
    else { Sx[0] = zero16; Sx[1] = zero16; }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) Sx[qb] = mfma32<BF16>(kf[kk], qf[qb][kk], Sx[qb]);
    if constexpr (ONE) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          const uint32_t o = bo + (uint32_t)(qb * 2048 + t2 * 1024);
          Sx[qb] = mfma32<BF16>(bias_clamp_frag(lds_rd_tr(btA[0] + o, btA[1] + o), blim2), selB[t2], Sx[qb]);
        }
    } else {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          const u32x2 bh_ = bias_clamp_frag(lds_rd_tr_half(btA[jj & 1] + bo + (uint32_t)(qb * 2048 + (jj >> 1) * 1024)), blim2);
          Sx[qb] = mfma32<BF16>(u32x4{bh_[0], bh_[1], bh_[0], bh_[1]}, selB[jj], Sx[qb]);
        }
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) DPx[qb] = mfma32<BF16>(vf[kk], dof[qb][kk], kk == 0 ? nd16[qb] : DPx[qb]);
  };
  auto product_step = [&](const uint32_t so) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const u32x4 kt = rd_tr(so, t2, db);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) mfma_acc_agpr<BF16>(dq[qb][db], kt, DSB[qb][t2]);
      }
  };
  // the exchange of one step's dS (buffer `xo` = 0 or XB): write this wave's tile | (barrier) | sum one quarter over the four waves, store it
  auto lds_wr128 = [](uint32_t addr, const u32x4 v) {
    typedef u32x4 __attribute__((address_space(3))) * p_t;
    *(p_t)(uintptr_t)addr = v;
  };
  auto x_write = [&](const uint32_t xo) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int i = 0; i < 2; ++i) lds_wr128(xw[i] + xo + (uint32_t)(qb * 2048), DSB[qb][i]);
  };
  // the reduced quarter tile leaves: one 16-byte (PARTIAL: two) store per lane; vo: this lane's offset or the out-of-range marker
  auto x_store = [&](const f32x4 (&acc)[2], const uint32_t vo, const uint32_t so) {
    if constexpr (PARTIAL) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[0]), dbrs, vo, so, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[1]), dbrs, vo + 16u, so, 0);
    } else {
      const u32x4 w4 = {pack2<BF16>(acc[0][0], acc[0][1]), pack2<BF16>(acc[0][2], acc[0][3]), pack2<BF16>(acc[1][0], acc[1][1]), pack2<BF16>(acc[1][2], acc[1][3])};
      if constexpr (FAT5_QDB_ABL & 64) __builtin_amdgcn_raw_buffer_store_b128(w4, dbrs, (uint32_t)(l * 16 + w * 1024), 0, 0);  // (always the same 4 KiB)
      else if constexpr (FAT5_QDB_ABL & 128) __builtin_amdgcn_raw_buffer_store_b128(w4, dbrs, vo, so, 2);  // (nt)
      else __builtin_amdgcn_raw_buffer_store_b128(w4, dbrs, vo, so, 0);
    }
  };
  // step: the key step the buffer holds (-1: nothing to store); full: every key of the step exists
  auto x_reduce_store = [&](const uint32_t xo, const int step, const bool full) {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int ac = 0; ac < 2; ++ac) acc[ac] = mfma16<BF16>(sel16, lds_rd128(xr[ac] + xo + (uint32_t)(m * 2 * BT)), acc[ac]);
    // (step -1: everything out of range -> dropped by the hardware)
    const uint32_t so = step >= 0 ? (uint32_t)(step * 32 * ESZ) : dbrec;
    const bool ok = row_ok && (full || step * 32 + 8 * (l >> 4) < N);
    if constexpr (!(FAT5_QDB_ABL & 1)) x_store(acc, ok ? dvo : 0x80000000u, so);
  };
  // general softmax stage of the key step at nb (key tail / causal diagonal): S, DP -> DSB
  auto softmax_generic = [&](const int nb) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16& s = S[qb];
      const f32x16& dp = DP[qb];
      const int qrow = qw0 + 32 * qb + lq;
      const float nl = nL2[qb];
      // (the scores arrive with bias / scale inside)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], c2, nl);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r]) * dp[r];  // dS = P (dP - delta)   (:713)
      const int lim = a.causal ? min(N - 1, qrow + P) : N - 1;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (nb + qdb_key(r, 0) + 16 * hi > lim) s[r] = 0.f;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) DSB[qb][t2] = pack8<BF16>(s, t2);
    }
  };
  uint32_t xpar = 0u;  // exchange buffer of the step whose dS is in DSB (toggles every step)
  uint32_t vnext = (uint32_t)IMG;  // byte offset (inside the V ring) of the V image of the step after the one in its softmax stage
  auto generic_iter = [&](const int t) {
    const uint32_t kp = (uint32_t)(((t + 3) & 3) * IMG), kc = (uint32_t)((t & 3) * IMG), kn = (uint32_t)(((t + 1) & 3) * IMG);
    x_write(xpar);
    product_step(kp);
    sync_step(t);
    x_reduce_store(xpar, t - 1, (t - 1) * 32 + 32 <= N);
    xpar ^= (uint32_t)Cfg::XB;
    f32x16 Sn[2], DPn[2];
    score_step(kn, vnext, (uint32_t)(((t + 1) & 3) * BT), (t + 1) * 32, Sn, DPn);
    vnext = vnext == (uint32_t)(2 * IMG) ? 0u : vnext + (uint32_t)IMG;
    softmax_generic(t * 32);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      S[qb] = Sn[qb];
      DP[qb] = DPn[qb];
    }
    TRK[0] = rd_tr(kc, 0, 0);
    TRK[1] = rd_tr(kc, 0, 1);
  };

  // One pipelined iteration = 24 MFMA gaps (attn_bwd64.h: attn_bwd_q64_body::fast_iter) + the dense bias + the exchange:
  //   MFMA   g < 8: dQ^T[qb][db] += K^T(t2, db) . dS^T[qb][t2] of step i-1; 8..15: S^T[qb] of step i+1, TWO MFMAs per gap (query block 0 at the head of
  //          the gap, query block 1 behind its LDS section; the two accumulators alternate strictly): k-steps 0..3, then + bias / scale (the four 8-row
  //          groups); 16..23: dP'^T[qb] (C = -delta on the first k-step)
  //   MFMA16 gaps 7..10: the batch sum of step i-1's dS quarter (two accumulators x two wave pairs)
  //   VALU   32 elements per lane opened evenly over gaps 0 .. 20: x = s * c2 - L2 | one gap later p = exp2(x) | one more: ds = p * dp' | pairs packed
  //          once both halves exist.  MK: the C operands of step i+1's scores (mask_c) in gaps 5..7
  //   LDS    gaps 0..3: the K^T fragments (t2 = 1) of step i-1 and this wave's four exchange writes; gap 4: the barrier E(i); gaps 5..13 the nine DMA
  //          pieces of step i+3; gaps 4..7 the K, 12..15 the V row-major fragments of step i+1; 5, 6: the four exchange reads; gaps 8..11: the bias
  //          fragments of step i+1 (two transposing reads each); gap 14: the dbias store of step i-1; gaps 20..23 the K^T fragments (t2 = 0) of step i
  constexpr int NG = 24;
  auto fast_iter = [&]<int SL, bool MK>(const int t) {
    constexpr uint32_t o_prev = ((SL + 3) & 3) * IMG, o_cur = SL * IMG, o_next = ((SL + 1) & 3) * IMG, b_next = ((SL + 1) & 3) * BT;
    const uint32_t v_next = (uint32_t)Cfg::VOFF + vnext;  // (the V ring has three slots: its offset is a run-time value, uniform)
    const uint32_t tt3 = (uint32_t)__builtin_amdgcn_readfirstlane(t + 3);
    f32x16 Sn[2], DPn[2];
    [[maybe_unused]] int lm[2] = {0, 0};
    u32x4 DSn[2][2], kf[KK], vf[KK], XR[2][2];
    u32x2 th[2][2], tn[2][2], bf[2][4];  // bf[qb][jj]: bias fragments (each read serves both halves of its operand)
    float X[32], Pv[32], Dv[32];
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const uint32_t xo = xpar;
    auto stA_ = [&]<int E>() { X[E] = asm_fma(S[E >> 4][E & 15], c2, nL2[E >> 4]); };
    auto stB_ = [&]<int E>() { Pv[E] = asm_exp2(X[E]); };
    auto stC_ = [&]<int E>() { Dv[E] = asm_mul(Pv[E], DP[E >> 4][E & 15]); };
    auto stD_ = [&]<int E0>() {
      constexpr int qb = E0 >> 4, r0 = E0 & 15;
      DSn[qb][r0 >> 3][(r0 & 7) >> 1] = asm_cvt_pk<BF16>(Dv[E0], Dv[E0 + 1]);
    };
    auto bfrag = [&](const int qb, const int jj) { return u32x4{bf[qb][jj][0], bf[qb][jj][1], bf[qb][jj][0], bf[qb][jj][1]}; };
    auto bfrag1 = [&](const int qb, const int t2) { return u32x4{bf[qb][2 * t2][0], bf[qb][2 * t2][1], bf[qb][2 * t2 + 1][0], bf[qb][2 * t2 + 1][1]}; };  // ONE: 16 rows
    static_for<NG>([&](auto gi) {
      constexpr int g = decltype(gi)::value;
      // ---- MFMA ----
      if constexpr (g < 8) {
        constexpr int p = g >> 1, t2 = p >> 1, db = p & 1, qb = g & 1;
        u32x4 fr;
        if constexpr (t2 == 0) fr = TRK[db];
        else fr = u32x4{th[db][0][0], th[db][0][1], th[db][1][0], th[db][1][1]};
        mfma_acc_agpr<BF16>(dq[qb][db], fr, DSB[qb][t2]);
      } else if constexpr (ONE && g < 12) {
        // (ONE: gaps 8..11 one k-step each, like the no-bias iteration; gaps 12..15 two MFMAs: k-steps 2, 3, then the two 16-row bias operands)
        constexpr int kk = (g - 8) >> 1, qb = g & 1;
        if constexpr (kk == 0 && !MK) Sn[qb] = mfma32<BF16>(kf[kk], qf[qb][kk], zero16);
        else Sn[qb] = mfma32<BF16>(kf[kk], qf[qb][kk], Sn[qb]);
      } else if constexpr (ONE && g < 16) {
        if constexpr (g < 14) Sn[0] = mfma32<BF16>(kf[g - 10], qf[0][g - 10], Sn[0]);
        else Sn[0] = mfma32<BF16>(bfrag1(0, g - 14), selB[g - 14], Sn[0]);
      } else if constexpr (g < 12) {
        // (gaps 8..15: two MFMAs each -- query block 0 here, query block 1 behind the gap's LDS section)
        if constexpr (g == 8 && !MK) Sn[0] = mfma32<BF16>(kf[0], qf[0][0], zero16);
        else Sn[0] = mfma32<BF16>(kf[g - 8], qf[0][g - 8], Sn[0]);  // (MK: Sn holds the mask, formed in gaps 5..7)
      } else if constexpr (g < 16) {
        Sn[0] = mfma32<BF16>(bfrag(0, g - 12), selB[g - 12], Sn[0]);
      } else {
        constexpr int kk = (g - 16) >> 1, qb = g & 1;
        if constexpr (kk == 0) DPn[qb] = mfma32<BF16>(vf[kk], dof[qb][kk], nd16[qb]);  // (nd16 lives for the whole loop: no WAR window)
        else DPn[qb] = mfma32<BF16>(vf[kk], dof[qb][kk], DPn[qb]);
      }
      __builtin_amdgcn_sched_barrier(0);  // (the MFMA opens its gap)
      // ---- the batch sum of step i-1 on the matrix pipe ----
      if constexpr (g >= 7 && g <= 10 && !(FAT5_QDB_ABL & 2)) {
        constexpr int m = (g - 7) >> 1, ac = (g - 7) & 1;
        acc[ac] = mfma16<BF16>(sel16, XR[m][ac], acc[ac]);
      }
      // ---- barrier + DMA ----
      if constexpr (g == 4) wait_step();
      if constexpr (g >= 5 && g < 5 + Cfg::PIECES) dma_piece(tt3, g - 5);  // (step i+3 into the slots the barrier has released, one piece per gap)
      // ---- LDS ----
      if constexpr (g < 4) {
        constexpr int db = g >> 1, half = g & 1;
        th[db][half] = lds_rd_tr_half(trA[half][db] + o_prev + (uint32_t)(16 * 2 * D));
        // this wave's dS of step i-1 into the exchange buffer (query block g >> 1, chunk g & 1)
        if constexpr (!(FAT5_QDB_ABL & 2)) lds_wr128(xw[g & 1] + xo + (uint32_t)((g >> 1) * 2048), DSB[g >> 1][g & 1]);
      } else if constexpr (g < 8) {
        kf[g - 4] = lds_rd128(rmA[g - 4] + o_next);
        if constexpr ((g == 5 || g == 6) && !(FAT5_QDB_ABL & 2)) {
          XR[g - 5][0] = lds_rd128(xr[0] + xo + (uint32_t)((g - 5) * 2 * BT));
          XR[g - 5][1] = lds_rd128(xr[1] + xo + (uint32_t)((g - 5) * 2 * BT));
        }
      } else if constexpr (g < 12) {
        // bias fragments of step i+1 (its tile is visible since E(i)): row groups jj = g - 8 of both query blocks
        constexpr int jj = g - 8;
        constexpr uint32_t o = b_next + (uint32_t)((jj >> 1) * 1024);
        bf[0][jj] = lds_rd_tr_half(btA[jj & 1] + o);
        bf[1][jj] = lds_rd_tr_half(btA[jj & 1] + o + 2048u);
      } else if constexpr (g < 16) {
        vf[g - 12] = lds_rd128(rmA[g - 12] + v_next);
      } else if constexpr (g >= NG - 4) {
        constexpr int db = (g - (NG - 4)) >> 1, half = g & 1;
        tn[db][half] = lds_rd_tr_half(trA[half][db] + o_cur);
      }
      // ---- the gap's second MFMA (query block 1) ----
      if constexpr (!ONE && g >= 8 && g < 16) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g == 8 && !MK) Sn[1] = mfma32<BF16>(kf[0], qf[1][0], zero16);
        else if constexpr (g < 12) Sn[1] = mfma32<BF16>(kf[g - 8], qf[1][g - 8], Sn[1]);
        else Sn[1] = mfma32<BF16>(bfrag(1, g - 12), selB[g - 12], Sn[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (ONE && g >= 12 && g < 16) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g < 14) Sn[1] = mfma32<BF16>(kf[g - 10], qf[1][g - 10], Sn[1]);
        else Sn[1] = mfma32<BF16>(bfrag1(1, g - 14), selB[g - 14], Sn[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // (the bias words are MFMA operands: clamped in their packed form two gaps after their reads, two gaps ahead of their MFMAs -- -inf * 0 would be NaN: attn_common.h)
      if constexpr (g >= 10 && g < 14) {
        bf[0][g - 10] = bias_clamp_frag(bf[0][g - 10], blim2);
        bf[1][g - 10] = bias_clamp_frag(bf[1][g - 10], blim2);
      }
      // ---- the dbias tile of step i-1 leaves ----
      if constexpr (g == 14 && !(FAT5_QDB_ABL & 1)) x_store(acc, row_ok ? dvo : 0x80000000u, t >= 1 ? (uint32_t)((t - 1) * 32 * ESZ) : dbrec);
      // ---- VALU ----
      if constexpr (MK) {  // the masked C operands of step i+1's scores: query block 0 in gaps 5, 6 (first MFMA: gap 8), query block 1 in gaps 6, 7 (gap 9)
        if constexpr (g == 5) {
          lm[0] = lim0 - (t + 1) * 32;
          lm[1] = lm[0] + 32;
        }
        if constexpr (g >= 5 && g <= 7) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if ((g == 5 && r < 8) || (g == 6 && r >= 8)) Sn[0][r] = (qdb_key(r, 0) > lm[0]) ? ninf_c : 0.f;
            if ((g == 6 && r < 8) || (g == 7 && r >= 8)) Sn[1][r] = (qdb_key(r, 0) > lm[1]) ? ninf_c : 0.f;
          }
        }
      }
      {
        constexpr auto lo = [](int gg) { return gg <= 0 ? 0 : (gg >= NG - 3 ? 32 : (32 * gg) / (NG - 3)); };
        static_for<lo(g - 2) - lo(g - 3)>([&](auto ei) {
          constexpr int e = lo(g - 3) + decltype(ei)::value;
          if constexpr ((e & 1) == 1) stD_.template operator()<e - 1>();
        });
        static_for<lo(g - 1) - lo(g - 2)>([&](auto ei) { stC_.template operator()<lo(g - 2) + decltype(ei)::value>(); });
        static_for<lo(g) - lo(g - 1)>([&](auto ei) { stB_.template operator()<lo(g - 1) + decltype(ei)::value>(); });
        static_for<lo(g + 1) - lo(g)>([&](auto ei) { stA_.template operator()<lo(g) + decltype(ei)::value>(); });
        if constexpr (g == NG - 1) {
          static_for<32 - lo(NG - 1)>([&](auto ei) { stB_.template operator()<lo(NG - 1) + decltype(ei)::value>(); });
          static_for<32 - lo(NG - 2)>([&](auto ei) { stC_.template operator()<lo(NG - 2) + decltype(ei)::value>(); });
          static_for<32 - lo(NG - 3)>([&](auto ei) {
            constexpr int e = lo(NG - 3) + decltype(ei)::value;
            if constexpr ((e & 1) == 1) stD_.template operator()<e - 1>();
          });
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      S[qb] = Sn[qb];
      DP[qb] = DPn[qb];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) DSB[qb][t2] = DSn[qb][t2];
    }
#pragma unroll
    for (int db = 0; db < DB; ++db) TRK[db] = u32x4{tn[db][0][0], tn[db][0][1], tn[db][1][0], tn[db][1][1]};
    xpar ^= (uint32_t)Cfg::XB;
    vnext = vnext == (uint32_t)(2 * IMG) ? 0u : vnext + (uint32_t)IMG;
  };

  if (nt > 0) {
    score_step(0u, 0u, 0u, 0, S, DP);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) DSB[qb][t2] = zero4;
    TRK[0] = zero4;
    TRK[1] = zero4;
    // every key of step u exists (no key tail)? / is visible to all 64 rows?  The same for every wave: they share their rows
    auto full = [&](const int u) { return u * 32 + 32 <= N; };
    auto vis = [&](const int u) { return !a.causal || u * 32 + 31 <= qw0 + P; };
    int t = 0;
    while (t < nt) {
      // steady state: four steps (K / bias ring slots 0..3) per trip, straight-line.  Iteration u forms the scores of step u + 1: while step t + 4 is
      // visible to every row no mask is needed; after that the trips carry the mask in the score MFMAs' C operand (visibility is monotone)
      while ((t & 3) == 0 && t + 4 <= nt && full(t + 3) && vis(t + 4)) {
        static_for<4>([&](auto si) { fast_iter.template operator()<decltype(si)::value, false>(t + decltype(si)::value); });
        t += 4;
      }
      while ((t & 3) == 0 && t + 4 <= nt && full(t + 3) && !vis(t + 4)) {
        static_for<4>([&](auto si) { fast_iter.template operator()<decltype(si)::value, true>(t + decltype(si)::value); });
        t += 4;
      }
      // (a key tail, and whatever does not fill an aligned trip)
      if (t < nt && !((t & 3) == 0 && t + 4 <= nt && full(t + 3))) {
        generic_iter(t);
        ++t;
      }
    }
    // drain: the last step's exchange and products
    x_write(xpar);
    product_step((uint32_t)(((nt - 1) & 3) * IMG));
    wait_dma_all();  // (the requests past the end: nothing may land in the ring once dQ goes through it)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    x_reduce_store(xpar, nt - 1, (nt - 1) * 32 + 32 <= N);
  }
  // causal: the key steps this row block never visits lie above the diagonal -- zeros by definition (reference :153, :160): written here, 16 rows per wave,
  // instead of a memset of the whole (H, M, N) tensor in front of the launch (403 MB at S = 4096).  (PARTIAL: the slab reduction knows the mask.)
  if constexpr (!PARTIAL) {
    if (a.causal) {
      for (int st = nt; st * 32 < N; ++st) {
        const bool ok = row_ok && st * 32 + 8 * (l >> 4) < N;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, dbrs, ok ? dvo : 0x80000000u, (uint32_t)(st * 32 * ESZ), 0);
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (asm MFMA -> accumulator reads below: see mfma_acc_agpr)
  wait_dma_all();  // (nothing may land in the ring once dQ goes through it)

  // dQ through the wave's (free) ring area: 8-byte pieces into a swizzled row-major image, out again as whole rows
  if (bvalid) {
    const float scale = a.scale;
    char* img = smem + w * Cfg::WRING;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int row = 32 * qb + lq;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 wv;
          wv[0] = pack2<BF16>(dq[qb][db][4 * g + 0] * scale, dq[qb][db][4 * g + 1] * scale);
          wv[1] = pack2<BF16>(dq[qb][db][4 * g + 2] * scale, dq[qb][db][4 * g + 3] * scale);
          *reinterpret_cast<u32x2*>(img + rm_off<D>(row, 4 * db + g) + 8 * hi) = wv;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 8 * i + (l >> 3), slot = l & 7;
      const u32x4 v4 = *reinterpret_cast<const u32x4*>(img + row * (2 * D) + slot * 16);
      if (qw0 + row < M) *reinterpret_cast<u32x4*>(dqb_ + (int64_t)(qw0 + row) * a.dqs[2] + ((slot ^ swz<D>(row)) << 3)) = v4;
    }
  }
}

template <int D, bool BF16, bool PARTIAL, bool ONE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void attn_bwd_qdb64_kernel(const AttnArgs a, void* dbias_out) {
  attn_bwd_qdb64_body<D, BF16, PARTIAL, ONE>(a, dbias_out, (int)blockIdx.x, true);
}

// The whole dense-bias backward in ONE launch (round 5, second half; the dense counterpart of attn_bwd_fused64_kernel): workgroups [0, n_kv_blocks) -- a multiple
// of eight, the padding exits -- run the 256-key dense dK/dV body (attn_bwd64.h), the others the dQ + dBias body above.  Separately the two launches of a short
// sequence leave most of the chip idle twice ((4,12,512): 96 workgroups each on 256 CUs) and those of a mid one each end in a half-empty round ((4,12,2048):
// 384 + 384 workgroups at one per CU).  The dK/dV half cannot wait for the dQ half's row statistics: bwd_stat2_kernel writes them ahead of the launch
// (the reference's _bwd_preprocess, flash_attention_v2_bias.py:516-556) -- the same values bit for bit, so a call that runs the stages as separate launches
// (a unit range, one stage only) returns identical results.
template <int D, bool BF16, bool PARTIAL, bool ONE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void attn_bwd_dfused64_kernel(const AttnArgs a, void* dbias_out) {
  const int nkv8 = (a.n_kv_blocks + 7) & ~7;
  if ((int)blockIdx.x < nkv8) {
    if ((int)blockIdx.x >= a.n_kv_blocks) return;
    int b, h, nblk;
    decode_unit(a, blockIdx.x, a.n_nblk, b, h, nblk, (FAT5_CAUSAL_ORDER && a.causal) ? 2 : 0);
    attn_bwd_kv64_body<D, BF16, FAT5_BIAS_DENSE, false, false, ONE>(a, b, h, nblk, nblk, false);
  } else {
    attn_bwd_qdb64_body<D, BF16, PARTIAL, ONE>(a, dbias_out, (int)blockIdx.x - nkv8, false);
  }
}

// Row statistics of the backward ahead of a launch whose dK/dV workgroups run beside the dQ ones: stat2[(b, h), step][0..31] = -L / scale (a dead or padded row: the
// value whose probability is zero), [32..63] = -delta = -rowsum(o * do) (reference _bwd_preprocess, :516-556).  One wave per 32-row step, lane = (row, half of the
// columns): the products of a row are summed in exactly the order of the dQ bodies' prologues (16-column groups ascending, low word first, the two halves last) --
// the same bits.  a.delta gets the plain delta as there.
template <int D, bool BF16>
__global__ __launch_bounds__(256) void bwd_stat2_kernel(const AttnArgs a) {
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lq = l & 31, hi = l >> 5;
  const int nst = (a.M + 31) / 32;
  const int64_t gs = (int64_t)blockIdx.x * 4 + w;  // (b, h, step), step minor
  if (gs >= (int64_t)a.B * a.H * nst) return;
  const int bh = (int)(gs / nst), st_i = (int)(gs - (int64_t)bh * nst);
  const int b = bh / a.H, h = bh - b * a.H;
  const int qrow = 32 * st_i + lq, qr = min(qrow, a.M - 1);
  const uint16_t* orow = a.o + (int64_t)b * a.os[0] + (int64_t)h * a.os[1] + (int64_t)qr * a.os[2];
  const uint16_t* dorow = a.dout + (int64_t)b * a.dos[0] + (int64_t)h * a.dos[1] + (int64_t)qr * a.dos[2];
  u32x4 of[D / 16], dof[D / 16];
#pragma unroll
  for (int kk = 0; kk < D / 16; ++kk) {
    of[kk] = *reinterpret_cast<const u32x4*>(orow + 16 * kk + 8 * hi);
    dof[kk] = *reinterpret_cast<const u32x4*>(dorow + 16 * kk + 8 * hi);
  }
  const float Lq = a.lse[(int64_t)bh * a.M + qr];
  float dsum = 0.f;
#pragma unroll
  for (int kk = 0; kk < D / 16; ++kk)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dsum = fmaf(cvt_lo<BF16>(of[kk][j]), cvt_lo<BF16>(dof[kk][j]), dsum);
      dsum = fmaf(cvt_hi<BF16>(of[kk][j]), cvt_hi<BF16>(dof[kk][j]), dsum);
    }
  const float delta = pair_sum(dsum);
  if (hi == 0) {
    if (a.delta && qrow < a.M) a.delta[(int64_t)bh * a.M + qrow] = delta;
    float* st = a.stat2 + ((int64_t)bh * nst + st_i) * 64 + lq;
    const bool live = qrow < a.M && !(Lq < kDeadRowLse);
    st[0] = live ? -Lq / a.scale : (a.scale > 0.f ? -INFINITY : INFINITY);
    st[32] = qrow < a.M ? -delta : 0.f;
  }
}

// dbias (H, M, N) in the bias dtype = sum over the `ngrp` fp32 slabs the PARTIAL kernel wrote, in slab order.  Causal: tiles the kernel never visits
// (key step 32 kb of row block 64 rb with 32 kb >= min(N, 64 rb + 64 + P)) hold garbage and are zeros by definition.
template <bool BF16>
__global__ __launch_bounds__(256) void dbias_partial_reduce_kernel(const float* __restrict__ part, uint16_t* __restrict__ out, int ngrp, int64_t HMN,
                                                                    int M, int N, int causal, int P) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // 8 consecutive keys of one row (N % 8 == 0)
  if (c * 8 >= HMN) return;
  const int64_t e0 = c * 8;
  bool vis = true;
  if (causal) {
    const int64_t row = (e0 / N) % M;
    const int n0 = (int)(e0 % N);
    const int n_end = (int)min((int64_t)N, (row / 64) * 64 + 64 + P);
    vis = (n0 / 32) * 32 < n_end;
  }
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (vis) {
    for (int g = 0; g < ngrp; ++g) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(part + (int64_t)g * HMN + e0), x1 = *reinterpret_cast<const f32x4*>(part + (int64_t)g * HMN + e0 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] += x0[j];
        acc[4 + j] += x1[j];
      }
    }
  }
  u32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = pack2<BF16>(acc[2 * j], acc[2 * j + 1]);
  *reinterpret_cast<u32x4*>(out + e0) = o;
}

}  // namespace fat5
