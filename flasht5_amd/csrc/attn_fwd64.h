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
//    probabilities (one FMA + v_exp_f32 per element, packed to 16 bit), the matrix pipe runs O^T += V^T P^T of block i-1
//    and S^T = K Q^T of block i+1 -- 16 MFMAs that do not depend on this block's VALU work.
//  * K and V live in separate 4-slot LDS rings filled global -> LDS by DMA three tiles (K) / two tiles (V) ahead;
//    one barrier per 64-key tile.
//  * bf16, round 3: ONE pipelined sweep over every tile all of whose keys are visible to all rows of the workgroup -- far
//    tiles (constant bias) and the tiles of the T5 band alike -- with NO running row maximum.  FlashAttention's reference
//    point m is free: exp2(x - m) with ANY m is exact as long as nothing overflows or is flushed, and bf16 P + fp32
//    accumulators share fp32's exponent range.  The sweep uses m = 0 for every row: p = exp2(s * c2 + bias), no row maximum, no
//    lane exchange, no rescale, no baseline tiles.  At the end a row whose sum left [2^-40, 2^100) (log2-scores above ~ +100 or
//    all below ~ -40: |s * sm_scale| beyond ~28 .. 69 nats) sends its workgroup through the exact running-maximum pass again
//    (second pass, unpipelined).  Band tiles read their per-element bias from the padded table copies in LDS (attn_common.h)
//    inside the pipelined stream: the value is the FMA's addend (no extra VALU op), fetched three MFMA gaps ahead.
//    Tiles with masked keys (N tail, causal diagonal) run unpipelined with the same reference point.  fp16 P overflows at
//    2^16: its sweep uses the row maxima of the first tile as reference point (round 4; see OPT below).
//  * KSPLIT (mid sequence lengths): the waves of a workgroup pair up on 64 query rows, each taking ONE 32-key block of every
//    staged tile, and merge (m, l, O) through LDS at the end: 128-row workgroups of half-length waves.  With 64 rows x N keys
//    per wave (4,12,2048,64) is 1536 waves on 1024 SIMDs at two waves per SIMD: half the SIMDs carry two waves for the whole
//    kernel, the other half one.  3072 half-length waves are 3 per SIMD: two side by side, then one alone at the full issue rate.
#pragma once
#include "attn_common.h"
#include <utility>
#include "attn_fwd.h"

#ifndef FAT5_TRACE
#define FAT5_TRACE 0
#endif
#if FAT5_TRACE  // developer build: thread 0 keeps s_memtime stamps of the phases and leaves them in the first O row of its workgroup (tools/trace64.py --fwd)
#define FAT5_FSTAMP(slot) do { if (threadIdx.x == 0) fstamp[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define FAT5_FSTAMP(slot) do { } while (0)
#endif
#ifndef FAT5_FWD_ABL
#define FAT5_FWD_ABL 0  // developer ablations of the pipelined block (timing / counters only, wrong results unless noted): 1 no row sums, 2 row sums by v_add_f32 (correct results), 4 v_mov for v_exp, 8 no conversion to 16 bit (the words stay zero)
#endif
#ifndef FAT5_MIX_PRIO
#define FAT5_MIX_PRIO 0  // mixed launch: 1 = the key-split (half-length) waves run at raised priority (s_setprio 3), 2 = the full-length ones.  Measured at (4,12,2048): 57.3 / 55.0 us against 55.2 without -- no gain, off
#endif
#ifndef FAT5_FWD_NS_KSPLIT
#define FAT5_FWD_NS_KSPLIT 4
#endif
namespace fat5 {

// D3 (head_dim 128 with a dense bias, round 5): three ring slots per operand (96 KB) so that the two-tile bias ring (64 KB) fits the CU's 160 KB; the exact-pass
// flag then lives in the first word of the bias ring (free once the sweep is over: see the body)
template <int D, bool KSPLIT = false, bool D3 = false>
struct Fwd64Cfg {
  static constexpr int NW = 4, BM = KSPLIT ? 32 * NW : 64 * NW, BN = 64, NT = 64 * NW, NS = D3 ? 3 : (KSPLIT ? FAT5_FWD_NS_KSPLIT : 4);  // NS ring slots per operand
  static constexpr int TILE = rm_bytes<D, BN>();
  static constexpr int KOFF = 0, VOFF = NS * TILE, FLAG = 2 * NS * TILE, TAB = FLAG + (D3 ? 0 : 16);
  static constexpr int MERGE = 32 * 64 * 4 + 1024;  // KSPLIT: per wave, one query block's O^T (32 registers x 64 lanes) + (m, l) -- inside the rings
  static_assert(NW * MERGE <= 2 * NS * TILE, "merge area lives in the K / V rings");
  // dense bias (round 4): the workgroup's (BM rows x 64 keys) 16-bit bias tile of every key tile travels global -> LDS like K / V,
  // into a ring of two tiles behind the K / V rings (source-swizzled like a D = 64 image: BiasTileReader)
  static constexpr int BIASB = BM * BN * 2, NSB = 2;
  static size_t smem(int R, int bias_mode) {
    return TAB + (bias_mode == FAT5_BIAS_RPE1D ? rpe_table_bytes(R) + 16 : 0) + (bias_mode == FAT5_BIAS_DENSE ? (size_t)NSB * BIASB : 0);
  }
};

// LDS access by integer address (base VGPR + compile-time constant -> the constant lands in the instruction's offset field;
// pointer arithmetic on the dynamic-LDS symbol costs a v_add per access instead)
FAT5_DEV u32x4 lds_rd128(uint32_t addr) {
  typedef const u32x4 __attribute__((address_space(3))) * p_t;
  return *(p_t)(uintptr_t)addr;
}
FAT5_DEV u32x2 lds_rd64(uint32_t addr) {
  typedef const u32x2 __attribute__((address_space(3))) * p_t;
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
FAT5_DEV float asm_mulf(float a, float b) {
  float r;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
FAT5_DEV float asm_shl16(uint32_t w) {  // low bf16 of a packed pair -> fp32
  float r;
  asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(r) : "v"(w));
  return r;
}
FAT5_DEV float asm_and_hi(uint32_t w) {  // high bf16 of a packed pair -> fp32
  float r;
  asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(r) : "v"(w));
  return r;
}
template <bool BF16>
FAT5_DEV uint32_t asm_cvt_pk(float a, float b) {
  uint32_t r;
  if constexpr (BF16) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));  // (gfx950: one instruction, round-to-nearest-even like the cast)
  return r;
}
// acc += A(16x32) . B(32x16), in place (asm: a builtin may pick a fresh destination, and no hazard padding exists between asm ops)
template <bool BF16>
FAT5_DEV void mfma16_acc(f32x4& acc, const u32x4 A, const u32x4 B) {
  if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B));
}
// acc += A . B with the accumulator tuple in AGPRs.  The dK^T / dV^T accumulators (128 registers) are touched by nothing but
// MFMAs: hipcc's VGPR-form MFMA selection would keep them in VGPRs and spill everything else through v_accvgpr moves.
// No hazard padding is generated for asm: same-accumulator MFMAs need none, A / B come from LDS reads (waitcnt is inserted
// for asm operands) or from VALU results that are many instructions old; the epilogue pads before it reads the tuples.
template <bool BF16>
FAT5_DEV void mfma_acc_agpr(f32x16& acc, const u32x4 A, const u32x4 B) {
  if constexpr (BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B));
}
FAT5_DEV u32x2 lds_rd_tr_half(uint32_t addr) {
  typedef s16x4_t __attribute__((address_space(3))) * p_t;
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((p_t)(uintptr_t)addr));
}
FAT5_DEV void wait_dma_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }


// SPREAD (head_dim 128): the ring requests of a tile leave one piece per MFMA gap instead of eight back to back in front of the tile.  Measured (us):
// (4,12,1024,128) -- 192 workgroups, one partial round, every CU in the same phase -- 32.4 vs 38.7; (16,12,1024) 106.9 vs 105.5, (4,12,8192) 1299 vs 1281:
// the launcher takes it for grids of at most one round.
template <int D, bool BF16, int BIAS, bool KSPLIT, bool SPREAD = false>
FAT5_DEV void attn_fwd64_body(const AttnArgs& a, const int b, const int h, const int m0) {
  static_assert(BIAS != FAT5_BIAS_DENSE || !KSPLIT, "dense bias: 256-row workgroups only");
  constexpr bool DENSE = BIAS == FAT5_BIAS_DENSE;
  // W1 (round 5, head_dim 128): one wave per SIMD -- O^T (128 registers per lane) lives in AGPRs and is touched by asm MFMAs only
  // (mfma_acc_agpr), the Q fragments sit in AGPRs as well; a pipelined block has 32 MFMA gaps; Q arrives and O leaves as whole rows through LDS
  constexpr bool W1 = D == 128;
  static_assert(!W1 || !KSPLIT, "head_dim 128: 256-row workgroups");
  constexpr bool D3 = W1 && DENSE;
  static_assert(!D3 || !SPREAD, "dense: the ring requests leave in front of the tile, behind the bias tile's (counted waits)");
  using Cfg = Fwd64Cfg<D, KSPLIT, D3>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT, TILE = Cfg::TILE, NS = Cfg::NS;
  constexpr int KK = D / 16, DB = D / 32;
  constexpr int NKB = KSPLIT ? 1 : 2;  // 32-key blocks of a tile that THIS wave works on
  [[maybe_unused]] long long fstamp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  FAT5_FSTAMP(0);
#if FAT5_TRACE
  fstamp[8] = (long long)__builtin_amdgcn_s_getreg(63492) | ((long long)__builtin_amdgcn_s_getreg(6164) << 32) | ((long long)(KSPLIT ? 1 : 0) << 40);  // HW_ID | XCC_ID | kind
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* sFlag = reinterpret_cast<int*>(smem + Cfg::FLAG);
  float* sT = reinterpret_cast<float*>(smem + Cfg::TAB) + kRpePad;  // (entry d of copy 0 at sT[d + R]; see attn_common.h)

  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, lq = l & 31, hi = l >> 5;
  const int rg = KSPLIT ? (w >> 1) : w;  // 64-row group of this wave
  const int kh = KSPLIT ? (w & 1) : 0;   // KSPLIT: the 32-key block of every tile this wave owns (wave-uniform)
  const int M = a.M, N = a.N;
  if (m0 >= M) return;
  [[maybe_unused]] RpeTableRegs tabr;  // (the bias table's first round of loads leaves before every other request of the prologue)
  if constexpr (BIAS == FAT5_BIAS_RPE1D) tabr = rpe_table_load_first(a.rpe1d + (int64_t)h * (2 * a.R + 1), a.R, tid, NT);
  const uint16_t* qb_ = a.q + (int64_t)b * a.qs[0] + (int64_t)h * a.qs[1];
  const uint16_t* kb_ = a.k + (int64_t)b * a.ks[0] + (int64_t)h * a.ks[1];
  const uint16_t* vb_ = a.v + (int64_t)b * a.vs[0] + (int64_t)h * a.vs[1];
  uint16_t* ob_ = a.o + (int64_t)b * a.os[0] + (int64_t)h * a.os[1];
  const int64_t lse_off = ((int64_t)b * a.H + h) * a.M;

  const int P = N - M;  // bottom-right causal offset
  // the causal mask carried by the bias table (attn_bwd64.h: -inf above the diagonal when it lies inside the band): diagonal tiles run the pipelined band block
  [[maybe_unused]] const bool ctab = BIAS == FAT5_BIAS_RPE1D && a.causal && P < a.R && P >= -a.R;
  int n_end = N;
  if (a.causal) n_end = min(N, m0 + BM + P);
  const int nt = n_end > 0 ? (n_end + BN - 1) / BN : 0;

  const int qrow0 = m0 + 64 * rg;  // first query row of this wave; block qb covers rows qrow0 + 32*qb ..+31
  // Q fragments (B operand of S^T = K Q^T): Q[q][16kk + 8hi + j].  KSPLIT (short sequences, round 4): the pair's 64 query rows arrive
  // as whole rows by LDS-DMA -- each wave of the pair four of the eight 1-KiB pieces -- in V ring slots 2, 3 (free until the first
  // iteration requests V tile 2) and are read back as fragments after the prologue's wait (rows past M arrive as zeros); the
  // 256-row form loads them straight from global (its prologue is < 1 % of a long sweep).
  u32x4 qf[2][KK];
  const uint32_t qimg = (uint32_t)(uintptr_t)smem + (uint32_t)(Cfg::VOFF + (NS - 2) * TILE + rg * 64 * 2 * D);
  if constexpr (KSPLIT) {
    static_assert(2 * TILE >= (Cfg::NW / 2) * 64 * 2 * D, "the Q images of the workgroup's 64-row groups fit the last two V slots");
    using SDma = DmaStage<D, 64, 64>;
    static_assert(SDma::PER == 8 && SDma::NV == 2, "eight 1-KiB pieces of 8 rows");
    SDma sq;
    sq.init(a.qs[2], l);
    const __amdgpu_buffer_rsrc_t qrs = make_rows_rsrc(qb_, a.qs[2], M, D);
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)qimg);
    const uint32_t r0 = (uint32_t)qrow0 * (uint32_t)a.qs[2] * 2u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pi = 4 * kh + i;  // (wave-uniform)
      dma16_asm(qrs, dst + (uint32_t)(pi * 1024), sq.voff[i % 2], r0 + sq.piece_step * (uint32_t)(pi / 2));
    }
  } else if constexpr (W1) {
    // the wave's 64 rows through the K ring (free until stage_first): 16 pieces of 4 rows, read back as fragments at once
    using SDma = DmaStage<D, 64, 64>;
    static_assert(SDma::PER == 16 && SDma::NV == 4, "sixteen 1-KiB pieces of 4 rows");
    static_assert(Cfg::NW * 64 * 2 * D <= 2 * NS * TILE, "the Q images fit the K / V rings (contiguous)");
    SDma sq;
    sq.init(a.qs[2], l);
    const __amdgpu_buffer_rsrc_t qrs = make_rows_rsrc(qb_, a.qs[2], M, D);
    const uint32_t qi = (uint32_t)(uintptr_t)smem + (uint32_t)(w * 64 * 2 * D);
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)qi);
    const uint32_t r0 = (uint32_t)qrow0 * (uint32_t)a.qs[2] * 2u;
#pragma unroll
    for (int i = 0; i < 16; ++i) dma16_asm(qrs, dst + (uint32_t)(i * 1024), sq.voff[i % 4], r0 + sq.piece_step * (uint32_t)(i / 4));
    FragAddr<D> fq;
    fq.init(l);
    wait_dma_all();
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) qf[qb][kk] = lds_rd128(qi + (uint32_t)(fq.rm[kk] + qb * 32 * 2 * D));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // (stage_first overwrites the images)
  } else {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qr = min(qrow0 + 32 * qb + lq, M - 1);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
        qf[qb][kk] = *reinterpret_cast<const u32x4*>(qb_ + (int64_t)qr * a.qs[2] + 16 * kk + 8 * hi);
    }
  }
  const float* sTa = sT;  // this lane's aligned copy of the RPE table (rows 32 apart share the alignment)
  if constexpr (BIAS == FAT5_BIAS_RPE1D) sTa = sT + ((a.R - (qrow0 + lq)) & 3) * rpe_n1p(a.R);

  FragAddr<D> fa;
  fa.init(l);

  // fp16 (round 4): the same sweep with a per-row reference point -- the row maxima of the first tile (a scores-only pre-pass) --
  // instead of 0: fp16 probabilities overflow at 2^16, so a later score more than ~11 nats above the first tile's maximum of its
  // row sends the workgroup through the exact pass (the check the bf16 sweep has for 2^100); what lies far BELOW that maximum
  // is flushed, as negligible beside the row's own first tile as it is in the exact algorithm.  (dense bias: bf16 only)
  constexpr bool OPT = FAT5_OPTIMISTIC && (BF16 || !DENSE);
  // reference point of the sweep: 0 for every row (bf16 with the T5 table: the band blocks would pay an add per element; T5 logits are small), else the rows'
  // first-tile maxima -- always in fp16, in bf16 when the launcher asks for it (large sm_scale * sqrt(D)): the addend of the FMA either way, no extra instruction
  constexpr bool REF0 = BF16 && BIAS == FAT5_BIAS_RPE1D;
  const bool ref_first = !BF16 || (!REF0 && a.ref_first != 0);
  if (OPT && !D3 && tid == 0) *sFlag = 0;

  f32x16 oacc[2][DB];
  float m_run[2];
  float l_run[2][2];
  // Row sums of the pipelined blocks: lacc[qb] += SEL . P^T words (the B operands of the P.V products, i.e. the ROUNDED
  // probabilities) with one 16x16x32 MFMA per four packed words instead of 16 v_add_f32. As a 32x16 B operand the words of lanes
  // n, n+16, n+32, n+48 form column n; SEL row 4G has ones for the k-groups of parity G%2, every other row is zero, so register 0 of
  // lane l ends up with the sum over both key halves (lanes l%32 and l%32 + 32) of ITS query column; registers 1..3 stay zero.
  f32x4 lacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  [[maybe_unused]] float lv[2] = {0.f, 0.f};  // (FAT5_FWD_ABL & 2)
  auto abl_exp2 = [](float x) {
    if constexpr ((FAT5_FWD_ABL & 4) != 0) {
      float r;
      asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(x));
      return r;
    } else {
      return asm_exp2(x);
    }
  };
  u32x4 sel;
  {
    const int i16 = l & 15;
    const uint32_t one2 = pack2<BF16>(1.f, 1.f);
    const uint32_t wv = ((i16 & 3) == 0 && (((l >> 4) ^ (i16 >> 2)) & 1) == 0) ? one2 : 0u;
    sel = u32x4{wv, wv, wv, wv};
    asm volatile("" : "+v"(sel));  // (resident: never re-materialised by VALU moves right in front of an asm consumer)
  }

  DmaStage<D, BN, NT> kst, vst;
  kst.init(a.ks[2], tid);
  vst.init(a.vs[2], tid);
  const __amdgpu_buffer_rsrc_t krs = make_rows_rsrc(kb_, a.ks[2], N, D);
  const __amdgpu_buffer_rsrc_t vrs = make_rows_rsrc(vb_, a.vs[2], N, D);
  const uint32_t kstride_b = (uint32_t)a.ks[2] * 2u, vstride_b = (uint32_t)a.vs[2] * 2u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)w * 1024u);
  using Dma = DmaStage<D, BN, NT>;
  static_assert(Dma::NV == 1, "one voffset per lane");
  auto dma_tile = [&](const Dma& st, __amdgpu_buffer_rsrc_t rs, uint32_t tile_off, uint32_t lds_tile) {
#pragma unroll
    for (int i = 0; i < Dma::PER; ++i) dma16_asm(rs, wave_lds + lds_tile + (uint32_t)(NT * 16 * i), st.voff[0], tile_off + st.piece_step * i);
  };
  auto dma_k = [&](int t, int slot) { dma_tile(kst, krs, (uint32_t)(t * BN) * kstride_b, (uint32_t)(Cfg::KOFF + slot * TILE)); };
  auto dma_v = [&](int t, int slot) { dma_tile(vst, vrs, (uint32_t)(t * BN) * vstride_b, (uint32_t)(Cfg::VOFF + slot * TILE)); };
  // dense bias: tile t (rows m0 .. m0 + BM - 1, keys 64 t ..) into bias slot t & 1.  Every wave fetches the rows IT reads (its 64 query
  // rows: eight 1-KiB pieces of 8 rows), so its own counted vmcnt covers them -- the first groups of the next tile are read before
  // the iteration's closing barrier.  Rows past M and bytes past the last row's N keys read as zeros; key columns past N inside earlier
  // rows read the next row's values (finite): tiles with such keys run the masked, unpipelined pass.
  using BDma = DmaStage<BN, 64, 64, true>;
  BDma bst;
  constexpr int BOFF = Cfg::TAB;  // (the table region: unused in dense mode)
  __amdgpu_buffer_rsrc_t brs = krs;
  if constexpr (DENSE) {
    static_assert(BDma::NV == 2 && BDma::PER == 8, "eight pieces of 8 rows, two swizzle phases");
    bst.init(a.bs[2], l);
    brs = make_rows_rsrc(a.bias + (int64_t)b * a.bs[0] + (int64_t)h * a.bs[1] + (int64_t)(m0 + 64 * rg) * a.bs[2], a.bs[2], M - m0 - 64 * rg, N);
  }
  const uint32_t bias_lds = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(Cfg::TAB + rg * 64 * BN * 2));
  auto dma_b = [&](int t) {
    if constexpr (DENSE) {
      const uint32_t base = bias_lds + (uint32_t)((t & 1) * Cfg::BIASB);
#pragma unroll
      for (int i = 0; i < BDma::PER; ++i) dma16_asm(brs, base + (uint32_t)(1024 * i), bst.voff[i % 2], (uint32_t)(t * BN) * 2u + bst.piece_step * (i / 2));
    }
  };
  // Ring protocol.  Tile t lives in slot t % NS of both rings.  Iteration t (between two barriers) may read K(t), K(t+1) and
  // V(t); it starts by issuing K(t+NS-1) and V(t+NS-2) into the slots of K(t-1) / V(t-2), whose last readers are behind the
  // barrier that ended iteration t-1.  Before its closing barrier every wave waits for its own pieces of K(t+2) and V(t+1)
  // -- everything but the requests issued in THIS iteration (counted vmcnt: LDS-DMA requests retire in order), so a
  // request has a full iteration more than it needs to land.
  // Dense: the bias tile of t + 1 is requested FIRST in iteration t (into the slot of tile t - 1), so that the counted wait at the end
  // of the iteration -- everything but this iteration's K / V requests -- covers it: it is read from the first block of iteration t + 1.
  auto stage_first = [&]() {
    if (nt > 0) dma_b(0);
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
      if (i < nt) dma_k(i, i);
#pragma unroll
    for (int i = 0; i < NS - 2; ++i)
      if (i < nt) dma_v(i, i);
  };
  auto begin_iter = [&](int t, int slot, [[maybe_unused]] bool in_pipe = false) {
    const int sk = slot == 0 ? NS - 1 : slot - 1;                    // (t + NS - 1) % NS
    const int sv = slot <= 1 ? slot + NS - 2 : slot - 2;             // (t + NS - 2) % NS
    if (t + 1 < nt) dma_b(t + 1);
    if constexpr ((FAT5_FWD_ABL & 16) != 0) return;
    if constexpr (W1 && SPREAD) {
      if (!in_pipe) {
        if (t + NS - 1 < nt) dma_k(t + NS - 1, sk);
        if (t + NS - 2 < nt) dma_v(t + NS - 2, sv);
      }
      return;
    }
    if (t + NS - 1 < nt) dma_k(t + NS - 1, sk);
    if (t + NS - 2 < nt) dma_v(t + NS - 2, sv);
  };
  // everything but the K / V requests of iteration t has landed (this wave's pieces): the bias tile of t + 1 among it.  (Round 5: counted by what the iteration did
  // issue -- the fixed count of round 4 let the last tiles' bias reads run ahead of their DMA)
  auto bias_wait = [&](int t) {
    constexpr int PER = Dma::PER;
    if (t + NS - 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else if (t + NS - 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else wait_dma_all();
  };
  // THREE slots (D3): the requests of iteration t are K(t+2) and V(t+1) -- both are read by iteration t + 1 (see above: K(t'), K(t'+1), V(t') with t' = t + 1), by waves
  // other than the ones that fetched the pieces, so they have to be complete in front of THIS barrier: no request stays in flight across it.  (Found by the race screen,
  // tools/stress.py, behind a backward that leaves the caches cold: with the counted wait of the four-slot ring a wave read pieces another wave had requested one
  // iteration earlier and not yet waited for -- (16,12,1024,128) dense: 3 wrong forwards in 60.)  The requests still have the whole iteration to land.
  auto end_iter = [&](int t) {
    constexpr int PER = Dma::PER;
    if constexpr (NS == 3) wait_dma_all();
    else if (t + NS - 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else if (t + NS - 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else wait_dma_all();
    __syncthreads();
  };
  FAT5_FSTAMP(1);
  stage_first();
  if constexpr (BIAS == FAT5_BIAS_RPE1D) rpe_table_fill_rest(sT - kRpePad, a.rpe1d + (int64_t)h * (2 * a.R + 1), a.R, tid, NT, tabr, ctab ? P : 0x7fffffff);
  FAT5_FSTAMP(2);
  wait_dma_all();
  __syncthreads();
  FAT5_FSTAMP(3);
  if constexpr (KSPLIT) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) qf[qb][kk] = lds_rd128(qimg + (uint32_t)(fa.rm[kk] + qb * 32 * 2 * D));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // (the first iteration requests V tile 2 into the slot the Q images sit in)
  }
  // per-lane LDS addresses of the fragment reads (ring base folded in; slot / block / step offsets are immediates;
  // KSPLIT: the wave's own 32-key block of every tile is folded in as well)
  const uint32_t khoff = (uint32_t)(kh * 32 * 2 * D);
  uint32_t rmA[KK], trA[2][DB];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    rmA[kk] = lds0 + (uint32_t)(Cfg::KOFF + fa.rm[kk]) + khoff;
    asm volatile("" : "+v"(rmA[kk]));  // opaque: one register per address, constants go to the offset fields
  }
#pragma unroll
  for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      trA[j2][db] = lds0 + (uint32_t)(Cfg::VOFF + fa.tr[j2][db]) + khoff;
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
    for (int kk = 0; kk < KK; ++kk) {
      if constexpr (W1) asm volatile("" : "+a"(qf[qb][kk]));  // MFMA-only operands: AGPRs
      else asm volatile("" ::"v"(qf[qb][kk]));  // (see attn_fwd.h: retire the loads in the waitcnt model)
    }
  const float c2 = a.scale * kLog2e;
  const bool fold_ok = c2 > 0.f;
  // Masked blocks inside the pipelined sweep (round 5; bias none / dense): the FMA's addend of a masked (key, row) position is -inf (p = 0) -- the causal diagonal
  // and the key tail cost two VALU ops (compare, select) per element of the blocks that carry a mask instead of an unpipelined tile each.  limq: last visible key
  // of the lane's row, minus the lane's 4 hi (register r <-> key crow(r, 0) + 4 hi).
  constexpr bool PMASK = BIAS != FAT5_BIAS_RPE1D;
  int limq[2] = {0, 0};
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) limq[qb] = (a.causal ? min(N - 1, qrow0 + 32 * qb + lq + P) : N - 1) - 4 * hi;
  float cst_neg = 0.f, cst_pos = 0.f;
  // band tiles of the pipelined sweep: LDS byte address of entry 0 of this lane's table copy, and the lane's position term
  // (index of block-relative key 0 for query block qb; see tile_exact: R + nb + 4 hi - qrow - ((R - qrow) & 3))
  uint32_t tabA = 0;
  int posb[2] = {0, 0};
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    cst_neg = sT[0];
    cst_pos = sT[2 * a.R];
    tabA = (uint32_t)(uintptr_t)sTa;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qrow = qrow0 + 32 * qb + lq;
      posb[qb] = a.R + 4 * hi - qrow - ((a.R - qrow) & 3);
    }
  }
  const int clamp_hi = rpe_n1p(a.R) - kRpePad - 28;  // (rpe_clamp_asc)
  auto tab_addr = [&](int qb, int nb) {  // LDS address of the lane's 28-entry window of block nb (clamped into the padded copy)
    return tabA + 4u * (uint32_t)min(max(posb[qb] + nb, -kRpePad), clamp_hi);
  };
  // dense bias: LDS byte address of this lane's 8 bytes of (key block kb, key group gg) of query block 0 in bias slot 0 -- keys
  // 32 kb + 8 gg + 4 hi + (0..3) of row 64 rg + lq; query block 1 sits 32 rows = 4096 bytes further, slot 1 Cfg::BIASB further
  uint32_t bA[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
  if constexpr (DENSE) {
    BiasTileReader brd;
    brd.init(64 * rg + lq, hi);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        bA[kb][gg] = lds0 + (uint32_t)(BOFF + brd.base[kb] + brd.goff[gg]);
        asm volatile("" : "+v"(bA[kb][gg]));
      }
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // ------------------------------------------------------------------------------------------------------------------
  // Unpipelined tile: MODE 0 generic (RPE band, N tail, causal diagonal), MODE 1 all-visible constant-bias tile.
  // Both query blocks per 32-key block: K / V fragments are read once and used twice.
  // NOMAX: reference point 0 for every row (the optimistic sweep's masked tiles); otherwise the exact running maximum.
  // ------------------------------------------------------------------------------------------------------------------
  auto acc_fence = [&]() {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int i = 0; i < DB; ++i) asm volatile("" : "+a"(oacc[qb][i]));
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int i = 0; i < DB; ++i) asm volatile("" : "+a"(oacc[qb][i]));
  };
  auto pv_mfma = [&](f32x16& acc, const u32x4 A, const u32x4 B) {
    if constexpr (W1) mfma_acc_agpr<BF16>(acc, A, B);
    else acc = mfma32<BF16>(A, B, acc);
  };
  auto tile_exact = [&]<int MODE, bool NOMAX, bool MAXONLY = false>(int t, int slot, float cst) {
    const int n0 = t * BN;
    const uint32_t soff = (uint32_t)(slot * TILE);
#pragma unroll
    for (int kbi = 0; kbi < NKB; ++kbi) {
      const int kb = KSPLIT ? kh : kbi;         // key block inside the tile
      const int kbo = KSPLIT ? 0 : kbi;         // ... as an address offset (KSPLIT: folded into the lane bases)
      const int nb = n0 + 32 * kb;
      f32x16 s[2];
      {
        u32x4 kf[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) kf[kk] = rd_k(soff, kbo, kk);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) s[qb] = mfma32<BF16>(kf[kk], qf[qb][kk], kk == 0 ? zero16 : s[qb]);
      }
      u32x4 vf[2][DB];
      if constexpr (!MAXONLY) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int db = 0; db < DB; ++db) vf[t2][db] = rd_v(soff, kbo, t2, db);
      }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int qr0 = qrow0 + 32 * qb, qrow = qr0 + lq;
        f32x16& sq = s[qb];
        float mul, add, mcand = 0.f;
        if constexpr (MODE == 1) {
          mul = c2;
          add = cst;
          if constexpr (!NOMAX) mcand = fmaf(max16(sq), c2, cst);
        } else {
          bool folded = fold_ok;
          if constexpr (DENSE) {
            // (unpipelined: masked tiles and the exact second pass) this lane's 16 bias values of the block from the staged tile
            folded = false;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              u32x2 wv = lds_rd64(bA[kb][g] + (uint32_t)((t & 1) * Cfg::BIASB + qb * 4096));
              wv[0] = bias_clamp2<BF16>(wv[0]);
              wv[1] = bias_clamp2<BF16>(wv[1]);
              sq[4 * g + 0] = fmaf(sq[4 * g + 0], c2, bias_log2(cvt_lo<BF16>(wv[0])));
              sq[4 * g + 1] = fmaf(sq[4 * g + 1], c2, bias_log2(cvt_hi<BF16>(wv[0])));
              sq[4 * g + 2] = fmaf(sq[4 * g + 2], c2, bias_log2(cvt_lo<BF16>(wv[1])));
              sq[4 * g + 3] = fmaf(sq[4 * g + 3], c2, bias_log2(cvt_hi<BF16>(wv[1])));
            }
          } else if constexpr (BIAS == FAT5_BIAS_RPE1D) {
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
          mul = folded ? c2 : 1.f;
          add = 0.f;
          if constexpr (!NOMAX) {
            const float m16 = max16(sq);
            mcand = folded ? m16 * c2 : m16;
          }
        }
        if constexpr (MAXONLY) {  // (the fp16 sweep's reference point: row maxima only)
          m_run[qb] = fmaxf(m_run[qb], pair_max(mcand));
          continue;
        }
        float ad = add - ((m_run[qb] == -INFINITY) ? 0.f : m_run[qb]);  // (NOMAX: the sweep's fixed reference point -- 0 in bf16)
        if constexpr (!NOMAX) {
          mcand = pair_max(mcand);
          if (__any(mcand > m_run[qb] + FAT5_DEFER_THR)) {
            const float m_new = fmaxf(m_run[qb], mcand);
            const float alpha = fast_exp2(m_run[qb] - ((m_new == -INFINITY) ? 0.f : m_new));
            l_run[qb][0] *= alpha;
            l_run[qb][1] *= alpha;
            if constexpr (W1) acc_fence();  // (asm MFMA -> VALU access of the accumulators: no padding is generated)
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
              for (int r = 0; r < 16; ++r) oacc[qb][i][r] *= alpha;
            if constexpr (W1) acc_fence();
            m_run[qb] = m_new;
          }
          ad = add - ((m_run[qb] == -INFINITY) ? 0.f : m_run[qb]);
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float x0 = fast_exp2(fmaf(sq[r], mul, ad)), x1 = fast_exp2(fmaf(sq[r + 1], mul, ad));
          sq[r] = x0;
          sq[r + 1] = x1;
          if constexpr (NOMAX) {
            // the sweep's row sums are sums of the ROUNDED probabilities (the operands of P.V); so are these: o is then a convex combination of V rows whatever
            // the reference point -- a row with ONE visible key returns that V row exactly, as the running-maximum form does (p = 1)
            const uint32_t w2 = pack2<BF16>(x0, x1);
            l_run[qb][0] += cvt_lo<BF16>(w2);
            l_run[qb][1] += cvt_hi<BF16>(w2);
          } else {
            l_run[qb][0] += x0;
            l_run[qb][1] += x1;
          }
        }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          u32x4 pb = pack8<BF16>(sq, t2);
          if constexpr (W1) asm volatile("s_nop 1" : "+v"(pb));  // (VALU write -> asm MFMA read: two wait states by hand)
#pragma unroll
          for (int db = 0; db < DB; ++db) pv_mfma(oacc[qb][db], vf[t2][db], pb);
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
  //            its chunk 14
  //   TN, tadr0   band tiles: table entries of the first eight keys (query block 0) of the block whose softmax is due -- fetched
  //            during the previous block -- and the LDS address of that block's window for query block 0
  // ------------------------------------------------------------------------------------------------------------------
  f32x16 S[2];
  u32x4 PB[2][2], VF[2][DB];
  float xc[2], pc[2];
  u32x4 TN[2];
  uint32_t tadr0 = 0;
  [[maybe_unused]] u32x2 BN0 = {0u, 0u}, BN1 = {0u, 0u};  // dense: the first two bias groups of the block whose softmax is due

  // One pipelined 32-key block i = 16 MFMA gaps.  Gap g holds, all mutually independent:
  //   MFMA g          g < 8: O^T += VF . PB (block i-1; VF was fetched during the previous block)   g >= 8: S' = K(KS, KB) . Q^T (block i+1)
  //   VALU            chunk g: x = s*c2 + bias (2 elements) | chunk g-1: p = exp2(x) | chunk g-2: pack to 16 bit
  //   LDS             gaps 0..3: one K fragment of (KS, KB); gaps 8..15: one ds_read_b64_tr_b16 of the V^T fragments of
  //                   (VS, VB) = block i; BAND: one 16-byte read of table entries in every odd gap, three gaps ahead of its FMAs
  //                   (gaps 13, 15: the first two reads of the NEXT block, whose first key is nbS + NSTEP)
  // No instruction waits on another one of its own gap and consecutive MFMAs never share an accumulator.  The VALU ops
  // are volatile asm: hipcc neither reorders nor packs them (v_pk_*_f32 beside MFMAs is an anti-lever, MI355X_MICROARCH.md)
  // and pads no hazards for them -- the youngest MFMA result they read (S'[0], finished by MFMA 14) is two MFMA issue
  // periods old when chunk 0 of the next block reads it.
  constexpr int NSTEP = KSPLIT ? 64 : 32;  // first key of a wave's next block minus first key of this one
  auto pipe_block = [&]<int KS, int KB, int VS, int VB, bool BAND, bool MK = false>(const float ad0, const float ad1, const int nbS, [[maybe_unused]] const int tt = 0) {
    static_assert(!(MK && BAND), "masked pipelined blocks: bias none / dense");
    // MK: the block whose softmax is due (first key nbS) carries a mask: element r of query block qb is masked when crow(r, 0) > dlm[qb]
    [[maybe_unused]] int dlm[2] = {0, 0};
    if constexpr (MK) {
      dlm[0] = limq[0] - nbS;
      dlm[1] = limq[1] - nbS;
    }
    auto madd = [&]<int QB, int R>(const float v) {  // the addend of element (QB, R)
      if constexpr (MK) return crow(R, 0) > dlm[QB] ? -INFINITY : v;
      else return v;
    };
    static_assert(D == 64 || D == 128, "gap schedules written for D = 64 (16 MFMAs, 16 two-element chunks per block) and D = 128 (32 MFMAs)");
    constexpr uint32_t koff = KS * TILE + KB * 32 * 2 * D, voff = VS * TILE + VB * 32 * 2 * D;
    if constexpr (W1) {
      // 32 gaps.  MFMA g < 16: O^T[qb][db] += V^T(t2, db) . P^T[qb][t2] of block i-1 (t2 outer, d-blocks, query blocks inner); g >= 16: S' = K(kk) Q^T of
      // block i+1.  VALU: chunk c (two elements) opens in gap 2c (arguments), exp in gap 2c+1, packed in gap 2c+2 (chunk 15: gap 0 of the next
      // block, `pc`).  LDS: gaps 0..7 the K fragments, gaps 16..31 the halves of the V^T fragments of block i; BAND: table entries >= 3 gaps ahead.
      u32x4 kf[KK];
      f32x16 Sn[2];
      u32x4 PBn[2][2];
      u32x2 vh[2][DB][2];
      float X[16][2], Pr[16][2];
      u32x4 T[2][4];
      uint32_t tadr1 = 0;
      if constexpr (BAND) {
        T[0][0] = TN[0];
        T[0][1] = TN[1];
      }
      // DENSE: the block's bias words [query block][4-key group] from this wave's rows of the staged tile (slot = tile parity: a run-time offset with three
      // ring slots); the first two groups were fetched during the previous block.  Per element one shift / mask and one multiply (log2 units) in front of the FMA,
      // placed one gap ahead of it
      [[maybe_unused]] u32x2 Bw[2][4];
      [[maybe_unused]] float Bf[16][2];
      [[maybe_unused]] const uint32_t bcur = (uint32_t)((tt & 1) * Cfg::BIASB), bnxt = VB == 0 ? bcur : (uint32_t)Cfg::BIASB - bcur;
      if constexpr (DENSE) {
        Bw[0][0] = BN0;
        Bw[0][1] = BN1;
      }
      auto bias_f = [&]<int C>() {  // chunk C: its two bias values in log2 units
        constexpr int cq = C >> 3, cr = 2 * (C & 7);
        const uint32_t wd = Bw[cq][cr >> 2][(cr & 3) >> 1];
        Bf[C][0] = asm_fma(asm_shl16(wd), kLog2e, cq == 0 ? ad0 : ad1);  // (ad = - the row's reference point)
        Bf[C][1] = asm_fma(asm_and_hi(wd), kLog2e, cq == 0 ? ad0 : ad1);
      };
      static_for<32>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        if constexpr (g < 16) {
          constexpr int qb = g & 1, db = (g >> 1) & 3, t2 = g >> 3;
          mfma_acc_agpr<BF16>(oacc[qb][db], VF[t2][db], PB[qb][t2]);
        } else {
          constexpr int idx = g - 16, qb = idx & 1, kk = idx >> 1;
          if constexpr (kk == 0) Sn[qb] = mfma32<BF16>(kf[kk], qf[qb][kk], zero16);
          else Sn[qb] = mfma32<BF16>(kf[kk], qf[qb][kk], Sn[qb]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- LDS ----
        if constexpr (g < KK) {
          kf[g] = lds_rd128(rmA[g] + koff);
        } else if constexpr (g >= 16) {
          constexpr int v = g - 16, t2 = v >> 3, db = (v >> 1) & 3, j2 = v & 1;
          vh[t2][db][j2] = lds_rd_tr_half(trA[j2][db] + voff + (uint32_t)(16 * t2 * 2 * D));
        }
        // the tile's ring requests, one 1-KiB piece per gap in the gaps without an LDS read (instead of eight back to back in front of the tile: SPREAD):
        // first block K(tt + NS - 1), second block V(tt + NS - 2) -- the order begin_iter issues them in (end_iter counts on it)
        if constexpr (SPREAD && g >= 8 && g < 8 + Dma::PER) {
          constexpr int i = g - 8;
          constexpr int sk = (VS + NS - 1) % NS, sv = (VS + NS - 2) % NS;
          if constexpr (VB == 0) {
            if (tt + NS - 1 < nt) dma16_asm(krs, wave_lds + (uint32_t)(Cfg::KOFF + sk * TILE + NT * 16 * i), kst.voff[0], (uint32_t)((tt + NS - 1) * BN) * kstride_b + kst.piece_step * i);
          } else {
            if (tt + NS - 2 < nt) dma16_asm(vrs, wave_lds + (uint32_t)(Cfg::VOFF + sv * TILE + NT * 16 * i), vst.voff[0], (uint32_t)((tt + NS - 2) * BN) * vstride_b + vst.piece_step * i);
          }
        }
        if constexpr (DENSE) {  // Bw[qb][j] is first unpacked in gap 16 qb + 4 j - 1
          if constexpr (g == 2) Bw[0][2] = lds_rd64(bA[VB][2] + bcur);
          else if constexpr (g == 6) Bw[0][3] = lds_rd64(bA[VB][3] + bcur);
          else if constexpr (g == 10) Bw[1][0] = lds_rd64(bA[VB][0] + bcur + 4096u);
          else if constexpr (g == 14) Bw[1][1] = lds_rd64(bA[VB][1] + bcur + 4096u);
          else if constexpr (g == 18) Bw[1][2] = lds_rd64(bA[VB][2] + bcur + 4096u);
          else if constexpr (g == 22) Bw[1][3] = lds_rd64(bA[VB][3] + bcur + 4096u);
          else if constexpr (g == 25 && VB == 1) bias_wait(tt);  // (the next block opens tile t + 1)
          else if constexpr (g == 26) BN0 = lds_rd64(bA[KB][0] + bnxt);  // the next block = (KS, KB)
          else if constexpr (g == 30) BN1 = lds_rd64(bA[KB][1] + bnxt);
        }
        if constexpr (BAND) {  // T[qb][j] is first used in gap 16 qb + 4 j
          if constexpr (g == 1) T[0][2] = lds_rd128(tadr0 + 64u);
          else if constexpr (g == 5) T[0][3] = lds_rd128(tadr0 + 96u);
          else if constexpr (g == 8) tadr1 = tab_addr(1, nbS);
          else if constexpr (g == 9) T[1][0] = lds_rd128(tadr1);
          else if constexpr (g == 13) T[1][1] = lds_rd128(tadr1 + 32u);
          else if constexpr (g == 17) T[1][2] = lds_rd128(tadr1 + 64u);
          else if constexpr (g == 21) T[1][3] = lds_rd128(tadr1 + 96u);
          else if constexpr (g == 24) tadr0 = tab_addr(0, nbS + NSTEP);
          else if constexpr (g == 25) TN[0] = lds_rd128(tadr0);
          else if constexpr (g == 29) TN[1] = lds_rd128(tadr0 + 32u);
        }
        // ---- VALU ----
        if constexpr ((g & 1) == 0) {
          // pack of chunk c-1 (gap 0: chunk 15 of the previous block -> the last word of PB[1][1])
          constexpr int c = (g >> 1) - 1;
          if constexpr (c < 0) {
            PB[1][1][3] = asm_cvt_pk<BF16>(pc[0], pc[1]);
          } else {
            constexpr int cq = c >> 3, cr = 2 * (c & 7);
            PBn[cq][cr >> 3][(cr & 7) >> 1] = asm_cvt_pk<BF16>(Pr[c][0], Pr[c][1]);
          }
          // arguments of chunk g / 2
          constexpr int cc = g >> 1, cq = cc >> 3, cr = 2 * (cc & 7);
          if constexpr (DENSE) {
            if constexpr (cc == 0) bias_f.template operator()<0>();
            X[cc][0] = asm_fma(S[cq][cr], c2, madd.template operator()<cq, cr>(Bf[cc][0]));
            X[cc][1] = asm_fma(S[cq][cr + 1], c2, madd.template operator()<cq, cr + 1>(Bf[cc][1]));
          } else if constexpr (BAND) {
            float t0 = __uint_as_float(T[cq][cr >> 2][cr & 3]), t1 = __uint_as_float(T[cq][cr >> 2][(cr & 3) + 1]);
            if constexpr (!BF16) {
              asm_add(t0, cq == 0 ? ad0 : ad1);
              asm_add(t1, cq == 0 ? ad0 : ad1);
            }
            X[cc][0] = asm_fma(S[cq][cr], c2, t0);
            X[cc][1] = asm_fma(S[cq][cr + 1], c2, t1);
          } else {
            X[cc][0] = asm_fma(S[cq][cr], c2, madd.template operator()<cq, cr>(cq == 0 ? ad0 : ad1));
            X[cc][1] = asm_fma(S[cq][cr + 1], c2, madd.template operator()<cq, cr + 1>(cq == 0 ? ad0 : ad1));
          }
        } else {
          constexpr int c = g >> 1;
          Pr[c][0] = abl_exp2(X[c][0]);
          Pr[c][1] = abl_exp2(X[c][1]);
          if constexpr (DENSE && c < 15) bias_f.template operator()<c + 1>();
        }
        // row sums of the rounded probabilities: a group of four words two gaps or more after its last pack
        if constexpr ((FAT5_FWD_ABL & 1) != 0) {
        } else if constexpr (g == 3) mfma16_acc<BF16>(lacc[1], sel, PB[1][1]);
        else if constexpr (g == 11) mfma16_acc<BF16>(lacc[0], sel, PBn[0][0]);
        else if constexpr (g == 19) mfma16_acc<BF16>(lacc[0], sel, PBn[0][1]);
        else if constexpr (g == 27) mfma16_acc<BF16>(lacc[1], sel, PBn[1][0]);
        __builtin_amdgcn_sched_barrier(0);
      });
      pc[0] = Pr[15][0]; pc[1] = Pr[15][1];
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
      return;
    } else {
    u32x4 kf[KK];
    f32x16 Sn[2];
    u32x4 PBn[2][2];
    u32x2 vh[2][DB][2];  // halves of the transposed fragments
    float X[16][2], Pr[16][2];
    u32x4 T[2][4];       // BAND: table entries of this block, [query block][8-key group]
    uint32_t tadr1 = 0;
    if constexpr (BAND) {
      T[0][0] = TN[0];
      T[0][1] = TN[1];
    }
    // DENSE: the bias words of this block (= (VS, VB)), [query block][4-key group] = two 2-element words each; the first two groups were
    // fetched during the previous block (BN0, BN1).  The values enter in log2 units: one shift / mask and one multiply per element on
    // top of the FMA (no clamp: a finfo.min bias overflows to -inf here, p = 0; a row of nothing but such entries has l = 0 and sends
    // its workgroup through the exact pass, which clamps like the 32-row body)
    [[maybe_unused]] u32x2 Bw[2][4];
    if constexpr (DENSE) {
      Bw[0][0] = BN0;
      Bw[0][1] = BN1;
    }
    constexpr uint32_t bcur = (uint32_t)((VS & 1) * Cfg::BIASB), bnxt = (uint32_t)((KS & 1) * Cfg::BIASB);
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
      if constexpr (DENSE) {  // chunk pair (2p, 2p + 1) = query block p >> 2, key group p & 3: fetched three gaps ahead
        if constexpr (g == 1) Bw[0][2] = lds_rd64(bA[VB][2] + bcur);
        else if constexpr (g == 3) Bw[0][3] = lds_rd64(bA[VB][3] + bcur);
        else if constexpr (g == 5) Bw[1][0] = lds_rd64(bA[VB][0] + bcur + 4096u);
        else if constexpr (g == 7) Bw[1][1] = lds_rd64(bA[VB][1] + bcur + 4096u);
        else if constexpr (g == 9) Bw[1][2] = lds_rd64(bA[VB][2] + bcur + 4096u);
        else if constexpr (g == 11) Bw[1][3] = lds_rd64(bA[VB][3] + bcur + 4096u);
        else if constexpr (g == 12 && VB == 1) bias_wait(tt);  // (the next block opens tile t + 1: this wave's pieces of its bias tile, requested ahead of this iteration's K / V)
        else if constexpr (g == 13) BN0 = lds_rd64(bA[KB][0] + bnxt);  // the next block = (KS, KB)
        else if constexpr (g == 15) BN1 = lds_rd64(bA[KB][1] + bnxt);
      }
      if constexpr (BAND) {
        if constexpr (g == 1) T[0][2] = lds_rd128(tadr0 + 64u);
        else if constexpr (g == 3) T[0][3] = lds_rd128(tadr0 + 96u);
        else if constexpr (g == 4) tadr1 = tab_addr(1, nbS);
        else if constexpr (g == 5) T[1][0] = lds_rd128(tadr1);
        else if constexpr (g == 7) T[1][1] = lds_rd128(tadr1 + 32u);
        else if constexpr (g == 9) T[1][2] = lds_rd128(tadr1 + 64u);
        else if constexpr (g == 11) T[1][3] = lds_rd128(tadr1 + 96u);
        else if constexpr (g == 12) tadr0 = tab_addr(0, nbS + NSTEP);
        else if constexpr (g == 13) TN[0] = lds_rd128(tadr0);
        else if constexpr (g == 15) TN[1] = lds_rd128(tadr0 + 32u);
      }
      // ---- VALU: pack of chunk g-2 ----
      {
        constexpr int c = g - 2;  // -2, -1: chunks 14, 15 of the previous block (query block 1, words 2, 3 of PB[1][1])
        constexpr int cq = c < 0 ? 1 : (c >> 3), cr = c < 0 ? 2 * (c + 8) : 2 * (c & 7);
        float p0, p1;
        if constexpr (c == -2) { p0 = pc[0]; p1 = pc[1]; }
        else if constexpr (c == -1) { p0 = Pr[15][0]; p1 = Pr[15][1]; }  // (slot 15 of this block's array is free until gap 15)
        else { p0 = Pr[c][0]; p1 = Pr[c][1]; }
        uint32_t wd = 0u;
        if constexpr ((FAT5_FWD_ABL & 8) == 0) wd = asm_cvt_pk<BF16>(p0, p1);
        if constexpr (c < 0) PB[1][1][(cr & 7) >> 1] = wd;
        else PBn[cq][cr >> 3][(cr & 7) >> 1] = wd;
      }
      // ---- VALU: exp of chunk g-1 ----
      if constexpr (g == 0) {
        Pr[15][0] = abl_exp2(xc[0]);
        Pr[15][1] = abl_exp2(xc[1]);
      } else {
        Pr[g - 1][0] = abl_exp2(X[g - 1][0]);
        Pr[g - 1][1] = abl_exp2(X[g - 1][1]);
      }
      if constexpr ((FAT5_FWD_ABL & 2) != 0) {  // (sums of the unrounded probabilities, the round-2 form)
        constexpr int e = g == 0 ? 15 : g - 1;
        asm_add(lv[g == 0 ? 1 : (e >> 3)], Pr[e][0]);
        asm_add(lv[g == 0 ? 1 : (e >> 3)], Pr[e][1]);
      }
      // ---- VALU: exponent arguments of chunk g ----
      {
        constexpr int cq = g >> 3, cr = 2 * (g & 7);
        if constexpr (DENSE) {
          const uint32_t wd = Bw[cq][cr >> 2][(cr & 3) >> 1];
          float b0, b1;
          if constexpr (BF16) {
            b0 = asm_fma(asm_shl16(wd), kLog2e, cq == 0 ? ad0 : ad1);  // (ad = - the row's reference point)
            b1 = asm_fma(asm_and_hi(wd), kLog2e, cq == 0 ? ad0 : ad1);
          } else {
            const f16x2_t hv = __builtin_bit_cast(f16x2_t, wd);
            b0 = (float)hv[0] * kLog2e;
            b1 = (float)hv[1] * kLog2e;
          }
          X[g][0] = asm_fma(S[cq][cr], c2, madd.template operator()<cq, cr>(b0));
          X[g][1] = asm_fma(S[cq][cr + 1], c2, madd.template operator()<cq, cr + 1>(b1));
        } else if constexpr (BAND) {
          float t0 = __uint_as_float(T[cq][cr >> 2][cr & 3]), t1 = __uint_as_float(T[cq][cr >> 2][(cr & 3) + 1]);
          if constexpr (!BF16) {  // (fp16: table entry minus the row's reference point -- in bf16 the reference point is 0)
            asm_add(t0, cq == 0 ? ad0 : ad1);
            asm_add(t1, cq == 0 ? ad0 : ad1);
          }
          X[g][0] = asm_fma(S[cq][cr], c2, t0);
          X[g][1] = asm_fma(S[cq][cr + 1], c2, t1);
        } else {
          X[g][0] = asm_fma(S[cq][cr], c2, madd.template operator()<cq, cr>(cq == 0 ? ad0 : ad1));
          X[g][1] = asm_fma(S[cq][cr + 1], c2, madd.template operator()<cq, cr + 1>(cq == 0 ? ad0 : ad1));
        }
      }
      // chunk c is packed in gap c + 2: a group of four words is summed two gaps after its last one (the previous block's last group,
      // whose words 2, 3 were packed in gaps 0, 1, in gap 3); nothing reads lacc before the end of the sweep
      if constexpr ((FAT5_FWD_ABL & 3) != 0) {
      } else if constexpr (g == 3) mfma16_acc<BF16>(lacc[1], sel, PB[1][1]);
      else if constexpr (g == 7) mfma16_acc<BF16>(lacc[0], sel, PBn[0][0]);
      else if constexpr (g == 11) mfma16_acc<BF16>(lacc[0], sel, PBn[0][1]);
      else if constexpr (g == 14) mfma16_acc<BF16>(lacc[1], sel, PBn[1][0]);
      __builtin_amdgcn_sched_barrier(0);  // pin the hand-placed MFMA / VALU / LDS interleave of this gap
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
    }
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
    const float q0 = W1 ? 0.f : fast_exp2(xc[0]), q1 = W1 ? 0.f : fast_exp2(xc[1]);
    if constexpr (W1) {  // (only the last chunk's pack is pending)
      PB[1][1][3] = pack2<BF16>(pc[0], pc[1]);
    } else {
      PB[1][1][2] = pack2<BF16>(pc[0], pc[1]);
      PB[1][1][3] = pack2<BF16>(q0, q1);
    }
    asm volatile("s_nop 1" : "+v"(PB[1][1]));  // (VALU write -> asm MFMA read: two wait states by hand)
    if constexpr ((FAT5_FWD_ABL & 3) == 0) mfma16_acc<BF16>(lacc[1], sel, PB[1][1]);
    if constexpr ((FAT5_FWD_ABL & 2) != 0) lv[1] += q0 + q1;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) pv_mfma(oacc[qb][db], VF[t2][db], PB[qb][t2]);
    pipe_reset();
  };
  // fold the matrix-pipe row sums into l_run (both key halves hold the full sum and pair_sum adds the halves: half each, exact)
  auto merge_lacc = [&]() {
    asm volatile("s_nop 11" : "+v"(lacc[0]), "+v"(lacc[1]));  // (asm MFMA -> VALU read: no padding is generated; tied so no read moves above it)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      l_run[qb][0] += 0.5f * lacc[qb][0];
      lacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr ((FAT5_FWD_ABL & 2) != 0) {  // (per-lane sums over the lane's own key half: pair_sum adds the halves)
        l_run[qb][0] += lv[qb];
        lv[qb] = 0.f;
      }
    }
  };

  // Pipelined tiles [t, te), t a multiple of NS on entry (ring slot 0).  Tile-iteration t runs the wave's blocks of tile t: it
  // reads K(t) and K(t+1) (scores one block ahead) and V(t) -- inside the ring protocol's window.  Steady state: NS tiles (ring
  // slots 0 .. NS-1) per trip, straight-line -- a slot switch inside the loop makes the register allocator reconcile the bodies at
  // every merge (tuple copies, spilled accumulators).
  auto pipe_run = [&]<bool BAND, bool MK = false>(int& t, const int te, int& slot, const float cst) {
    // the exponent's addend per query block: the tile-constant bias minus the row's reference point (bf16: 0; fp16: m_run, fixed during the sweep)
    const float ad0 = REF0 ? cst : cst - m_run[0], ad1 = REF0 ? cst : cst - m_run[1];
    auto one_tile = [&]<int SL>(int tt) {
      constexpr int S1 = (SL + 1) % NS;
      begin_iter(tt, SL, true);
      if constexpr (KSPLIT) {
        pipe_block.template operator()<S1, 0, SL, 0, BAND, MK>(ad0, ad1, tt * BN + 32 * kh);  // (the wave's key block: folded into its lane bases)
      } else {
        pipe_block.template operator()<SL, 1, SL, 0, BAND, MK>(ad0, ad1, tt * BN, tt);
        pipe_block.template operator()<S1, 0, SL, 1, BAND, MK>(ad0, ad1, tt * BN + 32, tt);
      }
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
  };
  auto exact_range = [&]<int MODE, bool NOMAX>(int& t, const int te, int& slot, const float cst) {
    for (; t < te; ++t) {
      begin_iter(t, slot);
      tile_exact.template operator()<MODE, NOMAX>(t, slot, cst);
      slot = slot == NS - 1 ? 0 : slot + 1;
      end_iter(t);
    }
  };

  // Tile classes (workgroup-uniform).  t_full: tiles all of whose keys are visible to every row of the workgroup.
  //   exact pass:       [0, ta) constant bias cst_a | [ta, tb0) generic | [tb0, tb1) constant cst_b | [tb1, nt) generic
  //   optimistic sweep: [0, tbA) constant | [tbA, tbB) band (rounded outwards to multiples of NS tiles: a band-mode tile reads
  //                     the padded table whatever its position) | [tbB, t_full) constant | [t_full, nt) masked, unpipelined
  int t_full = N / BN;
  if (a.causal && !ctab) t_full = min(t_full, max(0, (m0 + P + 1) / BN));
  t_full = min(t_full, nt);
  int ta = 0, tb0 = 0, tb1 = 0, tbA = 0, tbB = 0;
  float cst_a = 0.f, cst_b = 0.f;
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    const int lim_a = m0 - a.R - (BN - 1);
    ta = lim_a >= 0 ? min(t_full, lim_a / BN + 1) : 0;
    const int lo = m0 + BM - 1 + a.R;
    tb0 = min(t_full, max(ta, (lo + BN - 1) / BN));
    tb1 = t_full;
    cst_a = cst_neg;
    cst_b = cst_pos;
    tbA = ta / NS * NS;
    tbB = min(t_full, (tb0 + NS - 1) / NS * NS);
  } else if constexpr (DENSE) {
    ta = 0;              // (exact pass: every tile generic; optimistic sweep: every unmasked tile in the pipelined dense mode)
    tbA = tbB = t_full;
  } else {
    ta = t_full;
    tbA = tbB = t_full;  // (no band: one constant range, bias 0)
  }
  if (tb1 < tb0) tb0 = tb1 = ta;

  FAT5_FSTAMP(4);
  for (int pass = 0;; ++pass) {
    const bool nomax = OPT && pass == 0;
    if constexpr (W1) {
      if (pass > 0) acc_fence();
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[qb][i][r] = 0.f;
      m_run[qb] = (nomax && !ref_first) ? 0.f : -INFINITY;
      l_run[qb][0] = l_run[qb][1] = 0.f;
    }
    if (pass > 0) {
      __syncthreads();  // every wave is done with the rings (and, KSPLIT, with the merge area inside them)
      stage_first();
      wait_dma_all();
      __syncthreads();
    }
    int t = 0, slot = 0;
    if constexpr (OPT) {
      if (nomax) {
        if (ref_first) {
          // fp16, and bf16 on request (a.ref_first): the reference point of every row = its maximum over the first tile (scores only); a row that sees none of its
          // keys there, or nothing but masked-out bias entries (finfo.min: left padding), keeps 0
          if (nt > 0) tile_exact.template operator()<0, false, true>(0, 0, 0.f);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb)
            if (!(m_run[qb] > -0x1p100f)) m_run[qb] = 0.f;
        }
        // PMASK: whole trips of unmasked tiles first, then every tile up to nt in the masked form of the pipelined block (the visible tiles of its first trip
        // pay the mask arithmetic as well: nothing masked)
        const int te1 = PMASK ? t_full / NS * NS : t_full;
        if (PMASK ? nt > 0 : t_full > 0) {
          // fill: scores of the wave's first block, nothing pending
          {
            u32x4 kf[KK];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) kf[kk] = rd_k(0u, 0, kk);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
              for (int qb = 0; qb < 2; ++qb) S[qb] = mfma32<BF16>(kf[kk], qf[qb][kk], kk == 0 ? zero16 : S[qb]);
            pipe_reset();
          }
          if constexpr (BIAS == FAT5_BIAS_RPE1D) {
            pipe_run.template operator()<false>(t, tbA, slot, cst_a);
            if (t < tbB) {  // entering the band: the first block's table entries for query block 0 (later blocks prefetch their successor's)
              tadr0 = tab_addr(0, t * BN + 32 * kh);
              TN[0] = lds_rd128(tadr0);
              TN[1] = lds_rd128(tadr0 + 32u);
            }
            pipe_run.template operator()<true>(t, tbB, slot, 0.f);
            pipe_run.template operator()<false>(t, t_full, slot, cst_b);
          } else {
            if constexpr (DENSE) {  // the first block's first two bias groups (later blocks prefetch their successor's)
              BN0 = lds_rd64(bA[0][0]);
              BN1 = lds_rd64(bA[0][1]);
            }
            pipe_run.template operator()<false>(t, te1, slot, 0.f);
            if constexpr (PMASK) pipe_run.template operator()<false, true>(t, nt, slot, 0.f);
          }
          pipe_drain();
          merge_lacc();
        }
        exact_range.template operator()<0, true>(t, nt, slot, 0.f);
      }
    }
    if (!nomax) {
      if (fold_ok) {
        exact_range.template operator()<1, false>(t, ta, slot, cst_a);
        exact_range.template operator()<0, false>(t, min(max(tb0, ta), nt), slot, 0.f);
        exact_range.template operator()<1, false>(t, tb1, slot, cst_b);
      }
      exact_range.template operator()<0, false>(t, nt, slot, 0.f);
    }

    FAT5_FSTAMP(5);
    if constexpr (KSPLIT) {
      // ---- merge the two key halves of every 64-row group through LDS: wave kh keeps query block kh, hands the other one over ----
      __syncthreads();  // the rings are free
      char* mine = smem + w * Cfg::MERGE;
      const char* theirs = smem + (w ^ 1) * Cfg::MERGE;
      auto put = [&]<int QB>() {
#pragma unroll
        for (int i = 0; i < DB * 4; ++i) {
          const f32x4 v4 = {oacc[QB][i >> 2][4 * (i & 3)], oacc[QB][i >> 2][4 * (i & 3) + 1], oacc[QB][i >> 2][4 * (i & 3) + 2], oacc[QB][i >> 2][4 * (i & 3) + 3]};
          *reinterpret_cast<f32x4*>(mine + (i * 64 + l) * 16) = v4;
        }
        *reinterpret_cast<float2*>(mine + DB * 4 * 64 * 16 + l * 8) = make_float2(m_run[QB], l_run[QB][0] + l_run[QB][1]);
      };
      auto get = [&]<int QB>() {
        const float2 ml = *reinterpret_cast<const float2*>(theirs + DB * 4 * 64 * 16 + l * 8);
        const float m_new = fmaxf(m_run[QB], ml.x);
        const float mref = (m_new == -INFINITY) ? 0.f : m_new;
        const float sa = fast_exp2(m_run[QB] - mref), sb = fast_exp2(ml.x - mref);
#pragma unroll
        for (int i = 0; i < DB * 4; ++i) {
          const f32x4 v4 = *reinterpret_cast<const f32x4*>(theirs + (i * 64 + l) * 16);
#pragma unroll
          for (int j = 0; j < 4; ++j) oacc[QB][i >> 2][4 * (i & 3) + j] = oacc[QB][i >> 2][4 * (i & 3) + j] * sa + v4[j] * sb;
        }
        l_run[QB][0] = (l_run[QB][0] + l_run[QB][1]) * sa + ml.y * sb;
        l_run[QB][1] = 0.f;
        m_run[QB] = m_new;
      };
      if (kh == 0) put.template operator()<1>(); else put.template operator()<0>();
      __syncthreads();
      if (kh == 0) get.template operator()<0>(); else get.template operator()<1>();
    }

    if constexpr (!OPT) {
      break;
    } else {
      if (!nomax) break;
      // the sweep's reference point was 0 for every row: a row sum outside [2^-40, 2^100) (overflow / flushed terms) -> exact pass
      bool bad = false;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        if (KSPLIT && qb != kh) continue;
        const float lt = pair_sum(l_run[qb][0] + l_run[qb][1]);
        const bool visible = !(a.causal && qrow0 + 32 * qb + lq + P < 0);  // (a row without any visible key has l = 0 exactly)
        bad = bad || !(lt < 0x1p100f) || (visible && lt < 0x1p-40f);
      }
      if constexpr (D3) {  // (the flag word is the first word of the bias ring: every wave is done with its tiles first)
        __syncthreads();
        if (tid == 0) *sFlag = 0;
        __syncthreads();
      }
      if (bad && (FAT5_FWD_ABL & 13) == 0) *sFlag = 1;  // (ablations with wrong sums must not take the exact pass)
      __syncthreads();
      if (*sFlag == 0) break;
    }
  }

  FAT5_FSTAMP(6);
  // ---- epilogue: o = acc / l, L = m + ln(l) ----
  if constexpr (KSPLIT) {
    // Short sequences (round 4): O leaves through an LDS image of the wave's 32 rows -- 8-byte pieces into the swizzled row-major image,
    // out again as whole 128-byte rows; a lane storing "its" row straight to global touches 64 cache lines per instruction, and at
    // 512 keys the store tail is a visible part of the kernel.  The image sits behind the merge area (which the partner wave may
    // still be reading); every wave is behind the sweep's last barrier.  (The 256-row form keeps the direct stores: its instantiations
    // sit at the register cap of two waves per SIMD and the tail is < 1 % of a long sweep.)
    static_assert(Cfg::NW * Cfg::MERGE + Cfg::NW * 32 * 2 * D <= 2 * NS * TILE, "the O images fit behind the merge area");
    char* oimg = smem + Cfg::NW * Cfg::MERGE + w * (32 * 2 * D);
    {
      const int qb = kh;
      const float l_tot = pair_sum((kh == 0 ? l_run[0][0] + l_run[0][1] : l_run[1][0] + l_run[1][1]));
      const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
      const int qrow = qrow0 + 32 * qb + lq;
      auto put = [&]<int QB>() {
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            u32x2 wv;
            wv[0] = pack2<BF16>(oacc[QB][db][4 * g + 0] * inv, oacc[QB][db][4 * g + 1] * inv);
            wv[1] = pack2<BF16>(oacc[QB][db][4 * g + 2] * inv, oacc[QB][db][4 * g + 3] * inv);
            *reinterpret_cast<u32x2*>(oimg + rm_off<D>(lq, 4 * db + g) + 8 * hi) = wv;
          }
      };
      if (kh == 0) put.template operator()<0>(); else put.template operator()<1>();
      if (qrow < M && hi == 0) a.lse[lse_off + qrow] = l_tot > 0.f ? ((kh == 0 ? m_run[0] : m_run[1]) + fast_log2(l_tot)) * kLn2 : -INFINITY;
    }
    const int orow0 = qrow0 + 32 * kh;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + (l >> 3), slot = l & 7;
      const u32x4 v4 = *reinterpret_cast<const u32x4*>(oimg + row * (2 * D) + slot * 16);
      if (orow0 + row < M) *reinterpret_cast<u32x4*>(ob_ + (int64_t)(orow0 + row) * a.os[2] + ((slot ^ swz<D>(row)) << 3)) = v4;
    }
  } else if constexpr (W1) {
    // O through an LDS image of the wave's 64 rows (the rings are free behind the barrier), out again as whole 256-byte rows
    acc_fence();
    __syncthreads();
    char* oimg = smem + w * (64 * 2 * D);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qrow = qrow0 + 32 * qb + lq;
      const float l_tot = pair_sum(l_run[qb][0] + l_run[qb][1]);
      const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 wv;
          wv[0] = pack2<BF16>(oacc[qb][db][4 * g + 0] * inv, oacc[qb][db][4 * g + 1] * inv);
          wv[1] = pack2<BF16>(oacc[qb][db][4 * g + 2] * inv, oacc[qb][db][4 * g + 3] * inv);
          *reinterpret_cast<u32x2*>(oimg + rm_off<D>(32 * qb + lq, 4 * db + g) + 8 * hi) = wv;
        }
      if (qrow < M && hi == 0) a.lse[lse_off + qrow] = l_tot > 0.f ? (m_run[qb] + fast_log2(l_tot)) * kLn2 : -INFINITY;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 4 * i + (l >> 4), slot = l & 15;
      const u32x4 v4 = *reinterpret_cast<const u32x4*>(oimg + row * (2 * D) + slot * 16);
      if (qrow0 + row < M) *reinterpret_cast<u32x4*>(ob_ + (int64_t)(qrow0 + row) * a.os[2] + ((slot ^ swz<D>(row)) << 3)) = v4;
    }
  } else {
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
#if FAT5_TRACE
  FAT5_FSTAMP(7);
  if (threadIdx.x == 0) {
    long long* dst = reinterpret_cast<long long*>(ob_ + (int64_t)m0 * a.os[2]);
#pragma unroll
    for (int i = 0; i < 9; ++i) dst[i] = fstamp[i];
  }
#endif
}

template <int D, bool BF16, int BIAS, bool KSPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))  // (two waves per SIMD: <= 256 registers)
void attn_fwd64_kernel(const AttnArgs a) {
  int b, h, mblk;
  decode_unit(a, blockIdx.x, a.n_mblk, b, h, mblk, (FAT5_CAUSAL_ORDER && a.causal) ? 1 : 0);
  attn_fwd64_body<D, BF16, BIAS, KSPLIT>(a, b, h, mblk * Fwd64Cfg<D, KSPLIT>::BM);
}
// head_dim 128 (round 5): one wave per SIMD (512 registers per lane)
template <int D, bool BF16, int BIAS, bool SPREAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void attn_fwd64_w1_kernel(const AttnArgs a) {
  int b, h, mblk;
  decode_unit(a, blockIdx.x, a.n_mblk, b, h, mblk, (FAT5_CAUSAL_ORDER && a.causal) ? 1 : 0);
  attn_fwd64_body<D, BF16, BIAS, false, SPREAD>(a, b, h, mblk * Fwd64Cfg<D, false>::BM);
}
// dense bias: the two-tile bias ring (64 KB) beside the K / V rings leaves room for ONE workgroup per CU -- one wave per SIMD
template <int D, bool BF16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void attn_fwd64_dense_kernel(const AttnArgs a) {
  int b, h, mblk;
  decode_unit(a, blockIdx.x, a.n_mblk, b, h, mblk, (FAT5_CAUSAL_ORDER && a.causal) ? 1 : 0);
  attn_fwd64_body<D, BF16, FAT5_BIAS_DENSE, false>(a, b, h, mblk * Fwd64Cfg<D, false>::BM);
}
// Both workgroup forms in ONE launch (round 4) for problems of 1 .. 2 64-row waves per SIMD.  With 1.5 waves of 64 rows per SIMD either
// pure form leaves half of the SIMDs with twice the work of the others ((4,12,2048): 58 us where the matrix pipe's share is ~32).  Here
// the first `mix_na` workgroups are 256-row ones (four waves, one per SIMD, each with all keys of its 64 rows), the others key-split
// 128-row ones (four waves, one per SIMD, each with half the keys of 64 rows): one of each per CU gives every SIMD 1 + 1/2 units.
// Pair k of XCD x (unit 8 k + x) gets a_lo (+1 for the first mix_k_hi pairs of the XCD) 256-row workgroups from row 0 on and 128-row
// workgroups for the rest of its rows; both kinds of a pair sit on the pair's XCD (K / V stay in one L2).
template <int D, bool BF16, int BIAS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void attn_fwd64_mixed_kernel(const AttnArgs a) {
  const int M = a.M;
  const int a_lo = a.mix_a_lo, k_hi = a.mix_k_hi;
  const int b_hi = (max(M - 256 * (a_lo + 1), 0) + 127) / 128, b_lo = (max(M - 256 * a_lo, 0) + 127) / 128;
  // launch order inside an XCD (sequence number idx = blockIdx >> 3): all 256-row workgroups first.  The dispatcher deals the workgroups
  // of an XCD round-robin over its 32 CUs (traced: HW_REG_HW_ID per workgroup in the FAT5_TRACE build), so with na_x = nb_x = 32 CU c gets
  // 256-row workgroup c and 128-row workgroup c -- one of each; alternating the kinds in launch order pairs like with like (66 vs 62.5 us).
  if (a.unit_count > 0) {
    // a unit range (sharded runs): every row must go through the SAME form as in the unsharded launch (bit-identical results) -- the
    // pair's share of 256-row workgroups follows from its place in the full problem.  (a_lo + 1) + b_lo slots per pair, the unused exit.
    const int sp = a_lo + 1 + b_lo;
    const int pi = blockIdx.x / sp, sl = blockIdx.x - pi * sp;
    const int u = a.unit_begin + pi, h = u / a.B, b = u - h * a.B;
    const int k = (b * a.H + h) >> 3, ak = a_lo + (k < k_hi ? 1 : 0), bk = k < k_hi ? b_hi : b_lo;
    if (sl < a_lo + 1) {
      if (sl < ak) attn_fwd64_body<D, BF16, BIAS, false>(a, b, h, 256 * sl);
    } else if (sl - (a_lo + 1) < bk) {
      attn_fwd64_body<D, BF16, BIAS, true>(a, b, h, 256 * ak + 128 * (sl - (a_lo + 1)));
    }
    return;
  }
  const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int na_x = a.mix_na >> 3;
  const bool kindA = idx < na_x;
  const int i = kindA ? idx : idx - na_x;
  if (kindA) {
    const int thr = k_hi * (a_lo + 1);
    int k, j;
    if (i < thr) { k = fast_div(i, a_lo + 1, a.mg_mix[0]); j = i - k * (a_lo + 1); }
    else { const int i2 = i - thr; k = fast_div(i2, a_lo, a.mg_mix[1]); j = i2 - k * a_lo; k += k_hi; }
    const int ui = 8 * k + x, bb = fast_div(ui, a.H, a.mg_H);
    if (FAT5_MIX_PRIO == 2) __builtin_amdgcn_s_setprio(3);
    attn_fwd64_body<D, BF16, BIAS, false>(a, bb, ui - bb * a.H, 256 * j);
  } else {
    if (FAT5_MIX_PRIO == 1) __builtin_amdgcn_s_setprio(3);
    const int thr = k_hi * b_hi;
    int k, j, r0;
    if (i < thr) { k = fast_div(i, b_hi, a.mg_mix[2]); j = i - k * b_hi; r0 = 256 * (a_lo + 1); }
    else { const int i2 = i - thr; k = fast_div(i2, b_lo, a.mg_mix[3]); j = i2 - k * b_lo; k += k_hi; r0 = 256 * a_lo; }
    const int ui = 8 * k + x, bb = fast_div(ui, a.H, a.mg_H);
    attn_fwd64_body<D, BF16, BIAS, true>(a, bb, ui - bb * a.H, r0 + 128 * j);
  }
}

}  // namespace fat5
