// FlashAttention-2 backward, dK / dV / bias-table-gradient body for long sequences on gfx950: 64 keys per wave,
// software-pipelined query-step loop.
//
// Same contract as attn_bwd_kv_kernel (attn_bwd.h; replaces the reference Triton `_bwd_kv_kernel`,
// src/model/ops/flash_attention_v2_bias.py:559-745) for bias = none / in-kernel T5 RPE, bf16 / fp16, D = 64.  What is different, and why:
//
//  * LDS bandwidth.  In the 32-keys-per-wave body every wave reads the whole Q and dO tile twice (row-major for S / dP,
//    transposed for dK / dV): 1 KiB of LDS per MFMA, exactly the CU's 128 B/clk at 100 % MFMA issue -- measured 42 % MFMA
//    utilisation with the waves parked on LDS 44 % of the time.  Here a wave owns TWO 32-key blocks: every fragment read feeds
//    two MFMAs (512 B per MFMA), and consecutive MFMAs always target different accumulators (see attn_fwd64.h).
//  * Registers: dK^T / dV^T of 64 keys (128) + K / V operand fragments (64) + scores -> one wave per SIMD (512 registers),
//    so nothing hides latency but the instruction stream itself: three-stage software pipeline over 32-row query steps i.
//    While the VALU pipe turns S', dP' of step i into P and dS (one FMA, v_exp_f32, one multiply per element, packed to bf16),
//    the matrix pipe runs dV^T += dO^T P, dK^T += Q^T dS of step i-1 and S' = Q K^T - L/scale, dP' = dO V^T - delta of step i+1:
//    32 MFMAs per step, one element of VALU work, about one LDS read in every MFMA gap.
//  * Q / dO steps and their row statistics travel global -> LDS by DMA into a 4-slot ring three steps ahead; one barrier
//    per step.  The statistics (-L/scale and -delta, the MFMA accumulators' INITIAL values) come in exactly that form from the
//    dQ kernel (AttnArgs::stat2).
//  * All-visible constant-bias steps (everything outside the RPE band / the causal diagonal / a key tail) run the pipelined
//    iteration; the others run the same pipeline stages one after the other with the general softmax (table lookups, masks,
//    per-diagonal sums through the skew tile of attn_bwd.h).
#pragma once
#include "attn_common.h"
#include "attn_fwd64.h"  // static_for, integer-address LDS reads, asm LDS-DMA, pinned VALU ops
#include "diag_sum.h"    // per-diagonal sums of dS on the VALU (DPP row rotations)

#ifndef FAT5_ABL
#define FAT5_ABL 0  // developer ablations of the dK/dV body (timing only, wrong results): 1 no diagonal ops, 2 no finish / hand-over, 4 no table reads, 8 no table fill / zeroing in the prologue, 16 no partial sums in the epilogue, 32 no far-bin MFMAs, 64 SELF: no statistics production in the pipelined iteration, 128 SELF: counted vmcnt (step j+1 landed) instead of vmcnt(0)
#endif

#ifndef FAT5_DMA_SPREAD
#define FAT5_DMA_SPREAD 0  // 1: the pipelined iterations issue the LDS-DMA pieces of a step one per MFMA gap instead of all behind the step's barrier.  What pays in
                           // attn_bwd_qdb64.h (nine pieces per wave and step: -130 us of 2.1 ms) LOSES here at two to seven pieces (tools/build_variant.py A/B, twice:
                           // cfg2 backward 32.5 vs 30.3 us, (4,12,2048) 201 vs 195, (4,12,8192) 2520 vs 2475): a uniform branch and a readfirstlane per piece in the gaps
#endif
#ifndef FAT5_TRACE
#define FAT5_TRACE 0  // developer build: thread 0 of every workgroup stamps s_memtime at its phase boundaries into the delta scratch (tools/trace64.py)
#endif

namespace fat5 {

#if FAT5_TRACE
#define FAT5_STAMP(slot) do { if (threadIdx.x == 0) reinterpret_cast<long long*>(a.delta)[(int64_t)blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define FAT5_STAMP(slot) do { } while (0)
#endif

// HALF (mid sequence lengths): the four waves form two PAIRS; both pairs own the same 128 keys (64 per wave) and each walks one
// half of the query steps through a ring of its own, then the second pair hands its dK^T / dV^T over through LDS.  With 256-key
// workgroups at one per CU (one wave per SIMD), (4,12,2048,64) is 384 workgroups on 256 CUs: two rounds, the second half empty.
// 768 half-length workgroups are three full rounds of half the length.
// SELF (round 4: the dK/dV half of a launch shared with the dQ body, attn_bwd_fused64_kernel): the body forms its row statistics
// itself instead of reading the dQ kernel's (AttnArgs::stat2) -- the only dependency between the two kernels.  Every step brings its
// O rows and its 32 raw log-sum-exp values along; two steps ahead of their use every wave turns 8 of the step's rows into
// -L/scale and -delta = -rowsum(o * do) (reference _bwd_preprocess, flash_attention_v2_bias.py:516-556) in the slot's statistics area.
// DENSE (round 5): a dense additive bias (the reference's own operator, flash_attention_v2_bias.py:436-443 / :652-729).  Every step's slot carries, per
// wave, the (32 query rows x the wave's 64 keys) 16-bit bias tile as one more row-major image -- fetched by the wave itself (four 1-KiB LDS-DMA pieces,
// covered by its own counted vmcnt: no barrier).  The bias is added ON THE MATRIX PIPE: S' = Q K^T + E B - L/scale with B the bias tile read as a
// transposed operand fragment (lane = key, k-slot = row: the ds_read_b64_tr_b16 addressing of the Q^T / dO^T fragments) and E[row][k] = 1/scale where
// k-slot k is that row -- two more 32x32x16 MFMAs per key block and step (+12.5 % pipe time, inside the slack of the VALU-bound gaps) instead of a
// shift / mask, a multiply by log2(e) and an add per element (+64 VALU instructions per step: measured +32 % on the kernel).  1/scale enters as TWO
// 16-bit terms (hi + lo, 2^-17 relative: the reference benchmarks at sm_scale 1.3) in two k-slots that both see the same bias row: a fragment holds
// one 8-byte transposing read twice, one MFMA covers 8 rows -- four per key block and step, eight reads as before.
template <int D, bool HALF = false, bool SELF = false, bool DENSE = false>
struct Bwd64Cfg {
  static_assert(!(HALF && SELF), "the self-sufficient variant exists for 256-key workgroups only");
  static_assert(!(DENSE && (HALF || SELF)), "dense bias: 256-key workgroups with the dQ kernel's statistics only");
  static constexpr int NW = 4, BNK = HALF ? 128 : 64 * NW, QT = 32, NT = 64 * NW, NS = 4;
  static constexpr int IMG = rm_bytes<D, QT>();  // one 32-row image (Q or dO)
  static constexpr int IMGS = SELF ? 3 : 2;      // Q | dO (| O)
  // statistics of a step: one private 1 KiB DMA piece per wave (256 B used: [32] -L/scale, [32] -delta); SELF: one shared 1 KiB DMA
  // piece of raw L (its first 128 bytes become -L/scale in place) + [32] -delta behind it
  static constexpr int STATB = SELF ? 1024 + 256 : NW * 1024;
  static constexpr int DLOFF = SELF ? 1024 : 128;  // byte offset of -delta inside the statistics area
  static constexpr int BIASO = IMGS * IMG + STATB;  // DENSE: the waves' bias images of the step
  static constexpr int SLOT = BIASO + (DENSE ? NW * IMG : 0);
  static constexpr int RING = NS * SLOT;
  static constexpr int RINGS = HALF ? 2 : 1;     // one ring of query steps per wave pair
  static constexpr int RINGB = RINGS * RING;
  static_assert(!HALF || 2 * 128 * 64 * 4 <= RINGB, "the hand-over of dK^T / dV^T (128 registers x 64 lanes per wave) reuses the rings");
  // staging (round 4, see BwdQ64Cfg): the 64 keys of a wave arrive as whole rows of K | V by LDS-DMA and dK | dV leave as whole rows
  // through the same images (HALF: the two pairs share the images of their common keys)
  static constexpr int STG_T = 64 * 2 * D;       // one tensor's 64 rows
  static constexpr int STG = (HALF ? 2 : NW) * 2 * STG_T;
  static __host__ __device__ constexpr int rpe0(bool stage) { return RINGB + (stage ? STG : 0); }  // the RPE state sits behind the ring(s) and the staging images
  // rpe: four aligned table copies + one private diagonal accumulator per (wave, key block)
  // (+ one scratch word per thread: the target of the lanes that have no near diagonal to store)
  static __host__ __device__ size_t rpe_off(int R) { return (rpe_table_bytes(R) + (size_t)(2 * R + 1) * 4 * 2 * NW + NT * 4 + 63) / 64 * 64; }
  static size_t smem(int R, int bias_mode, bool stage = false) { return rpe0(stage) + (bias_mode == FAT5_BIAS_RPE1D ? rpe_off(R) : 0); }
};

FAT5_DEV float asm_mul(float a, float b) {
  float r;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}


// (b, h, nblk): the key block of this workgroup; part_row: its row among the a.part_stride partial diagonal-sum rows of (b, h);
// part_zero_next: the row behind it is nobody's and has to read zero (a 256-key workgroup of a launch that counts rows in
// 128-key units: attn_bwd_kv64_mixed_kernel)
// ONE (DENSE only): 1 / scale is itself a 16-bit value (1: T5, 8: the default 1 / sqrt(64)) -- one selector term, 16 bias rows per MFMA: two bias MFMAs per key
// block and step instead of four (the launcher checks the scale; +5 % on the kernel for the general two-term form)
// NODIAG (round 6; the dK/dV half of a one-launch backward whose dQ half forms the table gradient's diagonal sums: attn_bwd_q64_body<..., QDG>): nothing of the
// per-diagonal machinery below -- element operations, finish / hand-over, far-bin MFMAs, LDS arrays, partial rows -- is generated
template <int D, bool BF16, int BIAS, bool HALF, bool SELF = false, bool ONE = false, bool NODIAG = false>
FAT5_DEV void attn_bwd_kv64_body(const AttnArgs& a, const int b, const int h, const int nblk, const int part_row, const bool part_zero_next) {
  static_assert(D == 64, "gap schedule written for D = 64");
  constexpr bool DENSE = BIAS == FAT5_BIAS_DENSE;
  constexpr bool DG = BIAS == FAT5_BIAS_RPE1D && !NODIAG;  // this body forms the per-diagonal sums of the table gradient
  FAT5_STAMP(0);
  using Cfg = Bwd64Cfg<D, HALF, SELF, DENSE>;
  constexpr int BNK = Cfg::BNK, NT = Cfg::NT, IMG = Cfg::IMG, SLOT = Cfg::SLOT;
  constexpr int KK = D / 16, DB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, lq = l & 31, hi = l >> 5;  // (w: provably wave-uniform)
  const int pr = HALF ? (w >> 1) : 0;  // HALF: wave pair = which half of the query steps
  const int wp = HALF ? (w & 1) : w;   // wave inside its pair / workgroup = which 64 keys
  const int bh = b * a.H + h;
  const int M = a.M, N = a.N;
  const int n0 = nblk * BNK;
  [[maybe_unused]] RpeTableRegs tabr;  // (the bias table's first round of loads leaves before every other request of the prologue)
  if constexpr (BIAS == FAT5_BIAS_RPE1D && !(FAT5_ABL & 8)) tabr = rpe_table_load_first(a.rpe1d + (int64_t)h * (2 * a.R + 1), a.R, tid, 64 * Cfg::NW);
  const uint16_t* qb = a.q + (int64_t)b * a.qs[0] + (int64_t)h * a.qs[1];
  const uint16_t* kb_ = a.k + (int64_t)b * a.ks[0] + (int64_t)h * a.ks[1];
  const uint16_t* vb = a.v + (int64_t)b * a.vs[0] + (int64_t)h * a.vs[1];
  const uint16_t* dob = a.dout + (int64_t)b * a.dos[0] + (int64_t)h * a.dos[1];
  uint16_t* dkb = a.dk + (int64_t)b * a.dks[0] + (int64_t)h * a.dks[1];
  uint16_t* dvb = a.dv + (int64_t)b * a.dvs[0] + (int64_t)h * a.dvs[1];
  const int P = N - M;
  // causal mask by the bias table itself (attn_common.h: rpe_table_fill_rest): the diagonal d = P lies inside the band (-R <= P < R: the far-negative
  // constant stays visible, the far-positive one is masked), entries above it are -inf
  [[maybe_unused]] const bool ctab = BIAS == FAT5_BIAS_RPE1D && a.causal && P < a.R && P >= -a.R;
  const int kw0 = n0 + 64 * wp;  // first key of this wave; key block kb covers kw0 + 32*kb .. +31
  const float ninf_c = a.scale > 0.f ? -INFINITY : INFINITY;  // a score this large (in S' units) is a zero probability

  // K and V fragments (B operands) of this lane's two keys: staged (whole rows by LDS-DMA into the images of this wave's keys, read
  // back after the prologue's wait; keys past N arrive as zeros -- their scores are masked) or straight from global
  const bool stg = a.lds_stage != 0;
  const uint32_t stg_k = (uint32_t)(uintptr_t)smem + (uint32_t)(Cfg::RINGB + wp * 2 * Cfg::STG_T);  // K image | V image of keys kw0 .. kw0 + 63
  u32x4 kf[2][KK], vf[2][KK];
  if (stg) {
    using SDma = DmaStage<D, 64, 64>;
    static_assert(SDma::PER == 8 && SDma::NV == 2, "eight 1-KiB pieces of 8 rows per tensor");
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)stg_k);
    if (!HALF || pr == 0) {
      SDma sk;
      sk.init(a.ks[2], l);
      const __amdgpu_buffer_rsrc_t rs = make_rows_rsrc(kb_, a.ks[2], N, D);
      const uint32_t r0 = (uint32_t)kw0 * (uint32_t)a.ks[2] * 2u;
#pragma unroll
      for (int i = 0; i < SDma::PER; ++i) dma16_asm(rs, dst + (uint32_t)(i * 1024), sk.voff[i % 2], r0 + sk.piece_step * (i / 2));
    }
    if (!HALF || pr == 1) {
      SDma sv;
      sv.init(a.vs[2], l);
      const __amdgpu_buffer_rsrc_t rs = make_rows_rsrc(vb, a.vs[2], N, D);
      const uint32_t r0 = (uint32_t)kw0 * (uint32_t)a.vs[2] * 2u;
#pragma unroll
      for (int i = 0; i < SDma::PER; ++i) dma16_asm(rs, dst + (uint32_t)(Cfg::STG_T + i * 1024), sv.voff[i % 2], r0 + sv.piece_step * (i / 2));
    }
  } else {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int kr = min(kw0 + 32 * kb + lq, N - 1);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        kf[kb][kk] = *reinterpret_cast<const u32x4*>(kb_ + (int64_t)kr * a.ks[2] + 16 * kk + 8 * hi);
        vf[kb][kk] = *reinterpret_cast<const u32x4*>(vb + (int64_t)kr * a.vs[2] + 16 * kk + 8 * hi);
      }
    }
  }

  // ---- RPE state in LDS (see attn_bwd.h: table copies, private diagonal accumulators) ----
  float* sT = reinterpret_cast<float*>(smem + Cfg::rpe0(stg)) + kRpePad;  // (entry d of copy 0 at sT[d + R]; see attn_common.h)
  const int n1 = 2 * a.R + 1;
  float* sD0 = sT - kRpePad + 4 * rpe_n1p(a.R);
  const uint32_t one2s = pack2<BF16>(1.f, 1.f);
  u32x4 ones = {one2s, one2s, one2s, one2s};
  // (opaque: a constant tuple is re-materialised by v_mov right in front of the asm MFMA that reads it — and no wait states are
  // generated between a VALU write and an asm consumer)
  asm volatile("" : "+v"(ones));
  // Per-diagonal sums of dS (the gradient of the bias generator) on the VALU: diag_sum.h.  One carry per key block along a run of
  // consecutive band / masked steps.  The homes of key block 1 at a step are those of key block 0 one step earlier, so the two meet
  // in registers (dprev0) and ONE value per lane and step leaves: near diagonals into the (wave, half-wave)'s private array -- every
  // diagonal exactly once, a plain store --, everything beyond the band into two per-lane sums.  Branch-free: a lane without a near
  // diagonal stores into a scratch slot of its own.
  DiagCarry dcar[2];
  diag_carry_zero(dcar[0]);
  diag_carry_zero(dcar[1]);
  float dprev0 = 0.f;
  bool diag_run = false;  // (wave-uniform) a run is open: dcar / dprev0 hold partial diagonals of the step at diag_mb
  int diag_mb = 0;
  // borrow masks of the pinned form (destination lane: position p' of its row of 16 took its value from p' + ql0 - 16 of the row above in key order)
  float dmask[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (DG) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int ql0 = i < 3 ? i + 1 : i + 5;  // 1, 2, 3, 8, 9, 10, 11
      dmask[i] = ((l & 15) + ql0 >= 16) ? 1.f : 0.f;
      asm volatile("" : "+v"(dmask[i]));
    }
  }
  float far_neg = 0.f, far_pos = 0.f;
  const bool want_drpe = DG && (a.drpe_part != nullptr);
  float* const dsum = sD0 + (2 * w + hi) * n1 + a.R - 4 * hi + lq;  // (this lane's diagonal base + lane - 4 hi at dsum[base])
  float* const dtrash = sD0 + 2 * Cfg::NW * n1 + tid;
  // E: the finished sums of the diagonals base + (lane & 31) - 4 hi
  auto diag_store = [&](const float E, const int base) {
    const int d = base + lq - 4 * hi;
    far_neg += d <= -a.R ? E : 0.f;
    far_pos += d >= a.R ? E : 0.f;
    *((d > -a.R && d < a.R) ? dsum + base : dtrash) = E;
  };
  // end of a step at query row mb: st[kb] = the step's accumulators of key block kb
  auto diag_step_end = [&](const DiagStep (&st)[2], const int mb) {
    const float F0 = diag_finish_halves(dcar[0], st[0], l), F1 = diag_finish_halves(dcar[1], st[1], l);
    diag_store(F1 + dprev0, kw0 + 32 - mb);
    dprev0 = F0;
    diag_run = true;
    diag_mb = mb;
  };
  // the end of a run: the diagonals still in the carries (the two 32-diagonal windows below the last step's) leave
  auto diag_flush = [&]() {
    if (diag_run) {
      diag_store(dcar[1].cur + dprev0, kw0 - diag_mb);
      diag_store(dcar[0].cur, kw0 - diag_mb - 32);
      diag_carry_zero(dcar[0]);
      diag_carry_zero(dcar[1]);
      dprev0 = 0.f;
      diag_run = false;
    }
  };

  f32x16 dk[2][DB], dv[2][DB];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[kb][i][r] = 0.f; dv[kb][i][r] = 0.f; }

  // query-step range: causal => only rows with q + P >= n0 see this key block
  int m_lo = 0;
  if (a.causal) m_lo = max(0, n0 - P) / 32 * 32;
  const int nst_all = (M + 31) / 32;
  int mt0 = m_lo / 32;
  int nsteps = nst_all - mt0;
  if constexpr (HALF) {
    // each pair takes half of the steps (the second half may end with one step past the last row: its Q / dO rows and its
    // statistics read as zeros through the buffer descriptors -> dS = 0, dO = 0: it contributes exactly nothing)
    const int hs = (nsteps + 1) / 2;
    mt0 += pr * hs;
    nsteps = hs;
  }

  // ---- ring: step j (query rows 32*(mt0+j) ..+31) lives in slot j % 4: [Q image | dO image | 4 private statistics pieces] ----
  using Dma = DmaStage<D, Cfg::QT, HALF ? 128 : NT>;  // (HALF: a pair's 128 threads fill the pair's ring)
  static_assert(Dma::PER == (HALF ? 2 : 1) && Dma::NV == 1, "one (HALF: two) 16-byte piece(s) per thread and image");
  Dma qst, dost;
  qst.init(a.qs[2], HALF ? (tid & 127) : tid);
  dost.init(a.dos[2], HALF ? (tid & 127) : tid);
  const __amdgpu_buffer_rsrc_t qrs = make_rows_rsrc(qb, a.qs[2], M, D);
  const __amdgpu_buffer_rsrc_t dors = make_rows_rsrc(dob, a.dos[2], M, D);
  // statistics: 64 floats per step ([32] -L/scale, [32] -delta), whole steps (the dQ kernel pads the last one); SELF: the step's 32 raw
  // log-sum-exp values (rows past M read as zeros through the descriptor) and its O rows
  const __amdgpu_buffer_rsrc_t strs =
      SELF ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.lse + (int64_t)bh * M), 0, __builtin_amdgcn_readfirstlane(M * 4), 0x00020000)
           : make_rows_rsrc(reinterpret_cast<const uint16_t*>(a.stat2 + (int64_t)bh * nst_all * 64), 128, nst_all, 128);
  const uint16_t* ob = SELF ? a.o + (int64_t)b * a.os[0] + (int64_t)h * a.os[1] : dob;
  Dma ost;
  ost.init(SELF ? a.os[2] : a.dos[2], tid);
  const __amdgpu_buffer_rsrc_t ors = make_rows_rsrc(ob, SELF ? a.os[2] : a.dos[2], M, D);
  const uint32_t qstride_b = (uint32_t)a.qs[2] * 2u, dostride_b = (uint32_t)a.dos[2] * 2u, ostride_b = (uint32_t)(SELF ? a.os[2] : a.dos[2]) * 2u;
  const uint32_t svoff = (uint32_t)(l & (SELF ? 7 : 15)) * 16u;  // (the other lanes re-read the same 128 / 256 bytes: no out-of-range reliance)
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t ring0 = (uint32_t)(pr * Cfg::RING);  // this pair's ring
  const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds0 + ring0 + (uint32_t)wp * 1024u);
  // DENSE: this wave's bias tile of a step (32 rows x its 64 keys) as a D = 64 image: four pieces of 8 rows, two swizzle phases
  using BDma = DmaStage<64, 32, 64>;
  [[maybe_unused]] BDma bst;
  [[maybe_unused]] __amdgpu_buffer_rsrc_t brs = qrs;
  [[maybe_unused]] uint32_t bias_lds = 0u, bstride_b = 0u;
  if constexpr (DENSE) {
    static_assert(BDma::PER == 4 && BDma::NV == 2, "four 1-KiB pieces of 8 rows");
    bst.init(a.bs[2], l);
    brs = make_rows_rsrc(a.bias + (int64_t)b * a.bs[0] + (int64_t)h * a.bs[1], a.bs[2], M, N);
    bias_lds = __builtin_amdgcn_readfirstlane(lds0 + ring0 + (uint32_t)(Cfg::BIASO + wp * IMG));
    bstride_b = (uint32_t)a.bs[2] * 2u;
  }
  // The LDS-DMA pieces of one step, in request order (the counted waits below rely on it): [DENSE: 4 bias pieces] PER x (Q, dO [, O]) | statistics.
  // (FAT5_DMA_SPREAD: issued one per MFMA gap by the pipelined iterations -- measured, not kept: see the macro)
  constexpr int NPIECE = (DENSE ? 4 : 0) + Dma::PER * (SELF ? 3 : 2) + 1;
  auto dma_piece = [&](const uint32_t mt, const uint32_t slot_off, const int k) {
    constexpr int NB = DENSE ? 4 : 0, NI = SELF ? 3 : 2;
    if (k < NB) {
      if constexpr (DENSE) dma16_asm(brs, bias_lds + slot_off + (uint32_t)(k * 1024), bst.voff[k % 2], mt * 32u * bstride_b + (uint32_t)kw0 * 2u + bst.piece_step * (uint32_t)(k / 2));
    } else if (k < NB + Dma::PER * NI) {
      const int i = (k - NB) / NI, which = (k - NB) % NI;
      if (which == 0) dma16_asm(qrs, wave_lds + slot_off + (uint32_t)(i * 2048), qst.voff[0], mt * 32u * qstride_b + qst.piece_step * i);
      else if (which == 1) dma16_asm(dors, wave_lds + slot_off + (uint32_t)(IMG + i * 2048), dost.voff[0], mt * 32u * dostride_b + dost.piece_step * i);
      else dma16_asm(ors, wave_lds + slot_off + (uint32_t)(2 * IMG + i * 2048), ost.voff[0], mt * 32u * ostride_b + ost.piece_step * i);
    } else {
      if constexpr (SELF) dma16_asm(strs, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + ring0)) + slot_off + (uint32_t)(3 * IMG), svoff, mt * 128u);  // (every wave the same 128 bytes into the shared piece)
      else dma16_asm(strs, wave_lds + slot_off + (uint32_t)(2 * IMG), svoff, mt * 256u);
    }
  };
  auto dma_step = [&](int j, uint32_t slot_off_) {
    const uint32_t mt = (uint32_t)__builtin_amdgcn_readfirstlane(mt0 + j);
    const uint32_t slot_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot_off_);  // (provably wave-uniform: it feeds M0)
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) dma_piece(mt, slot_off, k);
  };
  // SELF: this wave's 8 rows of the step in the slot at `so` -> -L/scale (in place of the raw L) and -delta.  Eight lanes per row, each
  // one 16-byte piece of the O and dO images (same slot of both: the images share the swizzle), a butterfly over the eight.
  const float nis_c = -1.f / a.scale;
  struct StatIn { u32x4 dov, ov; float Lr; };
  const int prow = 8 * w + (l >> 3);
  auto stats_read = [&](const uint32_t so) {  // (three LDS reads; their values are needed several gaps later)
    const uint32_t pa = lds0 + ring0 + so + (uint32_t)(prow * 2 * D + (l & 7) * 16);
    StatIn x;
    x.dov = lds_rd128(pa + (uint32_t)IMG);
    x.ov = lds_rd128(pa + (uint32_t)(2 * IMG));
    x.Lr = reinterpret_cast<const float*>(smem + pr * Cfg::RING + so + 3 * IMG)[prow];
    return x;
  };
  auto stats_value = [&](const StatIn& x, const int j) {  // even lanes: -delta of the lane's row, odd lanes: -L/scale
    float pd = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pd = fmaf(cvt_lo<BF16>(x.ov[i]), cvt_lo<BF16>(x.dov[i]), pd);
      pd = fmaf(cvt_hi<BF16>(x.ov[i]), cvt_hi<BF16>(x.dov[i]), pd);
    }
    pd = dpp_add<0xB1>(pd);   // lane ^ 1
    pd = dpp_add<0x4E>(pd);   // lane ^ 2
    pd = dpp_add<0x141>(pd);  // the other quad of each 8
    const bool live = (mt0 + j) * 32 + prow < M && !(x.Lr < kDeadRowLse);
    const float nl = live ? x.Lr * nis_c : ninf_c;
    return (l & 1) ? nl : -pd;
  };
  auto stats_write = [&](const float v, const uint32_t so) {
    reinterpret_cast<float*>(smem + pr * Cfg::RING + so + 3 * IMG)[(l & 1) ? prow : Cfg::DLOFF / 4 + prow] = v;
  };
  auto produce_stats = [&](const int j, const uint32_t so) { stats_write(stats_value(stats_read(so), j), so); };
  // E(j): step j+1 (SELF: j+2) has landed and is visible to every wave; every wave is done with step j-1, whose slot takes step j+3
  auto sync_wait = [&](int j) {
    if ((!SELF || (FAT5_ABL & 128)) && j + 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
    else wait_dma_all();
    __syncthreads();
  };
  auto sync_step = [&](int j, uint32_t slot3_off) {
    sync_wait(j);
    if (j + 3 < nsteps) dma_step(j + 3, slot3_off);
  };

  if (nsteps > 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < nsteps) dma_step(i, (uint32_t)(i * SLOT));
  }
  // slot 3 is read (against an all-zero P / dS) before anything lands in it: finite contents
#pragma unroll
  for (int rg = 0; rg < Cfg::RINGS; ++rg)
    for (int i = tid; i < SLOT / 16; i += NT) reinterpret_cast<u32x4*>(smem + rg * Cfg::RING + 3 * SLOT)[i] = u32x4{0u, 0u, 0u, 0u};
  if constexpr (BIAS == FAT5_BIAS_RPE1D && !(FAT5_ABL & 8)) {
    rpe_table_fill_rest(sT - kRpePad, a.rpe1d + (int64_t)h * n1, a.R, tid, NT, tabr, ctab ? P : 0x7fffffff);
    if constexpr (DG)
      for (int i = tid; i < n1 * 2 * Cfg::NW; i += NT) sD0[i] = 0.f;
  }
  wait_dma_all();
  __syncthreads();
  FragAddr<D> fa;
  fa.init(l);
  if (stg) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const uint32_t ad = stg_k + (uint32_t)(fa.rm[kk] + kb * 32 * 2 * D);
        kf[kb][kk] = lds_rd128(ad);
        vf[kb][kk] = lds_rd128(ad + (uint32_t)Cfg::STG_T);
      }
  }
  FAT5_STAMP(1);
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) asm volatile("" : "+a"(kf[kb][kk]), "+a"(vf[kb][kk]));  // MFMA-only operands: AGPRs

  // per-lane LDS addresses (ring base folded in; slot / image / step offsets are immediates)
  uint32_t rmA[KK], trA[2][DB], stA;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    rmA[kk] = lds0 + ring0 + (uint32_t)fa.rm[kk];
    asm volatile("" : "+v"(rmA[kk]));
  }
#pragma unroll
  for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      trA[j2][db] = lds0 + ring0 + (uint32_t)fa.tr[j2][db];
      asm volatile("" : "+v"(trA[j2][db]));
    }
  stA = lds0 + ring0 + (uint32_t)(SELF ? 3 * IMG + 16 * hi : 2 * IMG + wp * 1024 + 16 * hi);  // this lane's rows 8g + 4hi ..+3: float4 g at +32g (-L/scale), +DLOFF+32g (-delta)
  asm volatile("" : "+v"(stA));
  // DENSE: the transposing-read addresses of this wave's bias image (slot / 16-row offsets are immediates): btr[j2][kb] + 2048 t2 gives the lane the bias of
  // rows 8 jj + 4 hi + (0..3), jj = 2 t2 + j2, at its key of block kb: that read, twice, is the B operand (jj, kb) -- and the selector operands E(jj)
  [[maybe_unused]] uint32_t btr[2][2] = {{0u, 0u}, {0u, 0u}};
  [[maybe_unused]] u32x4 selA[4];
  // (the raw bias words are MFMA operands: -inf entries are clamped in their packed 16-bit form first -- -inf * 0 in the selector product would be NaN: attn_common.h)
  [[maybe_unused]] const uint32_t blim2 = bias_mfma_limit<BF16>(a.scale);
  if constexpr (DENSE) {
    // E(jj): A operand, lane = row lq; k-slot (hi, j) <-> bias row 8 jj + 4 hi + (j & 3), holding 1/scale's leading 16 bits for j < 4 and the next 16 for j >= 4
    const float invf = 1.f / a.scale;
    uint32_t ih, il;
    split16<BF16>(invf, ih, il);
#pragma unroll
    for (int jj = 0; jj < (ONE ? 2 : 4); ++jj) {
      uint32_t wv[4];
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2) {  // word j2 = k-slots 2 j2, 2 j2 + 1
        // (ONE: operand jj = t2 covers rows 16 t2 + 8 (j >> 2) + 4 hi + (j & 3), every slot 1 / scale)
        const int r0 = ONE ? 16 * jj + 8 * ((2 * j2) >> 2) + 4 * hi + ((2 * j2) & 3) : 8 * jj + 4 * hi + ((2 * j2) & 3);
        const uint32_t val = (ONE || j2 < 2) ? ih : il;
        wv[j2] = (r0 == lq ? val : 0u) | (r0 + 1 == lq ? val << 16 : 0u);
      }
      selA[jj] = u32x4{wv[0], wv[1], wv[2], wv[3]};
      asm volatile("" : "+v"(selA[jj]));
    }
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        btr[j2][kb] = trA[j2][kb] + (uint32_t)(Cfg::BIASO + wp * IMG);
        asm volatile("" : "+v"(btr[j2][kb]));
      }
  }
  if constexpr (SELF) {  // the statistics of the first two steps (step 2's follow in iteration 0)
    produce_stats(0, 0u);
    produce_stats(1, (uint32_t)SLOT);
    __syncthreads();
  }
  auto rd_f4 = [&](uint32_t addr) { return __builtin_bit_cast(f32x4, lds_rd128(addr)); };
  auto put4 = [](f32x16& x, int g, const f32x4 v) { x[4 * g] = v[0]; x[4 * g + 1] = v[1]; x[4 * g + 2] = v[2]; x[4 * g + 3] = v[3]; };
  auto rd_tr = [&](uint32_t off, int t2, int db) {
    const uint32_t o = off + (uint32_t)(16 * t2 * 2 * D);
    return lds_rd_tr(trA[0][db] + o, trA[1][db] + o);
  };

  const float c2 = a.scale * kLog2e;
  float cst_neg = 0.f, cst_pos = 0.f;
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    cst_neg = sT[0];
    cst_pos = sT[2 * a.R];
  }
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // ------------------------------------------------------------------------------------------------------------------
  // Pipeline state between two iterations (iteration i = query step i is in its softmax stage):
  //   S, DP      S' = Q K^T - L/scale and dP' = dO V^T - delta of step i (lane = key, register r <-> row crow(r, hi))
  //   PB, DS     P and dS of step i-1, rounded to bf16 like the reference (:702 / :720), as B operands
  //   TRD, TRQ   first transposed fragments (t2 = 0, db = 0) of step i-1's dO and Q images
  // ------------------------------------------------------------------------------------------------------------------
  f32x16 S[2], DP[2];
  u32x4 PB[2][2], DS[2][2], TRD, TRQ;
  // band steps of the pipelined loop: the first table entries (key block 0, rows 0..3) of the step whose softmax is due and the
  // address of that window (both formed one iteration ahead), and this lane's addressing of its padded table copy: entry of row mb + crow(r, hi), r = 4 gg + i, is component
  // 3 - i of the 16 bytes at tab_addr(kb, mb) - 32 gg (the window runs DOWN with the row; see softmax_generic)
  u32x4 TN0;
  uint32_t tadr0 = 0u;
  uint32_t tabB[2] = {0u, 0u};
  int tpos[2] = {0, 0};
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int krow = kw0 + 32 * kb + lq, al = (a.R + krow - 3) & 3;
      tabB[kb] = (uint32_t)(uintptr_t)(sT + al * rpe_n1p(a.R));
      tpos[kb] = a.R + krow - 4 * hi - 3 - al;
    }
  }
  const int tclamp_lo = -kRpePad + 24, tclamp_hi = rpe_n1p(a.R) - kRpePad - 4;  // (rpe_clamp_desc)
  auto tab_addr = [&](int kb, int mb) { return tabB[kb] + 4u * (uint32_t)min(max(tpos[kb] - mb, tclamp_lo), tclamp_hi); };
  // sum of the dS of a pipelined range (one far bin) on the matrix pipe: ones(16x32) . dS words as a 32x16 B operand; every row of
  // the 16x16 result = the column sums, so the sum over lanes and registers is 16x the sum of all the words' elements, whatever
  // their layout (one v_dot2c_f32_bf16 per word instead measured 9 % of this kernel)
  [[maybe_unused]] f32x4 facc4 = {0.f, 0.f, 0.f, 0.f};

  // The causal mask as part of the score MFMAs' C operand (round 5; bias none / dense -- with the T5 bias the table carries it): S' = Q K^T + C with
  // C = -L/scale of the row, or -inf (+inf for a negative scale) where the key is masked: exp2(-inf) = 0 makes p and dS exact zeros, so a step on the
  // causal diagonal runs the PIPELINED iteration (one compare + one select per element of the C operand) instead of the unpipelined general one --
  // half of all steps of a 256-key workgroup at 1024 keys.  Key k0 + lane is visible to row mb + c(r) + 4 hi iff c(r) >= t with t = key - P - mb - 4 hi.
  constexpr bool CMASK = BIAS != FAT5_BIAS_RPE1D;
  const int tbase = kw0 + lq - P - 4 * hi;
  auto mask_c = [&](const f32x16& nl, const int t, f32x16& out) {
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r] = ((r & 3) + 8 * (r >> 2) < t) ? ninf_c : nl[r];
  };
  // scores of the step (query rows mb ..) in the slot at byte offset `so`
  auto score_step = [&](const uint32_t so, f32x16 (&Sx)[2], f32x16 (&DPx)[2], const int mb) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        put4(Sx[kb], g, rd_f4(stA + so + (uint32_t)(32 * g)));
        put4(DPx[kb], g, rd_f4(stA + so + (uint32_t)(Cfg::DLOFF + 32 * g)));
      }
    if (CMASK && a.causal) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) mask_c(Sx[kb], tbase + 32 * kb - mb, Sx[kb]);
    }
    u32x4 qa[KK], da[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      qa[kk] = lds_rd128(rmA[kk] + so);
      da[kk] = lds_rd128(rmA[kk] + so + (uint32_t)IMG);
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) Sx[kb] = mfma32<BF16>(qa[kk], kf[kb][kk], Sx[kb]);
    if constexpr (DENSE) {  // + bias / scale (see Bwd64Cfg)
      if constexpr (ONE) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
            Sx[kb] = mfma32<BF16>(selA[t2], bias_clamp_frag(lds_rd_tr(btr[0][kb] + so + (uint32_t)(2048 * t2), btr[1][kb] + so + (uint32_t)(2048 * t2)), blim2), Sx[kb]);
      } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            const u32x2 bh_ = bias_clamp_frag(lds_rd_tr_half(btr[jj & 1][kb] + so + (uint32_t)(2048 * (jj >> 1))), blim2);
            Sx[kb] = mfma32<BF16>(selA[jj], u32x4{bh_[0], bh_[1], bh_[0], bh_[1]}, Sx[kb]);
          }
      }
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) DPx[kb] = mfma32<BF16>(da[kk], vf[kb][kk], DPx[kb]);
  };
  // dV^T += dO^T P, dK^T += Q^T dS of the pending step, whose images are in the slot at `so`
  auto product_step = [&](const uint32_t so) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const u32x4 dot = rd_tr(so + (uint32_t)IMG, t2, db), qt = rd_tr(so, t2, db);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) mfma_acc_agpr<BF16>(dv[kb][db], dot, PB[kb][t2]);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) mfma_acc_agpr<BF16>(dk[kb][db], qt, DS[kb][t2]);
      }
  };

  // general softmax stage of the step at query row mb: S, DP -> PB, DS (+ per-diagonal sums of dS)
  auto softmax_generic = [&](const int mb, [[maybe_unused]] const uint32_t so) {
    DiagStep gst[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16& s = S[kb];
      const f32x16& dp = DP[kb];
      const int k0 = kw0 + 32 * kb, krow = k0 + lq;
      if constexpr (DENSE) {
        // (the scores arrive with the bias inside: score_step / the pipelined iteration add it on the matrix pipe)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] *= c2;
      } else if constexpr (BIAS == FAT5_BIAS_RPE1D) {
        // far, edge and band blocks alike: four aligned 16-byte reads of this lane's padded table copy, window clamped (attn_common.h)
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
      f32x16 p;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = fast_exp2(s[r]);
        s[r] = p[r] * dp[r];
      }
      const bool nmask = k0 + 32 > N;
      const bool cmask = a.causal && (k0 + 31 > mb + P);
      if (nmask || cmask) {
        // key visible to query m iff krow < N and (causal) krow <= m + P  <=>  crow(r, hi) >= thr
        int thr = a.causal ? (krow - P - mb) : -(1 << 30);
        if (krow >= N) thr = 1 << 30;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool ok = crow(r, hi) >= thr;
          p[r] = ok ? p[r] : 0.f;
          s[r] = ok ? s[r] : 0.f;
        }
      }
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        PB[kb][t2] = pack8<BF16>(p, t2);
        DS[kb][t2] = pack8<BF16>(s, t2);
      }
      if constexpr (DG) {
        // the block's dS (fp32, masked elements zero) onto its diagonals; whole blocks beyond the band included
        diag_step_zero(gst[kb]);
        static_for<16>([&](auto ri) { diag_elem<decltype(ri)::value>(gst[kb], s[decltype(ri)::value], l & 15); });
      }
    }
    if constexpr (DG) diag_step_end(gst, mb);
  };

  // the stages of one iteration one after the other (band / masked steps)
  auto generic_iter = [&](const int j) {
    const uint32_t o_prev = (uint32_t)(((j + 3) & 3) * SLOT), o_cur = (uint32_t)((j & 3) * SLOT), o_next = (uint32_t)(((j + 1) & 3) * SLOT);
    product_step(o_prev);
    sync_step(j, o_prev);
    if constexpr (SELF) produce_stats(j + 2, (uint32_t)(((j + 2) & 3) * SLOT));
    f32x16 Sn[2], DPn[2];
    score_step(o_next, Sn, DPn, (mt0 + j + 1) * 32);
    softmax_generic((mt0 + j) * 32, o_cur);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      S[kb] = Sn[kb];
      DP[kb] = DPn[kb];
    }
    TRD = rd_tr(o_cur + (uint32_t)IMG, 0, 0);
    TRQ = rd_tr(o_cur, 0, 0);
  };

  // One pipelined iteration = 32 MFMA gaps.  Gap g holds, all mutually independent:
  //   MFMA   g < 16: products of step i-1 -- per (t2, db): dV^T[kb0], dV^T[kb1] (fragment dO^T), dK^T[kb0], dK^T[kb1] (Q^T);
  //          16..23: S'[kb] of step i+1 (k-steps outer, C = -L/scale on the first); 24..31: dP'[kb] (C = -delta)
  //   VALU   element g of step i (key block g >> 4, register g & 15): x = s*c2 + cst | element g-1: p = exp2(x) |
  //          element g-2: ds = p*dp' | even g: elements g-4, g-3 packed to bf16 (P and dS words), dS summed on the matrix pipe (`facc4`)
  //   LDS    gaps 0..11 the transposed fragments of the remaining three (t2, db) pairs of step i-1; gap 12 the barrier
  //          E(i) + the DMA of step i+3; gaps 12..22 statistics (straight into the accumulators of S', dP': the MFMA C operand)
  //          and row-major fragments of step i+1; gaps 28..31 the first
  //          transposed fragments of step i (consumed by the next iteration's gaps 0..3)
  // The VALU ops are volatile asm (see attn_fwd64.h); the youngest MFMA results they read are dP'[0] (finished by MFMA 30 of
  // the previous iteration, first read in gap 2) and dP'[1] (MFMA 31, first read in gap 18).
  // BAND (round 3): the same iteration for steps that cross the T5 band (all keys visible): the bias of element (q, k) is the FMA's
  // addend, read from this lane's padded table copy (attn_common.h) -- four 16-byte reads per key block, each issued three gaps ahead of
  // its FMAs (the first one during the previous iteration: TN0) -- and the per-diagonal sums of the step's rounded dS follow the
  // iteration as a block (diag_sums on DS).  A band step was the general, unpipelined iteration before: ~2.9x a pipelined step.
  // DN (round 5): the same iteration with a DENSE bias: the scores of step j+1 get + bias / scale from eight more MFMAs (Bwd64Cfg) -- gaps 16..23 hold two
  // MFMAs each, key block 0 at the head of the gap and key block 1 behind its LDS section (the two accumulators alternate strictly): k-steps 0..3, then
  // the four 8-row groups of the bias --, their B operands from eight transposing reads of the wave's bias image of step j+1 in gaps 13..19; the softmax
  // stage is the one without bias
  // MK (round 5): the scores of step j+1 are formed with the causal mask in their C operand (mask_c): a trip on the diagonal
  auto fast_iter = [&]<int SL, bool BAND, bool DN = false, bool MK = false>(const int j, const float cst) {
    constexpr uint32_t o_prev = ((SL + 3) & 3) * SLOT, o_cur = SL * SLOT, o_next = ((SL + 1) & 3) * SLOT;
    u32x4 T[2][4];
    [[maybe_unused]] u32x2 bbh[2][4];  // [kb][jj]: B operands of the bias MFMAs (each read serves both halves of its operand)
    uint32_t tadr1 = 0u;
    if constexpr (BAND) T[0][0] = TN0;
    // BAND: the step's dS onto its diagonals (diag_sum.h), element e three gaps after its exponent argument: one rotating add for
    // everything, one rotating multiply-add by the borrow mask (both read Dv[e], written one gap earlier: no DPP hazard)
    [[maybe_unused]] DiagStep dst[2];
    if constexpr (BAND && DG) {
      diag_step_zero(dst[0]);
      diag_step_zero(dst[1]);
    }
    [[maybe_unused]] StatIn sin_;
    [[maybe_unused]] float sval_ = 0.f;
    f32x16 Sn[2], DPn[2];
    [[maybe_unused]] f32x16 NL, DL;
    [[maybe_unused]] f32x16 NLm[2];
    [[maybe_unused]] int tm[2] = {0, 0};
    u32x4 PBn[2][2], DSn[2][2], qa[KK], da[KK];
    u32x2 trh[4][2][2], tnd[2], tnq[2];
    float X[32], Pv[32], Dv[32];
    auto diag_el = [&]<int E>() {
      constexpr int r = E & 15, ql0 = diag_ql0(r);
      diag_elem_u<r>(dst[E >> 4], Dv[E]);
      if constexpr (ql0 != 0) diag_elem_bm<r>(dst[E >> 4], Dv[E], dmask[ql0 < 8 ? ql0 - 1 : ql0 - 5]);
    };
    auto pack_pair = [&]<int E0>() {
      constexpr int kb = E0 >> 4, r0 = E0 & 15, t2 = r0 >> 3, wd = (r0 & 7) >> 1;
      PBn[kb][t2][wd] = asm_cvt_pk<BF16>(Pv[E0], Pv[E0 + 1]);
      const uint32_t dsw = asm_cvt_pk<BF16>(Dv[E0], Dv[E0 + 1]);
      DSn[kb][t2][wd] = dsw;
    };
    static_for<32>([&](auto gi) {
      constexpr int g = decltype(gi)::value;
      // ---- MFMA ----
      if constexpr (g < 16) {
        constexpr int p = g >> 2, t2 = p >> 1, db = p & 1, jj = g & 3, kb = jj & 1, wh = jj >> 1;
        u32x4 fr;
        if constexpr (p == 0) fr = wh == 0 ? TRD : TRQ;
        else fr = u32x4{trh[p][wh][0][0], trh[p][wh][0][1], trh[p][wh][1][0], trh[p][wh][1][1]};
        if constexpr (wh == 0) mfma_acc_agpr<BF16>(dv[kb][db], fr, PB[kb][t2]);
        else mfma_acc_agpr<BF16>(dk[kb][db], fr, DS[kb][t2]);
      } else if constexpr (g < 24 && DN && ONE && g >= 20) {
        // (ONE: gaps 16..19 one k-step each, like the no-bias iteration; gaps 20..23 two MFMAs: k-steps 2, 3, then the two 16-row bias operands)
        if constexpr (g < 22) Sn[0] = mfma32<BF16>(qa[g - 18], kf[0][g - 18], Sn[0]);
        else Sn[0] = mfma32<BF16>(selA[g - 22], u32x4{bbh[0][2 * (g - 22)][0], bbh[0][2 * (g - 22)][1], bbh[0][2 * (g - 22) + 1][0], bbh[0][2 * (g - 22) + 1][1]}, Sn[0]);
      } else if constexpr (g < 24 && DN && !ONE) {
        // (the first of the gap's two MFMAs: key block 0; the second one follows the gap's LDS section)
        if constexpr (g == 16) Sn[0] = mfma32<BF16>(qa[0], kf[0][0], MK ? NLm[0] : NL);
        else if constexpr (g < 20) Sn[0] = mfma32<BF16>(qa[g - 16], kf[0][g - 16], Sn[0]);
        else Sn[0] = mfma32<BF16>(selA[g - 20], u32x4{bbh[0][g - 20][0], bbh[0][g - 20][1], bbh[0][g - 20][0], bbh[0][g - 20][1]}, Sn[0]);
      } else if constexpr (g < 24) {
        constexpr int kk = (g - 16) >> 1, kb = g & 1;
        if constexpr (kk == 0) Sn[kb] = mfma32<BF16>(qa[kk], kf[kb][kk], MK ? NLm[kb] : NL);
        else Sn[kb] = mfma32<BF16>(qa[kk], kf[kb][kk], Sn[kb]);  // (accumulator preloaded with -L/scale)
      } else {
        constexpr int kk = (g - 24) >> 1, kb = g & 1;
        if constexpr (kk == 0) DPn[kb] = mfma32<BF16>(da[kk], vf[kb][kk], DL);
        else DPn[kb] = mfma32<BF16>(da[kk], vf[kb][kk], DPn[kb]);  // (preloaded with -delta)
      }
      __builtin_amdgcn_sched_barrier(0);  // (the MFMA opens its gap)
      // (SELF: the statistics of step j+2, which has landed behind E(j) and is read two iterations from now: reads, arithmetic and the
      //  store in gaps of their own)
      if constexpr (SELF && g == 13 && !(FAT5_ABL & 64)) sin_ = stats_read(((SL + 2) & 3) * SLOT);
      if constexpr (SELF && g == 23 && !(FAT5_ABL & 64)) sval_ = stats_value(sin_, j + 2);
      if constexpr (SELF && g == 27 && !(FAT5_ABL & 64)) stats_write(sval_, ((SL + 2) & 3) * SLOT);
      if constexpr (g == 19) asm volatile("" ::"v"(NL));  // (keeps NL's registers out of reach of the VALU ops of gaps 16..18)
      if constexpr (MK && g == 20) asm volatile("" ::"v"(NLm[0]), "v"(NLm[1]));
      if constexpr (MK) {  // the masked C operands of step j+1's scores: key block 0 in gaps 13, 14 (first MFMA: gap 16), key block 1 in gaps 15, 16 (gap 17)
        if constexpr (g == 13) {
          tm[0] = tbase - (mt0 + j + 1) * 32;
          tm[1] = tm[0] + 32;
        }
        if constexpr (DN) {  // (two MFMAs per gap from gap 16 on: both masks by then)
          if constexpr (g >= 13 && g <= 15) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if ((g == 13 && r < 8) || (g == 14 && r >= 8)) NLm[0][r] = ((r & 3) + 8 * (r >> 2) < tm[0]) ? ninf_c : NL[r];
              if ((g == 14 && r < 8) || (g == 15 && r >= 8)) NLm[1][r] = ((r & 3) + 8 * (r >> 2) < tm[1]) ? ninf_c : NL[r];
            }
          }
        } else if constexpr (g >= 13 && g <= 16) {
          constexpr int kb = (g - 13) >> 1, r0 = 8 * ((g - 13) & 1);
#pragma unroll
          for (int r = r0; r < r0 + 8; ++r) NLm[kb][r] = ((r & 3) + 8 * (r >> 2) < tm[kb]) ? ninf_c : NL[r];
        }
      }
      if constexpr (g == 27) asm volatile("" ::"v"(DL));
      // ---- barrier + DMA (step j+3 into the slot of step j-1, one piece per gap) ----
      if constexpr (g == 12) {
        if constexpr (FAT5_DMA_SPREAD) sync_wait(j);
        else sync_step(j, o_prev);
      }
      if constexpr (FAT5_DMA_SPREAD && g > 12 && g - 13 < NPIECE) {
        if (j + 3 < nsteps) dma_piece((uint32_t)__builtin_amdgcn_readfirstlane(mt0 + j + 3), o_prev, g - 13);
      }
      // ---- LDS ----
      if constexpr (g < 12) {
        constexpr int pp = (g >> 2) + 1, t2 = pp >> 1, db = pp & 1, wh = (g >> 1) & 1, half = g & 1;
        trh[pp][wh][half] = lds_rd_tr_half(trA[half][db] + o_prev + (uint32_t)((wh == 0 ? IMG : 0) + 16 * t2 * 2 * D));
      } else if constexpr (g == 12) {
        put4(NL, 0, rd_f4(stA + o_next));
        put4(NL, 1, rd_f4(stA + o_next + 32u));
        put4(NL, 2, rd_f4(stA + o_next + 64u));
        put4(NL, 3, rd_f4(stA + o_next + 96u));
      } else if constexpr (g >= 13 && g <= 16) {
        qa[g - 13] = lds_rd128(rmA[g - 13] + o_next);
      } else if constexpr (g == 17) {
        put4(DL, 0, rd_f4(stA + o_next + (uint32_t)Cfg::DLOFF));
        put4(DL, 1, rd_f4(stA + o_next + (uint32_t)(Cfg::DLOFF + 32)));
        put4(DL, 2, rd_f4(stA + o_next + (uint32_t)(Cfg::DLOFF + 64)));
        put4(DL, 3, rd_f4(stA + o_next + (uint32_t)(Cfg::DLOFF + 96)));
      } else if constexpr (g >= 19 && g <= 22) {
        da[g - 19] = lds_rd128(rmA[g - 19] + o_next + (uint32_t)IMG);
      } else if constexpr (g >= 28) {
        constexpr int half = g & 1;
        if constexpr (g < 30) tnd[half] = lds_rd_tr_half(trA[half][0] + o_cur + (uint32_t)IMG);
        else tnq[half] = lds_rd_tr_half(trA[half][0] + o_cur);
      }
      if constexpr (DN) {
        // bias fragments of step j+1 (its tile has landed since E(j)): row group jj of key block kb -- kb 0 in gaps 13, 14, 16, 17; kb 1 in 14, 15, 18, 19
        if constexpr (g == 13) bbh[0][0] = lds_rd_tr_half(btr[0][0] + o_next);
        else if constexpr (g == 14) {
          bbh[0][1] = lds_rd_tr_half(btr[1][0] + o_next);
          bbh[1][0] = lds_rd_tr_half(btr[0][1] + o_next);
        } else if constexpr (g == 15) bbh[1][1] = lds_rd_tr_half(btr[1][1] + o_next);
        else if constexpr (g == 16) bbh[0][2] = lds_rd_tr_half(btr[0][0] + o_next + 2048u);
        else if constexpr (g == 17) bbh[0][3] = lds_rd_tr_half(btr[1][0] + o_next + 2048u);
        else if constexpr (g == 18) bbh[1][2] = lds_rd_tr_half(btr[0][1] + o_next + 2048u);
        else if constexpr (g == 19) bbh[1][3] = lds_rd_tr_half(btr[1][1] + o_next + 2048u);
        // ... clamped three gaps after their reads (two v_pk_min_u16 per fragment), ahead of the MFMAs of gaps 20..23 that take them
        if constexpr (g == 16) bbh[0][0] = bias_clamp_frag(bbh[0][0], blim2);
        else if constexpr (g == 17) {
          bbh[0][1] = bias_clamp_frag(bbh[0][1], blim2);
          bbh[1][0] = bias_clamp_frag(bbh[1][0], blim2);
        } else if constexpr (g == 18) bbh[1][1] = bias_clamp_frag(bbh[1][1], blim2);
        else if constexpr (g == 19) bbh[0][2] = bias_clamp_frag(bbh[0][2], blim2);
        else if constexpr (g == 20) bbh[0][3] = bias_clamp_frag(bbh[0][3], blim2);
        else if constexpr (g == 21) bbh[1][2] = bias_clamp_frag(bbh[1][2], blim2);
        else if constexpr (g == 22) bbh[1][3] = bias_clamp_frag(bbh[1][3], blim2);
        // the gap's second MFMA: key block 1
        if constexpr (!ONE && g >= 16 && g < 24) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (g == 16) Sn[1] = mfma32<BF16>(qa[0], kf[1][0], MK ? NLm[1] : NL);
          else if constexpr (g < 20) Sn[1] = mfma32<BF16>(qa[g - 16], kf[1][g - 16], Sn[1]);
          else Sn[1] = mfma32<BF16>(selA[g - 20], u32x4{bbh[1][g - 20][0], bbh[1][g - 20][1], bbh[1][g - 20][0], bbh[1][g - 20][1]}, Sn[1]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (ONE && g >= 20 && g < 24) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (g < 22) Sn[1] = mfma32<BF16>(qa[g - 18], kf[1][g - 18], Sn[1]);
          else Sn[1] = mfma32<BF16>(selA[g - 22], u32x4{bbh[1][2 * (g - 22)][0], bbh[1][2 * (g - 22)][1], bbh[1][2 * (g - 22) + 1][0], bbh[1][2 * (g - 22) + 1][1]}, Sn[1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // ---- VALU ----
      {
      if constexpr (g >= 4 && (g & 1) == 0) pack_pair.template operator()<g - 4>();
      if constexpr (g >= 2) Dv[g - 2] = asm_mul(Pv[g - 2], DP[(g - 2) >> 4][(g - 2) & 15]);
      if constexpr (g >= 1) Pv[g - 1] = asm_exp2(X[g - 1]);
      if constexpr (BAND && !(FAT5_ABL & 4)) X[g] = asm_fma(S[g >> 4][g & 15], c2, __uint_as_float(T[g >> 4][(g & 15) >> 2][3 - (g & 3)]));
      else X[g] = asm_fma(S[g >> 4][g & 15], c2, cst);
      if constexpr (BAND && !(FAT5_ABL & 4)) {  // table entries: key block kb, rows 4 gg .. 4 gg + 3 are first used in gap 16 kb + 4 gg
        if constexpr (g == 1) T[0][1] = lds_rd128(tadr0 - 32u);
        else if constexpr (g == 5) T[0][2] = lds_rd128(tadr0 - 64u);
        else if constexpr (g == 9) T[0][3] = lds_rd128(tadr0 - 96u);
        else if constexpr (g == 10) tadr1 = tab_addr(1, (mt0 + j) * 32);
        else if constexpr (g == 13) T[1][0] = lds_rd128(tadr1);
        else if constexpr (g == 18) T[1][1] = lds_rd128(tadr1 - 32u);
        else if constexpr (g == 21) T[1][2] = lds_rd128(tadr1 - 64u);
        else if constexpr (g == 25) T[1][3] = lds_rd128(tadr1 - 96u);
        else if constexpr (g == 28) tadr0 = tab_addr(0, (mt0 + j + 1) * 32);  // the next step: rows 32 further down, window 32 entries lower
        else if constexpr (g == 29) TN0 = lds_rd128(tadr0);
      }
      if constexpr (BAND && DG && g >= 3 && !(FAT5_ABL & 1)) diag_el.template operator()<g - 3>();
      if constexpr (g == 31) {  // the tail of the step: its last elements finish inside this iteration (dependent ops back to back)
        Pv[31] = asm_exp2(X[31]);
        Dv[30] = asm_mul(Pv[30], DP[1][14]);
        pack_pair.template operator()<28>();
        Dv[31] = asm_mul(Pv[31], DP[1][15]);
        pack_pair.template operator()<30>();
        if constexpr (BAND && DG && !(FAT5_ABL & 1)) {
          diag_el.template operator()<29>();
          diag_el.template operator()<30>();
          asm volatile("s_nop 1" ::: "memory");  // (Dv[31] was written two instructions ago: DPP operands need two wait states)
          diag_el.template operator()<31>();
        }
      }
      }
      // far-bin sum of the step's dS: one 16x16x32 MFMA (16 cycles of the pipe, inside the gap's slack) per four packed words once they
      // are complete — 16 v_dot2c_f32_bf16 per step measured 9 % of this kernel. The last group's words come from the asm ops just
      // above (no hazard padding for asm producers: two wait states by hand)
      if constexpr (DG && !BAND && !(FAT5_ABL & 32) && (g == 12 || g == 20 || g == 28 || g == 31)) {
        constexpr int grp = g == 31 ? 3 : (g - 12) >> 3;
        if constexpr (g == 31) asm volatile("s_nop 1" ::: "memory");
        // (asm, accumulating in place: a builtin may pick a fresh destination, and a C operand that is not the destination is still
        // being read when the asm VALU ops behind it — which get no hazard padding — overwrite it)
        if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(facc4) : "v"(ones), "v"(DSn[grp >> 1][grp & 1]));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(facc4) : "v"(ones), "v"(DSn[grp >> 1][grp & 1]));
      }
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      S[kb] = Sn[kb];
      DP[kb] = DPn[kb];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        PB[kb][t2] = PBn[kb][t2];
        DS[kb][t2] = DSn[kb][t2];
      }
    }
    TRD = u32x4{tnd[0][0], tnd[0][1], tnd[1][0], tnd[1][1]};
    TRQ = u32x4{tnq[0][0], tnq[0][1], tnq[1][0], tnq[1][1]};
    if constexpr (BAND && DG && !(FAT5_ABL & 2)) {
      const int mb = (mt0 + j) * 32;
      asm volatile("s_nop 1" : "+v"(dst[1].u1), "+v"(dst[1].b1));  // (asm producers: see above)
      diag_step_end(dst, mb);
    }
  };

  if (nsteps > 0) {
    // fill: scores of the first step, nothing pending
    score_step(0u, S, DP, mt0 * 32);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) { PB[kb][t2] = zero4; DS[kb][t2] = zero4; }
    TRD = zero4;
    TRQ = zero4;
    FAT5_STAMP(2);

    // this wave's 64 keys x the step's 32 rows: all visible and one constant bias?  (wave-uniform; rows past M contribute
    // nothing by their statistics).  side: +1 far-positive (k - q >= R), -1 far-negative / no bias
    auto classify = [&](const int j, int& side) {
      const int mb = (mt0 + j) * 32;
      bool fast = kw0 + 64 <= N && (!a.causal || kw0 + 63 <= mb + P);
      side = -1;
      if constexpr (BIAS == FAT5_BIAS_RPE1D) {
        const bool fpos = kw0 - (mb + 31) >= a.R, fneg = kw0 + 63 - mb <= -a.R;
        fast = fast && (fpos || fneg);
        side = fpos ? 1 : -1;
      }
      return fast;
    };
    // all of the wave's 64 keys visible to all 32 rows of the step (no key tail, no causal mask)?  Monotone in j: later steps see more
    auto all_visible = [&](const int j) { return kw0 + 64 <= N && (!a.causal || ctab || kw0 + 63 <= (mt0 + j) * 32 + P); };  // (ctab: the table masks)
    // (single-body inner loops, not one loop over `class ? A : B`: the register allocator keeps one assignment per loop and pays its
    //  copies only at the few transitions)
    int j = 0;
    if constexpr (BIAS != FAT5_BIAS_RPE1D) {
      while (j < nsteps) {
        // trips of four steps that see every key run the pipelined dense iteration (visibility is monotone in j); everything else -- the causal
        // diagonal, a key tail, the remainder of the sweep -- the general one
        // (causal: trips on the diagonal first -- the mask rides in the score MFMAs' C operand --, then the all-visible ones; a key tail stays general)
        while (j + 4 <= nsteps && (j & 3) == 0 && kw0 + 64 <= N && !all_visible(j)) {
          static_for<4>([&](auto si) { fast_iter.template operator()<decltype(si)::value, false, DENSE, true>(j + decltype(si)::value, 0.f); });
          j += 4;
        }
        while (j + 4 <= nsteps && (j & 3) == 0 && all_visible(j)) {
          static_for<4>([&](auto si) { fast_iter.template operator()<decltype(si)::value, false, DENSE>(j + decltype(si)::value, 0.f); });
          j += 4;
        }
        if (j < nsteps && !(j + 4 <= nsteps && (j & 3) == 0 && kw0 + 64 <= N)) {
          generic_iter(j);
          ++j;
        }
      }
    }
    while (j < nsteps) {
      int side, side3;
      // steady state: four steps (ring slots 0..3) per trip, straight-line; the fast steps of one side are contiguous, so the
      // first and the last step of a trip decide for all four.  Steps that do not fill an aligned trip run the general iteration.
      while (j + 4 <= nsteps && (j & 3) == 0 && classify(j, side) && classify(j + 3, side3) && side3 == side) {
        const float cst = BIAS == FAT5_BIAS_RPE1D ? (side > 0 ? cst_pos : cst_neg) : 0.f;
        if constexpr (DG) diag_flush();
        static_for<4>([&](auto si) { fast_iter.template operator()<decltype(si)::value, false>(j + decltype(si)::value, cst); });
        j += 4;
        if constexpr (DG) {
          asm volatile("s_nop 15" : "+v"(facc4));  // (asm MFMA -> VALU read of its result: no padding is generated; tied to the tuple so that no read moves above it)
          const float fsum = ((facc4[0] + facc4[1]) + (facc4[2] + facc4[3])) * 0.0625f;
          facc4 = f32x4{0.f, 0.f, 0.f, 0.f};
          if (side > 0) far_pos += fsum; else far_neg += fsum;
        }
      }
      if constexpr (BIAS == FAT5_BIAS_RPE1D) {
        // trips that touch the band but see every key (band-mode iterations)
        while (j + 4 <= nsteps && (j & 3) == 0 && all_visible(j) && !(classify(j, side) && classify(j + 3, side3) && side3 == side)) {
          // a trip that touches the band: the first table entries of its first step, then four band-mode iterations (the carry chain of
          // the diagonal sums simply continues: no flush)
          tadr0 = tab_addr(0, (mt0 + j) * 32);
          TN0 = lds_rd128(tadr0);
          static_for<4>([&](auto si) { fast_iter.template operator()<decltype(si)::value, true>(j + decltype(si)::value, 0.f); });
          j += 4;
        }
      }
      if constexpr (BIAS == FAT5_BIAS_RPE1D) {
        // (after a band trip the far side may follow at once: no general iteration in between, it would break the trips' alignment)
        if (j < nsteps && !(j + 4 <= nsteps && (j & 3) == 0 && classify(j, side) && classify(j + 3, side3) && side3 == side)) {
          generic_iter(j);
          ++j;
        }
      } else {
        if (j < nsteps) {
          generic_iter(j);
          ++j;
        }
      }
    }
    // drain: the products of the last step
    FAT5_STAMP(3);
    product_step((uint32_t)(((nsteps - 1) & 3) * SLOT));
  }

  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (asm MFMA -> accumulator reads below: see mfma_acc_agpr)
  FAT5_STAMP(4);
  // ---- partial per-diagonal sums of this key block.  Staged 256-key form: AFTER the dK / dV rows are on their way (the stores drain
  // while the partial rows are summed; the images and the diagonal arrays are different LDS areas) ----
  auto partial_rows = [&]() {
    if constexpr (DG && !(FAT5_ABL & 16)) {
      if (want_drpe) {
        diag_flush();
        far_neg = wave_sum(far_neg);
        far_pos = wave_sum(far_pos);
        if (l == 0) {
          sD0[(2 * w) * n1] += far_neg;
          sD0[(2 * w) * n1 + 2 * a.R] += far_pos;
        }
        // (LDS only: a __syncthreads here would also wait for the dK / dV stores when they are already in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float* out = a.drpe_part + ((int64_t)bh * a.part_stride + part_row) * n1;
        for (int i2 = tid; i2 < n1; i2 += NT) {
          float acc = 0.f;
#pragma unroll
          for (int ww = 0; ww < 2 * Cfg::NW; ++ww) acc += sD0[ww * n1 + i2];
          out[i2] = acc;
          if (part_zero_next) out[n1 + i2] = 0.f;
        }
      }
    }
  };
  const bool stores_first = stg && !HALF;
  if (!stores_first) partial_rows();

  FAT5_STAMP(5);
  if constexpr (HALF) {
    // ---- the second pair hands its dK^T / dV^T over (lane-major 16-byte pieces: conflict-free), the first adds and stores ----
    __syncthreads();  // every wave is done with the rings
    char* mg = smem + wp * (128 * 64 * 4);
    if (pr == 1) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int i = ((kb * DB + db) * 4 + g) * 2;
            *reinterpret_cast<f32x4*>(mg + (i * 64 + l) * 16) = f32x4{dk[kb][db][4 * g], dk[kb][db][4 * g + 1], dk[kb][db][4 * g + 2], dk[kb][db][4 * g + 3]};
            *reinterpret_cast<f32x4*>(mg + ((i + 1) * 64 + l) * 16) = f32x4{dv[kb][db][4 * g], dv[kb][db][4 * g + 1], dv[kb][db][4 * g + 2], dv[kb][db][4 * g + 3]};
          }
    }
    __syncthreads();
    if (pr == 1) return;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int i = ((kb * DB + db) * 4 + g) * 2;
          const f32x4 xk = *reinterpret_cast<const f32x4*>(mg + (i * 64 + l) * 16), xv = *reinterpret_cast<const f32x4*>(mg + ((i + 1) * 64 + l) * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dk[kb][db][4 * g + e] += xk[e];
            dv[kb][db][4 * g + e] += xv[e];
          }
        }
  }
  const float scale = a.scale;
  // DENSE (round 5): the ring with the bias images leaves no room for staging images behind it -- the outputs leave through the (drained) ring
  // itself, behind one more barrier (another wave's last product reads may still be in flight)
  const bool stg_out = stg || DENSE;
  if (DENSE && !stg) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (stg_out) {
    // dK | dV through the images of this wave's keys (their fragments have long been read; HALF: the other pair returned above and its
    // last LDS access is behind the hand-over's barrier): 8-byte pieces into the swizzled row-major images, out again as whole rows
    char* img = smem + (stg ? Cfg::RINGB : 0) + wp * 2 * Cfg::STG_T;
    static_assert(!DENSE || Cfg::NW * 2 * Cfg::STG_T <= Cfg::RING, "the output images fit the ring");
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int row = 32 * kb + lq;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 wk, wv;
          wk[0] = pack2<BF16>(dk[kb][db][4 * g + 0] * scale, dk[kb][db][4 * g + 1] * scale);
          wk[1] = pack2<BF16>(dk[kb][db][4 * g + 2] * scale, dk[kb][db][4 * g + 3] * scale);
          wv[0] = pack2<BF16>(dv[kb][db][4 * g + 0], dv[kb][db][4 * g + 1]);
          wv[1] = pack2<BF16>(dv[kb][db][4 * g + 2], dv[kb][db][4 * g + 3]);
          *reinterpret_cast<u32x2*>(img + rm_off<D>(row, 4 * db + g) + 8 * hi) = wk;
          *reinterpret_cast<u32x2*>(img + Cfg::STG_T + rm_off<D>(row, 4 * db + g) + 8 * hi) = wv;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 8 * i + (l >> 3), slot = l & 7;
      const u32x4 k4 = *reinterpret_cast<const u32x4*>(img + row * (2 * D) + slot * 16);
      const u32x4 v4 = *reinterpret_cast<const u32x4*>(img + Cfg::STG_T + row * (2 * D) + slot * 16);
      if (kw0 + row < N) {
        *reinterpret_cast<u32x4*>(dkb + (int64_t)(kw0 + row) * a.dks[2] + ((slot ^ swz<D>(row)) << 3)) = k4;
        *reinterpret_cast<u32x4*>(dvb + (int64_t)(kw0 + row) * a.dvs[2] + ((slot ^ swz<D>(row)) << 3)) = v4;
      }
    }
    if (stores_first) partial_rows();
    FAT5_STAMP(6);
    return;
  }
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int krow = kw0 + 32 * kb + lq;
    if (krow < N) {
      uint16_t* dkrow = dkb + (int64_t)krow * a.dks[2];
      uint16_t* dvrow = dvb + (int64_t)krow * a.dvs[2];
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 wk, wv;
          wk[0] = pack2<BF16>(dk[kb][db][4 * g + 0] * scale, dk[kb][db][4 * g + 1] * scale);
          wk[1] = pack2<BF16>(dk[kb][db][4 * g + 2] * scale, dk[kb][db][4 * g + 3] * scale);
          wv[0] = pack2<BF16>(dv[kb][db][4 * g + 0], dv[kb][db][4 * g + 1]);
          wv[1] = pack2<BF16>(dv[kb][db][4 * g + 2], dv[kb][db][4 * g + 3]);
          *reinterpret_cast<u32x2*>(dkrow + 32 * db + 8 * g + 4 * hi) = wk;
          *reinterpret_cast<u32x2*>(dvrow + 32 * db + 8 * g + 4 * hi) = wv;
        }
    }
  }
  FAT5_STAMP(6);
}

// =============================================================================================
// dQ (+ delta, + the row statistics of the dK/dV body): 64 query rows per wave, software-pipelined key-step loop
// =============================================================================================
// Same contract as attn_bwd_q_kernel (attn_bwd.h; replaces the reference `_bwd_preprocess` + `_bwd_q_kernel`,
// flash_attention_v2_bias.py:516-556, :748-905).  A wave owns two 32-row query blocks (Q / dO fragments and the dQ^T
// accumulators in AGPRs), one wave per SIMD; per 32-key step i the matrix pipe runs dQ^T += K^T dS^T of step i-1 and
// S^T = K Q^T, dP'^T = V dO^T - delta of step i+1 (24 MFMAs, every K / V fragment read from LDS feeds two of them) while the
// VALU pipe turns step i's scores into dS (FMA, v_exp_f32, multiply per element; bf16 pack per pair).  K / V steps come
// global -> LDS by DMA three steps ahead into a 4-slot ring, one barrier per step.  Every wave picks the pipelined or the
// general iteration by ITS 64 rows (wave-uniform): only the steps that cross the wave's own RPE band / causal diagonal / the key
// tail run the general softmax.
template <int D>
struct BwdQ64Cfg {
  static constexpr int NW = 4, BM = 64 * NW, KT = 32, NT = 64 * NW, NS = 4;
  static constexpr int IMG = rm_bytes<D, KT>();  // one 32-key image (K or V)
  static constexpr int SLOT = 2 * IMG;
  static constexpr int RING = NS * SLOT;
  // staging (round 4): the wave's 64 rows of Q, dO and O arrive as whole rows by LDS-DMA (one 1-KiB piece = 8 rows of 128 bytes) and
  // are read back as operand fragments; dQ leaves the same way.  A lane loading "its" row straight from global touches 64 cache
  // lines per instruction: 13.5 k cycles of prologue measured (tools/trace64.py), a third of the kernel at 512 keys.
  static constexpr int STG_T = 64 * 2 * D;      // one tensor's 64 rows of one wave
  static constexpr int STG = NW * 3 * STG_T;
  // qdg: the body also forms the table gradient's per-diagonal sums (QDG below): one private array of 2R+1 sums per (wave, half-wave) + one scratch word per thread
  static size_t smem(int R, int bias_mode, bool stage = false, bool qdg = false) {
    return RING + (stage ? STG : 0) + (bias_mode == FAT5_BIAS_RPE1D ? rpe_table_bytes(R) + 16 : 0) +
           (qdg && bias_mode == FAT5_BIAS_RPE1D ? ((size_t)(2 * R + 1) * 4 * 2 * NW + NT * 4 + 63) / 64 * 64 : 0);
  }
};

// QDG (round 6; T5 bias only): the per-diagonal sums of dS -- the gradient of the bias generator, drpe1d[h][d] = sum over (q, k) with k - q = d of dS[q][k] -- are
// formed HERE instead of in the dK/dV body (attn_bwd_kv64_body<..., NODIAG>).  Why: in the one-launch backward of a short sequence (attn_bwd_fused64_kernel, cfg2:
// 96 dK/dV + 96 dQ workgroups, all resident) the launch lasts as long as its LONGEST workgroup, and that is a dK/dV one -- 57.4 k cycles against 41.0 k for dQ
//  -- of which the diagonal machinery is ~8 k; here it rides in the shorter workgroup.  The arithmetic is diag_sum.h's, MIRRORED: this body holds dS^T
// (lane = query row, register r <-> key crow(r, hi)), the dK/dV body dS (lane = key, register <-> query row), so with d' = row - key = -d and
// base' = (first row of the query block) - (first key of the step) an element lies on d' = base' + 16 a + p - 16 qh - ql0 - 4 hi -- the formula of diag_sum.h with the
// roles of rows and keys exchanged.  A step moves 32 keys up: base' - 32, the same hand-over between consecutive steps; query block 1 of a step has the homes
// query block 0 had one step earlier.  Sums leave at index R - d' of the wave's private arrays (d' >= R: the far-negative bin, d' <= -R: the far-positive one).
template <int D, bool BF16, int BIAS, bool QDG = false>
FAT5_DEV void attn_bwd_q64_body(const AttnArgs& a, const int bid) {
  static_assert(D == 64 && BIAS != FAT5_BIAS_DENSE, "gap schedule written for D = 64, bias none / rpe1d");
  static_assert(!QDG || BIAS == FAT5_BIAS_RPE1D, "the diagonal sums belong to the T5 bias");
  FAT5_STAMP(0);
  using Cfg = BwdQ64Cfg<D>;
  constexpr int BM = Cfg::BM, NT = Cfg::NT, IMG = Cfg::IMG, SLOT = Cfg::SLOT;
  constexpr int KK = D / 16, DB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bool stg = a.lds_stage != 0;
  float* sT = reinterpret_cast<float*>(smem + Cfg::RING + (stg ? Cfg::STG : 0)) + kRpePad;  // (entry d of copy 0 at sT[d + R]; see attn_common.h)

  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, lq = l & 31, hi = l >> 5;  // (w: provably wave-uniform)
  int b, h, mblk;
  decode_unit(a, bid, a.n_mblk, b, h, mblk, (FAT5_CAUSAL_ORDER && a.causal) ? 1 : 0);
  const int M = a.M, N = a.N;
  const int m0 = mblk * BM;
  if (m0 >= M) return;
  [[maybe_unused]] RpeTableRegs tabr;
  if constexpr (BIAS == FAT5_BIAS_RPE1D) tabr = rpe_table_load_first(a.rpe1d + (int64_t)h * (2 * a.R + 1), a.R, tid, NT);
  const uint16_t* qb_ = a.q + (int64_t)b * a.qs[0] + (int64_t)h * a.qs[1];
  const uint16_t* kb_ = a.k + (int64_t)b * a.ks[0] + (int64_t)h * a.ks[1];
  const uint16_t* vb_ = a.v + (int64_t)b * a.vs[0] + (int64_t)h * a.vs[1];
  const uint16_t* ob_ = a.o + (int64_t)b * a.os[0] + (int64_t)h * a.os[1];
  const uint16_t* dob_ = a.dout + (int64_t)b * a.dos[0] + (int64_t)h * a.dos[1];
  uint16_t* dqb_ = a.dq + (int64_t)b * a.dqs[0] + (int64_t)h * a.dqs[1];
  const int64_t stat_off = ((int64_t)b * a.H + h) * a.M;
  const int P = N - M;
  [[maybe_unused]] const bool ctab = BIAS == FAT5_BIAS_RPE1D && a.causal && P < a.R && P >= -a.R;  // (the causal mask by the bias table itself: see the dK/dV body)
  int n_end = N;
  if (a.causal) n_end = min(N, m0 + BM + P);
  const int nt = n_end > 0 ? (n_end + 31) / 32 : 0;
  const int qw0 = m0 + 64 * w;  // first query row of this wave; block qb covers qw0 + 32*qb .. +31
  FAT5_STAMP(10);

  // Q and dO fragments (B operands), delta = rowsum(o * do) (reference _bwd_preprocess, :516-556), row statistics
  u32x4 qf[2][KK], dof[2][KK], off_[2][KK];
  float nL2[2];
  // dP'^T = V dO^T - delta.  The pipelined loop takes -delta as a C operand (nd16: 16 registers per query block, live for the whole
  // loop; 1077 vs 1108 us at cfg3 for the alternative); the general iteration forms it with one extra MFMA per query block, ones(32 x 16) x D3 with
  // D3[j][q] = the j-th 16-bit piece of -delta_q (hi + mid + lo: 24+ bits, exact to fp32) -- a 16-register broadcast of -delta per
  // block as the C operand would hold 32 VGPRs for the whole loop (C and D of an MFMA share one register file)
  u32x4 d3[2];
  [[maybe_unused]] f32x16 nd16[2];
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  float Lq_[2];  // (in flight beside the operand loads: one memory round trip for the whole prologue)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) Lq_[qb] = a.lse[stat_off + min(qw0 + 32 * qb + lq, M - 1)];
  // staged: this wave's rows qw0 .. qw0 + 63 of Q | dO | O as three swizzled row-major images (rows past M arrive as zeros)
  const uint32_t stg_w = lds0 + (uint32_t)(Cfg::RING + w * 3 * Cfg::STG_T);
  using SDma = DmaStage<D, 64, 64>;
  static_assert(SDma::PER == 8 && SDma::NV == 2, "eight 1-KiB pieces of 8 rows per tensor");
  if (stg) {
    SDma sq, sdo, so;
    sq.init(a.qs[2], l);
    sdo.init(a.dos[2], l);
    so.init(a.os[2], l);
    const __amdgpu_buffer_rsrc_t qrs = make_rows_rsrc(qb_, a.qs[2], M, D), dors = make_rows_rsrc(dob_, a.dos[2], M, D), ors = make_rows_rsrc(ob_, a.os[2], M, D);
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)stg_w);
    const uint32_t q0 = (uint32_t)qw0 * (uint32_t)a.qs[2] * 2u, do0 = (uint32_t)qw0 * (uint32_t)a.dos[2] * 2u, o0 = (uint32_t)qw0 * (uint32_t)a.os[2] * 2u;
#pragma unroll
    for (int i = 0; i < SDma::PER; ++i) {
      dma16_asm(qrs, dst + (uint32_t)(i * 1024), sq.voff[i % 2], q0 + sq.piece_step * (i / 2));
      dma16_asm(dors, dst + (uint32_t)(Cfg::STG_T + i * 1024), sdo.voff[i % 2], do0 + sdo.piece_step * (i / 2));
      dma16_asm(ors, dst + (uint32_t)(2 * Cfg::STG_T + i * 1024), so.voff[i % 2], o0 + so.piece_step * (i / 2));
    }
  } else {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qrow_c = min(qw0 + 32 * qb + lq, M - 1);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        qf[qb][kk] = *reinterpret_cast<const u32x4*>(qb_ + (int64_t)qrow_c * a.qs[2] + 16 * kk + 8 * hi);
        dof[qb][kk] = *reinterpret_cast<const u32x4*>(dob_ + (int64_t)qrow_c * a.dos[2] + 16 * kk + 8 * hi);
        off_[qb][kk] = *reinterpret_cast<const u32x4*>(ob_ + (int64_t)qrow_c * a.os[2] + 16 * kk + 8 * hi);
        if constexpr (QDG) {
          // rows past M hold copies of row M - 1 here (the staged form gets zeros from the descriptor): with dO = O = 0 their dP', delta and so their dS are
          // exact zeros -- the diagonal sums must not see them (their dQ is never stored either way)
          if (qw0 + 32 * qb + lq >= M) {
            dof[qb][kk] = u32x4{0u, 0u, 0u, 0u};
            off_[qb][kk] = u32x4{0u, 0u, 0u, 0u};
          }
        }
      }
    }
  }
  FAT5_STAMP(11);
  // (the fragments of the staged form are read after the prologue's wait; everything derived from them follows below)
  auto derive_rows = [&]() {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qrow = qw0 + 32 * qb + lq;
      float dsum = 0.f;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const u32x4 of = off_[qb][kk];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dsum = fmaf(cvt_lo<BF16>(of[j]), cvt_lo<BF16>(dof[qb][kk][j]), dsum);
          dsum = fmaf(cvt_hi<BF16>(of[j]), cvt_hi<BF16>(dof[qb][kk][j]), dsum);
        }
      }
      const float delta = pair_sum(dsum);
      if (!FAT5_TRACE && a.delta && qrow < M && hi == 0) a.delta[stat_off + qrow] = delta;
      const float Lq = Lq_[qb];
      nL2[qb] = (Lq < kDeadRowLse) ? -INFINITY : -Lq * kLog2e;  // (dead rows: see attn_bwd.h)
      if (a.stat2 && hi == 0 && qrow < (M + 31) / 32 * 32) {
        float* st = a.stat2 + (((int64_t)b * a.H + h) * ((M + 31) / 32) + (qrow >> 5)) * 64 + (qrow & 31);
        const bool live = qrow < M && !(Lq < kDeadRowLse);
        st[0] = live ? -Lq / a.scale : (a.scale > 0.f ? -INFINITY : INFINITY);
        st[32] = qrow < M ? -delta : 0.f;
      }
      {
        const float nd = -delta;
        const uint32_t p0 = to16<BF16>(nd);
        const float r1 = nd - cvt16<BF16>((uint16_t)p0);
        const uint32_t p1 = to16<BF16>(r1);
        const float r2 = r1 - cvt16<BF16>((uint16_t)p1);
        const uint32_t p2 = to16<BF16>(r2);
        d3[qb] = hi == 0 ? u32x4{p0 | (p1 << 16), p2, 0u, 0u} : u32x4{0u, 0u, 0u, 0u};  // (k-index 8*hi + j of the B operand)
#pragma unroll
        for (int r = 0; r < 16; ++r) nd16[qb][r] = nd;
      }
    }
  };
  if (!stg) derive_rows();
  const uint32_t one2 = pack2<BF16>(1.f, 1.f);
  u32x4 ones4 = {one2, one2, one2, one2};
  const float* sTa[2] = {sT, sT};
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) sTa[qb] = sT + ((a.R - (qw0 + 32 * qb + lq)) & 3) * rpe_n1p(a.R);
  }

  // ---- QDG: state of the per-diagonal sums (see the dK/dV body; everything mirrored: rows <-> keys) ----
  [[maybe_unused]] const int n1 = 2 * a.R + 1;
  [[maybe_unused]] float* const sD0 = sT - kRpePad + 4 * rpe_n1p(a.R);  // behind the four table copies: 2 NW private arrays of n1 sums, then one scratch word per thread
  [[maybe_unused]] DiagCarry dcar[2];
  diag_carry_zero(dcar[0]);
  diag_carry_zero(dcar[1]);
  [[maybe_unused]] float dprev0 = 0.f;
  [[maybe_unused]] bool diag_run = false;  // (wave-uniform) a run is open: dcar / dprev0 hold partial diagonals of the step at diag_nb
  [[maybe_unused]] int diag_nb = 0;
  [[maybe_unused]] float dmask[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (QDG) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int ql0 = i < 3 ? i + 1 : i + 5;  // 1, 2, 3, 8, 9, 10, 11
      dmask[i] = ((l & 15) + ql0 >= 16) ? 1.f : 0.f;
      asm volatile("" : "+v"(dmask[i]));
    }
  }
  [[maybe_unused]] float far_neg = 0.f, far_pos = 0.f;
  [[maybe_unused]] const bool want_drpe = QDG && (a.drpe_part != nullptr);
  [[maybe_unused]] float* const dsum = sD0 + (2 * w + hi) * n1 + a.R + 4 * hi - lq;  // (diagonal d' = base' + lane - 4 hi, i.e. d = -d', at dsum[-base'])
  [[maybe_unused]] float* const dtrash = sD0 + 2 * Cfg::NW * n1 + tid;
  // E: the finished sums of the diagonals d' = base + (lane & 31) - 4 hi
  [[maybe_unused]] auto diag_store = [&](const float E, const int base) {
    const int dm = base + lq - 4 * hi;
    far_neg += dm >= a.R ? E : 0.f;    // d = -d' <= -R
    far_pos += dm <= -a.R ? E : 0.f;   // d >= R
    *((dm > -a.R && dm < a.R) ? dsum - base : dtrash) = E;
  };
  // end of the key step at nb: st[qb] = the step's accumulators of query block qb
  [[maybe_unused]] auto diag_step_end = [&](const DiagStep (&st)[2], const int nb) {
    const float F0 = diag_finish_halves(dcar[0], st[0], l), F1 = diag_finish_halves(dcar[1], st[1], l);
    diag_store(F1 + dprev0, qw0 + 32 - nb);
    dprev0 = F0;
    diag_run = true;
    diag_nb = nb;
  };
  [[maybe_unused]] auto diag_flush = [&]() {
    if (diag_run) {
      diag_store(dcar[1].cur + dprev0, qw0 - diag_nb);
      diag_store(dcar[0].cur, qw0 - diag_nb - 32);
      diag_carry_zero(dcar[0]);
      diag_carry_zero(dcar[1]);
      dprev0 = 0.f;
      diag_run = false;
    }
  };
  [[maybe_unused]] f32x4 facc4 = {0.f, 0.f, 0.f, 0.f};  // far-bin sums of the pipelined far steps on the matrix pipe (see the dK/dV body)
  if constexpr (QDG) {
    // the private arrays start at zero: 16-byte stores, issued here -- under the latency of the staging requests above -- and made visible by the prologue's
    // barrier (the area lies behind the table copies: nothing else of the prologue touches it; the last store may run up to 12 bytes into the scratch words)
    f32x4* z = reinterpret_cast<f32x4*>(sD0);
    for (int i = tid; i < (n1 * 2 * Cfg::NW + 3) / 4; i += NT) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  f32x16 dq[2][DB];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[qb][i][r] = 0.f;

  // ---- ring: key step t (keys 32t ..+31) lives in slot t % 4: [K image | V image] ----
  using Dma = DmaStage<D, Cfg::KT, NT>;
  static_assert(Dma::PER == 1 && Dma::NV == 1, "one 16-byte piece per thread and image");
  Dma kst, vst;
  kst.init(a.ks[2], tid);
  vst.init(a.vs[2], tid);
  const __amdgpu_buffer_rsrc_t krs = make_rows_rsrc(kb_, a.ks[2], N, D);
  const __amdgpu_buffer_rsrc_t vrs = make_rows_rsrc(vb_, a.vs[2], N, D);
  const uint32_t kstride_b = (uint32_t)a.ks[2] * 2u, vstride_b = (uint32_t)a.vs[2] * 2u;
  const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(tid >> 6) * 1024u);
  auto dma_step = [&](int t, uint32_t slot_off) {
    const uint32_t tt = (uint32_t)__builtin_amdgcn_readfirstlane(t);
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave_lds + slot_off));
    dma16_asm(krs, dst, kst.voff[0], tt * 32u * kstride_b);
    dma16_asm(vrs, dst + (uint32_t)IMG, vst.voff[0], tt * 32u * vstride_b);
  };
  // E(t): step t+1 has landed and is visible to every wave; every wave is done with step t-1, whose slot takes step t+3
  auto sync_wait = [&](int t) {
    if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else wait_dma_all();
    __syncthreads();
  };
  auto sync_step = [&](int t, uint32_t slot3_off) {
    sync_wait(t);
    if (t + 3 < nt) dma_step(t + 3, slot3_off);
  };
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (i < nt) dma_step(i, (uint32_t)(i * SLOT));
  for (int i = tid; i < SLOT / 16; i += NT) reinterpret_cast<u32x4*>(smem + 3 * SLOT)[i] = u32x4{0u, 0u, 0u, 0u};  // (see the dK/dV body)
  // (round 6, measured and dropped -- profiles/r06_prologue_ab.log: the ring requests right behind the staging requests, the fragment reads + delta behind a counted
  //  wait for the private staging images only, the lse loads youngest: cfg2 backward 27.9 vs 27.0 us on the same box, (2,12,1024) 38.5 vs 37.6 -- the order below stays)
  FAT5_STAMP(7);
  if constexpr (BIAS == FAT5_BIAS_RPE1D) rpe_table_fill_rest(sT - kRpePad, a.rpe1d + (int64_t)h * (2 * a.R + 1), a.R, tid, NT, tabr, ctab ? P : 0x7fffffff);
  FAT5_STAMP(8);
  wait_dma_all();
  __syncthreads();
  FAT5_STAMP(9);
  FragAddr<D> fa;
  fa.init(l);
  if (stg) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const uint32_t ad = stg_w + (uint32_t)(fa.rm[kk] + qb * 32 * 2 * D);
        qf[qb][kk] = lds_rd128(ad);
        dof[qb][kk] = lds_rd128(ad + (uint32_t)Cfg::STG_T);
        off_[qb][kk] = lds_rd128(ad + (uint32_t)(2 * Cfg::STG_T));
      }
    derive_rows();
  }
  FAT5_STAMP(1);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) asm volatile("" : "+a"(qf[qb][kk]), "+a"(dof[qb][kk]));  // MFMA-only operands: AGPRs
  asm volatile("" : "+a"(d3[0]), "+a"(d3[1]), "+a"(ones4));

  uint32_t rmA[KK], trA[2][DB];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) rmA[kk] = lds0 + (uint32_t)fa.rm[kk];
#pragma unroll
  for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
    for (int db = 0; db < DB; ++db) trA[j2][db] = lds0 + (uint32_t)fa.tr[j2][db];

  const float c2 = a.scale * kLog2e;
  float cst_neg = 0.f, cst_pos = 0.f;
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
    cst_neg = sT[0];
    cst_pos = sT[2 * a.R];
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // band steps of the pipelined loop (round 4): LDS byte address of entry 0 of this lane's padded table copy per query block and the
  // lane's position term -- the entries of keys nb + 4 hi + 8 gg + (0..3), gg = 0..3, are the 16 bytes at tab_addr(qb, nb) + 32 gg
  // (the window runs UP with the key; see softmax_generic) -- and the first entries of the step whose softmax is due (fetched one
  // iteration ahead)
  uint32_t tabA[2] = {0u, 0u};
  int posb[2] = {0, 0};
  u32x4 TN0;
  uint32_t tadr0 = 0u;
  if constexpr (BIAS == FAT5_BIAS_RPE1D) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qrow = qw0 + 32 * qb + lq;
      tabA[qb] = (uint32_t)(uintptr_t)sTa[qb];
      posb[qb] = a.R + 4 * hi - qrow - ((a.R - qrow) & 3);
    }
  }
  const int tclamp_hi = rpe_n1p(a.R) - kRpePad - 28;  // (rpe_clamp_asc)
  auto tab_addr = [&](int qb, int nb) { return tabA[qb] + 4u * (uint32_t)min(max(posb[qb] + nb, -kRpePad), tclamp_hi); };

  // Pipeline state between two iterations (iteration i = key step i is in its softmax stage):
  //   S, DP     S^T = K Q^T and dP'^T = V dO^T - delta of step i (lane = query row, register r <-> key crow(r, hi))
  //   DSB       dS^T of step i-1 rounded to bf16 (:720), as B operands;  TRK  the K^T fragments (t2 = 0; db = 0, 1) of step i-1
  f32x16 S[2], DP[2];
  u32x4 DSB[2][2], TRK[2];

  auto rd_tr = [&](uint32_t off, int t2, int db) {
    const uint32_t o = off + (uint32_t)(16 * t2 * 2 * D);
    return lds_rd_tr(trA[0][db] + o, trA[1][db] + o);
  };
  auto score_step = [&](const uint32_t so, f32x16 (&Sx)[2], f32x16 (&DPx)[2]) {
    u32x4 kf[KK], vf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      kf[kk] = lds_rd128(rmA[kk] + so);
      vf[kk] = lds_rd128(rmA[kk] + so + (uint32_t)IMG);
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) Sx[qb] = mfma32<BF16>(kf[kk], qf[qb][kk], kk == 0 ? zero16 : Sx[qb]);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) DPx[qb] = mfma32<BF16>(ones4, d3[qb], zero16);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) DPx[qb] = mfma32<BF16>(vf[kk], dof[qb][kk], DPx[qb]);
  };
  // dQ^T[d][q] += K^T[d][key] dS^T[key][q] of the pending step, whose K image is in the slot at `so`
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
  // general softmax stage of the key step at nb: S, DP -> DSB
  auto softmax_generic = [&](const int nb) {
    [[maybe_unused]] DiagStep gst[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16& s = S[qb];
      const f32x16& dp = DP[qb];
      const int qr0 = qw0 + 32 * qb, qrow = qr0 + lq;
      const float nl = nL2[qb];
      if constexpr (BIAS == FAT5_BIAS_RPE1D) {
        // far, edge and band blocks alike: four aligned 16-byte reads of this lane's padded table copy, window clamped (attn_common.h)
        const int R = a.R;
        const float4* tp4 = reinterpret_cast<const float4*>(sTa[qb] + rpe_clamp_asc(R + nb + 4 * hi - qrow - ((R - qrow) & 3), R));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bq = tp4[2 * g];
          s[4 * g + 0] = fmaf(s[4 * g + 0], c2, bq.x + nl);
          s[4 * g + 1] = fmaf(s[4 * g + 1], c2, bq.y + nl);
          s[4 * g + 2] = fmaf(s[4 * g + 2], c2, bq.z + nl);
          s[4 * g + 3] = fmaf(s[4 * g + 3], c2, bq.w + nl);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], c2, nl);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r]) * dp[r];  // dS = P (dP - delta)   (:713)
      const bool nmask = nb + 32 > N;
      const bool cmask = a.causal && (nb + 31 > qr0 + P);
      if (nmask || cmask) {
        const int lim = a.causal ? min(N - 1, qrow + P) : N - 1;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (nb + crow(r, hi) > lim) s[r] = 0.f;
      }
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) DSB[qb][t2] = pack8<BF16>(s, t2);
      if constexpr (QDG) {
        // the block's dS (fp32, masked elements zero; rows past M carry dO = 0, delta = 0 -> dS = 0) onto its diagonals; whole blocks beyond the band included
        diag_step_zero(gst[qb]);
        static_for<16>([&](auto ri) { diag_elem<decltype(ri)::value>(gst[qb], s[decltype(ri)::value], l & 15); });
      }
    }
    if constexpr (QDG) diag_step_end(gst, nb);
  };
  auto generic_iter = [&](const int t) {
    const uint32_t o_prev = (uint32_t)(((t + 3) & 3) * SLOT), o_cur = (uint32_t)((t & 3) * SLOT), o_next = (uint32_t)(((t + 1) & 3) * SLOT);
    product_step(o_prev);
    sync_step(t, o_prev);
    f32x16 Sn[2], DPn[2];
    score_step(o_next, Sn, DPn);
    softmax_generic(t * 32);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      S[qb] = Sn[qb];
      DP[qb] = DPn[qb];
    }
    TRK[0] = rd_tr(o_cur, 0, 0);
    TRK[1] = rd_tr(o_cur, 0, 1);
  };

  // One pipelined iteration = 26 MFMA gaps.  Gap g holds, all mutually independent:
  //   MFMA   g < 8: dQ^T[qb][db] += K^T(t2, db) . dS^T[qb][t2] of step i-1 (fragment pairs (t2, db) outer, query blocks inner);
  //          8..15: S^T[qb] of step i+1 (k-steps outer); 16, 17: dP'^T[qb] = -delta; 18..25: dP'^T[qb] += V dO^T
  //   VALU   32 elements per lane (query block e >> 4) opened evenly over gaps 0 .. NG-4: x = s*c2 + (cst - L2) | one gap later
  //          p = exp2(x) | one more: ds = p*dp' | pairs packed to bf16 once both halves exist
  //   LDS    gaps 0..3 the K^T fragments (t2 = 1) of step i-1; gap 4 the barrier E(i) + the DMA of step i+3; gaps 4..7 the K,
  //          gaps 12..15 the V row-major fragments of step i+1; gaps 22..25 the K^T fragments (t2 = 0) of step i
  // SL = the step's ring slot, a compile-time constant: the steady state runs four steps (slots 0..3) per trip, straight-line --
  // slot offsets are instruction immediates and S / Sn (...) trade registers from one step to the next instead of being copied.
  // BAND (round 4): the same iteration for steps that cross the T5 band (every key visible): the bias of element (k, q) comes from this
  // lane's padded table copy -- four 16-byte reads per query block, each issued three or more gaps ahead of its first use (the first one
  // during the previous iteration: TN0) -- plus the row's -L2 (one v_add in front of the FMA: the rounding of the general iteration).
  // A band step was the general, unpipelined iteration before (the 64-row body lost to the 32-row one below 8192 keys for it).
  constexpr int NG = 24, G_DP = 16;  // (G_DP: first gap of the dP k-steps)
  auto fast_iter = [&]<int SL, bool BAND>(const int t, const float ad0, const float ad1) {
    constexpr uint32_t o_prev = (uint32_t)(((SL + 3) & 3) * SLOT), o_cur = (uint32_t)(SL * SLOT), o_next = (uint32_t)(((SL + 1) & 3) * SLOT);
    f32x16 Sn[2], DPn[2];
    u32x4 DSn[2][2], kf[KK], vf[KK];
    u32x2 th[2][2], tn[2][2];  // [db][half]: K^T fragments t2 = 1 of step i-1 / t2 = 0 of step i
    float X[32], Pv[32], Dv[32];
    u32x4 T[2][4];
    uint32_t tadr1 = 0u;
    if constexpr (BAND) T[0][0] = TN0;
    // QDG, band steps: the step's dS onto its diagonals (diag_sum.h), element e one gap after its multiply: one rotating add for everything, one rotating
    // multiply-add by the borrow mask (both read Dv[e], written a gap earlier: no DPP hazard)
    [[maybe_unused]] DiagStep dst[2];
    if constexpr (BAND && QDG) {
      diag_step_zero(dst[0]);
      diag_step_zero(dst[1]);
    }
    [[maybe_unused]] auto stE_ = [&]<int E>() {
      constexpr int r = E & 15, ql0 = diag_ql0(r);
      diag_elem_u<r>(dst[E >> 4], Dv[E]);
      if constexpr (ql0 != 0) diag_elem_bm<r>(dst[E >> 4], Dv[E], dmask[ql0 < 8 ? ql0 - 1 : ql0 - 5]);
    };
    auto stA_ = [&]<int E>() {
      if constexpr (BAND) {
        float tn_ = __uint_as_float(T[E >> 4][(E & 15) >> 2][E & 3]);
        asm_add(tn_, (E >> 4) == 0 ? ad0 : ad1);  // (ad = -L2 of the lane's row: entry + ad, then the FMA -- as the general iteration rounds)
        X[E] = asm_fma(S[E >> 4][E & 15], c2, tn_);
      } else {
        X[E] = asm_fma(S[E >> 4][E & 15], c2, (E >> 4) == 0 ? ad0 : ad1);
      }
    };
    auto stB_ = [&]<int E>() { Pv[E] = asm_exp2(X[E]); };
    auto stC_ = [&]<int E>() { Dv[E] = asm_mul(Pv[E], DP[E >> 4][E & 15]); };
    auto stD_ = [&]<int E0>() {
      constexpr int qb = E0 >> 4, r0 = E0 & 15;
      DSn[qb][r0 >> 3][(r0 & 7) >> 1] = asm_cvt_pk<BF16>(Dv[E0], Dv[E0 + 1]);
    };
    static_for<NG>([&](auto gi) {
      constexpr int g = decltype(gi)::value;
      // ---- MFMA ----
      if constexpr (g < 8) {
        constexpr int p = g >> 1, t2 = p >> 1, db = p & 1, qb = g & 1;
        u32x4 fr;
        if constexpr (t2 == 0) fr = TRK[db];
        else fr = u32x4{th[db][0][0], th[db][0][1], th[db][1][0], th[db][1][1]};
        mfma_acc_agpr<BF16>(dq[qb][db], fr, DSB[qb][t2]);
      } else if constexpr (g < 16) {
        constexpr int kk = (g - 8) >> 1, qb = g & 1;
        if constexpr (kk == 0) Sn[qb] = mfma32<BF16>(kf[kk], qf[qb][kk], zero16);
        else Sn[qb] = mfma32<BF16>(kf[kk], qf[qb][kk], Sn[qb]);
      } else if constexpr (g < G_DP) {
        DPn[g - 16] = mfma32<BF16>(ones4, d3[g - 16], zero16);
      } else {
        constexpr int kk = (g - G_DP) >> 1, qb = g & 1;
        if constexpr (kk == 0) DPn[qb] = mfma32<BF16>(vf[kk], dof[qb][kk], nd16[qb]);  // (nd16 lives for the whole loop: no WAR window)
        else DPn[qb] = mfma32<BF16>(vf[kk], dof[qb][kk], DPn[qb]);
      }
      __builtin_amdgcn_sched_barrier(0);  // (the MFMA opens its gap: without this it may sink below the gap's VALU work and pair up with the next one)
      // ---- barrier + DMA (the K and the V piece of step i+3 in gaps of their own: see the dK/dV body) ----
      if constexpr (g == 4) {
        if constexpr (FAT5_DMA_SPREAD) sync_wait(t);
        else sync_step(t, o_prev);
      }
      if constexpr (FAT5_DMA_SPREAD && (g == 5 || g == 6)) {
        if (t + 3 < nt) {
          const uint32_t tt = (uint32_t)__builtin_amdgcn_readfirstlane(t + 3);
          const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave_lds + o_prev));
          if constexpr (g == 5) dma16_asm(krs, dst, kst.voff[0], tt * 32u * kstride_b);
          else dma16_asm(vrs, dst + (uint32_t)IMG, vst.voff[0], tt * 32u * vstride_b);
        }
      }
      // ---- LDS ----
      if constexpr (g < 4) {
        constexpr int db = g >> 1, half = g & 1;
        th[db][half] = lds_rd_tr_half(trA[half][db] + o_prev + (uint32_t)(16 * 2 * D));
      } else if constexpr (g < 8) {
        kf[g - 4] = lds_rd128(rmA[g - 4] + o_next);
      } else if constexpr (g >= 12 && g < 16) {
        vf[g - 12] = lds_rd128(rmA[g - 12] + o_next + (uint32_t)IMG);
      } else if constexpr (g >= NG - 4) {
        constexpr int db = (g - (NG - 4)) >> 1, half = g & 1;
        tn[db][half] = lds_rd_tr_half(trA[half][db] + o_cur);
      }
      if constexpr (BAND) {  // table entries: (query block, key group gg) is first used in gap 0, 3, 5, 8 | 11, 13, 16, 19
        if constexpr (g == 0) T[0][1] = lds_rd128(tadr0 + 32u);
        else if constexpr (g == 2) T[0][2] = lds_rd128(tadr0 + 64u);
        else if constexpr (g == 5) T[0][3] = lds_rd128(tadr0 + 96u);
        else if constexpr (g == 6) tadr1 = tab_addr(1, t * 32);
        else if constexpr (g == 8) T[1][0] = lds_rd128(tadr1);
        else if constexpr (g == 9) T[1][1] = lds_rd128(tadr1 + 32u);
        else if constexpr (g == 11) T[1][2] = lds_rd128(tadr1 + 64u);
        else if constexpr (g == 16) T[1][3] = lds_rd128(tadr1 + 96u);
        else if constexpr (g == 17) tadr0 = tab_addr(0, (t + 1) * 32);  // the next step: keys 32 further up
        else if constexpr (g == 18) TN0 = lds_rd128(tadr0);
      }
      // ---- VALU ----
      {
        // first element whose stage A sits in gap gg: the 32 elements open in gaps 0 .. NG-4, so that the last one's multiply and pack
        // (two and three gaps later) still fall inside this iteration -- no dependent tail behind the last MFMA
        constexpr auto lo = [](int gg) { return gg <= 0 ? 0 : (gg >= NG - 3 ? 32 : (32 * gg) / (NG - 3)); };
        static_for<lo(g - 2) - lo(g - 3)>([&](auto ei) {  // stage D: pairs whose odd half was multiplied one gap ago
          constexpr int e = lo(g - 3) + decltype(ei)::value;
          if constexpr ((e & 1) == 1) stD_.template operator()<e - 1>();
          if constexpr (BAND && QDG) stE_.template operator()<e>();  // (stage E: the element onto its diagonal)
        });
        static_for<lo(g - 1) - lo(g - 2)>([&](auto ei) { stC_.template operator()<lo(g - 2) + decltype(ei)::value>(); });
        static_for<lo(g) - lo(g - 1)>([&](auto ei) { stB_.template operator()<lo(g - 1) + decltype(ei)::value>(); });
        static_for<lo(g + 1) - lo(g)>([&](auto ei) { stA_.template operator()<lo(g) + decltype(ei)::value>(); });
        if constexpr (g == NG - 1) {  // the tail of the step (dependent ops back to back)
          static_for<32 - lo(NG - 1)>([&](auto ei) { stB_.template operator()<lo(NG - 1) + decltype(ei)::value>(); });
          static_for<32 - lo(NG - 2)>([&](auto ei) { stC_.template operator()<lo(NG - 2) + decltype(ei)::value>(); });
          static_for<32 - lo(NG - 3)>([&](auto ei) {
            constexpr int e = lo(NG - 3) + decltype(ei)::value;
            if constexpr ((e & 1) == 1) stD_.template operator()<e - 1>();
          });
        }
      }
      // QDG, far steps: the sum of the step's rounded dS for its far bin on the matrix pipe -- one 16x16x32 MFMA (ones x four packed words: every row of the result
      // = the column sums) per group of words once it is complete: group k = (query block k >> 1, half k & 1) is packed in gaps 8, 13, 18, 23
      if constexpr (QDG && !BAND && (g == 9 || g == 14 || g == 19)) {
        constexpr int grp = (g - 9) / 5;
        if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(facc4) : "a"(ones4), "v"(DSn[grp >> 1][grp & 1]));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(facc4) : "a"(ones4), "v"(DSn[grp >> 1][grp & 1]));
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (QDG && !BAND) {
      asm volatile("s_nop 1" ::: "memory");  // (the last group's words come from the asm ops just above: two wait states by hand)
      if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(facc4) : "a"(ones4), "v"(DSn[1][1]));
      else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(facc4) : "a"(ones4), "v"(DSn[1][1]));
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      S[qb] = Sn[qb];
      DP[qb] = DPn[qb];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) DSB[qb][t2] = DSn[qb][t2];
    }
#pragma unroll
    for (int db = 0; db < DB; ++db) TRK[db] = u32x4{tn[db][0][0], tn[db][0][1], tn[db][1][0], tn[db][1][1]};
    if constexpr (BAND && QDG) {
      asm volatile("s_nop 1" : "+v"(dst[1].u1), "+v"(dst[1].b1));  // (asm producers: no hazard padding is generated for them)
      diag_step_end(dst, t * 32);
    }
  };

  if (nt > 0) {
    score_step(0u, S, DP);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) DSB[qb][t2] = zero4;
    TRK[0] = zero4;
    TRK[1] = zero4;
    FAT5_STAMP(2);
    // this wave's 64 rows x the step's 32 keys: all visible and one constant bias?  (wave-uniform)  side: -1 far-negative / no bias, +1 far-positive
    auto classify = [&](const int t, int& side) {
      const int nb = t * 32;
      bool fast = nb + 32 <= N && (!a.causal || nb + 31 <= qw0 + P);
      side = -1;
      if constexpr (BIAS == FAT5_BIAS_RPE1D) {
        const bool fneg = nb + 31 - qw0 <= -a.R, fpos = nb - (qw0 + 63) >= a.R;
        fast = fast && (fneg || fpos);
        side = fneg ? -1 : 1;
      }
      return fast;
    };
    // every key of the step visible to all 64 rows of the wave (no key tail, no causal mask)?  Monotone: earlier steps see more
    auto all_visible = [&](const int t) { return t * 32 + 32 <= N && (!a.causal || ctab || t * 32 + 31 <= qw0 + P); };  // (ctab: the table masks)
    // (single-body inner loops, not one loop over `fast ? A : B`: the register allocator keeps one assignment per loop and
    //  pays its copies only at the few transitions)
    int t = 0;
    while (t < nt) {
      int side, side3;
      // steady state: four steps (ring slots 0..3) per trip; the fast steps of one side are contiguous, so the first and the last
      // step of a trip decide for all four.  Steps that do not fill an aligned trip run the general iteration.
      while (t + 4 <= nt && (t & 3) == 0 && classify(t, side) && classify(t + 3, side3) && side3 == side) {
        const float cst = BIAS == FAT5_BIAS_RPE1D ? (side > 0 ? cst_pos : cst_neg) : 0.f;
        const float ad0 = cst + nL2[0], ad1 = cst + nL2[1];
        if constexpr (QDG) diag_flush();
        static_for<4>([&](auto si) { fast_iter.template operator()<decltype(si)::value, false>(t + decltype(si)::value, ad0, ad1); });
        t += 4;
        if constexpr (QDG) {
          asm volatile("s_nop 15" : "+v"(facc4));  // (asm MFMA -> VALU read of its result: no padding is generated; tied to the tuple so that no read moves above it)
          const float fsum = ((facc4[0] + facc4[1]) + (facc4[2] + facc4[3])) * 0.0625f;
          facc4 = f32x4{0.f, 0.f, 0.f, 0.f};
          if (side > 0) far_pos += fsum; else far_neg += fsum;
        }
      }
      if constexpr (BIAS == FAT5_BIAS_RPE1D) {
        // trips that touch the band but see every key (band-mode iterations)
        while (t + 4 <= nt && (t & 3) == 0 && all_visible(t + 3) && !(classify(t, side) && classify(t + 3, side3) && side3 == side)) {
          tadr0 = tab_addr(0, t * 32);
          TN0 = lds_rd128(tadr0);
          static_for<4>([&](auto si) { fast_iter.template operator()<decltype(si)::value, true>(t + decltype(si)::value, nL2[0], nL2[1]); });
          t += 4;
        }
        // (after a band trip the far side may follow at once: no general iteration in between, it would break the trips' alignment)
        if (t < nt && !(t + 4 <= nt && (t & 3) == 0 && classify(t, side) && classify(t + 3, side3) && side3 == side)) {
          generic_iter(t);
          ++t;
        }
      } else {
        if (t < nt) {
          generic_iter(t);
          ++t;
        }
      }
    }
    FAT5_STAMP(3);
    product_step((uint32_t)(((nt - 1) & 3) * SLOT));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (asm MFMA -> accumulator reads below: see mfma_acc_agpr)
  FAT5_STAMP(4);

  const float scale = a.scale;
  if (stg) {
    // dQ through the wave's (free) Q image: 8-byte pieces into the swizzled row-major image, out again as whole rows -- eight stores
    // of eight full 128-byte rows each instead of sixteen that touch 64 rows
    char* img = smem + Cfg::RING + w * 3 * Cfg::STG_T;
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
  } else {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qrow = qw0 + 32 * qb + lq;
      if (qrow < M) {
        uint16_t* drow = dqb_ + (int64_t)qrow * a.dqs[2];
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            u32x2 wv;
            wv[0] = pack2<BF16>(dq[qb][db][4 * g + 0] * scale, dq[qb][db][4 * g + 1] * scale);
            wv[1] = pack2<BF16>(dq[qb][db][4 * g + 2] * scale, dq[qb][db][4 * g + 3] * scale);
            *reinterpret_cast<u32x2*>(drow + 32 * db + 8 * g + 4 * hi) = wv;
          }
      }
    }
  }
  FAT5_STAMP(12);
  if constexpr (QDG) {
    // ---- partial per-diagonal sums of this row block (after the dQ rows are on their way: the stores drain while the partial row is summed; the images and
    // the diagonal arrays are different LDS areas): row `mblk` of the a.part_stride partial rows of (b, h) ----
    if (want_drpe) {
      diag_flush();
      far_neg = wave_sum(far_neg);
      far_pos = wave_sum(far_pos);
      if (l == 0) {
        sD0[(2 * w) * n1] += far_neg;
        sD0[(2 * w) * n1 + 2 * a.R] += far_pos;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (LDS only: a __syncthreads here would also wait for the dQ stores)
      __builtin_amdgcn_s_barrier();
      FAT5_STAMP(13);
      float* out = a.drpe_part + ((int64_t)(b * a.H + h) * a.part_stride + mblk) * n1;
      for (int i2 = tid; i2 < n1; i2 += NT) {
        float acc = 0.f;
#pragma unroll
        for (int ww = 0; ww < 2 * Cfg::NW; ++ww) acc += sD0[ww * n1 + i2];
        out[i2] = acc;
      }
    }
  }
  FAT5_STAMP(6);
}

template <int D, bool BF16, int BIAS, bool QDG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void attn_bwd_q64_kernel(const AttnArgs a) {
  attn_bwd_q64_body<D, BF16, BIAS, QDG>(a, blockIdx.x);
}

// Both backward kernels in ONE launch for problems whose grids leave the chip's last round mostly empty (mid sequence lengths) or
// do not fill it at all (cfg2: 96 + 96 workgroups): workgroups [0, n_kv_blocks) run the dK/dV body in its self-sufficient form,
// the others the dQ body; one workgroup per CU either way (512 registers per lane), the longer ones first.
// QDG (round 6, T5 bias): the dQ workgroups form the table gradient's per-diagonal sums, the dK/dV ones none (see attn_bwd_q64_body): taken where every workgroup of
// the launch is resident at once, i.e. where the launch lasts as long as its longest workgroup -- a dK/dV one
template <int D, bool BF16, int BIAS, bool QDG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void attn_bwd_fused64_kernel(const AttnArgs a) {
  if ((int)blockIdx.x < a.n_kv_blocks) {
    int b, h, nblk;
    decode_unit(a, blockIdx.x, a.n_nblk, b, h, nblk, (FAT5_CAUSAL_ORDER && a.causal) ? 2 : 0);
    attn_bwd_kv64_body<D, BF16, BIAS, false, true, false, QDG>(a, b, h, nblk, nblk, false);
  } else {
    attn_bwd_q64_body<D, BF16, BIAS, QDG>(a, blockIdx.x - a.n_kv_blocks);
  }
}

template <int D, bool BF16, int BIAS, bool HALF, bool ONE = false, bool NODIAG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void attn_bwd_kv64_kernel(const AttnArgs a) {
  int b, h, nblk;
  decode_unit(a, blockIdx.x, a.n_nblk, b, h, nblk, (FAT5_CAUSAL_ORDER && a.causal) ? 2 : 0);
  // (part_rows2: a 256-key launch over some units of a problem whose other units run half-length -- a unit range of a mixed launch)
  const bool two = !HALF && a.part_rows2;
  attn_bwd_kv64_body<D, BF16, BIAS, HALF, false, ONE, NODIAG>(a, b, h, nblk, two ? 2 * nblk : nblk, two && 2 * nblk + 1 < a.part_stride);
}

// Both variants in one launch.  One workgroup per CU (512 registers per lane): `w` 256-key workgroups take ceil(w / 256) rounds, the
// last of them mostly empty at mid sizes ((4,12,2048,64): 384 = 1.5 rounds).  Here the first a.mix_full (b, h) pairs of every XCD
// run as 256-key workgroups -- whole rounds -- and the others as 128-key half-length ones (two per key block, ~0.65-0.73 of a full
// workgroup's time each), which fill the last round evenly: 1 + 0.7 instead of 2 rounds.  Workgroup -> XCD as in decode_block
// (blockIdx % 8; pairs dealt round-robin), full pairs first inside every XCD.  Pairs are numbered HEAD-major here (u = h * B + b,
// the numbering of unit ranges): the full-length pairs are then the units [0, 8 * mix_full) and a unit-range launch of the same
// problem splits into at most one 256-key and one half-length launch that run every pair through the same body -- sharded and
// unsharded results stay bit-identical.  Partial diagonal sums: a.part_stride = 128-key rows per pair; a 256-key workgroup j
// writes row 2j and zeroes row 2j + 1.
template <int D, bool BF16, int BIAS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void attn_bwd_kv64_mixed_kernel(const AttnArgs a) {
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int nth = a.part_stride, ntf = (a.N + 255) / 256;
  const int nfull = a.mix_full * ntf;
  int pair, tile;
  const bool full = idx < nfull;
  if (full) {
    pair = idx / ntf;
    tile = idx - pair * ntf;
  } else {
    const int i2 = idx - nfull;
    pair = a.mix_full + i2 / nth;
    tile = i2 - (i2 / nth) * nth;
  }
  const int ui = pair * 8 + xcd;
  const int h = ui / a.B, b = ui - h * a.B;
  if (full) attn_bwd_kv64_body<D, BF16, BIAS, false>(a, b, h, tile, 2 * tile, 2 * tile + 1 < nth);
  else attn_bwd_kv64_body<D, BF16, BIAS, true>(a, b, h, tile, tile, false);
}

}  // namespace fat5
