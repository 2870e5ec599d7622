// Dense-bias gradient by in-kernel batch reduction (gfx950).
//
// The reference materialises dS for every batch element -- a (B, H, M, N) tensor in the bias dtype written by `_bwd_kv_kernel`
// (src/model/ops/flash_attention_v2_bias.py:149-162, :725-729) -- and sums it over the batch afterwards (:214-215): O(B*H*S^2)
// memory and traffic (6.4 GB + a second pass at (4,12,8192,64)).  Here a workgroup owns one (head, 32*NW query rows) strip of
// the (1, H, M, N) bias gradient and loops over the batch INSIDE: per 64-key tile it recomputes, for each batch element,
//     S^T = K Q^T,   P = exp2(S^T c2 + bias log2e - L log2e),   dP^T = V dO^T - delta,   dS^T = P * dP^T
// (the dQ body's orientation and arithmetic, attn_bwd.h), rounds dS^T to the input dtype exactly like the reference (:720) and
// adds it to an fp32 register tile; after the last batch element the tile is rounded once and leaves as whole 128-byte rows.
// Cost: two GEMMs per (batch element, tile) that the dQ / dK/dV bodies also compute -- in exchange the workspace drops to
// O(B*H*S) (delta), nothing of size B*H*M*N is written or re-read, and each bias tile is fetched once for all B elements.
//
// Q / dO fragments of up to BC = 4 batch elements live in registers; larger batches run in chunks of BC whose partial sums
// pass through an fp32 (H, M, N) scratch (read-modify-write by the owning workgroup only: no atomics, fixed order).
//
// SPLIT (round 4, D <= 64): the chunk's batch elements are divided between TWO groups of NW waves -- group g takes elements g, g + 2 --
// that walk the same (strip, key tile) sequence side by side, each with its own double-buffered K / V stage, sharing the bias tile.
// A wave then holds the fragments of two elements (64 registers at D = 64) and the workgroup runs at two waves per SIMD instead of
// one: one group's MFMAs overlap the other's softmax VALU.  After the tile's last step group 1 hands its fp32 partial tile to group 0
// through LDS (fixed order: (e0 + e2) + (e1 + e3)), which rounds and stores as before.
#pragma once
#include "attn_common.h"
#include "attn_fwd.h"  // load_bias_block

namespace fat5 {

template <int D, int NW, bool SPLIT = false>
struct BwdDbiasCfg {
  static constexpr int BM = 32 * NW, BN = 64, NT = 64 * NW, BC = 4;  // (NT: threads of ONE group)
  static constexpr int NG = SPLIT ? 2 : 1, BCW = BC / NG;            // groups; batch elements of a chunk per wave
  static constexpr int KRM = rm_bytes<D, BN>();
  static constexpr int STAGE = 2 * KRM;      // K + V tile of one (batch element, key tile)
  static constexpr int BIASB = BM * BN * 2;  // one (BM x 64) 16-bit bias tile
  static constexpr int MERGE = SPLIT ? NT * 32 * 4 : 0;  // group 1's fp32 partial tile: 32 values per lane
  static size_t smem() { return NG * 2 * STAGE + 2 * BIASB + MERGE; }
};

template <int D, bool BF16, int NW, bool SPLIT>
FAT5_DEV void attn_bwd_dbias_body(const AttnArgs& a, uint16_t* __restrict__ dbias, float* __restrict__ scratch) {
  using Cfg = BwdDbiasCfg<D, NW, SPLIT>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT, BC = Cfg::BC, BCW = Cfg::BCW, NG = Cfg::NG;
  constexpr int KK = D / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sB = smem + NG * 2 * Cfg::STAGE;  // [2][BM][64] 16-bit bias tiles (tile parity)
  [[maybe_unused]] char* sX = sB + 2 * Cfg::BIASB;  // SPLIT: [NT][8] f32x4, group 1's partial tile

  // g = group (wave-uniform), w = 32-row block of the strip, tid = thread index inside the group
  const int wall = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int g = SPLIT ? wall / NW : 0, w = wall - g * NW;
  const int tid = (int)threadIdx.x - g * NT, l = tid & 63, lq = l & 31, hi = l >> 5;
  // blockIdx.x = (head, query strip, key split): the key tiles of a strip are independent pieces of the output, so short
  // sequences split them over a.n_nblk workgroups to fill the chip (a.n_nblk = 1: one workgroup walks the whole strip)
  const int nsplit = a.n_nblk;
  const int sp = blockIdx.x % nsplit, hm = blockIdx.x / nsplit;
  const int h = hm / a.n_mblk, mblk = hm - h * a.n_mblk;
  const int M = a.M, N = a.N, B = a.B;
  const int m0 = mblk * BM;
  const int P = N - M;
  int n_end = N;
  if (a.causal) n_end = min(N, m0 + BM + P);
  const int nt_all = n_end > 0 ? (n_end + BN - 1) / BN : 0;           // visited key tiles of the strip
  const int nt_cap = (N + BN - 1) / BN;                                // all key tiles (causal: the rest is zero-filled)
  const int per = (nt_cap + nsplit - 1) / nsplit;
  const int t_lo = sp * per, t_hi_cap = min(nt_cap, t_lo + per);       // this workgroup's tiles [t_lo, t_hi_cap)
  const int nt = max(t_lo, min(nt_all, t_hi_cap));                     // ... of which [t_lo, nt) are visited
  const int qrow0 = m0 + 32 * w, qrow = qrow0 + lq, qrow_c = min(qrow, M - 1);

  // bias tile staging (shared by all batch elements of a key tile) and this lane's reader
  using BDma = DmaStage<BN, BM, NT, true>;
  BDma bdm;
  BiasTileReader brd;
  const bool bias_dma = a.bias_dma != 0;
  const bool bias_issuer = g == 0;  // (the bias tile is shared: group 0 fetches it; every barrier is behind the issuing waves' vmcnt(0))
  const uint16_t* bias_h = a.bias + (int64_t)h * a.bs[1];
  const uint16_t* brow = bias_h + (int64_t)qrow_c * a.bs[2];
  __amdgpu_buffer_rsrc_t brs = make_rows_rsrc(bias_h + (int64_t)m0 * a.bs[2], a.bs[2], bias_dma ? M - m0 : 0, N);
  bdm.init(a.bs[2], tid);
  brd.init(32 * w + lq, hi);
  uint16_t* dbtile = dbias + (int64_t)h * M * N;          // (1, H, M, N) contiguous
  uint16_t* dbrow = dbtile + (int64_t)qrow_c * N;
  float* scrow = scratch ? scratch + ((int64_t)h * M + qrow_c) * N : nullptr;
  const bool rows16 = (N % 8 == 0) && ((reinterpret_cast<uintptr_t>(dbias) & 15) == 0);

  FragAddr<D> fa;
  fa.init(l);
  DmaStage<D, BN, NT> kst, vst;
  kst.init(a.ks[2], tid, a.dvalid);
  vst.init(a.vs[2], tid, a.dvalid);
  const uint32_t kstride_b = (uint32_t)a.ks[2] * 2u, vstride_b = (uint32_t)a.vs[2] * 2u;
  const float c2 = a.scale * kLog2e;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  for (int c0 = 0; c0 < B; c0 += BC) {
    const int nbc = min(BC, B - c0);
    const bool first_chunk = c0 == 0, last_chunk = c0 + BC >= B;
    // ---- this chunk's per-batch-element operands: Q, dO fragments (B operands), -L log2e, -delta ----
    // this wave's elements of the chunk: local index I <-> chunk element EL(I) (SPLIT: g, g + 2)
    auto EL = [&](int I) { return SPLIT ? g + NG * I : I; };
    const int nloc = SPLIT ? (nbc - g + 1) / 2 : nbc;     // elements this group works on
    const int nsteps = SPLIT ? (nbc + 1) / 2 : nbc;       // steps per tile (the larger group's count)
    u32x4 qf[BCW][KK], dof[BCW][KK];
    float nL2[BCW];
    float ndel[BCW];
#pragma unroll
    for (int i = 0; i < BCW; ++i) {
      const int b = min(c0 + EL(i), B - 1);
      const uint16_t* qb = a.q + (int64_t)b * a.qs[0] + (int64_t)h * a.qs[1] + (int64_t)qrow_c * a.qs[2];
      const uint16_t* dob = a.dout + (int64_t)b * a.dos[0] + (int64_t)h * a.dos[1] + (int64_t)qrow_c * a.dos[2];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        qf[i][kk] = load_frag16(qb, kk, hi, a.dvalid);
        dof[i][kk] = load_frag16(dob, kk, hi, a.dvalid);
      }
      const int64_t so = ((int64_t)b * a.H + h) * M + qrow_c;
      const float Lq = a.lse[so];
      nL2[i] = (Lq < kDeadRowLse) ? -INFINITY : -Lq * kLog2e;
      ndel[i] = -a.delta[so];
    }
    // K / V descriptors of the chunk's batch elements (wave-uniform)
    auto kv_issue = [&](int t, int i, int buf) {  // (i: local element index; nothing to fetch when the group has no such element)
      if (i >= nloc) return;
      const int b = c0 + EL(i);
      const uint16_t* kb_ = a.k + (int64_t)b * a.ks[0] + (int64_t)h * a.ks[1];
      const uint16_t* vb_ = a.v + (int64_t)b * a.vs[0] + (int64_t)h * a.vs[1];
      const __amdgpu_buffer_rsrc_t krs = make_rows_rsrc(kb_, a.ks[2], N, a.dvalid);
      const __amdgpu_buffer_rsrc_t vrs = make_rows_rsrc(vb_, a.vs[2], N, a.dvalid);
      char* dst = smem + (g * 2 + buf) * Cfg::STAGE;
      kst.issue(krs, (uint32_t)(t * BN) * kstride_b, dst, tid);
      vst.issue(vrs, (uint32_t)(t * BN) * vstride_b, dst + Cfg::KRM, tid);
    };
    __syncthreads();  // (previous chunk's last readers of the LDS buffers)
    if (nt > t_lo) {
      kv_issue(t_lo, 0, 0);
      if (bias_dma && bias_issuer) bdm.issue(brs, (uint32_t)(t_lo * BN) * 2u, sB + (t_lo & 1) * Cfg::BIASB, tid);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BCW; ++i)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) asm volatile("" ::"v"(qf[i][kk]), "v"(dof[i][kk]));  // (see attn_fwd.h: waitcnt model)

    int j = 0;  // running (tile, batch element) step: K/V buffer = j & 1
    for (int t = t_lo; t < nt; ++t) {
      const int n0 = t * BN;
      const char* sBt = sB + (t & 1) * Cfg::BIASB;
      f32x16 acc[2] = {zero16, zero16};
      // one batch element of this tile: recompute dS^T, round like the reference, accumulate
      auto one = [&]<int I>() {
        const int buf = j & 1;
        f32x16 ndelta16;  // dP^T accumulators start at -delta (rebuilt per step: four live copies would cost 64 registers)
#pragma unroll
        for (int r = 0; r < 16; ++r) ndelta16[r] = ndel[I];
        const char* sK = smem + (g * 2 + buf) * Cfg::STAGE;
        const char* sV = sK + Cfg::KRM;
        // next (tile, element): its K/V into the other buffer; a new tile also brings its bias tile
        const bool last_i = (I + 1 >= nsteps);
        if (!last_i) {
          kv_issue(t, I + 1, buf ^ 1);
        } else if (t + 1 < nt) {
          kv_issue(t + 1, 0, buf ^ 1);
          if (bias_dma && bias_issuer) bdm.issue(brs, (uint32_t)(n0 + BN) * 2u, sB + ((t + 1) & 1) * Cfg::BIASB, tid);
        }
        if (I < nloc)
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
            s = mfma32<BF16>(kf[kk], qf[I][kk], kk == 0 ? zero16 : s);
            dp = mfma32<BF16>(vf[kk], dof[I][kk], kk == 0 ? ndelta16 : dp);
          }
          float bv[16];
          if (bias_dma) brd.template load<BF16>(sBt, kb, bv);
          else load_bias_block<BF16>(brow, nb, hi, N, a.bias_vec4 && (nb + 32 <= N), bv);
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = fast_exp2(fmaf(s[r], c2, bias_log2(bv[r]) + nL2[I])) * dp[r];
          const bool nmask = nb + 32 > N;
          const bool cmask = a.causal && (nb + 31 > qrow0 + P);
          if (nmask || cmask) {
            const int lim = a.causal ? min(N - 1, qrow + P) : N - 1;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (nb + crow(r, hi) > lim) s[r] = 0.f;
          }
          // the reference rounds every batch element's dS to the bias dtype before the batch sum (:720, :214)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const uint32_t pk = pack2<BF16>(s[r], s[r + 1]);
            acc[kb][r] += cvt_lo<BF16>(pk);
            acc[kb][r + 1] += cvt_hi<BF16>(pk);
          }
        }
        __syncthreads();  // this step's K/V buffer may be refilled; the next step's tiles have landed (vmcnt(0) first)
        ++j;
      };
      one.template operator()<0>();
      if (nsteps > 1) one.template operator()<1>();
      if constexpr (BCW > 2) {
        if (nsteps > 2) one.template operator()<2>();
        if (nsteps > 3) one.template operator()<3>();
      }
      if constexpr (SPLIT) {
        // group 1's partial tile joins group 0's (every step ends with a barrier: the hand-over area's previous readers are done)
        f32x4* xw = reinterpret_cast<f32x4*>(sX) + tid;
        if (g == 1) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) xw[(kb * 4 + q4) * NT] = f32x4{acc[kb][4 * q4], acc[kb][4 * q4 + 1], acc[kb][4 * q4 + 2], acc[kb][4 * q4 + 3]};
        }
        __syncthreads();
        if (g == 0 && nbc > 1) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const f32x4 x = xw[(kb * 4 + q4) * NT];
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[kb][4 * q4 + e] += x[e];
            }
        }
      }

      // ---- combine with earlier chunks, hand the tile on (group 0) ----
      if (g == 0)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int nb = n0 + 32 * kb;
        if (!first_chunk && qrow < M) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = nb + crow(r, hi);
            if (n < N) acc[kb][r] += scrow[n];
          }
        }
        if (!last_chunk) {
          if (qrow < M) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int n = nb + crow(r, hi);
              if (n < N) scrow[n] = acc[kb][r];
            }
          }
          continue;
        }
        u32x4 dsv[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) dsv[t2] = pack8<BF16>(acc[kb], t2);
        const bool via_lds = bias_dma && rows16 && (n0 + BN <= N);
        if (via_lds) {
          brd.store(const_cast<char*>(sBt), kb, dsv);  // park in the consumed bias tile (this wave's rows only)
        } else if (qrow < M) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = nb + crow(r, hi);
            if (n < N) dbrow[n] = to16<BF16>(acc[kb][r]);
          }
        }
      }
      if (g == 0 && last_chunk && bias_dma && rows16 && (n0 + BN <= N)) {
        // this wave's 32 rows x 128 bytes leave as whole rows: 16-byte pieces, 8 lanes per row
        const char* tl = sBt + (32 * w) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int jj = l + 64 * i, row = jj >> 3, slot = jj & 7;
          const u32x4 vv = *reinterpret_cast<const u32x4*>(tl + row * 128 + slot * 16);
          const int m = m0 + 32 * w + row;
          const int n = n0 + ((slot ^ swz<64>(row)) << 3);
          if (m < M) *reinterpret_cast<u32x4*>(dbtile + (int64_t)m * N + n) = vv;
        }
      }
      // the bias buffer of this parity is refilled by the DMA issued at the top of tile t+1's LAST step -- which is its first
      // step when the chunk holds one batch element: every wave's copy-out above must be behind a barrier by then
      if (last_chunk && bias_dma) __syncthreads();
    }
    // causal: key tiles above the diagonal are never visited; their gradient is zero (reference zero-fills ds, :153,:160)
    if (g == 0 && last_chunk && a.causal && qrow < M) {
      for (int n = nt * BN + 4 * hi; n < min(N, t_hi_cap * BN); n += 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < N) dbrow[n + e] = 0;
      }
    }
  }
}

template <int D, bool BF16, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(1)))  // (Q / dO fragments of four batch elements: 128 registers; two waves per SIMD would spill 129)
void attn_bwd_dbias_kernel(const AttnArgs a, uint16_t* dbias, float* scratch) {
  attn_bwd_dbias_body<D, BF16, NW, false>(a, dbias, scratch);
}
// two groups of NW waves sharing the batch: two waves per SIMD
template <int D, bool BF16, int NW>
__global__ __launch_bounds__(128 * NW) __attribute__((amdgpu_waves_per_eu(2, 2)))
void attn_bwd_dbias_split_kernel(const AttnArgs a, uint16_t* dbias, float* scratch) {
  attn_bwd_dbias_body<D, BF16, NW, true>(a, dbias, scratch);
}

}  // namespace fat5
