// Common device helpers for the gfx950 attention kernels (fwd, bwd_kv, bwd_q).
// CDNA4 only: 64-wide wavefronts, v_mfma_f32_32x32x16_{bf16,f16}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "../../include/fat5.h"

namespace fat5 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

#define FAT5_DEV __device__ __forceinline__

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, typename F>
FAT5_DEV void static_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// ------------------------------------------------------------------------------------------
// MFMA 32x32x16 (A: 32x16, B: 16x32, C/D: 32x32 fp32).
//   A lane l: row  = l & 31, k = 8*(l >> 5) + j, j = 0..7   (8 x 16-bit = one u32x4)
//   B lane l: col  = l & 31, k = 8*(l >> 5) + j
//   C lane l: col  = l & 31, row = crow(r, l >> 5) = (r & 3) + 8*(r >> 2) + 4*(l >> 5), r = 0..15
// The k-slot <-> "real" contraction index mapping is free as long as A and B agree; the kernels
// exploit this to feed a C-layout tile straight back in as an operand with no cross-lane traffic.
// ------------------------------------------------------------------------------------------
template <bool BF16>
FAT5_DEV f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
  if constexpr (BF16) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
// MFMA 16x16x32 (A: 16x32, B: 32x16, C/D: 16x16 fp32): B lane l: col = l & 15, eight k values per lane, the four
// 16-lane groups cover k = 0..31; C lane l: col = l & 15, rows 4*(l >> 4) + i.
template <bool BF16>
FAT5_DEV f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (BF16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
}

FAT5_DEV constexpr int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <bool BF16>
FAT5_DEV uint32_t pack2(float a, float b) {
  if constexpr (BF16) {
    bf16x2_t r = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, r);
  } else {
    f16x2_t r = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, r);
  }
}

template <bool BF16>
FAT5_DEV float cvt_lo(uint32_t w) {  // low 16 bits -> fp32
  if constexpr (BF16) {
    return __uint_as_float(w << 16);
  } else {
    f16x2_t r = __builtin_bit_cast(f16x2_t, w);
    return (float)r[0];
  }
}
template <bool BF16>
FAT5_DEV float cvt_hi(uint32_t w) {  // high 16 bits -> fp32
  if constexpr (BF16) {
    return __uint_as_float(w & 0xffff0000u);
  } else {
    f16x2_t r = __builtin_bit_cast(f16x2_t, w);
    return (float)r[1];
  }
}
// x as two 16-bit terms of the operand dtype, x ~ hi + lo (the dense bodies' 1 / scale on the matrix pipe): bf16 by truncation (8 + 8 mantissa bits, 2^-17 relative),
// fp16 by rounding (11 bits + whatever the second term still resolves: never worse than one fp16 rounding of the bias itself); host and device
template <bool BF16>
__host__ __device__ inline void split16(float x, uint32_t& hi, uint32_t& lo) {
  if constexpr (BF16) {
    uint32_t xb, rb;
    __builtin_memcpy(&xb, &x, 4);
    hi = xb >> 16;
    const uint32_t hb = hi << 16;
    float hf;
    __builtin_memcpy(&hf, &hb, 4);
    const float r = x - hf;
    __builtin_memcpy(&rb, &r, 4);
    lo = rb >> 16;
  } else {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    uint16_t hb, lb;
    __builtin_memcpy(&hb, &h, 2);
    __builtin_memcpy(&lb, &l, 2);
    hi = hb;
    lo = lb;
  }
}
// ... and whether x is ONE such term exactly (the one-selector-term kernels)
template <bool BF16>
__host__ __device__ inline bool is_one16(float x) {
  uint32_t hi, lo;
  split16<BF16>(x, hi, lo);
  return (lo & 0x7fffu) == 0u;
}

template <bool BF16>
FAT5_DEV float cvt16(uint16_t h) {
  return cvt_lo<BF16>((uint32_t)h);
}
template <bool BF16>
FAT5_DEV uint16_t to16(float a) {
  return (uint16_t)(pack2<BF16>(a, 0.f) & 0xffffu);
}

// dense bias in log2 units.  A bias of finfo(bf16).min (what `use_masking` writes, reference modeling_flash_t5.py:266-270) times
// log2e overflows fp32; the reference scales (s - m) instead and stays finite, so a fully masked row is a uniform softmax
// there -- keep the product finite to give the same.
// The clamp is applied to the PACKED 16-bit words (one v_pk_min_u16 per two values: a bf16 below -1.99e38 = 0xFF16 has a larger
// unsigned pattern; positive values and every fp16 value -- finfo.min = -65504 -- are unaffected), then one multiply per value.
typedef __attribute__((ext_vector_type(2))) uint16_t u16x2_t;
template <bool BF16>
FAT5_DEV uint32_t bias_clamp2(uint32_t w) {  // two packed bias values
  if constexpr (BF16) {
    const u16x2_t lim = {(uint16_t)0xFF16u, (uint16_t)0xFF16u};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, w), lim));
  } else {
    return w;
  }
}
template <bool BF16>
FAT5_DEV uint16_t bias_clamp1(uint16_t h) { return BF16 ? (h < (uint16_t)0xFF16u ? h : (uint16_t)0xFF16u) : h; }
FAT5_DEV float bias_log2(float b) { return b * kLog2e; }  // (b already clamped in its 16-bit form)
// Bias on the matrix pipe (S' = Q K^T + E B with a 0 / (1 / scale) selector E: attn_bwd64.h, attn_bwd_qdb64.h): the raw 16-bit bias words are MFMA operands, so a
// -inf entry (an additive mask written as -inf instead of finfo.min) would meet the selector's zeros as -inf * 0 = NaN and poison every score of its k-slot group
// (ADVICE r5).  The words are clamped in their packed form first, one v_pk_min_u16 per two values like bias_clamp2: bf16 to -min(2e38, 2e38 |scale|) -- the product
// with 1 / scale then stays finite in fp32 as well --, fp16 to finfo.min (-65504 / scale is far inside fp32).  NaN words with the sign bit set clamp too; positive
// values are untouched.  These bodies recompute p = exp2(S' c2) with -L / scale already inside S': a masked key ends at ~ -2e38 -> p = 0, no difference of two huge
// numbers is ever formed.  (The FORWARD keeps its per-element bias add -- round 6 tried the selector form in the 32-row body, profiles/r06_dense_fwd_ab.log: -4 % at
// (4,12,512), +4 % at (4,12,2048), +11 % at the reference's B = 16 causal shape with its two-term scale 1.3 -- the body is not VALU-bound at two waves per SIMD -- and
// a row masked ENTIRELY by finfo.min came out wrong: the MFMA accumulation of q.k onto -2e38 does not absorb the small term symmetrically the way the fp32 FMA's
// round-to-nearest does (results one ulp = 2e31 apart by the sign of q.k), so x - max(x) is no longer 0 on such a row.)
template <bool BF16>
FAT5_DEV uint32_t bias_mfma_limit(float scale) {
  if constexpr (BF16) {
    const uint32_t h = __float_as_uint(-fminf(2.0e38f, 2.0e38f * fabsf(scale))) >> 16;  // (truncated: the smaller magnitude)
    return h | (h << 16);
  } else {
    return 0xFBFFFBFFu;
  }
}
FAT5_DEV uint32_t pk_min_u16(uint32_t w, uint32_t lim2) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, w), __builtin_bit_cast(u16x2_t, lim2)));
}
FAT5_DEV u32x2 bias_clamp_frag(u32x2 f, uint32_t lim2) { return u32x2{pk_min_u16(f[0], lim2), pk_min_u16(f[1], lim2)}; }
FAT5_DEV u32x4 bias_clamp_frag(u32x4 f, uint32_t lim2) {
  return u32x4{pk_min_u16(f[0], lim2), pk_min_u16(f[1], lim2), pk_min_u16(f[2], lim2), pk_min_u16(f[3], lim2)};
}
// backward: a query row with lse below this has every key masked (by the causal rule: -inf; by a finfo.min bias: ~ -2e38);
// its probabilities are treated as zero (dq = 0 for the row, no contribution to dk / dv / dbias)
constexpr float kDeadRowLse = -1.0e30f;
FAT5_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
FAT5_DEV float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// max / sum over the lane pair (lane, lane ^ 32) with v_permlane32_swap: a VALU op (~2 issue slots) instead of a
// ds_bpermute round trip through the LDS crossbar (~100+ cycles of exposed latency in front of the rescale branch)
FAT5_DEV float pair_max(float v) {
  const uint32_t u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
FAT5_DEV float pair_sum(float v) {
  const uint32_t u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Sum over the 64 lanes of a wave, result in every lane, fixed order: DPP quad / mirror steps inside each row of 16,
// then v_permlane16_swap / v_permlane32_swap across rows -- six VALU-side steps instead of six ds_bpermute round trips.
template <int CTRL>
FAT5_DEV float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
FAT5_DEV float wave_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]  (lane ^ 1)
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]  (lane ^ 2)
  v = dpp_add<0x141>(v);  // row_half_mirror: the other quad of each 8
  v = dpp_add<0x140>(v);  // row_mirror: the other 8 of each 16
  {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return pair_sum(v);
}

// ------------------------------------------------------------------------------------------
// LDS images.
// Row-major [rows][D] 16-bit tile with 16-byte chunks XOR-swizzled so that a ds_read_b128 of
// "32 different rows, same chunk" (the MFMA A/B fragment read) is bank-conflict free.
//   bank row = 256 B = 16 chunks; a row holds C = D/8 chunks; rows per bank row = 16 / C.
// ------------------------------------------------------------------------------------------
template <int D>
FAT5_DEV constexpr int swz(int row) {
  // One XOR pattern serves both read shapes of a row-major image:
  //  * ds_read_b128 fragment reads (16 rows that differ mod 16, same chunk)  -> 16 distinct 16-byte slots
  //  * ds_read_b64_tr_b16 transposed reads (rows 4a..4a+3, 64 contiguous bytes each) -> 4 distinct 64-byte
  //    quarters of the 256-byte bank row
  constexpr int C = D / 8;
  if constexpr (C == 16) return ((row & 3) << 2) | ((row >> 2) & 3);
  else if constexpr (C == 8) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
  else return (row * C / 16) & (C - 1);
}
template <int D>
FAT5_DEV int rm_off(int row, int chunk) {  // byte offset of 16-B chunk `chunk` of row `row`
  return row * (2 * D) + ((chunk ^ swz<D>(row)) << 4);
}

// global 16-byte load of 8 consecutive 16-bit elements; zero when !valid
FAT5_DEV u32x4 gload16(const uint16_t* p, bool valid) {
  u32x4 z = {0, 0, 0, 0};
  return valid ? *reinterpret_cast<const u32x4*>(p) : z;
}

// ------------------------------------------------------------------------------------------
// XCD-aware work-item decode.  Workgroup `bid` lands on XCD bid % 8 (observed, speed only).
// All tiles of one (b,h) are given to one XCD so K/V (fwd, bwd_q) or Q/dO (bwd_kv) of that pair
// stay in that XCD's L2.  Bijective for every (nbh, ntile).
// ------------------------------------------------------------------------------------------
FAT5_DEV int fast_div(int n, int d, uint32_t mg) { return mg ? (int)__umulhi((uint32_t)n, mg) : n / d; }
__host__ inline uint32_t div_magic(long d, long n_max) {  // n_max: the largest dividend
  if (d <= 1 || d >= (1L << 31) || n_max < 0 || (unsigned long long)n_max * (unsigned long long)d >= (1ull << 32)) return 0u;
  return (uint32_t)(((1ull << 32) + (unsigned long long)d - 1) / (unsigned long long)d);
}
// order (causal problems: the tiles of a pair have unequal lengths): 0 = pair-major, a pair's tiles side by side (its K / V stay in the XCD's L2);
// 1 = tile-major, the LAST tile of every pair of the XCD first (row blocks of a causal problem: the ones that see the most keys); 2 = tile-major, the FIRST tile
// first (key blocks: the ones most rows see) -- longest-first list scheduling: in launch order the dispatcher hands a freed CU the next workgroup, and with
// pair-major order the last workgroups to start include full-length ones (24 pairs x 4 row blocks on the 32 CUs of an XCD: 51 tile-times against 43
// longest-first, 41 ideal).  Measured, causal forward (profiles/r05_causal_order_fwd_ab.log, us): (16,12,1024,128) dense 123.8 -> 101.3, (4,12,2048,128) 81.9 -> 62.6,
// (16,12,1024,64) dense 73.5 -> 59.6, (4,12,2048,64) 48.1 -> 39.9, (4,12,4096,64) 140.7 -> 116.0 -- 11 .. 33 % on every causal shape but one ((16,12,512) T5 table: +2 %).
// Results do not depend on the order (every workgroup computes its own tile).
FAT5_DEV void decode_block(int bid, int nbh, int ntile, int& bh, int& tile, uint32_t mg_tile = 0, int order = 0) {
  const int nx = 8;
  if ((nbh % nx) == 0) {
    const int xcd = bid % nx;
    const int idx = bid / nx;           // sequence number inside this XCD
    const int per = nbh / nx;           // (b,h) pairs per XCD
    int pair;
    if (order) {
      const int tq = idx / per;
      pair = idx - tq * per;
      tile = order == 1 ? ntile - 1 - tq : tq;
    } else {
      pair = fast_div(idx, ntile, mg_tile);  // which of this XCD's pairs
      tile = idx - pair * ntile;
    }
    bh = pair * nx + xcd;               // pairs dealt round-robin to XCDs
  } else if (order) {
    const int tq = bid / nbh;
    bh = bid - tq * nbh;
    tile = order == 1 ? ntile - 1 - tq : tq;
  } else {
    bh = fast_div(bid, ntile, mg_tile);
    tile = bid - bh * ntile;
  }
}


// ------------------------------------------------------------------------------------------
// RPE table in LDS: FOUR copies of the (2R+1) log2-scaled entries, copy c shifted left by c elements
// (copy_c[m] = T[m + c]).  A lane's 16 gathers per 32x32 block are four runs of 4 consecutive entries whose
// alignment (mod 4) is a per-lane constant, so it reads its own copy with 4 aligned ds_read_b128 instead of
// 16 ds_read_b32.  Copy 0 doubles as the plain table (constants, clamped edge path).
// Every copy is PADDED by kRpePad saturated entries on both sides (T[0] below, T[2R] above): a block that is partly or
// entirely outside the band reads the same way as one inside it -- one straight-line bias path per block, the window
// start clamped into the table (rpe_clamp_*) instead of a three-way branch over far / interior / edge blocks that keeps
// hipcc from scheduling across it.  Kernels hold `sT` = raw base + kRpePad, so that sT[d + R] is entry d of copy 0 for
// d in [-R - kRpePad, R + kRpePad] and copy c starts at sT + c * rpe_n1p(R) - c (the index expressions of the unpadded layout).
// ------------------------------------------------------------------------------------------
constexpr int kRpePad = 96;  // >= 63 (an edge block's overhang) + 32 (a clamped window stays in the constant region); multiple of 4
__host__ __device__ constexpr int rpe_n1p(int R) { return (2 * R + 1 + 2 * kRpePad + 3) & ~3; }
__host__ __device__ constexpr size_t rpe_table_bytes(int R) { return (size_t)4 * rpe_n1p(R) * 4; }
// window start (relative to sT + copy * rpe_n1p(R), a multiple of 4) of a lane's 28-entry run, clamped so that the run stays
// inside the padded copy: ascending runs [pos, pos + 27], descending runs [pos - 24, pos + 3]
FAT5_DEV int rpe_clamp_asc(int pos, int R) { return min(max(pos, -kRpePad), rpe_n1p(R) - kRpePad - 28); }
FAT5_DEV int rpe_clamp_desc(int pos, int R) { return min(max(pos, -kRpePad + 24), rpe_n1p(R) - kRpePad - 4); }
// `sT_raw`: the table's raw LDS base (16-byte aligned)
FAT5_DEV void rpe_table_fill(float* sT_raw, const float* rpe1d_h, int R, int tid, int nthreads) {
  const int n1 = 2 * R + 1, n1p = rpe_n1p(R);
  // Copy by copy (no division by the run-time copy length: ~40 instructions per entry in the first version), two entries of each of
  // the four copies per round: eight global loads in flight (a prologue that waits for each load before the next costs a memory
  // round trip per iteration: +4.6 k cycles measured at cfg2)
  for (int m0 = tid; m0 < n1p; m0 += 2 * nthreads) {
    float vv[8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int m = m0 + u * nthreads;
        vv[2 * c + u] = rpe1d_h[min(max(m + c - kRpePad, 0), n1 - 1)];
      }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int m = m0 + u * nthreads;
        if (m < n1p) sT_raw[c * n1p + m] = vv[2 * c + u] * kLog2e;
      }
  }
}

// The same fill in two halves for the prologues that have other requests to send first: the loads of the first round go out at the top
// of the kernel (eight per thread, values parked in registers), the LDS stores -- and further rounds of a long table -- follow where the
// single-call form would sit.  One memory round trip less on the way to the first tile (cfg2: ~1 k cycles per workgroup).
struct RpeTableRegs { float vv[8]; };
FAT5_DEV RpeTableRegs rpe_table_load_first(const float* rpe1d_h, int R, int tid, int nthreads) {
  const int n1 = 2 * R + 1;
  RpeTableRegs r;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) r.vv[2 * c + u] = rpe1d_h[min(max(tid + u * nthreads + c - kRpePad, 0), n1 - 1)];
  return r;
}
// dcut: entries of relative positions d = key - row ABOVE dcut are stored as -inf (p = 0, dS = 0): with a causal mask whose diagonal lies inside the
// band (d <= P visible, P < R) the masked elements of a diagonal step then need no instruction of their own -- the step runs the pipelined band iteration
FAT5_DEV void rpe_table_fill_rest(float* sT_raw, const float* rpe1d_h, int R, int tid, int nthreads, const RpeTableRegs& first, int dcut = 0x7fffffff) {
  const int n1 = 2 * R + 1, n1p = rpe_n1p(R);
  const int mcut = dcut == 0x7fffffff ? 0x7fffffff : dcut + R + kRpePad;  // (copy c, entry m holds d = m + c - kRpePad - R)
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = tid + u * nthreads;
      if (m < n1p) sT_raw[c * n1p + m] = (m + c > mcut) ? -INFINITY : first.vv[2 * c + u] * kLog2e;
    }
  for (int m0 = tid + 2 * nthreads; m0 < n1p; m0 += 2 * nthreads) {
    float vv[8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int u = 0; u < 2; ++u) vv[2 * c + u] = rpe1d_h[min(max(m0 + u * nthreads + c - kRpePad, 0), n1 - 1)];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int m = m0 + u * nthreads;
        if (m < n1p) sT_raw[c * n1p + m] = (m + c > mcut) ? -INFINITY : vv[2 * c + u] * kLog2e;
      }
  }
}

}  // namespace fat5

namespace fat5 {

// ------------------------------------------------------------------------------------------
// Device-side argument block (built by the host from fat5_attn_params).
// ------------------------------------------------------------------------------------------
struct AttnArgs {
  const uint16_t *q, *k, *v, *dout;
  uint16_t *o, *dq, *dk, *dv;
  float* lse;
  float* delta;             // (B,H,M) fp32 scratch (bwd)
  float* stat2;             // (B*H, ceil(M/32), 2, 32) fp32 scratch: -L/scale and -delta per 32-row step, the form the 64-key dK/dV body DMAs (or nullptr)
  const uint16_t* bias;     // dense
  uint16_t* ds_out;         // dense dS output (B', H', M, N) or nullptr
  const float* rpe1d;       // (H, 2R+1)
  float* drpe_part;         // (B*H*nblk_kv, 2R+1) partial diagonal sums or nullptr
  const int32_t *cu_q, *cu_k;
  int64_t qs[3], ks[3], vs[3], os[3], dos[3], dqs[3], dks[3], dvs[3];
  int64_t bs[3];            // bias strides [b,h,m]
  int64_t dss[3];           // ds_out strides [b,h,m]
  int32_t B, H, M, N;
  int32_t total_q, total_k;
  int32_t causal, R;
  int32_t bias_vec4;        // dense bias rows can be read with aligned 8-byte loads
  int32_t ds_vec4;          // dense dS rows can be written with aligned 8-byte stores
  int32_t ds_vec8;          // ... and with aligned 16-byte stores (rows start 16-byte aligned)
  int32_t bias_dma;         // dense bias rows are 16-byte aligned: tiles can go global -> LDS directly
  int32_t n_mblk, n_nblk;   // tiles per (b,h) for the m-parallel / n-parallel kernels
  int32_t n_kv_blocks;      // fused backward launch: workgroups [0, n_kv_blocks) run the dK/dV body
  int32_t mix_full;         // attn_bwd_kv64_mixed_kernel: (b, h) pairs per XCD that run as 256-key workgroups
  int32_t part_stride;      // 64-key dK/dV bodies: partial diagonal-sum rows per (b, h) (= n_nblk unless 256-key and half-length workgroups share a problem)
  int32_t part_rows2;       // ... and a 256-key workgroup j owns rows 2j, 2j + 1 of them
  int32_t diag_q;           // one-launch 64-wide backward, T5 bias: the dQ workgroups form the partial diagonal sums (part_stride = their row blocks per (b, h)), the dK/dV ones none
  int32_t unit_begin, unit_count;  // > 0: only units [unit_begin, +unit_count), u = h * B + b (include/fat5.h)
  int32_t batch_inner;      // dense bias shared by the batch: the B workgroups of one (head, tile) run side by side on one XCD
  uint32_t mg_mblk, mg_nblk, mg_H;   // 2^32 / d rounded up for d = n_mblk, n_nblk, H, or 0: the launchers fill them where the workgroup index is small enough for
                                     // q = umulhi(n, mg) to be the exact quotient (n d < 2^32); the index decode of a workgroup is then two instructions per division
                                     // instead of ~35 (a prologue of ~500 scalar instructions is 2.7 k cycles of the 20 k a cfg2 forward workgroup lives)
  uint32_t mg_mix[4];                // ... of the mixed forward launch's divisors a_lo + 1, a_lo, b_hi, b_lo
  int32_t mix_na, mix_a_lo, mix_k_hi;  // forward, mixed launch (attn_fwd64_mixed_kernel): 256-row workgroups in total / per pair (low) / pairs per XCD with one more
  int32_t dvalid;           // valid head-dim columns: = D except head_dim 16, which runs the D = 32 instantiations with columns 16..31 read as zeros and never written
  int32_t ref_first;        // forward sweep, bf16 without the T5 table: reference point of every row = its maximum over the first tile instead of 0 (large logits: see attn_fwd64.h)
  int32_t lds_stage;        // 64-wide backward bodies: the register-resident operands arrive / the outputs leave through wave-private LDS images (set by the launcher when the LDS fits)
  float scale;
};

// workgroup index -> (batch, head, tile).  The grid covers the call's units x tiles (all B * H units, or a unit range).
#ifndef FAT5_CAUSAL_ORDER
#define FAT5_CAUSAL_ORDER 1  // 1: the kernels of a causal problem launch their longest workgroups first (decode_block: order 1 for row blocks -- forward, dQ --, 2 for key blocks -- dK/dV); 0: pair-major (rounds 1-4)
#endif
FAT5_DEV void decode_unit(const AttnArgs& a, int bid, int ntile, int& b, int& h, int& tile, int order = 0) {
  const uint32_t mg_tile = ntile == a.n_mblk ? a.mg_mblk : (ntile == a.n_nblk ? a.mg_nblk : 0u);
  if (a.batch_inner) {
    // A batch-broadcast dense bias tile (1, h, m-tile, n-tile) is read by all B batch elements: give the B workgroups of one
    // (head, tile) consecutive slots of ONE XCD, so the tile crosses the fabric once and is then served by that XCD's L2
    // (with the per-(b,h) mapping below each batch element sits on another XCD: B x the bias traffic, the dominant stream
    // of the dense mode).  K/V (or Q/dO) of one (b, h) are then read by several XCDs instead: 1/8 of the bias bytes at most.
    const int groups = a.H * ntile;
    int gid;
    if ((groups & 7) == 0) {
      const int xcd = bid & 7, idx = bid >> 3;
      b = idx % a.B;
      gid = (idx / a.B) * 8 + xcd;
    } else {
      b = bid % a.B;
      gid = bid / a.B;
    }
    if (order) {
      // causal: tile-major over the heads, longest tiles first -- consecutive groups (= the XCDs at any moment) work on the same tile level.  (The head-major deal
      // below sends tile (8 k + xcd) % ntile to an XCD: with ntile a power of two every XCD sees only one or two tile indices, i.e. lengths -- the XCDs with the
      // long ones decide the launch.)
      const int tq = gid / a.H;
      h = gid - tq * a.H;
      tile = order == 1 ? ntile - 1 - tq : tq;
      return;
    }
    h = gid / ntile;
    tile = gid - h * ntile;
    return;
  }
  int ui;
  decode_block(bid, a.unit_count > 0 ? a.unit_count : a.B * a.H, ntile, ui, tile, mg_tile, order);
  if (a.unit_count > 0) {
    const int u = a.unit_begin + ui;
    h = u / a.B;
    b = u - h * a.B;
  } else {
    b = fast_div(ui, a.H, a.mg_H);
    h = ui - b * a.H;
  }
}

// Buffer resource over rows [0, nrows) of one (b,h) slice: bytes past the last row's D elements are out of range
// (hardware returns 0).  Inputs are made provably wave-uniform so no waterfall loop is generated around the loads.
FAT5_DEV __amdgpu_buffer_rsrc_t make_rows_rsrc(const uint16_t* base, int64_t row_stride, int nrows, int D) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  const int64_t bytes = nrows > 0 ? ((int64_t)(nrows - 1) * row_stride + D) * 2 : 0;
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// 16-byte operand fragment of a row: columns 16 kk + 8 hi .. + 7 (zeros beyond the valid columns)
FAT5_DEV u32x4 load_frag16(const uint16_t* row, int kk, int hi, int dvalid) {
  if (16 * kk + 8 * hi >= dvalid) return u32x4{0u, 0u, 0u, 0u};
  return *reinterpret_cast<const u32x4*>(row + 16 * kk + 8 * hi);
}

// Row staging for row-major images only (one 16-byte chunk of one row per work item).
template <int D, int ROWS, int NT>
struct RowStage {
  static constexpr int C = D / 8;
  static constexpr int ITEMS = ROWS * C;
  static constexpr int PER = (ITEMS + NT - 1) / NT;
  u32x4 r[PER];
  FAT5_DEV void load(const uint16_t* base, int64_t row_stride, int row0, int limit, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = tid + NT * i;
      const int c = id % C, row = row0 + id / C;
      const bool in = (ITEMS % NT == 0) || (id < ITEMS);
      r[i] = gload16(base + (int64_t)row * row_stride + c * 8, in && row < limit);
    }
  }
  // rows >= limit read row limit-1 again (finite data; callers mask those rows' contributions exactly)
  FAT5_DEV void load_clamped(const uint16_t* base, int64_t row_stride, int row0, int limit, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = tid + NT * i;
      if ((ITEMS % NT != 0) && id >= ITEMS) continue;
      const int row = min(row0 + id / C, limit - 1);
      r[i] = *reinterpret_cast<const u32x4*>(base + (int64_t)row * row_stride + (id % C) * 8);
    }
  }
  // Buffer-descriptor path (the one the kernels use): voffset = this thread's constant byte offset inside a
  // tile, soffset = wave-uniform byte offset of the tile.  No per-load VALU address math, no temporaries that the
  // compiler could alias with MFMA operands, and rows past the descriptor's end read as ZERO in hardware.
  FAT5_DEV void load_buf(__amdgpu_buffer_rsrc_t rsrc, uint32_t tile_byte_off, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if ((ITEMS % NT != 0) && tid + NT * i >= ITEMS) continue;
      r[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[i] * 2u, tile_byte_off, 0));
    }
  }
  // unguarded variant for full tiles: `tile` = wave-uniform pointer to the tile's first row (SGPR base),
  // goff[i] = this thread's constant element offset inside a tile -> global_load with saddr + 32-bit voffset
  uint32_t goff[PER];
  int loff[PER];
  FAT5_DEV void init(int64_t row_stride, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = tid + NT * i;
      goff[i] = (uint32_t)((id / C) * row_stride + (id % C) * 8);
      loff[i] = rm_off<D>(id / C, id % C);
    }
  }
  FAT5_DEV void load_full(const uint16_t* tile, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if ((ITEMS % NT != 0) && tid + NT * i >= ITEMS) continue;
      r[i] = *reinterpret_cast<const u32x4*>(tile + goff[i]);
    }
  }
  FAT5_DEV void store_rm(char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if ((ITEMS % NT != 0) && tid + NT * i >= ITEMS) continue;
      *reinterpret_cast<u32x4*>(lds + loff[i]) = r[i];
    }
  }
};

// ------------------------------------------------------------------------------------------
// Direct global -> LDS tile staging (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write pass.
// The hardware writes lane L's 16 bytes at (wave-uniform LDS base) + 16*L, i.e. the image is lane-linear; the
// XOR swizzle of the row-major image is therefore applied to the SOURCE: the lane that fills slot c' of row r
// fetches chunk c' ^ swz(r) (readers look for chunk c at slot c ^ swz(r); same involution on both sides).
// Work item id = tid + NT*i -> row id / C, slot id % C; rows of consecutive i differ by NT/C (a multiple of 16,
// so the swizzle term is the same for every i and one VGPR addresses all pieces).  Rows past the descriptor's
// end arrive as zeros.  Completion: the issuing wave's vmcnt, then a barrier for the other waves' reads.
// ------------------------------------------------------------------------------------------
template <int D, int ROWS, int NT, bool SWZ = true>
struct DmaStage {
  static constexpr int C = D / 8;
  static constexpr int ITEMS = ROWS * C;
  static constexpr int PER = ITEMS >= NT ? ITEMS / NT : 1;  // (ITEMS < NT: the trailing waves have nothing to fetch)
  static constexpr int RP = NT / C;                       // rows covered by one piece of the whole workgroup
  static constexpr int NV = (RP % 16 == 0 || !SWZ) ? 1 : 16 / RP; // distinct swizzle phases among the pieces
  static_assert((ITEMS % NT == 0 || (NT % ITEMS == 0 && ITEMS % 64 == 0)) && NT % C == 0 && (16 % RP == 0 || RP % 16 == 0) &&
                    PER % NV == 0, "unsupported tile split");
  uint32_t voff[NV];    // this lane's byte offset inside a tile for pieces 0 .. NV-1
  uint32_t piece_step;  // bytes between the rows of pieces i and i + NV (wave-uniform)
  // dvalid < D: the source rows hold only dvalid columns -- the pieces beyond them get an out-of-range offset (bit 31: every descriptor
  // here is shorter than 2 GiB) and arrive as zeros, like rows past the end
  FAT5_DEV void init(int64_t row_stride, int tid, int dvalid = D) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int row = tid / C + RP * v, slot = tid % C;
      const int src = SWZ ? (slot ^ swz<D>(row)) : slot;  // (the image is lane-linear: the swizzle is applied to the source chunk)
      voff[v] = (uint32_t)(row * row_stride * 2 + (src << 4));
      if (8 * src >= dvalid) voff[v] = 0x80000000u;
    }
    piece_step = (uint32_t)(RP * NV * row_stride * 2);
  }
  // tile at byte offset tile_off of the descriptor -> LDS image at `img` (workgroup-uniform)
  FAT5_DEV void issue(__amdgpu_buffer_rsrc_t rsrc, uint32_t tile_off, char* img, int tid) const {
#if defined(__HIP_DEVICE_COMPILE__)  // (hipcc's host pass mis-instantiates templates that reach this builtin)
    typedef __attribute__((address_space(3))) void* lds_t;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if (ITEMS < NT && 64 * wave >= ITEMS) continue;  // wave-uniform
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_t)(uintptr_t)(uint32_t)(uintptr_t)(img + (NT * i + 64 * wave) * 16), 16,
                                               voff[i % NV], tile_off + piece_step * (i / NV), 0, 0);
    }
#endif
  }
  // the same pieces into registers (operands that pair with a DMA'd tile piece by piece)
  FAT5_DEV u32x4 load_piece(__amdgpu_buffer_rsrc_t rsrc, uint32_t tile_off, int i) const {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i % NV], tile_off + piece_step * (i / NV), 0));
  }
  // LDS byte offset of the 16 bytes this thread's piece i lands at
  FAT5_DEV static constexpr int own_off(int tid, int i) { return (tid + NT * i) * 16; }
};

// ------------------------------------------------------------------------------------------
// Dense-bias tile of a query-parallel body (forward, dQ): (32*NW query rows) x 64 keys, 16-bit, staged global -> LDS
// by DmaStage<64, rows, NT> (source-swizzled like a D = 64 image).  Lane (row, hi) needs, per 32-key block kb, the
// four 8-byte groups of keys 32*kb + 8*g + 4*hi .. +3: chunk 4*kb + g, half hi.  With slot = chunk ^ swz<64>(row) the
// XOR splits into a kb part (bit 2) and a g part (bits 0-1), so two base offsets and four group offsets per lane
// address everything with plain adds; the 32 rows of a wave spread over all banks (unswizzled they would share one).
// ------------------------------------------------------------------------------------------
struct BiasTileReader {
  int base[2];  // byte offset of (row, kb), half hi
  int goff[4];  // byte offset of group g
  FAT5_DEV void init(int row, int hi) {
    const int sw = swz<64>(row);
    base[0] = row * 128 + ((sw & 4) << 4) + 8 * hi;
    base[1] = row * 128 + ((4 ^ (sw & 4)) << 4) + 8 * hi;
#pragma unroll
    for (int g = 0; g < 4; ++g) goff[g] = (g ^ (sw & 3)) << 4;
  }
  // overwrite this lane's positions of block kb with four packed 8-byte groups (the dS tile reuses the bias tile)
  FAT5_DEV void store(char* tile, int kb, const u32x4 (&v)[2]) const {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x2 w2 = {v[g >> 1][2 * (g & 1)], v[g >> 1][2 * (g & 1) + 1]};
      *reinterpret_cast<u32x2*>(tile + base[kb] + goff[g]) = w2;
    }
  }
  template <bool BF16>
  FAT5_DEV void load(const char* tile, int kb, float (&bv)[16]) const {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u32x2 w = *reinterpret_cast<const u32x2*>(tile + base[kb] + goff[g]);
      w[0] = bias_clamp2<BF16>(w[0]);
      w[1] = bias_clamp2<BF16>(w[1]);
      bv[4 * g + 0] = cvt_lo<BF16>(w[0]);
      bv[4 * g + 1] = cvt_hi<BF16>(w[0]);
      bv[4 * g + 2] = cvt_lo<BF16>(w[1]);
      bv[4 * g + 3] = cvt_hi<BF16>(w[1]);
    }
  }
};

// Per-lane LDS byte offsets of the fragment reads, hoisted out of the tile loops.  With the swizzle of
// swz<D>() the XOR term of a fragment address depends on the lane only (not on the 32-row block / 16-row
// step), so a handful of VGPRs + immediate offsets address every read of a tile.
template <int D>
struct FragAddr {
  static constexpr int KK = D / 16, DB = D / 32;
  int rm[KK];      // ds_read_b128: row (lane&31) of a 32-row block, chunk 2kk+hi
  int tr[2][DB];   // ds_read_b64_tr_b16: rows 8*j2 + 4*hi + e, d-block db
  FAT5_DEV void init(int lane) {
    const int lq = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) rm[kk] = rm_off<D>(lq, 2 * kk + hi);
    const int i = lane & 15, e = i >> 2, c = i & 3, g = (lane >> 4) & 1;
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const int row = 8 * j2 + 4 * hi + e;
#pragma unroll
      for (int db = 0; db < DB; ++db)
        tr[j2][db] = row * (2 * D) + (((4 * db + 2 * g + (c >> 1)) ^ swz<D>(row)) << 4) + 8 * (c & 1);
    }
  }
};
// row-major fragment (A or B operand, contraction over the row's elements): 32-row block `blk`
template <int D>
FAT5_DEV u32x4 ld_rm(const char* img, const FragAddr<D>& fa, int blk, int kk) {
  return *reinterpret_cast<const u32x4*>(img + fa.rm[kk] + blk * (32 * 2 * D));
}
// transposed fragment (contraction over rows): rows 32*blk + 16*t + {8*j2 + 4*hi + (j&3)}, d-block db
template <int D>
FAT5_DEV u32x4 ld_tr(const char* img, const FragAddr<D>& fa, int blk, int t, int db) {
  typedef s16x4_t __attribute__((address_space(3))) * lds_ptr_t;
  const char* p0 = img + fa.tr[0][db] + (32 * blk + 16 * t) * (2 * D);
  const char* p1 = img + fa.tr[1][db] + (32 * blk + 16 * t) * (2 * D);
  const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(uintptr_t)(uint32_t)(uintptr_t)p0);
  const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(uintptr_t)(uint32_t)(uintptr_t)p1);
  const u32x2 a2 = __builtin_bit_cast(u32x2, a), b2 = __builtin_bit_cast(u32x2, b);
  u32x4 r = {a2[0], a2[1], b2[0], b2[1]};
  return r;
}

// pack 8 fp32 C-layout registers (r = 8t .. 8t+7) into one 16-bit operand fragment
template <bool BF16>
FAT5_DEV u32x4 pack8(const f32x16& x, int t) {
  u32x4 r;
  r[0] = pack2<BF16>(x[8 * t + 0], x[8 * t + 1]);
  r[1] = pack2<BF16>(x[8 * t + 2], x[8 * t + 3]);
  r[2] = pack2<BF16>(x[8 * t + 4], x[8 * t + 5]);
  r[3] = pack2<BF16>(x[8 * t + 6], x[8 * t + 7]);
  return r;
}

template <int D, int ROWS>
constexpr int rm_bytes() { return ROWS * 2 * D; }

}  // namespace fat5
