// Common device helpers for the gfx950 attention kernels (fwd, bwd_kv, bwd_q).
// CDNA4 only: 64-wide wavefronts, v_mfma_f32_32x32x16_{bf16,f16}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fat5.h"

namespace fat5 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define FAT5_DEV __device__ __forceinline__

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
template <bool BF16>
FAT5_DEV float cvt16(uint16_t h) {
  return cvt_lo<BF16>((uint32_t)h);
}
template <bool BF16>
FAT5_DEV uint16_t to16(float a) {
  return (uint16_t)(pack2<BF16>(a, 0.f) & 0xffffu);
}

FAT5_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
FAT5_DEV float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

FAT5_DEV float xchg32(float v) {  // value held by the partner lane (lane ^ 32)
  return __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------------------
// LDS images.
// Row-major [rows][D] 16-bit tile with 16-byte chunks XOR-swizzled so that a ds_read_b128 of
// "32 different rows, same chunk" (the MFMA A/B fragment read) is bank-conflict free.
//   bank row = 256 B = 16 chunks; a row holds C = D/8 chunks; rows per bank row = 16 / C.
// ------------------------------------------------------------------------------------------
template <int D>
FAT5_DEV constexpr int swz(int row) {
  constexpr int C = D / 8;
  if constexpr (C >= 16) return row & (C - 1) & 15;
  else return (row * C / 16) & (C - 1);
}
template <int D>
FAT5_DEV int rm_off(int row, int chunk) {  // byte offset of 16-B chunk `chunk` of row `row`
  return row * (2 * D) + ((chunk ^ swz<D>(row)) << 4);
}

// Transposed image [D][ROWS + 4] 16-bit (row stride 2*ROWS + 8 bytes): element (d, r).
// 8-byte reads of "32 different d, same 4 consecutive r" are conflict free (stride = 34 dwords
// for ROWS = 64; odd multiple of 2 dwords in general).
template <int ROWS>
FAT5_DEV constexpr int tr_stride() { return 2 * ROWS + 8; }

// global 16-byte load of 8 consecutive 16-bit elements; zero when !valid
FAT5_DEV u32x4 gload16(const uint16_t* p, bool valid) {
  u32x4 z = {0, 0, 0, 0};
  return valid ? *reinterpret_cast<const u32x4*>(p) : z;
}

struct TensorView {  // one (b,h) slice: row pointer arithmetic in elements
  const uint16_t* base;
  int64_t row_stride;
};

// ------------------------------------------------------------------------------------------
// XCD-aware work-item decode.  Workgroup `bid` lands on XCD bid % 8 (observed, speed only).
// All tiles of one (b,h) are given to one XCD so K/V (fwd, bwd_q) or Q/dO (bwd_kv) of that pair
// stay in that XCD's L2.  Bijective for every (nbh, ntile).
// ------------------------------------------------------------------------------------------
FAT5_DEV void decode_block(int bid, int nbh, int ntile, int& bh, int& tile) {
  const int total = nbh * ntile;
  const int nx = 8;
  if ((nbh % nx) == 0) {
    const int xcd = bid % nx;
    const int idx = bid / nx;           // sequence number inside this XCD
    const int per = nbh / nx;           // (b,h) pairs per XCD
    const int pair = idx / ntile;       // which of this XCD's pairs
    tile = idx % ntile;
    bh = pair * nx + xcd;               // pairs dealt round-robin to XCDs
    (void)per;
    (void)total;
  } else {
    bh = bid / ntile;
    tile = bid % ntile;
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
  int32_t n_mblk, n_nblk;   // tiles per (b,h) for the m-parallel / n-parallel kernels
  float scale;
};

// ------------------------------------------------------------------------------------------
// Tile staging: each work item = rows (2p, 2p+1) x one 16-byte chunk, loaded once from global
// into registers and written to LDS as a row-major swizzled image and/or a transposed image.
// ------------------------------------------------------------------------------------------
template <int D, int ROWS, int NT>
struct PairStage {
  static constexpr int C = D / 8;
  static constexpr int ITEMS = (ROWS / 2) * C;
  static constexpr int PER = (ITEMS + NT - 1) / NT;
  u32x4 r0[PER], r1[PER];

  // rows [row0, row0 + ROWS) of a (rows, D) tensor; rows >= limit read as zero
  FAT5_DEV void load(const uint16_t* base, int64_t row_stride, int row0, int limit, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = tid + NT * i;
      const int c = id % C, p = id / C;
      const int ra = row0 + 2 * p, rb = ra + 1;
      const bool in = (ITEMS % NT == 0) || (id < ITEMS);
      r0[i] = gload16(base + (int64_t)ra * row_stride + c * 8, in && ra < limit);
      r1[i] = gload16(base + (int64_t)rb * row_stride + c * 8, in && rb < limit);
    }
  }
  // row-major swizzled image [ROWS][D]
  FAT5_DEV void store_rm(char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = tid + NT * i;
      if ((ITEMS % NT != 0) && id >= ITEMS) continue;
      const int c = id % C, p = id / C;
      *reinterpret_cast<u32x4*>(lds + rm_off<D>(2 * p, c)) = r0[i];
      *reinterpret_cast<u32x4*>(lds + rm_off<D>(2 * p + 1, c)) = r1[i];
    }
  }
  // transposed image [D][ROWS + 4]: element (d, row)
  FAT5_DEV void store_tr(char* lds, int tid) const {
    constexpr int TRS = tr_stride<ROWS>();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = tid + NT * i;
      if ((ITEMS % NT != 0) && id >= ITEMS) continue;
      const int c = id % C, p = id / C;
      char* dst = lds + (8 * c) * TRS + 4 * p;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t a = r0[i][j >> 1], b = r1[i][j >> 1];
        const uint32_t wv = (j & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
        *reinterpret_cast<uint32_t*>(dst + j * TRS) = wv;
      }
    }
  }
};

// fragment reads -----------------------------------------------------------------------------
// row-major image: rows on lanes (lq), 16-bit k-slots 16*kk + 8*hi + j
template <int D>
FAT5_DEV u32x4 frag_rm(const char* lds, int row, int kk, int hi) {
  return *reinterpret_cast<const u32x4*>(lds + rm_off<D>(row, 2 * kk + hi));
}
// transposed image: rows (d) on lanes, k-slots j <-> source row (j&3) + 8*(j>>2) + base
template <int ROWS>
FAT5_DEV u32x4 frag_tr(const char* lds, int d, int base) {
  constexpr int TRS = tr_stride<ROWS>();
  const char* p = lds + d * TRS + 2 * base;
  const u32x2 a = *reinterpret_cast<const u32x2*>(p);
  const u32x2 b = *reinterpret_cast<const u32x2*>(p + 16);
  u32x4 r = {a[0], a[1], b[0], b[1]};
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
template <int D, int ROWS>
constexpr int tr_bytes() { return D * tr_stride<ROWS>(); }

}  // namespace fat5
