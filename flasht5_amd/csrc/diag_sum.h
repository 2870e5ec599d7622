// Per-diagonal sums of 32 x 32 blocks held in the MFMA accumulator layout, on the VALU with DPP row rotations -- no LDS round trip.
//
// Used by the dK/dV bodies for the gradient of the T5 bias generator: drpe1d[h][d] = sum over (q, k) with k - q = d of dS[q][k]
// (reference: the bias gradient is the dS tensor summed over the batch, src/model/ops/flash_attention_v2_bias.py:214-215; the
// Toeplitz form of the bias, src/utils/positional_encoding.py:100-101, turns that sum into sums along diagonals).
// Rounds 1-3 wrote every block into a skewed LDS tile (16 ds_write_b16), read it back transposed (8 reads) and summed the columns
// with four 16x16x32 MFMAs, strictly in order behind the step's softmax: ~1,150 cycles per block measured at one wave per SIMD,
// 2.4x the cost of a whole pipelined step without it.
//
// Layout (attn_common.h): lane L = 32 hi + 16 a + p holds key column n = 16 a + p of the block; register r holds query row
// rho = crow(r, hi) = 16 qh + ql0 + 4 hi with qh = r >> 3, ql0 = (r & 3) + 8 ((r >> 2) & 1).  With base = (first key of the
// block) - (first row of the step), element (r, L) lies on diagonal
//     d = base + 16 a + p - 16 qh - ql0 - 4 hi.
// Step 1, per element: rotate the register right by 16 - ql0 inside its row of 16 lanes (DPP row_ror, fused into the add): the
// value lands at p' = (p - ql0) mod 16, borrow beta = [p < ql0], so d = base + 16 (a - qh - beta) + p' - 4 hi.  Two accumulators
// per qh: U (everything) and B (the borrowed part, masked on the SOURCE lane with a constant 64-bit lane mask).
// Step 2, once per step: classes c = qh + beta: A0 = U0 - B0, A1 = B0 + U1 - B1, A2 = B1; lane L' of A_c holds diagonal
// hb(L') - 16 c with the lane's "home" diagonal hb = base + 16 a + p' - 4 hi.  The next step (32 rows further down) has
// base' = base - 32, i.e. hb' = hb - 32: A2 is the next step's home (carried in `nxt`), A1 belongs to the partner lane (a ^ 1):
// its current home where a = 1, its next home where a = 0 (one v_permlane16_swap).  After the step `cur` is complete: either all
// 64 lanes store their home diagonal (one array of sums per half-wave: diag_finish_halves), or the hi = 1 half (homes 4 lower) is
// folded into the hi = 0 half by a rotation by 4 over 32 lanes (v_permlane32_swap, row_ror:12, a row swap and a bank-masked move),
// the four values that fall off the low end join the carry, and lanes 0..31 hand diagonals base + n to the caller (diag_finish).
// Every diagonal of a run of consecutive steps is handed over exactly once; two more calls with an all-zero step end a run.
#pragma once
#include "attn_common.h"

namespace fat5 {

struct DiagStep {   // one 32 x 32 block, zeroed at the start of its step
  float u0, u1, b0, b1;
};
struct DiagCarry {  // one key block across consecutive steps, zeroed at the start of a run
  float cur, nxt;
};
FAT5_DEV void diag_step_zero(DiagStep& s) { s.u0 = s.u1 = s.b0 = s.b1 = 0.f; }
FAT5_DEV void diag_carry_zero(DiagCarry& c) { c.cur = c.nxt = 0.f; }

template <int CTRL, int BANK = 0xf>
FAT5_DEV float dpp_take(float old, float v) {  // lanes of the banks in BANK receive the permuted value, the others keep `old`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, BANK, false));
}
constexpr int kDppRowRor = 0x120;  // row_ror:n = 0x120 + n: lane i of a row of 16 receives lane (i - n) mod 16
__host__ __device__ constexpr int diag_ql0(int r) { return (r & 3) + 8 * ((r >> 2) & 1); }
// lanes whose position inside their row of 16 is below n (the lanes that borrow when rotated down by n)
__host__ __device__ constexpr uint64_t diag_borrow_mask(int n) { return ((1ull << n) - 1ull) * 0x0001000100010001ull; }

// element of register R (compiler-visible form: the general, unpipelined steps)
template <int R>
FAT5_DEV void diag_elem(DiagStep& s, const float x, const int p16) {
  constexpr int ql0 = diag_ql0(R);
  float& u = (R >> 3) ? s.u1 : s.u0;
  float& b = (R >> 3) ? s.b1 : s.b0;
  if constexpr (ql0 == 0) {
    u += x;
  } else {
    u += dpp_take<kDppRowRor + 16 - ql0>(0.f, x);
    b += dpp_take<kDppRowRor + 16 - ql0>(0.f, p16 < ql0 ? x : 0.f);
  }
}
// the same as single pinned instructions for the hand-placed gap streams (attn_bwd64.h).  No hazard padding exists around asm:
// a DPP operand must have been written at least two instructions earlier -- the callers keep a gap between producer and consumer.
template <int R>
FAT5_DEV void diag_elem_u(DiagStep& s, const float x) {
  constexpr int ql0 = diag_ql0(R);
  float& u = (R >> 3) ? s.u1 : s.u0;
  if constexpr (ql0 == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(u) : "v"(x));
  else asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:%2 row_mask:0xf bank_mask:0xf" : "+v"(u) : "v"(x), "n"(16 - ql0));
}
// borrowed part with a VGPR mask (1.0 on the lanes whose value crossed the row's end): one multiply-add (the v_cndmask form with an
// SGPR lane mask measures 7.7 issue cycles against 2.7, tools/mb_diag.hip)
template <int R>
FAT5_DEV void diag_elem_bm(DiagStep& s, const float x, const float m) {
  constexpr int ql0 = diag_ql0(R);
  float& b = (R >> 3) ? s.b1 : s.b0;
  if constexpr (ql0 != 0) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:%3 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(x), "v"(m), "n"(16 - ql0));
}
template <int R>
FAT5_DEV float diag_elem_mask(const float x) {  // the borrowed part of register R (zero elsewhere); nothing for ql0 = 0
  constexpr int ql0 = diag_ql0(R);
  float t = 0.f;
  if constexpr (ql0 != 0) {
    const uint64_t m = diag_borrow_mask(ql0);
    asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(t) : "v"(x), "s"(m));
  }
  return t;
}
template <int R>
FAT5_DEV void diag_elem_b(DiagStep& s, const float t) {
  constexpr int ql0 = diag_ql0(R);
  float& b = (R >> 3) ? s.b1 : s.b0;
  if constexpr (ql0 != 0) asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:%2 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(t), "n"(16 - ql0));
}

// End of a step, first half: folds the step's accumulators into the carry and returns the now complete sums of the lanes' home
// diagonals hb = base + (lane & 31) - 4 hi (base = first key of the block - first row of the step), all 64 lanes.
FAT5_DEV float diag_finish_halves(DiagCarry& c, const DiagStep& s, const int lane) {
  const bool a1 = (lane & 16) != 0;
  const float A0 = s.u0 - s.b0, A1 = s.b0 + (s.u1 - s.b1), A2 = s.b1;
  // partner rows: r[0] = {row0, row0, row2, row2}, r[1] = {row1, row1, row3, row3} of A1
  const uint32_t a1u = __float_as_uint(A1);
  const auto sw = __builtin_amdgcn_permlane16_swap(a1u, a1u, false, false);
  const float cur = c.cur + A0 + (a1 ? 0.f : __uint_as_float(sw[1]));
  c.cur = c.nxt + A2 + (a1 ? __uint_as_float(sw[0]) : 0.f);
  c.nxt = 0.f;
  return cur;
}
// End of a step with the hi = 1 half (homes 4 lower) folded into the hi = 0 half: returns, in lanes 0..31, the complete sums of the
// diagonals base + lane; lanes 32..63 return garbage.  (Callers with room for one array of sums per half-wave skip the fold and
// store diag_finish_halves' 64 values.)
FAT5_DEV float diag_finish(DiagCarry& c, const DiagStep& s, const int lane) {
  const bool a1 = (lane & 16) != 0;
  const float cur = diag_finish_halves(c, s, lane);
  // W[n] = cur[32 + (n + 4) mod 32]
  const uint32_t cu = __float_as_uint(cur);
  const auto hs = __builtin_amdgcn_permlane32_swap(cu, cu, false, false);
  const float Z = __uint_as_float(hs[1]);                      // lanes 0..31: cur of lane 32 + n
  const float t1 = dpp_take<kDppRowRor + 12>(0.f, Z);          // t1[16 a + i] = Z[16 a + (i + 4) mod 16]
  const uint32_t t1u = __float_as_uint(t1);
  const auto ts = __builtin_amdgcn_permlane16_swap(t1u, t1u, false, false);
  const float t2 = a1 ? __uint_as_float(ts[0]) : __uint_as_float(ts[1]);  // t1 of the partner row
  const float W = dpp_take<0xE4, 0x8>(t1, t2);                  // (quad_perm identity; bank 3 = positions 12..15 take the partner row's)
  const int n = lane & 31;
  const bool lo = lane < 32;
  c.cur += (lo && n >= 28) ? W : 0.f;  // the four values that fall off the low end belong to the next step's homes
  return cur + (n <= 27 ? W : 0.f);
}

}  // namespace fat5
