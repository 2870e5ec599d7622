// Bandwidth-bound row-wise ops for gfx950: T5 RMSNorm and cross-entropy (+ label smoothing, z-loss).
// No MFMA: coalesced 16-byte HBM accesses, wave-shuffle / LDS reductions, fp32 math.
//
// Replaces the reference Triton kernels
//   _rmsnorm_fwd_kernel / _rmsnorm_bwd_kernel        (src/model/ops/rms_norm.py:25-131)
//   cross_entropy_fwd_kernel / cross_entropy_bwd_kernel (src/model/ops/cross_entropy_loss.py:35-162)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "attn_common.h"

namespace fat5 {

// element traits: 16-byte vectors of VEC elements <-> fp32
template <int DT>
struct Elem;
template <>
struct Elem<FAT5_F32> {
  typedef float T;
  static constexpr int VEC = 4;
  static FAT5_DEV void load(const T* p, float (&f)[VEC]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  static FAT5_DEV void store(T* p, const float (&f)[VEC]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  }
  static FAT5_DEV float ld1(const T* p) { return *p; }
  static FAT5_DEV void st1(T* p, float f) { *p = f; }
};
template <int DT>
struct Elem16 {
  typedef uint16_t T;
  static constexpr bool BF = (DT == FAT5_BF16);
  static constexpr int VEC = 8;
  static FAT5_DEV void load(const T* p, float (&f)[VEC]) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[2 * j] = cvt_lo<BF>(v[j]);
      f[2 * j + 1] = cvt_hi<BF>(v[j]);
    }
  }
  static FAT5_DEV void store(T* p, const float (&f)[VEC]) {
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = pack2<BF>(f[2 * j], f[2 * j + 1]);
    *reinterpret_cast<u32x4*>(p) = v;
  }
  static FAT5_DEV float ld1(const T* p) { return cvt16<BF>(*p); }
  static FAT5_DEV void st1(T* p, float f) { *p = to16<BF>(f); }
};
template <>
struct Elem<FAT5_F16> : Elem16<FAT5_F16> {};
template <>
struct Elem<FAT5_BF16> : Elem16<FAT5_BF16> {};

// wave_sum: attn_common.h (DPP + permlane swaps)
FAT5_DEV float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// load VEC weights starting at column c (w may have a different dtype than x)
template <int WDT, int VEC>
FAT5_DEV void load_w(const void* w, int64_t c, float (&f)[VEC]) {
  typedef Elem<WDT> W;
  const typename W::T* wp = reinterpret_cast<const typename W::T*>(w) + c;
  if constexpr (W::VEC == VEC) {
    W::load(wp, f);
  } else if constexpr (W::VEC * 2 == VEC) {  // fp32 weights, 16-bit activations
    float a[W::VEC], b[W::VEC];
    W::load(wp, a);
    W::load(wp + W::VEC, b);
#pragma unroll
    for (int j = 0; j < W::VEC; ++j) { f[j] = a[j]; f[W::VEC + j] = b[j]; }
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) f[j] = W::ld1(wp + j);
  }
}

// ---------------------------------------------------------------------------------------------
// RMSNorm forward: one wave per row, 16-byte accesses; x is re-read from L1 in the second pass.
// ---------------------------------------------------------------------------------------------
template <int XDT, int WDT, bool VECOK>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const void* __restrict__ x_, const void* __restrict__ w_,
                                                          void* __restrict__ y_, float* __restrict__ rstd,
                                                          int64_t rows, int n, int64_t xs, int64_t ys, float eps) {
  typedef Elem<XDT> X;
  constexpr int VEC = X::VEC;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const typename X::T* x = reinterpret_cast<const typename X::T*>(x_) + row * xs;
  typename X::T* y = reinterpret_cast<typename X::T*>(y_) + row * ys;
  float ss = 0.f;
  if constexpr (VECOK) {
    for (int c = lane * VEC; c < n; c += 64 * VEC) {
      float f[VEC];
      X::load(x + c, f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) ss = fmaf(f[j], f[j], ss);
    }
  } else {
    for (int c = lane; c < n; c += 64) {
      const float f = X::ld1(x + c);
      ss = fmaf(f, f, ss);
    }
  }
  ss = wave_sum(ss);
  const float r = 1.0f / sqrtf(ss / (float)n + eps);  // rms_norm.py:49-50
  if (lane == 0) rstd[row] = r;
  if constexpr (VECOK) {
    for (int c = lane * VEC; c < n; c += 64 * VEC) {
      float f[VEC], wv[VEC];
      X::load(x + c, f);
      load_w<WDT, VEC>(w_, c, wv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] = f[j] * r * wv[j];  // x_hat * w (:59-60)
      X::store(y + c, f);
    }
  } else {
    typedef Elem<WDT> W;
    const typename W::T* w = reinterpret_cast<const typename W::T*>(w_);
    for (int c = lane; c < n; c += 64) X::st1(y + c, X::ld1(x + c) * r * W::ld1(w + c));
  }
}

// ---------------------------------------------------------------------------------------------
// Residual add + RMSNorm forward (the residual epilogue of a T5 sub-layer fused into the next pre-norm,
// reference modeling_flash_t5.py:159-164 / :304-318): h = x + r rounded to the activation dtype (exactly what the separate add
// writes), y = rmsnorm(h) from the rounded sum -- bit-identical to `h = x + r; y = fast_rms_layernorm(h)`, one pass over x and r
// instead of two kernels and five tensor passes.  One wave per row; h is re-read from cache in the second pass.
// ---------------------------------------------------------------------------------------------
template <int XDT, int WDT, bool VECOK>
__global__ __launch_bounds__(256) void add_rmsnorm_fwd_kernel(const void* __restrict__ x_, const void* __restrict__ r_,
                                                              const void* __restrict__ w_, void* __restrict__ h_,
                                                              void* __restrict__ y_, float* __restrict__ rstd, int64_t rows, int n,
                                                              int64_t xs, int64_t rs, int64_t hs, int64_t ys, float eps) {
  typedef Elem<XDT> X;
  constexpr int VEC = X::VEC;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const typename X::T* x = reinterpret_cast<const typename X::T*>(x_) + row * xs;
  const typename X::T* r = reinterpret_cast<const typename X::T*>(r_) + row * rs;
  typename X::T* h = reinterpret_cast<typename X::T*>(h_) + row * hs;
  typename X::T* y = reinterpret_cast<typename X::T*>(y_) + row * ys;
  float ss = 0.f;
  if constexpr (VECOK) {
    for (int c = lane * VEC; c < n; c += 64 * VEC) {
      float f[VEC], g[VEC];
      X::load(x + c, f);
      X::load(r + c, g);
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] += g[j];
      X::store(h + c, f);
      X::load(h + c, f);  // (the rounded sum: same thread, same address)
#pragma unroll
      for (int j = 0; j < VEC; ++j) ss = fmaf(f[j], f[j], ss);
    }
  } else {
    for (int c = lane; c < n; c += 64) {
      X::st1(h + c, X::ld1(x + c) + X::ld1(r + c));
      const float f = X::ld1(h + c);
      ss = fmaf(f, f, ss);
    }
  }
  ss = wave_sum(ss);
  const float rr = 1.0f / sqrtf(ss / (float)n + eps);
  if (lane == 0) rstd[row] = rr;
  if constexpr (VECOK) {
    for (int c = lane * VEC; c < n; c += 64 * VEC) {
      float f[VEC], wv[VEC];
      X::load(h + c, f);
      load_w<WDT, VEC>(w_, c, wv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] = f[j] * rr * wv[j];
      X::store(y + c, f);
    }
  } else {
    typedef Elem<WDT> W;
    const typename W::T* w = reinterpret_cast<const typename W::T*>(w_);
    for (int c = lane; c < n; c += 64) X::st1(y + c, X::ld1(h + c) * rr * W::ld1(w + c));
  }
}

// Register-resident variant (n <= NCH * 64 * VEC, 16-byte friendly rows): the rounded sum stays in registers between the two
// passes -- x and r are read once, h and y written once, nothing is re-read (108 -> 97 us at (65536, 1024) bf16, 5.6 TB/s of its four tensor passes).
template <int DT>
FAT5_DEV void round_to_dtype(float (&f)[Elem<DT>::VEC], u32x4& packed) {
  if constexpr (DT == FAT5_F32) {
    packed = u32x4{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
  } else {
    constexpr bool BF = (DT == FAT5_BF16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      packed[j] = pack2<BF>(f[2 * j], f[2 * j + 1]);
      f[2 * j] = cvt_lo<BF>(packed[j]);
      f[2 * j + 1] = cvt_hi<BF>(packed[j]);
    }
  }
}
template <int XDT, int WDT, int NCH>
__global__ __launch_bounds__(256) void add_rmsnorm_fwd_reg_kernel(const void* __restrict__ x_, const void* __restrict__ r_,
                                                                  const void* __restrict__ w_, void* __restrict__ h_,
                                                                  void* __restrict__ y_, float* __restrict__ rstd, int64_t rows, int n,
                                                                  int64_t xs, int64_t rs, int64_t hs, int64_t ys, float eps) {
  typedef Elem<XDT> X;
  constexpr int VEC = X::VEC;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const typename X::T* x = reinterpret_cast<const typename X::T*>(x_) + row * xs;
  const typename X::T* r = reinterpret_cast<const typename X::T*>(r_) + row * rs;
  typename X::T* h = reinterpret_cast<typename X::T*>(h_) + row * hs;
  typename X::T* y = reinterpret_cast<typename X::T*>(y_) + row * ys;
  float f[NCH][VEC], g[NCH][VEC];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * VEC;
    if (c < n) {
      X::load(x + c, f[i]);
      X::load(r + c, g[i]);
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * VEC;
    if (c < n) {
      u32x4 packed;
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[i][j] += g[i][j];
      round_to_dtype<XDT>(f[i], packed);
      *reinterpret_cast<u32x4*>(h + c) = packed;
#pragma unroll
      for (int j = 0; j < VEC; ++j) ss = fmaf(f[i][j], f[i][j], ss);
    }
  }
  ss = wave_sum(ss);
  const float rr = 1.0f / sqrtf(ss / (float)n + eps);
  if (lane == 0) rstd[row] = rr;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * VEC;
    if (c < n) {
      float wv[VEC];
      load_w<WDT, VEC>(w_, c, wv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[i][j] = f[i][j] * rr * wv[j];
      X::store(y + c, f[i]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward through the UNIT-weight norm xhat = x * rstd, for the backward of the pre-norm fused into a projection GEMM
// (fused_linear.py): gy = dL/dxhat ->  dx = (gy - xhat * mean(xhat * gy)) * rstd, and xhat itself (the A operand of the
// dout^T xhat weight-gradient GEMM) in the same pass.  Wave per row, everything of the row in registers.
// ---------------------------------------------------------------------------------------------
template <int XDT, int NCH>
__global__ __launch_bounds__(256) void rmsnorm_unit_bwd_kernel(const void* __restrict__ gy_, const void* __restrict__ x_,
                                                               const float* __restrict__ rstd, void* __restrict__ dx_,
                                                               void* __restrict__ xhat_, int64_t rows, int n, int64_t gys, int64_t xs,
                                                               int64_t dxs, int64_t xhs, const void* __restrict__ dres_, int64_t drs) {
  typedef Elem<XDT> X;
  constexpr int VEC = X::VEC;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const typename X::T* x = reinterpret_cast<const typename X::T*>(x_) + row * xs;
  const typename X::T* gy = reinterpret_cast<const typename X::T*>(gy_) + row * gys;
  typename X::T* dx = reinterpret_cast<typename X::T*>(dx_) + row * dxs;
  typename X::T* xh_out = reinterpret_cast<typename X::T*>(xhat_) + row * xhs;
  // dres: the gradient that reaches x along the residual connection (h + sublayer(norm(h))): added here instead of by a kernel of its own
  const typename X::T* dres = dres_ ? reinterpret_cast<const typename X::T*>(dres_) + row * drs : nullptr;
  const float r = rstd[row];
  float xh[NCH][VEC], gv[NCH][VEC];
  float c1 = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * VEC;
    if (c < n) {
      X::load(x + c, xh[i]);
      X::load(gy + c, gv[i]);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        xh[i][j] *= r;
        c1 = fmaf(xh[i][j], gv[i][j], c1);
      }
    }
  }
  c1 = wave_sum(c1) / (float)n;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * VEC;
    if (c < n) {
      float o[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = (gv[i][j] - xh[i][j] * c1) * r;
      if (dres) {
        float dr[VEC];
        X::load(dres + c, dr);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] += dr[j];
      }
      X::store(dx + c, o);
      X::store(xh_out + c, xh[i]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Gated activation of the T5 v1.1 feed-forward (reference FlashT5DenseGatedAct.forward, modeling_flash_t5.py:139-142:
// act(wi_0(x)) * wi_1(x) with act = GELU(approximate='tanh') or ReLU, :134): one pass over the two projections instead of an
// activation kernel + a multiply (forward) and an activation-backward kernel + two multiplies + a concatenation (backward).
// h0 / h1 may be the two halves of ONE (rows, 2F) projection output (fused_linear.py stacks wi_0 and wi_1), and dh0 / dh1 the two
// halves of its gradient: everything is addressed by row strides.  fp32 arithmetic, one rounding per output element.
//   gelu_tanh(x) = 0.5 x (1 + tanh(k (x + c x^3))),  k = sqrt(2 / pi),  c = 0.044715   (torch.nn.GELU(approximate='tanh'))
// ---------------------------------------------------------------------------------------------
// 0.5 (1 + tanh u) = sigmoid(2u) = 1 / (1 + e^(-2u)) =: s, and 1 - s = e^(-2u) s: no cancellation in either tail
// (1 - 2 / (1 + e^(2u)) loses the small e^(2u) against the 1 for u << 0: 1e-3 relative at u = -5).
template <int ACT>
FAT5_DEV void act_and_grad(float x, float& a, float& da) {
  if constexpr (ACT == FAT5_ACT_RELU) {
    a = fmaxf(x, 0.f);
    da = x > 0.f ? 1.f : 0.f;
  } else {
    constexpr float k = 0.7978845608028654f, c = 0.044715f;
    const float x2 = x * x;
    const float u = fmaxf(k * x * fmaf(c, x2, 1.f), -40.f);  // (e^80 is finite in fp32: below, s is 0 to every output precision anyway)
    const float e = __builtin_amdgcn_exp2f(u * -2.885390081777927f);  // e^(-2u)
    const float sg = __builtin_amdgcn_rcpf(1.f + e);
    a = x * sg;
    da = fmaf(x * sg * (e * sg), 2.f * k * fmaf(3.f * c, x2, 1.f), sg);
  }
}
// one thread = one 16-byte chunk; blockIdx.y = row
template <int DT, int ACT>
__global__ __launch_bounds__(256) void gated_act_fwd_kernel(const void* __restrict__ h0_, const void* __restrict__ h1_, void* __restrict__ out_,
                                                            int F, int64_t s0, int64_t s1, int64_t so) {
  typedef Elem<DT> X;
  constexpr int VEC = X::VEC;
  const int c = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (c >= F) return;
  const int64_t row = blockIdx.y;
  float a[VEC], b[VEC], o[VEC];
  X::load(reinterpret_cast<const typename X::T*>(h0_) + row * s0 + c, a);
  X::load(reinterpret_cast<const typename X::T*>(h1_) + row * s1 + c, b);
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float act, dact;
    act_and_grad<ACT>(a[j], act, dact);
    o[j] = act * b[j];
  }
  X::store(reinterpret_cast<typename X::T*>(out_) + row * so + c, o);
}
template <int DT, int ACT>
__global__ __launch_bounds__(256) void gated_act_bwd_kernel(const void* __restrict__ do_, const void* __restrict__ h0_, const void* __restrict__ h1_,
                                                            void* __restrict__ dh0_, void* __restrict__ dh1_, int F, int64_t sd, int64_t s0,
                                                            int64_t s1, int64_t sg0, int64_t sg1) {
  typedef Elem<DT> X;
  constexpr int VEC = X::VEC;
  const int c = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (c >= F) return;
  const int64_t row = blockIdx.y;
  float g[VEC], a[VEC], b[VEC], d0[VEC], d1[VEC];
  X::load(reinterpret_cast<const typename X::T*>(do_) + row * sd + c, g);
  X::load(reinterpret_cast<const typename X::T*>(h0_) + row * s0 + c, a);
  X::load(reinterpret_cast<const typename X::T*>(h1_) + row * s1 + c, b);
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float act, dact;
    act_and_grad<ACT>(a[j], act, dact);
    d0[j] = g[j] * b[j] * dact;
    d1[j] = g[j] * act;
  }
  X::store(reinterpret_cast<typename X::T*>(dh0_) + row * sg0 + c, d0);
  X::store(reinterpret_cast<typename X::T*>(dh1_) + row * sg1 + c, d1);
}

// ---------------------------------------------------------------------------------------------
// RMSNorm backward: persistent waves over strided rows; dw accumulated per lane in registers,
// reduced across the workgroup's waves through LDS, one fp32 partial row per workgroup.
//   NCH = max 16-byte chunks per lane (n <= NCH * 64 * VEC)
// ---------------------------------------------------------------------------------------------
template <int XDT, int WDT, int NCH>
__global__ __launch_bounds__(512) void rmsnorm_bwd_kernel(const void* __restrict__ dy_, const void* __restrict__ x_,
                                                          const void* __restrict__ w_, const float* __restrict__ rstd,
                                                          void* __restrict__ dx_, float* __restrict__ dw_part,
                                                          int64_t rows, int n, int64_t dys, int64_t xs, int64_t dxs,
                                                          const void* __restrict__ dres_ = nullptr, int64_t drs = 0) {
  // dres_ (optional): the gradient that reaches the normalised tensor through its OTHER consumer (the residual stream of the fused
  // add + norm): dx = round(dx_norm) + dres, rounded again -- what autograd's accumulation of the two branch gradients computes
  typedef Elem<XDT> X;
  constexpr int VEC = X::VEC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sdw = reinterpret_cast<float*>(smem);  // [n]
  const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int64_t wave_id = (int64_t)blockIdx.x * nwave + wv_;
  const int64_t nwaves = (int64_t)gridDim.x * nwave;
  float dw[NCH][VEC], wreg[NCH][VEC];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * VEC;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { dw[i][j] = 0.f; wreg[i][j] = 0.f; }
    if (c < n) load_w<WDT, VEC>(w_, c, wreg[i]);
  }
  const float inv_n = 1.0f / (float)n;
  for (int64_t row = wave_id; row < rows; row += nwaves) {
    const typename X::T* x = reinterpret_cast<const typename X::T*>(x_) + row * xs;
    const typename X::T* dy = reinterpret_cast<const typename X::T*>(dy_) + row * dys;
    typename X::T* dx = reinterpret_cast<typename X::T*>(dx_) + row * dxs;
    const float r = rstd[row];
    float xh[NCH][VEC], wdy[NCH][VEC];
    float c1 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (lane + 64 * i) * VEC;
      if (c < n) {
        float xf[VEC], dyf[VEC];
        X::load(x + c, xf);
        X::load(dy + c, dyf);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          xh[i][j] = xf[j] * r;                       // xhat (rms_norm.py:113)
          wdy[i][j] = wreg[i][j] * dyf[j];            // :117
          dw[i][j] = fmaf(dyf[j], xh[i][j], dw[i][j]);  // :119
          c1 = fmaf(xh[i][j], wdy[i][j], c1);
        }
      }
    }
    c1 = wave_sum(c1) * inv_n;  // :121
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (lane + 64 * i) * VEC;
      if (c < n) {
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = (wdy[i][j] - xh[i][j] * c1) * r;  // :122
        if (dres_) {  // (workgroup-uniform)
          float g[VEC];
          X::store(dx + c, o);
          X::load(dx + c, o);  // rounded like the separate kernel's output
          X::load(reinterpret_cast<const typename X::T*>(dres_) + row * drs + c, g);
#pragma unroll
          for (int j = 0; j < VEC; ++j) o[j] += g[j];
        }
        X::store(dx + c, o);
      }
    }
  }
  // workgroup reduction of dw in wave order (deterministic)
  for (int wsel = 0; wsel < nwave; ++wsel) {
    if (wv_ == wsel) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * VEC;
        if (c < n) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) sdw[c + j] = (wsel == 0 ? 0.f : sdw[c + j]) + dw[i][j];
        }
      }
    }
    __syncthreads();
  }
  float* out = dw_part + (int64_t)blockIdx.x * n;
  for (int c = threadIdx.x; c < n; c += blockDim.x) out[c] = sdw[c];
}

// Generic fallback (n or strides not 16-byte friendly): scalar accesses, dw accumulated with LDS atomics.
template <int XDT, int WDT>
__global__ __launch_bounds__(256) void rmsnorm_bwd_scalar_kernel(const void* __restrict__ dy_, const void* __restrict__ x_,
                                                                 const void* __restrict__ w_, const float* __restrict__ rstd,
                                                                 void* __restrict__ dx_, float* __restrict__ dw_part,
                                                                 int64_t rows, int n, int64_t dys, int64_t xs, int64_t dxs,
                                                                 const void* __restrict__ dres_ = nullptr, int64_t drs = 0) {
  typedef Elem<XDT> X;
  typedef Elem<WDT> W;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sdw = reinterpret_cast<float*>(smem);
  for (int c = threadIdx.x; c < n; c += blockDim.x) sdw[c] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, nwave = blockDim.x >> 6;
  const int64_t wave_id = (int64_t)blockIdx.x * nwave + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * nwave;
  const typename W::T* w = reinterpret_cast<const typename W::T*>(w_);
  const float inv_n = 1.0f / (float)n;
  for (int64_t row = wave_id; row < rows; row += nwaves) {
    const typename X::T* x = reinterpret_cast<const typename X::T*>(x_) + row * xs;
    const typename X::T* dy = reinterpret_cast<const typename X::T*>(dy_) + row * dys;
    typename X::T* dx = reinterpret_cast<typename X::T*>(dx_) + row * dxs;
    const float r = rstd[row];
    float c1 = 0.f;
    for (int c = lane; c < n; c += 64) c1 = fmaf(X::ld1(x + c) * r, W::ld1(w + c) * X::ld1(dy + c), c1);
    c1 = wave_sum(c1) * inv_n;
    for (int c = lane; c < n; c += 64) {
      const float xh = X::ld1(x + c) * r, dyf = X::ld1(dy + c);
      X::st1(dx + c, (W::ld1(w + c) * dyf - xh * c1) * r);
      if (dres_) X::st1(dx + c, X::ld1(dx + c) + X::ld1(reinterpret_cast<const typename X::T*>(dres_) + row * drs + c));
      atomicAdd(&sdw[c], dyf * xh);
    }
  }
  __syncthreads();
  float* out = dw_part + (int64_t)blockIdx.x * n;
  for (int c = threadIdx.x; c < n; c += blockDim.x) out[c] = sdw[c];
}

// dw[c] = sum_p part[p][c]  -> weight dtype (rms_norm.py:234).  Latency-bound: the partial rows were just written by other
// CUs, every read is a ~1 us round trip -- so all reads of a thread are in flight before its first add.  One workgroup =
// 64 columns x 16 row groups (1024 threads); thread (g, col) owns the partial rows p = g (mod 16), up to 16 of them per
// round, summed in a fixed order; the 16 groups are combined through LDS in a fixed tree.  Deterministic.
template <int WDT>
__global__ __launch_bounds__(1024) void rmsnorm_dw_reduce_kernel(const float* __restrict__ part, void* __restrict__ dw_,
                                                                 int nparts, int n) {
  typedef Elem<WDT> W;
  __shared__ float sacc[16][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (c < n) {
    for (int p0 = g; p0 < nparts; p0 += 256) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int pp = p0 + 16 * u;
        v[u] = (pp < nparts) ? part[(int64_t)pp * n + c] : 0.f;
      }
      const float t0 = (v[0] + v[1]) + (v[2] + v[3]), t1 = (v[4] + v[5]) + (v[6] + v[7]);
      const float t2 = (v[8] + v[9]) + (v[10] + v[11]), t3 = (v[12] + v[13]) + (v[14] + v[15]);
      acc += (t0 + t1) + (t2 + t3);
    }
  }
  sacc[g][lane] = acc;
  __syncthreads();
  if (g == 0 && c < n) {
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      t[k] = (sacc[4 * k][lane] + sacc[4 * k + 1][lane]) + (sacc[4 * k + 2][lane] + sacc[4 * k + 3][lane]);
    W::st1(reinterpret_cast<typename W::T*>(dw_) + c, (t[0] + t[1]) + (t[2] + t[3]));
  }
}

// ---------------------------------------------------------------------------------------------
// Cross-entropy forward: one 256-thread workgroup per row, online max / sum-exp per thread over
// 16-byte chunks, then a workgroup combine.
// ---------------------------------------------------------------------------------------------
template <int DT, bool VECOK>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const void* __restrict__ logits_, const int64_t* __restrict__ labels,
                                                     float* __restrict__ losses, float* __restrict__ z_losses,
                                                     float* __restrict__ lse_, int n_cols, int64_t row_stride,
                                                     float smoothing, float logit_scale, float lse_square_scale,
                                                     int64_t ignore_index, int use_precomputed_lse) {
  typedef Elem<DT> X;
  constexpr int VEC = X::VEC;
  __shared__ float sm[4], sl[4], ssum[4];
  const int64_t row = blockIdx.x;
  const typename X::T* x = reinterpret_cast<const typename X::T*>(logits_) + row * row_stride;
  const int tid = threadIdx.x, lane = tid & 63, wv_ = tid >> 6;
  const bool has_smooth = smoothing > 0.f;
  float lse;
  float sum_logits = 0.f;
  if (!use_precomputed_lse) {
    float m = -INFINITY, l = 0.f;
    if constexpr (VECOK) {
      for (int c = tid * VEC; c < n_cols; c += 256 * VEC) {
        float f[VEC];
        X::load(x + c, f);
        float cm = -INFINITY;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          f[j] *= logit_scale;
          cm = fmaxf(cm, f[j]);
          sum_logits += f[j];
        }
        const float mn = fmaxf(m, cm);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc += fast_exp2((f[j] - mn) * kLog2e);
        l = l * fast_exp2((m - mn) * kLog2e) + acc;
        m = mn;
      }
    } else {
      for (int c = tid; c < n_cols; c += 256) {
        const float f = X::ld1(x + c) * logit_scale;
        sum_logits += f;
        const float mn = fmaxf(m, f);
        l = l * fast_exp2((m - mn) * kLog2e) + fast_exp2((f - mn) * kLog2e);
        m = mn;
      }
    }
    // combine (m, l) across the wave, then across the 4 waves
    const float wm = wave_max(m);
    l = (m == -INFINITY) ? 0.f : l * fast_exp2((m - wm) * kLog2e);
    l = wave_sum(l);
    sum_logits = wave_sum(sum_logits);
    if (lane == 0) { sm[wv_] = wm; sl[wv_] = l; ssum[wv_] = sum_logits; }
    __syncthreads();
    const float gm = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float gl = 0.f;
    sum_logits = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gl += (sm[i] == -INFINITY) ? 0.f : sl[i] * fast_exp2((sm[i] - gm) * kLog2e);
      sum_logits += ssum[i];
    }
    lse = logf(gl) + gm;  // cross_entropy_loss.py:78
    if (tid == 0) lse_[row] = lse;
  } else {
    lse = lse_[row];
  }
  if (tid == 0) {
    const int64_t label = labels[row];
    float loss = 0.f, z = 0.f;
    if (label != ignore_index) {  // :83-106 (single rank: class_start_idx = 0, total_classes = n_cols)
      if (label >= 0 && label < n_cols) {
        const float ll = X::ld1(x + label) * logit_scale;
        loss = has_smooth ? (lse - smoothing * sum_logits / (float)n_cols - (1.f - smoothing) * ll) : (lse - ll);
      } else {
        loss = has_smooth ? smoothing * (lse - sum_logits / (float)n_cols) : 0.f;
      }
      z = lse_square_scale * lse * lse;
      loss += z;
    }
    losses[row] = loss;
    z_losses[row] = z;
  }
}

// ---------------------------------------------------------------------------------------------
// Cross-entropy backward: elementwise over (row, 2048-column block); may run in place.
// ---------------------------------------------------------------------------------------------
template <int DT, bool VECOK>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ dlosses, int64_t dloss_stride,
                                                     const void* logits_, const float* __restrict__ lse_,
                                                     const int64_t* __restrict__ labels, void* dlogits_, int n_cols,
                                                     int64_t row_stride, int64_t drow_stride, float smoothing,
                                                     float logit_scale, float lse_square_scale, int64_t ignore_index) {
  typedef Elem<DT> X;
  constexpr int VEC = X::VEC;
  const int64_t row = blockIdx.x;
  const typename X::T* x = reinterpret_cast<const typename X::T*>(logits_) + row * row_stride;
  typename X::T* dx = reinterpret_cast<typename X::T*>(dlogits_) + row * drow_stride;
  const int64_t label = labels[row];
  const float dloss = (label != ignore_index) ? dlosses[row * dloss_stride] : 0.f;  // :145-148
  const float lse = lse_[row];
  const float g = dloss * logit_scale;
  const float zf = 1.f + 2.f * lse_square_scale * lse;  // probs += 2*zs*lse*probs (:153-154)
  const bool has_smooth = smoothing > 0.f;
  const float sp = 1.f - smoothing, sn = smoothing / (float)n_cols;
  const float nl = -lse * kLog2e, ls2 = logit_scale * kLog2e;
  const int c0 = blockIdx.y * (256 * VEC);
  if constexpr (VECOK) {
    const int c = c0 + threadIdx.x * VEC;
    if (c < n_cols) {
      float f[VEC];
      X::load(x + c, f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float p = fast_exp2(fmaf(f[j], ls2, nl)) * zf;
        if (has_smooth) p = ((c + j == label) ? p - sp : p) - sn;  // :156-159
        else p = (c + j == label) ? p - 1.f : p;                   // :161
        f[j] = g * p;
      }
      X::store(dx + c, f);
    }
  } else {
    for (int j = 0; j < VEC; ++j) {
      const int c = c0 + j * 256 + threadIdx.x;
      if (c < n_cols) {
        float p = fast_exp2(fmaf(X::ld1(x + c), ls2, nl)) * zf;
        if (has_smooth) p = ((c == label) ? p - sp : p) - sn;
        else p = (c == label) ? p - 1.f : p;
        X::st1(dx + c, g * p);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Cross-entropy forward AND backward of a row in one launch (round 4; the mean-loss form of lm_head -> loss knows every row's
// upstream gradient before the forward runs, lm_head_cross_entropy.py): the two kernels above back to back inside one workgroup --
// same arithmetic in the same order, so lse / losses / dlogits are bit-identical to the two launches -- with the row read ONCE:
// HOLD keeps the row's logits in registers between the passes (up to 16 chunks of 256 x VEC columns: 32768 columns of a
// 16-bit dtype); otherwise the second pass re-reads the row (L2).  dlogits may alias logits (each thread rewrites its own chunks).
// Needs the vectorised layout (the C entry point falls back to the two launches otherwise).
// ---------------------------------------------------------------------------------------------
template <int DT, bool HOLD>
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const void* logits_, const int64_t* __restrict__ labels,
                                                         const float* __restrict__ dlosses, int64_t dloss_stride,
                                                         float* __restrict__ losses, float* __restrict__ z_losses, float* __restrict__ lse_,
                                                         void* dlogits_, int n_cols, int64_t row_stride, int64_t drow_stride,
                                                         float smoothing, float logit_scale, float lse_square_scale, int64_t ignore_index) {
  typedef Elem<DT> X;
  constexpr int VEC = X::VEC, NCH = 16;
  __shared__ float sm[4], sl[4], ssum[4];
  const int64_t row = blockIdx.x;
  const typename X::T* x = reinterpret_cast<const typename X::T*>(logits_) + row * row_stride;
  typename X::T* dx = reinterpret_cast<typename X::T*>(dlogits_) + row * drow_stride;
  const int tid = threadIdx.x, lane = tid & 63, wv_ = tid >> 6;
  const bool has_smooth = smoothing > 0.f;
  float fr[HOLD ? NCH : 1][VEC];
  float sum_logits = 0.f, m = -INFINITY, l = 0.f;
  auto chunk = [&](const float (&f)[VEC]) {  // (ce_fwd_kernel's per-chunk update; f stays unscaled for the second pass)
    float t[VEC];
    float cm = -INFINITY;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      t[j] = f[j] * logit_scale;
      cm = fmaxf(cm, t[j]);
      sum_logits += t[j];
    }
    const float mn = fmaxf(m, cm);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc += fast_exp2((t[j] - mn) * kLog2e);
    l = l * fast_exp2((m - mn) * kLog2e) + acc;
    m = mn;
  };
  if constexpr (HOLD) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int c = (k * 256 + tid) * VEC;
      if (c < n_cols) {
        X::load(x + c, fr[k]);
        chunk(fr[k]);
      }
    }
  } else {
    for (int c = tid * VEC; c < n_cols; c += 256 * VEC) {
      float f[VEC];
      X::load(x + c, f);
      chunk(f);
    }
  }
  const float wm = wave_max(m);
  l = (m == -INFINITY) ? 0.f : l * fast_exp2((m - wm) * kLog2e);
  l = wave_sum(l);
  sum_logits = wave_sum(sum_logits);
  if (lane == 0) { sm[wv_] = wm; sl[wv_] = l; ssum[wv_] = sum_logits; }
  __syncthreads();
  const float gm = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  float gl = 0.f;
  sum_logits = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    gl += (sm[i] == -INFINITY) ? 0.f : sl[i] * fast_exp2((sm[i] - gm) * kLog2e);
    sum_logits += ssum[i];
  }
  const float lse = logf(gl) + gm;
  const int64_t label = labels[row];
  if (tid == 0) {
    lse_[row] = lse;
    float loss = 0.f, z = 0.f;
    if (label != ignore_index) {
      if (label >= 0 && label < n_cols) {
        const float ll = X::ld1(x + label) * logit_scale;
        loss = has_smooth ? (lse - smoothing * sum_logits / (float)n_cols - (1.f - smoothing) * ll) : (lse - ll);
      } else {
        loss = has_smooth ? smoothing * (lse - sum_logits / (float)n_cols) : 0.f;
      }
      z = lse_square_scale * lse * lse;
      loss += z;
    }
    losses[row] = loss;
    z_losses[row] = z;
  }
  __syncthreads();  // (in place: the label's logit has been read before anything of the row is overwritten)
  // ---- backward (ce_bwd_kernel's arithmetic on the stored logits) ----
  const float dloss = (label != ignore_index) ? dlosses[row * dloss_stride] : 0.f;
  const float g = dloss * logit_scale;
  const float zf = 1.f + 2.f * lse_square_scale * lse;
  const float sp = 1.f - smoothing, sn = smoothing / (float)n_cols;
  const float nl = -lse * kLog2e, ls2 = logit_scale * kLog2e;
  auto grad = [&](float (&f)[VEC], const int c, const float mul) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float p = fast_exp2(fmaf(f[j], mul, nl)) * zf;
      if (has_smooth) p = ((c + j == label) ? p - sp : p) - sn;
      else p = (c + j == label) ? p - 1.f : p;
      f[j] = g * p;
    }
    X::store(dx + c, f);
  };
  if constexpr (HOLD) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int c = (k * 256 + tid) * VEC;
      if (c < n_cols) {
        grad(fr[k], c, ls2);
      }
    }
  } else {
    for (int c = tid * VEC; c < n_cols; c += 256 * VEC) {
      float f[VEC];
      X::load(x + c, f);
      grad(f, c, ls2);
    }
  }
}

}  // namespace fat5
