// Fused multi-tensor AdamWScale step for gfx950 (bandwidth-bound, no MFMA).
//
// Replaces the reference optimizer's per-tensor op sequence (src/utils/adamw_scaled.py:154-211, foreach form :213-281: ~14
// elementwise launches per parameter tensor -- ~3,000 launches for FAT5-base) by TWO launches over all tensors of a group:
//   adamw_sumsq_kernel   per 8192-element chunk: sum of squares of the parameters (fp32) -> one partial per chunk
//   adamw_update_kernel  per chunk: fixed-order sum of the tensor's partials -> rms(p) -> step size; then per element
//                        m, v update, denominator, (Kahan-compensated) parameter update, decoupled weight decay
// Arithmetic follows the reference op by op INCLUDING its intermediate roundings: with 16-bit parameters the reference keeps
// m, v, the Kahan compensation and every intermediate of an in-place op in the parameter dtype, so each `rnd()` below is one
// in-place torch op of adamw_scaled.py (fp32 opmath inside the op, one rounding at its end).
// Algorithmic bytes per element (e = 2 or 4): sumsq e; update 4e read + 3e write (+ 2e with Kahan).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fat5.h"
#include "attn_common.h"

namespace fat5 {

constexpr int kAdamChunk = 8192;  // elements per workgroup (256 threads x 32)

template <int DT> struct adt;
template <> struct adt<FAT5_F32> { typedef float type; };
template <> struct adt<FAT5_F16> { typedef _Float16 type; };
template <> struct adt<FAT5_BF16> { typedef __bf16 type; };

template <int DT>
FAT5_DEV float rnd(float x) {  // the rounding at the end of one in-place op on a tensor of this dtype
  if constexpr (DT == FAT5_F32) return x;
  else return (float)(typename adt<DT>::type)x;
}

FAT5_DEV float block_sum_256(float v, float* red) {  // fixed-order sum over 256 threads, result in every thread
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  const float r = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return r;
}

// binary search: tensor of chunk c (chunk_begin is non-decreasing, chunk_begin[n] = total chunks)
FAT5_DEV int tensor_of_chunk(const fat5_adamw_tensor* __restrict__ tab, int n, int c) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].chunk_begin <= c) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

// GRAD: sum of squares of the gradients instead (global-norm clipping, see fat5_adamw_grad_sumsq)
template <int DT, bool GRAD = false>
__global__ __launch_bounds__(256) void adamw_sumsq_kernel(const fat5_adamw_tensor* __restrict__ tab, int n, float* __restrict__ partial) {
  typedef typename adt<DT>::type T;
  __shared__ float red[4];
  const int c = blockIdx.x;
  const fat5_adamw_tensor t = tab[tensor_of_chunk(tab, n, c)];
  const int64_t e0 = (int64_t)(c - t.chunk_begin) * kAdamChunk;
  const int64_t cnt = min((int64_t)kAdamChunk, t.numel - e0);
  const T* p = reinterpret_cast<const T*>(GRAD ? t.g : t.p) + e0;
  float acc = 0.f;
  constexpr int V = 16 / sizeof(T);
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    const int64_t nv = cnt / V;
    for (int64_t i = threadIdx.x; i < nv; i += 256) {
      const uint4 w = reinterpret_cast<const uint4*>(p)[i];
      const T* x = reinterpret_cast<const T*>(&w);
#pragma unroll
      for (int j = 0; j < V; ++j) acc = fmaf((float)x[j], (float)x[j], acc);
    }
    for (int64_t i = nv * V + threadIdx.x; i < cnt; i += 256) acc = fmaf((float)p[i], (float)p[i], acc);
  } else {
    for (int64_t i = threadIdx.x; i < cnt; i += 256) acc = fmaf((float)p[i], (float)p[i], acc);
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[c] = s;
}

template <typename T, int N>
struct __attribute__((aligned(sizeof(T) * N))) avec { T x[N]; };

// DT: dtype of parameters, gradients and the Kahan compensation; SDT: dtype of m, v and of the denominator (the reference's
// `use_state_dtype`, adamw_scaled.py:101-103: fp16 / bf16 states beside parameters of another dtype; SDT == DT otherwise)
// plain_step: `correct_bias=False` (:177): the reference's step size is then `lr * max(1e-3, rms(p))` = a Python float times a
// 0-dim tensor of p's dtype -> ROUNDED to p's dtype (with bias correction a float32 tensor enters the product and it stays fp32);
// when rms < 1e-3 Python's max() returns the float and the product lr * 1e-3 is a double (`lr_small`, cast once)
template <int DT, int SDT, bool KAHAN>
__global__ __launch_bounds__(256) void adamw_update_kernel(const fat5_adamw_tensor* __restrict__ tab, int n,
                                                           const float* __restrict__ partial, float beta1, float beta2, float a1,
                                                           float a2, float wdf_, float eps, const float* __restrict__ grad_coef,
                                                           int plain_step, float lr_small_, const float* __restrict__ dev_scalars) {
  typedef typename adt<DT>::type T;
  typedef typename adt<SDT>::type S;
  __shared__ float red[4];
  const int c = blockIdx.x;
  const int ti = tensor_of_chunk(tab, n, c);
  // global-norm clipping folded in: every gradient enters as rnd(g * coef), the value `clip_grad_norm_`'s in-place g.mul_(coef)
  // leaves behind (torch.nn.utils.clip_grad_norm_, the reference's `max_grad_norm: 1.0`); the gradients themselves stay untouched
  const bool clip = grad_coef != nullptr;
  const float gcoef = clip ? *grad_coef : 1.f;
  const fat5_adamw_tensor t = tab[ti];
  // dev_scalars: the three values that change from step to step -- {step prefactor, -lr * weight_decay, lr * 1e-3} -- read from
  // device memory instead of the launch arguments / the table, so that a captured launch (HIP graph replay) follows the schedule
  const float prefactor = dev_scalars ? dev_scalars[0] : t.step_prefactor;
  const float wdf = dev_scalars ? dev_scalars[1] : wdf_;
  const float lr_small = dev_scalars ? dev_scalars[2] : lr_small_;
  // ---- rms(p) of the whole tensor from its chunk partials, fixed order; reference :69-70, :184 ----
  const int nc = (int)((t.numel + kAdamChunk - 1) / kAdamChunk);
  float acc = 0.f;
  for (int i = threadIdx.x; i < nc; i += 256) acc += partial[t.chunk_begin + i];
  const float sumsq = block_sum_256(acc, red);
  // p.norm(2): fp32 accumulation, result in p's dtype; "/ numel ** 0.5": one more op in p's dtype
  const float norm = rnd<DT>(sqrtf(sumsq));
  const float rms = rnd<DT>(norm / (float)sqrt((double)t.numel));
  float neg_step = -(prefactor * fmaxf(1e-3f, rms));  // float32 x p-dtype scalars promote to float32 (:184)
  if (plain_step) neg_step = rms > 1e-3f ? -rnd<DT>(prefactor * rms) : -lr_small;  // (max(1e-3, rms): the tensor only when it is larger)
  // (a1 = 1 - beta1, a2 = 1 - beta2, wdf = -lr * weight_decay: formed in double by the host like the reference's Python, cast once)

  const int64_t e0 = (int64_t)(c - t.chunk_begin) * kAdamChunk;
  const int64_t cnt = min((int64_t)kAdamChunk, t.numel - e0);
  T* p = reinterpret_cast<T*>(t.p) + e0;
  const T* g = reinterpret_cast<const T*>(t.g) + e0;
  S* m = reinterpret_cast<S*>(t.m) + e0;
  S* v = reinterpret_cast<S*>(t.v) + e0;
  T* k = KAHAN ? reinterpret_cast<T*>(t.k) + e0 : nullptr;

  auto one = [&](float pf, float gf, float mf, float vf, float kf, float& po, float& mo, float& vo, float& ko) {
    if (clip) gf = rnd<DT>(gf * gcoef);
    mf = rnd<SDT>(mf * beta1);                        // exp_avg.mul_(beta1)                         :173
    mf = rnd<SDT>(fmaf(a1, gf, mf));                  //        .add_(grad, alpha=1-beta1)
    vf = rnd<SDT>(vf * beta2);                        // exp_avg_sq.mul_(beta2)                      :174
    vf = rnd<SDT>(fmaf(a2 * gf, gf, vf));             //        .addcmul_(grad, grad, value=1-beta2)
    float den = rnd<SDT>(sqrtf(vf));                  // exp_avg_sq.sqrt()                           :175
    den = rnd<SDT>(den + eps);                        //        .add_(eps)
    const float upd = neg_step * (mf / den);          // value * (exp_avg / denom)
    if constexpr (KAHAN) {
      kf = rnd<DT>(kf + upd);                         // kahan_comp.addcdiv_(exp_avg, denom, value=-step_size)   :190
      const float old = pf;                           // grad.copy_(p)                                            :193
      pf = rnd<DT>(pf + kf);                          // p.add_(kahan_comp)                                       :194
      const float err = rnd<DT>(old - pf);            // grad.sub_(p)                                             :197
      kf = rnd<DT>(kf + err);                         // kahan_comp.add_(grad)                                    :198
    } else {
      pf = rnd<DT>(pf + upd);                         // p.addcdiv_(exp_avg, denom, value=-step_size)             :200
    }
    if (wdf != 0.f) pf = rnd<DT>(fmaf(wdf, pf, pf));  // p.add_(p, alpha=-lr*weight_decay)                :210
    po = pf; mo = mf; vo = vf; ko = kf;
  };

  constexpr int V = 16 / sizeof(T);  // elements per thread and trip: 16 bytes of p / g / k, V state elements (8, 16 or 32 bytes)
  typedef avec<T, V> PV;
  typedef avec<S, V> SV;
  const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | (KAHAN ? reinterpret_cast<uintptr_t>(k) : 0)) & 15) == 0 &&
                       ((reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & (sizeof(SV) - 1)) == 0;
  int64_t done = 0;
  if (aligned) {
    const int64_t nv = cnt / V;
    for (int64_t i = threadIdx.x; i < nv; i += 256) {
      PV wp = reinterpret_cast<PV*>(p)[i], wk;
      const PV wg = reinterpret_cast<const PV*>(g)[i];
      SV wm = reinterpret_cast<SV*>(m)[i], wv = reinterpret_cast<SV*>(v)[i];
      if constexpr (KAHAN) wk = reinterpret_cast<PV*>(k)[i];
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float po, mo, vo, ko;
        one((float)wp.x[j], (float)wg.x[j], (float)wm.x[j], (float)wv.x[j], KAHAN ? (float)wk.x[j] : 0.f, po, mo, vo, ko);
        wp.x[j] = (T)po; wm.x[j] = (S)mo; wv.x[j] = (S)vo;
        if constexpr (KAHAN) wk.x[j] = (T)ko;
      }
      reinterpret_cast<PV*>(p)[i] = wp;
      reinterpret_cast<SV*>(m)[i] = wm;
      reinterpret_cast<SV*>(v)[i] = wv;
      if constexpr (KAHAN) reinterpret_cast<PV*>(k)[i] = wk;
    }
    done = nv * V;
  }
  for (int64_t i = done + threadIdx.x; i < cnt; i += 256) {
    float po, mo, vo, ko;
    one((float)p[i], (float)g[i], (float)m[i], (float)v[i], KAHAN ? (float)k[i] : 0.f, po, mo, vo, ko);
    p[i] = (T)po; m[i] = (S)mo; v[i] = (S)vo;
    if constexpr (KAHAN) k[i] = (T)ko;
  }
}

}  // namespace fat5
