// Native host path of the attention operators: at::Tensor in / out, C++ autograd functions, straight onto the C ABI of
// libfat5.so (include/fat5.h).  Same behaviour as the ctypes path in flasht5_amd/flash_attention_v2_bias.py (which stays the
// reference implementation of the host logic and the path torch.compile traces); this one exists because an eager call through
// Python + ctypes costs ~100 us of host time per direction -- several times the S = 512 kernels -- most of it interpreter work
// and, in the backward, the hand-over between the autograd engine's device thread and the GIL.
//
// Mirrors reference src/model/ops/flash_attention_v2_bias.py:27-80 (fwd op), :91-217 (bwd op), :228-271 (autograd function).
// Built by flasht5_amd/build.py into flasht5_amd/lib/_fat5_torch.so (g++, links libfat5.so; no device code here).
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <atomic>
#include <map>
#include <mutex>
#include <tuple>

#include <hip/hip_runtime_api.h>

#include <vector>
#include "fat5.h"

namespace {

using at::Tensor;
using OptT = c10::optional<Tensor>;

bool kernel_ready(const Tensor& t) {  // last-dim stride 1, 16-byte aligned base, other strides multiples of 8 elements
  if (t.stride(-1) != 1 || (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) != 0) return false;
  for (int64_t i = 0; i + 1 < t.dim(); ++i)
    if (t.stride(i) % 8 != 0) return false;
  return true;
}
Tensor prep(const Tensor& t) { return kernel_ready(t) ? t : t.contiguous(); }
Tensor empty_like_ready(const Tensor& t) {
  Tensor e = at::empty_like(t);
  return kernel_ready(e) ? e : at::empty(t.sizes(), t.options());
}
// Gradients of PACKED projections: when the (B, H, S, D) inputs are the slices [:, :, i] of one (B, S, n, H, D) projection output
// (n = 3: q, k, v of a self-attention block from one GEMM; n = 2: k, v of a cross-attention block), their gradients are
// allocated as the same slices of ONE (B, S, n, H, D) buffer: the kernels write them in place (any strides with unit inner
// stride are fine for them) and the caller hands the buffer to the projection's backward GEMMs as it is -- no zero-filled
// full-size tensor per slice, no strided copies, no additions of the slices' gradients.
bool packed_slices(const std::vector<const Tensor*>& ts) {
  const Tensor& a = *ts[0];
  const int64_t n = (int64_t)ts.size(), H = a.size(1), S = a.size(2), D = a.size(3), row = n * H * D;
  if (a.stride(3) != 1 || a.stride(1) != D || a.stride(2) != row || a.stride(0) != S * row) return false;
  for (int64_t i = 1; i < n; ++i) {
    const Tensor& t = *ts[i];
    if (t.sizes() != a.sizes() || t.strides() != a.strides() || t.scalar_type() != a.scalar_type()) return false;
    if (reinterpret_cast<const char*>(t.data_ptr()) - reinterpret_cast<const char*>(a.data_ptr()) != i * H * D * (int64_t)a.element_size()) return false;
  }
  return true;
}
std::vector<Tensor> empty_packed_like(const std::vector<const Tensor*>& ts) {
  const Tensor& a = *ts[0];
  const int64_t n = (int64_t)ts.size(), H = a.size(1), S = a.size(2), D = a.size(3);
  Tensor base = at::empty({a.size(0), S, n, H, D}, a.options());
  std::vector<Tensor> out;
  for (int64_t i = 0; i < n; ++i) out.push_back(base.select(2, i).permute({0, 2, 1, 3}));
  return out;
}
void set3(int64_t (&d)[3], const Tensor& t) {
  d[0] = t.stride(0);
  d[1] = t.stride(1);
  d[2] = t.stride(2);
}
int dtype_code(const Tensor& t) {
  if (t.scalar_type() == at::kHalf) return FAT5_F16;
  if (t.scalar_type() == at::kBFloat16) return FAT5_BF16;
  TORCH_CHECK(false, "q, k, v must share dtype float16 or bfloat16");
}
void check_rc(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (fat5 status ", rc, "): ", fat5_last_error()); }

// fat5_attn_params.variant of every descriptor built here: 0 in production; tests / profilers set it through
// flasht5_amd._lib.set_variant (the same hook drives the ctypes path)
std::atomic<int> g_variant{0};

void base_params(fat5_attn_params& p, const Tensor& q, const Tensor& k, const Tensor& v, bool causal, double scale) {
  memset(&p, 0, sizeof(p));
  p.variant = g_variant.load(std::memory_order_relaxed);
  p.B = (int)q.size(0); p.H = (int)q.size(1); p.M = (int)q.size(2); p.N = (int)k.size(2); p.D = (int)q.size(3);
  p.dtype = dtype_code(q);
  p.causal = causal ? 1 : 0;
  p.sm_scale = (float)scale;
  p.q = q.data_ptr(); p.k = k.data_ptr(); p.v = v.data_ptr();
  set3(p.q_stride, q); set3(p.k_stride, k); set3(p.v_stride, v);
}
void set_bias(fat5_attn_params& p, const Tensor& bias) {  // broadcast (1|B, 1|H, M, N) via zero strides (reference :45-52)
  p.bias_mode = FAT5_BIAS_DENSE;
  p.bias = bias.data_ptr();
  p.bias_stride[0] = bias.size(0) == 1 ? 0 : bias.stride(0);
  p.bias_stride[1] = bias.size(1) == 1 ? 0 : bias.stride(1);
  p.bias_stride[2] = bias.stride(2);
}

// backward scratch: one growing buffer per (device, stream) -- eager launches on one stream are ordered, so consecutive calls
// share it.  While the stream is being CAPTURED into a HIP graph the buffer comes from the caching allocator instead (the
// graph's private pool): a cached buffer baked into a graph would be freed by a later, larger eager call and the replays would
// write into memory someone else owns.  The map is leaked on purpose (no Tensor destructors after HIP has shut down).
Tensor workspace(size_t nbytes, const Tensor& like, hipStream_t stream) {
  const int64_t n = (int64_t)std::max<size_t>(nbytes, 256);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
    return at::empty({n}, like.options().dtype(at::kByte));
  static std::mutex mu;
  static auto* cache = new std::map<std::pair<int, void*>, Tensor>();
  std::lock_guard<std::mutex> g(mu);
  auto key = std::make_pair((int)like.get_device(), (void*)stream);
  auto it = cache->find(key);
  if (it == cache->end() || it->second.numel() < n) {
    Tensor ws = at::empty({n}, like.options().dtype(at::kByte));
    (*cache)[key] = ws;
    return ws;
  }
  return it->second;
}

std::tuple<Tensor, Tensor> attn_fwd(const Tensor& q_, const Tensor& k_, const Tensor& v_, const OptT& bias_, const OptT& rpe1d,
                                    int64_t radius, bool causal, double scale) {
  Tensor q = prep(q_), k = prep(k_), v = prep(v_);
  Tensor o = empty_like_ready(q);  // reference :58
  Tensor L = at::empty({q.size(0), q.size(1), q.size(2)}, q.options().dtype(at::kFloat));  // reference :59
  fat5_attn_params p;
  base_params(p, q, k, v, causal, scale);
  p.o = o.data_ptr(); p.lse = (float*)L.data_ptr(); set3(p.o_stride, o);
  Tensor bias;
  if (bias_.has_value() && bias_->defined()) {
    bias = bias_->stride(-1) == 1 ? *bias_ : bias_->contiguous();
    set_bias(p, bias);
  } else if (rpe1d.has_value() && rpe1d->defined()) {
    p.bias_mode = FAT5_BIAS_RPE1D; p.rpe1d = (const float*)rpe1d->data_ptr(); p.rpe_radius = (int)radius;
  }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(q.device());  // (ROCm builds of torch call the device type "cuda")
  check_rc(fat5_attn_fwd(&p, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(q.get_device()).stream()), "fat5_attn_fwd");
  return {o, L};
}

std::tuple<Tensor, Tensor, Tensor, Tensor> attn_bwd(const Tensor& o_, const Tensor& do_, const Tensor& q_, const Tensor& k_,
                                                    const Tensor& v_, const OptT& bias_, const OptT& rpe1d, int64_t radius,
                                                    const Tensor& L, bool causal, double scale, bool need_dbias,
                                                    const OptT& bucket, int64_t num_buckets) {
  Tensor q = prep(q_), k = prep(k_), v = prep(v_), o = prep(o_), dout = prep(do_);
  Tensor dq, dk, dv;  // reference :140-141,:191
  if (q.sizes() == k.sizes() && packed_slices({&q, &k, &v})) {
    auto g = empty_packed_like({&q, &k, &v});
    dq = g[0]; dk = g[1]; dv = g[2];
  } else if (packed_slices({&k, &v})) {
    auto g = empty_packed_like({&k, &v});
    dq = empty_like_ready(q); dk = g[0]; dv = g[1];
  } else {
    dq = empty_like_ready(q); dk = empty_like_ready(k); dv = empty_like_ready(v);
  }
  fat5_attn_params p;
  base_params(p, q, k, v, causal, scale);
  p.o = o.data_ptr(); p.lse = (float*)L.data_ptr(); set3(p.o_stride, o);
  p.dout = dout.data_ptr(); p.dq = dq.data_ptr(); p.dk = dk.data_ptr(); p.dv = dv.data_ptr();
  set3(p.do_stride, dout); set3(p.dq_stride, dq); set3(p.dk_stride, dk); set3(p.dv_stride, dv);
  Tensor bias, dbias;
  if (bias_.has_value() && bias_->defined()) {
    bias = bias_->stride(-1) == 1 ? *bias_ : bias_->contiguous();
    set_bias(p, bias);
    if (need_dbias) {
      dbias = at::empty(bias.sizes(), bias.options());  // shape / dtype of bias (:149,:224)
      p.dbias = dbias.data_ptr(); p.dbias_batch = (int)bias.size(0); p.dbias_heads = (int)bias.size(1);
    }
  } else if (rpe1d.has_value() && rpe1d->defined()) {
    p.bias_mode = FAT5_BIAS_RPE1D; p.rpe1d = (const float*)rpe1d->data_ptr(); p.rpe_radius = (int)radius;
    if (need_dbias) {
      if (bucket.has_value() && bucket->defined()) {  // table gradient straight from the reduction launch
        dbias = at::empty({num_buckets, q.size(1)}, q.options().dtype(at::kFloat));
        p.rpe_bucket = (const int32_t*)bucket->data_ptr(); p.drpe_table = (float*)dbias.data_ptr(); p.rpe_num_buckets = (int)num_buckets;
      } else {
        dbias = at::empty_like(*rpe1d);
        p.drpe1d = (float*)dbias.data_ptr();
      }
    }
  }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(q.device());  // (ROCm builds of torch call the device type "cuda")
  hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(q.get_device()).stream();
  Tensor ws = workspace(fat5_attn_bwd_workspace_bytes(&p), q, stream);
  p.workspace = ws.data_ptr(); p.workspace_bytes = (size_t)ws.numel();
  check_rc(fat5_attn_bwd(&p, stream), "fat5_attn_bwd");
  return {dq, dk, dv, dbias};
}

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// FlashAttentionAdditiveBias (reference :228-271): o = attention(q, k, v, bias); gradients for q, k, v, bias
struct BiasFn : public torch::autograd::Function<BiasFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& q, const Tensor& k, const Tensor& v, const OptT& bias, bool causal, double scale) {
    auto [o, L] = attn_fwd(q, k, v, bias, c10::nullopt, 0, causal, scale);
    const bool has_bias = bias.has_value() && bias->defined();
    ctx->save_for_backward({q, k, v, has_bias ? *bias : Tensor(), o, L});
    ctx->saved_data["causal"] = causal;
    ctx->saved_data["scale"] = scale;
    return o;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto s = ctx->get_saved_variables();
    const bool has_bias = s[3].defined();
    auto [dq, dk, dv, ds] = attn_bwd(s[4], grads[0], s[0], s[1], s[2], has_bias ? OptT(s[3]) : OptT(), c10::nullopt, 0, s[5],
                                     ctx->saved_data["causal"].toBool(), ctx->saved_data["scale"].toDouble(), has_bias,
                                     c10::nullopt, 0);
    return {dq, dk, dv, has_bias ? ds : Tensor(), Tensor(), Tensor()};
  }
};

// (num_buckets, H) table -> (H, 2R+1) fp32 generator, one launch.  Built on EVERY forward call and never cached: whatever
// updates the table in place without touching its autograd version counter (this package's fused AdamWScale writes parameters
// through raw pointers; `p.data = ...` swaps) is seen by the next forward.
Tensor rpe1d_of(const Tensor& table_, const Tensor& bucket, int64_t radius, int64_t num_buckets) {
  Tensor table = table_.detach();
  const auto st = table.scalar_type();
  if (!(st == at::kFloat || st == at::kHalf || st == at::kBFloat16)) table = table.to(at::kFloat);
  table = table.contiguous();
  const int64_t H = table.size(1);
  Tensor r1 = at::empty({H, 2 * radius + 1}, table.options().dtype(at::kFloat));
  c10::hip::HIPGuardMasqueradingAsCUDA guard(table.device());
  check_rc(fat5_rpe1d_from_table(table.data_ptr(), st == at::kHalf ? FAT5_F16 : (st == at::kBFloat16 ? FAT5_BF16 : FAT5_F32),
                                 (const int32_t*)bucket.data_ptr(), (float*)r1.data_ptr(), (int)H, (int)radius, (int)num_buckets,
                                 c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(table.get_device()).stream()),
           "fat5_rpe1d_from_table");
  return r1;
}

// RPE mode on the (num_buckets, H) table: the kernels read the (H, 2R+1) generator built above; the table gradient comes
// scattered into buckets from the reduction launch (`bucket`: (2R+1,) int32)
struct RpeTableFn : public torch::autograd::Function<RpeTableFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& table,
                        const Tensor& bucket, int64_t radius, int64_t num_buckets, bool causal, double scale) {
    TORCH_CHECK(bucket.scalar_type() == at::kInt && bucket.is_contiguous() && bucket.numel() == 2 * radius + 1 && bucket.device() == q.device(),
                "rpe bucket index must be a contiguous int32 (2R+1,) vector on q's device");
    TORCH_CHECK(table.dim() == 2 && table.size(0) == num_buckets && table.size(1) == q.size(1) && table.device() == q.device(),
                "rpe_table must be (num_buckets, n_heads) on q's device");
    Tensor rpe1d = rpe1d_of(table, bucket, radius, num_buckets);
    auto [o, L] = attn_fwd(q, k, v, c10::nullopt, rpe1d, radius, causal, scale);
    ctx->save_for_backward({q, k, v, o, L, rpe1d, bucket});
    ctx->saved_data["causal"] = causal;
    ctx->saved_data["scale"] = scale;
    ctx->saved_data["radius"] = radius;
    ctx->saved_data["num_buckets"] = num_buckets;
    ctx->saved_data["tdtype"] = (int64_t)table.scalar_type();
    return o;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto s = ctx->get_saved_variables();
    const bool need = ctx->needs_input_grad(3);
    auto [dq, dk, dv, dt] = attn_bwd(s[3], grads[0], s[0], s[1], s[2], c10::nullopt, s[5], ctx->saved_data["radius"].toInt(), s[4],
                                     ctx->saved_data["causal"].toBool(), ctx->saved_data["scale"].toDouble(), need, s[6],
                                     ctx->saved_data["num_buckets"].toInt());
    if (need) dt = dt.to((at::ScalarType)ctx->saved_data["tdtype"].toInt());
    return {dq, dk, dv, need ? dt : Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// RPE mode on the 1-D generator itself (differentiable in rpe1d; `r1` = its detached fp32 contiguous form)
struct Rpe1dFn : public torch::autograd::Function<Rpe1dFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& rpe1d, const Tensor& r1,
                        int64_t radius, bool causal, double scale) {
    auto [o, L] = attn_fwd(q, k, v, c10::nullopt, r1, radius, causal, scale);
    ctx->save_for_backward({q, k, v, o, L, r1});
    ctx->saved_data["causal"] = causal;
    ctx->saved_data["scale"] = scale;
    ctx->saved_data["radius"] = radius;
    ctx->saved_data["rdtype"] = (int64_t)rpe1d.scalar_type();
    return o;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto s = ctx->get_saved_variables();
    const bool need = ctx->needs_input_grad(3);
    auto [dq, dk, dv, d1] = attn_bwd(s[3], grads[0], s[0], s[1], s[2], c10::nullopt, s[5], ctx->saved_data["radius"].toInt(), s[4],
                                     ctx->saved_data["causal"].toBool(), ctx->saved_data["scale"].toDouble(), need, c10::nullopt, 0);
    if (need) d1 = d1.to((at::ScalarType)ctx->saved_data["rdtype"].toInt());
    return {dq, dk, dv, need ? d1 : Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// ------------------------------------------------------------------------------------------------
// RMSNorm (reference src/model/ops/rms_norm.py:250-287) and the residual add fused into it: same host logic as
// flasht5_amd/rms_norm.py.  124 norm launches per FAT5-base step: their Python + dispatcher time is what this saves.
// ------------------------------------------------------------------------------------------------
int any_dtype_code(const Tensor& t) {
  if (t.scalar_type() == at::kFloat) return FAT5_F32;
  if (t.scalar_type() == at::kHalf) return FAT5_F16;
  if (t.scalar_type() == at::kBFloat16) return FAT5_BF16;
  TORCH_CHECK(false, "unsupported dtype (float32, float16, bfloat16)");
}
bool rows_ok(const Tensor& t) { return t.stride(-1) == 1 && (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) == 0; }
Tensor rows2d(const Tensor& t) {
  Tensor r = t.reshape({-1, t.size(-1)});
  return rows_ok(r) ? r : r.contiguous();
}

std::tuple<Tensor, Tensor> rms_fwd(const Tensor& X, const Tensor& W_, double eps) {
  const int64_t M = X.size(0), N = X.size(1);
  Tensor W = W_.contiguous();
  Tensor Y = at::empty({M, N}, X.options()), rstd = at::empty({M}, X.options().dtype(at::kFloat));
  if (M == 0) return {Y, rstd};
  c10::hip::HIPGuardMasqueradingAsCUDA guard(X.device());
  check_rc(fat5_rmsnorm_fwd(X.data_ptr(), W.data_ptr(), Y.data_ptr(), (float*)rstd.data_ptr(), M, N, X.stride(0), Y.stride(0), (float)eps,
                            any_dtype_code(X), any_dtype_code(W), c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(X.get_device()).stream()),
           "fat5_rmsnorm_fwd");
  return {Y, rstd};
}
std::tuple<Tensor, Tensor, Tensor> add_rms_fwd(const Tensor& X, const Tensor& R, const Tensor& W_, double eps) {
  const int64_t M = X.size(0), N = X.size(1);
  Tensor W = W_.contiguous();
  Tensor H = at::empty({M, N}, X.options()), Y = at::empty({M, N}, X.options()), rstd = at::empty({M}, X.options().dtype(at::kFloat));
  if (M == 0) return {H, Y, rstd};
  c10::hip::HIPGuardMasqueradingAsCUDA guard(X.device());
  check_rc(fat5_add_rmsnorm_fwd(X.data_ptr(), R.data_ptr(), W.data_ptr(), H.data_ptr(), Y.data_ptr(), (float*)rstd.data_ptr(), M, N, X.stride(0),
                                R.stride(0), H.stride(0), Y.stride(0), (float)eps, any_dtype_code(X), any_dtype_code(W),
                                c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(X.get_device()).stream()),
           "fat5_add_rmsnorm_fwd");
  return {H, Y, rstd};
}
// dres: undefined -> plain RMSNorm backward
std::tuple<Tensor, Tensor> rms_bwd(const Tensor& dy_, const Tensor& x, const Tensor& W_, const Tensor& rstd, const Tensor& dres_) {
  const int64_t M = x.size(0), N = x.size(1);
  Tensor dy = dy_.scalar_type() == x.scalar_type() ? dy_ : dy_.to(x.scalar_type());
  if (!rows_ok(dy)) dy = dy.contiguous();
  Tensor dres = dres_;
  if (dres.defined()) {
    if (dres.scalar_type() != x.scalar_type()) dres = dres.to(x.scalar_type());
    if (!rows_ok(dres)) dres = dres.contiguous();
  }
  Tensor W = W_.contiguous();
  Tensor dx = at::empty({M, N}, x.options()), dw = at::empty({N}, W.options());
  if (M == 0) return {dx, dw.zero_()};
  const size_t nbytes = fat5_rmsnorm_bwd_workspace_bytes(M, N);
  Tensor ws = at::empty({(int64_t)std::max<size_t>(nbytes, 16)}, x.options().dtype(at::kByte));
  c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  check_rc(fat5_add_rmsnorm_bwd(dy.data_ptr(), x.data_ptr(), W.data_ptr(), (const float*)rstd.data_ptr(), dres.defined() ? dres.data_ptr() : nullptr,
                                dx.data_ptr(), dw.data_ptr(), M, N, dy.stride(0), x.stride(0), dres.defined() ? dres.stride(0) : 0, dx.stride(0),
                                any_dtype_code(x), any_dtype_code(W), ws.data_ptr(), (size_t)ws.numel(),
                                c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(x.get_device()).stream()),
           "fat5_rmsnorm_bwd");
  return {dx, dw};
}

struct RmsNormFn : public torch::autograd::Function<RmsNormFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& X, const Tensor& W, double eps) {
    Tensor x2 = rows2d(X);
    auto [y, rstd] = rms_fwd(x2, W, eps);
    ctx->save_for_backward({x2, W, rstd});  // y is recomputed in backward, like the reference (:261-262)
    ctx->saved_data["shape"] = X.sizes().vec();
    return y.reshape(X.sizes());
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto s = ctx->get_saved_variables();
    auto [dx, dw] = rms_bwd(grads[0].reshape({-1, s[0].size(1)}), s[0], s[1], s[2], Tensor());
    return {dx.reshape(ctx->saved_data["shape"].toIntVector()), dw, Tensor()};
  }
};
struct AddRmsNormFn : public torch::autograd::Function<AddRmsNormFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& X, const Tensor& R, const Tensor& W, double eps) {
    auto [h, y, rstd] = add_rms_fwd(rows2d(X), rows2d(R), W, eps);
    ctx->save_for_backward({h, W, rstd});
    ctx->saved_data["shape"] = X.sizes().vec();
    return {h.reshape(X.sizes()), y.reshape(X.sizes())};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto s = ctx->get_saved_variables();
    const auto shape = ctx->saved_data["shape"].toIntVector();
    const Tensor& dH = grads[0];
    const Tensor& dY = grads[1];
    if (!dY.defined()) return {dH, dH, Tensor(), Tensor()};  // only the residual stream is used downstream
    const int64_t n = s[0].size(1);
    auto [dx, dw] = rms_bwd(dY.reshape({-1, n}), s[0], s[1], s[2], dH.defined() ? dH.reshape({-1, n}) : Tensor());
    Tensor d = dx.reshape(shape);
    return {d, d, dw, Tensor()};
  }
};
Tensor rmsnorm_apply(const Tensor& X, const Tensor& W, double eps) { return RmsNormFn::apply(X, W, eps); }
std::tuple<Tensor, Tensor> add_rmsnorm_apply(const Tensor& X, const Tensor& R, const Tensor& W, double eps) {
  auto r = AddRmsNormFn::apply(X, R, W, eps);
  return {r[0], r[1]};
}

Tensor bias_apply(const Tensor& q, const Tensor& k, const Tensor& v, const OptT& bias, bool causal, double scale) {
  return BiasFn::apply(q, k, v, bias, causal, scale);
}
Tensor rpe_table_apply(const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& table, const Tensor& bucket, int64_t radius,
                       int64_t num_buckets, bool causal, double scale) {
  return RpeTableFn::apply(q, k, v, table, bucket, radius, num_buckets, causal, scale);
}
Tensor rpe1d_apply(const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& rpe1d, const Tensor& r1, int64_t radius, bool causal,
                   double scale) {
  return Rpe1dFn::apply(q, k, v, rpe1d, r1, radius, causal, scale);
}

// ------------------------------------------------------------------------------------------------
// The fused projections and the passes around them (flasht5_amd/fused_linear.py, gated_act.py, attention_module.unpack_heads) as
// C++ autograd functions: same host logic as the Python classes, whose per-call cost -- above all in the BACKWARD, which the
// engine runs on its device thread and which has to take the GIL for every Python-defined node -- was what kept the FAT5-base
// step host-bound in eager mode (22-28 ms of enqueue time around 15 ms of kernels).
// ------------------------------------------------------------------------------------------------
hipStream_t cur_stream(const Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream(); }
bool lin_ready(const Tensor& t) { return t.stride(-1) == 1 && (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) == 0 && t.stride(0) % 8 == 0; }
Tensor lin_rows(const Tensor& t) {
  Tensor r = t.reshape({-1, t.size(-1)});
  return lin_ready(r) ? r : r.contiguous();
}
int lin_dtype(const Tensor& t) {
  TORCH_CHECK(t.scalar_type() == at::kHalf || t.scalar_type() == at::kBFloat16, "fused linear: float16 or bfloat16");
  return t.scalar_type() == at::kHalf ? FAT5_F16 : FAT5_BF16;
}
// [w0; w1; w2] diag(g) in one launch (fat5_fold_weights); g undefined: the plain stack
Tensor fold_weights(const std::vector<Tensor>& ws_, const Tensor& g_) {
  TORCH_CHECK(ws_.size() >= 1 && ws_.size() <= 3, "fold_weights: one to three weights");
  std::vector<Tensor> ws;
  for (const Tensor& w : ws_) ws.push_back(lin_ready(w) ? w : w.contiguous());
  const int64_t K = ws[0].size(1);
  int64_t n[3] = {0, 0, 0}, ld[3] = {0, 0, 0}, ntot = 0;
  const void* ptr[3] = {nullptr, nullptr, nullptr};
  for (size_t i = 0; i < ws.size(); ++i) { n[i] = ws[i].size(0); ld[i] = ws[i].stride(0); ptr[i] = ws[i].data_ptr(); ntot += n[i]; }
  Tensor out = at::empty({ntot, K}, ws[0].options());
  Tensor g = g_.defined() ? g_.to(ws[0].scalar_type()).contiguous() : Tensor();
  c10::hip::HIPGuardMasqueradingAsCUDA guard(out.device());
  check_rc(fat5_fold_weights(ptr[0], ptr[1], ptr[2], n[0], n[1], n[2], ld[0], ld[1], ld[2], g.defined() ? g.data_ptr() : nullptr, out.data_ptr(), K,
                             lin_dtype(out), cur_stream(out)), "fat5_fold_weights");
  return out;
}
// rmsnorm_linear (fused_linear.py::RMSNormLinear): inputs x, norm_weight, w0, w1?, w2? (undefined = absent)
struct RmsNormLinearFn : public torch::autograd::Function<RmsNormLinearFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& x, const Tensor& norm_weight, double eps, bool with_residual, const Tensor& w0,
                               const OptT& w1_, const OptT& w2_) {
    const Tensor w1 = w1_.has_value() ? *w1_ : Tensor(), w2 = w2_.has_value() ? *w2_ : Tensor();
    std::vector<Tensor> ws{w0};
    if (w1.defined()) ws.push_back(w1);
    if (w2.defined()) ws.push_back(w2);
    Tensor x2 = lin_rows(x);
    // norm kernel (y rounded to the activation dtype like the reference's layer_norm output, rstd kept) + ONE library GEMM on the stacked weights
    // (round 6: the hand-written GEMM with the norm in its prologue lost to hipBLASLt on every FAT5-base shape -- fused_linear.py)
    auto [y, rstd] = rms_fwd(x2, norm_weight, eps);
    Tensor wc = ws.size() == 1 ? ws[0] : fold_weights(ws, Tensor());
    Tensor out = at::matmul(y, wc.t());
    ctx->save_for_backward({x2, norm_weight, rstd, w0, w1, w2});
    ctx->saved_data["shape"] = x.sizes().vec();
    ctx->set_materialize_grads(false);
    std::vector<int64_t> oshape = x.sizes().vec();
    oshape.back() = wc.size(0);
    Tensor o = out.reshape(oshape);
    if (with_residual) return {o, x.view_as(x)};
    return {o};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto s = ctx->get_saved_variables();
    const Tensor &x2 = s[0], &g = s[1], &rstd = s[2];
    const Tensor dout = grads[0];
    Tensor dres = grads.size() > 1 ? grads[1] : Tensor();
    const auto shape = ctx->saved_data["shape"].toIntVector();
    if (!dout.defined()) return {dres, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};  // only the residual alias was used
    const int64_t M = x2.size(0), K = x2.size(1);
    Tensor d2 = lin_rows(dout.scalar_type() == x2.scalar_type() ? dout : dout.to(x2.scalar_type()));
    if (dres.defined()) dres = lin_rows(dres.scalar_type() == x2.scalar_type() ? dres : dres.to(x2.scalar_type()));
    std::vector<Tensor> wsv;
    for (int i = 0; i < 3; ++i)
      if (s[3 + i].defined()) wsv.push_back(s[3 + i]);
    Tensor wg = fold_weights(wsv, g);  // [W_i] diag(g)
    Tensor gy = at::matmul(d2, wg);    // dL/dxhat: (M, K)
    Tensor dx = at::empty({M, K}, x2.options()), xhat = at::empty({M, K}, x2.options());
    {
      c10::hip::HIPGuardMasqueradingAsCUDA guard(x2.device());
      check_rc(fat5_rmsnorm_unit_bwd(gy.data_ptr(), x2.data_ptr(), (const float*)rstd.data_ptr(), dx.data_ptr(), xhat.data_ptr(), M, K, gy.stride(0),
                                     x2.stride(0), K, K, dres.defined() ? dres.data_ptr() : nullptr, dres.defined() ? dres.stride(0) : 0,
                                     lin_dtype(x2), cur_stream(x2)), "fat5_rmsnorm_unit_bwd");
    }
    Tensor dg, dw[3];
    const bool need_g = ctx->needs_input_grad(1);
    // (needs_input_grad counts the TENSOR inputs that are present: x, norm_weight, w0 [, w1 [, w2]])
    const bool need_w[3] = {ctx->needs_input_grad(2), s[4].defined() && ctx->needs_input_grad(3), s[5].defined() && ctx->needs_input_grad(4)};
    if (need_g || need_w[0] || need_w[1] || need_w[2]) {
      Tensor dwg = at::matmul(d2.t(), xhat);  // gradient of the folded weight [W_i] diag(g): (N, K)
      std::vector<Tensor> ws;
      for (int i = 0; i < 3; ++i)
        if (s[3 + i].defined()) ws.push_back(lin_ready(s[3 + i]) ? s[3 + i] : s[3 + i].contiguous());
      int64_t n[3] = {0, 0, 0}, ld[3] = {0, 0, 0}, ntot = 0;
      const void* ptr[3] = {nullptr, nullptr, nullptr};
      void* dptr[3] = {nullptr, nullptr, nullptr};
      Tensor dws[3];
      for (size_t i = 0; i < ws.size(); ++i) {
        n[i] = ws[i].size(0); ld[i] = ws[i].stride(0); ptr[i] = ws[i].data_ptr(); ntot += n[i];
        dws[i] = at::empty({n[i], K}, x2.options());
        dptr[i] = dws[i].data_ptr();
      }
      Tensor gq = g.to(x2.scalar_type()).contiguous();
      Tensor dgq = at::empty({K}, x2.options());
      const size_t sbytes = fat5_fold_weights_bwd_scratch_bytes(ntot, K);
      Tensor scratch = at::empty({(int64_t)std::max<size_t>(sbytes, 4) / 4}, x2.options().dtype(at::kFloat));
      c10::hip::HIPGuardMasqueradingAsCUDA guard(x2.device());
      check_rc(fat5_fold_weights_bwd(dwg.data_ptr(), ptr[0], ptr[1], ptr[2], n[0], n[1], n[2], ld[0], ld[1], ld[2], gq.data_ptr(), dptr[0], dptr[1],
                                     dptr[2], dgq.data_ptr(), K, lin_dtype(x2), scratch.data_ptr(), (size_t)scratch.numel() * 4, cur_stream(x2)),
               "fat5_fold_weights_bwd");
      if (need_g) dg = dgq.to(g.scalar_type());
      for (size_t i = 0; i < ws.size(); ++i)
        if (need_w[i]) dw[i] = dws[i].to(s[3 + i].scalar_type());
    }
    return {ctx->needs_input_grad(0) ? dx.reshape(shape) : Tensor(), dg, Tensor(), Tensor(), dw[0], dw[1], dw[2]};
  }
};

// linear_residual (fused_linear.py::LinearResidual)
struct LinearResidualFn : public torch::autograd::Function<LinearResidualFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& a, const Tensor& weight, const Tensor& residual) {
    Tensor a2 = lin_rows(a), r2 = lin_rows(residual);
    Tensor out = at::addmm(r2, a2, weight.t());  // (the residual add in the library GEMM's epilogue)
    ctx->save_for_backward({a2, weight});
    ctx->saved_data["ashape"] = a.sizes().vec();
    return out.reshape(residual.sizes());
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto s = ctx->get_saved_variables();
    const Tensor &a2 = s[0], &W = s[1];
    const Tensor& dout = grads[0];
    Tensor d2 = dout.reshape({-1, dout.size(-1)});
    Tensor da, dW;
    if (ctx->needs_input_grad(0)) da = at::matmul(d2, W).reshape(ctx->saved_data["ashape"].toIntVector());
    if (ctx->needs_input_grad(1)) dW = at::matmul(d2.t(), a2).to(W.scalar_type());
    return {da, dW, ctx->needs_input_grad(2) ? dout : Tensor()};
  }
};

// gated activation (gated_act.py): h0, h1 the two projections -- for the packed form the two halves of one (rows, 2F) tensor
Tensor gated_fwd(const Tensor& h0, const Tensor& h1, int64_t act) {
  Tensor a = lin_rows(h0), b = lin_rows(h1);
  Tensor out = at::empty(a.sizes(), a.options());
  if (a.size(0)) {
    c10::hip::HIPGuardMasqueradingAsCUDA guard(a.device());
    check_rc(fat5_gated_act_fwd(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.size(0), a.size(1), a.stride(0), b.stride(0), out.stride(0), (int)act,
                                any_dtype_code(a), cur_stream(a)), "fat5_gated_act_fwd");
  }
  return out.reshape(h0.sizes());
}
Tensor gated_bwd(const Tensor& dout, const Tensor& h0, const Tensor& h1, int64_t act) {  // -> (rows, 2F): [dh0 | dh1]
  Tensor a = lin_rows(h0), b = lin_rows(h1);
  Tensor g = lin_rows(dout.scalar_type() == a.scalar_type() ? dout : dout.to(a.scalar_type()));
  const int64_t F = a.size(1);
  Tensor dh = at::empty({a.size(0), 2 * F}, a.options());
  if (a.size(0)) {
    c10::hip::HIPGuardMasqueradingAsCUDA guard(a.device());
    check_rc(fat5_gated_act_bwd(g.data_ptr(), a.data_ptr(), b.data_ptr(), dh.data_ptr(), (char*)dh.data_ptr() + F * dh.element_size(), a.size(0), F,
                                g.stride(0), a.stride(0), b.stride(0), 2 * F, 2 * F, (int)act, any_dtype_code(a), cur_stream(a)),
             "fat5_gated_act_bwd");
  }
  return dh;
}
struct GatedActPackedFn : public torch::autograd::Function<GatedActPackedFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& h, int64_t act) {
    const int64_t F = h.size(-1) / 2;
    ctx->save_for_backward({h});
    ctx->saved_data["act"] = act;
    return gated_fwd(h.narrow(-1, 0, F), h.narrow(-1, F, F), act);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto s = ctx->get_saved_variables();
    const Tensor& h = s[0];
    const int64_t F = h.size(-1) / 2;
    return {gated_bwd(grads[0], h.narrow(-1, 0, F), h.narrow(-1, F, F), ctx->saved_data["act"].toInt()).reshape(h.sizes()), Tensor()};
  }
};
struct GatedActFn : public torch::autograd::Function<GatedActFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& h0, const Tensor& h1, int64_t act) {
    ctx->save_for_backward({h0, h1});
    ctx->saved_data["act"] = act;
    return gated_fwd(h0, h1, act);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto s = ctx->get_saved_variables();
    const int64_t F = s[0].size(-1);
    Tensor dh = gated_bwd(grads[0], s[0], s[1], ctx->saved_data["act"].toInt());
    return {dh.narrow(1, 0, F).reshape(s[0].sizes()), dh.narrow(1, F, F).reshape(s[1].sizes()), Tensor()};
  }
};

// unpack_heads (attention_module.py::_UnpackHeads): (B, S, n H D) -> n views (B, H, S, D); the backward hands the packed gradient
// buffer on when the incoming gradients are its slices (attn_bwd above allocates them that way), one stack otherwise
struct UnpackHeadsFn : public torch::autograd::Function<UnpackHeadsFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& x, int64_t n, int64_t H) {
    const int64_t B = x.size(0), S = x.size(1), D = x.size(2) / (n * H);
    ctx->saved_data["dims"] = std::vector<int64_t>{B, S, n, H, D};
    ctx->set_materialize_grads(false);
    Tensor p = x.view({B, S, n, H, D});
    variable_list out;
    for (int64_t i = 0; i < n; ++i) out.push_back(p.select(2, i).permute({0, 2, 1, 3}));
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list gs) {
    const auto d = ctx->saved_data["dims"].toIntVector();
    const int64_t B = d[0], S = d[1], n = d[2], H = d[3], D = d[4], row = n * H * D;
    bool all = true;
    for (const Tensor& g : gs) all = all && g.defined();
    if (all) {
      const Tensor& g0 = gs[0];
      bool same = true;
      for (int64_t i = 0; i < n && same; ++i) {
        const Tensor& g = gs[i];
        same = g.storage().data() == g0.storage().data() && g.scalar_type() == g0.scalar_type() && g.stride(0) == S * row && g.stride(1) == D &&
               g.stride(2) == row && g.stride(3) == 1 && g.storage_offset() == g0.storage_offset() + i * H * D;
      }
      if (same && (int64_t)g0.storage().nbytes() >= (g0.storage_offset() + B * S * row) * (int64_t)g0.element_size())
        return {g0.as_strided({B, S, row}, {S * row, row, 1}, g0.storage_offset()), Tensor(), Tensor()};
    }
    Tensor ref;
    for (const Tensor& g : gs)
      if (g.defined()) ref = g;
    std::vector<Tensor> parts;
    for (const Tensor& g : gs) parts.push_back((g.defined() ? g : at::zeros_like(ref)).permute({0, 2, 1, 3}));
    return {at::stack(parts, 2).reshape({B, S, row}), Tensor(), Tensor()};
  }
};

std::vector<Tensor> rmsnorm_linear_apply(const Tensor& x, const Tensor& norm_weight, double eps, bool with_residual, const std::vector<Tensor>& ws) {
  TORCH_CHECK(ws.size() >= 1 && ws.size() <= 3, "rmsnorm_linear: one to three weights");
  return RmsNormLinearFn::apply(x, norm_weight, eps, with_residual, ws[0], ws.size() > 1 ? OptT(ws[1]) : OptT(), ws.size() > 2 ? OptT(ws[2]) : OptT());
}
Tensor linear_residual_apply(const Tensor& a, const Tensor& weight, const Tensor& residual) { return LinearResidualFn::apply(a, weight, residual); }
Tensor gated_act_packed_apply(const Tensor& h, int64_t act) { return GatedActPackedFn::apply(h, act); }
Tensor gated_act_apply(const Tensor& h0, const Tensor& h1, int64_t act) { return GatedActFn::apply(h0, h1, act); }
std::vector<Tensor> unpack_heads_apply(const Tensor& x, int64_t n, int64_t H) { return UnpackHeadsFn::apply(x, n, H); }

}  // namespace

PYBIND11_MODULE(_fat5_torch, m) {
  m.doc() = "native host path of the flasht5_amd attention operators (at::Tensor -> libfat5.so C ABI)";
  m.def("attn_fwd", &attn_fwd);
  m.def("attn_bwd", &attn_bwd);
  m.def("bias_apply", &bias_apply);
  m.def("rpe_table_apply", &rpe_table_apply);
  m.def("rpe1d_apply", &rpe1d_apply);
  m.def("rpe1d_of", &rpe1d_of);
  m.def("rmsnorm_apply", &rmsnorm_apply);
  m.def("rmsnorm_linear_apply", &rmsnorm_linear_apply);
  m.def("linear_residual_apply", &linear_residual_apply);
  m.def("gated_act_packed_apply", &gated_act_packed_apply);
  m.def("gated_act_apply", &gated_act_apply);
  m.def("unpack_heads_apply", &unpack_heads_apply);
  m.def("add_rmsnorm_apply", &add_rmsnorm_apply);
  m.def("sizeof_attn_params", []() { return (int64_t)sizeof(fat5_attn_params); });
  m.def("set_variant", [](int64_t bits) { g_variant.store((int)bits); });
}
