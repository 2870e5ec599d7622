/*
 * fat5.h -- C ABI of libfat5.so: MI355X-native (gfx950) FlashAttention-2 with an additive
 * T5 relative-position bias (forward + backward), T5 RMSNorm and cross-entropy + z-loss.
 *
 * Every entry point replaces one `flasht5::*` custom op of the reference (catie-aq/flashT5);
 * the reference interface each one stands in for is cited next to it (paths relative to the
 * reference tree).  The reference has no native code: its FFI for this path is the Python
 * `torch.library.custom_op` layer, so the binding a maintainer would add is a ctypes stub
 * (see INTEGRATION.md).
 *
 * Conventions
 *  - plain pointers and sizes only; no torch / HIP C++ types (hipStream_t is passed as void*).
 *  - the caller owns every buffer (inputs, outputs, workspace); the library never allocates or
 *    frees device memory.  Its only process-wide state is idempotent launch configuration (the largest dynamic-LDS size
 *    already requested per kernel).  It reads NO environment variables: results depend on the arguments of the call alone,
 *    never on call history, and every entry point is safe to call from several threads.  All launches are asynchronous on the given
 *    stream; the library never synchronises.
 *  - return value: FAT5_OK (0) or a negative error; the message of the last error on the
 *    calling thread is available from fat5_last_error().  Nothing throws across the ABI.
 *  - strides are in ELEMENTS.  The innermost (head_dim / feature / vocab) stride must be 1.
 */
#ifndef FAT5_H
#define FAT5_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FAT5_VERSION 114 /* 0.1.4 (114: FAT5_V_QDIAG_ON / _OFF, fat5_chip_cus, fat5_linear_fused removed, -inf-safe bias operands in the bodies that add the bias on the matrix pipe; 113: head_dim 16 native, FAT5_V_DBIAS_NOSPLIT; 112: FAT5_V_FUSED64_ON / _OFF); 0.1.1: per-call kernel-variant bits (no environment variables), fat5_rpe1d_from_table,
                            AdamWScale state dtype / flags; 111: fat5_fold_weights_bwd takes scratch, fat5_gated_act_*, fat5_adamw_scale_step_dev */

enum fat5_status {
  FAT5_OK = 0,
  FAT5_EINVAL = -1, /* unsupported head_dim / dtype / stride / alignment / shape */
  FAT5_EHIP = -2,   /* a HIP runtime call failed (text in fat5_last_error) */
  FAT5_EWORKSPACE = -3 /* workspace missing or too small */
};

enum fat5_dtype { FAT5_F16 = 1, FAT5_BF16 = 2, FAT5_F32 = 0 };

/* fat5_attn_params.variant bits (testing / profiling only) */
enum fat5_variant {
  FAT5_V_FWD64_ON = 1, FAT5_V_FWD64_OFF = 2,   /* forward: 64-rows-per-wave pipelined body (attn_fwd64.h) wherever it applies / never */
  FAT5_V_KV64_ON = 4, FAT5_V_KV64_OFF = 8,     /* backward dK/dV: 64-keys-per-wave pipelined body (attn_bwd64.h) */
  FAT5_V_Q64_ON = 16, FAT5_V_Q64_OFF = 32,     /* backward dQ: 64-rows-per-wave pipelined body */
  FAT5_V_DBIAS_STAGED = 64, FAT5_V_DBIAS_INKERNEL = 128, /* dense (1,H,M,N) dbias: staged dS + reduction / batch-inner kernel */
  FAT5_V_NO_FUSE = 256,                         /* backward: never the single side-by-side dQ | dK/dV launch */
  FAT5_V_NO_SPLIT = 512,                        /* forward: never the two-waves-per-32-rows short-sequence body */
  FAT5_V_FWD64_KSPLIT_ON = 1024, FAT5_V_FWD64_KSPLIT_OFF = 2048, /* 64-row forward: key-split (two waves per 64 rows) variant always / never */
  FAT5_V_KV64_HALF_ON = 4096, FAT5_V_KV64_HALF_OFF = 8192,       /* 64-key dK/dV body: half-length (128-key workgroup) variant always / never */
  FAT5_V_KV64_MIX_ON = 16384, FAT5_V_KV64_MIX_OFF = 32768,       /* 64-key dK/dV body: 256-key and half-length workgroups in one launch wherever legal / never */
  FAT5_V_FWD64_MIX_ON = 524288, FAT5_V_FWD64_MIX_OFF = 1048576,  /* 64-row forward: 256-row and key-split 128-row workgroups in ONE launch wherever legal / never */
  FAT5_V_DBIAS_NOSPLIT = 262144,                                 /* batch-inner dbias kernel: the one-group form (one wave per SIMD) of rounds 2-3 */
  FAT5_V_QDB64_ON = 2097152, FAT5_V_QDB64_OFF = 4194304,        /* dense (1,H,M,N) bias, bf16, D = 64: dQ and the batch-reduced dbias in one kernel, four batch elements per workgroup (attn_bwd_qdb64.h) wherever legal / never */
  FAT5_V_QDIAG_ON = 8388608, FAT5_V_QDIAG_OFF = 16777216,       /* T5 bias, one-launch 64-wide backward: the table gradient's per-diagonal sums formed by the dQ workgroups (attn_bwd_q64_body<QDG>) wherever that launch runs / never */
  FAT5_V_FUSED64_ON = 65536, FAT5_V_FUSED64_OFF = 131072         /* backward: the 64-wide dK/dV and dQ bodies in ONE launch (the dK/dV half forms its row statistics itself) wherever legal / never; dense (1,H,M,N) bias: the dense dK/dV body beside the dQ + dBias body in one launch behind a small row-statistics kernel */
};

enum fat5_bias_mode {
  FAT5_BIAS_NONE = 0,  /* bias=None (reference HAS_BIAS=False, e.g. T5 cross-attention) */
  FAT5_BIAS_DENSE = 1, /* additive bias (B|1, H|1, M, N), same dtype as q (reference path) */
  FAT5_BIAS_RPE1D = 2  /* Toeplitz bias given by its clamped generator (H, 2R+1) fp32:
                          bias[h][m][n] = rpe1d[h][clamp(n - m, -R, R) + R]  (linear memory) */
};

/*
 * Attention problem descriptor, shared by forward and backward.
 *
 * Replaces: flasht5::flash_attn_v2_fwd  (src/model/ops/flash_attention_v2_bias.py:27-80)
 *           flasht5::flash_attn_v2_bwd  (src/model/ops/flash_attention_v2_bias.py:91-217)
 * Semantics: o = softmax(q k^T * sm_scale + bias [+ bottom-right causal mask]) v;
 *            lse = natural-log LSE per row, fp32, (B,H,M) contiguous (:476,:59);
 *            fully masked rows (causal and M > N) give o = 0, lse = -inf (:470-473).
 *            dbias is the gradient of the UNSCALED additive term (dS), reduced over every
 *            broadcast dimension of bias (mathematically correct also for (1,1,M,N): SURVEY Q4).
 */
typedef struct fat5_attn_params {
  /* ---- problem ---- */
  int32_t B, H, M, N, D; /* D in {16, 32, 64, 128} (16: the D = 32 kernels with columns 16..31 read as zeros and never written; rows are 32 bytes: strides stay multiples of 8 elements) */
  int32_t dtype;         /* FAT5_F16 | FAT5_BF16 : dtype of q,k,v,o,do,dq,dk,dv and of dense bias */
  int32_t causal;        /* bottom-right aligned: key n visible to query m iff m + (N-M) >= n */
  int32_t bias_mode;     /* enum fat5_bias_mode */
  float sm_scale;
  int32_t rpe_radius;    /* R for FAT5_BIAS_RPE1D: 1..2048 forward; the backward needs its per-wave accumulators in LDS
                            and accepts what fits 160 KiB (R <= 1024 for every head_dim; FAT5_EINVAL beyond) */
  /* ---- forward tensors ---- */
  const void* q; /* (B,H,M,D) strides q_stride[b,h,m] */
  const void* k; /* (B,H,N,D) */
  const void* v; /* (B,H,N,D) */
  void* o;       /* (B,H,M,D) */
  float* lse;    /* (B,H,M) contiguous fp32 */
  int64_t q_stride[3], k_stride[3], v_stride[3], o_stride[3];
  const void* bias;        /* DENSE: (Bb,Hb,M,N), Bb in {1,B}, Hb in {1,H}; n-stride 1 */
  int64_t bias_stride[3];  /* [b,h,m]; 0 for a broadcast dimension */
  const float* rpe1d;      /* RPE1D: (H, 2R+1) fp32 contiguous */
  /* ---- packed var-len (optional; reference has none -- SURVEY 8(f) n2) ----
   * when cu_seqlens_q != NULL: B = number of sequences, q/o are (total_q, H, D) addressed as
   * base + (cu_seqlens_q[b] + m) * stride[2] + h * stride[1] (stride[0] ignored), k/v likewise
   * with cu_seqlens_k; M/N are the MAXIMUM lengths; lse is (H, total_q).  Forward and backward (dq like q, dk/dv like
   * k/v, dout like o); bias_mode FAT5_BIAS_NONE or FAT5_BIAS_RPE1D (relative positions count from each sequence's
   * own start: bias[m][n] = rpe1d[h][clamp(n - m, -R, R) + R] with m, n local to the sequence). */
  const int32_t* cu_seqlens_q;
  const int32_t* cu_seqlens_k;
  int32_t total_q, total_k;
  /* ---- backward tensors (ignored by fat5_attn_fwd) ---- */
  const void* dout; /* (B,H,M,D) */
  void* dq;         /* (B,H,M,D) */
  void* dk;         /* (B,H,N,D) */
  void* dv;         /* (B,H,N,D) */
  int64_t do_stride[3], dq_stride[3], dk_stride[3], dv_stride[3];
  void* dbias;              /* DENSE: same shape/dtype as bias, contiguous; may be NULL */
  int32_t dbias_batch, dbias_heads; /* Bb, Hb of bias/dbias */
  float* drpe1d;            /* RPE1D: (H, 2R+1) fp32, overwritten; may be NULL */
  /* RPE1D, optional: gradient of the T5 table itself, scattered in the same reduction launch.
   * rpe_bucket[i] = bucket id of clamped relative position i - R (i in [0, 2R]); drpe_table is
   * (rpe_num_buckets, H) fp32, overwritten: drpe_table[b][h] = sum_{i: rpe_bucket[i]=b} drpe1d[h][i]
   * (what autograd's embedding backward computes from the dense dbias). */
  const int32_t* rpe_bucket;
  float* drpe_table;
  int32_t rpe_num_buckets;
  /* ---- unit range (optional): the B*H independent (batch, head) problems in head-major order, u = h * B + b -- the order in
   * which SURVEY 8(e) deals them to ranks (the reference's grid axes 1, 2: flash_attention_v2_bias.py:57,164,192).
   * unit_count > 0: forward and backward touch ONLY units [unit_begin, unit_begin + unit_count): every tensor keeps the full
   * (B, H, ...) geometry and strides, other units' slices are neither read nor written; lse and the workspace keep their
   * full-problem layout; drpe1d / drpe_table receive the partial sums over this range's units (heads without a unit in the
   * range get zeros) -- ranks then add their partial tables with ONE all-reduce.  Not with cu_seqlens; dense dbias only in
   * its unreduced (B, H, M, N) form.  unit_count == 0: the whole problem. */
  int32_t unit_begin;
  int32_t unit_count;
  int32_t variant;          /* 0 in production: the library picks every kernel variant from the problem alone.  Tests and
                               profilers OR fat5_variant bits here to force / forbid a body for THIS call (no environment
                               variables, no process-wide switches: the choice is part of the call). */
  void* workspace;          /* size from fat5_attn_bwd_workspace_bytes(); 256-B aligned */
  size_t workspace_bytes;
} fat5_attn_params;

int fat5_version(void);
/* compute units of the current device as the dispatch rules see them (their workgroup-count thresholds are rounds of the chip and scale with it);
 * 256 -- the MI355X the rules were measured on -- when no device is present.  Host-only tests that pin dispatch choices check it. */
int fat5_chip_cus(void);
const char* fat5_last_error(void);
/* sizeof(fat5_attn_params) as compiled into the library (bindings check their mirror against it). */
size_t fat5_sizeof_attn_params(void);

/* forward: writes o, lse. */
int fat5_attn_fwd(const fat5_attn_params* p, void* hip_stream);
/* bytes of scratch the backward needs for this problem (delta, dS staging, partial sums). */
size_t fat5_attn_bwd_workspace_bytes(const fat5_attn_params* p);
/* backward: reads q,k,v,o,lse,dout(,bias|rpe1d); writes dq,dk,dv(,dbias|drpe1d). */
int fat5_attn_bwd(const fat5_attn_params* p, void* hip_stream);
/* number of MAIN kernel launches fat5_attn_bwd uses for this problem: 1 = dQ and dK/dV halves side by side in one
 * launch (short sequences: both grids fit the chip together), 2 = dQ kernel then dK/dV kernel; 0 on invalid params.
 * Not counted: the reduction launch behind them (table gradient / staged dS / fp32 dbias slabs of B > 4) and, for the one-launch form of the
 * dense (1,H,M,N) bias, the small row-statistics kernel ahead of it (bwd_stat2_kernel) -- `fat5_attn_describe` names the bodies, profilers use
 * this number only to name the dominant kernel. */
int fat5_attn_bwd_launches(const fat5_attn_params* p);
/* Which kernel bodies this problem runs -- "fwd=64row-ksplit dq=32row dkdv=64key-mixed:4 fused=0 dbias=direct" -- written to `out`
 * (n bytes).  Host-only (no device, no pointer of `p` is followed): tests pin the dispatch rules with it. */
int fat5_attn_describe(const fat5_attn_params* p, char* out, size_t n);
/* the same backward, one stage at a time (profiling / stream overlap).  Order matters:
 * FAT5_BWD_DQ (writes delta + dq) must precede FAT5_BWD_DKDV (reads delta; writes dk, dv, dS / partial
 * diagonal sums), which must precede FAT5_BWD_REDUCE (dbias / drpe1d).  fat5_attn_bwd == FAT5_BWD_ALL.
 * FAT5_BWD_DQ | FAT5_BWD_DKDV in one call may use the single side-by-side launch (fat5_attn_bwd_launches). */
enum fat5_bwd_stage { FAT5_BWD_DQ = 1, FAT5_BWD_DKDV = 2, FAT5_BWD_REDUCE = 4, FAT5_BWD_ALL = 7 };
int fat5_attn_bwd_stages(const fat5_attn_params* p, int stages, void* hip_stream);

/* T5 table -> Toeplitz generator of FAT5_BIAS_RPE1D: rpe1d[h][i] = (float) table[rpe_bucket[i]][h], i in [0, 2R]
 * (the embedding lookup of RelativePositionalEncoding.compute_bias, src/utils/positional_encoding.py:100-101, on the
 * 2R+1 distinct clamped relative positions instead of M*N of them).  table: (num_buckets, H) contiguous, table_dtype in
 * {FAT5_F32, FAT5_F16, FAT5_BF16}; rpe_bucket: (2R+1,) int32; rpe1d: (H, 2R+1) fp32, overwritten.  One small launch. */
int fat5_rpe1d_from_table(const void* table, int table_dtype, const int32_t* rpe_bucket, float* rpe1d, int32_t H,
                          int32_t rpe_radius, int32_t num_buckets, void* hip_stream);

/*
 * T5 RMSNorm.  Replaces flasht5::rmsnorm_triton_fwd / _bwd (src/model/ops/rms_norm.py:134-236).
 *   y = x * rsqrt(mean(x^2) + eps) * w   (fp32 math, y in x's dtype);  rstd (rows,) fp32.
 *   dx = (w*dy - xhat*mean(xhat*w*dy)) * rstd;  dw = sum_rows(dy * xhat)  (fp32 partials, cast to w dtype)
 * x_dtype / w_dtype in {FAT5_F32, FAT5_F16, FAT5_BF16}; dy has x's dtype.
 */
int fat5_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int64_t n,
                     int64_t x_row_stride, int64_t y_row_stride, float eps, int x_dtype, int w_dtype,
                     void* hip_stream);
/* Residual add + RMSNorm (the residual epilogue of a T5 sub-layer fused into the next pre-norm, reference
 * src/model/modeling_flash_t5.py:159-164 + :304-318 / :95-98): h = x + r rounded to x_dtype, y = rmsnorm(h) * w, rstd saved --
 * bit-identical to the separate add followed by fat5_rmsnorm_fwd.  x, r, h, y: (rows, n) with row strides in elements. */
int fat5_add_rmsnorm_fwd(const void* x, const void* r, const void* w, void* h, void* y, float* rstd, int64_t rows, int64_t n,
                         int64_t x_row_stride, int64_t r_row_stride, int64_t h_row_stride, int64_t y_row_stride, float eps,
                         int x_dtype, int w_dtype, void* hip_stream);
/* its backward: dx = round(rmsnorm_bwd_dx(dy, h, w, rstd)) + dres (dres = gradient reaching h through the residual stream, may be
 * NULL), dw as fat5_rmsnorm_bwd; dx is the gradient of BOTH x and r.  Workspace: fat5_rmsnorm_bwd_workspace_bytes(rows, n). */
int fat5_add_rmsnorm_bwd(const void* dy, const void* h, const void* w, const float* rstd, const void* dres, void* dx, void* dw,
                         int64_t rows, int64_t n, int64_t dy_row_stride, int64_t h_row_stride, int64_t dres_row_stride,
                         int64_t dx_row_stride, int x_dtype, int w_dtype, void* workspace, size_t workspace_bytes,
                         void* hip_stream);
size_t fat5_rmsnorm_bwd_workspace_bytes(int64_t rows, int64_t n);
int fat5_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, void* dx, void* dw,
                     int64_t rows, int64_t n, int64_t dy_row_stride, int64_t x_row_stride,
                     int64_t dx_row_stride, int x_dtype, int w_dtype, void* workspace,
                     size_t workspace_bytes, void* hip_stream);

/*
 * Stacked projections of a T5 block: layer_norm -> Wq / Wk / Wv (src/model/modeling_flash_t5.py:304-318, :95-98) and
 * layer_norm -> wi_0 / wi_1 (:159-160) as ONE library GEMM on the stacked weight.  (Versions <= 113 exported fat5_linear_fused, a hand-written
 * MFMA GEMM with the norm in its prologue: slower than the library GEMM on every FAT5-base shape, removed in 114.)
 */
/* The stacked projection weight in one launch: out = [w0; w1; w2] (rows stacked: (n0 + n1 + n2, K),
 * contiguous) with every row multiplied elementwise by the norm weight g (K,) (g == NULL: the plain stack) -- e.g. Wq, Wk, Wv of a
 * T5 attention block and its layer_norm weight.  n1 / n2 may be 0.  All tensors share `dtype` (16-bit); products rounded once. */
int fat5_fold_weights(const void* w0, const void* w1, const void* w2, int64_t n0, int64_t n1, int64_t n2, int64_t ld0, int64_t ld1,
                      int64_t ld2, const void* g, void* out, int64_t K, int dtype, void* hip_stream);
/* Backward pieces of Linear(RMSNorm(x; g), [w0; w1; w2]) around its two library GEMMs:
 *  - fat5_rmsnorm_unit_bwd: gy = dout (W diag g) = dL/dxhat, xhat = x * rstd ->  dx = (gy - xhat * mean_k(xhat * gy)) * rstd (the
 *    backward of rms_norm.py:113-124 with unit weight), and xhat itself -- the operand of the dout^T xhat GEMM -- in the same pass;
 *    x_dtype tensors, n <= 2048 (16-bit) / 1024 (fp32).  dres (optional, NULL = none): the gradient arriving at x along the
 *    residual connection of the sub-layer (h + f(norm(h))), added to dx in the same pass (fp32 sum, one rounding).
 *  - fat5_fold_weights_bwd: dwg (N, K) = dout^T xhat, the gradient of the folded weight [w0; w1; w2] diag(g) ->
 *    dw_i = dwg rows * g (contiguous (n_i, K), any of them may be NULL), dg[k] = sum_n dwg[n][k] * w[n][k] (fp32, fixed order; may be NULL).
 *    K a multiple of 64.  dg needs `scratch` of fat5_fold_weights_bwd_scratch_bytes(n0 + n1 + n2, K) bytes (row slabs are summed by
 *    separate workgroups, the slab sums added in order by a second launch); contents need no initialisation. */
int fat5_rmsnorm_unit_bwd(const void* gy, const void* x, const float* rstd, void* dx, void* xhat, int64_t rows, int64_t n,
                          int64_t gy_row_stride, int64_t x_row_stride, int64_t dx_row_stride, int64_t xhat_row_stride,
                          const void* dres, int64_t dres_row_stride, int dtype, void* hip_stream);
int fat5_fold_weights_bwd(const void* dwg, const void* w0, const void* w1, const void* w2, int64_t n0, int64_t n1, int64_t n2, int64_t ld0,
                          int64_t ld1, int64_t ld2, const void* g, void* dw0, void* dw1, void* dw2, void* dg, int64_t K, int dtype,
                          void* scratch, size_t scratch_bytes, void* hip_stream);
size_t fat5_fold_weights_bwd_scratch_bytes(int64_t n_total, int64_t K);

/*
 * Gated activation of the T5 v1.1 feed-forward: out = act(h0) * h1 (reference FlashT5DenseGatedAct.forward,
 * src/model/modeling_flash_t5.py:139-142; act = GELU(approximate='tanh') when config.use_gelu_act, else ReLU, :134) and its backward
 *   dh0 = dout * h1 * act'(h0),  dh1 = dout * act(h0)
 * in one pass each.  All tensors (rows, F) in `dtype`, addressed by row strides (elements): h0 / h1 may be the halves of one
 * (rows, 2F) projection output and dh0 / dh1 the halves of its gradient.  F and the strides multiples of the 16-byte vector
 * (8 elements; 4 for fp32), bases 16-byte aligned.
 */
enum fat5_act { FAT5_ACT_GELU_TANH = 0, FAT5_ACT_RELU = 1 };
int fat5_gated_act_fwd(const void* h0, const void* h1, void* out, int64_t rows, int64_t F, int64_t h0_row_stride, int64_t h1_row_stride,
                       int64_t out_row_stride, int act, int dtype, void* hip_stream);
int fat5_gated_act_bwd(const void* dout, const void* h0, const void* h1, void* dh0, void* dh1, int64_t rows, int64_t F,
                       int64_t dout_row_stride, int64_t h0_row_stride, int64_t h1_row_stride, int64_t dh0_row_stride,
                       int64_t dh1_row_stride, int act, int dtype, void* hip_stream);

/*
 * Cross-entropy + label smoothing + z-loss.  Replaces flasht5::cross_entropy_triton_fwd / _bwd
 * (src/model/ops/cross_entropy_loss.py:164-274), single-rank path (SPLIT = False).
 *   lse = log sum exp(logits*logit_scale);  loss = lse - logit[label]  (smoothed variant :90-95)
 *   z_loss = lse_square_scale * lse^2 (added to loss); rows with label == ignore_index give 0.
 *   dlogits = dloss*logit_scale*(softmax*(1 + 2*lse_square_scale*lse) - onehot/smoothing terms);
 *   dlogits may alias logits (in-place backward, :247).
 * labels are int64.
 */
int fat5_ce_fwd(const void* logits, const int64_t* labels, float* losses, float* z_losses, float* lse,
                int64_t rows, int64_t n_cols, int64_t row_stride, float smoothing, float logit_scale,
                float lse_square_scale, int64_t ignore_index, int use_precomputed_lse, int dtype,
                void* hip_stream);
int fat5_ce_bwd(const float* dlosses, int64_t dloss_stride, const void* logits, const float* lse,
                const int64_t* labels, void* dlogits, int64_t rows, int64_t n_cols,
                int64_t row_stride, int64_t dlogits_row_stride, float smoothing, float logit_scale,
                float lse_square_scale, int64_t ignore_index, int dtype, void* hip_stream);
/* Both in one launch, the row read once (round 4; for callers that know d loss / d losses before the forward: a mean or sum loss --
 * flasht5_amd/lm_head_cross_entropy.py).  losses / z_losses / lse / dlogits are bit-identical to fat5_ce_fwd followed by fat5_ce_bwd;
 * dlogits may be logits (in place). */
int fat5_ce_fwd_bwd(const void* logits, const int64_t* labels, const float* dlosses, int64_t dloss_stride, float* losses,
                    float* z_losses, float* lse, void* dlogits, int64_t rows, int64_t n_cols, int64_t row_stride,
                    int64_t dlogits_row_stride, float smoothing, float logit_scale, float lse_square_scale, int64_t ignore_index,
                    int dtype, void* hip_stream);

/*
 * AdamWScale step over a group of tensors.  Replaces the reference optimizer's per-tensor / foreach op
 * sequences (src/utils/adamw_scaled.py:154-211, :213-281) with two launches for the whole group.
 *   m = beta1 m + (1-beta1) g;  v = beta2 v + (1-beta2) g^2;  denom = sqrt(v) + eps
 *   step = step_prefactor * max(1e-3, rms(p));     step_prefactor = lr [* sqrt(1-beta2^t) / (1-beta1^t)], computed by the caller
 *   p -= step * m / denom   (with `kahan`: through the compensation tensor k, :188-198);   p += -lr * weight_decay * p
 * Intermediate roundings are the reference's (every in-place op rounds to the tensor dtype; fp32 math inside an op).
 * Parameters, gradients and k of one call share `dtype`; m and v have `state_dtype` (== dtype by default; FAT5_F16 / FAT5_BF16 for the
 * reference's `use_state_dtype`, :101-103 -- 16-bit moments beside parameters of another dtype; every op then rounds to the dtype
 * of the tensor it writes, fp32 math inside).  `flags`: FAT5_ADAMW_KAHAN (16-bit parameters only); FAT5_ADAMW_PLAIN_STEP =
 * `correct_bias=False` (the reference's step size lr * max(1e-3, rms(p)) is then rounded to the parameter dtype, :177-184).
 * `table` is a DEVICE array of n_tensors descriptors (+ one terminator whose chunk_begin is the total chunk count);
 * chunk_begin[i] = sum over j < i of ceil(numel[j] / 8192);  `partials` = device scratch of total-chunks floats.
 */
typedef struct fat5_adamw_tensor {
  void* p;              /* parameters, updated in place */
  const void* g;        /* gradients */
  void* m;              /* exp_avg */
  void* v;              /* exp_avg_sq (m, v: state_dtype) */
  void* k;              /* Kahan compensation (kahan != 0), else NULL */
  int64_t numel;
  int32_t chunk_begin;
  float step_prefactor;
} fat5_adamw_tensor;
enum fat5_adamw_flags { FAT5_ADAMW_KAHAN = 1, FAT5_ADAMW_PLAIN_STEP = 2 };
/* hyper-parameters as doubles: the reference passes Python floats, and e.g. (1 - beta2) is formed in double before the op casts it */
int fat5_adamw_scale_step(const fat5_adamw_tensor* table, int32_t n_tensors, int32_t n_chunks, float* partials, double lr,
                          double beta1, double beta2, double weight_decay, double eps, int dtype, int state_dtype, int flags,
                          void* hip_stream);
/* Global-norm gradient clipping folded into the step (torch.nn.utils.clip_grad_norm_ + AdamWScale.step in one pass; the
 * reference trains with `max_grad_norm: 1.0`, configs/flan/fat5-flan-base.yaml): fat5_adamw_grad_sumsq writes one partial sum of
 * squares of the GRADIENTS per 8192-element chunk (the caller sums them over every group, forms
 * coef = min(1, max_norm / (sqrt(sum) + 1e-6)) on the device) and fat5_adamw_scale_step_clipped reads that fp32 scalar:
 * every gradient enters the update as round_to_dtype(g * coef), the value clip_grad_norm_'s in-place multiply would have left.
 * The gradient tensors themselves are not modified. */
int fat5_adamw_grad_sumsq(const fat5_adamw_tensor* table, int32_t n_tensors, int32_t n_chunks, float* partials, int dtype,
                          void* hip_stream);
int fat5_adamw_scale_step_clipped(const fat5_adamw_tensor* table, int32_t n_tensors, int32_t n_chunks, float* partials, double lr,
                                  double beta1, double beta2, double weight_decay, double eps, int dtype, int state_dtype,
                                  int flags, const float* grad_coef, void* hip_stream);
/* The same step with its step-dependent scalars in DEVICE memory -- dev_scalars[0] = the step prefactor lr * sqrt(1 - beta2^t) /
 * (1 - beta1^t) (or lr with FAT5_ADAMW_PLAIN_STEP; overrides every table entry's step_prefactor: one step count for the group),
 * [1] = -lr * weight_decay (0 when weight_decay is 0), [2] = lr * 1e-3 -- so that a launch captured in a HIP graph follows the
 * learning-rate schedule and the bias correction: the host writes three floats before each replay (stream-ordered), nothing in the
 * graph changes.  grad_coef may be NULL (no clipping). */
int fat5_adamw_scale_step_dev(const fat5_adamw_tensor* table, int32_t n_tensors, int32_t n_chunks, float* partials, const float* dev_scalars,
                              double beta1, double beta2, double eps, int dtype, int state_dtype, int flags, const float* grad_coef,
                              void* hip_stream);
size_t fat5_sizeof_adamw_tensor(void);

#ifdef __cplusplus
}
#endif
#endif /* FAT5_H */
