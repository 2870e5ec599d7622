"""FlashAttention-2 with an additive (T5 relative-position) bias on MI355X.

Host-side mirror of the reference operator file src/model/ops/flash_attention_v2_bias.py: the same
public callable `flash_attention_v2_bias(q, k, v, bias, causal=False, sm_scale=None)` backed by the same
`torch.autograd.Function` signature (`FlashAttentionAdditiveBias`, reference :228-271) and a pair of
custom ops with fake impls (reference :27-89, :91-226) -- but the kernels are hand-written gfx950 HIP
behind the C ABI of libfat5.so instead of Triton.

Extra, linear-memory entry point: `flash_attention_v2_rpe(q, k, v, rpe_table, ...)` takes the
`(num_buckets, H)` T5 table itself (what the reference's external `fa2_rpe` backend takes,
modeling_flash_t5.py:275-279); neither the `(1,H,M,N)` bias nor its gradient is ever materialised.
"""
import ctypes
import math
from typing import Optional, Tuple

import torch

from . import _lib
from . import positional_encoding as _pe

__all__ = ["flash_attention_v2_bias", "FlashAttentionAdditiveBias", "flash_attention_v2_rpe",
           "FlashAttentionRPE", "flash_attn_varlen_fwd"]


def _prep(t):
    return t if _lib.kernel_ready(t) else t.contiguous()


def _bias_strides(bias, B, H):
    # broadcast (1|B, 1|H, M, N) via zero strides (reference :45-52)
    sb = bias.stride(0) if bias.shape[0] == B and B != 1 else (0 if bias.shape[0] == 1 else bias.stride(0))
    sh = bias.stride(1) if bias.shape[1] == H and H != 1 else (0 if bias.shape[1] == 1 else bias.stride(1))
    if bias.shape[0] == 1:
        sb = 0
    if bias.shape[1] == 1:
        sh = 0
    return _lib.c_i64x3(sb, sh, bias.stride(2))


def _base_params(q, k, v, causal, sm_scale):
    B, H, M, D = q.shape
    N = k.shape[2]
    p = _lib.AttnParams()
    p.B, p.H, p.M, p.N, p.D = B, H, M, N, D
    p.dtype = _lib.dtype_code(q.dtype)
    p.causal = int(bool(causal))
    p.sm_scale = float(sm_scale)
    p.q, p.k, p.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
    p.q_stride, p.k_stride, p.v_stride = _lib.strides3(q), _lib.strides3(k), _lib.strides3(v)
    p.variant = _lib._variant
    return p


def _check_inputs(q, k, v):
    if not (q.is_cuda and k.is_cuda and v.is_cuda):
        raise RuntimeError("flasht5_amd attention needs tensors on the HIP device (no CPU fallback)")
    if q.dtype not in (torch.float16, torch.bfloat16) or k.dtype != q.dtype or v.dtype != q.dtype:
        raise TypeError("q, k, v must share dtype float16 or bfloat16")
    if q.dim() != 4 or k.dim() != 4 or v.dim() != 4:
        raise ValueError("q, k, v must be (B, H, S, D)")
    if k.shape != v.shape or k.shape[:2] != q.shape[:2] or k.shape[3] != q.shape[3] or k.device != q.device or v.device != q.device:
        raise ValueError(f"q {tuple(q.shape)}, k {tuple(k.shape)}, v {tuple(v.shape)}: batch, heads, head_dim and device must agree")


def _check_bias(bias, q, k):
    """dense bias (B|1, H|1, M, N) in q's dtype on q's device (the kernels trust these: reference :45-52 broadcasts the same way)"""
    B, H, M, _ = q.shape
    N = k.shape[2]
    if bias.dim() != 4 or bias.shape[0] not in (1, B) or bias.shape[1] not in (1, H) or tuple(bias.shape[2:]) != (M, N):
        raise ValueError(f"bias must be (1|{B}, 1|{H}, {M}, {N}), got {tuple(bias.shape)}")
    if bias.dtype != q.dtype:
        raise TypeError("bias must have the dtype of q")
    if bias.device != q.device:
        raise ValueError("bias must live on q's device")


def _check_rpe1d(rpe1d, H, radius, device):
    if rpe1d.dtype != torch.float32 or not rpe1d.is_contiguous() or rpe1d.device != device:
        raise TypeError("rpe1d must be a contiguous float32 tensor on q's device")
    if tuple(rpe1d.shape) != (H, 2 * int(radius) + 1) or radius < 1:
        raise ValueError(f"rpe1d must be (n_heads, 2 * radius + 1) = ({H}, {2 * int(radius) + 1}), got {tuple(rpe1d.shape)}")


# Backward scratch: one growing buffer per (device, stream).  Eager launches on one stream are ordered, so consecutive backward
# calls can share it; a call on another stream gets its own.  (A torch.empty per call cost ~4 us of the ~14 us host time of
# an eager call and made the caching allocator the busiest part of a 12 us kernel's launch.)
# NOT while the stream is being captured into a HIP graph (torch.cuda.graph, torch.compile(mode="reduce-overhead")): a cached
# buffer baked into a graph would be freed when a later, larger eager call regrows the cache, and the graph's replays would
# write into memory that belongs to someone else -- there the buffer comes from the allocator (the graph's private pool).
_WS_CACHE = {}


def _workspace(nbytes, device):
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device()))
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _WS_CACHE[key] = ws
    return ws


def _attn_fwd(q, k, v, bias, rpe1d, radius, causal, sm_scale):
    _check_inputs(q, k, v)
    q, k, v = _prep(q), _prep(k), _prep(v)
    B, H, M, D = q.shape
    o = torch.empty_like(q)  # reference :58
    if not _lib.kernel_ready(o):
        o = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    L = torch.empty((B, H, M), device=q.device, dtype=torch.float32)  # reference :59
    p = _base_params(q, k, v, causal, sm_scale)
    p.o, p.lse, p.o_stride = o.data_ptr(), L.data_ptr(), _lib.strides3(o)
    if bias is not None:
        _check_bias(bias, q, k)
        if bias.stride(-1) != 1:
            bias = bias.contiguous()
        p.bias_mode, p.bias, p.bias_stride = _lib.BIAS_DENSE, bias.data_ptr(), _bias_strides(bias, B, H)
    elif rpe1d is not None:
        _check_rpe1d(rpe1d, H, radius, q.device)
        p.bias_mode, p.rpe1d, p.rpe_radius = _lib.BIAS_RPE1D, rpe1d.data_ptr(), radius
    with _lib.on_device(q.device):
        _lib.check(_lib.load().fat5_attn_fwd(ctypes.byref(p), _lib.stream_ptr(q.device)), "fat5_attn_fwd")
    return o, L


def packed_slices(ts):
    """True when the (B, H, S, D) tensors `ts` are the slices [:, :, i] of ONE (B, S, n, H, D) projection output (n = len(ts): q, k, v
    of a self-attention block from one GEMM; k, v of a cross-attention block).  Their gradients are then allocated as the same
    slices of one (B, S, n, H, D) buffer -- the kernels write them in place, the projection's backward GEMMs take the buffer as it
    is: no zero-filled full-size tensor per slice, no strided copies, no additions (attention_module.unpack_heads)."""
    a, n = ts[0], len(ts)
    _, H, S, D = a.shape
    row = n * H * D
    if a.stride() != (S * row, D, row, 1):
        return False
    return all(t.shape == a.shape and t.stride() == a.stride() and t.dtype == a.dtype and
               t.data_ptr() - a.data_ptr() == i * H * D * a.element_size() for i, t in enumerate(ts))


def empty_packed_like(ts):
    a, n = ts[0], len(ts)
    B, H, S, D = a.shape
    base = torch.empty((B, S, n, H, D), dtype=a.dtype, device=a.device)
    return tuple(base[:, :, i].permute(0, 2, 1, 3) for i in range(n))


def _attn_bwd(o, do, q, k, v, bias, rpe1d, radius, L, causal, sm_scale, need_dbias, rpe_bucket=None, num_buckets=0):
    q, k, v, o, do = _prep(q), _prep(k), _prep(v), _prep(o), _prep(do)
    B, H, M, D = q.shape
    N = k.shape[2]
    def like(t):  # reference :140-141,:191
        e = torch.empty_like(t)
        return e if _lib.kernel_ready(e) else torch.empty(t.shape, dtype=t.dtype, device=t.device)
    # gradients of PACKED projections are allocated as slices of one buffer (see packed_slices)
    if q.shape == k.shape and packed_slices((q, k, v)):
        dq, dk, dv = empty_packed_like((q, k, v))
    elif packed_slices((k, v)):
        dq, (dk, dv) = like(q), empty_packed_like((k, v))
    else:
        dq, dk, dv = like(q), like(k), like(v)
    p = _base_params(q, k, v, causal, sm_scale)
    p.o, p.lse, p.o_stride = o.data_ptr(), L.data_ptr(), _lib.strides3(o)
    p.dout, p.dq, p.dk, p.dv = do.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    p.do_stride, p.dq_stride, p.dk_stride, p.dv_stride = (_lib.strides3(t) for t in (do, dq, dk, dv))
    dbias = None
    if bias is not None:
        _check_bias(bias, q, k)
        if bias.stride(-1) != 1:
            bias = bias.contiguous()
        p.bias_mode, p.bias, p.bias_stride = _lib.BIAS_DENSE, bias.data_ptr(), _bias_strides(bias, B, H)
        if need_dbias:
            dbias = torch.empty(bias.shape, dtype=bias.dtype, device=bias.device)  # shape/dtype of bias (:149,:224)
            p.dbias, p.dbias_batch, p.dbias_heads = dbias.data_ptr(), bias.shape[0], bias.shape[1]
    elif rpe1d is not None:
        _check_rpe1d(rpe1d, H, radius, q.device)
        p.bias_mode, p.rpe1d, p.rpe_radius = _lib.BIAS_RPE1D, rpe1d.data_ptr(), radius
        if need_dbias:
            if rpe_bucket is not None:  # table gradient straight from the reduction launch
                dbias = torch.empty((num_buckets, H), dtype=torch.float32, device=q.device)
                p.rpe_bucket, p.drpe_table, p.rpe_num_buckets = rpe_bucket.data_ptr(), dbias.data_ptr(), num_buckets
            else:
                dbias = torch.empty_like(rpe1d)
                p.drpe1d = dbias.data_ptr()
    lib = _lib.load()
    with _lib.on_device(q.device):
        ws = _workspace(lib.fat5_attn_bwd_workspace_bytes(ctypes.byref(p)), q.device)
        p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel()
        _lib.check(lib.fat5_attn_bwd(ctypes.byref(p), _lib.stream_ptr(q.device)), "fat5_attn_bwd")
    return dq, dk, dv, dbias


# ------------------------------------------------------------------------------------------------
# custom ops (own namespace; optionals declared `Tensor?` -- SURVEY Q8) with fake impls so that
# FakeTensor / torch.compile tracing works like for the reference's flasht5::* ops.
# ------------------------------------------------------------------------------------------------
@torch.library.custom_op("fat5::flash_attn_v2_fwd", mutates_args=(), device_types="cuda")
def flash_attn_v2_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, bias: Optional[torch.Tensor],
                      causal: bool, sm_scale: float) -> Tuple[torch.Tensor, torch.Tensor]:
    return _attn_fwd(q, k, v, bias, None, 0, causal, sm_scale)


@torch.library.register_fake("fat5::flash_attn_v2_fwd")
def _flash_attn_v2_fwd_fake(q, k, v, bias, causal, sm_scale):
    B, H, M, D = q.shape
    return torch.empty_like(q), torch.empty((B, H, M), dtype=torch.float32, device=q.device)


@torch.library.custom_op("fat5::flash_attn_v2_bwd", mutates_args=(), device_types="cuda")
def flash_attn_v2_bwd(o: torch.Tensor, do: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                      bias: Optional[torch.Tensor], L: torch.Tensor, causal: bool, sm_scale: float
                      ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    dq, dk, dv, ds = _attn_bwd(o, do, q, k, v, bias, None, 0, L, causal, sm_scale, bias is not None)
    if ds is None:
        ds = torch.empty(0, dtype=q.dtype, device=q.device)
    return dq, dk, dv, ds


@torch.library.register_fake("fat5::flash_attn_v2_bwd")
def _flash_attn_v2_bwd_fake(o, do, q, k, v, bias, L, causal, sm_scale):
    ds = torch.empty_like(bias) if bias is not None else torch.empty(0, dtype=q.dtype, device=q.device)
    return torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), ds


def _tracing():
    """a compiler / FakeTensor trace is recording: go through the registered custom ops (they carry the fake impls); plain
    eager calls skip the dispatcher round trip (~10 us of host time per op, more than the S = 512 kernels take)"""
    return torch.compiler.is_compiling()


class FlashAttentionAdditiveBias(torch.autograd.Function):
    """Same contract as the reference class (flash_attention_v2_bias.py:228-271)."""

    @staticmethod
    def forward(ctx, q, k, v, bias, causal, sm_scale):
        Dq, Dk, Dv = q.shape[-1], k.shape[-1], v.shape[-1]
        assert Dq == Dk == Dv
        assert Dk in {16, 32, 64, 128}
        if sm_scale is None:
            sm_scale = 1.0 / math.sqrt(Dq)
        if _tracing():
            o, L = torch.ops.fat5.flash_attn_v2_fwd(q, k, v, bias, bool(causal), float(sm_scale))
        else:
            o, L = _attn_fwd(q, k, v, bias, None, 0, bool(causal), float(sm_scale))
        ctx.save_for_backward(q, k, v, bias, o, L)
        ctx.sm_scale = sm_scale
        ctx.causal = causal
        return o

    @staticmethod
    def backward(ctx, do, *ignored):
        q, k, v, bias, o, L = ctx.saved_tensors
        if _tracing():
            dq, dk, dv, ds = torch.ops.fat5.flash_attn_v2_bwd(o, do, q, k, v, bias, L, bool(ctx.causal), float(ctx.sm_scale))
        else:
            dq, dk, dv, ds = _attn_bwd(o, do, q, k, v, bias, None, 0, L, bool(ctx.causal), float(ctx.sm_scale), bias is not None)
        return dq, dk, dv, (ds if bias is not None else None), None, None, None, None


def flash_attention_v2_bias(q, k, v, bias, causal=False, sm_scale=None):
    """FlashAttention-2 forward/backward with additive bias (reference :274-288).

    q: (B, H, M, D); k, v: (B, H, N, D) (strided views are fine); bias: (B|1, H|1, M, N) or None.
    Differentiable in q, k, v and bias.  Returns o: (B, H, M, D)."""
    nat = None if _tracing() else _lib.native()
    if nat is None:
        return FlashAttentionAdditiveBias.apply(q, k, v, bias, causal, sm_scale)
    # eager: the same checks, then the C++ autograd function (csrc/torch_binding.cpp) straight onto the C ABI
    D = q.shape[-1]
    assert D == k.shape[-1] == v.shape[-1]
    assert D in {16, 32, 64, 128}
    _check_inputs(q, k, v)
    if bias is not None:
        _check_bias(bias, q, k)
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)
    o = nat.bias_apply(q, k, v, bias, bool(causal), float(sm_scale))
    return o


# ------------------------------------------------------------------------------------------------
# linear-memory T5 RPE mode
# ------------------------------------------------------------------------------------------------
@torch.library.custom_op("fat5::flash_attn_rpe1d_fwd", mutates_args=(), device_types="cuda")
def flash_attn_rpe1d_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, rpe1d: torch.Tensor, radius: int, causal: bool,
                         sm_scale: float) -> Tuple[torch.Tensor, torch.Tensor]:
    return _attn_fwd(q, k, v, None, rpe1d, int(radius), causal, sm_scale)


@torch.library.register_fake("fat5::flash_attn_rpe1d_fwd")
def _flash_attn_rpe1d_fwd_fake(q, k, v, rpe1d, radius, causal, sm_scale):
    B, H, M, D = q.shape
    return torch.empty_like(q), torch.empty((B, H, M), dtype=torch.float32, device=q.device)


@torch.library.custom_op("fat5::flash_attn_rpe1d_bwd", mutates_args=(), device_types="cuda")
def flash_attn_rpe1d_bwd(o: torch.Tensor, do: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, rpe1d: torch.Tensor,
                         L: torch.Tensor, radius: int, causal: bool, sm_scale: float, need_drpe: bool,
                         rpe_bucket: Optional[torch.Tensor], num_buckets: int
                         ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """dq, dk, dv and the bias gradient: (H, 2R+1) diagonal sums, or -- with `rpe_bucket` -- the (num_buckets, H) table
    gradient scattered by the same reduction launch; empty when `need_drpe` is False"""
    dq, dk, dv, d1 = _attn_bwd(o, do, q, k, v, None, rpe1d, int(radius), L, causal, sm_scale, need_drpe, rpe_bucket, int(num_buckets))
    if d1 is None:
        d1 = torch.empty(0, dtype=torch.float32, device=q.device)
    return dq, dk, dv, d1


@torch.library.register_fake("fat5::flash_attn_rpe1d_bwd")
def _flash_attn_rpe1d_bwd_fake(o, do, q, k, v, rpe1d, L, radius, causal, sm_scale, need_drpe, rpe_bucket, num_buckets):
    if not need_drpe:
        d1 = torch.empty(0, dtype=torch.float32, device=q.device)
    elif rpe_bucket is not None:
        d1 = torch.empty((num_buckets, q.shape[1]), dtype=torch.float32, device=q.device)
    else:
        d1 = torch.empty_like(rpe1d)
    return torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), d1


def _rpe_fwd(q, k, v, r1, radius, causal, sm_scale):
    if _tracing():
        return torch.ops.fat5.flash_attn_rpe1d_fwd(q, k, v, r1, int(radius), bool(causal), float(sm_scale))
    return _attn_fwd(q, k, v, None, r1, int(radius), bool(causal), float(sm_scale))


def _rpe_bwd(o, do, q, k, v, r1, L, radius, causal, sm_scale, need, bucket=None, num_buckets=0):
    if _tracing():
        dq, dk, dv, d1 = torch.ops.fat5.flash_attn_rpe1d_bwd(o, do, q, k, v, r1, L, int(radius), bool(causal), float(sm_scale),
                                                             bool(need), bucket, int(num_buckets))
        return dq, dk, dv, (d1 if need else None)
    return _attn_bwd(o, do, q, k, v, None, r1, int(radius), L, bool(causal), float(sm_scale), bool(need), bucket, int(num_buckets))


def _rpe1d_of(rpe_table, R, bidirectional, num_buckets, max_distance):
    """(H, 2R+1) fp32 generator of the table: ONE launch (fat5_rpe1d_from_table) on every call.  Deliberately not cached across
    calls: an autograd version counter does not see every update of the table (this package's fused AdamWScale writes parameters
    through raw pointers; `p.data = ...` swaps), and a stale generator would train silently wrong.  A layer stack that wants to
    build it once per step does so explicitly: `RelativePositionalEncoding.forward_1d()` + `flash_attention_v2_rpe1d`."""
    if _tracing():  # torch ops a compiler can trace
        idx = _pe.bucket_index(R, bidirectional, num_buckets, max_distance, rpe_table.device)
        return rpe_table.detach().index_select(0, idx).transpose(0, 1).float().contiguous()
    t = rpe_table.detach()
    if t.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        t = t.float()
    t = t.contiguous()
    H = t.shape[1]
    idx32 = _pe.bucket_index32(R, bidirectional, num_buckets, max_distance, t.device)
    r1 = torch.empty((H, 2 * R + 1), dtype=torch.float32, device=t.device)
    with _lib.on_device(t.device):
        _lib.check(_lib.load().fat5_rpe1d_from_table(t.data_ptr(), _lib.dtype_code(t.dtype), idx32.data_ptr(), r1.data_ptr(), H, R,
                                                     int(num_buckets), _lib.stream_ptr(t.device)), "fat5_rpe1d_from_table")
    return r1


class FlashAttentionRPE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, rpe_table, bidirectional, num_buckets, max_distance, causal, sm_scale):
        D = q.shape[-1]
        assert D in {16, 32, 64, 128}
        if sm_scale is None:
            sm_scale = 1.0 / math.sqrt(D)
        R = _pe.rpe_radius(max_distance)
        if R > _lib.MAX_RPE_RADIUS:
            raise ValueError(f"max_distance {max_distance} exceeds the RPE-mode limit {_lib.MAX_RPE_RADIUS}; use the dense bias")
        if rpe_table.shape != (num_buckets, q.shape[1]):
            raise ValueError("rpe_table must be (num_buckets, n_heads)")
        rpe1d = _rpe1d_of(rpe_table, R, bidirectional, num_buckets, max_distance)
        o, L = _rpe_fwd(q, k, v, rpe1d, R, causal, sm_scale)
        ctx.save_for_backward(q, k, v, o, L, rpe1d, _pe.bucket_index32(R, bidirectional, num_buckets, max_distance, q.device))
        ctx.meta = (R, causal, sm_scale, num_buckets, rpe_table.dtype)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, L, rpe1d, idx = ctx.saved_tensors
        R, causal, sm_scale, num_buckets, tdtype = ctx.meta
        need = ctx.needs_input_grad[3]
        # the (H, 2R+1) diagonal sums are scattered into the (num_buckets, H) table by the reduction launch itself
        dq, dk, dv, dtable = _rpe_bwd(o, do, q, k, v, rpe1d, L, R, causal, sm_scale, need, idx, num_buckets)
        if dtable is not None:
            dtable = dtable.to(tdtype)
        return dq, dk, dv, dtable, None, None, None, None, None


def flash_attention_v2_rpe(q, k, v, rpe_table, bidirectional=True, num_buckets=32, max_distance=128,
                           causal=False, sm_scale=None):
    """Attention with the T5 relative-position bias generated in-kernel from the `(num_buckets, H)` table
    (`RelativePositionalEncoding.relative_attention_bias.weight` of the reference).  Equivalent to
    `flash_attention_v2_bias(q, k, v, compute_bias(table, M, N), ...)` with O(S) memory; differentiable in
    q, k, v and the table."""
    nat = None if _tracing() else _lib.native()
    if nat is None:
        return FlashAttentionRPE.apply(q, k, v, rpe_table, bidirectional, num_buckets, max_distance, causal, sm_scale)
    D = q.shape[-1]
    assert D in {16, 32, 64, 128}
    _check_inputs(q, k, v)
    R = _pe.rpe_radius(max_distance)
    if R > _lib.MAX_RPE_RADIUS:
        raise ValueError(f"max_distance {max_distance} exceeds the RPE-mode limit {_lib.MAX_RPE_RADIUS}; use the dense bias")
    if rpe_table.shape != (num_buckets, q.shape[1]):
        raise ValueError("rpe_table must be (num_buckets, n_heads)")
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)
    if rpe_table.device != q.device:
        raise ValueError("rpe_table must live on q's device")
    idx = _pe.bucket_index32(R, bidirectional, num_buckets, max_distance, q.device)
    # (the C++ function builds the (H, 2R+1) generator from the table itself, one launch per call, never cached)
    o = nat.rpe_table_apply(q, k, v, rpe_table, idx, R, int(num_buckets), bool(causal), float(sm_scale))
    return o


class FlashAttentionRPE1D(torch.autograd.Function):
    """RPE mode on the 1-D generator itself: `rpe1d (H, 2R+1)` fp32 with bias[h][m][n] = rpe1d[h][clamp(n-m,-R,R)+R].
    Differentiable in q, k, v and rpe1d, so ONE generator (built once per step from the T5 table by
    `rpe1d_from_table` / `RelativePositionalEncoding.forward_1d`) can feed every layer: autograd sums the per-layer
    `(H, 2R+1)` gradients and scatters them into the `(num_buckets, H)` table once."""

    @staticmethod
    def forward(ctx, q, k, v, rpe1d, radius, causal, sm_scale):
        D = q.shape[-1]
        assert D in {16, 32, 64, 128}
        if sm_scale is None:
            sm_scale = 1.0 / math.sqrt(D)
        if radius > _lib.MAX_RPE_RADIUS:
            raise ValueError(f"radius {radius} exceeds the RPE-mode limit {_lib.MAX_RPE_RADIUS}; use the dense bias")
        if rpe1d.shape != (q.shape[1], 2 * radius + 1):
            raise ValueError("rpe1d must be (n_heads, 2 * radius + 1)")
        r1 = rpe1d.detach().float().contiguous()
        o, L = _rpe_fwd(q, k, v, r1, radius, causal, sm_scale)
        ctx.save_for_backward(q, k, v, o, L, r1)
        ctx.meta = (radius, causal, sm_scale, rpe1d.dtype)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, L, r1 = ctx.saved_tensors
        radius, causal, sm_scale, rdtype = ctx.meta
        dq, dk, dv, d1 = _rpe_bwd(o, do, q, k, v, r1, L, radius, causal, sm_scale, ctx.needs_input_grad[3])
        return dq, dk, dv, (d1.to(rdtype) if d1 is not None else None), None, None, None


def flash_attention_v2_rpe1d(q, k, v, rpe1d, radius, causal=False, sm_scale=None):
    """Attention with the Toeplitz bias generated in-kernel from `rpe1d (H, 2*radius+1)` (see FlashAttentionRPE1D)."""
    nat = None if _tracing() else _lib.native()
    if nat is None:
        return FlashAttentionRPE1D.apply(q, k, v, rpe1d, int(radius), causal, sm_scale)
    D = q.shape[-1]
    radius = int(radius)
    assert D in {16, 32, 64, 128}
    _check_inputs(q, k, v)
    if radius > _lib.MAX_RPE_RADIUS:
        raise ValueError(f"radius {radius} exceeds the RPE-mode limit {_lib.MAX_RPE_RADIUS}; use the dense bias")
    if rpe1d.shape != (q.shape[1], 2 * radius + 1):
        raise ValueError("rpe1d must be (n_heads, 2 * radius + 1)")
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)
    r1 = rpe1d.detach().float().contiguous()
    _check_rpe1d(r1, q.shape[1], radius, q.device)
    o = nat.rpe1d_apply(q, k, v, rpe1d, r1, radius, bool(causal), float(sm_scale))
    return o


# ------------------------------------------------------------------------------------------------
# packed var-len attention (config 4: decoder cross-attention over cu_seqlens), forward + backward
# ------------------------------------------------------------------------------------------------
def _varlen_ok(t):
    return t.stride(-1) == 1 and t.data_ptr() % 16 == 0 and t.stride(0) % 8 == 0 and t.stride(1) % 8 == 0


def _check_varlen(q, k, v, cu_q, cu_k, rpe1d, radius):
    if not (q.is_cuda and k.is_cuda and v.is_cuda):
        raise RuntimeError("flasht5_amd attention needs tensors on the HIP device (no CPU fallback)")
    if q.dtype not in (torch.float16, torch.bfloat16) or k.dtype != q.dtype or v.dtype != q.dtype:
        raise TypeError("q, k, v must share dtype float16 or bfloat16")
    if q.dim() != 3 or k.dim() != 3 or v.dim() != 3 or k.shape != v.shape or k.shape[1:] != q.shape[1:]:
        raise ValueError("packed attention takes q (total_q, H, D) and k, v (total_k, H, D)")
    if q.shape[-1] not in (16, 32, 64, 128):
        raise ValueError("head_dim must be 16, 32, 64 or 128")
    for name, cu in (("cu_seqlens_q", cu_q), ("cu_seqlens_k", cu_k)):
        # the kernels dereference these on the device
        if cu.device != q.device or cu.dtype != torch.int32 or cu.dim() != 1 or not cu.is_contiguous():
            raise TypeError(f"{name} must be a contiguous int32 vector on q's device")
    if cu_q.numel() != cu_k.numel() or cu_q.numel() < 2:
        raise ValueError("cu_seqlens_q and cu_seqlens_k must both have (number of sequences + 1) entries")
    if rpe1d is not None:
        _check_rpe1d(rpe1d, q.shape[1], radius, q.device)


def _varlen_params(q, k, v, cu_q, cu_k, max_q, max_k, causal, sm_scale):
    Tq, H, D = q.shape
    p = _lib.AttnParams()
    p.B, p.H, p.M, p.N, p.D = cu_q.numel() - 1, H, int(max_q), int(max_k), D
    p.dtype, p.causal, p.sm_scale = _lib.dtype_code(q.dtype), int(bool(causal)), float(sm_scale)
    p.q, p.k, p.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
    for name, t in (("q_stride", q), ("k_stride", k), ("v_stride", v)):
        setattr(p, name, _lib.c_i64x3(0, t.stride(1), t.stride(0)))
    p.cu_seqlens_q, p.cu_seqlens_k, p.total_q, p.total_k = cu_q.data_ptr(), cu_k.data_ptr(), Tq, k.shape[0]
    p.variant = _lib._variant
    return p


def _as_cu(cu, device):
    return cu if (cu.dtype == torch.int32 and cu.device == device and cu.is_contiguous()) else cu.to(device=device, dtype=torch.int32).contiguous()


def _varlen_fwd_impl(q, k, v, cu_q, cu_k, max_q, max_k, causal, sm_scale, rpe1d, radius):
    _check_varlen(q, k, v, cu_q, cu_k, rpe1d, radius)
    q, k, v = (t if _varlen_ok(t) else t.contiguous() for t in (q, k, v))
    Tq, H, D = q.shape
    o = torch.empty((Tq, H, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((H, Tq), dtype=torch.float32, device=q.device)
    p = _varlen_params(q, k, v, cu_q, cu_k, max_q, max_k, causal, sm_scale)
    p.o, p.lse = o.data_ptr(), lse.data_ptr()
    p.o_stride = _lib.c_i64x3(0, o.stride(1), o.stride(0))
    if rpe1d is not None:
        p.bias_mode, p.rpe1d, p.rpe_radius = _lib.BIAS_RPE1D, rpe1d.data_ptr(), int(radius)
    with _lib.on_device(q.device):
        _lib.check(_lib.load().fat5_attn_fwd(ctypes.byref(p), _lib.stream_ptr(q.device)), "fat5_attn_fwd(varlen)")
    return o, lse


def _varlen_bwd_impl(do, q, k, v, o, lse, cu_q, cu_k, max_q, max_k, causal, sm_scale, rpe1d, radius, need_drpe):
    _check_varlen(q, k, v, cu_q, cu_k, rpe1d, radius)
    q, k, v, o, do = (t if _varlen_ok(t) else t.contiguous() for t in (q, k, v, o, do))
    dq, dk, dv = (torch.empty(t.shape, dtype=t.dtype, device=t.device) for t in (q, k, v))
    p = _varlen_params(q, k, v, cu_q, cu_k, max_q, max_k, causal, sm_scale)
    p.o, p.lse, p.dout, p.dq, p.dk, p.dv = (t.data_ptr() for t in (o, lse, do, dq, dk, dv))
    for name, t in (("o_stride", o), ("do_stride", do), ("dq_stride", dq), ("dk_stride", dk), ("dv_stride", dv)):
        setattr(p, name, _lib.c_i64x3(0, t.stride(1), t.stride(0)))
    drpe = None
    if rpe1d is not None:
        p.bias_mode, p.rpe1d, p.rpe_radius = _lib.BIAS_RPE1D, rpe1d.data_ptr(), int(radius)
        if need_drpe:
            drpe = torch.empty_like(rpe1d)
            p.drpe1d = drpe.data_ptr()
    lib = _lib.load()
    with _lib.on_device(q.device):
        ws = _workspace(lib.fat5_attn_bwd_workspace_bytes(ctypes.byref(p)), q.device)
        p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel()
        _lib.check(lib.fat5_attn_bwd(ctypes.byref(p), _lib.stream_ptr(q.device)), "fat5_attn_bwd(varlen)")
    return dq, dk, dv, drpe


@torch.library.custom_op("fat5::flash_attn_varlen_fwd", mutates_args=(), device_types="cuda")
def _varlen_fwd_op(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor,
                   max_seqlen_q: int, max_seqlen_k: int, causal: bool, sm_scale: float, rpe1d: Optional[torch.Tensor],
                   radius: int) -> Tuple[torch.Tensor, torch.Tensor]:
    return _varlen_fwd_impl(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal, sm_scale, rpe1d, radius)


@torch.library.register_fake("fat5::flash_attn_varlen_fwd")
def _varlen_fwd_fake(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal, sm_scale, rpe1d, radius):
    Tq, H, D = q.shape
    return torch.empty((Tq, H, D), dtype=q.dtype, device=q.device), torch.empty((H, Tq), dtype=torch.float32, device=q.device)


@torch.library.custom_op("fat5::flash_attn_varlen_bwd", mutates_args=(), device_types="cuda")
def _varlen_bwd_op(do: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor, lse: torch.Tensor,
                   cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor, max_seqlen_q: int, max_seqlen_k: int, causal: bool,
                   sm_scale: float, rpe1d: Optional[torch.Tensor], radius: int, need_drpe: bool
                   ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    dq, dk, dv, d1 = _varlen_bwd_impl(do, q, k, v, o, lse, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal,
                                      sm_scale, rpe1d, radius, need_drpe)
    if d1 is None:
        d1 = torch.empty(0, dtype=torch.float32, device=q.device)
    return dq, dk, dv, d1


@torch.library.register_fake("fat5::flash_attn_varlen_bwd")
def _varlen_bwd_fake(do, q, k, v, o, lse, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal, sm_scale, rpe1d, radius,
                     need_drpe):
    d1 = torch.empty_like(rpe1d) if (need_drpe and rpe1d is not None) else torch.empty(0, dtype=torch.float32, device=q.device)
    return torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), d1


def flash_attn_varlen_fwd(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal=False, sm_scale=None,
                          rpe1d=None, radius=0):
    """q: (total_q, H, D); k, v: (total_k, H, D); cu_seqlens_*: int32 (nseq+1,) on the device (other dtypes / devices are
    converted); optional T5 bias generator rpe1d (H, 2*radius+1) fp32 (positions count from each sequence's own start).
    Returns o (total_q, H, D) and lse (H, total_q)."""
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(q.shape[-1])
    cq, ck = _as_cu(cu_seqlens_q, q.device), _as_cu(cu_seqlens_k, q.device)
    if _tracing():
        return torch.ops.fat5.flash_attn_varlen_fwd(q, k, v, cq, ck, int(max_seqlen_q), int(max_seqlen_k), bool(causal), float(sm_scale),
                                                    rpe1d, int(radius))
    return _varlen_fwd_impl(q, k, v, cq, ck, int(max_seqlen_q), int(max_seqlen_k), bool(causal), float(sm_scale), rpe1d, int(radius))


def flash_attn_varlen_bwd(do, q, k, v, o, lse, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal=False,
                          sm_scale=None, rpe1d=None, radius=0, need_drpe=False):
    """Backward of the packed (cu_seqlens) attention: returns dq (total_q, H, D), dk, dv (total_k, H, D)
    (and drpe1d (H, 2*radius+1) or None when a generator is given)."""
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(q.shape[-1])
    cq, ck = _as_cu(cu_seqlens_q, q.device), _as_cu(cu_seqlens_k, q.device)
    if _tracing():
        dq, dk, dv, d1 = torch.ops.fat5.flash_attn_varlen_bwd(do, q, k, v, o, lse, cq, ck, int(max_seqlen_q), int(max_seqlen_k), bool(causal),
                                                              float(sm_scale), rpe1d, int(radius), bool(need_drpe))
        d1 = d1 if (need_drpe and rpe1d is not None) else None
    else:
        dq, dk, dv, d1 = _varlen_bwd_impl(do, q, k, v, o, lse, cq, ck, int(max_seqlen_q), int(max_seqlen_k), bool(causal),
                                          float(sm_scale), rpe1d, int(radius), bool(need_drpe))
    return (dq, dk, dv, d1) if rpe1d is not None else (dq, dk, dv)


class FlashAttentionVarlen(torch.autograd.Function):
    """Packed batches: q (total_q, H, D), k/v (total_k, H, D), int32 cu_seqlens on the device;
    optionally the T5 bias generator `rpe1d (H, 2*radius+1)` (differentiable), positions local to each sequence.
    (The reference has no cu_seqlens path at all: it pads, `data_collator_ul2.py:49-87`.)"""

    @staticmethod
    def forward(ctx, q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal, sm_scale, rpe1d, radius):
        r1 = rpe1d.detach().float().contiguous() if rpe1d is not None else None
        cq, ck = _as_cu(cu_seqlens_q, q.device), _as_cu(cu_seqlens_k, q.device)
        o, lse = flash_attn_varlen_fwd(q, k, v, cq, ck, max_seqlen_q, max_seqlen_k, causal, sm_scale, r1, radius)
        ctx.save_for_backward(q, k, v, o, lse, cq, ck, *([r1] if r1 is not None else []))
        ctx.meta = (int(max_seqlen_q), int(max_seqlen_k), bool(causal), sm_scale, int(radius),
                    rpe1d.dtype if rpe1d is not None else None)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, cq, ck, *rest = ctx.saved_tensors
        mq, mk, causal, sm_scale, radius, rdtype = ctx.meta
        if rest:
            dq, dk, dv, d1 = flash_attn_varlen_bwd(do, q, k, v, o, lse, cq, ck, mq, mk, causal, sm_scale, rest[0], radius,
                                                   ctx.needs_input_grad[9])
            return dq, dk, dv, None, None, None, None, None, None, (d1.to(rdtype) if d1 is not None else None), None
        dq, dk, dv = flash_attn_varlen_bwd(do, q, k, v, o, lse, cq, ck, mq, mk, causal, sm_scale)
        return dq, dk, dv, None, None, None, None, None, None, None, None


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal=False, sm_scale=None,
                           rpe1d=None, radius=0):
    """Differentiable packed attention (argument order of the flash_attn var-len entry point the reference's `fa2`
    attention types would call), optionally with the in-kernel T5 bias of `flash_attention_v2_rpe1d`."""
    return FlashAttentionVarlen.apply(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal, sm_scale,
                                      rpe1d, int(radius))


# ------------------------------------------------------------------------------------------------
# pre-planned fwd+bwd (fixed buffers, prebuilt C descriptors): what a training loop / hipGraph replays
# ------------------------------------------------------------------------------------------------
class AttentionPlan:
    """Fixed-shape forward+backward with all buffers preallocated and the `fat5_attn_params` descriptors
    built once: each call is one C-ABI call (fwd) or one per stage (bwd) -- capturable in a HIP graph.

    mode: "none" | "dense" (bias tensor) | "rpe" (rpe1d (H, 2R+1) fp32 + radius)."""

    def __init__(self, q, k, v, do, *, bias=None, rpe1d=None, radius=0, causal=False, sm_scale=None, need_dbias=True,
                 rpe_bucket=None, num_buckets=0, units=None, variant=None):
        """units = (begin, count): run only the head-major unit range u = h * B + b in [begin, begin + count) (a rank's shard,
        `flasht5_amd.sharding.unit_range`); the bias gradient then holds this range's partial sums."""
        _check_inputs(q, k, v)
        self.q, self.k, self.v, self.do = _prep(q), _prep(k), _prep(v), _prep(do)
        B, H, M, D = q.shape
        self.shape = (B, H, M, k.shape[2], D)
        sm_scale = (1.0 / math.sqrt(D)) if sm_scale is None else sm_scale
        dev = q.device
        self.o = torch.empty_like(self.q)
        self.lse = torch.empty((B, H, M), dtype=torch.float32, device=dev)
        self.dq, self.dk, self.dv = torch.empty_like(self.q), torch.empty_like(self.k), torch.empty_like(self.v)
        self.bias, self.rpe1d, self.dbias = bias, rpe1d, None
        p = _base_params(self.q, self.k, self.v, causal, sm_scale)
        p.o, p.lse, p.o_stride = self.o.data_ptr(), self.lse.data_ptr(), _lib.strides3(self.o)
        p.dout, p.dq, p.dk, p.dv = self.do.data_ptr(), self.dq.data_ptr(), self.dk.data_ptr(), self.dv.data_ptr()
        p.do_stride, p.dq_stride, p.dk_stride, p.dv_stride = (_lib.strides3(t) for t in (self.do, self.dq, self.dk, self.dv))
        if bias is not None:
            p.bias_mode, p.bias, p.bias_stride = _lib.BIAS_DENSE, bias.data_ptr(), _bias_strides(bias, B, H)
            if need_dbias:
                self.dbias = torch.empty_like(bias)
                p.dbias, p.dbias_batch, p.dbias_heads = self.dbias.data_ptr(), bias.shape[0], bias.shape[1]
        elif rpe1d is not None:
            p.bias_mode, p.rpe1d, p.rpe_radius = _lib.BIAS_RPE1D, rpe1d.data_ptr(), int(radius)
            if need_dbias and rpe_bucket is not None:  # (num_buckets, H) table gradient from the reduction launch
                self.rpe_bucket = rpe_bucket
                self.dbias = torch.empty((num_buckets, H), dtype=torch.float32, device=dev)
                p.rpe_bucket, p.drpe_table, p.rpe_num_buckets = rpe_bucket.data_ptr(), self.dbias.data_ptr(), num_buckets
            elif need_dbias:
                self.dbias = torch.empty_like(rpe1d)
                p.drpe1d = self.dbias.data_ptr()
        # an EMPTY shard (more ranks than units): the C ABI reads unit_count == 0 as "the whole problem", so such a plan launches
        # nothing at all -- its outputs stay untouched and its bias gradient is zero (the rank still joins the all-reduce)
        self.empty = units is not None and int(units[1]) == 0
        if units is not None:
            if int(units[1]) < 0 or int(units[0]) < 0 or int(units[0]) + int(units[1]) > B * H:
                raise ValueError(f"unit range {tuple(units)} outside the {B * H} (batch, head) units")
            p.unit_begin, p.unit_count = int(units[0]), int(units[1])
        if self.empty and self.dbias is not None:
            self.dbias.zero_()
        if variant is not None:  # tests / profilers: force or forbid a kernel body for this plan (include/fat5.h `enum fat5_variant`)
            p.variant = int(variant)
        self.lib = _lib.load()
        nbytes = self.lib.fat5_attn_bwd_workspace_bytes(ctypes.byref(p))
        self.ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        p.workspace, p.workspace_bytes = self.ws.data_ptr(), self.ws.numel()
        self.p = p
        self.device = dev

    def set_variant(self, bits):
        """tests / profilers: kernel-variant bits of this plan's calls; the workspace is re-sized for the new choice"""
        self.p.variant = int(bits)
        nbytes = self.lib.fat5_attn_bwd_workspace_bytes(ctypes.byref(self.p))
        if nbytes > self.ws.numel():
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self.p.workspace, self.p.workspace_bytes = self.ws.data_ptr(), self.ws.numel()

    def forward(self):
        if self.empty:
            return self.o
        _lib.check(self.lib.fat5_attn_fwd(ctypes.byref(self.p), _lib.stream_ptr(self.device)), "fat5_attn_fwd")
        return self.o

    def describe(self):
        """the kernel bodies this plan's calls run, e.g. {"fwd": "64row-ksplit", "dq": "64row", "dkdv": "64key", "fused": "1", ...} (fat5_attn_describe)"""
        buf = ctypes.create_string_buffer(256)
        _lib.check(self.lib.fat5_attn_describe(ctypes.byref(self.p), buf, 256), "fat5_attn_describe")
        return dict(kv.split("=") for kv in buf.value.decode().split())

    def bwd_launches(self):
        """1: dQ and dK/dV halves share one launch (short sequences); 2: two kernels (fat5_attn_bwd_launches)."""
        return int(self.lib.fat5_attn_bwd_launches(ctypes.byref(self.p)))

    def backward(self, stages=7):
        if self.empty:
            return self.dq, self.dk, self.dv, self.dbias
        _lib.check(self.lib.fat5_attn_bwd_stages(ctypes.byref(self.p), int(stages), _lib.stream_ptr(self.device)),
                   "fat5_attn_bwd")
        return self.dq, self.dk, self.dv, self.dbias
