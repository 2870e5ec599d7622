"""A `FlashT5Attention`-compatible module on top of the MI355X operators.

Mirrors the interface of the reference module (src/model/modeling_flash_t5.py:166-287): same constructor signature
`(config, has_positional_encoding=False, is_causal=False)`, same parameter names (`Wq`, `Wk`, `Wv`, `o`,
`pe_encoding.relative_attention_bias`) so reference checkpoints load, same `forward(hidden_states, mask=None,
key_value_states=None, position_bias=None) -> (output, position_bias)` with the position bias of block 0 handed on to
the following blocks.  Two attention types:

  * "triton"   -- the reference's name for its `flash_attention_v2_bias` path: dense `(1|B, H, M, N)` bias, here
                  through `flasht5_amd.flash_attention_v2_bias` (drop-in);
  * "fat5_rpe" -- linear memory: the handed-on "position bias" is the `(H, 2R+1)` generator (with its radius), built
                  once by block 0 and consumed in-kernel by every block (`flash_attention_v2_rpe1d`); nothing of size
                  S x S is allocated in either direction (the role of the reference's external "fa2_rpe" type, :272-279).

Only the T5 relative-position producer lives here; other producers' dense outputs (ALiBi, FIRE, ...) can be passed as
`position_bias`.  Dropout is not supported (like the reference's Triton path, :201)."""
import math

import torch
from torch import nn

from .flash_attention_v2_bias import flash_attention_v2_bias, flash_attention_v2_rpe1d
from .positional_encoding import RelativePositionalEncoding


def _cfg(config, name, default):
    return getattr(config, name, default)


class _UnpackHeads(torch.autograd.Function):
    """(B, S, n * H * D) projection output -> n views (B, H, S, D) (slice i = heads of projection i).  Same values as
    `x.view(B, S, n, H, D)[:, :, i].permute(0, 2, 1, 3)`; the difference is the backward: autograd's select would allocate a
    zero-filled (B, S, n, H, D) tensor per slice, copy the slice's gradient in and add the n tensors up.  The attention
    backward writes the gradients of packed slices into the slices of ONE buffer (flash_attention_v2_bias.packed_slices): when
    the incoming gradients are exactly those, the buffer itself is the result (no kernel at all); otherwise one stack."""

    @staticmethod
    def forward(ctx, x, n, H):
        B, S, W = x.shape
        D = W // (n * H)
        ctx.dims = (B, S, n, H, D)
        p = x.view(B, S, n, H, D)
        return tuple(p[:, :, i].permute(0, 2, 1, 3) for i in range(n))

    @staticmethod
    def backward(ctx, *gs):
        B, S, n, H, D = ctx.dims
        row = n * H * D
        g0 = gs[0]
        if all(g is not None for g in gs):
            same = all(g.untyped_storage().data_ptr() == g0.untyped_storage().data_ptr() and g.dtype == g0.dtype and
                       g.stride() == (S * row, D, row, 1) and g.storage_offset() == g0.storage_offset() + i * H * D
                       for i, g in enumerate(gs))
            if same and g0.untyped_storage().nbytes() >= (g0.storage_offset() + B * S * row) * g0.element_size():
                return g0.as_strided((B, S, row), (S * row, row, 1), g0.storage_offset()), None, None
        ref = next(g for g in gs if g is not None)
        parts = [(g if g is not None else torch.zeros_like(ref)).permute(0, 2, 1, 3) for g in gs]  # (B, S, H, D) each
        return torch.stack(parts, 2).reshape(B, S, row), None, None


def unpack_heads(x, n, n_heads):
    """the n head-major (B, H, S, D) views of a packed (B, S, n * H * D) projection output (see _UnpackHeads)"""
    from . import _lib
    nat = None if (torch.compiler.is_compiling() or not x.is_cuda) else _lib.native()
    if nat is not None:  # (the same function in C++: csrc/torch_binding.cpp::UnpackHeadsFn)
        return tuple(nat.unpack_heads_apply(x, int(n), int(n_heads)))
    return _UnpackHeads.apply(x, n, n_heads)


class FlashT5Attention(nn.Module):
    def __init__(self, config, has_positional_encoding=False, is_causal=False):
        super().__init__()
        self.is_decoder = _cfg(config, "is_decoder", False)
        self.has_positional_encoding = has_positional_encoding
        self.is_causal = is_causal
        self.d_model = config.d_model
        self.key_value_proj_dim = config.d_kv
        self.n_heads = config.num_heads
        self.inner_dim = self.n_heads * self.key_value_proj_dim
        self.attention_type = _cfg(config, "attention_type", "triton")
        self.position_encoding_type = _cfg(config, "position_encoding_type", "t5")
        scale = _cfg(config, "attention_scale", None)
        # (the reference's default is 1/sqrt(n_heads), modeling_flash_t5.py:184 -- kept for checkpoint parity)
        self.softmax_scale = scale if scale is not None else 1.0 / math.sqrt(self.n_heads)
        self.use_full_bias_size = _cfg(config, "use_full_bias_size", False)
        self.use_masking = _cfg(config, "use_masking", False)
        if self.attention_type not in ("triton", "fat5_rpe"):
            raise ValueError(f"attention_type {self.attention_type!r}: this module implements 'triton' (dense bias) and 'fat5_rpe'")
        if _cfg(config, "attention_dropout_rate", 0.0) != 0.0:
            raise ValueError("attention dropout is not supported by the fused kernels")
        if self.attention_type == "fat5_rpe" and (self.position_encoding_type != "t5" or self.use_masking):
            raise ValueError("fat5_rpe needs the T5 relative-position encoding and no key masking (use var-len batches)")
        self.pe_encoding = None
        if self.position_encoding_type == "t5" and has_positional_encoding:
            self.pe_encoding = RelativePositionalEncoding(
                config.relative_attention_num_buckets, config.relative_attention_max_distance, self.n_heads,
                _cfg(config, "max_sequence_length", 0), bidirectional=not self.is_decoder,
                randomized_position=_cfg(config, "use_randomized_position_encoding", False))
        elif self.position_encoding_type != "t5" and has_positional_encoding:
            raise ValueError("only the T5 producer is built in; pass other encodings' dense bias as position_bias")
        self.Wq = nn.Linear(self.d_model, self.inner_dim, bias=False)
        self.Wk = nn.Linear(self.d_model, self.inner_dim, bias=False)
        self.Wv = nn.Linear(self.d_model, self.inner_dim, bias=False)
        self.o = nn.Linear(self.inner_dim, self.d_model, bias=False)

    def forward(self, hidden_states, mask=None, key_value_states=None, position_bias=None):
        B, M = hidden_states.shape[:2]
        src = hidden_states if key_value_states is None else key_value_states
        N = src.shape[1]
        # (B, S, H, D) storage viewed as (B, H, S, D): the kernels take the strided views as they are
        q = self.Wq(hidden_states).view(B, M, self.n_heads, self.key_value_proj_dim).permute(0, 2, 1, 3)
        k = self.Wk(src).view(B, N, self.n_heads, self.key_value_proj_dim).permute(0, 2, 1, 3)
        v = self.Wv(src).view(B, N, self.n_heads, self.key_value_proj_dim).permute(0, 2, 1, 3)
        out, position_bias = self._attend(q, k, v, hidden_states.dtype, mask, key_value_states is None, position_bias)
        return self.o(out), position_bias

    def forward_fused(self, hidden_states, norm_weight, eps, mask=None, key_value_states=None, position_bias=None):
        """One whole T5 attention sub-layer, `h + o(attention(layer_norm(h)))` (reference modeling_flash_t5.py:304-318 / :321-349),
        with the pre-norm inside the projection GEMM and the residual add as the output projection's epilogue
        (`fused_linear.rmsnorm_linear` / `linear_residual`): `hidden_states` is the UN-normalised residual stream,
        `norm_weight` / `eps` the sub-layer's `layer_norm`.  Returns (new residual stream, position_bias)."""
        from .fused_linear import rmsnorm_linear, linear_residual
        B, M = hidden_states.shape[:2]
        H, Dh = self.n_heads, self.key_value_proj_dim
        if key_value_states is None:  # self-attention: ONE GEMM for q, k, v
            qkv, res = rmsnorm_linear(hidden_states, norm_weight, (self.Wq.weight, self.Wk.weight, self.Wv.weight), eps, return_residual=True)
            q, k, v = unpack_heads(qkv, 3, H)
        else:                         # cross-attention: the decoder side is normed, the encoder output is not (:330-336)
            N = key_value_states.shape[1]
            q, res = rmsnorm_linear(hidden_states, norm_weight, self.Wq.weight, eps, return_residual=True)
            q = q.view(B, M, H, Dh).permute(0, 2, 1, 3)
            k, v = unpack_heads(torch.nn.functional.linear(key_value_states, torch.cat((self.Wk.weight, self.Wv.weight), 0)), 2, H)
        out, position_bias = self._attend(q, k, v, hidden_states.dtype, mask, key_value_states is None, position_bias)
        return linear_residual(out, self.o.weight, res), position_bias

    def _attend(self, q, k, v, dtype, mask, is_self, position_bias):
        """attention of projected (B, H, S, D) views -> (B, M, inner_dim), plus the position bias to hand on"""
        B, _, M, _ = q.shape
        N = k.shape[2]
        key_value_states = None if is_self else True
        hidden_dtype = dtype
        if self.attention_type == "fat5_rpe":
            if position_bias is None and self.pe_encoding is not None:
                position_bias = self.pe_encoding.forward_1d()
            if position_bias is None:
                # no producer and nothing handed on: T5 cross-attention (reference :207,:324 -> bias=None, HAS_BIAS=False).
                # A SELF-attention layer without a producer must be handed block 0's (rpe1d, radius): training it silently
                # without the T5 bias would be a wiring bug of the caller, not a mode.
                if key_value_states is None and self.position_encoding_type == "t5":
                    raise ValueError("fat5_rpe self-attention without a bias producer needs position_bias=(rpe1d, radius) from the "
                                     "block that owns the RelativePositionalEncoding")
                out = flash_attention_v2_bias(q, k, v, None, self.is_causal, self.softmax_scale)
            else:
                rpe1d, radius = position_bias
                out = flash_attention_v2_rpe1d(q, k, v, rpe1d, radius, self.is_causal, self.softmax_scale)
        else:
            if position_bias is None and self.pe_encoding is not None:
                position_bias = self.pe_encoding.compute_bias(M, N, device=q.device).contiguous().to(q.dtype)
            bias = position_bias
            if bias is not None and self.use_full_bias_size:
                bias = bias.expand(B, self.n_heads, M, N).contiguous()
                position_bias = bias
            if bias is not None and mask is not None and self.use_masking:
                m = mask.unsqueeze(1)
                if m.dim() == 3:
                    m = m.unsqueeze(3)
                bias = torch.where(m, bias, torch.finfo(hidden_dtype).min)
                position_bias = bias
            out = flash_attention_v2_bias(q, k, v, bias, self.is_causal, self.softmax_scale)
        return out.permute(0, 2, 1, 3).reshape(B, M, self.inner_dim), position_bias
