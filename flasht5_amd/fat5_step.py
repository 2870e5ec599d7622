"""Config 5 of BASELINE.json: a FAT5-base UL2 training step (seq 1024, data parallel) built from the path's three operators.

A minimal encoder-decoder with the reference's structure (src/model/modeling_flash_t5.py: `FlashT5LayerFF` :148-164,
`FlashT5LayerSelfAttention` / `FlashT5LayerCrossAttention` :297-349, `FlashT5Block` :352-392, `FlashT5Stack` :394-464,
`FlashT5ForConditionalGeneration` :604-736) and its parameter names, so a reference checkpoint's state dict loads:

  * RMSNorm        -> `FlashT5LayerNorm`        (fast_rms_layernorm, HIP)
  * attention      -> `FlashT5Attention`        (flash_attention_v2_rpe1d / flash_attention_v2_bias, HIP); the T5 relative-position
                      bias is produced by block 0 of each stack and handed to the following blocks (:403-405, :452-455);
                      cross-attention has no bias (`bias=None`, :207,:324)
  * loss           -> `FlashT5CrossEntropyLoss` (cross_entropy_loss with z-loss and label smoothing, HIP)
  * everything else is plain torch-ROCm: `nn.Linear` (hipBLASLt), `nn.Embedding`, tanh-GELU gating, residual adds.

This is the step DRIVER of the hot path, not a model zoo: no generation, heads, dropout (0 in every reference config),
HF plumbing or checkpoint conversion.  Data parallelism = one process per GPU; `allreduce_gradients` is the step's one
exchange (RCCL over xGMI): a single flat fp32 all-reduce that carries the two `(32, H)` relative-position tables first.
"""
import math
from dataclasses import dataclass

import torch
import torch.distributed as dist
from torch import nn

from .attention_module import FlashT5Attention
from .modules import FlashT5LayerNorm, FlashT5CrossEntropyLoss


@dataclass
class FAT5Config:
    """model_args of configs/flan/fat5-flan-base.yaml (vocabulary: the 32768-entry tokenizer of examples/minipile)"""
    vocab_size: int = 32768
    d_model: int = 768
    d_kv: int = 64
    d_ff: int = 2048
    num_heads: int = 12
    num_layers: int = 12
    num_decoder_layers: int = 12
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    max_sequence_length: int = 1024
    layer_norm_epsilon: float = 1e-6
    z_loss: float = 1e-4
    label_smoothing: float = 0.1
    attention_scale: float = 1.0
    attention_type: str = "fat5_rpe"      # "fat5_rpe": O(S) bias memory; "triton": the reference's dense-bias operator
    position_encoding_type: str = "t5"
    use_glu_mlp: bool = True
    use_gelu_act: bool = True
    decoder_start_token_id: int = 0
    pad_token_id: int = 0
    crossentropy_inplace_backward: bool = True
    fuse_lm_head_ce: bool = False          # lm_head + loss in row chunks: the (B*T, vocab) logits are never materialised
    fuse_add_norm: bool = False            # every residual add runs inside the next pre-norm (fused_add_rms_layernorm): same bits, fewer passes
    fuse_norm_linear: bool = False         # stacked projections (q | k | v, wi_0 | wi_1: one library GEMM each, packed gradients), the residual add in the
                                           # output projection's GEMM epilogue, fused backward passes around them (fused_linear.py)
    fuse_gated_act: bool = True            # act(wi_0 x) * wi_1 x in one kernel forward, one backward (gated_act.py / fat5_gated_act_*)
    is_decoder: bool = False


class FAT5GatedAct(nn.Module):  # reference FlashT5DenseGatedAct / FlashT5DenseAct (:114-146)
    def __init__(self, c):
        super().__init__()
        self.glu = c.use_glu_mlp
        if self.glu:
            self.wi_0 = nn.Linear(c.d_model, c.d_ff, bias=False)
            self.wi_1 = nn.Linear(c.d_model, c.d_ff, bias=False)
        else:
            self.wi = nn.Linear(c.d_model, c.d_ff, bias=False)
        self.act = nn.GELU(approximate="tanh") if c.use_gelu_act else nn.ReLU()
        self.act_name = "gelu_tanh" if c.use_gelu_act else "relu"
        self.fuse = bool(getattr(c, "fuse_gated_act", True))

    def fused_ok(self, x):
        return self.fuse and x.is_cuda and x.dtype in (torch.float32, torch.float16, torch.bfloat16)

    def forward(self, x):
        if self.glu:
            if self.fused_ok(x) and self.wi_0.weight.shape[0] % 8 == 0:
                from .gated_act import gated_act
                return gated_act(self.wi_0(x), self.wi_1(x), self.act_name)
            return self.act(self.wi_0(x)) * self.wi_1(x)
        return self.act(self.wi(x))


def _pre_norm(ln, h, pending):
    """pre-norm of a sub-layer with the previous sub-layer's residual add folded in: (h + pending, layer_norm(h + pending))"""
    if pending is None:
        return h, ln(h)
    from .rms_norm import fused_add_rms_layernorm
    return fused_add_rms_layernorm(h, pending, ln.weight, ln.variance_epsilon)


class FAT5LayerFF(nn.Module):  # :148-164
    def __init__(self, c):
        super().__init__()
        self.act = FAT5GatedAct(c)
        self.layer_norm = FlashT5LayerNorm(c.d_model, eps=c.layer_norm_epsilon)
        self.wo = nn.Linear(c.d_ff, c.d_model, bias=False)

    def forward(self, h):
        return h + self.wo(self.act(self.layer_norm(h)))

    def forward_fused(self, h):
        """the same sub-layer with the norm inside the wi GEMM and the residual add as wo's epilogue"""
        from .fused_linear import rmsnorm_linear, linear_residual
        a = self.act
        if a.glu:
            g, h = rmsnorm_linear(h, self.layer_norm.weight, (a.wi_0.weight, a.wi_1.weight), self.layer_norm.variance_epsilon, return_residual=True)
            if a.fused_ok(g) and a.wi_0.weight.shape[0] % 8 == 0:
                from .gated_act import gated_act_packed
                t = gated_act_packed(g, a.act_name)  # (its gradient is ONE (…, 2 d_ff) tensor: no concatenation in front of the backward GEMMs)
            else:
                g0, g1 = g.split(a.wi_0.weight.shape[0], dim=-1)
                t = a.act(g0) * g1
        else:
            t, h = rmsnorm_linear(h, self.layer_norm.weight, a.wi.weight, self.layer_norm.variance_epsilon, return_residual=True)
            t = a.act(t)
        return linear_residual(t, self.wo.weight, h)

    def forward_deferred(self, h, pending):
        """(h, delta): the residual stream after the pending add, and this sub-layer's output whose add is left to the next pre-norm"""
        h, n = _pre_norm(self.layer_norm, h, pending)
        return h, self.wo(self.act(n))


class FAT5LayerSelfAttention(nn.Module):  # :297-318
    def __init__(self, c, has_positional_encoding):
        super().__init__()
        self.self_attention = FlashT5Attention(c, has_positional_encoding=has_positional_encoding, is_causal=c.is_decoder)
        self.layer_norm = FlashT5LayerNorm(c.d_model, eps=c.layer_norm_epsilon)

    def forward(self, h, position_bias=None):
        a, position_bias = self.self_attention(self.layer_norm(h), position_bias=position_bias)
        return h + a, position_bias

    def forward_fused(self, h, position_bias=None):
        return self.self_attention.forward_fused(h, self.layer_norm.weight, self.layer_norm.variance_epsilon, position_bias=position_bias)

    def forward_deferred(self, h, pending, position_bias=None):
        h, n = _pre_norm(self.layer_norm, h, pending)
        a, position_bias = self.self_attention(n, position_bias=position_bias)
        return h, a, position_bias


class FAT5LayerCrossAttention(nn.Module):  # :321-349
    def __init__(self, c):
        super().__init__()
        self.cross_attention = FlashT5Attention(c, has_positional_encoding=False)
        self.layer_norm = FlashT5LayerNorm(c.d_model, eps=c.layer_norm_epsilon)

    def forward(self, h, key_value_states):
        a, _ = self.cross_attention(self.layer_norm(h), key_value_states=key_value_states)
        return h + a

    def forward_fused(self, h, key_value_states):
        return self.cross_attention.forward_fused(h, self.layer_norm.weight, self.layer_norm.variance_epsilon, key_value_states=key_value_states)[0]

    def forward_deferred(self, h, pending, key_value_states):
        h, n = _pre_norm(self.layer_norm, h, pending)
        a, _ = self.cross_attention(n, key_value_states=key_value_states)
        return h, a


class FAT5Block(nn.Module):  # :352-392
    def __init__(self, c, has_positional_encoding):
        super().__init__()
        self.is_decoder = c.is_decoder
        self.self_attention_layer = FAT5LayerSelfAttention(c, has_positional_encoding)
        if self.is_decoder:
            self.cross_attention_layer = FAT5LayerCrossAttention(c)
        self.ff_layer = FAT5LayerFF(c)

    def forward(self, h, position_bias=None, encoder_hidden_states=None):
        h, position_bias = self.self_attention_layer(h, position_bias)
        if self.is_decoder and encoder_hidden_states is not None:
            h = self.cross_attention_layer(h, encoder_hidden_states)
        return self.ff_layer(h), position_bias

    def forward_fused(self, h, position_bias=None, encoder_hidden_states=None):
        h, position_bias = self.self_attention_layer.forward_fused(h, position_bias)
        if self.is_decoder and encoder_hidden_states is not None:
            h = self.cross_attention_layer.forward_fused(h, encoder_hidden_states)
        return self.ff_layer.forward_fused(h), position_bias

    def forward_deferred(self, h, pending, position_bias=None, encoder_hidden_states=None):
        h, pending, position_bias = self.self_attention_layer.forward_deferred(h, pending, position_bias)
        if self.is_decoder and encoder_hidden_states is not None:
            h, pending = self.cross_attention_layer.forward_deferred(h, pending, encoder_hidden_states)
        h, pending = self.ff_layer.forward_deferred(h, pending)
        return h, pending, position_bias


class FAT5Stack(nn.Module):  # :394-464
    def __init__(self, c, embed_tokens, n_layers):
        super().__init__()
        self.embed_tokens = embed_tokens
        self.block = nn.ModuleList([FAT5Block(c, has_positional_encoding=(i == 0)) for i in range(n_layers)])
        self.final_layer_norm = FlashT5LayerNorm(c.d_model, eps=c.layer_norm_epsilon)
        self.fuse_add_norm = c.fuse_add_norm
        self.fuse_norm_linear = c.fuse_norm_linear

    def forward(self, input_ids, encoder_hidden_states=None):
        h = self.embed_tokens(input_ids)
        if torch.is_autocast_enabled() and h.is_cuda:  # :424-425
            h = h.to(torch.get_autocast_gpu_dtype())
        position_bias = None  # produced by block 0, shared by the others (:452-455)
        if self.fuse_norm_linear and h.dtype in (torch.float16, torch.bfloat16):
            for blk in self.block:
                h, position_bias = blk.forward_fused(h, position_bias, encoder_hidden_states)
            return self.final_layer_norm(h)
        if self.fuse_add_norm:
            pending = None  # the last sub-layer's output: its residual add happens inside the next pre-norm
            for blk in self.block:
                h, pending, position_bias = blk.forward_deferred(h, pending, position_bias, encoder_hidden_states)
            return _pre_norm(self.final_layer_norm, h, pending)[1]
        for blk in self.block:
            h, position_bias = blk(h, position_bias, encoder_hidden_states)
        return self.final_layer_norm(h)


class FAT5ForConditionalGeneration(nn.Module):  # :604-736 (training forward only)
    def __init__(self, config: FAT5Config):
        super().__init__()
        import copy
        self.config = config
        self.shared = nn.Embedding(config.vocab_size, config.d_model)
        enc = copy.copy(config)
        enc.is_decoder = False
        dec = copy.copy(config)
        dec.is_decoder = True
        self.encoder = FAT5Stack(enc, self.shared, config.num_layers)
        self.decoder = FAT5Stack(dec, self.shared, config.num_decoder_layers)
        self.lm_head = nn.Linear(config.d_model, config.vocab_size, bias=False)
        self.loss_fct = FlashT5CrossEntropyLoss(z_loss_factor=config.z_loss, label_smoothing=config.label_smoothing,
                                                inplace_backward=config.crossentropy_inplace_backward)
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self):
        """the reference's `_init_weights` (:482-517): Mesh-TF style, factor 1.0"""
        c = self.config
        self.shared.weight.normal_(0.0, 1.0)
        self.lm_head.weight.normal_(0.0, c.d_model ** -0.5)
        for m in self.modules():
            if isinstance(m, FlashT5LayerNorm):
                m.weight.fill_(1.0)
            elif isinstance(m, FAT5GatedAct):
                for w in ((m.wi_0, m.wi_1) if m.glu else (m.wi,)):
                    w.weight.normal_(0.0, c.d_model ** -0.5)
            elif isinstance(m, FAT5LayerFF):
                m.wo.weight.normal_(0.0, c.d_ff ** -0.5)
            elif isinstance(m, FlashT5Attention):
                m.Wq.weight.normal_(0.0, (c.d_model * c.d_kv) ** -0.5)
                m.Wk.weight.normal_(0.0, c.d_model ** -0.5)
                m.Wv.weight.normal_(0.0, c.d_model ** -0.5)
                m.o.weight.normal_(0.0, (c.num_heads * c.d_kv) ** -0.5)
                if m.pe_encoding is not None:
                    m.pe_encoding.relative_attention_bias.weight.normal_(0.0, c.d_model ** -0.5)

    def _shift_right(self, labels):  # HF T5 convention used by the reference (:713-714)
        c = self.config
        shifted = labels.new_zeros(labels.shape)
        shifted[..., 1:] = labels[..., :-1]
        shifted[..., 0] = c.decoder_start_token_id
        return shifted.masked_fill(shifted == -100, c.pad_token_id)

    def rpe_tables(self):
        """the two (num_buckets, H) relative-position tables (encoder, decoder): the bias gradients of the step"""
        return [self.encoder.block[0].self_attention_layer.self_attention.pe_encoding.relative_attention_bias.weight,
                self.decoder.block[0].self_attention_layer.self_attention.pe_encoding.relative_attention_bias.weight]

    def forward(self, input_ids, labels):
        enc = self.encoder(input_ids)
        dec = self.decoder(self._shift_right(labels), encoder_hidden_states=enc)
        if self.config.fuse_lm_head_ce:
            from .lm_head_cross_entropy import lm_head_cross_entropy
            c = self.config
            return lm_head_cross_entropy(dec, self.lm_head.weight, labels, label_smoothing=c.label_smoothing,
                                         lse_square_scale=c.z_loss, reduction="mean")[0]
        return self.loss_fct(self.lm_head(dec), labels)


def allreduce_gradients(model: nn.Module, group=None, average=True):
    """The data-parallel exchange of one step: ONE flat fp32 all-reduce (SUM, then 1/world) of every parameter gradient,
    the `(32, H)` bias tables at the front of the buffer (what DDP's first bucket carries in the reference's runs,
    SURVEY 2 #14a).  In place; returns the flat reduced buffer (tests look at its head)."""
    params = [p for p in model.parameters() if p.grad is not None]
    if hasattr(model, "rpe_tables"):
        head = [p for p in model.rpe_tables() if p.grad is not None]
        ids = {id(p) for p in head}
        params = head + [p for p in params if id(p) not in ids]
    if not params:
        return None
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return None  # one rank: nothing to exchange (and no 1 GB flat copy of a FAT5-base gradient set)
    flat = torch.cat([p.grad.detach().reshape(-1).float() for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
    return flat


def train_step(model, input_ids, labels, optimizer=None, group=None, max_grad_norm=1.0):
    """forward + backward (+ gradient all-reduce, clip like the reference's `max_grad_norm: 1.0`, optimizer step)"""
    loss = model(input_ids, labels)
    loss.backward()
    allreduce_gradients(model, group)
    if optimizer is not None:
        if max_grad_norm and getattr(optimizer, "max_grad_norm", None) is None:  # (AdamWScale(max_grad_norm=...) clips inside its step)
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    return loss.detach()



class GraphedTrainStep:
    """`train_step` captured ONCE in HIP graphs and replayed per batch: forward, backward, global-norm clipping and the fused
    AdamWScale step of a fixed-shape batch are ~3,400 kernel launches whose enqueue time (24-29 ms of Python, dispatcher and
    hipLaunchKernel for FAT5-base at B = 4) exceeds their 17 ms of kernel time; a replay enqueues them in well under a millisecond.

        step = GraphedTrainStep(model, AdamWScale(model.parameters(), ..., max_grad_norm=1.0))
        for input_ids, labels in loader:       # every batch of ONE shape
            loss = step(input_ids, labels)     # (a static tensor, overwritten by the next call: .item() / .clone() to keep it)

    The first `warmup` calls run the eager `train_step` (they also create the optimizer state), the next call captures and replays, every later call only replays.  What changes from step to step reaches the graph
    through device memory: the batch (copied into static buffers), and the optimizer's step-dependent scalars (learning rate,
    bias correction: `AdamWScale.graph_advance`, so LR schedulers keep working -- they set `param_group["lr"]` as always).
    Gradients stay allocated between steps (the graph owns them): `p.grad` holds the last step's unclipped gradients.

    Data parallel (`group` with more than one rank; `split=True` forces this form): two graphs -- forward + backward, then the
    optimizer step -- with the flat gradient all-reduce (RCCL, `allreduce_gradients`) between them, outside any graph.

    What a captured step fixes (checked before every replay where it can be): `betas` and `eps` of every parameter group are launch
    arguments of the captured kernels -- changing them afterwards raises; the gradient tensors belong to the graph -- a
    `zero_grad(set_to_none=True)` / re-assignment of `p.grad` between replays raises (the graph would keep writing the old buffers
    while `allreduce_gradients` skipped the parameter); clipping happens only if the optimizer was built with `max_grad_norm`
    (the eager warm-up steps behave the same: `train_step(..., max_grad_norm=None)`); one captured step per optimizer
    (`close()` -- also run when this object is destroyed -- releases it: `AdamWScale.release_captured_step()`)."""

    def __init__(self, model, optimizer, group=None, warmup=2, split=None):
        from .adamw_scaled import AdamWScale
        if not isinstance(optimizer, AdamWScale):
            raise TypeError("GraphedTrainStep needs flasht5_amd.AdamWScale (its captured step reads lr / bias correction from device memory)")
        self.model, self.optimizer, self.group = model, optimizer, group
        self.warmup = int(warmup)
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.split = multi if split is None else (bool(split) or multi)
        self.calls = 0
        self.graphs = None
        self.ids = self.labels = self.loss = None

    def _eager(self, input_ids, labels):
        # (on the CURRENT stream.  Warming up on a side stream, as the PyTorch recipe has it, made the first replay after a
        #  device-wide synchronize fault on this ROCm stack -- HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION, full-size model only;
        #  tools/graph_step_debug.py -- and buys nothing here: no DDP hooks, one autograd stream)
        return train_step(self.model, input_ids, labels, self.optimizer, self.group, max_grad_norm=None)

    def _capture(self, input_ids, labels):
        self.ids, self.labels = input_ids.clone(), labels.clone()
        self.optimizer.init_state()
        self.optimizer.zero_grad(set_to_none=True)  # the gradients are (re)allocated inside the graph's pool
        try:
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                self.loss = self.model(self.ids, self.labels)
                self.loss.backward()
                if not self.split:
                    self.optimizer.step()
                self.loss = self.loss.detach()
            graphs = [g1]
            if self.split:
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=g1.pool()):
                    self.optimizer.step()
                graphs.append(g2)
        except BaseException:
            # a capture that threw midway must not leave the optimizer "holding a captured step" for good
            self.optimizer.release_captured_step()
            raise
        self.graphs = graphs
        self._token = self.optimizer.capture_token()  # (close() releases THIS capture only: ADVICE r5)
        # what the capture baked in (see the class docstring)
        self._baked = [(tuple(g["betas"]), float(g["eps"])) for g in self.optimizer.param_groups]
        self._grads = [(p, p.grad) for g in self.optimizer.param_groups for p in g["params"] if p.grad is not None]

    def close(self):
        """Drop the captured step: the graphs go and the optimizer may capture again (another GraphedTrainStep, e.g. for a new batch shape)."""
        if self.graphs is not None:
            self.graphs = None
            self.optimizer.release_captured_step(getattr(self, "_token", None))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check_baked(self):
        for g, (betas, eps) in zip(self.optimizer.param_groups, self._baked):
            if tuple(g["betas"]) != betas or float(g["eps"]) != eps:
                raise RuntimeError("GraphedTrainStep: betas / eps are launch arguments of the captured optimizer step and cannot change "
                                   "after capture (lr and weight_decay can: they travel through device memory)")
        for p, gr in self._grads:
            if p.grad is not gr:
                raise RuntimeError("GraphedTrainStep: p.grad was reset or replaced between replays; the captured graph owns the gradient "
                                   "tensors (use zero_grad(set_to_none=False) if they have to be cleared)")

    def __call__(self, input_ids, labels):
        self.calls += 1
        if self.graphs is None:
            if self.calls <= self.warmup:
                return self._eager(input_ids, labels)
            self._capture(input_ids, labels)
        else:
            if input_ids.shape != self.ids.shape or labels.shape != self.labels.shape:
                raise ValueError(f"GraphedTrainStep was captured for batches {tuple(self.ids.shape)} / {tuple(self.labels.shape)}")
            self.ids.copy_(input_ids, non_blocking=True)
            self.labels.copy_(labels, non_blocking=True)
        self._check_baked()
        self.optimizer.graph_advance()
        self.graphs[0].replay()
        if self.split:
            allreduce_gradients(self.model, self.group)
            self.graphs[1].replay()
        return self.loss
