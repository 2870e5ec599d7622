"""`AdamWScale` on MI355X: the reference optimizer (src/utils/adamw_scaled.py:10-281 -- AdamW whose step is scaled by
max(1e-3, rms(p)), optional Kahan-compensated updates for 16-bit parameters, decoupled weight decay) with the same constructor
and state (`step`, `exp_avg`, `exp_avg_sq`, `kahan_comp`: reference checkpoints of the optimizer load), its whole step fused into
two HIP launches per (device, dtype) group of parameters through the C ABI (`fat5_adamw_scale_step`, csrc/adamw_kernels.h)
instead of ~14 elementwise launches per tensor.  Arithmetic and intermediate roundings follow the reference's per-tensor path
op by op (oracle/adamw_scale.py is its CPU restatement, pinned against the reference class)."""
import ctypes
import math
from typing import Iterable, Tuple

import torch
from torch import nn
from torch.optim import Optimizer

from . import _lib

__all__ = ["AdamWScale"]

CHUNK = 8192  # elements per workgroup (csrc/adamw_kernels.h kAdamChunk)


class _Desc(ctypes.Structure):
    """mirror of `fat5_adamw_tensor` (include/fat5.h)"""
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p), ("k", ctypes.c_void_p),
                ("numel", ctypes.c_int64), ("chunk_begin", ctypes.c_int32), ("step_prefactor", ctypes.c_float)]


class AdamWScale(Optimizer):
    """Same arguments as the reference class (:40-66).  `foreach` is accepted and ignored (the fused step replaces both of the
    reference's paths); `use_state_dtype` = torch.float16 / torch.bfloat16 keeps exp_avg / exp_avg_sq in that dtype beside
    parameters of another one (reference :101-103; any other value = the parameter dtype, like the reference).

    Extension (keyword only): `max_grad_norm` folds `torch.nn.utils.clip_grad_norm_(params, max_grad_norm)` -- what the reference's
    trainer runs before every optimizer step (`max_grad_norm: 1.0`) -- into the step: one more pass over the gradients for the global
    norm, the clip coefficient stays on the device and scales every gradient as it is read (rounded to the gradient dtype like the
    in-place multiply).  The gradient tensors are left untouched; `last_grad_norm` holds the norm (device tensor) for logging."""

    def __init__(self, params: Iterable[nn.parameter.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-6, weight_decay: float = 0.0, kahan_sum: bool = False, foreach: bool = False,
                 correct_bias: bool = True, use_state_dtype: torch.dtype = None, *, max_grad_norm: float = None):
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr} - should be >= 0.0")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter: {betas[0]} - should be in [0.0, 1.0)")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter: {betas[1]} - should be in [0.0, 1.0)")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps} - should be >= 0.0")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=foreach, kahan_sum=kahan_sum,
                        correct_bias=correct_bias, use_state_dtype=use_state_dtype)
        super().__init__(params, defaults)
        if max_grad_norm is not None and not max_grad_norm > 0.0:
            raise ValueError(f"Invalid max_grad_norm: {max_grad_norm} - should be > 0")
        self.max_grad_norm = max_grad_norm
        self.last_grad_norm = None
        if ctypes.sizeof(_Desc) != _lib.load().fat5_sizeof_adamw_tensor():
            raise ImportError("fat5_adamw_tensor layout mismatch between libfat5.so and its binding")

    @staticmethod
    def _prefactor(lr, beta1, beta2, step, correct_bias):
        """lr [* sqrt(1 - beta2^t) / (1 - beta1^t)] with the reference's types (:177-181): `step` is an int32 tensor there, so the
        bias corrections are float32 tensors and the product is float32"""
        if not correct_bias:
            return float(lr)  # (the kernel rounds lr * rms(p) to the parameter dtype like the reference's Python-float x tensor product)
        st = torch.as_tensor(step, dtype=torch.int32)
        bc1 = 1.0 - beta1 ** st
        bc2 = 1.0 - beta2 ** st
        return float(lr * math.sqrt(bc2) / bc1)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        if capturing:  # (a step recorded into a HIP graph: see graph_advance)
            # ONE captured step per optimizer: the graph bakes raw pointers to the arena's descriptor table and step scalars into its
            # launches, and graph_advance() writes the scalars of exactly one capture -- a second capture would hand the same arena
            # bytes out again (the first graph would then replay over another table) and leave the first graph's scalars stale
            if getattr(self, "_graph_jobs", None):
                raise RuntimeError("AdamWScale: this optimizer already holds a captured step; destroy that graph and call "
                                   "release_captured_step() before capturing another one")
            self._graph_jobs, self._graph_keep = [], []
            self._capture_generation = getattr(self, "_capture_generation", 0) + 1  # (ownership token of THIS capture: capture_token())
        jobs = []  # one per (group, device, dtype, kahan) bucket: descriptor table on the device, launched below
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            buckets = {}
            steps = []
            pre_cache = {}  # step count -> prefactor (all tensors of a group normally share one step count)
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("AdamWScale does not support sparse gradients")
                if not p.is_cuda:
                    raise RuntimeError("flasht5_amd.AdamWScale needs parameters on the HIP device (no CPU fallback)")
                state = self.state[p]
                if capturing and "kahan_comp" not in state:
                    raise RuntimeError("AdamWScale: the optimizer state must exist before a step is captured in a graph "
                                       "(run one eager step first, or call init_state())")
                if "kahan_comp" not in state:  # reference :96-113
                    # (the reference keeps `step` on p.device; it is only ever read on the host -- beta ** step -- so it lives there)
                    state["step"] = torch.tensor(0, dtype=torch.int32)
                    if group["use_state_dtype"] in (torch.float16, torch.bfloat16):  # :101-103
                        state["exp_avg"] = torch.zeros_like(p, dtype=group["use_state_dtype"])
                        state["exp_avg_sq"] = torch.zeros_like(p, dtype=group["use_state_dtype"])
                    else:
                        state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    kah = group["kahan_sum"] and p.dtype in (torch.float16, torch.bfloat16)
                    state["kahan_comp"] = torch.zeros_like(p, memory_format=torch.preserve_format) if kah else None
                elif state["step"].device.type != "cpu":  # a reference checkpoint (device-side steps): one read, then host-side
                    state["step"] = state["step"].cpu()
                if state["exp_avg"].dtype != state["exp_avg_sq"].dtype or not (state["exp_avg"].is_contiguous() and state["exp_avg_sq"].is_contiguous()):
                    raise RuntimeError("AdamWScale: exp_avg / exp_avg_sq must be contiguous and share one dtype")
                steps.append(state["step"])
                if not (p.is_contiguous() and p.grad.is_contiguous()):
                    raise RuntimeError("AdamWScale: parameters and gradients must be contiguous")
                g = p.grad if p.grad.dtype == p.dtype else p.grad.to(p.dtype)
                buckets.setdefault((p.device, p.dtype, state["exp_avg"].dtype, state["kahan_comp"] is not None), []).append((p, g, state))
            if capturing:
                # one step count per group (the captured launch reads ONE prefactor per bucket from device memory), advanced by
                # graph_advance() before every replay -- capture itself runs nothing
                if len({int(t) for t in steps}) > 1:
                    raise RuntimeError("AdamWScale: a captured step needs one step count per parameter group")
                if group["weight_decay"] < 0:
                    raise RuntimeError("AdamWScale: negative weight_decay")
            elif steps:
                torch._foreach_add_(steps, 1)  # reference :120
            for (device, dtype, sdtype, kahan), items in buckets.items():
                table = (_Desc * (len(items) + 1))()
                chunk = 0
                keep = []
                for i, (p, g, state) in enumerate(items):
                    d = table[i]
                    d.p, d.g, d.m, d.v = p.data_ptr(), g.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr()
                    d.k = state["kahan_comp"].data_ptr() if kahan else None
                    d.numel, d.chunk_begin = p.numel(), chunk
                    st = int(state["step"])
                    if st not in pre_cache:
                        pre_cache[st] = self._prefactor(group["lr"], beta1, beta2, st, group["correct_bias"])
                    d.step_prefactor = pre_cache[st]
                    chunk += (p.numel() + CHUNK - 1) // CHUNK
                    keep.append(g)
                table[len(items)].chunk_begin = chunk
                if chunk == 0:
                    continue
                if capturing:
                    # capture runs nothing, so the table (pointers into the graph's own pool: they never change) is uploaded by an
                    # ordinary copy AFTER the capture, in front of the first replay (graph_advance).  Table and step scalars live in
                    # an arena allocated BEFORE the capture (init_state): memory taken from the graph's pool during capture may
                    # alias activations that were freed earlier in the capture -- every replay's forward would overwrite it.
                    host = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8)
                    raw = self._arena_take(device, host.numel())
                    scalars = self._arena_take(device, 12).view(torch.float32)
                    self._graph_jobs.append((group, [st for _, _, st in items], scalars, raw, host))
                else:
                    raw = self._upload(device, table)
                    scalars = None
                partials = torch.empty(chunk, dtype=torch.float32, device=device)
                flags = (1 if kahan else 0) | (0 if group["correct_bias"] else 2)  # FAT5_ADAMW_KAHAN | FAT5_ADAMW_PLAIN_STEP
                jobs.append((device, dtype, (sdtype, flags, scalars), raw, len(items), chunk, partials, group, keep))
        if not jobs:
            return loss
        coef = None
        if self.max_grad_norm is not None:
            if len({j[0] for j in jobs}) != 1:
                raise RuntimeError("AdamWScale(max_grad_norm=...): parameters on several devices are not supported")
            total = None
            for device, dtype, kahan, raw, n, chunk, partials, group, keep in jobs:
                with _lib.on_device(device):
                    _lib.check(lib.fat5_adamw_grad_sumsq(raw.data_ptr(), n, chunk, partials.data_ptr(), _lib.dtype_code(dtype),
                                                         _lib.stream_ptr(device)), "fat5_adamw_grad_sumsq")
                s_ = partials.sum()
                total = s_ if total is None else total + s_
            norm = total.sqrt()
            self.last_grad_norm = norm
            coef = (self.max_grad_norm / (norm + 1e-6)).clamp(max=1.0).float().contiguous()  # clip_grad_norm_'s clip_coef_clamped
        for device, dtype, kahan, raw, n, chunk, partials, group, keep in jobs:
            beta1, beta2 = group["betas"]
            sdtype, flags, scalars = kahan
            if scalars is not None:
                with _lib.on_device(device):
                    _lib.check(lib.fat5_adamw_scale_step_dev(raw.data_ptr(), n, chunk, partials.data_ptr(), scalars.data_ptr(), float(beta1),
                                                             float(beta2), float(group["eps"]), _lib.dtype_code(dtype), _lib.dtype_code(sdtype),
                                                             int(flags), coef.data_ptr() if coef is not None else None,
                                                             _lib.stream_ptr(device)), "fat5_adamw_scale_step_dev")
                self._graph_keep += [partials, coef, keep]
                continue
            args = (raw.data_ptr(), n, chunk, partials.data_ptr(), float(group["lr"]), float(beta1), float(beta2),
                    float(group["weight_decay"]), float(group["eps"]), _lib.dtype_code(dtype), _lib.dtype_code(sdtype), int(flags))
            with _lib.on_device(device):
                if coef is None:
                    _lib.check(lib.fat5_adamw_scale_step(*args, _lib.stream_ptr(device)), "fat5_adamw_scale_step")
                else:
                    _lib.check(lib.fat5_adamw_scale_step_clipped(*args, coef.data_ptr(), _lib.stream_ptr(device)), "fat5_adamw_scale_step_clipped")
        return loss

    @torch.no_grad()
    def init_state(self):
        """allocate exp_avg / exp_avg_sq / kahan_comp / step of every parameter now (what the first step() does lazily, reference
        :96-113): a step can only be captured in a graph once its state exists"""
        for group in self.param_groups:
            for p in group["params"]:
                state = self.state[p]
                if "kahan_comp" in state or not p.requires_grad:
                    continue
                state["step"] = torch.tensor(0, dtype=torch.int32)
                sd = group["use_state_dtype"] if group["use_state_dtype"] in (torch.float16, torch.bfloat16) else p.dtype
                state["exp_avg"] = torch.zeros_like(p, dtype=sd, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(p, dtype=sd, memory_format=torch.preserve_format)
                kah = group["kahan_sum"] and p.dtype in (torch.float16, torch.bfloat16)
                state["kahan_comp"] = torch.zeros_like(p, memory_format=torch.preserve_format) if kah else None
        # descriptor tables + step scalars of a captured step (see step()): one arena per device, [tensor, bytes handed out].  An arena
        # a captured graph may be reading is never replaced or resized: it is allocated once and only while no step is captured.
        if not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            count = {}
            for group in self.param_groups:
                for p in group["params"]:
                    if p.is_cuda:
                        count[p.device] = count.get(p.device, 0) + 1
            need = {dev: (2 * n + 8 * len(self.param_groups) + 8) * 128 for dev, n in count.items()}
            have = getattr(self, "_graph_arena", None)
            if have is None or any(dev not in have or have[dev][0].numel() < nb for dev, nb in need.items()):
                if getattr(self, "_graph_jobs", None):
                    raise RuntimeError("AdamWScale.init_state: parameters were added while a captured step exists; destroy that graph "
                                       "and call release_captured_step() first")
                self._graph_arena = {dev: [torch.empty(nb, dtype=torch.uint8, device=dev), 0] for dev, nb in need.items()}

    def capture_token(self):
        """Ownership token of the captured step this optimizer currently holds (0: none).  Whoever captured keeps the token and hands it to
        `release_captured_step(token)`: a holder whose capture was already released -- and replaced by a newer one -- then releases nothing
        (an old GraphedTrainStep collected late must not clear the arena a live graph reads)."""
        return getattr(self, "_capture_generation", 0) if getattr(self, "_graph_jobs", None) else 0

    def release_captured_step(self, token=None):
        """Forget the captured step (the caller has destroyed every graph that holds it): its arena bytes may be handed out again.
        token (from `capture_token()` right after the capture): release only if the optimizer still holds THAT capture; None: unconditionally."""
        if token is not None and token != self.capture_token():
            return False
        self._graph_jobs, self._graph_keep = [], []
        for a in getattr(self, "_graph_arena", {}).values():
            a[1] = 0
        return True

    def _upload(self, device, table):
        """descriptor table -> device, ASYNCHRONOUSLY: through one of eight rotating pinned staging buffers (a copy from pageable memory
        makes the host wait for the stream, i.e. for the whole backward pass in front of it -- the host could never run ahead of the
        device).  A staging buffer is rewritten only after the copy issued from it eight uploads ago has completed."""
        n = ctypes.sizeof(table)
        ring = self.__dict__.setdefault("_staging", {}).setdefault(device, {"i": 0, "slot": 0, "pin": None, "ev": [None] * 8})
        if ring["pin"] is None or ring["slot"] < n:  # (ONE pinned allocation for the whole ring: hipHostMalloc costs milliseconds)
            for ev in ring["ev"]:
                if ev is not None:
                    ev.synchronize()
            ring["slot"] = max(2 * n, 1 << 16)
            ring["pin"] = torch.empty(8 * ring["slot"], dtype=torch.uint8).pin_memory()
            ring["ev"] = [None] * 8
        i = ring["i"]
        ring["i"] = (i + 1) % 8
        if ring["ev"][i] is not None:
            ring["ev"][i].synchronize()
        buf = ring["pin"][i * ring["slot"]:(i + 1) * ring["slot"]]
        ctypes.memmove(buf.data_ptr(), ctypes.addressof(table), n)
        raw = torch.empty(n, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            raw.copy_(buf[:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        ring["ev"][i] = ev
        return raw

    def _arena_take(self, device, nbytes):
        """`nbytes` of the pre-capture arena of `device` (64-byte aligned pieces, handed out in order)"""
        a = getattr(self, "_graph_arena", {}).get(device)
        if a is None:
            raise RuntimeError("AdamWScale: call init_state() before capturing a step (it allocates the buffers a captured step "
                               "keeps outside the graph's memory pool)")
        off = a[1]
        a[1] = off + (nbytes + 63) // 64 * 64
        if a[1] > a[0].numel():
            raise RuntimeError("AdamWScale: capture arena too small (parameters added after init_state()?)")
        return a[0][off:off + nbytes]

    @torch.no_grad()
    def graph_advance(self):
        """Before every replay of a graph that holds a captured step(): count the step (reference :120) and write the three
        step-dependent scalars of each bucket -- prefactor from the current `lr` and step count, -lr * weight_decay, lr * 1e-3 -- into
        its device tensor (stream-ordered in front of the replay; the values the eager step passes as launch arguments)."""
        if not getattr(self, "_graph_jobs", None):
            raise RuntimeError("AdamWScale.graph_advance: no captured step")
        for i, (group, states, scalars, raw, host) in enumerate(self._graph_jobs):
            if host is not None:  # first replay: the descriptor table of the bucket
                raw.copy_(host)
                self._graph_jobs[i] = (group, states, scalars, raw, None)
            torch._foreach_add_([st["step"] for st in states], 1)
            beta1, beta2 = group["betas"]
            lr, wd = float(group["lr"]), float(group["weight_decay"])
            pre = self._prefactor(lr, beta1, beta2, int(states[0]["step"]), group["correct_bias"])
            # (three fill launches -- the values travel as launch arguments; a copy from a host tensor would wait for the stream,
            #  i.e. for the previous replay, and the host could never run ahead of the device)
            for j, val in enumerate((pre, -lr * wd if wd > 0.0 else 0.0, lr * 1e-3)):
                scalars[j:j + 1].fill_(val)
