"""Multi-GPU decomposition of the attention path (one process per GPU, torch.distributed "nccl" = RCCL).

The path shards embarrassingly over the B*H independent (batch, head) attention problems (grid axes 1 and 2 of
every reference kernel, flash_attention_v2_bias.py:57,126,164,192): the forward needs no communication and the
backward needs exactly ONE reduction -- the bias gradient sums over the batch axis.

 * data parallel (config 5, what bench.py runs): every rank owns its own batch; after the local backward
   `allreduce_bias_grad` sums the (num_buckets, H) table gradient (1.5 KB) -- or the dense (1,H,M,N) dbias -- over
   xGMI with a single RCCL all-reduce.
 * unit sharding of ONE batch (strong scaling): `shard_units` deals contiguous head-major chunks of the B*H
   units; a rank that holds every batch element of a head owns that head's bias gradient outright, so only heads
   split across ranks need the all-reduce.
"""
from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_units(B: int, H: int, world: int, rank: int) -> List[Tuple[int, int]]:
    """(b, h) units of this rank: head-major order (unit u = h*B + b), contiguous balanced chunks."""
    total = B * H
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    stop = start + base + (1 if rank < rem else 0)
    return [(u % B, u // B) for u in range(start, stop)]


def unit_range(B: int, H: int, world: int, rank: int) -> Tuple[int, int]:
    """(unit_begin, unit_count) of this rank's chunk in head-major unit order -- what `fat5_attn_params.unit_begin/unit_count`
    (and `AttentionPlan(units=...)`) take: the shard of `shard_units` runs in ONE forward and ONE backward call.
    world > B * H leaves the trailing ranks an EMPTY chunk (count 0): `AttentionPlan` then launches nothing (in the raw C ABI
    unit_count == 0 means "the whole problem" -- never pass an empty chunk there)."""
    total = B * H
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def heads_needing_reduction(B: int, H: int, world: int) -> List[int]:
    """heads whose B batch elements are spread over more than one rank (their dbias needs the all-reduce)."""
    owners = {}
    for r in range(world):
        for (b, h) in shard_units(B, H, world, r):
            owners.setdefault(h, set()).add(r)
    return [h for h, rs in sorted(owners.items()) if len(rs) > 1]


def allreduce_bias_grad(grad: torch.Tensor, group=None, async_op: bool = False):
    """The one collective of the path: SUM of the bias(-table) gradient over ranks, accumulated in fp32.
    `grad` is reduced in place (fp32 tensors directly; low-precision tensors through an fp32 staging copy)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    if grad.dtype == torch.float32:
        return dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    acc = grad.float()
    dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    grad.copy_(acc)
    return None


class OverlappedGradReduce:
    """Data-parallel steps whose only exchange is a tiny gradient (the (num_buckets, H) table gradient, 1.5 KB): the
    all-reduce of step i runs on RCCL's stream while step i+1 computes, like DDP overlaps its buckets with the rest of
    the backward.  Two staging buffers alternate: `submit(grad)` snapshots the gradient (stream-ordered copy) and starts
    one asynchronous SUM all-reduce; the buffer is reused two steps later, after waiting for its previous reduction.
    Every step's gradient is reduced exactly once; `drain()` waits for whatever is still in flight and returns the
    results in submission order."""

    def __init__(self, like: torch.Tensor, group=None):
        self.group = group
        self.buf = [torch.empty_like(like, dtype=torch.float32) for _ in range(2)]
        self.work = [None, None]
        self.n = 0
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    def submit(self, grad: torch.Tensor):
        """returns the reduced gradient of the step submitted two calls ago (None for the first two calls)"""
        if not self.active:
            return grad
        j = self.n % 2
        done = None
        if self.work[j] is not None:
            self.work[j].wait()
            done = self.buf[j].clone() if self.keep_results else None
        self.buf[j].copy_(grad)
        self.work[j] = dist.all_reduce(self.buf[j], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.n += 1
        return done

    keep_results = False  # tests flip this to check every reduced value; the bench only needs the work done

    def drain(self):
        out = []
        if not self.active:
            return out
        order = [self.n % 2, (self.n + 1) % 2]  # older submission first
        for j in order:
            if self.work[j] is not None:
                self.work[j].wait()
                self.work[j] = None
                out.append(self.buf[j])
        return out


def gather_units(local: torch.Tensor, B: int, H: int, world: int, rank: int, group=None) -> torch.Tensor:
    """Reassemble a (B, H, ...) tensor from per-rank stacks of unit results (test / validation helper; the hot
    path never gathers activations)."""
    units = shard_units(B, H, world, rank)
    assert local.shape[0] == len(units)
    counts = [len(shard_units(B, H, world, r)) for r in range(world)]
    pad = max(counts)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    out = torch.empty((B, H) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        for i, (b, h) in enumerate(shard_units(B, H, world, r)):
            out[b, h] = bufs[r][i]
    return out
