"""T5 relative-position bias producers for the attention path.

Mirrors `RelativePositionalEncoding` of the reference (src/utils/positional_encoding.py:11-110):
same bucket formula, same dense `(1, H, M, N)` bias -- plus the linear-memory form consumed by the
RPE mode of the kernels: the Toeplitz generator `rpe1d[h, clamp(n - m, -R, R) + R]`.
"""
import math
from functools import lru_cache

import torch


def relative_position_bucket(relative_position, bidirectional=True, num_buckets=32, max_distance=128):
    """Same arithmetic as the reference `_relative_position_bucket` (positional_encoding.py:25-71):
    fp32 log ratio, truncation toward zero, clamp to the last bucket."""
    relative_buckets = torch.zeros_like(relative_position)
    if bidirectional:
        num_buckets //= 2
        relative_buckets = relative_buckets + (relative_position > 0).to(torch.long) * num_buckets
        relative_position = torch.abs(relative_position)
    else:
        relative_position = -torch.min(relative_position, torch.zeros_like(relative_position))
    max_exact = num_buckets // 2
    is_small = relative_position < max_exact
    safe = torch.clamp(relative_position, min=1).float()  # log(0) lanes are masked by is_small
    large = max_exact + (
        torch.log(safe / max_exact) / torch.log(torch.tensor(max_distance / max_exact)) * (num_buckets - max_exact)
    ).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return relative_buckets + torch.where(is_small, relative_position, large)


def compute_bias(table, query_length, key_length, bidirectional=True, num_buckets=32, max_distance=128,
                 context_position=None, memory_position=None):
    """Dense bias `(1, H, M, N)` from the `(num_buckets, H)` table (positional_encoding.py:73-102).
    `context_position (M,)` / `memory_position (N,)`: explicit (e.g. randomized, :79-89) positions instead of 0..M-1 / 0..N-1."""
    device = table.device
    ctx = torch.arange(query_length, dtype=torch.long, device=device) if context_position is None else context_position.to(device)
    mem = torch.arange(key_length, dtype=torch.long, device=device) if memory_position is None else memory_position.to(device)
    bucket = relative_position_bucket(mem[None, :] - ctx[:, None], bidirectional, num_buckets, max_distance)
    return table[bucket].permute(2, 0, 1).unsqueeze(0)


def randomized_positions(max_sequence_length, length):
    """`length` sorted distinct positions out of `max_sequence_length`, the first rooted at 0 -- the reference's draw
    (positional_encoding.py:80-88: `sort(randperm(max_len)[:length])`, element 0 overwritten with 0), same calls on the
    same (global CPU) generator, so a seeded run reproduces the reference's positions exactly."""
    idx, _ = torch.sort(torch.randperm(max_sequence_length)[:length])
    idx[0] = 0
    return idx


@lru_cache(maxsize=64)
def _bucket_index_cpu(radius, bidirectional, num_buckets, max_distance):
    delta = torch.arange(-radius, radius + 1, dtype=torch.long)
    return relative_position_bucket(delta, bidirectional, num_buckets, max_distance)


_IDX_CACHE = {}


def bucket_index(radius, bidirectional, num_buckets, max_distance, device):
    """bucket id of every clamped relative position delta in [-R, R]; cached per device."""
    key = (radius, bool(bidirectional), num_buckets, max_distance, str(device))
    idx = _IDX_CACHE.get(key)
    if idx is None:
        idx = _bucket_index_cpu(radius, bool(bidirectional), num_buckets, max_distance).to(device)
        _IDX_CACHE[key] = idx
    return idx


def bucket_index32(radius, bidirectional, num_buckets, max_distance, device):
    """int32 copy of `bucket_index` (what the C ABI's `rpe_bucket` takes)."""
    key = ("i32", radius, bool(bidirectional), num_buckets, max_distance, str(device))
    idx = _IDX_CACHE.get(key)
    if idx is None:
        idx = bucket_index(radius, bidirectional, num_buckets, max_distance, device).to(torch.int32).contiguous()
        _IDX_CACHE[key] = idx
    return idx


def rpe_radius(max_distance):
    """Beyond |n - m| >= max_distance every relative position falls in the last bucket of its side."""
    return int(max_distance)


def rpe1d_from_table(table, bidirectional=True, num_buckets=32, max_distance=128):
    """(num_buckets, H) table -> (H, 2R+1) fp32 generator with R = max_distance."""
    R = rpe_radius(max_distance)
    idx = bucket_index(R, bidirectional, num_buckets, max_distance, table.device)
    return table.index_select(0, idx).transpose(0, 1).float().contiguous()


class RelativePositionalEncoding(torch.nn.Module):
    """Mirror of the reference module (src/utils/positional_encoding.py:11-110; constructor arguments and the
    `relative_attention_bias` parameter name are the reference's, so its checkpoints load) with one more output form:

    * `forward(q, k, v)` -> `(q, k, v, bias)` with the dense `(1, H, M, N)` bias in q's dtype, like the reference (:103-110);
    * `forward_1d()` -> `(rpe1d, radius)`: the `(H, 2R+1)` fp32 generator of the linear-memory mode.  Build it ONCE per
      step and hand it to every layer's `flash_attention_v2_rpe1d`; the table gradient is then one accumulated scatter.

    `randomized_position` (reference :79-89: sorted random subsets of 0..max_sequence_length-1 as query / key positions)
    has no Toeplitz structure and is therefore served by the dense form only."""

    def __init__(self, relative_attention_num_buckets, relative_attention_max_distance, n_heads, max_sequence_length=0,
                 bidirectional=True, randomized_position=False):
        super().__init__()
        self.relative_attention_num_buckets = relative_attention_num_buckets
        self.relative_attention_max_distance = relative_attention_max_distance
        self.n_heads = n_heads
        self.max_sequence_length = max_sequence_length
        self.bidirectional = bidirectional
        self.randomized_position = randomized_position
        self.relative_attention_bias = torch.nn.Embedding(relative_attention_num_buckets, n_heads)

    def compute_bias(self, query_length, key_length, device=None):
        w = self.relative_attention_bias.weight
        if device is not None and w.device != torch.device(device):
            w = w.to(device)
        ctx = mem = None
        if self.randomized_position:  # context first, then memory: the reference's order of draws (:80-88)
            ctx = randomized_positions(self.max_sequence_length, query_length)
            mem = randomized_positions(self.max_sequence_length, key_length)
        return compute_bias(w, query_length, key_length, self.bidirectional, self.relative_attention_num_buckets,
                            self.relative_attention_max_distance, ctx, mem)

    def forward(self, q, k=None, v=None):
        query_length = q.shape[1]
        key_length = k.shape[1] if k is not None else query_length
        bias = self.compute_bias(query_length, key_length, device=q.device).contiguous().to(q.dtype)
        return q, k, v, bias

    def forward_1d(self):
        if self.randomized_position:
            raise NotImplementedError("randomized positions are not a function of n - m")
        r1 = rpe1d_from_table(self.relative_attention_bias.weight, self.bidirectional, self.relative_attention_num_buckets,
                              self.relative_attention_max_distance)
        return r1, rpe_radius(self.relative_attention_max_distance)

