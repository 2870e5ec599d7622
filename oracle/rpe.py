"""CPU restatement of the T5 relative-position bucket / bias builder.

TEST INFRASTRUCTURE ONLY.  Follows reference src/utils/positional_encoding.py:25-110.
Integer work is numpy int64; the log term is float32 exactly like the reference
(`relative_position.float()`, `.to(torch.long)` truncation, :61-65).

Pinned by the known answers in SURVEY 8(a8) and by fixtures generated from the imported
reference (tests/golden/make_golden.py).
"""
import math
import numpy as np
import torch


def relative_position_bucket(relative_position, bidirectional=True, num_buckets=32, max_distance=128):
    """reference positional_encoding.py:25-71.  relative_position = key_pos - query_pos (int array)."""
    rp = np.asarray(relative_position, dtype=np.int64)
    buckets = np.zeros_like(rp)
    nb = int(num_buckets)
    if bidirectional:
        nb //= 2                                            # :48
        buckets = buckets + (rp > 0).astype(np.int64) * nb  # :49
        rp = np.abs(rp)                                     # :50
    else:
        rp = -np.minimum(rp, 0)                             # :52
    max_exact = nb // 2                                     # :56
    is_small = rp < max_exact                               # :57
    # :60-65 -- fp32 log ratio, truncation toward zero.  rp == 0 -> log(0) = -inf -> the
    # reference's .to(long) yields INT64_MIN which is masked by is_small; guard it here.
    rpf = torch.from_numpy(np.maximum(rp, 1)).float()
    large = max_exact + (
        torch.log(rpf / max_exact) / torch.log(torch.tensor(max_distance / max_exact)) * (nb - max_exact)
    ).to(torch.long).numpy()
    large = np.minimum(large, nb - 1)                       # :66-68
    return buckets + np.where(is_small, rp, large)          # :70


def compute_bias(table, M, N, bidirectional=True, num_buckets=32, max_distance=128, context_position=None, memory_position=None):
    """reference positional_encoding.py:73-102.  table: (num_buckets, H) tensor.  Returns (1, H, M, N) in table dtype.
    context_position / memory_position: the query / key positions when they are not 0..M-1 / 0..N-1 (the reference's
    `randomized_position` branch, :79-89, draws sorted random subsets rooted at 0; the draw itself is not restated here --
    the fixtures carry the positions the reference drew)."""
    ctx = (np.arange(M, dtype=np.int64) if context_position is None else np.asarray(context_position, dtype=np.int64))[:, None]
    mem = (np.arange(N, dtype=np.int64) if memory_position is None else np.asarray(memory_position, dtype=np.int64))[None, :]
    bucket = relative_position_bucket(mem - ctx, bidirectional, num_buckets, max_distance)
    vals = table[torch.from_numpy(bucket)]          # (M, N, H)
    return vals.permute(2, 0, 1).unsqueeze(0)


def bias1d_from_table(table, M, N, bidirectional=True, num_buckets=32, max_distance=128):
    """The Toeplitz generator of compute_bias: bias[0,h,m,n] = bias1d[h, (n - m) + (M - 1)].
    Returns (H, M+N-1)."""
    delta = np.arange(-(M - 1), N, dtype=np.int64)
    bucket = relative_position_bucket(delta, bidirectional, num_buckets, max_distance)
    return table[torch.from_numpy(bucket)].transpose(0, 1).contiguous()


def toeplitz_from_bias1d(bias1d, M, N):
    """(H, M+N-1) -> (1, H, M, N)."""
    idx = (torch.arange(N)[None, :] - torch.arange(M)[:, None]) + (M - 1)
    return bias1d[:, idx].unsqueeze(0)


def table_grad_from_dbias1d(dbias1d, M, N, bidirectional=True, num_buckets=32, max_distance=128):
    """Scatter-add of the diagonal sums into the (num_buckets, H) table = what autograd's
    embedding backward does after the dense dbias."""
    delta = np.arange(-(M - 1), N, dtype=np.int64)
    bucket = torch.from_numpy(relative_position_bucket(delta, bidirectional, num_buckets, max_distance))
    H = dbias1d.shape[0]
    out = torch.zeros(num_buckets, H, dtype=dbias1d.dtype)
    out.index_add_(0, bucket, dbias1d.transpose(0, 1).contiguous())
    return out
