"""CPU restatement of the reference's eager attention and of the FA2 backward math.

TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Parity pin: `attn_ref` is checked against the imported reference
(`/root/reference/src/utils/attn_ref.py:3-29`) by `tests/golden/make_golden.py`
(run in the build container), whose outputs are frozen in `tests/golden/*.npz`
and re-checked by `tests/test_oracle_golden.py` on every run.

Each function cites the reference lines it restates.
"""
import math
import torch


def _expand_bias(b, B, H, M, N):
    # reference attn_ref.py:9-11 -- broadcast (1|B, 1|H, M, N) bias over batch / heads
    if b is None:
        return None
    return b.expand(B, H, M, N)


def causal_mask(M, N, device=None):
    """Bottom-right aligned causal mask: key n visible to query m iff m + (N - M) >= n.
    reference attn_ref.py:13-14,21-22 and flash_attention_v2_bias.py:447-449."""
    ms = torch.arange(M, device=device).unsqueeze(-1)
    ns = torch.arange(N, device=device)
    return (ms + (N - M)) >= ns


def attn_ref(q, k, v, b, sm_scale, causal=False, upcast=False):
    """Eager attention, same op order and dtypes as reference attn_ref.py:3-29
    (matmul in the input dtype -> scale -> +bias -> mask -> fp32 softmax -> cast -> matmul)."""
    if upcast:
        q, k, v = q.float(), k.float(), v.float()
        if b is not None:
            b = b.float()
    B, H, M, _ = q.shape
    N = k.shape[2]
    s = torch.matmul(q, k.transpose(2, 3))
    s = s * sm_scale
    if b is not None:
        s = s + _expand_bias(b, B, H, M, N)
    if causal:
        s = torch.where(causal_mask(M, N, q.device), s, torch.full_like(s, float("-inf")))
    p = torch.softmax(s.float(), dim=-1).to(q.dtype)
    return torch.matmul(p, v)


def attn_fwd_oracle(q, k, v, b, sm_scale, causal=False):
    """fp32 forward returning (o, L) with the kernel's conventions:
    L = m + ln(l) natural-log LSE (reference flash_attention_v2_bias.py:470-476);
    fully masked rows (causal, M > N) give o = 0, L = -inf (:470-473)."""
    qf, kf, vf = q.float(), k.float(), v.float()
    B, H, M, _ = q.shape
    N = k.shape[2]
    s = torch.matmul(qf, kf.transpose(2, 3)) * sm_scale
    if b is not None:
        s = s + _expand_bias(b.float(), B, H, M, N)
    if causal:
        s = s.masked_fill(~causal_mask(M, N, q.device), float("-inf"))
    L = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - L.unsqueeze(-1))
    p = torch.nan_to_num(p, nan=0.0)  # empty rows: exp(-inf - -inf)
    o = torch.matmul(p, vf)
    return o, L


def attn_bwd_oracle(q, k, v, b, o, L, do, sm_scale, causal=False):
    """fp32 backward with explicit FA2 formulas (no autograd):
    delta = rowsum(o*do)                      (reference :516-556)
    p  = exp(s - L)                           (:690)
    dv = p^T do                               (:702)
    dp = do v^T                               (:709-710)
    ds = p * (dp - delta)                     (:713)   gradient wrt the additive bias term
    dk = ds^T q * sm_scale                    (:722,:739)
    dq = ds k * sm_scale                      (:893,:901)
    dbias = ds reduced over the broadcast dims (:214-215; Q4: the mathematically
            correct head-sum is used for (1,1,M,N), unlike the reference's race).
    Returns dq, dk, dv, ds_full(B,H,M,N), dbias (shape of b or None)."""
    qf, kf, vf, of, dof = (t.float() for t in (q, k, v, o, do))
    B, H, M, _ = q.shape
    N = k.shape[2]
    s = torch.matmul(qf, kf.transpose(2, 3)) * sm_scale
    if b is not None:
        s = s + _expand_bias(b.float(), B, H, M, N)
    p = torch.exp(s - L.float().unsqueeze(-1))
    if causal:
        p = p.masked_fill(~causal_mask(M, N, q.device), 0.0)
    p = torch.nan_to_num(p, nan=0.0, posinf=0.0)
    delta = (of * dof).sum(-1, keepdim=True)
    dv = torch.matmul(p.transpose(2, 3), dof)
    dp = torch.matmul(dof, vf.transpose(2, 3))
    ds = p * (dp - delta)
    dk = torch.matmul(ds.transpose(2, 3), qf) * sm_scale
    dq = torch.matmul(ds, kf) * sm_scale
    dbias = None
    if b is not None:
        dbias = ds
        if b.shape[0] == 1 and B != 1:
            dbias = dbias.sum(0, keepdim=True)
        if b.shape[1] == 1 and H != 1:
            dbias = dbias.sum(1, keepdim=True)
    return dq, dk, dv, ds, dbias


def attn_varlen_oracle(q, k, v, cu_q, cu_k, sm_scale, causal=False):
    """Packed var-len attention = per-sequence loop of the eager path (config 4; the
    reference has no cu_seqlens path -- SURVEY 7.1 step 8 -- so this is the definition).
    q: (Tq, H, D), k/v: (Tk, H, D); cu_*: int lists of length nseq+1. Returns (Tq, H, D) fp32."""
    out = torch.zeros(q.shape, dtype=torch.float32)
    for i in range(len(cu_q) - 1):
        qs, qe, ks, ke = cu_q[i], cu_q[i + 1], cu_k[i], cu_k[i + 1]
        if qe == qs:
            continue
        qi = q[qs:qe].permute(1, 0, 2).unsqueeze(0)
        ki = k[ks:ke].permute(1, 0, 2).unsqueeze(0)
        vi = v[ks:ke].permute(1, 0, 2).unsqueeze(0)
        if ke == ks:
            continue
        oi, _ = attn_fwd_oracle(qi, ki, vi, None, sm_scale, causal)
        out[qs:qe] = oi[0].permute(1, 0, 2)
    return out


def attn_varlen_bwd_oracle(q, k, v, do, cu_q, cu_k, sm_scale, causal=False):
    """Gradients of `attn_varlen_oracle` (per-sequence loop of the FA2 backward formulas, `attn_bwd_oracle`).
    Returns dq (Tq, H, D), dk, dv (Tk, H, D) fp32; empty sequences contribute zeros."""
    dq = torch.zeros(q.shape, dtype=torch.float32)
    dk = torch.zeros(k.shape, dtype=torch.float32)
    dv = torch.zeros(v.shape, dtype=torch.float32)
    for i in range(len(cu_q) - 1):
        qs, qe, ks, ke = cu_q[i], cu_q[i + 1], cu_k[i], cu_k[i + 1]
        if qe == qs or ke == ks:
            continue
        qi = q[qs:qe].permute(1, 0, 2).unsqueeze(0)
        ki = k[ks:ke].permute(1, 0, 2).unsqueeze(0)
        vi = v[ks:ke].permute(1, 0, 2).unsqueeze(0)
        doi = do[qs:qe].permute(1, 0, 2).unsqueeze(0)
        oi, Li = attn_fwd_oracle(qi, ki, vi, None, sm_scale, causal)
        dqi, dki, dvi, _, _ = attn_bwd_oracle(qi, ki, vi, None, oi, Li, doi, sm_scale, causal)
        dq[qs:qe] = dqi[0].permute(1, 0, 2)
        dk[ks:ke] = dki[0].permute(1, 0, 2)
        dv[ks:ke] = dvi[0].permute(1, 0, 2)
    return dq, dk, dv

