"""Shared helpers for the GPU parity tests / tools (test infrastructure)."""
import math

import torch

import oracle


def make_inputs(B, H, M, N, D, dtype, bias_kind, seed=0, device="cuda", std=1.0, strided=False):
    g = torch.Generator(device="cpu").manual_seed(seed)

    def rnd(*shape):
        return (torch.randn(*shape, generator=g) * std).to(dtype).to(device)

    if strided:  # the model's real layout: (B,S,H,D) storage viewed as (B,H,S,D)  
        q = rnd(B, M, H, D).permute(0, 2, 1, 3)
        k = rnd(B, N, H, D).permute(0, 2, 1, 3)
        v = rnd(B, N, H, D).permute(0, 2, 1, 3)
        do = rnd(B, M, H, D).permute(0, 2, 1, 3)
    else:
        q, k, v, do = rnd(B, H, M, D), rnd(B, H, N, D), rnd(B, H, N, D), rnd(B, H, M, D)
    shape = {None: None, "1h": (1, H, M, N), "bh": (B, H, M, N), "11": (1, 1, M, N), "b1": (B, 1, M, N)}[bias_kind]
    b = rnd(*shape) if shape is not None else None
    return q, k, v, b, do


def maxdiff(a, b):
    """max |a - b|; positions where both hold the same value -- including the same +-inf (lse = -inf on fully masked rows) --
    count as 0; a NaN on either side, or an infinity facing a finite value, gives +inf so that every bound fails."""
    a, b = a.float(), b.float()
    d = (a - b).abs()
    d = torch.where(a == b, torch.zeros_like(d), d)
    d = torch.where(torch.isnan(d), torch.full_like(d, float("inf")), d)
    return d.max().item()


def eager_lowprec_errors(q, k, v, b, do, sm_scale, causal, ref):
    """Error of the eager low-precision path vs fp32 (the reference tests' yardstick, test_fa2_bias.py:26-28,64-67)."""
    leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)] + ([b.detach().clone().requires_grad_()] if b is not None else [])
    o = oracle.attn_ref(leaves[0], leaves[1], leaves[2], leaves[3] if b is not None else None, sm_scale, causal=causal, upcast=False)
    grads = torch.autograd.grad(o, leaves, do)
    out = {"o": maxdiff(o, ref["o"]), "dq": maxdiff(grads[0], ref["dq"]), "dk": maxdiff(grads[1], ref["dk"]),
           "dv": maxdiff(grads[2], ref["dv"])}
    if b is not None:
        out["db"] = maxdiff(grads[3], ref["db"])
    return out


def oracle_all(q, k, v, b, do, sm_scale, causal):
    o, L = oracle.attn_fwd_oracle(q, k, v, b, sm_scale, causal)
    dq, dk, dv, ds, db = oracle.attn_bwd_oracle(q, k, v, b, o, L, do, sm_scale, causal)
    return {"o": o, "L": L, "dq": dq, "dk": dk, "dv": dv, "db": db, "ds": ds}


def run_dense(q, k, v, b, do, sm_scale, causal):
    """Run the product path (HIP kernels through the C ABI) forward + backward; "L" = the forward's log-sum-exp output (part of
    the operator contract, reference flash_attention_v2_bias.py:59,:476), from the same forward entry point the backward reads it from."""
    from flasht5_amd import flash_attention_v2_bias
    from flasht5_amd.flash_attention_v2_bias import _attn_fwd
    leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)]
    bb = b.detach().clone().requires_grad_() if b is not None else None
    o = flash_attention_v2_bias(leaves[0], leaves[1], leaves[2], bb, causal, sm_scale)
    grads = torch.autograd.grad(o, leaves + ([bb] if bb is not None else []), do)
    qq, kk, vv = (t.detach() for t in leaves)
    o2, L = _attn_fwd(qq, kk, vv, b, None, 0, bool(causal), float(sm_scale if sm_scale is not None else 1.0 / math.sqrt(q.shape[-1])))
    assert torch.equal(o2, o.detach())  # (deterministic: the second forward is the first one bit for bit)
    out = {"o": o.detach(), "L": L, "dq": grads[0], "dk": grads[1], "dv": grads[2]}
    if bb is not None:
        out["db"] = grads[3]
    return out


def errors(got, ref):
    return {key: maxdiff(got[key], ref[key]) for key in got if key in ref and ref[key] is not None}


def tol_scale(ref_t):
    """bf16/fp16 output rounding grows with magnitude: tolerance = atol * max(1, max|ref|)."""
    return max(1.0, ref_t.float().abs().max().item())
