"""Generate the committed golden fixtures from the IMPORTED reference (build container only).

    TRITON_INTERPRET=1 python tests/golden/make_golden.py

* imports /root/reference (read-only) -- eager oracle `attn_ref`, `RelativePositionalEncoding`,
  `FlashT5LayerNorm` / `FlashT5CrossEntropyLoss` eager branches, and the Triton kernels themselves
  run under the Triton CPU interpreter (fp16/fp32 only; bf16 is broken in the interpreter);
* checks this repo's oracle/ restatement against them on the same seeded inputs (asserts);
* writes small .npz fixtures (inputs + expected outputs) next to this file.

The fixtures are DATA (arrays); no reference source text is stored.  /root/reference does not
exist on the GPU box -- only the .npz files travel.
"""
import os
import sys

os.environ.setdefault("TRITON_INTERPRET", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import math
import numpy as np
import torch

import oracle

from src.utils.attn_ref import attn_ref as ref_attn_ref                      # noqa: E402
from src.utils.positional_encoding import RelativePositionalEncoding        # noqa: E402
import src.model.ops.flash_attention_v2_bias as ref_fa                      # noqa: E402
import src.model.ops.rms_norm as ref_rms                                    # noqa: E402
import src.model.ops.cross_entropy_loss as ref_ce                           # noqa: E402
from src.model.modeling_flash_t5 import FlashT5LayerNorm, FlashT5CrossEntropyLoss  # noqa: E402


def npy(t):
    if t is None:
        return None
    if t.dtype == torch.bfloat16:
        # store bf16 as raw uint16 bit patterns
        return t.contiguous().view(torch.int16).numpy().view(np.uint16)
    return t.detach().contiguous().numpy()


def save(name, **arrs):
    arrs = {k: v for k, v in arrs.items() if v is not None}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path} ({os.path.getsize(path)/1024:.1f} KiB)")


def maxdiff(a, b):
    return (a.float() - b.float()).abs().max().item()


# ----------------------------------------------------------------------------------------------
# 1. T5 relative-position buckets / bias
# ----------------------------------------------------------------------------------------------
def gen_rpe():
    deltas = np.arange(-300, 301, dtype=np.int64)
    out = {}
    for bidir in (True, False):
        for nb, md in ((32, 128), (32, 256), (64, 128), (16, 64)):
            ref = RelativePositionalEncoding._relative_position_bucket(
                torch.from_numpy(deltas), bidirectional=bidir, num_buckets=nb, max_distance=md).numpy()
            mine = oracle.relative_position_bucket(deltas, bidir, nb, md)
            assert np.array_equal(ref, mine), (bidir, nb, md)
            out[f"bucket_{int(bidir)}_{nb}_{md}"] = ref.astype(np.int32)
    # known answers quoted in SURVEY 8(a8)
    d = np.arange(-200, 200, 25)
    assert oracle.relative_position_bucket(d, True, 32, 128).tolist() == \
        [15, 15, 15, 15, 15, 14, 13, 11, 0, 27, 29, 30, 31, 31, 31, 31]
    assert oracle.relative_position_bucket(d, False, 32, 128).tolist() == \
        [31, 31, 31, 31, 30, 27, 24, 19, 0, 0, 0, 0, 0, 0, 0, 0]

    torch.manual_seed(11)
    H = 2
    for bidir, (M, N) in ((True, (256, 256)), (True, (96, 160)), (False, (128, 128))):
        pe = RelativePositionalEncoding(32, 128, H, 512, bidirectional=bidir)
        with torch.no_grad():
            pe.relative_attention_bias.weight.normal_(0, 0.5)
        table = pe.relative_attention_bias.weight.detach().clone()     # (32, H)
        ref_bias = pe.compute_bias(M, N).detach()                       # (1,H,M,N)
        mine = oracle.compute_bias(table, M, N, bidir, 32, 128)
        assert torch.equal(ref_bias, mine)
        b1 = oracle.bias1d_from_table(table, M, N, bidir, 32, 128)
        assert torch.equal(oracle.toeplitz_from_bias1d(b1, M, N), ref_bias)
        out[f"table_{int(bidir)}_{M}_{N}"] = npy(table)
        out[f"bias1d_{int(bidir)}_{M}_{N}"] = npy(b1)
        out[f"bias_{int(bidir)}_{M}_{N}_row0"] = npy(ref_bias[0, :, 0, :])
        out[f"bias_{int(bidir)}_{M}_{N}_rowlast"] = npy(ref_bias[0, :, M - 1, :])
    # randomized positions (reference :79-89): the mirror module draws the same positions from the same seeded generator
    from flasht5_amd.positional_encoding import RelativePositionalEncoding as MirrorRPE, compute_bias as mirror_bias
    for bidir, (M, N, L) in ((True, (48, 80, 512)), (False, (64, 64, 256))):
        ref_pe = RelativePositionalEncoding(32, 128, H, L, bidirectional=bidir, randomized_position=True)
        with torch.no_grad():
            ref_pe.relative_attention_bias.weight.normal_(0, 0.5)
        mine_pe = MirrorRPE(32, 128, H, L, bidirectional=bidir, randomized_position=True)
        mine_pe.load_state_dict(ref_pe.state_dict())
        torch.manual_seed(1234 + M)
        ref_bias = ref_pe.compute_bias(M, N).detach()
        torch.manual_seed(1234 + M)
        assert torch.equal(mine_pe.compute_bias(M, N).detach(), ref_bias), ("randomized", bidir)
        # the positions themselves, recovered with the same draws, for the CPU test of the functional form
        torch.manual_seed(1234 + M)
        ctx, _ = torch.sort(torch.randperm(L)[:M]); ctx[0] = 0
        mem, _ = torch.sort(torch.randperm(L)[:N]); mem[0] = 0
        table = ref_pe.relative_attention_bias.weight.detach().clone()
        assert torch.equal(mirror_bias(table, M, N, bidir, 32, 128, ctx, mem), ref_bias)
        out[f"rand_table_{int(bidir)}"] = npy(table)
        out[f"rand_ctx_{int(bidir)}"] = ctx.numpy()
        out[f"rand_mem_{int(bidir)}"] = mem.numpy()
        out[f"rand_bias_{int(bidir)}"] = npy(ref_bias[0])
    out["deltas"] = deltas
    save("rpe_buckets", **out)


# ----------------------------------------------------------------------------------------------
# 2. attention: eager reference (fp32 + low precision) and autograd grads
# ----------------------------------------------------------------------------------------------
def attn_inputs(seed, B, H, M, N, D, dtype, bias_shape):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, H, M, D, generator=g).to(dtype)
    k = torch.randn(B, H, N, D, generator=g).to(dtype)
    v = torch.randn(B, H, N, D, generator=g).to(dtype)
    do = torch.randn(B, H, M, D, generator=g).to(dtype)
    b = None
    if bias_shape is not None:
        b = torch.randn(*bias_shape, generator=g).to(dtype)
    return q, k, v, b, do


def gen_attn_case(name, seed, B, H, M, N, D, dtype, bias_kind, causal, sm_scale, fwd_only=False):
    bias_shape = {None: None, "1h": (1, H, M, N), "bh": (B, H, M, N), "11": (1, 1, M, N)}[bias_kind]
    q, k, v, b, do = attn_inputs(seed, B, H, M, N, D, dtype, bias_shape)
    if dtype == torch.float32 and fwd_only:
        q, k, v, do = (t.bfloat16().float() for t in (q, k, v, do))
        b = b.bfloat16().float() if b is not None else None
    # --- reference eager, fp32 upcast, with autograd grads -------------------------------------
    # fp32 leaves so autograd's gradients are not rounded to the storage dtype
    leaves = [t.float().requires_grad_() for t in (q, k, v)] + ([b.float().requires_grad_()] if b is not None else [])
    qq, kk, vv = leaves[:3]
    bb = leaves[3] if b is not None else None
    o_ref = ref_attn_ref(qq, kk, vv, bb, sm_scale, causal=causal, upcast=True)
    larger_m_causal = causal and M > N
    if not larger_m_causal:
        grads = torch.autograd.grad(o_ref, leaves, do.float())
    else:
        grads = None  # reference eager gives NaN rows (Q2); pinned to the oracle's 0/-inf convention
    o_lp = ref_attn_ref(q, k, v, b, sm_scale, causal=causal, upcast=False)
    lp_err = [maxdiff(o_lp, o_ref.detach()) if not (causal and M > N) else 0.0, 0.0, 0.0, 0.0, 0.0]
    if larger_m_causal and dtype != torch.float32:
        # The reference's eager path NaNs on the M - N query rows that see no key (Q2).  Its low-precision error is still defined
        # on the rows that do: rows m >= M - N alone form the square causal problem (same keys, same bias rows), whose o, dq and
        # dbias rows are the full problem's rows and whose dk / dv are the full problem's (empty rows contribute nothing).
        lo = M - N
        qs, dos = q[:, :, lo:], do[:, :, lo:]
        bs = b[:, :, lo:] if b is not None else None
        lv32 = [t.float().requires_grad_() for t in (qs, k, v)] + ([bs.float().requires_grad_()] if b is not None else [])
        o32 = ref_attn_ref(lv32[0], lv32[1], lv32[2], lv32[3] if b is not None else None, sm_scale, causal=True, upcast=True)
        g32 = torch.autograd.grad(o32, lv32, dos.float())
        lvl = [t.clone().requires_grad_() for t in (qs, k, v)] + ([bs.clone().requires_grad_()] if b is not None else [])
        olp = ref_attn_ref(lvl[0], lvl[1], lvl[2], lvl[3] if b is not None else None, sm_scale, causal=True, upcast=False)
        glp = torch.autograd.grad(olp, lvl, dos)
        assert torch.isfinite(o32).all() and all(torch.isfinite(t).all() for t in g32)
        lp_err = [maxdiff(olp, o32.detach())] + [maxdiff(a_, b_) for a_, b_ in zip(glp, g32)] + [0.0] * (4 - len(g32))
    if dtype != torch.float32 and not (causal and M > N):
        lv = [t.clone().requires_grad_() for t in (q, k, v)] + ([b.clone().requires_grad_()] if b is not None else [])
        o_lp2 = ref_attn_ref(lv[0], lv[1], lv[2], lv[3] if b is not None else None, sm_scale, causal=causal, upcast=False)
        g_lp = torch.autograd.grad(o_lp2, lv, do)
        for i, gl in enumerate(g_lp):
            lp_err[1 + i] = maxdiff(gl, grads[i])
    # --- this repo's oracle must agree ----------------------------------------------------------
    o_mine, L_mine = oracle.attn_fwd_oracle(q, k, v, b, sm_scale, causal)
    o_mine_lp = oracle.attn_ref(q, k, v, b, sm_scale, causal=causal, upcast=False)
    if not larger_m_causal:
        assert maxdiff(o_mine, o_ref) < 2e-5, (name, maxdiff(o_mine, o_ref))
        assert torch.equal(o_mine_lp, o_lp), name
        assert maxdiff(oracle.attn_ref(q, k, v, b, sm_scale, causal=causal, upcast=True), o_ref) == 0.0
        dq, dk, dv, ds, dbias = oracle.attn_bwd_oracle(q, k, v, b, o_mine, L_mine, do, sm_scale, causal)
        tol = 2e-4 * max(1.0, math.sqrt(N / 64))  # fp32 summation-order noise on |g| ~ 10..50
        assert maxdiff(dq, grads[0]) < tol, (name, "dq", maxdiff(dq, grads[0]))
        assert maxdiff(dk, grads[1]) < tol, (name, "dk", maxdiff(dk, grads[1]))
        assert maxdiff(dv, grads[2]) < tol, (name, "dv", maxdiff(dv, grads[2]))
        if b is not None:
            assert maxdiff(dbias, grads[3]) < tol * 4, (name, "db", maxdiff(dbias, grads[3]))
    else:
        dq, dk, dv, ds, dbias = oracle.attn_bwd_oracle(q, k, v, b, o_mine, L_mine, do, sm_scale, causal)
        valid = torch.arange(M) + (N - M) >= 0
        assert maxdiff(o_mine[:, :, valid], o_ref.detach()[:, :, valid]) < 2e-5
    if fwd_only:
        save(name, q=npy(q.bfloat16()), k=npy(k.bfloat16()), v=npy(v.bfloat16()), bias=npy(b.bfloat16()),
             o=npy(o_mine), L=npy(L_mine), o_ref=npy(o_ref.detach()), eager_lp_err=np.array(lp_err, dtype=np.float64),
             meta=np.array([B, H, M, N, D, int(causal), 0]), sm_scale=np.array([sm_scale], dtype=np.float64))
        return
    save(name,
         q=npy(q), k=npy(k), v=npy(v), do=npy(do), bias=npy(b),
         o=npy(o_mine), L=npy(L_mine), eager_lp_err=np.array(lp_err, dtype=np.float64),
         o_ref=(npy(o_ref.detach()) if not larger_m_causal else None),  # the reference's own eager fp32 output
         dq=npy(grads[0] if grads else dq), dk=npy(grads[1] if grads else dk), dv=npy(grads[2] if grads else dv),
         dbias=npy(grads[3] if (grads and b is not None) else dbias),
         meta=np.array([B, H, M, N, D, int(causal), {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dtype]]),
         sm_scale=np.array([sm_scale], dtype=np.float64))


# ----------------------------------------------------------------------------------------------
# 3. attention: the reference's Triton kernels under the CPU interpreter (fp16)
# ----------------------------------------------------------------------------------------------
def run_triton_interp(q, k, v, b, do, causal, sm_scale, BM=32, BN=32):
    import triton
    B, H, M, D = q.shape
    N = k.shape[2]
    P_SEQ = N - M
    o = torch.empty_like(q)
    L = torch.empty(B, H, M, dtype=torch.float32)
    bbs = b.stride(0) if b.shape[0] == B else 0
    bhs = b.stride(1) if b.shape[1] == H else 0
    div_m, div_n = M % BM == 0, N % BN == 0
    grid = (triton.cdiv(M, BM), H, B)
    ref_fa._fwd_kernel[grid](
        q, k, v, b, sm_scale, L, o,
        *q.stride(), *k.stride(), *v.stride(), *o.stride(),
        bbs, bhs, b.stride(2), b.stride(3),
        B, H, M, N, P_SEQ,
        BLOCK_M=BM, BLOCK_N=BN, BLOCK_DMODEL=D, IS_CAUSAL=causal, LARGER_M=M > N,
        DIVISIBLE_M=div_m, DIVISIBLE_N=div_n, HAS_BIAS=True, num_warps=4, num_stages=1)
    delta = torch.empty_like(L)
    ref_fa._bwd_preprocess[grid](
        o, do, delta, *o.stride(), *do.stride(), *delta.stride(), M,
        BLOCK_M=BM, D_HEAD=D, DIVISIBLE_M=div_m)
    dk, dv, dq = torch.empty_like(k), torch.empty_like(v), torch.empty_like(q)
    batch_reduced = bbs == 0
    ds = torch.zeros((B, *b.shape[1:]), dtype=b.dtype)
    grid_kv = (triton.cdiv(N, BN), H, B)
    ref_fa._bwd_kv_kernel[grid_kv](
        q, k, v, b, sm_scale, do, dk, dv, ds, L, delta,
        *q.stride(), *k.stride(), *v.stride(),
        b.stride(0) if not batch_reduced else ds.stride(0), bhs, b.stride(2), b.stride(3),
        *do.stride(), *dk.stride(), *dv.stride(),
        B, H, M, N, P_SEQ, None,
        BLOCK_M=BM, BLOCK_DMODEL=D, BLOCK_N=BN, CAUSAL=causal,
        DIVISIBLE_M=div_m, DIVISIBLE_N=div_n, HAS_BIAS=True, RETURN_DS=True,
        IS_BATCH_REDUCED=batch_reduced, GROUP_SIZE_BIAS=B, num_stages=1, num_warps=4)
    ref_fa._bwd_q_kernel[grid](
        q, k, v, b, sm_scale, do, dq, L, delta,
        *q.stride(), *k.stride(), *v.stride(),
        bbs, bhs, b.stride(2), b.stride(3),
        *do.stride(), *dq.stride(),
        B, H, M, N, P_SEQ,
        BLOCK_M=BM, BLOCK_DMODEL=D, BLOCK_N=BN, CAUSAL=causal, LARGER_M=M > N,
        DIVISIBLE_M=div_m, DIVISIBLE_N=div_n, HAS_BIAS=True, num_stages=1, num_warps=4)
    if batch_reduced and B > 1:
        ds = ds.sum(0, keepdim=True)
    return o, L, dq, dk, dv, ds


def gen_triton_case(name, seed, B, H, M, N, D, bias_kind, causal, sm_scale, BM=32, BN=32):
    dtype = torch.float16
    bias_shape = {"1h": (1, H, M, N), "bh": (B, H, M, N)}[bias_kind]
    q, k, v, b, do = attn_inputs(seed, B, H, M, N, D, dtype, bias_shape)
    o, L, dq, dk, dv, ds = run_triton_interp(q, k, v, b, do, causal, sm_scale, BM, BN)
    o_mine, L_mine = oracle.attn_fwd_oracle(q, k, v, b, sm_scale, causal)
    gq, gk, gv, _, gb = oracle.attn_bwd_oracle(q, k, v, b, o_mine, L_mine, do, sm_scale, causal)
    errs = dict(o=maxdiff(o, o_mine), L=maxdiff(L, L_mine), dq=maxdiff(dq, gq), dk=maxdiff(dk, gk),
                dv=maxdiff(dv, gv), db=maxdiff(ds, gb))
    print(name, "triton-interp vs oracle:", {k_: f"{v_:.2e}" for k_, v_ in errs.items()})
    # fp16 outputs: allow output rounding (2^-11 relative) + fp16 P / dS rounding inside the kernel
    for key, ref_t in (("o", o_mine), ("dq", gq), ("dk", gk), ("dv", gv), ("db", gb)):
        assert errs[key] < 2e-3 * max(1.0, ref_t.abs().max().item()), (name, key, errs[key])
    assert errs["L"] < 1e-4
    save(name, q=npy(q), k=npy(k), v=npy(v), do=npy(do), bias=npy(b),
         o_triton=npy(o), L_triton=npy(L), dq_triton=npy(dq), dk_triton=npy(dk), dv_triton=npy(dv),
         dbias_triton=npy(ds), triton_vs_oracle_err=np.array([errs[k_] for k_ in ("o", "L", "dq", "dk", "dv", "db")]),
         meta=np.array([B, H, M, N, D, int(causal), 1]), sm_scale=np.array([sm_scale], dtype=np.float64))


# ----------------------------------------------------------------------------------------------
# 4. RMSNorm and CE: eager module branches + Triton kernels in the interpreter (fp32)
# ----------------------------------------------------------------------------------------------
def gen_rmsnorm():
    out = {}
    for tag, (rows, n) in (("a", (24, 768)), ("b", (7, 1024)), ("c", (5, 100))):
        g = torch.Generator().manual_seed(100 + rows)
        x = torch.randn(rows, n, generator=g)
        w = 1.0 + 0.1 * torch.randn(n, generator=g)
        dy = torch.randn(rows, n, generator=g)
        eps = 1e-6
        ln = FlashT5LayerNorm(n, eps=eps, use_triton_layernorm=False)
        with torch.no_grad():
            ln.weight.copy_(w)
        xx = x.clone().requires_grad_()
        y_ref = ln(xx)
        y_ref.backward(dy)
        dx_ref, dw_ref = xx.grad.clone(), ln.weight.grad.clone()
        y, rstd = oracle.rmsnorm_fwd_oracle(x, w, eps)
        dx, dw = oracle.rmsnorm_bwd_oracle(dy, x, w, rstd)
        assert maxdiff(y, y_ref) < 1e-5 and maxdiff(dx, dx_ref) < 1e-5 and maxdiff(dw, dw_ref) < 1e-4
        assert maxdiff(oracle.rmsnorm_eager(x, w, eps), y_ref) == 0.0
        # the Triton kernels under the interpreter
        Y = torch.empty_like(x)
        R = torch.empty(rows)
        import triton
        BN = triton.next_power_of_2(n)
        ref_rms._rmsnorm_fwd_kernel[(rows,)](x, Y, w, R, x.stride(0), Y.stride(0), n, eps, BN, n % BN == 0)
        progs = 4
        rpp = math.ceil(rows / progs)
        DX = torch.empty_like(x)
        DW = torch.zeros(progs, n)
        ref_rms._rmsnorm_bwd_kernel[(progs,)](x, w, dy, DX, DW, R, x.stride(0), dy.stride(0), DX.stride(0),
                                              rows, n, eps, rpp, BN, n % BN == 0)
        assert maxdiff(Y, y) < 1e-5 and maxdiff(R, rstd) < 1e-5
        assert maxdiff(DX, dx) < 1e-5 and maxdiff(DW.sum(0), dw) < 1e-4
        out.update({f"x_{tag}": npy(x), f"w_{tag}": npy(w), f"dy_{tag}": npy(dy), f"y_{tag}": npy(Y),
                    f"rstd_{tag}": npy(R), f"dx_{tag}": npy(DX), f"dw_{tag}": npy(DW.sum(0)),
                    f"y_eager_{tag}": npy(y_ref.detach()), f"dx_eager_{tag}": npy(dx_ref), f"dw_eager_{tag}": npy(dw_ref)})
    save("rmsnorm", **out)


def gen_ce():
    import triton
    out = {}
    cases = (("a", 16, 1000, 0.0, 0.0), ("b", 16, 1000, 0.1, 1e-4), ("c", 3, 32102, 0.1, 1e-4), ("d", 4, 4099, 0.0, 2.0))
    for tag, rows, V, smooth, zl in cases:
        g = torch.Generator().manual_seed(200 + V + rows)
        logits = torch.randn(rows, V, generator=g) * 2.0
        labels = torch.randint(0, V, (rows,), generator=g)
        labels[1] = -100
        labels[rows - 1] = -100
        dloss = torch.randn(rows, generator=g)
        loss, z, lse = oracle.ce_fwd_oracle(logits, labels, smooth, 1.0, zl, -100)
        dlg = oracle.ce_bwd_oracle(dloss, logits, lse, labels, smooth, 1.0, zl, -100)
        # reference Triton kernels under the interpreter
        BS = min(triton.next_power_of_2(V), 16 * 1024)
        losses = torch.empty(rows)
        zls = torch.empty(rows)
        lses = torch.empty(rows)
        ref_ce.cross_entropy_fwd_kernel[(rows,)](losses, lses, zls, logits, labels, smooth, 1.0, zl, -100, V, 0, V,
                                                 logits.stride(0), BLOCK_SIZE=BS, SPLIT=False, PRECOMPUTED_LSE=False)
        BS2 = min(triton.next_power_of_2(V), 4096)
        dl = torch.empty_like(logits)
        ref_ce.cross_entropy_bwd_kernel[(rows, triton.cdiv(V, BS2))](
            dl, dloss, logits, lses, labels, smooth, 1.0, zl, -100, V, 0, V,
            logits.stride(0), dl.stride(0), dloss.stride(0), BLOCK_SIZE=BS2)
        assert maxdiff(losses, loss) < 2e-4 * max(1.0, zl * 50) and maxdiff(lses, lse) < 1e-5 and maxdiff(zls, z) < 2e-4 * max(1, zl * 50)
        assert maxdiff(dl, dlg) < 1e-5 * max(1.0, zl * 50)
        # eager module semantics (mean over non-ignored; documented difference Q9) -- autograd pin of the math
        lg = logits.clone().requires_grad_()
        lsef = torch.logsumexp(lg, -1)
        keep = labels != -100
        per_row = torch.nn.functional.cross_entropy(lg, labels, reduction="none", label_smoothing=smooth, ignore_index=-100)
        per_row = per_row + torch.where(keep, zl * lsef * lsef, torch.zeros_like(lsef))
        assert maxdiff(per_row.detach(), loss) < 2e-4 * max(1.0, zl * 50), (tag, maxdiff(per_row.detach(), loss))
        per_row.backward(dloss)
        assert maxdiff(lg.grad, dlg) < 1e-5 * max(1.0, zl * 50)
        out.update({f"logits_{tag}": npy(logits), f"labels_{tag}": labels.numpy(), f"dloss_{tag}": npy(dloss),
                    f"loss_{tag}": npy(losses), f"z_{tag}": npy(zls), f"lse_{tag}": npy(lses), f"dlogits_{tag}": npy(dl),
                    f"cfg_{tag}": np.array([smooth, zl], dtype=np.float64)})
    save("cross_entropy", **out)


# ----------------------------------------------------------------------------------------------
# 6. AdamWScale optimizer step: the reference class on CPU vs the oracle restatement
# ----------------------------------------------------------------------------------------------
def gen_adamw():
    from src.utils.adamw_scaled import AdamWScale
    out = {}
    cases = [("fp32", torch.float32, False, 0.0), ("fp32_wd", torch.float32, False, 0.03), ("bf16", torch.bfloat16, False, 0.0),
             ("bf16_kahan_wd", torch.bfloat16, True, 0.03), ("fp16_kahan", torch.float16, True, 0.0),
             # round 3: `use_state_dtype` (:101-103) and `correct_bias=False` (:177)
             ("fp32_sbf16_wd", torch.float32, False, 0.03, torch.bfloat16, True), ("bf16_kahan_sfp16", torch.bfloat16, True, 0.0, torch.float16, True),
             ("bf16_plain", torch.bfloat16, False, 0.0, None, False), ("fp32_plain_wd", torch.float32, False, 0.03, None, False),
             ("fp16_kahan_plain_sbf16", torch.float16, True, 0.0, torch.bfloat16, False)]
    for case in cases:
        name, dtype, kahan, wd = case[:4]
        sdtype, correct = (case[4], case[5]) if len(case) > 4 else (None, True)
        g = torch.Generator().manual_seed(len(name) * 7 + 1)
        shapes = [(257, 33), (64,), (1, 12), (1000, 8)]
        p0 = [(torch.randn(*s, generator=g) * (0.02 if i != 2 else 1e-5)).to(dtype) for i, s in enumerate(shapes)]  # tensor 2: rms below the 1e-3 floor
        grads = [[(torch.randn(*s, generator=g) * 0.1).to(dtype) for s in shapes] for _ in range(3)]
        # the reference
        rp = [torch.nn.Parameter(t.clone()) for t in p0]
        opt = AdamWScale(rp, lr=0.01, betas=(0.9, 0.999), eps=1e-6, weight_decay=wd, kahan_sum=kahan, foreach=False,
                         correct_bias=correct, use_state_dtype=sdtype)
        for step in range(3):
            for t, gr in zip(rp, grads[step]):
                t.grad = gr.clone()
            opt.step()
        # the oracle
        mp = [t.clone() for t in p0]
        m = [torch.zeros_like(t, dtype=sdtype or t.dtype) for t in p0]
        v = [torch.zeros_like(t, dtype=sdtype or t.dtype) for t in p0]
        use_k = kahan and dtype in (torch.float16, torch.bfloat16)
        kc = [torch.zeros_like(t) if use_k else None for t in p0]
        for step in range(3):
            for i in range(len(mp)):
                oracle.adamw_scale_step(mp[i], grads[step][i].clone(), m[i], v[i], kc[i], step + 1, 0.01, 0.9, 0.999, wd, 1e-6, correct)
        for i in range(len(mp)):
            st = opt.state[rp[i]]
            assert torch.equal(mp[i], rp[i].detach()), (name, i, "p")
            assert torch.equal(m[i], st["exp_avg"]) and torch.equal(v[i], st["exp_avg_sq"]), (name, i, "state")
            if use_k:
                assert torch.equal(kc[i], st["kahan_comp"]), (name, i, "kahan")
            out[f"{name}__p0_{i}"] = npy(p0[i])
            out[f"{name}__p_{i}"] = npy(mp[i])
            out[f"{name}__m_{i}"] = npy(m[i])
            out[f"{name}__v_{i}"] = npy(v[i])
            if use_k:
                out[f"{name}__k_{i}"] = npy(kc[i])
            for step in range(3):
                out[f"{name}__g{step}_{i}"] = npy(grads[step][i])
        code = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
        out[f"{name}__cfg"] = np.array([code[dtype], int(kahan), wd, 0.01, 0.9, 0.999, 1e-6, code[sdtype] if sdtype is not None else -1, int(correct)])
    save("adamw_scale", **out)


def gen_triton_round3():
    """head dims 128 and 32, and M > N non-causal with a per-batch (B, H, M, N) bias (parity hole iii)"""
    gen_triton_case("triton_d128_nc_1h_fp16", 24, 1, 2, 64, 72, 128, "1h", False, 1.0 / math.sqrt(128))  # (the default scale)
    gen_triton_case("triton_d32_c_1h_fp16", 25, 2, 2, 64, 64, 32, "1h", True, 0.5)
    gen_triton_case("triton_mgtn_nc_bh_fp16", 26, 2, 2, 96, 48, 64, "bh", False, 1.0)


def gen_gated_act():
    """the reference's FlashT5DenseGatedAct (modeling_flash_t5.py:128-146) itself, GELU(tanh) and ReLU, fp32 and bf16: the two
    projection outputs (captured by forward hooks), the module output and autograd's gradients of the projections"""
    from src.model.configuration_flash_t5 import FlashT5Config
    from src.model.modeling_flash_t5 import FlashT5DenseGatedAct
    out = {}
    for act in ("gelu_tanh", "relu"):
        for dt, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
            torch.manual_seed(40 + len(out))
            cfg = FlashT5Config(d_model=64, d_ff=128, dropout_rate=0.0, use_gelu_act=(act == "gelu_tanh"), use_glu_mlp=True)
            m = FlashT5DenseGatedAct(cfg).to(dt).eval()
            with torch.no_grad():  # projections of a few units spread: both GELU tails and the region around 0
                m.wi_0.weight.mul_(6.0)
            x = torch.randn(24, 64).to(dt)
            cap = {}
            def grab(key):
                def hook(mod, inp, o):
                    o.retain_grad()
                    cap[key] = o
                return hook
            hooks = [m.wi_0.register_forward_hook(grab("h0")), m.wi_1.register_forward_hook(grab("h1"))]
            y = m(x)
            dy = torch.randn(y.shape).to(dt)
            y.backward(dy)
            for h in hooks:
                h.remove()
            k = f"{act}_{tag}_"
            for name, t in (("h0", cap["h0"]), ("h1", cap["h1"]), ("out", y), ("dout", dy), ("dh0", cap["h0"].grad), ("dh1", cap["h1"].grad)):
                out[k + name] = t.detach().float().numpy()  # (bf16 values are exact in fp32)
            e = (oracle.gated_act_oracle(cap["h0"].detach(), cap["h1"].detach(), act) - y.detach().double()).abs().max().item()
            print(f"gated_act {act} {tag}: |oracle - reference| = {e:.3e}")
    save("gated_act", **out)


def gen_norm_linear():
    """The reference's own `normed = layer_norm(hidden)` -> Wq / Wk / Wv sequence (FlashT5LayerNorm eager branch followed by three
    bias-free nn.Linear, modeling_flash_t5.py:95-112, :226-231, :304-318) and `hidden + o(attn)` (:316), fp32 and bf16, with
    autograd's gradients: pins oracle/fused_linear.py as a unit (parity iii)."""
    out = {}
    for dt, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        torch.manual_seed(77)
        rows, K, n = 24, 128, 64
        ln = FlashT5LayerNorm(K, eps=1e-6, use_triton_layernorm=False).to(dt)
        with torch.no_grad():
            ln.weight.copy_((1.0 + 0.1 * torch.randn(K)).to(dt))
        lins = [torch.nn.Linear(K, n, bias=False).to(dt) for _ in range(3)]
        wo = torch.nn.Linear(n, K, bias=False).to(dt)
        x = torch.randn(rows, K).to(dt).requires_grad_()
        normed = ln(x).type_as(x)
        qkv = torch.cat([l(normed) for l in lins], -1)
        dqkv = torch.randn(qkv.shape).to(dt)
        qkv.backward(dqkv)
        a = torch.randn(rows, n).to(dt).requires_grad_()
        res = torch.randn(rows, K).to(dt).requires_grad_()
        y = res + wo(a)
        dy = torch.randn(y.shape).to(dt)
        y.backward(dy)
        W = torch.cat([l.weight for l in lins], 0).detach()
        ref_o, _ = oracle.rmsnorm_linear_oracle(x.detach(), ln.weight.detach(), W, 1e-6)
        ref_r = oracle.rmsnorm_linear_reference_rounding(x.detach(), ln.weight.detach(), W, 1e-6)
        print(f"norm_linear {tag}: |oracle - reference| = {maxdiff(ref_o, qkv.detach()):.3e}, with the reference's rounding {maxdiff(ref_r, qkv.detach()):.3e};"
              f" residual {maxdiff(oracle.linear_residual_oracle(a.detach(), wo.weight.detach(), res.detach()), y.detach()):.3e}")
        if dt == torch.float32:
            assert maxdiff(ref_o, qkv.detach()) < 1e-4 and maxdiff(oracle.linear_residual_oracle(a.detach(), wo.weight.detach(), res.detach()), y.detach()) < 1e-5
        else:
            assert maxdiff(ref_r, qkv.detach()) <= 2.0 ** -7 * float(qkv.detach().float().abs().max())  # (one output rounding)
        k = tag + "_"
        f = lambda t: t.detach().float().numpy()   # (bf16 values are exact in fp32)
        out.update({k + "x": f(x), k + "g": f(ln.weight), k + "W": f(W), k + "qkv": f(qkv), k + "dqkv": f(dqkv), k + "dx": f(x.grad),
                    k + "dg": f(ln.weight.grad), k + "dW": f(torch.cat([l.weight.grad for l in lins], 0)),
                    k + "a": f(a), k + "wo": f(wo.weight), k + "res": f(res), k + "y": f(y), k + "dy": f(dy), k + "da": f(a.grad),
                    k + "dwo": f(wo.weight.grad), k + "dres": f(res.grad)})
    save("norm_linear", **out)


def main():
    torch.set_num_threads(8)
    only = set(sys.argv[sys.argv.index("--only") + 1].split(",")) if "--only" in sys.argv else None
    if only is not None:  # regenerate a subset: `--only adamw,triton3,mgtn,gated` (every generator is seeded: the others are unchanged)
        f16 = torch.float16
        if "adamw" in only:
            gen_adamw()
        if "mgtn" in only:
            gen_attn_case("attn_mgtn_c_1h_fp16", 5, 2, 2, 96, 64, 64, f16, "1h", True, 1.0)
        if "triton3" in only:
            gen_triton_round3()
        if "gated" in only:
            gen_gated_act()
        if "normlin" in only:
            gen_norm_linear()
        return
    gen_rpe()
    f32, f16, bf16 = torch.float32, torch.float16, torch.bfloat16
    # cfg1: t5-small encoder self-attn fwd (2,8,128,64) fp32 eager CPU numerics baseline
    gen_attn_case("attn_cfg1_fp32", 1, 2, 8, 128, 128, 64, f32, "1h", False, 1.0 / math.sqrt(8), fwd_only=True)
    # tails / M != N / both bias shapes / causal, low precision
    gen_attn_case("attn_t164_nc_1h_bf16", 2, 2, 2, 128, 164, 64, bf16, "1h", False, 1.0)
    gen_attn_case("attn_t164_nc_1h_fp16", 2, 1, 2, 128, 164, 64, f16, "1h", False, 1.0)
    gen_attn_case("attn_t100_c_bh_bf16", 3, 2, 2, 72, 100, 64, bf16, "bh", True, 1.0)
    gen_attn_case("attn_t100_c_bh_fp16", 3, 2, 2, 72, 100, 64, f16, "bh", True, 1.0)
    gen_attn_case("attn_mgtn_nc_bh_bf16", 4, 2, 2, 96, 64, 64, bf16, "bh", False, 1.0)
    gen_attn_case("attn_mgtn_c_1h_fp16", 5, 2, 2, 96, 64, 64, f16, "1h", True, 1.0)
    gen_attn_case("attn_nobias_c_bf16", 6, 1, 2, 80, 80, 64, bf16, None, True, 0.125)
    gen_attn_case("attn_11_nc_bf16", 7, 2, 3, 64, 96, 64, bf16, "11", False, 1.0)
    gen_attn_case("attn_d128_nc_bf16", 8, 1, 2, 64, 100, 128, bf16, "1h", False, 1.0)
    gen_attn_case("attn_d32_c_fp16", 9, 1, 2, 72, 72, 32, f16, "1h", True, 1.0)
    # the reference's Triton kernels (interpreter, fp16)
    gen_triton_case("triton_t80_nc_1h_fp16", 21, 2, 2, 64, 80, 64, "1h", False, 1.0)
    gen_triton_case("triton_t80_c_bh_fp16", 22, 2, 2, 64, 80, 64, "bh", True, 1.0)
    gen_triton_case("triton_t100_nc_1h_fp16", 23, 1, 2, 96, 100, 64, "1h", False, 0.5)
    gen_triton_round3()
    gen_rmsnorm()
    gen_ce()
    gen_adamw()
    gen_gated_act()
    gen_norm_linear()


if __name__ == "__main__":
    main()
