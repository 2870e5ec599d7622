"""Random problems for the PUBLIC operators at the library's own dispatch (no variant bits), shared by tools/fuzz_api.py and tests/test_fuzz_api_gpu.py (test
infrastructure): `flash_attention_v2_bias(q, k, v, bias, causal, sm_scale)` -- the reference's operator, every bias broadcast form it accepts ((1|B, 1|H, M, N),
reference flash_attention_v2_bias.py:45-52) or none -- and `flash_attention_v2_rpe` (T5 table in-kernel), through autograd: head_dim 16 / 32 / 64 / 128, 1 .. 700 rows
and keys (any remainder, M != N, single rows), batch 1 .. 5, 1 .. 4 heads, bf16 / fp16, (B,S,H,D)-strided or contiguous operands, causal or not, several scales, and --
dense mode -- masks the way the model builds them (finfo.min or -inf added into the bias for padded keys: modeling_flash_t5.py:267-277).  o, dq, dk, dv, dbias / dtable
against the fp32 oracle with the bounds of tests/test_attention_gpu.py; a tensor past the bound must still satisfy the reference's own rule (at most twice the error of
eager attention in the input dtype, tests/fa2_triton/test_fa2_bias.py:64-67).
"""
import torch
import oracle
from flasht5_amd import flash_attention_v2_bias, flash_attention_v2_rpe
from attn_helpers import make_inputs, oracle_all, maxdiff, eager_lowprec_errors
from test_attention_gpu import bound, gbound


def run_case(i, rng, large=False):
    """case i drawn from `rng` (random.Random): returns (description, list of failures -- empty = passed).  large: production-sized problems (4 .. 96 (batch, head)
    pairs, 256 .. 2304 rows / keys, mostly head_dim 64) -- the sizes at which the library's own dispatch reaches the 64-wide pipelined bodies and their launch forms"""
    if large:
        D = rng.choice([64, 64, 64, 128])
        B, H = rng.randint(1, 8), rng.choice([4, 8, 12, 12, 16])
        M = rng.randint(256, 2304)
        N = rng.randint(256, 2304) if rng.random() < 0.5 else M
        if rng.random() < 0.5:
            M, N = (M + 127) // 128 * 128, (N + 127) // 128 * 128
    else:
        D = rng.choice([16, 32, 64, 64, 64, 128])
        B, H = rng.randint(1, 5), rng.randint(1, 4)
        big = rng.random() < 0.6
        M = rng.randint(1, 700) if big else rng.randint(1, 70)
        N = rng.randint(1, 700) if big else rng.randint(1, 70)
        if rng.random() < 0.3:
            N = M
    causal = rng.random() < 0.4
    dtype = torch.bfloat16 if rng.random() < 0.7 else torch.float16
    scale = rng.choice([None, 0.125, 0.25, 1.0 / 3, 1.0, 1.3])
    kind = rng.choice(["none", "1h", "1h", "bh", "11", "b1", "rpe", "rpe"])
    strided = rng.random() < 0.5
    q, k, v, b, do = make_inputs(B, H, M, N, D, dtype, kind if kind not in ("none", "rpe") else None, seed=3000 + i, strided=strided)
    sc = scale if scale is not None else D ** -0.5
    if sc > 0.5:
        q = (q.float() * 0.35).to(dtype)
    mask = ""
    md = 128
    table = None
    if kind == "rpe":
        md = rng.choice([32, 64, 128, 128, 256])
        bidir = rng.random() < 0.7
        table = torch.randn(32, H, generator=torch.Generator().manual_seed(i)) * 0.5
        b = oracle.compute_bias(table, M, N, bidir, 32, md).contiguous().cuda()  # (fp32: the table mode never rounds the bias)
    elif b is not None and rng.random() < 0.35 and N > 1:
        # padded keys: the last `pad` keys of some batch elements (or of all, with a shared bias) masked the way the model does it
        neg = rng.choice([float("-inf"), torch.finfo(dtype).min])
        pad = rng.randint(1, max(1, N // 2))
        b = b.clone()
        if b.shape[0] > 1:
            b[rng.randrange(b.shape[0]), :, :, N - pad:] = neg
        else:
            b[:, :, :, N - pad:] = neg
        mask = f" mask={neg:.3g}x{pad}"
    ref = oracle_all(q, k, v, b, do, sc, causal)
    leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)]
    try:
        if kind == "rpe":
            tb = table.cuda().requires_grad_()
            o = flash_attention_v2_rpe(leaves[0], leaves[1], leaves[2], tb, bidir, 32, md, causal, scale)
            dq, dk, dv, dt = torch.autograd.grad(o, leaves + [tb], do)
        elif b is not None:
            bb = b.detach().clone().requires_grad_()
            o = flash_attention_v2_bias(leaves[0], leaves[1], leaves[2], bb, causal, scale)
            dq, dk, dv, db = torch.autograd.grad(o, leaves + [bb], do)
        else:
            o = flash_attention_v2_bias(leaves[0], leaves[1], leaves[2], None, causal, scale)
            dq, dk, dv = torch.autograd.grad(o, leaves, do)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        return f"D={D} B={B} H={H} M={M} N={N} causal={int(causal)} {kind}", [f"raised {type(e).__name__}: {e}"]
    got = {"o": o.detach(), "dq": dq, "dk": dk, "dv": dv}
    msgs = []
    lp = None
    fa2 = None
    for key in ("o", "dq", "dk", "dv") + (("db",) if kind not in ("none", "rpe") else ()):
        g = db if key == "db" else got[key]
        r = ref[key]
        if key == "db":
            r = r.to(torch.float32)
            if tuple(r.shape) != tuple(g.shape):  # (the oracle returns dS summed to the bias shape already; guard all the same)
                msgs.append(f"db shape {tuple(g.shape)} vs {tuple(r.shape)}")
                continue
        # a sum over b batch elements / h heads of rounded dS: one output rounding of a value up to (b h) times larger, b h times the roundings inside
        nsum = 1
        if key == "db":
            nsum = (B if b.shape[0] == 1 else 1) * (H if b.shape[1] == 1 else 1)
        lim = bound(r, dtype) if key == "o" else gbound(r, dtype) * (1 + (nsum if nsum > 1 else 0))
        # masked entries: finite-or-identical is all that can be asked of a gradient at a -inf bias (maxdiff treats equal infinities / exact zeros as 0)
        e = maxdiff(g, r)
        fin = torch.isfinite(g.float()).all()
        if fin and e <= lim:
            continue
        if lp is None:
            bl = b.to(dtype) if b is not None else None
            lp = eager_lowprec_errors(q, k, v, bl, do, sc, causal, ref)
        if not fin or e > 2 * lp.get(key, 0.0) * (nsum if key == "db" else 1) + 1e-5:
            # third yardstick, gradients only: the FA2 backward DEFINES delta = rowsum(o * do) on the STORED (rounded) o (reference _bwd_preprocess reads the 16-bit o,
            # flash_attention_v2_bias.py:516-556) -- on peaked rows (dP' ~ delta) that differs from the fp32-o gradient by ~2^-9 |o| |do| sqrt(D) per dS element, which
            # neither bound above models.  The same formulas fed with the kernel's own o are the reference kernel's arithmetic in fp32:
            if fin and key != "o":
                if fa2 is None:
                    dq2, dk2, dv2, _, db2 = oracle.attn_bwd_oracle(q, k, v, b, got["o"], ref["L"], do, sc, causal)
                    fa2 = {"dq": dq2, "dk": dk2, "dv": dv2, "db": db2}
                e2 = maxdiff(g, fa2[key].to(torch.float32))
                if e2 <= lim:
                    continue
            msgs.append(f"{key} {e:.3e} > {lim:.3e} and > 2 x eager {lp.get(key, float('nan')):.3e}" + ("" if fin else " (non-finite)"))
    if kind == "rpe":
        _, _, _, _, db_alg = oracle.attn_bwd_oracle(q, k, v, b, o.detach(), ref["L"], do, sc, causal)
        tl = table.clone().requires_grad_()
        oracle.compute_bias(tl, M, N, bidir, 32, md).backward(db_alg.cpu())
        want = tl.grad
        # (truth as in tests/test_attention_gpu.py::test_rpe_mode_matches_dense_oracle: the oracle's dS with delta formed from the STORED o, pushed through compute_bias)
        err = (dt.detach().cpu() - want).abs().max().item()
        lim = 2e-2 * max(1.0, want.abs().max().item())
        if not torch.isfinite(dt).all() or err > lim:
            msgs.append(f"dtable {err:.3e} > {lim:.3e}")
    desc = (f"D={D} B={B} H={H} M={M} N={N} causal={int(causal)} {kind}{' md=%d' % md if kind == 'rpe' else ''} {str(dtype)[6:]} scale={sc:.3f} "
            f"strided={int(strided)}{mask}")
    return desc, msgs


def run_varlen_case(i, rng):
    """Packed batches (`flash_attn_varlen_func`, SURVEY 8(f) n2 -- the reference pads instead, data_collator_ul2.py:49-87): 1 .. 7 sequences of 0 .. 400 tokens (empty and
    one-token sequences included), cross-attention (independent key lengths) without bias or self-attention with the in-kernel T5 bias (positions local to each
    sequence), head_dim 16 / 32 / 64 / 128, causal or not; o, dq, dk, dv (and the generator gradient) against the per-sequence fp32 oracle."""
    from flasht5_amd import flash_attn_varlen_func
    D = rng.choice([16, 32, 64, 64, 128])
    H = rng.randint(1, 6)
    nseq = rng.randint(1, 7)
    causal = rng.random() < 0.4
    with_rpe = rng.random() < 0.4
    scale = rng.choice([0.125, 0.25, D ** -0.5])
    lens_q = [rng.choice([0, 1, rng.randint(2, 60), rng.randint(61, 400)]) for _ in range(nseq)]
    lens_k = list(lens_q) if with_rpe else [rng.choice([0, 1, rng.randint(2, 60), rng.randint(61, 400)]) for _ in range(nseq)]
    if sum(lens_q) == 0:
        lens_q[0] = 17
        if with_rpe:
            lens_k[0] = 17
    if sum(lens_k) == 0:
        lens_k[0] = 23
    if causal:  # (bottom-right aligned mask: more rows than keys would leave rows without a visible key -- NaN in the eager reference)
        lens_q = [min(a_, b_) if b_ > 0 else a_ for a_, b_ in zip(lens_q, lens_k)]
        if sum(lens_q) == 0:
            lens_q[0] = lens_k[0] = 9
    cu_q, cu_k = [0], [0]
    for a_, b_ in zip(lens_q, lens_k):
        cu_q.append(cu_q[-1] + a_)
        cu_k.append(cu_k[-1] + b_)
    g = torch.Generator().manual_seed(5000 + i)
    dtype = torch.bfloat16
    q = torch.randn(cu_q[-1], H, D, generator=g).to(dtype).cuda()
    k = torch.randn(cu_k[-1], H, D, generator=g).to(dtype).cuda()
    v = torch.randn(cu_k[-1], H, D, generator=g).to(dtype).cuda()
    do = torch.randn(cu_q[-1], H, D, generator=g).to(dtype).cuda()
    R = rng.choice([8, 32, 128])
    r1 = (torch.randn(H, 2 * R + 1, generator=g) * 0.5).cuda()
    desc = f"varlen D={D} H={H} lens_q={lens_q} lens_k={lens_k} causal={int(causal)} {'rpe R=%d' % R if with_rpe else 'none'} scale={scale:.3f}"
    leaves = [t.clone().requires_grad_() for t in (q, k, v)] + ([r1.clone().requires_grad_()] if with_rpe else [])
    cq, ck = torch.tensor(cu_q, dtype=torch.int32).cuda(), torch.tensor(cu_k, dtype=torch.int32).cuda()
    try:
        if with_rpe:
            o = flash_attn_varlen_func(leaves[0], leaves[1], leaves[2], cq, ck, max(lens_q), max(lens_k), causal, scale, leaves[3], R)
        else:
            o = flash_attn_varlen_func(leaves[0], leaves[1], leaves[2], cq, ck, max(lens_q), max(lens_k), causal, scale)
        grads = torch.autograd.grad(o, leaves, do)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        return desc, [f"raised {type(e).__name__}: {e}"]
    # per-sequence fp32 truth through autograd of the eager restatement (bias gathered from the same generator)
    ref_leaves = [t.float().clone().requires_grad_() for t in (q, k, v)] + ([r1.clone().requires_grad_()] if with_rpe else [])
    ref_o = torch.zeros(cu_q[-1], H, D, device="cuda")
    total = None
    for s in range(nseq):
        qs, qe, ks, ke = cu_q[s], cu_q[s + 1], cu_k[s], cu_k[s + 1]
        if qe == qs or ke == ks:
            continue
        qi, ki, vi = ref_leaves[0][qs:qe].permute(1, 0, 2).unsqueeze(0), ref_leaves[1][ks:ke].permute(1, 0, 2).unsqueeze(0), ref_leaves[2][ks:ke].permute(1, 0, 2).unsqueeze(0)
        bias = None
        if with_rpe:
            idx = torch.clamp(torch.arange(ke - ks)[None, :] - torch.arange(qe - qs)[:, None], -R, R).cuda() + R
            bias = ref_leaves[3][:, idx].unsqueeze(0)
        oi = oracle.attn_ref(qi, ki, vi, bias, scale, causal=causal, upcast=True)[0].permute(1, 0, 2)
        ref_o[qs:qe] = oi.detach()
        t_ = (oi * do[qs:qe].float()).sum()
        total = t_ if total is None else total + t_
    if total is None:
        ref_grads = [torch.zeros_like(t) for t in ref_leaves]
    else:
        ref_grads = [g_ if g_ is not None else torch.zeros_like(t) for g_, t in zip(torch.autograd.grad(total, ref_leaves, allow_unused=True), ref_leaves)]
    msgs = []
    lowp = None

    def lowp_errors():
        """the reference tests' yardstick (tests/fa2_triton/test_fa2_bias.py:64-67): the error of eager attention in the input dtype, per sequence"""
        lv = [t.clone().requires_grad_() for t in (q, k, v)]
        lo = torch.zeros(cu_q[-1], H, D, device="cuda")
        tot = None
        for s_ in range(nseq):
            qs, qe, ks, ke = cu_q[s_], cu_q[s_ + 1], cu_k[s_], cu_k[s_ + 1]
            if qe == qs or ke == ks:
                continue
            bias = None
            if with_rpe:
                idx = torch.clamp(torch.arange(ke - ks)[None, :] - torch.arange(qe - qs)[:, None], -R, R).cuda() + R
                bias = r1[:, idx].unsqueeze(0).to(dtype)
            oi = oracle.attn_ref(lv[0][qs:qe].permute(1, 0, 2).unsqueeze(0), lv[1][ks:ke].permute(1, 0, 2).unsqueeze(0), lv[2][ks:ke].permute(1, 0, 2).unsqueeze(0),
                                 bias, scale, causal=causal, upcast=False)[0].permute(1, 0, 2)
            lo[qs:qe] = oi.detach().float()
            t_ = (oi.float() * do[qs:qe].float()).sum()
            tot = t_ if tot is None else tot + t_
        gl = torch.autograd.grad(tot, lv) if tot is not None else [torch.zeros_like(t) for t in lv]
        return {"o": maxdiff(lo, ref_o), "dq": maxdiff(gl[0], ref_grads[0]), "dk": maxdiff(gl[1], ref_grads[1]), "dv": maxdiff(gl[2], ref_grads[2])}

    for got, ref, key, lim in [(o, ref_o, "o", bound(ref_o, dtype))] + [(g_, r_, k_, gbound(r_, dtype)) for g_, r_, k_ in zip(grads[:3], ref_grads[:3], ("dq", "dk", "dv"))]:
        e = maxdiff(got, ref)
        if torch.isfinite(got.float()).all() and e <= lim:
            continue
        if lowp is None and torch.isfinite(got.float()).all():
            lowp = lowp_errors()
        if not torch.isfinite(got.float()).all() or e > 2 * lowp[key] + 1e-5:
            msgs.append(f"{key} {e:.3e} > {lim:.3e}" + (f" and > 2 x eager {lowp[key]:.3e}" if lowp is not None else " (non-finite)"))
    if with_rpe:
        lim = 1e-2 * max(1.0, ref_grads[3].abs().max().item()) + 3e-2
        e = maxdiff(grads[3], ref_grads[3])
        if torch.isfinite(grads[3]).all() and e > lim:
            # the FA2 definition (see run_case): dS with delta = rowsum(o * do) on the STORED o, summed along its diagonals -- far bins of a small radius collect
            # thousands of dS elements whose per-row delta offsets do not cancel
            alt = torch.zeros_like(ref_grads[3])
            for s_ in range(nseq):
                qs, qe, ks, ke = cu_q[s_], cu_q[s_ + 1], cu_k[s_], cu_k[s_ + 1]
                if qe == qs or ke == ks:
                    continue
                idx = torch.clamp(torch.arange(ke - ks)[None, :] - torch.arange(qe - qs)[:, None], -R, R).cuda() + R
                bias = r1[:, idx].unsqueeze(0)
                qi, ki, vi = q[qs:qe].permute(1, 0, 2).unsqueeze(0), k[ks:ke].permute(1, 0, 2).unsqueeze(0), v[ks:ke].permute(1, 0, 2).unsqueeze(0)
                oi, Li = oracle.attn_fwd_oracle(qi, ki, vi, bias, scale, causal)
                _, _, _, ds, _ = oracle.attn_bwd_oracle(qi, ki, vi, bias, o[qs:qe].detach().permute(1, 0, 2).unsqueeze(0), Li, do[qs:qe].permute(1, 0, 2).unsqueeze(0), scale, causal)
                alt.index_add_(1, idx.reshape(-1), ds[0].reshape(H, -1).float())
            e = maxdiff(grads[3], alt)
        if not torch.isfinite(grads[3]).all() or e > lim:
            msgs.append(f"drpe1d {e:.3e} > {lim:.3e}")
    return desc, msgs


def run_rowwise_case(i, rng):
    """The two bandwidth-bound operators around the attention path: `fast_rms_layernorm(X, W, eps)` (reference rms_norm.py:250-287) and `cross_entropy_loss(...)`
    with label smoothing / z-loss / logit scale / ignored and out-of-range labels (cross_entropy_loss.py:280-426) on random shapes -- 1 .. 3000 rows, any width --
    against the CPU restatement of their kernels, with the reference tests' tolerance (atol 1e-2, tests/test_rms_norm.py / test_cross_entropy.py)."""
    from flasht5_amd import fast_rms_layernorm, cross_entropy_loss
    msgs = []
    dtype = rng.choice([torch.bfloat16, torch.bfloat16, torch.float16, torch.float32])
    g = torch.Generator().manual_seed(7000 + i)
    if rng.random() < 0.5:
        rows, n = rng.randint(1, 3000), rng.choice([rng.randint(1, 300), rng.randint(301, 4100), 512, 768, 1024, 2048])
        lead = rng.random() < 0.5 and rows % 2 == 0
        x = torch.randn(rows, n, generator=g).to(dtype)
        w = (1.0 + 0.1 * torch.randn(n, generator=g)).to(dtype if rng.random() < 0.7 else torch.float32)
        dy = torch.randn(rows, n, generator=g).to(dtype)
        eps = rng.choice([1e-6, 1e-5])
        desc = f"rmsnorm rows={rows} n={n} {str(dtype)[6:]} w={str(w.dtype)[6:]} eps={eps:g} 3d={int(lead)}"
        y_ref, rstd = oracle.rmsnorm_fwd_oracle(x, w, eps)
        dx_ref, dw_ref = oracle.rmsnorm_bwd_oracle(dy, x, w, rstd)
        xs = x.cuda().view(2, rows // 2, n) if lead else x.cuda()
        xg, wg = xs.clone().requires_grad_(), w.cuda().clone().requires_grad_()
        try:
            y = fast_rms_layernorm(xg, wg, eps)
            dx, dw = torch.autograd.grad(y, [xg, wg], dy.cuda().view_as(y))
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            return desc, [f"raised {type(e).__name__}: {e}"]
        for got, ref, key, tol in ((y.reshape(rows, n), y_ref, "y", 1e-2), (dx.reshape(rows, n), dx_ref, "dx", 1e-2), (dw, dw_ref, "dw", 1e-2 * max(1.0, rows ** 0.5))):
            lim = tol * max(1.0, ref.float().abs().max().item())
            if not torch.isfinite(got.float()).all() or maxdiff(got.cpu(), ref) > lim:
                msgs.append(f"{key} {maxdiff(got.cpu(), ref):.3e} > {lim:.3e}")
        return desc, msgs
    rows, V = rng.randint(1, 600), rng.choice([rng.randint(2, 500), rng.randint(501, 40000), 32128, 32768])
    smooth, zl, scale = rng.choice([0.0, 0.0, 0.1]), rng.choice([0.0, 1e-4, 1.0]), rng.choice([1.0, 1.0, 0.5])
    logits = (torch.randn(rows, V, generator=g) * rng.choice([1.0, 4.0])).to(dtype)
    labels = torch.randint(0, V, (rows,), generator=g)
    for r in range(rows):  # ignored rows and labels outside the vocabulary (the reference's vocab-parallel convention: no picked logit)
        u = rng.random()
        if u < 0.1:
            labels[r] = -100
        elif u < 0.13:
            labels[r] = V + 5
    inplace = rng.random() < 0.3
    desc = f"ce rows={rows} V={V} {str(dtype)[6:]} smoothing={smooth} z={zl:g} logit_scale={scale} inplace={int(inplace)}"
    l_ref, z_ref, lse = oracle.ce_fwd_oracle(logits, labels, smooth, scale, zl, -100)
    dl = torch.randn(rows, generator=g)
    d_ref = oracle.ce_bwd_oracle(dl, logits, lse, labels, smooth, scale, zl, -100)
    lg = logits.cuda().clone().requires_grad_()
    try:
        losses, zz = cross_entropy_loss(lg, labels.cuda(), label_smoothing=smooth, logit_scale=scale, lse_square_scale=zl, inplace_backward=inplace)
        (dlg,) = torch.autograd.grad(losses, [lg], dl.cuda())
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        return desc, [f"raised {type(e).__name__}: {e}"]
    for got, ref, key in ((losses, l_ref, "loss"), (zz, z_ref, "z_loss"), (dlg, d_ref, "dlogits")):
        lim = 1e-2 * max(1.0, ref.float().abs().max().item())
        if not torch.isfinite(got.float()).all() or maxdiff(got.cpu(), ref) > lim:
            msgs.append(f"{key} {maxdiff(got.cpu(), ref):.3e} > {lim:.3e}")
    return desc, msgs
