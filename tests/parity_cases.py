"""Parity cases shared by the GPU tests (tests/test_parity_configs_gpu.py asserts err <= bound for every record) and by
tools/parity_report.py (writes the measured errors and the bounds used to profiles/parity_rNN.json).

A case is a function returning a list of records {"case", "tensor", "err", "bound"}: max-abs error of the HIP path (through
the C ABI) against the reference-derived truth -- committed golden fixtures (outputs of the reference's eager `attn_ref` +
autograd and of its Triton kernels under the interpreter), or the oracle restatement at sizes no fixture covers -- and the
bound the tests enforce:  (1e-3 + u * half-ulp(dtype)) * max(1, max|ref|), u = 1 forward / 3 gradients
(tests/test_attention_gpu.py header)."""
import math

import torch

import oracle
from attn_helpers import make_inputs, oracle_all, run_dense, maxdiff
from golden_io import load_attn, ATTN_CASES, TRITON_CASES

HALF_ULP = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


def bound(ref_t, dtype, atol=1e-3, ulps=1.0):
    return (atol + ulps * HALF_ULP[dtype]) * max(1.0, ref_t.float().abs().max().item())


def gbound(ref_t, dtype):
    return bound(ref_t, dtype, ulps=3.0)


def lse_bound(L_ref):
    fin = torch.isfinite(L_ref)
    return 1e-4 * max(1.0, L_ref[fin].abs().max().item() if fin.any() else 0.0)


def rec(case, tensor, err, bnd):
    return {"case": case, "tensor": tensor, "err": float(err), "bound": float(bnd)}


def _dev(c):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in c.items()}


# ---- committed fixtures ---------------------------------------------------------------------------------------------
def fixture_case(name):
    c = _dev(load_attn(name))
    dt = c["dtype"]
    got = run_dense(c["q"], c["k"], c["v"], c["bias"], c["do"], c["sm_scale"], c["causal"])
    out = [rec(name, "o", maxdiff(got["o"], c["o"]), bound(c["o"], dt)),
           rec(name, "lse", maxdiff(got["L"], c["L"]), lse_bound(c["L"]))]  # (part of the operator contract: reference :59, :476)
    lp = c["eager_lp_err"].tolist()
    for i, key in enumerate(("o", "dq", "dk", "dv")):  # the reference's own rule (tests/fa2_triton/test_fa2_bias.py:26-28), unmodified
        out.append(rec(name, key + " [reference rule: 2 x eager low-precision error + 1e-5]", maxdiff(got[key], c[key]), 2 * lp[i] + 1e-5))
    if "o_ref" in c:
        out.append(rec(name, "o_vs_reference_eager_fp32", maxdiff(got["o"], c["o_ref"]), bound(c["o_ref"], dt) + 2e-5))
    for key in ("dq", "dk", "dv"):
        out.append(rec(name, key, maxdiff(got[key], c[key]), gbound(c[key], dt)))
    if c["bias"] is not None:
        nsum = (c["B"] if c["bias"].shape[0] == 1 else 1) * (c["H"] if c["bias"].shape[1] == 1 else 1)
        out.append(rec(name, "dbias", maxdiff(got["db"], c["dbias"]), gbound(c["dbias"], dt) * (1 + nsum)))
    return out


def triton_case(name):
    """distance to the outputs of the reference's OWN Triton kernels (fp16, CPU interpreter) on the same inputs"""
    c = _dev(load_attn(name))
    got = run_dense(c["q"], c["k"], c["v"], c["bias"], c["do"], c["sm_scale"], c["causal"])
    out = []
    for key, tk in (("o", "o_triton"), ("dq", "dq_triton"), ("dk", "dk_triton"), ("dv", "dv_triton"), ("db", "dbias_triton")):
        scale = max(1.0, c[tk].float().abs().max().item())
        nsum = c["B"] if (key == "db" and c["bias"].shape[0] == 1) else 1
        out.append(rec(name, key + "_vs_triton", maxdiff(got[key], c[tk]), 2 * (1e-3 + HALF_ULP[torch.float16]) * scale * nsum))
    out.append(rec(name, "lse_vs_triton", maxdiff(got["L"], c["L_triton"]), lse_bound(c["L_triton"])))
    return out


def cfg1_case():
    from flasht5_amd import flash_attention_v2_bias
    c = _dev(load_attn("attn_cfg1_fp32"))
    o = flash_attention_v2_bias(c["q"].bfloat16(), c["k"].bfloat16(), c["v"].bfloat16(), c["bias"].bfloat16(), False, c["sm_scale"])
    return [rec("cfg1 (2,8,128,64) fwd", "o", maxdiff(o, c["o"]), bound(c["o"], torch.bfloat16)),
            rec("cfg1 (2,8,128,64) fwd", "o_vs_reference_eager_fp32", maxdiff(o, c["o_ref"]), bound(c["o_ref"], torch.bfloat16) + 2e-5)]


# ---- BASELINE.json configs at full size ---------------------------------------------------------------------------------
def _table(H, seed):
    return torch.randn(32, H, generator=torch.Generator().manual_seed(seed)) * 0.5


def _table_grad_truth(table, M, N, db_alg, bidir=True, md=128):
    tl = table.clone().requires_grad_()
    oracle.compute_bias(tl, M, N, bidir, 32, md).backward(db_alg.cpu())
    return tl.grad


def cfg2_case(mode="rpe"):
    """config 2 at FULL size: (4,12,512,64) bf16, (B,S,H,D)-strided, 32-bucket T5 bias: fwd + bwd + bias gradient -- the split
    forward and the side-by-side (fused) backward launch that the bench times."""
    from flasht5_amd import flash_attention_v2_rpe
    B, H, S, D, dt, scale = 4, 12, 512, 64, torch.bfloat16, 0.125
    name = f"cfg2 (4,12,512,64) {mode}"
    q, k, v, _, do = make_inputs(B, H, S, S, D, dt, None, seed=42, strided=True)
    table = _table(H, 1)
    bias = oracle.compute_bias(table, S, S, True, 32, 128).contiguous().cuda()          # fp32 (1,H,S,S)
    if mode == "dense":
        bias = bias.to(dt)  # the dense operator takes the bias in q's dtype (reference positional_encoding.py:108)
    ref = oracle_all(q, k, v, bias, do, scale, False)
    leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)]
    out = []
    if mode == "rpe":
        tb = table.cuda().requires_grad_()
        o = flash_attention_v2_rpe(leaves[0], leaves[1], leaves[2], tb, True, 32, 128, False, scale)
        dq, dk, dv, dt_ = torch.autograd.grad(o, leaves + [tb], do)
        _, _, _, _, db_alg = oracle.attn_bwd_oracle(q, k, v, bias, o.detach(), ref["L"], do, scale, False)
        want = _table_grad_truth(table, S, S, db_alg)
        out.append(rec(name, "dtable(32,12)", maxdiff(dt_.cpu(), want), 5e-3 * max(1.0, want.abs().max().item()) + 2e-2))
    else:
        got = run_dense(q, k, v, bias, do, scale, False)
        o, dq, dk, dv = got["o"], got["dq"], got["dk"], got["dv"]
        out.append(rec(name, "dbias(1,12,512,512)", maxdiff(got["db"], ref["db"]), gbound(ref["db"], dt) * (1 + B)))
    out += [rec(name, "o", maxdiff(o, ref["o"]), bound(ref["o"], dt))]
    out += [rec(name, key, maxdiff(g, ref[key]), gbound(ref[key], dt)) for key, g in (("dq", dq), ("dk", dk), ("dv", dv))]
    return out


def cfg3_case():
    """config 3: S = 8192.  The full (4,12,8192,64) problem runs through the default dispatch (pipelined forward, two-kernel
    backward); a (1, 2-head) slice is compared with the fp32 oracle run on the device, including the (H, 2R+1) diagonal sums
    of dS (the RPE bias gradient) from a stand-alone run of the slice."""
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import positional_encoding as pe
    B, H, S, D, dt, scale = 4, 12, 8192, 64, torch.bfloat16, 0.125
    name = "cfg3 (4,12,8192,64) rpe"
    q, k, v, _, do = make_inputs(B, H, S, S, D, dt, None, seed=8, strided=True)
    table = _table(H, 2).cuda()
    rpe1d = pe.rpe1d_from_table(table)
    plan = AttentionPlan(q, k, v, do, rpe1d=rpe1d, radius=128, sm_scale=scale)
    o = plan.forward().clone()
    dq, dk, dv, d1 = (t.clone() for t in plan.backward())
    torch.cuda.synchronize()
    sl = (slice(0, 1), slice(0, 2))
    bias = pe.compute_bias(table[:, :2], S, S).contiguous()
    qs, ks, vs, dos = (t[sl] for t in (q, k, v, do))
    ref_o, ref_L = oracle.attn_fwd_oracle(qs, ks, vs, bias, scale, False)
    rdq, rdk, rdv, _, _ = oracle.attn_bwd_oracle(qs, ks, vs, bias, ref_o, ref_L, dos, scale, False)
    out = [rec(name, "o[0,:2]", maxdiff(o[sl], ref_o), bound(ref_o, dt)),
           rec(name, "lse[0,:2]", maxdiff(plan.lse[sl], ref_L), 1e-3),
           rec(name, "dq[0,:2]", maxdiff(dq[sl], rdq), gbound(rdq, dt)),
           rec(name, "dk[0,:2]", maxdiff(dk[sl], rdk), gbound(rdk, dt)),
           rec(name, "dv[0,:2]", maxdiff(dv[sl], rdv), gbound(rdv, dt))]
    # a second slice at the other end of the grid: last batch element, last two heads (another XCD, the last workgroups)
    sl2 = (slice(B - 1, B), slice(H - 2, H))
    bias2 = pe.compute_bias(table[:, H - 2:], S, S).contiguous()
    q2, k2, v2, do2 = (t[sl2] for t in (q, k, v, do))
    ref_o2, ref_L2 = oracle.attn_fwd_oracle(q2, k2, v2, bias2, scale, False)
    rdq2, rdk2, rdv2, _, _ = oracle.attn_bwd_oracle(q2, k2, v2, bias2, ref_o2, ref_L2, do2, scale, False)
    out += [rec(name, "o[3,10:]", maxdiff(o[sl2], ref_o2), bound(ref_o2, dt)),
            rec(name, "lse[3,10:]", maxdiff(plan.lse[sl2], ref_L2), 1e-3),
            rec(name, "dq[3,10:]", maxdiff(dq[sl2], rdq2), gbound(rdq2, dt)),
            rec(name, "dk[3,10:]", maxdiff(dk[sl2], rdk2), gbound(rdk2, dt)),
            rec(name, "dv[3,10:]", maxdiff(dv[sl2], rdv2), gbound(rdv2, dt))]
    del ref_o2, ref_L2, rdq2, rdk2, rdv2, bias2
    # bias gradient of the slice: diagonal sums of the oracle's dS (delta from the kernel's stored o, as FA2 defines it)
    p2 = AttentionPlan(qs, ks, vs, dos, rpe1d=rpe1d[:2].contiguous(), radius=128, sm_scale=scale)
    o2 = p2.forward().clone()
    _, _, _, d1s = p2.backward()
    _, _, _, ds, _ = oracle.attn_bwd_oracle(qs, ks, vs, bias, o2, ref_L, dos, scale, False)   # ds (1,2,S,S) fp32
    R = 128
    delta = torch.clamp(torch.arange(S, device="cuda")[None, :] - torch.arange(S, device="cuda")[:, None], -R, R) + R
    want = torch.zeros(2, 2 * R + 1, device="cuda", dtype=torch.float64)
    for h in range(2):
        want[h].index_add_(0, delta.reshape(-1), ds[0, h].double().reshape(-1))
    want = want.float()
    # the two clamped end entries sum ~S^2/2 rounded dS values each: bound relative to the largest sum
    out.append(rec(name, "drpe1d(2,257) of the slice", maxdiff(d1s, want), 5e-3 * max(1.0, want.abs().max().item()) + 2e-2))
    # the far bins sum the dS that were rounded to bf16 for the dK GEMM (the reference's bias gradient is made of the same rounded
    # values, :720): n independent roundings, 4 sigma = 4 * 2^-9 / sqrt(3) * sqrt(sum ds^2); sum ds^2 of a head from the slice, x B;
    # x 3 for the heads the slice does not cover
    allow = 3 * max(4.0 * 2.0 ** -9 / 3 ** 0.5 * (q.shape[0] * (ds[0, hh].float() ** 2).sum().item()) ** 0.5 for hh in range(2))
    out.append(rec(name, "sum_k drpe1d[h,k] (softmax Jacobian rows sum to 0)", d1.sum(-1).abs().max().item(),
                   allow + 1e-3 * d1.abs().sum(-1).max().item() + 1e-2))
    return out


def cfg4_case():
    """config 4: packed decoder cross-attention, q_len <= 256 / kv_len <= 4096 via cu_seqlens, forward AND backward at kv 4096"""
    from flasht5_amd import flash_attn_varlen_func
    H, D, dt, scale = 12, 64, torch.bfloat16, 0.125
    name = "cfg4 varlen q256/kv4096"
    cu_q, cu_k = [0, 256, 512, 704, 768], [0, 4096, 7168, 11264, 12288]
    g = torch.Generator().manual_seed(4)
    q, k, v, do = (torch.randn(n, H, D, generator=g).to(dt).cuda() for n in (cu_q[-1], cu_k[-1], cu_k[-1], cu_q[-1]))
    leaves = [t.clone().requires_grad_() for t in (q, k, v)]
    o = flash_attn_varlen_func(leaves[0], leaves[1], leaves[2], torch.tensor(cu_q, dtype=torch.int32).cuda(),
                               torch.tensor(cu_k, dtype=torch.int32).cuda(), 256, 4096, False, scale)
    dq, dk, dv = torch.autograd.grad(o, leaves, do)
    ref_o = oracle.attn_varlen_oracle(q.cpu(), k.cpu(), v.cpu(), cu_q, cu_k, scale, False)
    rdq, rdk, rdv = oracle.attn_varlen_bwd_oracle(q.cpu(), k.cpu(), v.cpu(), do.cpu(), cu_q, cu_k, scale, False)
    return [rec(name, "o", maxdiff(o.cpu(), ref_o), bound(ref_o, dt)), rec(name, "dq", maxdiff(dq.cpu(), rdq), gbound(rdq, dt)),
            rec(name, "dk", maxdiff(dk.cpu(), rdk), gbound(rdk, dt)), rec(name, "dv", maxdiff(dv.cpu(), rdv), gbound(rdv, dt))]


def rowwise_cases():
    """RMSNorm / cross-entropy + z-loss against the fixtures the reference's own Triton kernels produced under the
    interpreter (fp32; tests/golden/make_golden.py gen_rmsnorm / gen_ce)"""
    from golden_io import load
    from flasht5_amd import fast_rms_layernorm, cross_entropy_loss
    out = []
    z = load("rmsnorm")
    for tag in ("a", "b", "c"):
        x, w, dy = (torch.from_numpy(z[f"{n}_{tag}"]).cuda() for n in ("x", "w", "dy"))
        xx, ww = x.clone().requires_grad_(), w.clone().requires_grad_()
        y = fast_rms_layernorm(xx, ww, 1e-6)
        y.backward(dy)
        out += [rec(f"rmsnorm fixture {tag} {tuple(x.shape)}", "y", maxdiff(y.cpu(), torch.from_numpy(z[f"y_{tag}"])), 1e-5),
                rec(f"rmsnorm fixture {tag} {tuple(x.shape)}", "dx", maxdiff(xx.grad.cpu(), torch.from_numpy(z[f"dx_{tag}"])), 1e-5),
                rec(f"rmsnorm fixture {tag} {tuple(x.shape)}", "dw", maxdiff(ww.grad.cpu(), torch.from_numpy(z[f"dw_{tag}"])), 1e-4)]
    z = load("cross_entropy")
    for tag in ("a", "b", "c", "d"):
        smoothing, zscale = (float(t) for t in z[f"cfg_{tag}"])
        logits = torch.from_numpy(z[f"logits_{tag}"]).cuda().requires_grad_()
        labels = torch.from_numpy(z[f"labels_{tag}"]).cuda()
        losses, zl = cross_entropy_loss(logits, labels, label_smoothing=smoothing, lse_square_scale=zscale)
        losses.backward(torch.from_numpy(z[f"dloss_{tag}"]).cuda())
        nm = f"cross-entropy fixture {tag} {tuple(logits.shape)} smoothing {smoothing} z {zscale}"
        sc = max(1.0, 50 * zscale)
        out += [rec(nm, "loss", maxdiff(losses.detach().cpu(), torch.from_numpy(z[f"loss_{tag}"])), 2e-4 * sc),
                rec(nm, "z_loss", maxdiff(zl.cpu(), torch.from_numpy(z[f"z_{tag}"])), 2e-4 * sc),
                rec(nm, "dlogits", maxdiff(logits.grad.cpu(), torch.from_numpy(z[f"dlogits_{tag}"])), 1e-5 * sc)]
    return out


# ---- the bodies of rounds 5 - 6, forced per call (per-tensor records, not only pass / fail) -----------------------------------
def body_case(name, B, H, M, N, D, mode, bits, causal=False, scale=0.125, dt=torch.bfloat16, want=None, minus_inf=False):
    """one problem through AttentionPlan with the variant bits `bits` against the fp32 oracle: o, lse, dq, dk, dv and the bias gradient (dense dbias / T5 table);
    `want`: entries fat5_attn_describe must report (the body under test really ran)"""
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import positional_encoding as pe
    q, k, v, _, do = make_inputs(B, H, M, N, D, dt, None, seed=B + M + 3 * N + D, strided=True)
    table = _table(H, 3)
    kw = {}
    if mode == "dense":
        g = torch.Generator().manual_seed(7)
        bias = (torch.randn(1, H, M, N, generator=g) * 0.7).to(dt).cuda()
        if minus_inf:  # an additive mask written as -inf: a band of keys for every row + scattered entries (every row keeps visible keys)
            bias[..., N - 72:N - 8] = float("-inf")
            bias = bias.masked_fill((torch.rand(1, H, M, N, generator=g) < 0.03).cuda() & (torch.arange(N).cuda() >= 32), float("-inf"))
        kw = dict(bias=bias)
        ref_bias = bias
    elif mode == "rpe":
        ref_bias = oracle.compute_bias(table, M, N, True, 32, 128).contiguous().cuda()
        kw = dict(rpe1d=pe.rpe1d_from_table(table.cuda(), True, 32, 128), radius=128, rpe_bucket=pe.bucket_index32(128, True, 32, 128, "cuda"), num_buckets=32)
    else:
        ref_bias = None
    plan = AttentionPlan(q, k, v, do, causal=causal, sm_scale=scale, variant=bits, **kw)
    d = plan.describe()
    for key, val in (want or {}).items():
        assert d[key] == val, (name, key, d)
    o = plan.forward().clone()
    dq, dk, dv, db = plan.backward()
    torch.cuda.synchronize()
    ref = oracle_all(q, k, v, ref_bias, do, scale, causal)
    out = [rec(name, "o", maxdiff(o, ref["o"]), bound(ref["o"], dt, ulps=2.0 if minus_inf else 1.0)), rec(name, "lse", maxdiff(plan.lse, ref["L"]), 10 * lse_bound(ref["L"]))]
    out += [rec(name, key, maxdiff(g_, ref[key]), gbound(ref[key], dt)) for key, g_ in (("dq", dq), ("dk", dk), ("dv", dv))]
    if mode == "dense":
        out.append(rec(name, f"dbias(1,{H},{M},{N})", maxdiff(db, ref["db"]), gbound(ref["db"], dt) * (1 + B)))
    elif mode == "rpe":
        _, _, _, _, db_alg = oracle.attn_bwd_oracle(q, k, v, ref_bias, o, ref["L"], do, scale, causal)
        want_t = _table_grad_truth(table, M, N, db_alg)
        out.append(rec(name, f"dtable(32,{H})", maxdiff(db.cpu(), want_t), 5e-3 * max(1.0, want_t.abs().max().item()) + 2e-2))
    return out


def round56_cases():
    from flasht5_amd import _lib as L
    two = L.V_QDB64_ON | L.V_KV64_ON | L.V_FUSED64_OFF
    one = L.V_QDB64_ON | L.V_KV64_ON | L.V_FUSED64_ON
    out = []
    out += body_case("r5 dense: dQ + dBias kernel | dense 64-key dK/dV, two launches (4,12,1024,64)", 4, 12, 1024, 1024, 64, "dense", two, want={"dq": "64row-batch4", "dkdv": "64key", "fused": "0"})
    out += body_case("r5 dense: both in ONE launch (4,12,512,64)", 4, 12, 512, 512, 64, "dense", one, want={"dq": "64row-batch4", "fused": "1"})
    out += body_case("r5 dense: B = 6, fp32 slabs + partial reduction (6,4,512,64)", 6, 4, 512, 512, 64, "dense", two, want={"dbias": "dq-kernel+partials"})
    out += body_case("r5 dense: reference benchmark's form, B = 16 causal sm_scale 1.3 (16,4,512,64)", 16, 4, 512, 512, 64, "dense", one, causal=True, scale=1.3, want={"dq": "64row-batch4"})
    out += body_case("r5 dense: fp16 (4,4,512,64) 64-key dK/dV", 4, 4, 512, 512, 64, "dense", L.V_KV64_ON, dt=torch.float16, want={"dkdv": "64key"})
    out += body_case("r6 dense with -inf mask entries, two launches (4,4,512,64)", 4, 4, 512, 512, 64, "dense", two, minus_inf=True, want={"dq": "64row-batch4", "dkdv": "64key"})
    out += body_case("r6 dense with -inf mask entries, one launch, sm_scale 1.3 (4,4,512,64)", 4, 4, 512, 512, 64, "dense", one, scale=1.3, minus_inf=True, want={"fused": "1"})
    out += body_case("r5 head_dim 128 pipelined forward, T5 table (2,4,1024,128)", 2, 4, 1024, 1024, 128, "rpe", L.V_FWD64_ON, scale=128 ** -0.5, want={"fwd": "64row"})
    out += body_case("r5 head_dim 128 pipelined forward, dense bias (2,4,1024,128)", 2, 4, 1024, 1024, 128, "dense", L.V_FWD64_ON, scale=128 ** -0.5, want={"fwd": "64row"})
    out += body_case("r5 head_dim 128 pipelined forward, causal none, sm_scale 1.3 (2,4,1024,128)", 2, 4, 1024, 1024, 128, "none", L.V_FWD64_ON, causal=True, scale=1.3, want={"fwd": "64row"})
    out += body_case("r6 table gradient from the dQ workgroups (cfg2's form) (4,12,512,64)", 4, 12, 512, 512, 64, "rpe", L.V_FUSED64_ON | L.V_QDIAG_ON, want={"fused": "1", "qdiag": "1"})
    out += body_case("r6 ... causal, ragged (2,4,1000,1100,64)", 2, 4, 1000, 1100, 64, "rpe", L.V_FUSED64_ON | L.V_QDIAG_ON, causal=True, want={"qdiag": "1"})
    out += body_case("r4 table gradient from the dK/dV workgroups (4,12,512,64)", 4, 12, 512, 512, 64, "rpe", L.V_FUSED64_ON | L.V_QDIAG_OFF, want={"fused": "1", "qdiag": "0"})
    return out


ALL_CASES = ([("fixture:" + n, (lambda n=n: fixture_case(n))) for n in ATTN_CASES] +
             [("triton:" + n, (lambda n=n: triton_case(n))) for n in TRITON_CASES] +
             [("cfg1", cfg1_case), ("cfg2:rpe", lambda: cfg2_case("rpe")), ("cfg2:dense", lambda: cfg2_case("dense")),
              ("cfg3", cfg3_case), ("cfg4", cfg4_case), ("rowwise", rowwise_cases), ("rounds5-6", round56_cases)])
