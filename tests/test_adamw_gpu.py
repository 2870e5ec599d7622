"""GPU parity of the fused AdamWScale step (csrc/adamw_kernels.h through fat5_adamw_scale_step) against the
fixtures produced by the REFERENCE optimizer class (tests/golden/make_golden.py gen_adamw: three steps on four tensors per
case -- fp32, fp32 + weight decay, bf16, bf16 + Kahan + weight decay, fp16 + Kahan; one tensor sits below the 1e-3 rms floor)
and against the oracle at FAT5-sized tensors."""
import numpy as np
import pytest
import torch

import oracle
from golden_io import load, _t

pytestmark = pytest.mark.gpu
DT = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}
ULP = {torch.float32: 2.0 ** -23, torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}
TINY = {torch.float32: 0.0, torch.float16: 2.0 ** -24, torch.bfloat16: 0.0}  # spacing of fp16 subnormals (v ~ 1e-5 lives there)


def _case(z, name):
    cfg = z[f"{name}__cfg"]
    dtype, kahan, wd, lr, b1, b2, eps = DT[int(cfg[0])], bool(cfg[1]), float(cfg[2]), float(cfg[3]), float(cfg[4]), float(cfg[5]), float(cfg[6])
    get = lambda key, i: _t(z[f"{name}__{key}_{i}"])  # noqa: E731
    return dtype, kahan, wd, lr, b1, b2, eps, get


@pytest.mark.parametrize("name", ["fp32", "fp32_wd", "bf16", "bf16_kahan_wd", "fp16_kahan", "fp32_sbf16_wd", "bf16_kahan_sfp16", "bf16_plain", "fp32_plain_wd", "fp16_kahan_plain_sbf16"])
def test_adamw_scale_reference_fixture(name):
    """Against the reference class's CPU results.  torch's CPU kernels round the `alpha` of a 16-bit add_ to 16 bit where its
    device kernels (and this one) keep it in fp32, and a Kahan pair (p, k) may split the same value one ulp of p differently:
    one unit in the last place of the tensor dtype is the bar for p, m, v; p + k is compared as a sum."""
    from flasht5_amd import AdamWScale
    z = load("adamw_scale")
    dtype, kahan, wd, lr, b1, b2, eps, get = _case(z, name)
    cfg = z[f"{name}__cfg"]
    sdtype = None if int(cfg[7]) < 0 else DT[int(cfg[7])]  # `use_state_dtype` (reference :101-103)
    correct = bool(cfg[8])                                  # `correct_bias` (:177)
    params = [torch.nn.Parameter(get("p0", i).cuda()) for i in range(4)]
    opt = AdamWScale(params, lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, kahan_sum=kahan, correct_bias=correct, use_state_dtype=sdtype)
    for step in range(3):
        for i, p in enumerate(params):
            p.grad = get(f"g{step}", i).cuda()
        opt.step()
    torch.cuda.synchronize()

    def ulp_err(got, want):  # max error in units of the last place of the tensor's largest magnitude
        w = want.float()
        return (got.float().cpu() - w).abs().max().item() / (ULP[got.dtype] * max(w.abs().max().item(), 1e-30))  # (the tensor's own dtype: states may differ from p)

    for i, p in enumerate(params):
        st = opt.state[p]
        assert all(torch.isfinite(t.float()).all() for t in (p, st["exp_avg"], st["exp_avg_sq"]))
        assert ulp_err(st["exp_avg"], get("m", i)) <= 1.0, (name, i, "m", ulp_err(st["exp_avg"], get("m", i)))
        assert ulp_err(st["exp_avg_sq"], get("v", i)) <= 1.0 or (st["exp_avg_sq"].float().cpu() - get("v", i).float()).abs().max().item() <= 2 * TINY[st["exp_avg_sq"].dtype], (name, i, "v")
        if st["exp_avg"].dtype == p.dtype:
            assert ulp_err(p.detach(), get("p", i)) <= 2.0, (name, i, "p", ulp_err(p.detach(), get("p", i)))
        else:
            # 16-bit moments beside wider parameters: the one-ulp (of the STATE dtype) freedom of m and v above moves each update
            # by that relative amount -- far more than an ulp of an fp32 parameter
            moved = (get("p", i).float() - get("p0", i).float()).abs().max().item()
            perr = (p.detach().float().cpu() - get("p", i).float()).abs().max().item()
            assert perr <= 3 * ULP[st["exp_avg"].dtype] * moved + 2 * ULP[dtype] * get("p", i).float().abs().max().item(), (name, i, "p", perr, moved)
        if kahan:
            got = p.detach().float().cpu() + st["kahan_comp"].float().cpu()
            want = get("p", i).float() + get("k", i).float()
            # the pair carries p0 + (sum of updates) to ~ulp(k); the updates themselves agree to one ulp of m (CPU alpha rounding)
            moved = (get("p", i).float() - get("p0", i).float()).abs().max().item()
            tol = ULP[dtype] * (2 * moved + 4 * get("k", i).float().abs().max().item()) + 1e-12
            assert (got - want).abs().max().item() <= tol, (name, i, "p+k", (got - want).abs().max().item(), tol)


@pytest.mark.parametrize("dtype,kahan", [(torch.bfloat16, True), (torch.bfloat16, False), (torch.float32, False), (torch.float16, True)])
def test_adamw_scale_large_tensors_vs_oracle_on_device(dtype, kahan):
    """FAT5-base sized tensors (lm_head 32768 x 768, a GLU weight, a norm weight, the (32,12) table, one chunk + 1 element) and an
    unaligned view.  The oracle's op sequence runs ON THE DEVICE here: the same torch device kernels the reference optimizer
    uses in training (fp32 alpha, tree-reduced norm -- torch's CPU norm is 0.12 % off on 25 M elements); the fused step must
    reproduce it to the last place (fma contraction and the reduction order of rms(p) are the only freedoms left)."""
    from flasht5_amd import AdamWScale
    g = torch.Generator().manual_seed(3)
    shapes = [(32768, 768), (2048, 768), (768,), (32, 12), (8193,)]
    base = [(torch.randn(*s, generator=g) * 0.05).to(dtype) for s in shapes]
    odd = (torch.randn(1001, generator=g) * 0.05).to(dtype)
    params = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    flat = torch.zeros(1002, dtype=dtype).cuda()
    flat[1:] = odd.cuda()
    pv = torch.nn.Parameter(flat[1:])  # 2-byte offset: the kernels' scalar path
    params.append(pv)
    opt = AdamWScale(params, lr=3e-3, weight_decay=0.01, kahan_sum=kahan)
    ref_p = [t.clone().cuda() for t in base] + [odd.clone().cuda()]
    m = [torch.zeros_like(t) for t in ref_p]
    v = [torch.zeros_like(t) for t in ref_p]
    use_k = kahan and dtype != torch.float32
    k = [torch.zeros_like(t) if use_k else None for t in ref_p]
    for step in range(3):
        grads = [(torch.randn(t.shape, generator=g) * 0.02).to(dtype).cuda() for t in ref_p]
        for p, gr in zip(params, grads):
            p.grad = gr.clone()
        opt.step()
        for i in range(len(ref_p)):
            oracle.adamw_scale_step(ref_p[i], grads[i].clone(), m[i], v[i], k[i], step + 1, 3e-3, 0.9, 0.999, 0.01, 1e-6, True)
    torch.cuda.synchronize()
    for i, p in enumerate(params):
        st = opt.state[p]
        n = p.numel()
        for got, want, key in ((st["exp_avg"], m[i], "m"), (st["exp_avg_sq"], v[i], "v"), (p.detach(), ref_p[i], "p")):
            err = (got.float() - want.float()).abs().max().item()
            lim = 2.0 * max(ULP[dtype] * want.float().abs().max().item(), TINY[dtype])
            # one unit in the last place: which multiply-adds the device kernels contract into fmas is the compiler's choice
            assert err <= lim, (i, key, n, err, lim)
        if use_k:
            got = p.detach().float() + st["kahan_comp"].float()
            want = ref_p[i].float() + k[i].float()
            moved = (ref_p[i].float() - (base[i].cuda().float() if i < len(base) else odd.cuda().float())).abs().max().item()
            tol = ULP[dtype] * (2 * moved + 4 * k[i].float().abs().max().item()) + 1e-12
            assert (got - want).abs().max().item() <= tol, (i, "p+k", (got - want).abs().max().item(), tol)


@pytest.mark.parametrize("gscale", [30.0, 1e-3])
def test_fused_gradient_clipping_matches_clip_then_step(gscale):
    """AdamWScale(max_grad_norm=1.0): the global-norm clip folded into the step against torch's clip_grad_norm_ followed by the
    plain step.  gscale 30: coef << 1 (every gradient scaled, rounded to bf16 like the in-place multiply); 1e-3: coef clamps to
    1 and the result is bit-identical to the unclipped step."""
    from flasht5_amd import AdamWScale
    g = torch.Generator().manual_seed(7)
    shapes = [(257, 129), (4096,), (32, 12), (1000, 64)]
    base = [(torch.randn(*sh, generator=g) * 0.05).bfloat16().cuda() for sh in shapes]
    grads = [(torch.randn(*sh, generator=g) * gscale * 0.01).bfloat16().cuda() for sh in shapes]

    def run(fused):
        ps = [torch.nn.Parameter(b.clone()) for b in base]
        opt = AdamWScale(ps, lr=1e-2, weight_decay=0.01, kahan_sum=True, **({"max_grad_norm": 1.0} if fused else {}))
        norms = []
        for it in range(3):
            for p_, g_ in zip(ps, grads):
                p_.grad = (g_ * (1.0 + 0.1 * it)).clone()
            if not fused:
                norms.append(torch.nn.utils.clip_grad_norm_(ps, 1.0).float())
            opt.step()
            if fused:
                norms.append(opt.last_grad_norm.clone())
        return [p_.detach().float() + opt.state[p_]["kahan_comp"].float() for p_ in ps], norms, ps

    (pf, nf, psf), (pu, nu, _) = run(True), run(False)
    for a, b in zip(nf, nu):
        assert abs(a.item() - b.item()) <= 2.0 ** -7 * b.item() + 1e-6  # (torch forms the norms in the gradient dtype: bf16 here)
    for a, b, b0 in zip(pf, pu, base):
        moved = (b - b0.float()).abs().max().item()
        # (the two clip coefficients differ by up to a bf16 ulp of the norm: a few gradients round the other way)
        assert (a - b).abs().max().item() <= 2.0 ** -5 * moved + 1e-6, ((a - b).abs().max().item(), moved)
    for p_, g_ in zip(psf, grads):
        assert torch.equal(p_.grad, (g_ * 1.2).clone())  # the gradients themselves are not modified
    if gscale < 1.0:  # coef == 1: same bits as the step without clipping
        ps = [torch.nn.Parameter(b.clone()) for b in base]
        opt = AdamWScale(ps, lr=1e-2, weight_decay=0.01, kahan_sum=True)
        for it in range(3):
            for p_, g_ in zip(ps, grads):
                p_.grad = (g_ * (1.0 + 0.1 * it)).clone()
            opt.step()
        for p_, q_ in zip(ps, psf):
            assert torch.equal(p_.detach(), q_.detach())


@pytest.mark.parametrize("path", ["native", "python"])
def test_rpe_table_forward_sees_fused_optimizer_updates(path):
    """`flash_attention_v2_rpe(q, k, v, table)` in a training loop with this package's fused AdamWScale.  The
    optimizer kernel writes the table through its raw pointer (no autograd version bump), so anything cached per table version
    would keep running the step-0 bias.  Three steps of the in-kernel-RPE path against the dense path (`compute_bias` of the live
    table, the reference's formulation: positional_encoding.py:73-110 + flash_attention_v2_bias) with its own copy of the table
    and optimizer: outputs agree at EVERY step and the tables stay together."""
    from flasht5_amd import AdamWScale, flash_attention_v2_bias, flash_attention_v2_rpe
    from flasht5_amd import positional_encoding as pe
    from flasht5_amd.flash_attention_v2_bias import FlashAttentionRPE
    B, H, S, D = 2, 4, 192, 64
    g = torch.Generator().manual_seed(21)
    q, k, v, do = (torch.randn(B, H, S, D, generator=g).bfloat16().cuda() for _ in range(4))
    t0 = torch.randn(32, H, generator=g) * 0.5
    ta, tb = torch.nn.Parameter(t0.clone().cuda()), torch.nn.Parameter(t0.clone().cuda())
    oa_, ob_ = AdamWScale([ta], lr=0.5), AdamWScale([tb], lr=0.5)  # (large steps: a stale bias would be off by ~0.25 per entry)
    first = None
    for step in range(3):
        if path == "native":
            o1 = flash_attention_v2_rpe(q, k, v, ta, True, 32, 128, False, 0.125)
        else:
            o1 = FlashAttentionRPE.apply(q, k, v, ta, True, 32, 128, False, 0.125)
        (o1.float() * do.float()).sum().backward()
        bias = pe.compute_bias(tb, S, S, True, 32, 128).to(torch.bfloat16).contiguous()
        o2 = flash_attention_v2_bias(q, k, v, bias, False, 0.125)
        (o2.float() * do.float()).sum().backward()
        e = (o1.float() - o2.float()).abs().max().item()
        assert e <= 2 * (1e-3 + 2.0 ** -8) * max(1.0, o2.float().abs().max().item()), (step, e)  # (two bf16 outputs, bf16-rounded bias on one side)
        if first is None:
            first = o1.detach().clone()
        oa_.step()
        ob_.step()
        ta.grad = tb.grad = None
    # the table really moved, the forward really followed it, and both loops ended in the same place
    assert (ta.detach() - t0.cuda()).abs().max().item() > 0.2
    assert (o1.detach().float() - first.float()).abs().max().item() > 0.02
    assert (ta.detach() - tb.detach()).abs().max().item() <= 0.05 * max(1.0, tb.detach().abs().max().item())


def test_captured_step_keeps_its_arena_and_refuses_a_second_capture():
    """a captured step bakes raw pointers to the optimizer's arena (descriptor table, step scalars) into its launches.
    init_state() must not replace that arena, a second capture must not hand the same bytes out again (and graph_advance() feeds
    one capture only): refused until release_captured_step()."""
    from flasht5_amd.adamw_scaled import AdamWScale
    p = torch.nn.Parameter(torch.randn(4096, device="cuda", dtype=torch.bfloat16))
    opt = AdamWScale([p], lr=1e-2)
    p.grad = torch.randn_like(p)
    opt.step()
    opt.init_state()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        opt.step()
    arena = opt._graph_arena[p.device][0]
    opt.init_state()
    assert opt._graph_arena[p.device][0] is arena
    before = p.detach().clone()
    opt.graph_advance()
    g.replay()
    torch.cuda.synchronize()
    assert not torch.equal(before, p.detach())
    with pytest.raises(RuntimeError, match="already holds a captured step"):
        _capture_again(opt)
    t1 = opt.capture_token()
    assert t1 > 0
    opt.release_captured_step()
    assert opt.capture_token() == 0
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        opt.step()
    # whoever held the FIRST capture (an old GraphedTrainStep collected late) must not release the second one's arena
    t2 = opt.capture_token()
    assert t2 > 0 and t2 != t1
    assert opt.release_captured_step(t1) is False and opt.capture_token() == t2 and len(opt._graph_jobs) > 0
    opt.graph_advance()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(p.detach().float()).all()
    assert opt.release_captured_step(t2) is True and opt.capture_token() == 0


def _capture_again(opt):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        opt.step()
