"""fat5_gated_act_fwd / _bwd (the T5 v1.1 gated feed-forward activation, reference modeling_flash_t5.py:139-142) through the C ABI
against the oracle (oracle/gated_act.py, pinned on the reference module by tests/golden/gated_act.npz) and against that fixture.

Tolerance: fp32 arithmetic inside, ONE rounding per output element -> (0.5 ulp of the output + the fast exp2 / rcp of the tanh,
~1e-6 relative) * |ref|, + a small absolute floor: 2^-8 (bf16) / 2^-11 (fp16) / 4e-6 (fp32) relative, written below."""
import pytest
import torch

import oracle
from golden_io import load

pytestmark = pytest.mark.gpu

REL = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11, torch.float32: 4e-6}


def _close(got, ref, dtype, what):
    ref = ref.double().cpu()
    err = (got.double().cpu() - ref).abs()
    tol = REL[dtype] * ref.abs() + (1e-5 if dtype == torch.float32 else 2e-4)
    assert bool((err <= tol).all()), f"{what}: err/tol {float((err / tol).max()):.2f}"


@pytest.mark.parametrize("act", ["gelu_tanh", "relu"])
@pytest.mark.parametrize("tag,dtype", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
def test_gated_act_against_reference_fixture(act, tag, dtype):
    """inputs and outputs of the reference's FlashT5DenseGatedAct itself (forward hooks + autograd, make_golden.py::gen_gated_act);
    the reference rounds act(h0) before the multiply and autograd rounds every product: a few ulps of the tensor dtype"""
    from flasht5_amd.gated_act import gated_act
    z = load("gated_act")
    g = {k: torch.from_numpy(z[f"{act}_{tag}_{k}"]).to(dtype).cuda() for k in ("h0", "h1", "out", "dout", "dh0", "dh1")}
    h0, h1 = g["h0"].clone().requires_grad_(), g["h1"].clone().requires_grad_()
    out = gated_act(h0, h1, act)
    out.backward(g["dout"])
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -23
    for got, ref, n in ((out, g["out"], 2), (h0.grad, g["dh0"], 4), (h1.grad, g["dh1"], 2)):
        err = (got.double() - ref.double()).abs()
        tol = n * ulp * ref.double().abs().clamp_min(1e-3) + (3e-6 if tag == "fp32" else 1e-3)
        assert bool((err <= tol).all()), float((err / tol).max())


@pytest.mark.parametrize("act", ["gelu_tanh", "relu"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("rows,F", [(4096, 2048), (37, 136), (1, 8)])
def test_gated_act_packed_and_two_tensor_forms(act, dtype, rows, F):
    from flasht5_amd.gated_act import gated_act, gated_act_packed
    g = torch.Generator().manual_seed(rows + F)
    h = (torch.randn(rows, 2 * F, generator=g) * 3).to(dtype).cuda()
    dout = torch.randn(rows, F, generator=g).to(dtype).cuda()
    ref = oracle.gated_act_oracle(h[:, :F].cpu(), h[:, F:].cpu(), act)
    d0, d1 = oracle.gated_act_bwd_oracle(dout.cpu(), h[:, :F].cpu(), h[:, F:].cpu(), act)
    hp = h.clone().requires_grad_()
    out = gated_act_packed(hp, act)
    out.backward(dout)
    _close(out, ref, dtype, "packed out")
    _close(hp.grad[:, :F], d0, dtype, "packed dh0")
    _close(hp.grad[:, F:], d1, dtype, "packed dh1")
    h0, h1 = h[:, :F].contiguous().requires_grad_(), h[:, F:].contiguous().requires_grad_()
    out2 = gated_act(h0, h1, act)
    out2.backward(dout)
    assert torch.equal(out2, out) and torch.equal(h0.grad, hp.grad[:, :F]) and torch.equal(h1.grad, hp.grad[:, F:])


def test_gated_act_shapes_strides_and_errors():
    from flasht5_amd.gated_act import gated_act, gated_act_packed
    # leading dims, a non-contiguous operand (copied to rows), the tails of the GELU (no NaN / inf for |x| large)
    h0 = torch.tensor([[-60.0, -12.0, -1.0, -0.0, 0.0, 1.0, 12.0, 60.0]] * 6, device="cuda").reshape(2, 3, 8).bfloat16()
    h1 = torch.full((2, 3, 16), 2.0, device="cuda").bfloat16()[..., ::2]
    out = gated_act(h0, h1, "gelu_tanh")
    assert out.shape == (2, 3, 8) and bool(torch.isfinite(out).all())
    ref = torch.nn.functional.gelu(h0.float(), approximate="tanh") * 2.0
    assert float((out.float() - ref).abs().max()) <= 2.0 ** -8 * 120
    assert abs(float(out[0, 0, 0])) < 1e-30 and float(out[0, 0, 7]) == 120.0  # (the kernel clamps the tanh argument at -40: e^-80, not 0)
    with pytest.raises(ValueError):
        gated_act(h0, h1, "swish")
    with pytest.raises(ValueError):
        gated_act(h0, h1[..., :4], "relu")
    with pytest.raises(ValueError):
        gated_act_packed(torch.zeros(4, 12, device="cuda").bfloat16(), "relu")  # halves of 6: not a multiple of the 8-element vector
    e = gated_act_packed(torch.zeros(0, 16, device="cuda").bfloat16(), "relu")
    assert e.shape == (0, 8)


def test_feed_forward_layer_uses_the_gated_kernel_and_matches_plain_torch():
    """FAT5LayerFF (reference FlashT5LayerFF, :148-164) with the gated kernel against the same layer written with torch ops, fp32
    oracle in between: both within the bf16 bound of the oracle, forward and input gradient"""
    from flasht5_amd import FAT5Config
    from flasht5_amd.fat5_step import FAT5LayerFF
    torch.manual_seed(0)
    outs = {}
    for fuse in (False, True):
        cfg = FAT5Config()
        cfg.fuse_gated_act = fuse
        torch.manual_seed(1)
        ff = FAT5LayerFF(cfg).cuda().bfloat16()
        x = torch.randn(2, 64, cfg.d_model, generator=torch.Generator().manual_seed(2)).cuda().bfloat16().requires_grad_()
        y = ff(x)
        y.backward(torch.ones_like(y))
        outs[fuse] = (y.detach().float(), x.grad.float())
        if fuse:
            a = ff.act
            xf = x.detach().double().cpu()
            w = ff.layer_norm.weight.detach().double().cpu()
            n = xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + ff.layer_norm.variance_epsilon) * w
            t = oracle.gated_act_oracle(n @ a.wi_0.weight.detach().double().cpu().t(), n @ a.wi_1.weight.detach().double().cpu().t())
            ref = xf + t @ ff.wo.weight.detach().double().cpu().t()
    for fuse in (False, True):
        err = (outs[fuse][0].double().cpu() - ref).abs().max().item()
        assert err <= 3 * 2.0 ** -8 * ref.abs().max().item(), (fuse, err)
    assert float((outs[True][1] - outs[False][1]).abs().max()) <= 3 * 2.0 ** -8 * float(outs[False][1].abs().max())
