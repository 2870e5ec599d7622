"""developer fuzz of the dense-bias bodies of round 5 (forced on): random ragged shapes, batch sizes 1 .. 9 (idle waves, fp32 slabs beyond four), causal or not, scales incl. the
reference benchmark's 1.3 and negative ones, bf16 / fp16 -- head_dim 64: forward (64-row body where legal), dQ + dBias (qdb64), the dense 64-key dK/dV body, as ONE launch or two
(random); head_dim 128: the pipelined forward with its bias ring (three ring slots).  o, dq, dk, dv, dbias against the oracle with the bounds of tests/test_dense64_gpu.py.
usage: python tools/fuzz_dense.py [n_cases] [seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from flasht5_amd import _lib
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from attn_helpers import make_inputs, oracle_all, maxdiff, eager_lowprec_errors
from test_attention_gpu import bound, gbound

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for i in range(n):
    D = 64 if rng.random() < 0.7 else 128
    B, H = rng.randint(1, 9), rng.choice([1, 2, 3])
    M, N = rng.randint(33, 1500), 8 * rng.randint(5, 190)  # (N % 8 == 0: the dense 64-wide bodies' condition)
    if rng.random() < 0.3:
        N = (M + 7) // 8 * 8
    causal = rng.random() < 0.4
    dtype = torch.bfloat16 if (D == 128 or rng.random() < 0.7) else torch.float16
    scale = rng.choice([0.125, 0.25, 1.0 / 3, 1.3, -0.5, 1.0, D ** -0.5])
    q, k, v, b, do = make_inputs(B, H, M, N, D, dtype, "1h", seed=2000 + i, strided=bool(i & 1))
    if abs(scale) > 0.5:
        q = (q.float() * 0.5).to(dtype)
    ref = oracle_all(q, k, v, b, do, scale, causal)
    one = rng.random() < 0.5
    if D == 64:
        bits = _lib.V_QDB64_ON | _lib.V_KV64_ON | (_lib.V_FUSED64_ON if one else _lib.V_FUSED64_OFF) | (_lib.V_FWD64_ON if rng.random() < 0.5 else 0)
    else:
        bits = _lib.V_FWD64_ON
    plan = AttentionPlan(q, k, v, do, bias=b, causal=causal, sm_scale=scale, variant=bits)
    plan.forward()
    plan.ws.view(torch.uint8).fill_(255)
    msgs = []
    lp = [None]  # beside the absolute bound the reference's own rule (tests/fa2_triton/test_fa2_bias.py): at most twice the error of eager attention in the input dtype
    def check(got, key, lim):
        e = maxdiff(got, ref[key])
        if torch.isfinite(got.float()).all() and e <= lim:
            return
        if lp[0] is None:
            lp[0] = eager_lowprec_errors(q, k, v, b, do, scale, causal, ref)
        if not torch.isfinite(got.float()).all() or e > 2 * lp[0][key] + 1e-5:
            msgs.append(f"{key} {e:.3e} > {lim:.3e} and > 2 x eager {lp[0][key]:.3e}")
    check(plan.o, "o", bound(ref["o"], dtype))
    if D == 64:
        dq, dk, dv, db = plan.backward()
        torch.cuda.synchronize()
        for got, key, mul in ((dq, "dq", 1), (dk, "dk", 1), (dv, "dv", 1), (db, "db", 1 + B)):
            check(got, key, gbound(ref[key], dtype) * mul)
    d = plan.describe()
    print(f"[{i}] D={D} B={B} H={H} M={M} N={N} causal={int(causal)} {str(dtype)[6:]} scale={scale:.4g} fwd={d['fwd']} dq={d['dq']} dkdv={d['dkdv']} fused={d['fused']}: " + ("OK" if not msgs else "FAIL " + "; ".join(msgs)), flush=True)
    bad += bool(msgs)
    del plan
print("fuzz_dense failures:", bad, "of", n)
