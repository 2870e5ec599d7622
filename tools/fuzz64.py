"""developer fuzz of the 64-wide bodies (forward, dK/dV, dQ forced on): random ragged shapes, causal or not, no bias / T5 bias with a
random radius; o, dq, dk, dv and the table gradient against the oracle with the bounds of tests/test_bwd64_gpu.py.
usage: [FUZZ_FORCE64=0] python tools/fuzz64.py [n_cases] [seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from flasht5_amd import _lib
if os.environ.get("FUZZ_FORCE64", "1") != "0":  # FUZZ_FORCE64=0: the default dispatch (32-row bodies at these sizes)
    # FUZZ_VARIANT: further fat5_variant bits, e.g. 8454144 = FUSED64_ON | QDIAG_ON (round 6: the one-launch backward with the table gradient's diagonal sums in the dQ workgroups)
    _lib.set_variant(_lib.V_FWD64_ON | _lib.V_KV64_ON | _lib.V_Q64_ON | int(os.environ.get("FUZZ_VARIANT", "0")))
import oracle
from attn_helpers import make_inputs, oracle_all, maxdiff, eager_lowprec_errors
from test_attention_gpu import bound, gbound, _rpe_case
from test_bwd64_gpu import _grads, _table_truth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for i in range(n):
    B, H = rng.choice([1, 2]), rng.choice([1, 2, 3])
    M, N = rng.randint(33, 2600), rng.randint(33, 2600)
    if rng.random() < 0.3:
        N = M
    causal = rng.random() < 0.4
    mode = rng.choice(["none", "rpe", "rpe"])
    md = rng.choice([32, 64, 128, 128, 256])
    dtype = torch.bfloat16 if rng.random() < 0.8 else torch.float16
    scale = rng.choice([0.125, 0.25, 1.0 / 3])
    if mode == "rpe":
        q, k, v, do, table, bias = _rpe_case(B, H, M, N, dtype, causal, True, md, seed=1000 + i)
    else:
        q, k, v, _, do = make_inputs(B, H, M, N, 64, dtype, None, seed=1000 + i, strided=bool(i & 1))
        table, bias = None, None
    ref = oracle_all(q, k, v, bias, do, scale, causal)
    got = _grads(q, k, v, do, causal, scale, table, True, md)
    msgs = []
    lp = None  # the reference's own rule beside the absolute bound: at most twice the error of eager attention in the input dtype
    for key in ("o", "dq", "dk", "dv"):
        lim = bound(ref[key], dtype) if key == "o" else gbound(ref[key], dtype)
        e = maxdiff(got[key], ref[key])
        if torch.isfinite(got[key].float()).all() and e <= lim:
            continue
        if lp is None:
            lp = eager_lowprec_errors(q, k, v, bias, do, scale, causal, ref)
        if not torch.isfinite(got[key].float()).all() or e > 2 * lp[key]:
            msgs.append(f"{key} {e:.3e} > {lim:.3e} and > 2 x eager {lp[key]:.3e}")
    if table is not None:
        want, allow = _table_truth(q, k, v, bias, got["o"], ref["L"], do, scale, causal, table, M, N, True, md)
        err = (got["dtable"].cpu() - want).abs()
        lim = allow + 2e-3 * max(1.0, want.abs().max().item()) + 1e-2
        if not bool((err <= lim).all()):
            msgs.append(f"dtable {err.max().item():.3e} (allow {lim.max().item():.3e})")
    tag = "OK " if not msgs else "BAD"
    bad += bool(msgs)
    print(f"{tag} case {i}: B={B} H={H} M={M} N={N} causal={int(causal)} {mode} md={md} {str(dtype)[6:]} scale={scale:.3f} {'; '.join(msgs)}", flush=True)
print(f"FUZZ64: {bad} bad of {n}")
