"""Loader for the committed golden fixtures (tests/golden/*.npz)."""
import os
import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DTYPES = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}


def _t(a, dtype=None):
    if a.dtype == np.uint16:  # raw bf16 bits
        return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
    return torch.from_numpy(a.copy())


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def load_attn(name):
    z = load(name)
    B, H, M, N, D, causal, dt = (int(x) for x in z["meta"])
    dtype = DTYPES[dt]
    out = {"B": B, "H": H, "M": M, "N": N, "D": D, "causal": bool(causal), "dtype": dtype,
           "sm_scale": float(z["sm_scale"][0])}
    for key in z:
        if key in ("meta", "sm_scale"):
            continue
        out[key] = _t(z[key])
    if dtype == torch.float32:  # cfg1: inputs stored as bf16 bits, used as fp32
        for key in ("q", "k", "v", "bias", "do"):
            if key in out:
                out[key] = out[key].float()
    out.setdefault("bias", None)
    return out


ATTN_CASES = [
    "attn_t164_nc_1h_bf16", "attn_t164_nc_1h_fp16", "attn_t100_c_bh_bf16", "attn_t100_c_bh_fp16",
    "attn_mgtn_nc_bh_bf16", "attn_mgtn_c_1h_fp16", "attn_nobias_c_bf16", "attn_11_nc_bf16",
    "attn_d128_nc_bf16", "attn_d32_c_fp16",
]
TRITON_CASES = ["triton_t80_nc_1h_fp16", "triton_t80_c_bh_fp16", "triton_t100_nc_1h_fp16",
                "triton_d128_nc_1h_fp16", "triton_d32_c_1h_fp16", "triton_mgtn_nc_bh_fp16"]  # (round 3: D = 128, D = 32, M > N per-batch bias)
