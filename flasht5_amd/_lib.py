"""ctypes binding of libfat5.so (the C ABI declared in include/fat5.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing the
ops raises -- a GPU box must never silently run something else.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# FAT5_LIB_VARIANT selects a developer A/B build (lib/libfat5_<name>.so); unset = the product library
_VAR = os.environ.get("FAT5_LIB_VARIANT", "")
LIB_PATH = os.path.join(_HERE, "lib", "libfat5" + ("_" + _VAR if _VAR else "") + ".so")

FAT5_F32, FAT5_F16, FAT5_BF16 = 0, 1, 2
BIAS_NONE, BIAS_DENSE, BIAS_RPE1D = 0, 1, 2
MAX_RPE_RADIUS = 1024  # forward alone takes 2048; the backward's per-wave diagonal accumulators must fit LDS (include/fat5.h)

_DT = {torch.float32: FAT5_F32, torch.float16: FAT5_F16, torch.bfloat16: FAT5_BF16}

c_i64x3 = ctypes.c_int64 * 3


class AttnParams(ctypes.Structure):
    """Mirror of `fat5_attn_params` (include/fat5.h) -- field order must match exactly."""
    _fields_ = [
        ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("M", ctypes.c_int32), ("N", ctypes.c_int32),
        ("D", ctypes.c_int32), ("dtype", ctypes.c_int32), ("causal", ctypes.c_int32),
        ("bias_mode", ctypes.c_int32), ("sm_scale", ctypes.c_float), ("rpe_radius", ctypes.c_int32),
        ("q", ctypes.c_void_p), ("k", ctypes.c_void_p), ("v", ctypes.c_void_p), ("o", ctypes.c_void_p),
        ("lse", ctypes.c_void_p),
        ("q_stride", c_i64x3), ("k_stride", c_i64x3), ("v_stride", c_i64x3), ("o_stride", c_i64x3),
        ("bias", ctypes.c_void_p), ("bias_stride", c_i64x3), ("rpe1d", ctypes.c_void_p),
        ("cu_seqlens_q", ctypes.c_void_p), ("cu_seqlens_k", ctypes.c_void_p),
        ("total_q", ctypes.c_int32), ("total_k", ctypes.c_int32),
        ("dout", ctypes.c_void_p), ("dq", ctypes.c_void_p), ("dk", ctypes.c_void_p), ("dv", ctypes.c_void_p),
        ("do_stride", c_i64x3), ("dq_stride", c_i64x3), ("dk_stride", c_i64x3), ("dv_stride", c_i64x3),
        ("dbias", ctypes.c_void_p), ("dbias_batch", ctypes.c_int32), ("dbias_heads", ctypes.c_int32),
        ("drpe1d", ctypes.c_void_p), ("rpe_bucket", ctypes.c_void_p), ("drpe_table", ctypes.c_void_p),
        ("rpe_num_buckets", ctypes.c_int32), ("unit_begin", ctypes.c_int32), ("unit_count", ctypes.c_int32),
        ("variant", ctypes.c_int32),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
    ]


EXPORTS = (
    "fat5_version", "fat5_chip_cus", "fat5_last_error", "fat5_sizeof_attn_params", "fat5_attn_fwd", "fat5_attn_bwd_workspace_bytes", "fat5_attn_bwd", "fat5_attn_bwd_launches",
    "fat5_attn_bwd_stages", "fat5_attn_describe", "fat5_rpe1d_from_table",
    "fat5_rmsnorm_fwd", "fat5_rmsnorm_bwd_workspace_bytes", "fat5_rmsnorm_bwd", "fat5_add_rmsnorm_fwd", "fat5_add_rmsnorm_bwd",
    "fat5_ce_fwd", "fat5_ce_bwd", "fat5_ce_fwd_bwd", "fat5_fold_weights", "fat5_fold_weights_bwd", "fat5_fold_weights_bwd_scratch_bytes", "fat5_rmsnorm_unit_bwd", "fat5_gated_act_fwd", "fat5_gated_act_bwd",
    "fat5_adamw_scale_step", "fat5_adamw_scale_step_clipped", "fat5_adamw_scale_step_dev", "fat5_adamw_grad_sumsq", "fat5_sizeof_adamw_tensor",
)

_lib = None


def load():
    """Load libfat5.so once; raise loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python flasht5_amd/build.py` (hipcc, gfx950). "
            "flasht5_amd has no non-HIP fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.fat5_version.restype = ctypes.c_int
    lib.fat5_last_error.restype = ctypes.c_char_p
    lib.fat5_attn_fwd.restype = ctypes.c_int
    lib.fat5_attn_fwd.argtypes = [ctypes.POINTER(AttnParams), ctypes.c_void_p]
    lib.fat5_attn_bwd.restype = ctypes.c_int
    lib.fat5_attn_bwd.argtypes = [ctypes.POINTER(AttnParams), ctypes.c_void_p]
    lib.fat5_attn_bwd_stages.restype = ctypes.c_int
    lib.fat5_attn_bwd_stages.argtypes = [ctypes.POINTER(AttnParams), ctypes.c_int, ctypes.c_void_p]
    lib.fat5_attn_bwd_launches.restype = ctypes.c_int
    lib.fat5_attn_bwd_launches.argtypes = [ctypes.POINTER(AttnParams)]
    lib.fat5_attn_describe.restype = ctypes.c_int
    lib.fat5_attn_describe.argtypes = [ctypes.POINTER(AttnParams), ctypes.c_char_p, ctypes.c_size_t]
    lib.fat5_attn_bwd_workspace_bytes.restype = ctypes.c_size_t
    lib.fat5_attn_bwd_workspace_bytes.argtypes = [ctypes.POINTER(AttnParams)]
    lib.fat5_rpe1d_from_table.restype = ctypes.c_int
    lib.fat5_rpe1d_from_table.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_void_p]
    i64, f32, vp, i32 = ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_int
    lib.fat5_rmsnorm_fwd.restype = ctypes.c_int
    lib.fat5_rmsnorm_fwd.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, f32, i32, i32, vp]
    lib.fat5_rmsnorm_bwd_workspace_bytes.restype = ctypes.c_size_t
    lib.fat5_rmsnorm_bwd_workspace_bytes.argtypes = [i64, i64]
    lib.fat5_rmsnorm_bwd.restype = ctypes.c_int
    lib.fat5_rmsnorm_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i32, i32, vp, ctypes.c_size_t, vp]
    lib.fat5_add_rmsnorm_fwd.restype = ctypes.c_int
    lib.fat5_add_rmsnorm_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, f32, i32, i32, vp]
    lib.fat5_add_rmsnorm_bwd.restype = ctypes.c_int
    lib.fat5_add_rmsnorm_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i32, i32, vp, ctypes.c_size_t, vp]
    lib.fat5_ce_fwd.restype = ctypes.c_int
    lib.fat5_ce_fwd.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, f32, f32, f32, i64, i32, i32, vp]
    lib.fat5_fold_weights_bwd.restype = ctypes.c_int
    lib.fat5_fold_weights_bwd.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, vp, vp, vp, i64, i32, vp, ctypes.c_size_t, vp]
    lib.fat5_fold_weights_bwd_scratch_bytes.restype = ctypes.c_size_t
    lib.fat5_gated_act_fwd.restype = ctypes.c_int
    lib.fat5_gated_act_fwd.argtypes = [vp, vp, vp, i64, i64, i64, i64, i64, i32, i32, vp]
    lib.fat5_gated_act_bwd.restype = ctypes.c_int
    lib.fat5_gated_act_bwd.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, i32, vp]
    lib.fat5_fold_weights_bwd_scratch_bytes.argtypes = [i64, i64]
    lib.fat5_rmsnorm_unit_bwd.restype = ctypes.c_int
    lib.fat5_rmsnorm_unit_bwd.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, vp, i64, i32, vp]
    lib.fat5_fold_weights.restype = ctypes.c_int
    lib.fat5_fold_weights.argtypes = [vp, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, i64, i32, vp]
    lib.fat5_ce_bwd.restype = ctypes.c_int
    lib.fat5_ce_bwd.argtypes = [vp, i64, vp, vp, vp, vp, i64, i64, i64, i64, f32, f32, f32, i64, i32, vp]
    lib.fat5_ce_fwd_bwd.restype = ctypes.c_int
    lib.fat5_ce_fwd_bwd.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, i64, i64, i64, i64, f32, f32, f32, i64, i32, vp]
    lib.fat5_adamw_scale_step.restype = ctypes.c_int
    f64 = ctypes.c_double
    lib.fat5_adamw_scale_step.argtypes = [vp, i32, i32, vp, f64, f64, f64, f64, f64, i32, i32, i32, vp]
    lib.fat5_adamw_scale_step_clipped.restype = ctypes.c_int
    lib.fat5_adamw_scale_step_dev.restype = ctypes.c_int
    lib.fat5_adamw_scale_step_dev.argtypes = [vp, i32, i32, vp, vp, f64, f64, f64, i32, i32, i32, vp, vp]
    lib.fat5_adamw_scale_step_clipped.argtypes = [vp, i32, i32, vp, f64, f64, f64, f64, f64, i32, i32, i32, vp, vp]
    lib.fat5_adamw_grad_sumsq.restype = ctypes.c_int
    lib.fat5_adamw_grad_sumsq.argtypes = [vp, i32, i32, vp, i32, vp]
    lib.fat5_sizeof_adamw_tensor.restype = ctypes.c_size_t
    lib.fat5_sizeof_attn_params.restype = ctypes.c_size_t
    if lib.fat5_sizeof_attn_params() != ctypes.sizeof(AttnParams):
        raise ImportError(f"fat5_attn_params layout mismatch: library {lib.fat5_sizeof_attn_params()} B, "
                          f"binding {ctypes.sizeof(AttnParams)} B")
    _lib = lib
    return lib


# fat5_attn_params.variant bits (include/fat5.h `enum fat5_variant`): tests and profilers force / forbid a kernel body per call
V_FWD64_ON, V_FWD64_OFF, V_KV64_ON, V_KV64_OFF, V_Q64_ON, V_Q64_OFF = 1, 2, 4, 8, 16, 32
V_DBIAS_STAGED, V_DBIAS_INKERNEL, V_NO_FUSE, V_NO_SPLIT = 64, 128, 256, 512
V_FWD64_KSPLIT_ON, V_FWD64_KSPLIT_OFF = 1024, 2048
V_KV64_HALF_ON, V_KV64_HALF_OFF = 4096, 8192
V_KV64_MIX_ON, V_KV64_MIX_OFF = 16384, 32768
V_FWD64_MIX_ON, V_FWD64_MIX_OFF = 524288, 1048576
V_FUSED64_ON, V_FUSED64_OFF = 65536, 131072
V_DBIAS_NOSPLIT = 262144
V_QDB64_ON, V_QDB64_OFF = 2097152, 4194304
V_QDIAG_ON, V_QDIAG_OFF = 8388608, 16777216  # T5 bias, one-launch backward: the per-diagonal sums of the table gradient in the dQ workgroups always / never
_variant = 0  # what the host mirror writes into every descriptor it builds; 0 = the library's own choice (production)


def set_variant(bits):
    """Test / profiling hook: OR of V_* bits carried by every subsequent attention call of this process (ctypes path and the
    C++ binding alike); returns the previous value.  Production code never calls this."""
    global _variant
    prev, _variant = _variant, int(bits)
    nat = native()
    if nat is not None:
        nat.set_variant(int(bits))
    return prev


class variant:
    """`with _lib.variant(_lib.V_FWD64_ON | _lib.V_KV64_ON): ...`"""

    def __init__(self, bits):
        self.bits = bits

    def __enter__(self):
        self.prev = set_variant(self.bits)

    def __exit__(self, *exc):
        set_variant(self.prev)
        return False


_native = False  # (False: not looked up yet; None: unavailable / disabled)


def native():
    """The C++ host path (csrc/torch_binding.cpp -> lib/_fat5_torch.so: at::Tensor in / out, C++ autograd functions on the C ABI),
    or None when it is not built or FAT5_TORCH_BINDING=0 -- the ctypes path below is then used for eager calls as well (same
    kernels, ~10x the host time per call)."""
    global _native
    if _native is False:
        _native = None
        if os.environ.get("FAT5_TORCH_BINDING", "1") != "0" and not _VAR:
            path = os.path.join(_HERE, "lib", "_fat5_torch.so")
            if os.path.exists(path):
                import importlib.util
                load()  # (libfat5.so first: same image for both paths)
                spec = importlib.util.spec_from_file_location("_fat5_torch", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                if mod.sizeof_attn_params() != ctypes.sizeof(AttnParams):
                    raise ImportError("lib/_fat5_torch.so was built against another include/fat5.h: rebuild (python flasht5_amd/build.py)")
                _native = mod
    return _native


def check(rc, what):
    if rc != 0:
        msg = load().fat5_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (fat5 status {rc}): {msg}")


def dtype_code(dt):
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError(f"unsupported dtype {dt}") from None


def stream_ptr(device):
    """raw hipStream_t of torch's current stream on `device` (the C ABI launches there, asynchronously)"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


class on_device:
    """`with on_device(t.device):` -- make the tensor's device current for the C-ABI call (the reference wraps its launches
    the same way, flash_attention_v2_bias.py:61,:128); a no-op, without touching the runtime, when it already is."""
    __slots__ = ("idx", "prev")

    def __init__(self, device):
        self.idx = device.index if device.index is not None else -1
        self.prev = -1

    def __enter__(self):
        if self.idx >= 0:
            cur = torch.cuda.current_device()
            if cur != self.idx:
                self.prev = cur
                torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


def strides3(t):
    """(b, h, s) element strides of a 4-D (B,H,S,D) tensor as a ctypes array."""
    return c_i64x3(t.stride(0), t.stride(1), t.stride(2))


def kernel_ready(t):
    """The kernels need last-dim stride 1, 16-byte aligned base and strides that are multiples of 8."""
    return (t.stride(-1) == 1 and t.data_ptr() % 16 == 0 and all(s % 8 == 0 for s in t.stride()[:-1]))


def describe(B, H, M, N, D=64, dtype=FAT5_BF16, causal=False, bias_mode=0, radius=0, need_dbias=False, variant=0, sm_scale=None):
    """Which kernel bodies the library would run for a problem -- {"fwd": "64row-ksplit", "dq": "32row", "dkdv": "64key-mixed:4",
    "fused": "0", "dbias": "direct"} -- from shapes alone (fat5_attn_describe: host-only, works without a GPU)."""
    p = AttnParams()
    p.B, p.H, p.M, p.N, p.D = B, H, M, N, D
    p.dtype, p.causal, p.bias_mode, p.rpe_radius, p.variant = dtype, int(causal), bias_mode, radius, variant
    p.sm_scale = float(D) ** -0.5 if sm_scale is None else float(sm_scale)  # (a zero scale keeps the dense bias off the matrix pipe)
    if bias_mode == BIAS_DENSE:
        p.bias = 16  # (never followed)
        p.bias_stride[0], p.bias_stride[1], p.bias_stride[2] = 0, M * N, N  # the model's (1, H, M, N) bias, shared by the batch
        if need_dbias:
            p.dbias, p.dbias_batch, p.dbias_heads = 16, 1, H
    elif bias_mode == BIAS_RPE1D:
        p.rpe1d = 16
        if need_dbias:
            p.drpe1d = 16
    buf = ctypes.create_string_buffer(256)
    check(load().fat5_attn_describe(ctypes.byref(p), buf, 256), "fat5_attn_describe")
    return dict(kv.split("=") for kv in buf.value.decode().split())
