"""Build libfat5.so (gfx950) in-tree with hipcc.  `python flasht5_amd/build.py [--force]` (run as a script: importing the package needs the library).

The built library lives at flasht5_amd/lib/libfat5.so (git-ignored, travels to the GPU box).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# developer A/B builds: FAT5_VARIANT=name FAT5_EXTRA_FLAGS="-DX=1" python flasht5_amd/build.py  -> lib/libfat5_name.so
VARIANT = os.environ.get("FAT5_VARIANT", "")
OBJ = os.path.join(HERE, "lib", "obj" + ("_" + VARIANT if VARIANT else ""))
LIB = os.path.join(HERE, "lib", "libfat5" + ("_" + VARIANT if VARIANT else "") + ".so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950 has a unified register file); without it every
# softmax / rescale operand costs a v_accvgpr_read/_write (25 % of the forward loop's VALU issue slots).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         "-Wno-unused-value", "-I", INCLUDE] + os.environ.get("FAT5_EXTRA_FLAGS", "").split()

# (source, object name, extra defines)
UNITS = [("fat5_api.hip", "fat5_api.o", [])]
for d in (32, 64, 128):
    UNITS.append(("attn_fwd_inst.hip", f"attn_fwd_d{d}.o", [f"-DFAT5_INST_D={d}"]))
    UNITS.append(("attn_bwd_inst.hip", f"attn_bwd_d{d}.o", [f"-DFAT5_INST_D={d}"]))
UNITS.append(("attn_fwd64_inst.hip", "attn_fwd64_d64.o", ["-DFAT5_INST_D=64"]))
UNITS.append(("attn_fwd64_inst.hip", "attn_fwd64_d128.o", ["-DFAT5_INST_D=128"]))
UNITS.append(("attn_bwd64_inst.hip", "attn_bwd64_d64.o", ["-DFAT5_INST_D=64"]))
UNITS.append(("attn_bwd_qdb64_inst.hip", "attn_bwd_qdb64_d64.o", []))


def _deps_mtime():
    m = 0.0
    for root in (CSRC, INCLUDE):
        for f in os.listdir(root):
            if f.endswith((".h", ".hip")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return max(m, os.path.getmtime(os.path.abspath(__file__)))


def _compile(unit):
    src, obj, defs = unit
    cmd = [HIPCC] + FLAGS + defs + ["-c", os.path.join(CSRC, src), "-o", os.path.join(OBJ, obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {obj}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return obj


def build_lib(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    newest = _deps_mtime()
    todo = [u for u in UNITS
            if force or not os.path.exists(os.path.join(OBJ, u[1])) or os.path.getmtime(os.path.join(OBJ, u[1])) < newest]
    if todo:
        if verbose:
            print(f"[fat5 build] compiling {len(todo)} unit(s) for gfx950 ...", flush=True)
        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            for obj in ex.map(_compile, todo):
                if verbose:
                    print(f"[fat5 build]   {obj}", flush=True)
    objs = [os.path.join(OBJ, u[1]) for u in UNITS]
    if todo or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[fat5 build] linked {LIB}", flush=True)
        # every kernel stub must resolve (a silently skipped template instantiation shows up as an undefined symbol)
        chk = subprocess.run([sys.executable, "-c", f"import ctypes; ctypes.CDLL({LIB!r})"], capture_output=True, text=True)
        if chk.returncode != 0:
            os.remove(LIB)
            raise RuntimeError(f"{LIB} does not load:\n{chk.stderr[-2000:]}")
    return LIB


TORCH_EXT = os.path.join(HERE, "lib", "_fat5_torch.so")


def build_torch_binding(force=False, verbose=True):
    """The native host path (csrc/torch_binding.cpp: at::Tensor -> C ABI, C++ autograd functions) as a Python extension module,
    compiled with g++ against the installed torch headers (host code only; it links libfat5.so through $ORIGIN)."""
    src = os.path.join(CSRC, "torch_binding.cpp")
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(INCLUDE, "fat5.h")))
    if not force and os.path.exists(TORCH_EXT) and os.path.getmtime(TORCH_EXT) >= newest:
        return TORCH_EXT
    import sysconfig
    import torch
    import pybind11
    tdir = os.path.dirname(torch.__file__)
    if verbose:
        print("[fat5 build] compiling the torch binding (g++, ~1-2 min) ...", flush=True)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-DTORCH_EXTENSION_NAME=_fat5_torch",
           "-I", os.path.join(tdir, "include"), "-I", os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
           "-I", "/opt/rocm/include", "-I", sysconfig.get_paths()["include"], "-I", pybind11.get_include(), "-I", INCLUDE,
           src, "-o", TORCH_EXT, "-L", os.path.join(tdir, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
           "-ltorch_python", "-L", os.path.dirname(LIB), "-l:" + os.path.basename(LIB), "-L", "/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.join(tdir, "lib"), "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"torch binding failed to build:\n{' '.join(cmd)}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}")
    if verbose:
        print(f"[fat5 build] linked {TORCH_EXT}", flush=True)
    return TORCH_EXT


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    if "--no-torch" not in sys.argv and not VARIANT:
        build_torch_binding(force="--force" in sys.argv)
