"""Developer A/B builds that differ from the product library in ONE translation unit (minutes saved: the other objects are copied).
    python tools/build_variant.py NAME UNIT.o "EXTRA FLAGS"      e.g.  python tools/build_variant.py trace attn_bwd64_d64.o "-DFAT5_TRACE=1"
-> flasht5_amd/lib/libfat5_NAME.so (select with FAT5_LIB_VARIANT=NAME).  The product objects must be current (python flasht5_amd/build.py)."""
import importlib.util, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, unit, flags = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
os.environ["FAT5_VARIANT"], os.environ["FAT5_EXTRA_FLAGS"] = name, flags
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "flasht5_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
base = os.path.join(ROOT, "flasht5_amd", "lib", "obj")
shutil.rmtree(b.OBJ, ignore_errors=True)
shutil.copytree(base, b.OBJ)
b._compile([u for u in b.UNITS if u[1] == unit][0])
r = subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", b.LIB] + [os.path.join(b.OBJ, u[1]) for u in b.UNITS], capture_output=True, text=True)
print(name, "->", b.LIB if r.returncode == 0 else r.stderr[-2000:])
