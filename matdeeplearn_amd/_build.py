"""Builds libmdl_hip.so (gfx950) from matdeeplearn_amd/csrc/*.hip with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
matdeeplearn_amd/lib/libmdl_hip.so is git-ignored but travels to the GPU box with the tree.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(LIBDIR, "libmdl_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]
# per-file extras: the edge-per-lane CGConv backward wants its MFMA results in VGPRs (see csrc/cgconv.hip)
FILE_FLAGS = {"cgconv_ep.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "cfconv.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
# translation units that #include another .hip file
FILE_DEPS = {}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build libmdl_hip.so)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "mdl_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources()) or _deps_mtime() > t


def build(force=False, verbose=True):
    """Compile every .hip translation unit for gfx950 and link the shared library."""
    if not force and not needs_build():
        return LIB
    cc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_t = _deps_mtime()

    def compile_one(src):
        base = os.path.basename(src)
        obj = os.path.join(OBJDIR, base[:-4] + ".o")
        src_t = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(CSRC, d)) for d in FILE_DEPS.get(base, [])])
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(src_t, hdr_t):
            cmd = [cc] + FLAGS + FILE_FLAGS.get(base, []) + ["-c", src, "-o", obj]
            if verbose:
                print("[mdl build]", " ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print("[mdl build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
