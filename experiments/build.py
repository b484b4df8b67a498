"""Builds experiments/lib/libmdl_hip_exp.so: the product sources (matdeeplearn_amd/csrc/*.hip) compiled with
-DMDL_EXPERIMENTS=1, which adds the measured-negative kernel variants of experiments/csrc/ (cooperative weight-stationary
CGConv kernels, the first edge-per-lane backward, the saved-gate pair, the W-split pair, the two-layer dense kernel, the
dynamic tail of kernel 2 with -DMDL_EP2_TAIL=25) and their environment switches.  Not part of build(); run by hand:

    python experiments/build.py [extra hipcc flags]
    MDL_HIP_LIB=experiments/lib/libmdl_hip_exp.so python -m pytest experiments/test_experiments.py -m gpu
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from matdeeplearn_amd import _build  # noqa: E402

LIB = os.path.join(HERE, "lib", "libmdl_hip_exp.so")


def build(extra=()):
    cc = _build._hipcc()
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)

    def one(src):
        base = os.path.basename(src)
        obj = os.path.join(objdir, base[:-4] + ".o")
        cmd = [cc] + _build.FLAGS + _build.FILE_FLAGS.get(base, []) + ["-DMDL_EXPERIMENTS=1"] + list(extra) + ["-c", src, "-o", obj]
        print("[exp build]", " ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, _build.sources()))
    subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, check=True)
    return LIB


if __name__ == "__main__":
    print(build(sys.argv[1:]))
