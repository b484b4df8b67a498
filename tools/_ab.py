"""Tooling-only: A/B scripts select another build of libmdl_hip.so with MDL_HIP_LIB=<path> in THEIR environment.  The package
itself reads no environment variable (matdeeplearn_amd._lib.use_library is the explicit call); this shim translates."""
import os


def apply():
    path = os.environ.get("MDL_HIP_LIB")
    if path:
        from matdeeplearn_amd import _lib
        _lib.use_library(path)
    opts = os.environ.get("MDL_OPS")                      # e.g. MDL_OPS="rsrc16=0,balance=0"
    if opts:
        from matdeeplearn_amd import ops
        ops.configure(**{kv.split("=")[0]: kv.split("=")[1] not in ("0", "false", "False") for kv in opts.split(",") if kv})
