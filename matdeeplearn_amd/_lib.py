"""ctypes binding of libmdl_hip.so (include/mdl_hip.h).  There is NO CPU fallback: if the HIP
library is missing, or a tensor is not on a HIP device, the ops raise."""
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libmdl_hip.so")


def use_library(path):
    """Bind another build of libmdl_hip.so (A/B builds of tools/, the experiments library) — call before the first op.
    The package reads no environment variable; tools/ and the test harness translate their own MDL_HIP_LIB into this call."""
    global LIB_PATH, _lib
    if _lib is not None and os.path.abspath(path) != os.path.abspath(LIB_PATH):
        raise MdlError("use_library(%s): %s is already loaded in this process" % (path, LIB_PATH))
    LIB_PATH = path

MDL_F32, MDL_BF16 = 0, 1
MDL_SUM, MDL_MEAN, MDL_MAX = 0, 1, 2
# execution flags (include/mdl_hip.h): the `flags` field of the struct entry points, OR-ed into `dtype` for the positional ones
MDL_DTYPE_MASK, MDL_DETERMINISTIC, MDL_K3_PER_WAVE, MDL_K3_EDGE_LANE = 0xFF, 0x100, 0x200, 0x400
MDL_MLP_F32_IO = 0x2000        # mdl_mlp_head_fwd / _bwd: the last output / its gradient as fp32 rows
MDL_SPLIT_BF16 = 0x1000        # CGConv kernels on fp32 storage: the K = 2C + G product as three bf16 MFMAs on (hi, lo) operands
MDL_BN_SHIFT_ROW = 0x800       # mdl_bn_apply_n: sums about the shift row their producer stored behind the totals rows
MDL_BN_REPLICAS = 16
REDUCE = {"sum": MDL_SUM, "add": MDL_SUM, "mean": MDL_MEAN, "max": MDL_MAX}

_vp, _i64, _i32, _f32, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_u32 = ctypes.c_uint32


class MdlCgConv(ctypes.Structure):
    """include/mdl_hip.h: arguments of mdl_cgconv_fwd_ex / mdl_cgconv_bwd_ex (unused fields stay zero)"""
    _fields_ = [("size", _u32), ("dtype", _i32), ("flags", _u32), ("aggr", _i32), ("N", _i64), ("E", _i64), ("C", _i32), ("G", _i32),
                ("x", _vp), ("edge_attr", _vp), ("rowptr", _vp), ("src", _vp), ("tgt", _vp), ("eperm", _vp), ("wpack", _vp),
                ("bpack", _vp), ("out", _vp), ("bn_sums", _vp), ("bn_shift", _vp), ("bn_rows", _vp), ("grad_out", _vp),
                ("r_tgt", _vp), ("r_src", _vp), ("r_src_dtype", _i32), ("ld_dwe", _i32), ("dwe", _vp), ("db", _vp),
                ("workspace", _vp), ("workspace_bytes", _sz), ("balance", _vp)]


class MdlCgNode(ctypes.Structure):
    """include/mdl_hip.h: arguments of mdl_cgconv_bwd_node_ex"""
    _fields_ = [("size", _u32), ("dtype", _i32), ("flags", _u32), ("zero_src", _i32), ("N", _i64), ("C", _i32), ("r_src_dtype", _i32),
                ("ld_dwn", _i32), ("reserved", _i32), ("x", _vp), ("grad_out", _vp), ("r_tgt", _vp), ("r_src", _vp), ("wn_t", _vp), ("dx", _vp), ("dwn", _vp)]


def _dp(t):
    return None if t is None else t.data_ptr()


def cg_args(**kw):
    """MdlCgConv from keyword arguments; tensors become device pointers"""
    a = MdlCgConv()
    a.size = ctypes.sizeof(MdlCgConv)
    for k, v in kw.items():
        setattr(a, k, _dp(v) if torch.is_tensor(v) else v)
    return a


def cg_node_args(**kw):
    a = MdlCgNode()
    a.size = ctypes.sizeof(MdlCgNode)
    for k, v in kw.items():
        setattr(a, k, _dp(v) if torch.is_tensor(v) else v)
    return a

# name -> (restype, argtypes); kept in one table so tests can check it against include/mdl_hip.h
PROTOTYPES = {
    "mdl_version": (_i32, []),
    "mdl_last_error_string": (ctypes.c_char_p, []),
    "mdl_debug_last_k3": (_i32, []),
    "mdl_rbf_expand": (_i32, [_vp, _vp, _f32, _vp, _i64, _i32, _i64, _i32, _vp]),
    "mdl_csr_rowptr": (_i32, [_vp, _i64, _i64, _vp, _vp]),
    "mdl_segment_reduce_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "mdl_segment_reduce_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp]),
    "mdl_segment_reduce_bwd_add": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp]),
    "mdl_cgconv_wpack_bytes": (_sz, [_i32, _i32, _i32]),
    "mdl_cgconv_pack_weights": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    "mdl_cgconv_pack_weights_node": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "mdl_cgconv_pack_weights_multi": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "mdl_cgconv_fwd": (_i32, [_vp] * 9 + [_i64, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mdl_cgconv_fwd_ex": (_i32, [ctypes.POINTER(MdlCgConv), _vp]),
    "mdl_cgconv_bwd_ex": (_i32, [ctypes.POINTER(MdlCgConv), _vp]),
    "mdl_cgconv_bwd_node_ex": (_i32, [ctypes.POINTER(MdlCgNode), _vp]),
    "mdl_cgconv_workspace_bytes": (ctypes.c_size_t, [_i64, _i64, _i32, _i32, _i32]),
    "mdl_cgconv_balance": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "mdl_cgconv_pack_node_weights": (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _vp]),
    "mdl_cgconv_assemble_grads": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mdl_assemble_batch": (_i32, [_vp] * 20 + [_i32, _i32, _i32, _i32, _i32, _vp]),
    "mdl_assemble_batch_padded": (_i32, [_vp] * 20 + [_i32, _i32, _i32, _i32, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "mdl_nnconv_msg_fwd": (_i32, [_vp] * 5 + [_i64, _i32, _i32, _i32, _vp]),
    "mdl_nnconv_msg_bwd": (_i32, [_vp] * 7 + [_i64, _i32, _i32, _i32, _vp]),
    "mdl_pad_batch_tail": (_i32, [_vp, _vp, _i32, _i64, _vp, _vp, _vp]),
    "mdl_assemble_transposed": (_i32, [_vp] * 13 + [_i32, _i64, _vp]),
    "mdl_pad_edge_tail": (_i32, [_vp, _vp, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mdl_loss_fwd_bwd": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "mdl_loss_fwd_bwd_rows": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "mdl_bn_sums_rows": (_i32, []),
    "mdl_bn_stats_n": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _vp]),
    "mdl_bn_apply_n": (_i32, [_vp] * 8 + [_i64, _i32, _f32, _f32, _vp, _i32, _vp]),
    "mdl_bn_bwd_stats_n": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _vp]),
    "mdl_bn_bwd_apply_n": (_i32, [_vp] * 6 + [_i64, _i32, _vp, _i32, _vp]),
    "mdl_bn_bwd_apply_relu_n": (_i32, [_vp] * 6 + [_i64, _i32, _vp, _i32, _vp]),
    "mdl_linear_act": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mdl_dense_bwd": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _i32, _vp]),
    "mdl_dense_bwd_ex": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "mdl_tn_scratch_bytes": (_sz, []),
    "mdl_linear_act_stats": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "mdl_linear_act_in": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mdl_ssp_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "mdl_linear_gather_act": (_i32, [_vp] * 10 + [_i64, _i32, _i32, _i32, _i32, _vp]),
    "mdl_linear_wide": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _i32, _vp]),
    "mdl_mlp_head_fwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i32, _vp]),
    "mdl_mlp_head_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i32, _vp]),
    "mdl_gru_gates_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "mdl_gru_gates_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "mdl_gemm_tn": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _vp]),
    "mdl_split_bf16": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "mdl_gemm_tn_colsum": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp]),
    "mdl_gemm_tn_act": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp]),
    "mdl_gemm_tn_ex": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _i32, _vp]),
    "mdl_gather_rows": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "mdl_gather_mul_reduce": (_i32, [_vp] * 7 + [_i64, _i64, _i32, _i32, _vp]),
    "mdl_gather_mul_reduce_dw": (_i32, [_vp] * 9 + [_i64, _i64, _i32, _vp]),
    "mdl_edge_mul": (_i32, [_vp] * 6 + [_i64, _i64, _i32, _vp]),
    "mdl_cfconv_supported": (_i32, [_i32, _i32, _i32]),
    "mdl_cfconv_wpack_bytes": (_sz, []),
    "mdl_cfconv_pack_weights": (_i32, [_vp] * 4 + [_i32, _i32, _vp, _vp]),
    "mdl_cfconv_fwd": (_i32, [_vp] * 10 + [_i64, _i64, _i32, _i32, _i32, _vp]),
    "mdl_cfconv_bwd_w_scratch_bytes": (_sz, []),
    "mdl_cfconv_bwd_w": (_i32, [_vp] * 13 + [_i64, _i64, _i32, _i32, _i32, _vp]),
}

_lib = None


class MdlError(RuntimeError):
    pass


def lib():
    """Load libmdl_hip.so once.  Fails loudly: the product has no other compute path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MdlError(
                "libmdl_hip.so not found at %s — build it with `python -m matdeeplearn_amd._build` "
                "(hipcc --offload-arch=gfx950).  matdeeplearn_amd has no CPU/eager fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise MdlError("%s failed (%d): %s" % (what, rc, lib().mdl_last_error_string().decode()))


def dtype_code(t):
    if t.dtype == torch.float32:
        return MDL_F32
    if t.dtype == torch.bfloat16:
        return MDL_BF16
    raise MdlError("unsupported dtype %s (float32 or bfloat16)" % t.dtype)


def require_hip(*tensors):
    """Every op starts here: operands must live on a HIP device, and on the CURRENT one — the kernels are launched on
    torch.cuda.current_stream() of the current device, so a tensor of another device would be touched from a foreign
    stream (one process per GPU: call torch.cuda.set_device(rank) first; ddp_setup / train_regular do)."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise MdlError("matdeeplearn_amd ops need tensors on a HIP device (got %s); there is no CPU path — "
                           "the CPU restatement lives in oracle/ and is test infrastructure only" % t.device)
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise MdlError("tensor on %s but the current HIP device is cuda:%d — call torch.cuda.set_device(%d) in "
                           "this process before using matdeeplearn_amd" % (t.device, cur, t.device.index))


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    """Raw hipStream_t of torch's current stream on the current device (no Stream object: this runs once per launch)."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
