"""GPU parity tests of the EXPERIMENTS build (experiments/lib/libmdl_hip_exp.so = the product sources with
-DMDL_EXPERIMENTS=1): measured-negative kernel variants kept reproducible.  Not collected by `pytest tests/`; run with

    python experiments/build.py && python -m pytest experiments/test_experiments.py -m gpu -q

Every test runs its body in a fresh interpreter with MDL_HIP_LIB pointing at the experiments library (the product package
binds ONE library per process) and, where a variant is chosen by the experiments build's environment switches, with them set."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP_LIB = os.path.join(ROOT, "experiments", "lib", "libmdl_hip_exp.so")

_PROTOS = r"""
import ctypes
from matdeeplearn_amd import _lib
_vp, _i64, _i32, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
L = _lib.lib()
for name, (res, args) in {
    "mdl_cgconv_gate_row_bytes": (_sz, [_i32, _i32, _i32]),
    "mdl_cgconv_wsplit_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "mdl_cgconv_pack_weights_split": (_i32, [_vp] * 4 + [_i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "mdl_cgconv_fwd_p": (_i32, [_vp] * 10 + [_i64, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mdl_cgconv_bwd_p": (_i32, [_vp] * 13 + [_i64, _i64, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "mdl_cgconv_fwd_save": (_i32, [_vp] * 9 + [_i64, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mdl_cgconv_bwd_saved": (_i32, [_vp] * 10 + [_i64, _i64, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "mdl_mlp2": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
}.items():
    fn = getattr(L, name)
    fn.restype, fn.argtypes = res, args
"""


def _run(code, **env):
    if not os.path.exists(EXP_LIB):
        pytest.skip("experiments library not built (python experiments/build.py)")
    # (the package reads no environment: the library path and the fp32 by-source buffer are explicit calls; the MDL_CG_* variables
    # that remain are read by the EXPERIMENTS build of the C library, `#if MDL_EXPERIMENTS`)
    pre = "from matdeeplearn_amd import _lib as _l, ops as _o\n_l.use_library(%r)\n" % EXP_LIB
    if env.pop("MDL_CG_RSRC16", "1") == "0":
        pre += "_o.configure(rsrc16=False)\n"
    r = subprocess.run([sys.executable, "-c", pre + code], env={**os.environ, **env}, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "EXP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("C", [64, 32])
def test_cooperative_kernels_match_oracle(C):
    """The weight-stationary kernels (cgconv_cb.inc: forward MDL_CG_CB=1, backward edge pass MDL_CG_CB_BWD=1; fp32 by-source
    sums) against the oracle, on graphs large enough for several workgroups, multi-tile groups, partial last tiles and sources
    outside the 64-node window."""
    _run("import torch; import tests.test_gpu_kernels as t\n"
         "t._cgconv_case(700, %d, 50, torch.bfloat16, True, seed=21, empty_frac=0.05)\n"
         "t._cgconv_case(90, %d, 50, torch.bfloat16, True, seed=22, aggr='add')\nprint('EXP_OK')" % (C, C),
         MDL_CG_CB="1", MDL_CG_CB_BWD="1", MDL_CG_RSRC16="0")


def test_first_edge_per_lane_backward_matches_oracle():
    """cgconv_ep.inc (MDL_CG_EP=1: phases one after the other, fp32 by-source sums) against the oracle."""
    _run("import torch; import tests.test_gpu_kernels as t\n"
         "t._cgconv_case(700, 64, 50, torch.bfloat16, True, seed=21, empty_frac=0.05)\n"
         "t._cgconv_case(2500, 64, 50, torch.bfloat16, True, seed=23, empty_frac=0.3)\n"
         "t._cgconv_case(90, 64, 50, torch.bfloat16, True, seed=22, aggr='add')\n"
         "t._cgconv_case(1500, 64, 50, torch.bfloat16, True, seed=25, empty_frac=0.0, window=400)\nprint('EXP_OK')",
         MDL_CG_EP="1", MDL_CG_RSRC16="0")


_SAVED_GATE = _PROTOS + r"""
import torch
from matdeeplearn_amd import ops
from tests.test_gpu_kernels import close
d = torch.device("cuda:0")
P, st = _lib.ptr, _lib.stream
for C in (64, 32):
    G, dt = 50, _lib.MDL_BF16
    g = torch.Generator().manual_seed(17)
    n = 9001
    tgt = torch.arange(n).repeat_interleave(9)
    src = (tgt + torch.randint(-90, 91, (tgt.numel(),), generator=g)).clamp_(0, n - 1)       # ~1/3 outside a 64-node window
    E = tgt.numel()
    csr = ops.build_csr(torch.stack([src, tgt]).to(d), n, assume_sorted=True)
    x = torch.randn(n, C, generator=g).to(d).to(torch.bfloat16)
    ea = torch.rand(E, G, generator=g).to(d).to(torch.bfloat16)
    gout = torch.randn(n, C, generator=g).to(d).to(torch.bfloat16)
    k = 3.0 / (2 * C + G) ** 0.5
    wf, ws = (torch.randn(C, 2 * C + G, generator=g) * k).to(d), (torch.randn(C, 2 * C + G, generator=g) * k).to(d)
    bf, bs = (torch.randn(C, generator=g) * 0.1).to(d), (torch.randn(C, generator=g) * 0.1).to(d)
    wpack = torch.empty(L.mdl_cgconv_wpack_bytes(C, G, dt), dtype=torch.uint8, device=d)
    bpack = torch.empty(2 * C, dtype=torch.float32, device=d)
    _lib.check(L.mdl_cgconv_pack_weights(P(wf), P(bf), P(ws), P(bs), C, G, P(wpack), P(bpack), dt, st()), "pack")
    rb = L.mdl_cgconv_gate_row_bytes(C, G, dt)
    assert rb == 4 * C and L.mdl_cgconv_gate_row_bytes(100, G, dt) == 0 and L.mdl_cgconv_gate_row_bytes(C, G, _lib.MDL_F32) == 0
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    gate = torch.full((E, rb // 2), float("nan"), dtype=torch.bfloat16, device=d)
    _lib.check(L.mdl_cgconv_fwd(P(x), P(ea), P(csr.rowptr), P(csr.src), P(csr.tgt), None, P(wpack), P(bpack), P(o1),
                                n, E, C, G, 1, dt, st()), "fwd")
    _lib.check(L.mdl_cgconv_fwd_save(P(x), P(ea), P(csr.rowptr), P(csr.src), P(csr.tgt), P(wpack), P(bpack), P(o2), P(gate),
                                     n, E, C, G, 1, dt, st()), "fwd_save")
    close(o2, o1, 1e-2, 1e-2)
    assert torch.isfinite(gate.float()).all()                        # every (edge, channel) pair was written exactly once
    res = []
    for saved in (False, True):
        r_tgt = torch.empty(n, 2 * C, device=d, dtype=torch.bfloat16)
        r_src = torch.zeros(n, 2 * C, device=d)
        dwe = torch.zeros(2 * C, 64, device=d)
        db = torch.zeros(2 * C, device=d)
        if saved:
            _lib.check(L.mdl_cgconv_bwd_saved(P(ea), P(csr.rowptr), P(csr.src), P(csr.tgt), P(gate), P(gout), P(r_tgt), P(r_src),
                                              P(dwe), P(db), n, E, C, G, 1, dt, None, 0, st()), "bwd_saved")
        else:
            _lib.check(L.mdl_cgconv_bwd_ex(_lib.cg_args(dtype=dt, aggr=1, N=n, E=E, C=C, G=G, x=x, edge_attr=ea, rowptr=csr.rowptr, src=csr.src,
                                                        tgt=csr.tgt, wpack=wpack, bpack=bpack, grad_out=gout, r_tgt=r_tgt, r_src=r_src,
                                                        r_src_dtype=_lib.MDL_F32, dwe=dwe, db=db), st()), "bwd")
        res.append((r_tgt, r_src, dwe, db))
    for a, b in zip(res[1], res[0]):
        close(a, b, 2e-2, 1e-2)
print("EXP_OK")
"""


def test_saved_gate_pair_matches_recompute_through_the_c_abi():
    """mdl_cgconv_fwd_save + mdl_cgconv_bwd_saved against mdl_cgconv_fwd + mdl_cgconv_bwd on the same operands: same output,
    r_tgt / r_src / dwe / db within bf16 rounding of the stored factors; graphs wider than the source window, a ragged tail."""
    _run(_SAVED_GATE)


_WSPLIT = _PROTOS + r"""
import torch
from matdeeplearn_amd import ops
from tests.test_gpu_kernels import close
d = torch.device("cuda:0")
P, st = _lib.ptr, _lib.stream
for C in (64, 32):
    G, dt = 50, _lib.MDL_BF16
    g = torch.Generator().manual_seed(19)
    n = 6001
    tgt = torch.arange(n).repeat_interleave(9)
    src = (tgt + torch.randint(-90, 91, (tgt.numel(),), generator=g)).clamp_(0, n - 1)
    E = tgt.numel()
    csr = ops.build_csr(torch.stack([src, tgt]).to(d), n, assume_sorted=True)
    x = torch.randn(n, C, generator=g).to(d).to(torch.bfloat16)
    ea = torch.rand(E, G, generator=g).to(d).to(torch.bfloat16)
    gout = torch.randn(n, C, generator=g).to(d).to(torch.bfloat16)
    k = 3.0 / (2 * C + G) ** 0.5
    wf, ws = (torch.randn(C, 2 * C + G, generator=g) * k).to(d), (torch.randn(C, 2 * C + G, generator=g) * k).to(d)
    bf, bs = (torch.randn(C, generator=g) * 0.1).to(d), (torch.randn(C, generator=g) * 0.1).to(d)
    wpack = torch.empty(L.mdl_cgconv_wpack_bytes(C, G, dt), dtype=torch.uint8, device=d)
    bpack = torch.empty(2 * C, dtype=torch.float32, device=d)
    _lib.check(L.mdl_cgconv_pack_weights(P(wf), P(bf), P(ws), P(bs), C, G, P(wpack), P(bpack), dt, st()), "pack")
    wpe = torch.empty(L.mdl_cgconv_wsplit_bytes(C, G, dt, 0), dtype=torch.uint8, device=d)
    wproj = torch.empty((2, 2 * C, C), dtype=torch.bfloat16, device=d)
    bpack2 = torch.empty(2 * C, dtype=torch.float32, device=d)
    _lib.check(L.mdl_cgconv_pack_weights_split(P(wf), P(bf), P(ws), P(bs), C, G, P(wpe), P(wproj), P(bpack2), dt, st()), "pack split")
    pt = torch.empty((n, 2 * C), dtype=torch.bfloat16, device=d)
    ps = torch.empty_like(pt)
    _lib.check(L.mdl_linear_act(P(x), P(wproj[0]), None, P(pt), n, C, 2 * C, 0, dt, st()), "P_t")
    _lib.check(L.mdl_linear_act(P(x), P(wproj[1]), None, P(ps), n, C, 2 * C, 0, dt, st()), "P_s")
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    _lib.check(L.mdl_cgconv_fwd(P(x), P(ea), P(csr.rowptr), P(csr.src), P(csr.tgt), None, P(wpack), P(bpack), P(o1), n, E, C, G, 1, dt, st()), "fwd")
    _lib.check(L.mdl_cgconv_fwd_p(P(x), P(pt), P(ps), P(ea), P(csr.rowptr), P(csr.src), P(csr.tgt), P(wpe), P(bpack2), P(o2), n, E, C, G, 1,
                                  dt, st()), "fwd_p")
    close(o2, o1, 3e-2, 3e-2)                       # (the projections are rounded to bf16 once more than the fused product)
    res = []
    for split in (False, True):
        r_tgt = torch.empty(n, 2 * C, device=d, dtype=torch.bfloat16)
        r_src = torch.zeros(n, 2 * C, device=d)
        dwe, db = torch.zeros(2 * C, 64, device=d), torch.zeros(2 * C, device=d)
        if split:
            _lib.check(L.mdl_cgconv_bwd_p(P(pt), P(ps), P(ea), P(csr.rowptr), P(csr.src), P(csr.tgt), P(wpe), P(bpack2), P(gout), P(r_tgt),
                                          P(r_src), P(dwe), P(db), n, E, C, G, 1, dt, None, 0, st()), "bwd_p")
        else:
            _lib.check(L.mdl_cgconv_bwd_ex(_lib.cg_args(dtype=dt, aggr=1, N=n, E=E, C=C, G=G, x=x, edge_attr=ea, rowptr=csr.rowptr, src=csr.src,
                                                        tgt=csr.tgt, wpack=wpack, bpack=bpack, grad_out=gout, r_tgt=r_tgt, r_src=r_src,
                                                        r_src_dtype=_lib.MDL_F32, dwe=dwe, db=db), st()), "bwd")
        res.append((r_tgt, r_src, dwe, db))
    for a, b in zip(res[1], res[0]):
        close(a, b, 3e-2, 3e-2)
print("EXP_OK")
"""


def test_w_split_pair_matches_the_fused_pair_through_the_c_abi():
    """mdl_cgconv_fwd_p / mdl_cgconv_bwd_p (per-node projections from two dense launches, per edge only the K = 64 edge-feature
    product) against mdl_cgconv_fwd / mdl_cgconv_bwd, which the product's oracle tests pin: output, r_tgt, r_src, dwe, db."""
    _run(_WSPLIT)


_MLP2 = _PROTOS + r"""
import torch
from tests.test_gpu_kernels import close
d = torch.device("cuda:0")
P, st = _lib.ptr, _lib.stream
g = torch.Generator().manual_seed(2)
dt = _lib.MDL_BF16
for (N, K, M1, M2, a1, a2) in ((5000 + 37, 50, 150, 150, 2, 0), (3000, 64, 100, 100, 1, 1), (130, 10, 32, 8, 2, 2)):
    x = torch.randn(N, K, generator=g).to(d).to(torch.bfloat16)
    w1 = (torch.randn(M1, K, generator=g) * 0.2).to(d).to(torch.bfloat16)
    w2 = (torch.randn(M2, M1, generator=g) * 0.1).to(d).to(torch.bfloat16)
    b1 = (torch.randn(M1, generator=g) * 0.1).to(d).to(torch.bfloat16)
    b2 = (torch.randn(M2, generator=g) * 0.1).to(d).to(torch.bfloat16)
    h0, y0 = torch.empty(N, M1, device=d, dtype=torch.bfloat16), torch.empty(N, M2, device=d, dtype=torch.bfloat16)
    _lib.check(L.mdl_linear_act(P(x), P(w1), P(b1), P(h0), N, K, M1, a1, dt, st()), "l1")
    _lib.check(L.mdl_linear_act(P(h0), P(w2), P(b2), P(y0), N, M1, M2, a2, dt, st()), "l2")
    h1 = torch.full_like(h0, float("nan"))
    y1 = torch.full_like(y0, float("nan"))
    _lib.check(L.mdl_mlp2(P(x), P(w1), P(b1), a1, P(w2), P(b2), a2, P(h1), P(y1), N, K, M1, M2, dt, st()), "mlp2")
    assert torch.equal(h0, h1), (N, K, M1, M2)
    close(y1, y0, 1e-2, 1e-3)
print("EXP_OK")
"""


def test_two_layer_dense_kernel_matches_two_launches():
    """mdl_mlp2 (Linear -> activation -> Linear [-> activation] with the intermediate tile in LDS) against two mdl_linear_act
    launches: hidden rows bit-identical, outputs equal to the rounding of the bf16 hidden rows they are computed from."""
    _run(_MLP2)
