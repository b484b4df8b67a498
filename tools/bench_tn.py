"""Event timing of the tall-skinny TN GEMM and the node-level backward kernel on the bench shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matdeeplearn_amd import _lib
import _ab; _ab.apply()      # (tools/_ab.py: MDL_HIP_LIB / MDL_OPS of the A/B scripts -> explicit calls)
L = _lib.lib(); P = _lib.ptr; st = _lib.stream
d = torch.device("cuda:0")
N = 209768
def t(name, fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-22s %.1f us" % (name, e0.elapsed_time(e1) * 1e3 / iters))
for (M, K, rows) in ((64, 114, N), (64, 64, 8192)):
    a = torch.randn(rows, M, device=d).to(torch.bfloat16); b = torch.randn(rows, K, device=d).to(torch.bfloat16)
    c = torch.zeros(M, K, device=d)
    t("gemm_tn %dx%dx%d" % (M, K, rows), lambda: L.mdl_gemm_tn(P(a), a.stride(0), M, P(b), b.stride(0), K, P(c), rows, _lib.MDL_BF16, st()))
C = 64
x = torch.randn(N, C, device=d).to(torch.bfloat16); g = torch.randn(N, C, device=d).to(torch.bfloat16)
rt = torch.randn(N, 2 * C, device=d).to(torch.bfloat16); rs = torch.randn(N, 2 * C, device=d)
wn = torch.randn(C, 4 * C, device=d).to(torch.bfloat16); dx = torch.empty_like(x); dwn = torch.zeros(4 * C, C, device=d)
t("cgconv_bwd_node", lambda: L.mdl_cgconv_bwd_node_ex(_lib.cg_node_args(dtype=_lib.MDL_BF16, N=N, C=C, r_src_dtype=_lib.MDL_F32, x=x, grad_out=g,
                                                                        r_tgt=rt, r_src=rs, wn_t=wn, dx=dx, dwn=dwn), st()))
