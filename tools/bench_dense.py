"""Event timing of the streaming dense kernels (TN GEMM, fused Linear) on the SchNet / MEGNet edge shapes.
usage: python tools/bench_dense.py [rows ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matdeeplearn_amd import _lib
import _ab; _ab.apply()      # (tools/_ab.py: MDL_HIP_LIB / MDL_OPS of the A/B scripts -> explicit calls)
L = _lib.lib(); P = _lib.ptr; st = _lib.stream
d = torch.device("cuda:0")


def t(name, fn, bytes_, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("%-34s %8.1f us   %6.2f TB/s" % (name, us, bytes_ / us / 1e6))


rows_list = [int(a) for a in sys.argv[1:]] or [1_498_398, 374_600]
for rows in rows_list:
    for (M, K) in ((150, 150), (150, 50), (100, 100), (64, 114)):
        a = torch.randn(rows, M, device=d).to(torch.bfloat16); b = torch.randn(rows, K, device=d).to(torch.bfloat16)
        c = torch.zeros(M, K, device=d); cs = torch.zeros(M, device=d)
        t("gemm_tn %dx%d rows %d" % (M, K, rows), lambda: L.mdl_gemm_tn_colsum(P(a), a.stride(0), M, P(b), b.stride(0), K, P(c), P(cs), rows, _lib.MDL_BF16, st()),
          rows * (M + K) * 2)
    for (K, M, act) in ((50, 150, 2), (150, 150, 0), (100, 100, 1), (114, 64, 1)):
        x = torch.randn(rows, K, device=d).to(torch.bfloat16); w = torch.randn(M, K, device=d).to(torch.bfloat16)
        bb = torch.randn(M, device=d).to(torch.bfloat16); o = torch.empty(rows, M, device=d, dtype=torch.bfloat16)
        t("linear %d->%d act %d rows %d" % (K, M, act, rows), lambda: L.mdl_linear_act(P(x), P(w), P(bb), P(o), rows, K, M, act, _lib.MDL_BF16, st()),
          rows * (M + K) * 2)
    for (K, M, xact) in ((100, 100, 1), (150, 150, 2)):
        g = torch.randn(rows, K, device=d).to(torch.bfloat16); y = torch.randn(rows, K, device=d).to(torch.bfloat16)
        w = torch.randn(M, K, device=d).to(torch.bfloat16); o = torch.empty(rows, M, device=d, dtype=torch.bfloat16)
        t("linear_in %d->%d xact %d rows %d" % (K, M, xact, rows),
          lambda: L.mdl_linear_act_in(P(g), P(y), xact, P(w), None, P(o), rows, K, M, 0, _lib.MDL_BF16, st()), rows * (M + 2 * K) * 2)
    for (K, M, act, xout) in ((100, 100, 1, 0), (100, 100, 0, 1), (150, 150, 0, 2), (150, 150, 2, 0)):
        g = torch.randn(rows, M, device=d).to(torch.bfloat16); y = torch.randn(rows, M, device=d).to(torch.bfloat16)
        x = torch.randn(rows, K, device=d).to(torch.bfloat16); w = torch.randn(M, K, device=d).to(torch.bfloat16)
        dx = torch.empty(rows, K, device=d, dtype=torch.bfloat16); dw = torch.zeros(M, K, device=d); db = torch.zeros(M, device=d)
        t("dense_bwd %d<-%d act %d xout %d rows %d" % (K, M, act, xout, rows),
          lambda: L.mdl_dense_bwd(P(g), M, M, P(y) if act else None, M, act, P(x), K, K, P(w), P(dx), K, xout, None, P(dw), P(db), rows,
                                  _lib.MDL_BF16, st()), rows * ((2 if act else 1) * M + 2 * K) * 2)
# the _ex forms (partial sums through the scratch buffer + reduce launch) against the forms above
scr = torch.empty(L.mdl_tn_scratch_bytes(), dtype=torch.uint8, device=d)
for rows in rows_list:
    for (M, K) in ((150, 150), (100, 100), (64, 114)):
        a = torch.randn(rows, M, device=d).to(torch.bfloat16); b = torch.randn(rows, K, device=d).to(torch.bfloat16)
        c = torch.zeros(M, K, device=d); cs = torch.zeros(M, device=d)
        t("gemm_tn_ex %dx%d rows %d" % (M, K, rows), lambda: L.mdl_gemm_tn_ex(P(a), a.stride(0), M, None, 0, 0, P(b), b.stride(0), K, P(c), P(cs), P(scr), rows, _lib.MDL_BF16, st()),
          rows * (M + K) * 2)
    for (K, M, act, xout) in ((100, 100, 1, 0), (100, 100, 0, 1), (150, 150, 0, 2)):
        g = torch.randn(rows, M, device=d).to(torch.bfloat16); y = torch.randn(rows, M, device=d).to(torch.bfloat16)
        x = torch.randn(rows, K, device=d).to(torch.bfloat16); w = torch.randn(M, K, device=d).to(torch.bfloat16)
        dx = torch.empty(rows, K, device=d, dtype=torch.bfloat16); dw = torch.zeros(M, K, device=d); db = torch.zeros(M, device=d)
        t("dense_bwd_ex %d<-%d act %d xout %d rows %d" % (K, M, act, xout, rows),
          lambda: L.mdl_dense_bwd_ex(P(g), M, M, P(y) if act else None, M, act, P(x), K, K, P(w), P(dx), K, xout, None, P(dw), P(db), P(scr), rows,
                                     _lib.MDL_BF16, st()), rows * ((2 if act else 1) * M + 2 * K) * 2)
