"""debug: host-side cost per call of mdl_dense_bwd vs mdl_dense_bwd_ex (+ scratch), and of ops._tn_scratch"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from matdeeplearn_amd import _lib, ops
L = _lib.lib(); P = _lib.ptr; st = _lib.stream
d = torch.device("cuda:0")
rows, M, K = 82000, 100, 100
g = torch.randn(rows, M, device=d).to(torch.bfloat16); x = torch.randn(rows, K, device=d).to(torch.bfloat16); w = torch.randn(M, K, device=d).to(torch.bfloat16)
dx = torch.empty(rows, K, device=d, dtype=torch.bfloat16); dw = torch.zeros(M, K, device=d); db = torch.zeros(M, device=d)
scr = torch.empty(L.mdl_tn_scratch_bytes(), dtype=torch.uint8, device=d)
def a(): L.mdl_dense_bwd(P(g), M, M, None, M, 0, P(x), K, K, P(w), P(dx), K, 0, None, P(dw), P(db), rows, _lib.MDL_BF16, st())
def b(): L.mdl_dense_bwd_ex(P(g), M, M, None, M, 0, P(x), K, K, P(w), P(dx), K, 0, None, P(dw), P(db), P(scr), rows, _lib.MDL_BF16, st())
def c(): ops._tn_scratch(d)
for name, fn in (("mdl_dense_bwd", a), ("mdl_dense_bwd_ex + scratch", b), ("ops._tn_scratch", c), ("mdl_dense_bwd", a), ("mdl_dense_bwd_ex + scratch", b)):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-28s host %.1f us/call   incl. device drain %.1f us/call" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
