"""Event timing of the NNConv (K7) kernels at the cfg5 MPNN shapes: Y = x W2r (mdl_linear_wide), the message forward / backward,
and the two library products of the backward (dx = dY W2r^T, dW2r = x^T dY).  usage: python tools/bench_nnconv.py [N E]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matdeeplearn_amd import _lib, ops
import _ab; _ab.apply()
L = _lib.lib(); P = _lib.ptr; st = _lib.stream
d = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 52000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 806000
C, D3 = 100, 100


def t(name, fn, bytes_, iters=10):
    for _ in range(25): fn()          # (clocks: a cold device runs the first milliseconds 20 % slow)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("%-40s %8.1f us   %6.2f TB/s" % (name, us, bytes_ / us / 1e6))


x = torch.randn(N, C, device=d).to(torch.bfloat16)
w = (torch.randn(C, C * D3, device=d) * 0.1).to(torch.bfloat16)
wt = w.t().contiguous()
Y = torch.empty(N, C * D3, device=d, dtype=torch.bfloat16)
t("linear_wide Y = x W2r [%d x %d]" % (N, C * D3), lambda: L.mdl_linear_wide(P(x), P(wt), P(Y), N, C, C * D3, _lib.MDL_BF16, st()), N * C * D3 * 2)
dY = torch.randn(N, C * D3, device=d).to(torch.bfloat16)
t("library dx = dY W2r^T", lambda: dY @ wt, N * C * D3 * 2)
t("library dW2r = x^T dY", lambda: x.t() @ dY, N * C * D3 * 2)
# message kernels on a random graph (E edges, targets sorted): m[e] = Y[src_e].view(C, D3) @ h[e]
tgt = torch.sort(torch.randint(0, N, (E,), device=d)).values
src = torch.randint(0, N, (E,), device=d)
ei = torch.stack([src, tgt])
csr = ops.build_csr(ei, N, assume_sorted=True)
h = torch.randn(E, D3, device=d).to(torch.bfloat16)
Yg = Y.detach().clone().requires_grad_(True)
hg = h.detach().clone().requires_grad_(True)
m = ops.nnconv_msg(Yg, hg, csr, C)
gm = torch.randn_like(m)
t("nnconv_msg fwd", lambda: ops.nnconv_msg(Y, h, csr, C), N * C * D3 * 2 + E * (D3 + C) * 2)
def fb():
    Yg.grad = None; hg.grad = None
    ops.nnconv_msg(Yg, hg, csr, C).backward(gm)
t("nnconv_msg fwd + bwd", fb, 3 * N * C * D3 * 2 + E * (2 * D3 + 2 * C) * 2)
