"""mdl_linear_wide at equal output bytes (1 GB) and K = 100 for different output widths M: is the 20-KB row stride of NNConv's Y what
holds the kernel at 2.3 TB/s?  (tools; round 6)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matdeeplearn_amd import _lib
import _ab; _ab.apply()
L = _lib.lib(); P = _lib.ptr; st = _lib.stream
d = torch.device("cuda:0")
K = 100
TOTAL = 52000 * 10000
for M in (192, 384, 768, 1536, 3072, 9984, 10000):
    N = TOTAL // M
    x = torch.randn(N, K, device=d).to(torch.bfloat16)
    wt = (torch.randn(M, K, device=d) * 0.1).to(torch.bfloat16)
    Y = torch.empty(N, M, device=d, dtype=torch.bfloat16)
    fn = lambda: L.mdl_linear_wide(P(x), P(wt), P(Y), N, K, M, _lib.MDL_BF16, st())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    print("M = %5d  N = %8d   %8.1f us   %5.2f TB/s (output + x bytes)" % (M, N, us, (N * M * 2 + N * K * 2) / us / 1e6))
