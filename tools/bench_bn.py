"""Event timing of the BatchNorm kernels on the bench shapes (N nodes x 64 channels, bf16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matdeeplearn_amd import _lib
import _ab; _ab.apply()      # (tools/_ab.py: MDL_HIP_LIB / MDL_OPS of the A/B scripts -> explicit calls)
L = _lib.lib(); P = _lib.ptr; st = _lib.stream
d = torch.device("cuda:0")
N, C = 209768, 64
x = torch.randn(N, C, device=d).to(torch.bfloat16)
dy = torch.randn(N, C, device=d).to(torch.bfloat16)
y = torch.empty_like(x)
dx = torch.empty_like(x)
R = L.mdl_bn_sums_rows() if hasattr(L, "mdl_bn_sums_rows") else 2
sums = torch.zeros(R, C, device=d)
save = torch.zeros(2, C, device=d)
gw = torch.ones(C, device=d); gb = torch.zeros(C, device=d)
def t(name, fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-14s %.1f us" % (name, e0.elapsed_time(e1) * 1e3 / iters))
t("bn_stats", lambda: L.mdl_bn_stats_n(P(x), P(sums), N, C, None, _lib.MDL_BF16, st()))
t("bn_apply", lambda: L.mdl_bn_apply_n(P(x), P(sums), P(gw), P(gb), P(save), None, None, P(y), N, C, 1e-5, 0.1, None, _lib.MDL_BF16, st()))
t("bn_bwd_stats", lambda: L.mdl_bn_bwd_stats_n(P(dy), P(x), P(save), P(sums), N, C, None, _lib.MDL_BF16, st()))
t("bn_bwd_apply", lambda: L.mdl_bn_bwd_apply_n(P(dy), P(x), P(save), P(sums), P(gw), P(dx), N, C, None, _lib.MDL_BF16, st()))
t("fill 27MB", lambda: y.zero_())
t("copy 27MB", lambda: y.copy_(x))
