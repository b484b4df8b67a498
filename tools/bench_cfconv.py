#!/usr/bin/env python3
"""K4 micro-benchmark: the fused CFConv forward against the three-pass sequence on the SchNet bench batch (cfg3: MOF-like, 1024 graphs)."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matdeeplearn_amd import ops, nn as mnn
from matdeeplearn_amd.process import synthetic_mof

ap = argparse.ArgumentParser()
ap.add_argument("--graphs", type=int, default=1024)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--F", type=int, default=150)
ap.add_argument("--nograd", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
ds = synthetic_mof(1200, seed=0).to(dev)
b = ds.collate(np.arange(a.graphs), edge_dtype=torch.bfloat16, x_dtype=torch.bfloat16)
print("N=%d E=%d" % (b.num_nodes, b.num_edges))
torch.manual_seed(0)
conv = mnn.InteractionBlock(100, 50, a.F, 8.0).to(dev)
x = (torch.randn(b.num_nodes, 100, device=dev) * 0.5).to(torch.bfloat16)
cut = mnn.cosine_cutoff(b.edge_weight, 8.0)
for fused in (True, False, True, False):
    ops._CFCONV_FUSED = fused
    keys = ["cfconv_fwd", "gmr_fwd"]
    ev = {k: [] for k in keys}
    t = []
    for it in range(a.iters + 3):
        ops.KERNEL_EVENTS = ev if it >= 3 else None
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        if a.nograd:
            with torch.no_grad():
                y = conv.conv(x, b.edge_index, b.edge_weight, b.edge_attr, csr=b.csr, cut=cut)
        else:
            xr = x.clone().requires_grad_(True)
            y = conv.conv(xr, b.edge_index, b.edge_weight, b.edge_attr, csr=b.csr, cut=cut)
        e.record()
        if it >= 3:
            t.append((s, e))
    ops.KERNEL_EVENTS = None
    torch.cuda.synchronize()
    tt = sorted(s.elapsed_time(e) * 1e3 for s, e in t)
    line = "fused=%d  CFConv forward (lin1 + filter + aggregate + lin2): median %.1f us  min %.1f us" % (fused, tt[len(tt) // 2], tt[0])
    for k, v in ev.items():
        if v:
            u = sorted(s.elapsed_time(e) * 1e3 for s, e in v)
            line += " | %s median %.1f" % (k, u[len(u) // 2])
    print(line)

# raw C-ABI timings: which of the two activation tensors are written
from matdeeplearn_amd import _lib
import _ab; _ab.apply()      # (tools/_ab.py: MDL_HIP_LIB / MDL_OPS of the A/B scripts -> explicit calls)
L, P, st = _lib.lib(), _lib.ptr, _lib.stream
E, N, F = b.num_edges, b.num_nodes, a.F
h = (torch.randn(N, F, device=dev) * 0.5).to(torch.bfloat16)
wpack = torch.empty(L.mdl_cfconv_wpack_bytes(), dtype=torch.uint8, device=dev)
m0, m2 = conv.mlp[0], conv.mlp[2]
_lib.check(L.mdl_cfconv_pack_weights(P(m0.weight), P(m0.bias), P(m2.weight), P(m2.bias), F, 50, P(wpack), st()), "pack")
out = torch.empty(N, F, dtype=torch.bfloat16, device=dev)
a1 = torch.empty(E, F, dtype=torch.bfloat16, device=dev)
w = torch.empty(E, F, dtype=torch.bfloat16, device=dev)
cutf = cut.float().contiguous()
for name, pa, pw in (("none", None, None), ("a1", a1, None), ("w", None, w), ("both", a1, w), ("none", None, None)):
    ts = []
    for it in range(a.iters + 3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        _lib.check(L.mdl_cfconv_fwd(P(b.edge_attr), P(cutf), P(h), P(b.csr.rowptr), P(b.csr.src), P(b.csr.tgt), P(wpack), P(out), P(pa), P(pw),
                                    N, E, F, 50, _lib.MDL_BF16, st()), "cfconv")
        e.record()
        if it >= 3:
            ts.append((s, e))
    torch.cuda.synchronize()
    u = sorted(s.elapsed_time(e) * 1e3 for s, e in ts)
    print("mdl_cfconv_fwd stores=%-5s median %.1f us  min %.1f us" % (name, u[len(u) // 2], u[0]))
    if hasattr(L, "mdl_debug_read_cf"):
        import ctypes
        buf = (ctypes.c_longlong * 16)()
        L.mdl_debug_read_cf(buf)
        v = list(buf)
        n = max(v[8], 1)
        names = ["top: issue gathers + prefetch, commit rbf", "one-hot + GEMM1 + ssp", "vmcnt(0)", "a1 staging + stores", "GEMM2 blocks", "group epilogue", "group top (rowptr)"]
        print("   per tile (wave 0, %d tiles, %d cycles/tile):" % (n, sum(v[:7]) / n), {names[k]: round(v[k] / n) for k in range(7)})
