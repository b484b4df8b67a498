#!/usr/bin/env python3
"""K4b micro-benchmark: forward + backward of one InteractionBlock on the SchNet bench batch (cfg3: MOF-like, 1024 graphs) in its
three forms (unfused / fused forward with stored activations / recompute), then the raw C-ABI timings of mdl_cfconv_bwd_w and of
the transposed mdl_cfconv_fwd."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matdeeplearn_amd import ops, nn as mnn, _lib
from matdeeplearn_amd.process import synthetic_mof

ap = argparse.ArgumentParser()
ap.add_argument("--graphs", type=int, default=1024)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--F", type=int, default=150)
ap.add_argument("--modes", default="recompute,stored,unfused,recompute")
a = ap.parse_args()
dev = torch.device("cuda:0")
ds = synthetic_mof(1200, seed=0).to(dev)
b = ds.collate(np.arange(a.graphs), edge_dtype=torch.bfloat16, x_dtype=torch.bfloat16)
print("N=%d E=%d" % (b.num_nodes, b.num_edges))
torch.manual_seed(0)
conv = mnn.InteractionBlock(100, 50, a.F, 8.0).to(dev)
x = (torch.randn(b.num_nodes, 100, device=dev) * 0.5).to(torch.bfloat16)
gy = torch.randn(b.num_nodes, 100, device=dev).to(torch.bfloat16)
cut = mnn.cosine_cutoff(b.edge_weight, 8.0)
b.csr.transposed()
for mode in a.modes.split(","):
    ops.configure(cfconv_fused=mode != "unfused", cfconv_recompute=mode == "recompute")
    ops._CFCONV_RECOMPUTE_MIN_F = 0                   # (time the requested form whatever the default dispatch would pick at this width)
    ev = {k: [] for k in ("cfconv_fwd", "gmr_fwd", "cfconv_bwd_w", "cfconv_bwd_h")}
    t = []
    for it in range(a.iters + 3):
        ops.KERNEL_EVENTS = ev if it >= 3 else None
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        xr = x.clone().requires_grad_(True)
        conv.zero_grad(set_to_none=True)
        s.record()
        y = conv.conv(xr, b.edge_index, b.edge_weight, b.edge_attr, csr=b.csr, cut=cut)
        y.backward(gy[:, :y.shape[1]] if y.shape[1] <= gy.shape[1] else gy.repeat(1, 2)[:, :y.shape[1]])
        e.record()
        if it >= 3:
            t.append((s, e))
    ops.KERNEL_EVENTS = None
    torch.cuda.synchronize()
    tt = sorted(s.elapsed_time(e) * 1e3 for s, e in t)
    line = "%-10s CFConv fwd + bwd: median %.1f us  min %.1f us" % (mode, tt[len(tt) // 2], tt[0])
    for k, v in ev.items():
        if v:
            u = sorted(s.elapsed_time(e) * 1e3 for s, e in v)
            line += " | %s median %.1f" % (k, u[len(u) // 2])
    print(line)
