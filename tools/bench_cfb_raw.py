#!/usr/bin/env python3
"""raw C-ABI timing of mdl_cfconv_bwd_w (and the transposed mdl_cfconv_fwd) on the SchNet bench batch; MDL_HIP_LIB selects a variant build"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from matdeeplearn_amd import ops, nn as mnn, _lib
import _ab; _ab.apply()
from matdeeplearn_amd.process import synthetic_mof
dev = torch.device("cuda:0")
ds = synthetic_mof(1200, seed=0).to(dev)
b = ds.collate(np.arange(1024), edge_dtype=torch.bfloat16, x_dtype=torch.bfloat16)
L, P, st = _lib.lib(), _lib.ptr, _lib.stream
E, N, F, G = b.num_edges, b.num_nodes, 150, 50
torch.manual_seed(0)
conv = mnn.InteractionBlock(100, 50, F, 8.0).to(dev)
h = (torch.randn(N, F, device=dev) * 0.5).to(torch.bfloat16)
g = (torch.randn(N, F, device=dev) * 0.5).to(torch.bfloat16)
cut = mnn.cosine_cutoff(b.edge_weight, 8.0).float().contiguous()
wpack = torch.empty(L.mdl_cfconv_wpack_bytes(), dtype=torch.uint8, device=dev)
m0, m2 = conv.mlp[0], conv.mlp[2]
_lib.check(L.mdl_cfconv_pack_weights(P(m0.weight), P(m0.bias), P(m2.weight), P(m2.bias), F, G, P(wpack), st()), "pack")
scratch = None if os.environ.get('NOSCRATCH') else torch.empty(L.mdl_cfconv_bwd_w_scratch_bytes(), dtype=torch.uint8, device='cuda:0')
outs = [torch.zeros(s_, dtype=torch.float32, device=dev) for s_ in ((F, G), (F,), (F, F), (F,))]
csr = b.csr
ts = []
for it in range(23):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    _lib.check(L.mdl_cfconv_bwd_w(P(b.edge_attr), P(cut), P(h), P(g), P(csr.rowptr), P(csr.src), P(csr.tgt), P(wpack), P(outs[0]), P(outs[1]),
                                  P(outs[2]), P(outs[3]), P(scratch), N, E, F, G, _lib.MDL_BF16, st()), "bwd_w")
    e.record()
    if it >= 3:
        ts.append((s, e))
torch.cuda.synchronize()
u = sorted(s.elapsed_time(e) * 1e3 for s, e in ts)
print("%s mdl_cfconv_bwd_w E=%d: median %.1f us  min %.1f us" % (os.environ.get("MDL_HIP_LIB", "product").split("/")[-1], E, u[len(u) // 2], u[0]))
