"""Compare the cooperative column-block CGConv kernels against the per-wave kernels on a small random graph."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from matdeeplearn_amd import ops, nn as mnn
torch.manual_seed(0)
N, C, G, deg = int(os.environ.get("N", 64)), int(os.environ.get("C", 64)), 50, int(os.environ.get("DEG", 12))
dev = "cuda"
rng = np.random.default_rng(0)
src, tgt = [], []
for n in range(N):
    nb = rng.choice(N, size=min(deg, N - 1), replace=False)
    for j in nb:
        src.append(int(j)); tgt.append(n)
    src.append(n); tgt.append(n)
ei = torch.tensor([src, tgt], dtype=torch.int64, device=dev)
E = ei.shape[1]
x = torch.randn(N, C, device=dev).to(torch.bfloat16)
ea = torch.rand(E, G, device=dev).to(torch.bfloat16)
conv = mnn.CGConv(C, G).to(dev)
csr = ops.build_csr(ei, N, assume_sorted=True)
MODE = os.environ.get("MODE", "all")
with torch.no_grad():
    for lin in (conv.lin_f, conv.lin_s):
        if MODE == "bias": lin.weight.zero_()
        if MODE == "tgt": lin.weight[:, C:].zero_()
        if MODE == "src": lin.weight[:, :C].zero_(); lin.weight[:, 2 * C:].zero_()
        if MODE == "e": lin.weight[:, :2 * C].zero_()
        if MODE in ("tgt", "src", "e"): lin.bias.zero_()
def run(cb, bwd=False):
    os.environ["MDL_CG_CB"] = str(cb)
    xx = x.clone().requires_grad_(True)
    out = ops.cgconv(xx, ei, ea, conv.lin_f.weight, conv.lin_f.bias, conv.lin_s.weight, conv.lin_s.bias, "mean", csr=csr)
    return out, xx
try:
    o0, _ = run(0); o1, _ = run(1)
except Exception as e:
    print("ops.cgconv signature?", e); raise
DBG = int(os.environ.get("DBG", 0))
if DBG:
    # kernel built with -DMDL_CB_DEBUG=1/2 aggregates the raw pre-activation f (1) or s (2) instead of the gated message
    lin = conv.lin_f if DBG == 1 else conv.lin_s
    z = torch.cat([x[ei[1]].float(), x[ei[0]].float(), ea.float()], dim=1)
    pre = (z @ lin.weight.float().t() + lin.bias.float()) * 1.4426950408889634 * 0.6931471805599453
    agg = torch.zeros(N, C, device=dev).index_add_(0, ei[1], pre)
    cnt = torch.zeros(N, device=dev).index_add_(0, ei[1], torch.ones(E, device=dev)).clamp(min=1)
    o0 = (x.float() + agg / cnt[:, None]).to(torch.bfloat16)
d = (o0.float() - o1.float()).abs()
print("N %d E %d  max diff %.4f  mean %.5f  ref absmax %.3f" % (N, E, d.max().item(), d.mean().item(), o0.float().abs().max().item()))
bad_rows = (d.max(dim=1).values > 0.05).nonzero().flatten().tolist()
bad_cols = (d.max(dim=0).values > 0.05).nonzero().flatten().tolist()
print("bad rows (%d):" % len(bad_rows), bad_rows[:64])
print("bad cols (%d):" % len(bad_cols), bad_cols[:64])
if bad_rows:
    r = bad_rows[0]
    print("row", r, "ref", o0[r, :8].float().tolist(), "\n      new", o1[r, :8].float().tolist(), "\n      x  ", x[r, :8].float().tolist())

o2, _ = run(1)
print("new vs new (determinism): max diff %.4f" % (o1.float() - o2.float()).abs().max().item())
idx = (d > 0.05).nonzero()[:12].tolist()
for r, c in idx:
    print("  (%d,%d) ref %.4f new %.4f  x %.4f  indeg %d" % (r, c, o0[r, c].float(), o1[r, c].float(), x[r, c].float(), int((ei[1] == r).sum())))
print("rowptr[:12]", csr.rowptr[:12].tolist())
rm = d.max(dim=1).values
print("per-row max diff (first 40 rows):", [round(float(v), 3) for v in rm[:40]])
