import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_kernels import rand_graph
from matdeeplearn_amd import ops, _lib
from oracle import ops as oops
d = torch.device("cuda:0")
n, C, G = 200, 64, 50
dtype = torch.bfloat16
g = torch.Generator().manual_seed(n + C + G)
ei = rand_graph(n, n + C + G, sort=True, empty_frac=0.1)
E = ei.shape[1]
rnd = lambda *s: torch.randn(*s, generator=g)
x = rnd(n, C).to(dtype).float()
ea = torch.rand(E, G, generator=g).to(dtype).float()
k = 1.0 / (2 * C + G) ** 0.5
wf, ws = (rnd(C, 2 * C + G) * k * 3).to(dtype).float(), (rnd(C, 2 * C + G) * k * 3).to(dtype).float()
bf, bs = rnd(C) * 0.1, rnd(C) * 0.1
gout = rnd(n, C).to(dtype).float()
ref = oops.cgconv(x, ei, ea, wf, bf, ws, bs, "mean")
csr = ops.build_csr(ei.to(d), n, assume_sorted=True)
xd = x.to(d).to(dtype)
args = [t.to(d) for t in (wf, bf, ws, bs)]
with torch.no_grad():
    o_plain = ops.cgconv(xd, None, ea.to(d).to(dtype), *args, "mean", csr=csr)
xg = xd.clone().requires_grad_(True)
o_save = ops.cgconv(xg, None, ea.to(d).to(dtype), *args, "mean", csr=csr)
torch.cuda.synchronize()
o_save_before = o_save.detach().clone()
print("plain vs ref", float((o_plain.float().cpu() - ref).abs().max()))
print("save  vs ref", float((o_save_before.float().cpu() - ref).abs().max()))
dd = (o_save_before.float().cpu() - ref).abs()
rows = torch.nonzero(dd.max(1).values > 0.3).flatten().tolist()
print("bad rows", rows[:40], "of", n)
deg = (csr.rowptr[1:] - csr.rowptr[:-1]).cpu()
print("deg of bad rows", [int(deg[r]) for r in rows[:40]])
print("rowptr of bad rows", [int(csr.rowptr[r]) for r in rows[:40]])
cols = torch.nonzero(dd.max(0).values > 0.3).flatten().tolist()
print("bad cols", cols)
(o_save.float() * gout.to(d)).sum().backward()
torch.cuda.synchronize()
print("save after bwd vs before", float((o_save.detach().float() - o_save_before.float()).abs().max()))
# ---- which messages are wrong?
import torch.nn.functional as F
row, col = ei[0], ei[1]
z = torch.cat([x[col], x[row], ea], 1)
m = torch.sigmoid(F.linear(z, wf, bf)) * F.softplus(F.linear(z, ws, bs))       # [E, C]
rp = csr.rowptr.cpu().tolist()
W = min((E + 63) // 64, n)
bounds = []
import bisect
for wi in range(W + 1):
    b = E * wi // W
    bounds.append(bisect.bisect_left(rp, b) if 0 < wi < W else (0 if wi == 0 else n))
print("range starts", bounds[:40])
diff = (o_save_before.float().cpu() - ref)
for r in rows[:6]:
    dg = rp[r + 1] - rp[r]
    dv = diff[r] * dg                       # = sum of (wrong - right) messages
    es = list(range(rp[r], rp[r + 1]))
    # is dv == -m_e for some e (missing), or == m_e' - 0 for a foreign edge?
    best = min(((float((dv + m[e]).abs().max()), e) for e in es))
    bestf = min(((float((dv - m[e]).abs().max()), e) for e in range(max(0, rp[r] - 40), min(E, rp[r + 1] + 40))))
    print("row", r, "deg", dg, "edges", es[0], es[-1], "| missing-edge fit", best, "| extra-edge fit", bestf, "| |dv|max", float(dv.abs().max()))
