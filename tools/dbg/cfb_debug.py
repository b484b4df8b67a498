"""debug: mdl_cfconv_bwd_w per-block errors against fp64"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from matdeeplearn_amd import _lib, ops
from test_gpu_kernels import rand_graph
d = torch.device("cuda:0")
L, P, st = _lib.lib(), _lib.ptr, _lib.stream
n, F, G = int(os.environ.get("N", 700)), 150, 50
g = torch.Generator().manual_seed(11)
ei = rand_graph(n, 23, sort=True, empty_frac=0.2, max_in=40)
E = ei.shape[1]
csr = ops.build_csr(ei.to(d), n, assume_sorted=True)
rbf = torch.rand(E, G, generator=g).to(torch.bfloat16)
cut = torch.rand(E, generator=g)
h = torch.randn(n, F, generator=g).to(torch.bfloat16)
gout = torch.randn(n, F, generator=g).to(torch.bfloat16)
w1, b1 = torch.randn(F, G, generator=g) * 0.3, torch.randn(F, generator=g) * 0.2
w2, b2 = torch.randn(F, F, generator=g) * 0.1, torch.randn(F, generator=g) * 0.2
wpack = torch.empty(L.mdl_cfconv_wpack_bytes(), dtype=torch.uint8, device=d)
dv = [t.to(d).contiguous() for t in (rbf, cut, h, gout, w1, b1, w2, b2)]
_lib.check(L.mdl_cfconv_pack_weights(P(dv[4]), P(dv[5]), P(dv[6]), P(dv[7]), F, G, P(wpack), st()), "pack")
scratch = None if os.environ.get('NOSCRATCH') else torch.empty(L.mdl_cfconv_bwd_w_scratch_bytes(), dtype=torch.uint8, device='cuda:0')
outs = [torch.zeros(s_, dtype=torch.float32, device=d) for s_ in ((F, G), (F,), (F, F), (F,))]
flags = _lib.MDL_DETERMINISTIC if os.environ.get("DET") else 0
_lib.check(L.mdl_cfconv_bwd_w(P(dv[0]), P(dv[1]), P(dv[2]), P(dv[3]), P(csr.rowptr), P(csr.src), P(csr.tgt), P(wpack), P(outs[0]), P(outs[1]),
                              P(outs[2]), P(outs[3]), P(scratch), n, E, F, G, _lib.MDL_BF16 | flags, st()), "cfconv_bwd_w")
torch.cuda.synchronize()
bfr = lambda t: t.to(torch.bfloat16).double()
s_cpu, t_cpu = csr.src.cpu().long(), csr.tgt.cpu().long()
a1 = bfr(torch.nn.functional.softplus(bfr(rbf) @ bfr(w1).t() + bfr(b1)) - np.log(2.0))
dw = bfr(gout.double()[t_cpu] * h.double()[s_cpu] * cut.double().view(-1, 1))
da = bfr((dw @ bfr(w2)) * (1.0 - torch.exp(-(a1 + np.log(2.0)))))
refs = {"dw2": dw.t() @ a1, "db2": dw.sum(0), "dw1": da.t() @ bfr(rbf), "db1": da.sum(0)}
got = {"dw1": outs[0], "db1": outs[1], "dw2": outs[2], "db2": outs[3]}
print("E", E, "tiles", (E + 63) // 64)
for k in ("dw2", "db2", "dw1", "db1"):
    a, b = got[k].double().cpu(), refs[k]
    print(k, "scale %.3e  max err %.3e" % (float(b.abs().max()), float((a - b).abs().max())))
    if a.dim() == 2:
        R, C = (a.shape[0] + 31) // 32, (a.shape[1] + 31) // 32
        for r in range(R):
            print("   ", " ".join("%9.2e" % float((a - b)[32 * r:32 * r + 32, 32 * c:32 * c + 32].abs().max()) for c in range(C)),
                  " | ratio", " ".join("%6.2f" % float((a[32 * r:32 * r + 32, 32 * c:32 * c + 32].abs().sum() / (b[32 * r:32 * r + 32, 32 * c:32 * c + 32].abs().sum() + 1e-30))) for c in range(C)))
    else:
        print("    got", a[:6].tolist(), "ref", b[:6].tolist())
