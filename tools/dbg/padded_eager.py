import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from matdeeplearn_amd import models, ops
from matdeeplearn_amd.process import synthetic_bulk, StaticBatch, static_capacity
dev = torch.device("cuda:0")
ds = synthetic_bulk(400, seed=13).to(dev)
B = 48
rng = np.random.default_rng(2)
ids = rng.choice(len(ds), size=B, replace=False)
name = sys.argv[1]; cd = sys.argv[2]; dt = torch.bfloat16 if cd == "bf16" else torch.float32
kw = dict(SchNet=dict(dim1=32, dim2=32, dim3=48, gc_count=2, post_fc_count=2), CGCNN=dict(dim1=32, dim2=32, gc_count=2, post_fc_count=2),
          MEGNet=dict(dim1=32, dim2=32, dim3=32, gc_count=2, gc_fc_count=1, post_fc_count=2),
          GCN=dict(dim1=32, dim2=32, gc_count=2, post_fc_count=2), MPNN=dict(dim1=32, dim2=32, dim3=24, gc_count=2, post_fc_count=2))[name]
torch.manual_seed(4)
m = getattr(models, name)(ds, compute_dtype=cd, **kw).to(dev); m.train()
n_cap, e_cap = static_capacity(ds, B)
sb = StaticBatch(ds, B, n_cap, e_cap, x_dtype=dt, edge_dtype=dt)
caps = {}
def hook(nm):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        caps.setdefault(nm, []).append(o.detach().float().clone() if torch.is_tensor(o) else None)
    return f
for nm, mod in m.named_modules():
    if nm and nm.count(".") <= 2: mod.register_forward_hook(hook(nm))
def run(batch, tr):
    m.zero_grad(set_to_none=True)
    ops.NO_INDEX_CACHE = True
    with ops.true_rows(tr), ops.zero_arena(dev):
        out = m(batch)
        loss = torch.nn.functional.mse_loss(out[:B], batch.y[:B]); loss.backward()
    return out.detach().float()[:B].clone(), [p.grad.float().clone() for p in m.parameters() if p.requires_grad]
be = ds.collate(ids, edge_dtype=dt, x_dtype=dt)
oe, ge = run(be, None); ce = {k: v for k, v in caps.items()}; caps.clear()
sb.load(ids); bp = sb.assemble()
op, gp = run(bp, bp.true_rows); cp = dict(caps)
N, E = be.num_nodes, be.num_edges
print("pred max diff", float((oe - op).abs().max()), "scale", float(oe.abs().max()))
names = [k for k, p in m.named_parameters() if p.requires_grad]
gmax = max(float(g.abs().max()) for g in ge)
worst = sorted(((float((a - b).abs().max()) / gmax, k) for k, a, b in zip(names, gp, ge)), reverse=True)[:5]
print("worst grads", [(round(w, 5), k) for w, k in worst])
for k in ce:
    for a, b in zip(ce[k], cp[k]):
        if a is None: continue
        r = a.shape[0]
        d = float((a - b[:r]).abs().max()); s = float(a.abs().max())
        if d > 0: print("module", k, tuple(a.shape), tuple(b.shape), "diff %.4g scale %.4g" % (d, s))
