import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from matdeeplearn_amd import models, ops
from matdeeplearn_amd.process import synthetic_bulk
from matdeeplearn_amd.training import GraphedStep, make_optimizer
dev = torch.device("cuda:0")
ds = synthetic_bulk(400, seed=13).to(dev)
B = 48
rng = np.random.default_rng(2)
batches = [rng.choice(len(ds), size=B, replace=False) for _ in range(4)]
kw = dict(dim1=32, dim2=32, dim3=32, gc_count=2, gc_fc_count=1, post_fc_count=2)
cd, dt = "fp32", torch.float32
def run(mode):
    torch.manual_seed(4)
    m = models.MEGNet(ds, compute_dtype=cd, **kw).to(dev)
    if mode == "graph":
        o = make_optimizer(m.parameters(), "AdamW", lr=0.002, capturable=True)
        gs = GraphedStep(ds, m, o, B, compute_dtype=dt)
        out = []
        for ids in batches:
            gs.step(ids); out.append(float(gs.loss_value))
        return out
    o = make_optimizer(m.parameters(), "AdamW", lr=0.002, capturable=(mode == "eager_cap"))
    m.train(); out = []
    for ids in batches:
        batch = ds.collate(ids, edge_dtype=dt, x_dtype=dt)
        o.zero_grad(set_to_none=True)
        with ops.zero_arena(dev):
            loss = torch.nn.functional.l1_loss(m(batch), batch.y); loss.backward()
        o.step(); out.append(float(loss.detach()))
    return out
for mode in ("eager", "eager", "eager_cap", "graph", "graph"):
    print(mode, ["%.6f" % v for v in run(mode)])
