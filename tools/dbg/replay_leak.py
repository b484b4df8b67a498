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
name = sys.argv[1] if len(sys.argv) > 1 else "MEGNet"
kw = dict(SchNet=dict(dim1=32, dim2=32, dim3=48, gc_count=2, post_fc_count=2),
          MEGNet=dict(dim1=32, dim2=32, dim3=32, gc_count=2, gc_fc_count=1, post_fc_count=2),
          GCN=dict(dim1=32, dim2=32, gc_count=2, post_fc_count=2))[name]
cd, dt = "fp32", torch.float32
torch.manual_seed(4)
m_g = getattr(models, name)(ds, compute_dtype=cd, **kw).to(dev)
o_g = make_optimizer(m_g.parameters(), "AdamW", lr=0.002, capturable=True)
gs = GraphedStep(ds, m_g, o_g, B, compute_dtype=dt)
names = [k for k, p in m_g.named_parameters() if p.requires_grad]
for step, ids in enumerate(batches):
    m_e = copy.deepcopy(m_g); m_e.train()          # same weights as the graph model BEFORE its step
    batch = ds.collate(ids, edge_dtype=dt, x_dtype=dt)
    with ops.zero_arena(dev):
        loss = torch.nn.functional.l1_loss(m_e(batch), batch.y)
        loss.backward()
    ge = [p.grad.float() for p in m_e.parameters() if p.requires_grad]
    gs.step(ids)
    gmax = max(float(g.abs().max()) for g in ge)
    worst = sorted(((float((a.float() - b).abs().max()) / gmax, k) for k, a, b in zip(names, gs.static_grads, ge)), reverse=True)[:4]
    print("step", step, "N,E", gs.sb.true_nodes, gs.sb.true_edges, "cap", gs.sb.n_cap, gs.sb.e_cap, "loss e/g %.6f %.6f" % (float(loss), float(gs.loss_value)), "worst grad err/gmax:", [(round(w, 6), k) for w, k in worst])
