"""K3 / K2 launch durations (HIP events) on the PADDED static batch of the HIP-graph path, run eagerly: what the conv kernels
cost inside a replayed step (MDL_CG_BALANCE=0/1, MDL_CG_EP=0/2 in the environment for A/B runs)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from matdeeplearn_amd import models, ops
from matdeeplearn_amd.process import synthetic_bulk
from matdeeplearn_amd.training import GraphedStep, make_optimizer
dev = torch.device("cuda:0")
ds = synthetic_bulk(20000, seed=0).to(dev)
rng = np.random.default_rng(0)
B = 8192
torch.manual_seed(0)
m = models.CGCNN(ds, dim1=64, dim2=64, gc_count=4, post_fc_count=3, compute_dtype="bf16").to(dev)
o = make_optimizer(m.parameters(), "AdamW", lr=0.002, capturable=True)
gs = GraphedStep(ds, m, o, B, compute_dtype=torch.bfloat16)
ids = [rng.choice(len(ds), size=B, replace=False) for _ in range(12)]
ops.NO_INDEX_CACHE = True
ev = {"fwd": [], "bwd": [], "bwd_node": []}
for k in range(12):
    ops.KERNEL_EVENTS = ev if k >= 2 else None
    gs.sb.load(ids[k]); gs._zero_grad(); gs._body()
ops.KERNEL_EVENTS = None
torch.cuda.synchronize()
print("capacity", gs.sb.n_cap, gs.sb.e_cap, "true", gs.sb.true_nodes, gs.sb.true_edges)
for k, v in ev.items():
    t = sorted(s.elapsed_time(e) * 1e3 for s, e in v)
    print("%s: median %.1f us  min %.1f  max %.1f  (%d launches)" % (k, t[len(t) // 2], t[0], t[-1], len(t)))
