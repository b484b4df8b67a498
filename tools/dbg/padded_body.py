import os, sys, time, torch, numpy as np
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
m.train()
if os.environ.get("MODE") == "padded":
    ops.NO_INDEX_CACHE = True
    for k in range(12):
        gs.sb.load(ids[k]); gs._zero_grad(); gs._body()
else:
    for k in range(12):
        gs._eager(ids[k])
torch.cuda.synchronize()
