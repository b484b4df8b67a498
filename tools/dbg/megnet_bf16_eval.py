"""MEGNet_demo: bf16 vs fp32 forward of the HIP path at the same weights, train and eval mode, before / after a few steps."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from matdeeplearn_amd import models, ops
from matdeeplearn_amd.process import synthetic_bulk
from matdeeplearn_amd.training import make_optimizer
dev = torch.device("cuda:0")
ds = synthetic_bulk(3000, seed=0).to(dev)
kw = dict(dim1=100, dim2=100, dim3=100, pre_fc_count=1, gc_count=3, gc_fc_count=1, post_fc_count=3, pool="global_mean_pool",
          pool_order="early", batch_norm="True", batch_track_stats="True", act="relu", dropout_rate=0.0)
torch.manual_seed(0)
m16 = models.MEGNet(ds, compute_dtype="bf16", **kw).to(dev)
m32 = models.MEGNet(ds, compute_dtype="fp32", **kw).to(dev)
rng = np.random.default_rng(0)
def cmp(tag):
    m32.load_state_dict(m16.state_dict())
    ids = rng.choice(len(ds), size=512, replace=False)
    for mode in ("train", "eval"):
        outs = []
        for m, dt in ((m16, torch.bfloat16), (m32, torch.float32)):
            mm = copy.deepcopy(m); getattr(mm, mode)()
            with torch.no_grad():
                outs.append(mm(ds.collate(ids, edge_dtype=dt, x_dtype=dt)).float())
        a, b = outs
        print(tag, mode, "pred scale %.3f  max|diff| %.4f  mean|diff| %.4f" % (float(b.abs().max()), float((a - b).abs().max()), float((a - b).abs().mean())))
cmp("init")
opt = make_optimizer(m16.parameters(), "AdamW", lr=0.002)
m16.train()
for s in range(25):
    ids = rng.choice(len(ds), size=1024, replace=False)
    b = ds.collate(ids, edge_dtype=torch.bfloat16, x_dtype=torch.bfloat16)
    opt.zero_grad(set_to_none=True)
    loss = ops.loss("l1_loss", m16(b), b.y); loss.backward(); opt.step()
print("train loss after 25 steps", float(loss))
cmp("after 25 steps")
for k, v in m16.state_dict().items():
    if "running_var" in k and ("conv_list.2" in k or "conv_list.0.edge" in k): print(k, float(v.min()), float(v.max()))
# sensitivity of the fp32 path to bf16-level perturbations of its inputs only (one rounding at the input)
ids = rng.choice(len(ds), size=512, replace=False)
mm = copy.deepcopy(m32); mm.train()
with torch.no_grad():
    b = ds.collate(ids, edge_dtype=torch.float32, x_dtype=torch.float32)
    o0 = mm(b).float()
    b.edge_attr = b.edge_attr.to(torch.bfloat16).float()
    o1 = mm(b).float()
    # weights rounded to bf16 as well
    for p in mm.parameters(): p.copy_(p.to(torch.bfloat16).float())
    o2 = mm(b).float()
print("fp32 path, edge features rounded to bf16: max|diff| %.4f; + weights rounded: %.4f (scale %.3f)" % (float((o1 - o0).abs().max()), float((o2 - o0).abs().max()), float(o0.abs().max())))
