"""MEGNet_demo: fp32 vs bf16 compute mode trained on the same batches; where do the predictions differ afterwards?"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from matdeeplearn_amd import models, ops
from matdeeplearn_amd.process import synthetic_bulk
from matdeeplearn_amd.training import make_optimizer
DEV = "cuda:0"
ds = synthetic_bulk(512, seed=5)
z = np.add.reduceat(ds.z.astype(np.float64), ds.node_ptr[:-1]) / np.diff(ds.node_ptr)
ds.y = ((z - z.mean()) / z.std()).astype(np.float32).reshape(-1, 1)
ds = ds.to(DEV)
kw = dict(dim1=100, dim2=100, dim3=100, pre_fc_count=1, gc_count=4, gc_fc_count=1, post_fc_count=3)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(0)
batches = [rng.choice(448, size=64, replace=False) for _ in range(steps)]
held = np.arange(448, 512)
trained, curves = {}, {}
for cd, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
    torch.manual_seed(0)
    m = models.MEGNet(ds, compute_dtype=cd, **kw).to(DEV)
    opt = make_optimizer(m.parameters(), "AdamW", lr=0.0005)
    m.train(); losses = []
    for ids in batches:
        b = ds.collate(ids, edge_dtype=dt, x_dtype=dt)
        opt.zero_grad(set_to_none=True)
        with ops.zero_arena(torch.device(DEV)):
            loss = torch.nn.functional.l1_loss(m(b), b.y); loss.backward()
        opt.step(); losses.append(float(loss.detach()))
    trained[cd] = m; curves[cd] = np.array(losses)
print("loss fp32", curves["fp32"].round(3).tolist()); print("loss bf16", curves["bf16"].round(3).tolist())
def ev(weights_from, cd):
    dt = torch.bfloat16 if cd == "bf16" else torch.float32
    m = models.MEGNet(ds, compute_dtype=cd, **kw).to(DEV); m.load_state_dict(trained[weights_from].state_dict()); m.eval()
    with torch.no_grad():
        return m(ds.collate(held, edge_dtype=dt, x_dtype=dt)).float().cpu()
P = {(w, c): ev(w, c) for w in ("fp32", "bf16") for c in ("fp32", "bf16")}
for k, v in P.items(): print("weights %s eval %s: mean %.4f std %.4f absmax %.4f" % (k[0], k[1], v.mean(), v.std(), v.abs().max()))
d = lambda a, b: float((P[a] - P[b]).abs().max())
print("same weights (fp32-trained): bf16 eval vs fp32 eval", d(("fp32", "bf16"), ("fp32", "fp32")))
print("same weights (bf16-trained): bf16 eval vs fp32 eval", d(("bf16", "bf16"), ("bf16", "fp32")))
print("different training, both fp32 eval", d(("fp32", "fp32"), ("bf16", "fp32")))
# running stats
sa, sb = trained["fp32"].state_dict(), trained["bf16"].state_dict()
worst = sorted(((float((sa[k].float() - sb[k].float()).abs().max() / (sa[k].float().abs().max() + 1e-9)), k) for k in sa if sa[k].dtype.is_floating_point), reverse=True)[:8]
print("largest relative state differences", [(round(a, 3), k) for a, k in worst])
