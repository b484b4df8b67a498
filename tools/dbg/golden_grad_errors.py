#!/usr/bin/env python3
"""What do the model-level golden gradient checks actually deliver in fp32?  For every case of tests/golden/wrappers.npz and
megnet.npz: per parameter tensor the error against the reference-generated gradient, relative to the tensor's own scale (for
tensors that carry signal: scale > 1e-3 of the model's largest gradient) and relative to the model's largest gradient (all
tensors).  The test bounds (tests/test_gpu_model.py) are these maxima x 10."""
import json, os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from matdeeplearn_amd import models
import test_gpu_model as T

G = os.path.join(ROOT, "tests", "golden")
d = torch.device("cuda:0")
ns = types.SimpleNamespace
worst_rel, worst_abs = (0, None), (0, None)


def account(case, model, grads):
    global worst_rel, worst_abs
    gmax = max(float(np.abs(g).max()) for g in grads.values() if g.size)
    for k, p in model.named_parameters():
        g = grads[k]
        if g.size == 0:
            continue
        g = torch.from_numpy(g)
        s = float(g.abs().max()) + 1e-9
        err = float((p.grad.cpu() - g).abs().max())
        if s > 1e-3 * gmax and err / s > worst_rel[0]:
            worst_rel = (err / s, (case, k, err, s, gmax))
        if err / gmax > worst_abs[0]:
            worst_abs = (err / gmax, (case, k, err, s, gmax))


z = np.load(os.path.join(G, "wrappers.npz"))
meta_all = json.loads(bytes(z["meta"]).decode())
for case, meta in meta_all.items():
    cls = case.split("/")[0]
    b = ns(x=torch.from_numpy(z["x"]).to(d), edge_index=torch.from_numpy(z["edge_index"]).to(d),
           edge_attr=torch.from_numpy(z["edge_attr"]).to(d), edge_weight=torch.from_numpy(z["edge_weight"]).to(d),
           batch=torch.from_numpy(z["batch"]).to(d), u=torch.zeros(3, 3, device=d), num_graphs=3)
    y = torch.from_numpy(z["y"]).to(d)
    torch.manual_seed(4321)
    model = getattr(models, cls)(T.DS(), dim1=16, dim2=12, dim3=8, gc_count=2, **meta["kw"]).to(d).train()
    torch.nn.functional.l1_loss(model(b), y).backward()
    account(case, model, {k: z["%s/grad/%s" % (case, k)] for k, _ in model.named_parameters()})
print("wrappers.npz  worst err / tensor scale (signal tensors): %.3e %s" % worst_rel)
print("wrappers.npz  worst err / model gmax (all tensors):      %.3e %s" % worst_abs)
w1, w2 = worst_rel, worst_abs
worst_rel, worst_abs = (0, None), (0, None)
z = np.load(os.path.join(G, "megnet.npz"))
for tag, kw in [("bn", dict(batch_norm="True")), ("nobn", dict(batch_norm="False")), ("max", dict(batch_norm="False", pool="global_max_pool")),
                ("late", dict(batch_norm="True", pool_order="late"))]:
    B = int(z["batch"].max()) + 1
    b = ns(x=torch.from_numpy(z["x"]).to(d), edge_index=torch.from_numpy(z["edge_index"]).to(d), edge_attr=torch.from_numpy(z["edge_attr"]).to(d),
           u=torch.from_numpy(z["u"]).to(d), batch=torch.from_numpy(z["batch"]).to(d), num_graphs=B)
    y = torch.from_numpy(z["y"]).to(d)
    model = models.MEGNet(T.DS(), dim1=32, dim2=24, dim3=16, pre_fc_count=1, gc_count=2, gc_fc_count=1, post_fc_count=2, **kw)
    sd = {k[len(tag) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + "/sd/")}
    for k in sd:
        if k.endswith("running_mean"): sd[k] = torch.zeros_like(sd[k])
        elif k.endswith("running_var"): sd[k] = torch.ones_like(sd[k])
        elif k.endswith("num_batches_tracked"): sd[k] = torch.zeros_like(sd[k])
    model.load_state_dict(sd)
    model.to(d).train()
    torch.nn.functional.l1_loss(model(b), y).backward()
    account("megnet/" + tag, model, {k: z["%s/grad/%s" % (tag, k)] for k, _ in model.named_parameters()})
print("megnet.npz    worst err / tensor scale (signal tensors): %.3e %s" % worst_rel)
print("megnet.npz    worst err / model gmax (all tensors):      %.3e %s" % worst_abs)
