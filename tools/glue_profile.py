#!/usr/bin/env python3
"""Which torch-level operations (not HIP-library kernels) a model's training step still launches, with input shapes and device time:
torch.profiler over a few steps of a bench workload.  usage: python tools/glue_profile.py megnet|mpnn|schnet|gcn|cgcnn [batch [dim1 dim2]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from matdeeplearn_amd import models, ops, process
from matdeeplearn_amd.training import make_optimizer

name = sys.argv[1] if len(sys.argv) > 1 else "megnet"
cls_name, gen_name, n_graphs, B, mkw, _ = bench.WORKLOADS[name]
if len(sys.argv) > 2:
    B = int(sys.argv[2])
dev = torch.device("cuda:0")
ds = getattr(process, gen_name)(min(n_graphs, 4 * B), seed=0).to(dev)
torch.manual_seed(0)
kw = dict(mkw or dict(dim1=64, dim2=64, gc_count=4, post_fc_count=3))
if len(sys.argv) > 4:
    kw.update(dim1=int(sys.argv[3]), dim2=int(sys.argv[4]))
m = getattr(models, cls_name)(ds, compute_dtype="bf16", **kw).to(dev)
opt = make_optimizer(m.parameters(), "AdamW", lr=0.002)
rng = np.random.default_rng(0)


def step():
    ids = rng.choice(len(ds), size=B, replace=False)
    batch = ds.collate(ids, edge_dtype=torch.bfloat16, x_dtype=torch.bfloat16)
    opt.zero_grad(set_to_none=True)
    with ops.zero_arena(dev):
        loss = ops.loss("l1_loss", m(batch), batch.y)
        ops.backward(loss)
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = getattr(e, "self_cuda_time_total", 0)
    if t > 0:
        rows.append((t / 3.0, e.count / 3.0, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
print("%-9s %-7s %-42s %s" % ("us/step", "n/step", "op", "input shapes"))
for t, n, k, sh in rows[:int(os.environ.get("GLUE_ROWS", "45"))]:
    print("%9.1f %7.1f %-42s %s" % (t, n, k[:42], sh))
