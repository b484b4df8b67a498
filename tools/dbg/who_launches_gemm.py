#!/usr/bin/env python3
"""Which host call launches the per-layer fp32 library GEMM that remains in the bf16x3 / fp32 step?  torch.profiler with stacks."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from matdeeplearn_amd import models, ops
from matdeeplearn_amd.process import synthetic_bulk
from matdeeplearn_amd.training import FlatDataParallel, make_optimizer
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
ds = synthetic_bulk(9000, seed=0); ds.to(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
model = models.CGCNN(ds, compute_dtype=mode, dim1=64, dim2=64, pre_fc_count=1, gc_count=4, post_fc_count=3).to(dev).train()
dp = FlatDataParallel(model); opt = make_optimizer(model.parameters(), "AdamW", lr=0.002)
ids = np.arange(8192)
def step():
    b = ds.collate(ids, edge_dtype=torch.float32, x_dtype=torch.float32)
    dp.zero_grad()
    with ops.zero_arena(dev):
        loss = ops.loss("l1_loss", model(b), b.y); loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 150 and ("mm" in e.key or "linear" in e.key or "matmul" in e.key)]
for e in sorted(rows, key=lambda e: -e.device_time_total)[:12]:
    print("%-28s calls %3d  device %8.1f us  shapes %s" % (e.key[:28], e.count, e.device_time_total, str(e.input_shapes)[:120]))
for e in sorted(prof.key_averages(group_by_stack_n=6), key=lambda e: -e.device_time_total)[:40]:
    if ("mm" in e.key or "linear" in e.key) and e.device_time_total > 150:
        print("==", e.key, e.count, "%.1f us" % e.device_time_total)
        for fr in e.stack[:6]: print("     ", fr[:150])
