#!/usr/bin/env python3
"""The training step at the reference's batch size (config.yml:136: 100 graphs) as a HIP-graph replay, for rocprofv3
--kernel-trace --stats: which kernels the 0.6 ms go to.  Prints ms/step (device-synchronised at the ends only)."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matdeeplearn_amd import models
from matdeeplearn_amd.process import synthetic_bulk
from matdeeplearn_amd.training import GraphedStep, make_optimizer
import _ab; _ab.apply()
if os.environ.get("MDL_DENSE_MIN_ROWS"):      # (tool-level A/B of ops._DENSE_MIN_ROWS)
    from matdeeplearn_amd import ops as _ops
    _ops._DENSE_MIN_ROWS = int(os.environ["MDL_DENSE_MIN_ROWS"])      # (tools/_ab.py: MDL_HIP_LIB / MDL_OPS of the A/B scripts -> explicit calls)

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=100)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--graphs", type=int, default=4096)
ap.add_argument("--dim", type=int, default=64, help="dim1 (100 = the reference's default width, config.yml:123)")
ap.add_argument("--dim2", type=int, default=0)
ap.add_argument("--model", default="CGCNN", help="CGCNN | SchNet | MEGNet | GCN | MPNN with the reference's *_demo widths (dim 100)")
a = ap.parse_args()
dev = torch.device("cuda:0")
ds = synthetic_bulk(a.graphs, seed=0).to(dev)
rng = np.random.default_rng(0)
torch.manual_seed(0)
if a.model == "CGCNN":
    m = models.CGCNN(ds, dim1=a.dim, dim2=a.dim2 or a.dim, gc_count=4, post_fc_count=3, compute_dtype="bf16").to(dev)
else:
    kw = {"SchNet": dict(dim1=100, dim2=100, dim3=150, cutoff=8), "MEGNet": dict(dim1=100, dim2=100, dim3=100, gc_fc_count=1),
          "GCN": dict(dim1=100, dim2=150), "MPNN": dict(dim1=100, dim2=100, dim3=100)}[a.model]
    m = getattr(models, a.model)(ds, pre_fc_count=1, gc_count=4, post_fc_count=3, compute_dtype="bf16", **kw).to(dev)
o = make_optimizer(m.parameters(), "AdamW", lr=0.002, capturable=True)
gs = GraphedStep(ds, m, o, a.batch, compute_dtype=torch.bfloat16)
ids = [rng.choice(len(ds), size=a.batch, replace=False) for _ in range(64)]
for k in range(20):
    gs.step(ids[k])
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(a.steps):
    gs.step(ids[k % 64])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("B=%d: %.4f ms/step over %d replays (eager fallbacks %d)" % (a.batch, dt / a.steps * 1e3, a.steps, gs.eager_steps))
