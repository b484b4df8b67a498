"""Segment reductions at the sizes of a batch-100 MEGNet step (E = 32 k edge rows -> N = 2.5 k nodes -> B = 100 graphs, C = 100 / 64):
which of them take 25-40 us, and why.  usage: python tools/bench_segment_small.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from matdeeplearn_amd import ops
from matdeeplearn_amd.process import synthetic_bulk
import _ab; _ab.apply()
d = torch.device("cuda:0")
ds = synthetic_bulk(512, seed=0).to(d)
b = ds.collate(np.arange(100), edge_dtype=torch.bfloat16, x_dtype=torch.bfloat16)
N, E = b.num_nodes, b.num_edges
ei = b.edge_index
print("N", N, "E", E)


def t(name, fn, iters=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-58s %7.1f us" % (name, e0.elapsed_time(e1) * 1e3 / iters))


for C in (64, 100):
    e = torch.randn(E, C, device=d).to(torch.bfloat16)
    x = torch.randn(N, C, device=d).to(torch.bfloat16)
    t("C=%d scatter_mean(e [E], row = source) -> N" % C, lambda: ops.scatter(e, ei[0], 0, N, "mean"))
    t("C=%d scatter_mean(e [E], col = target, sorted) -> N" % C, lambda: ops.scatter(e, ei[1], 0, N, "mean", assume_sorted=True))
    t("C=%d scatter_sum(e [E], row) -> N" % C, lambda: ops.scatter(e, ei[0], 0, N, "sum"))
    t("C=%d scatter_mean(x [N], batch) -> B" % C, lambda: ops.scatter(x, b.batch, 0, 100, "mean", assume_sorted=True))
    bsrc = b.batch[ei[0]]
    t("C=%d scatter_mean(e [E], batch[row]) -> B (320 rows per segment)" % C, lambda: ops.scatter(e, bsrc, 0, 100, "mean", assume_sorted=True))

# the same graph-level reductions on the PADDED static batch of a captured step (process.StaticBatch: n_cap rows, pooling index from the loader)
from matdeeplearn_amd.process import StaticBatch, static_capacity
n_cap, e_cap = static_capacity(ds, 100, slack=3.5)
sb = StaticBatch(ds, 100, n_cap, e_cap, x_dtype=torch.bfloat16, edge_dtype=torch.bfloat16)
sb.load(np.arange(100)); sbb = sb.assemble(); torch.cuda.synchronize()
print("static: n_cap", n_cap, "e_cap", e_cap, "pool segments", sbb.pool_index.N, "rowptr tail", sbb.pool_index.rowptr[-3:].tolist())
for C in (64, 100):
    xp = torch.randn(n_cap, C, device=d).to(torch.bfloat16)
    t("C=%d static: scatter_mean(x [n_cap], batch) -> B + 1 (pool index)" % C, lambda: ops.scatter(xp, sb.batch_idx, 0, 101, "mean", seg_index=sbb.pool_index))
    t("C=%d static: scatter_sum(x [n_cap], batch) -> B + 1 (pool index)" % C, lambda: ops.scatter(xp, sb.batch_idx, 0, 101, "sum", seg_index=sbb.pool_index))

# ... and inside a captured graph (what the replayed step runs), 20 reductions per replay
xp = torch.randn(n_cap, 100, device=d).to(torch.bfloat16)
outs = []
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        ops.scatter(xp, sb.batch_idx, 0, 101, "mean", seg_index=sbb.pool_index)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        outs.append(ops.scatter(xp, sb.batch_idx, 0, 101, "mean", seg_index=sbb.pool_index))
t("captured: 20 x scatter_mean(x [n_cap, 100], batch) -> B + 1 per replay", lambda: g.replay(), iters=20)
ids2 = np.random.default_rng(0).choice(512, size=100, replace=False)
sb.load(ids2); sb.assemble(); torch.cuda.synchronize()
print("after a random batch: rowptr head", sbb.pool_index.rowptr[:4].tolist(), "tail", sbb.pool_index.rowptr[-3:].tolist())
t("captured, random batch loaded", lambda: g.replay(), iters=20)
