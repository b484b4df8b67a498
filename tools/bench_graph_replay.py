import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matdeeplearn_amd import models, ops
from matdeeplearn_amd.process import synthetic_bulk
from matdeeplearn_amd.training import GraphedStep, make_optimizer
import _ab; _ab.apply()      # (tools/_ab.py: MDL_HIP_LIB / MDL_OPS of the A/B scripts -> explicit calls)
dev = torch.device("cuda:0")
ds = synthetic_bulk(int(os.environ.get("GRAPHS", "20000")), seed=0).to(dev)
rng = np.random.default_rng(0)
for B in (100, 8192):
    torch.manual_seed(0)
    m = models.CGCNN(ds, dim1=64, dim2=64, gc_count=4, post_fc_count=3, compute_dtype="bf16").to(dev)
    o = make_optimizer(m.parameters(), "AdamW", lr=0.002, capturable=True)
    gs = GraphedStep(ds, m, o, B, compute_dtype=torch.bfloat16)
    ids = [rng.choice(len(ds), size=B, replace=False) for _ in range(40)]
    gs.step(ids[0])
    torch.cuda.synchronize()
    # (a) replay only, same batch
    t0 = time.perf_counter()
    for _ in range(30):
        gs.graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B=%d replay-only: host %.3f ms/step, total %.3f ms/step" % (B, (t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3))
    # (b) load + replay
    t0 = time.perf_counter()
    for k in range(30):
        gs.step(ids[1 + k])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B=%d load+replay: host %.3f ms/step, total %.3f ms/step" % (B, (t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3))
    # (c) eager body on the static buffers (no graph)
    ops.NO_INDEX_CACHE = True
    t0 = time.perf_counter()
    for k in range(30):
        gs.sb.load(ids[1 + k]); gs._zero_grad(); gs._body()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ops.NO_INDEX_CACHE = False
    print("B=%d eager padded body: host %.3f ms/step, total %.3f ms/step" % (B, (t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3))
