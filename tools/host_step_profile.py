"""Where does the HOST spend its time per training step once the loop has run for a while?  (cProfile over bench's step)"""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from matdeeplearn_amd import models, ops
from matdeeplearn_amd.process import synthetic_bulk, DeviceLoader, split_data
from matdeeplearn_amd.training import FlatDataParallel, make_optimizer
import _ab; _ab.apply()      # (tools/_ab.py: MDL_HIP_LIB / MDL_OPS of the A/B scripts -> explicit calls)
from bench import batch_stream
dev = torch.device("cuda:0")
ds = synthetic_bulk(int(os.environ.get("GRAPHS", "46744")), seed=0).to(dev)
tr, va, _ = split_data(len(ds), 0.8, 0.05, 0.15, seed=42)
B = 8192
stream = batch_stream(DeviceLoader(ds, tr, B, shuffle=True, seed=42), B)
torch.manual_seed(42)
model = models.CGCNN(ds, dim1=64, dim2=64, pre_fc_count=1, gc_count=4, post_fc_count=3, compute_dtype="bf16").to(dev)
dp = FlatDataParallel(model)
opt = make_optimizer(model.parameters(), "AdamW", lr=0.002)
PH = [0.0] * 6
def step(ids):
    c = time.perf_counter
    t0 = c()
    batch = ds.collate(ids, edge_dtype=torch.bfloat16, x_dtype=torch.bfloat16)
    t1 = c()
    dp.zero_grad()
    with ops.zero_arena(dev):
        out = model(batch)
        t2 = c()
        loss = torch.nn.functional.l1_loss(out, batch.y)
        t3 = c()
        loss.backward()
        t4 = c()
    opt.step()
    t5 = c()
    del batch, out, loss
    t6 = c()
    for k, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6))):
        PH[k] += b - a
model.train()
def run(n, ev=None):
    ops.KERNEL_EVENTS = ev
    t0 = time.perf_counter()
    for k in range(n):
        step(next(stream))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    ops.KERNEL_EVENTS = None
    return (t1 - t0) / n * 1e3, (time.perf_counter() - t0) / n * 1e3
for rep in range(8):
    ev = {"fwd": [], "bwd": []}
    a, b = run(20, ev)
    kf = sum(s.elapsed_time(e) for s, e in ev["fwd"]) / len(ev["fwd"]) * 1e3
    kb = sum(s.elapsed_time(e) for s, e in ev["bwd"]) / len(ev["bwd"]) * 1e3
    # device-side span of the chunk: first fwd start -> last bwd end
    span = ev["fwd"][0][0].elapsed_time(ev["bwd"][-1][1]) / 20
    print("chunk %d: host enqueue %.2f ms/step, wall %.2f ms/step | K2 %.0f us, K3 %.0f us | device span %.2f ms/step" % (rep, a, b, kf, kb, span),
          "| collate %.2f fwd %.2f loss %.2f bwd %.2f opt %.2f free %.2f" % tuple(v / 20 * 1e3 for v in PH),
          "| mallocs", torch.cuda.memory_stats(dev).get("num_device_alloc", 0))
    PH[:] = [0.0] * 6
sys.exit(0)
pr = cProfile.Profile()
pr.enable()
run(60)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
