"""cProfile of the HOST side of an eager training step at a small batch (where the step is host-bound): which Python functions the
~1.4 ms per step go to.  usage: python tools/host_profile_small.py [batch]"""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from matdeeplearn_amd import models, ops
from matdeeplearn_amd.process import synthetic_bulk
from matdeeplearn_amd.training import make_optimizer
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ds = synthetic_bulk(4096, seed=0).to(dev)
torch.manual_seed(0)
model = models.CGCNN(ds, dim1=64, dim2=64, pre_fc_count=1, gc_count=4, post_fc_count=3, compute_dtype="bf16").to(dev)
opt = make_optimizer(model.parameters(), "AdamW", lr=0.002)
rng = np.random.default_rng(0)
ids = [rng.choice(4096, size=B, replace=False) for _ in range(64)]


def step(k):
    batch = ds.collate(ids[k % 64], edge_dtype=torch.bfloat16, x_dtype=torch.bfloat16)
    opt.zero_grad(set_to_none=True)
    with ops.zero_arena(dev):
        loss = ops.loss("l1_loss", model(batch), batch.y)
        ops.backward(loss)
    opt.step()


for k in range(30):
    step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(200):
    step(k)
torch.cuda.synchronize()
print("eager step at batch %d: %.3f ms" % (B, (time.perf_counter() - t0) / 200 * 1e3))
pr = cProfile.Profile()
pr.enable()
for k in range(100):
    step(k)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue()[:9000])
