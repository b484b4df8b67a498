#!/usr/bin/env python3
"""Does a training step leave CYCLIC garbage that holds device tensors?  (bench.py disables the cyclic collector inside its
timed region; tensors kept alive by reference cycles are then never freed and the caching allocator has to hipMalloc new
blocks — seen as `device_mallocs` 1-5 in 20 timed steps.)  Runs steps with the collector off, then collects with DEBUG_SAVEALL
and prints what the garbage consists of."""
import gc, os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from matdeeplearn_amd import models, ops
from matdeeplearn_amd.process import synthetic_bulk
from matdeeplearn_amd.training import FlatDataParallel, make_optimizer

dev = torch.device("cuda:0")
ds = synthetic_bulk(2048, seed=0)
ds.to(dev)
torch.manual_seed(0)
model = models.CGCNN(ds, compute_dtype="bf16", dim1=64, dim2=64, pre_fc_count=1, gc_count=4, post_fc_count=3).to(dev)
dp = FlatDataParallel(model)
opt = make_optimizer(model.parameters(), "AdamW", lr=0.002)
rng = np.random.default_rng(0)
pending = {}


def step(ids, nxt):
    ahead = pending.pop(ids.tobytes(), None)
    batch = ds.take_ahead(ahead) if ahead is not None else ds.collate(ids, edge_dtype=torch.bfloat16, x_dtype=torch.bfloat16)
    pending.clear()
    pending[nxt.tobytes()] = ds.collate_ahead(nxt, edge_dtype=torch.bfloat16, x_dtype=torch.bfloat16)
    dp.zero_grad()
    with ops.zero_arena(dev):
        out = model(batch)
        loss = ops.loss("l1_loss", out, batch.y)
        loss.backward()
    if dp.reduce_grads_async():
        dp.finish()
    opt.step()


ids = [rng.permutation(2048)[:512] for _ in range(12)]
for k in range(4):
    step(ids[k], ids[k + 1])
torch.cuda.synchronize()
gc.collect()
gc.disable()
a0 = torch.cuda.memory_allocated()
for k in range(4, 10):
    step(ids[k], ids[k + 1])
torch.cuda.synchronize()
a1 = torch.cuda.memory_allocated()
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
kinds = collections.Counter(type(o).__name__ for o in gc.garbage)
tens = [o for o in gc.garbage if torch.is_tensor(o)]
print("allocated before / after 6 steps without the collector: %.1f / %.1f MB" % (a0 / 2**20, a1 / 2**20))
print("unreachable objects found:", n, "| tensors among them:", len(tens), "| device bytes: %.1f MB" % (sum(t.numel() * t.element_size() for t in tens if t.is_cuda) / 2**20))
print(kinds.most_common(25))
for o in gc.garbage:
    if type(o).__name__ in ("Batch", "EdgeCSR", "function", "cell") :
        print(type(o).__name__, getattr(o, "__qualname__", ""), [type(r).__name__ for r in gc.get_referents(o)][:8])
        if type(o).__name__ == "function":
            print("   closure:", [type(c.cell_contents).__name__ for c in (o.__closure__ or ())])
gc.set_debug(0)
gc.garbage.clear()
