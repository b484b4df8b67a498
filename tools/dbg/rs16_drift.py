"""Loss curves of CGCNN 64x4 (bf16) trained with bf16 vs fp32 by-source sums and a second fp32-sums run (control):
what tests/test_gpu_workloads.py::test_cfg2_bf16_by_source_sums_train_like_fp32_sums bounds."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from matdeeplearn_amd import models, ops
from matdeeplearn_amd.process import synthetic_bulk
from matdeeplearn_amd.training import make_optimizer
from tests.test_gpu_workloads import _composition_targets
DEV = "cuda:0"
ds = _composition_targets(synthetic_bulk(1024, seed=6)).to(DEV)
kw = dict(dim1=64, dim2=64, pre_fc_count=1, gc_count=4, post_fc_count=3)
rng = np.random.default_rng(1)
batches = [rng.choice(896, size=128, replace=False) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30)]
curves = {}
for tag, flag, det in (("bf16_sums", True, False), ("fp32_sums", False, False), ("fp32_again", False, False), ("fp32_det", False, True), ("bf16_det", True, True)):
    ops._RSRC16 = flag
    ops.set_deterministic(det)
    torch.manual_seed(0)
    m = models.CGCNN(ds, compute_dtype="bf16", **kw).to(DEV)
    opt = make_optimizer(m.parameters(), "AdamW", lr=0.002)
    m.train()
    losses = []
    for ids in batches:
        b = ds.collate(ids, edge_dtype=torch.bfloat16, x_dtype=torch.bfloat16)
        opt.zero_grad(set_to_none=True)
        with ops.zero_arena(torch.device(DEV)):
            loss = torch.nn.functional.l1_loss(m(b), b.y)
            loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    curves[tag] = np.array(losses)
    print(tag, np.round(curves[tag], 3).tolist())
ops.set_deterministic(False)
c = curves["fp32_sums"]
for k in curves:
    rel = np.abs(curves[k] - c) / np.abs(c)
    print("%-11s vs fp32_sums: mean %.4f max %.4f first10 %.4f" % (k, rel.mean(), rel.max(), rel[:10].max()))
rel = np.abs(curves["bf16_det"] - curves["fp32_det"]) / np.abs(curves["fp32_det"])
print("bf16_det vs fp32_det: mean %.4f max %.4f first10 %.4f" % (rel.mean(), rel.max(), rel[:10].max()))
