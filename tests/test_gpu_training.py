"""GPU end-to-end: the harness trains the HIP CGCNN on the reference's Pt10 test structures (config 1
shape: CGCNN_demo hyper-parameters) and the val MAE matches the oracle trained identically on CPU."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _pt10(n=160):
    from matdeeplearn_amd.process import from_structures
    z = np.load(os.path.join(G, "pt10_dataset.npz"))
    structs = [dict(positions=z["positions"][s], numbers=z["numbers"][s], cell=z["cell"][s], pbc=z["pbc"][s]) for s in range(n)]
    return from_structures(structs, z["y"][:n], [str(v) for v in z["ids"][:n]])


def test_harness_trains_hip_cgcnn_and_tracks_the_oracle():
    from matdeeplearn_amd.training import train_regular
    from oracle import models as omodels, ops as oops
    training = dict(target_index=0, loss="l1_loss", train_ratio=0.8, val_ratio=0.1, test_ratio=0.1, verbosity=0)
    mp = dict(model="CGCNN", dim1=32, dim2=32, pre_fc_count=1, gc_count=2, post_fc_count=1, epochs=3, lr=0.002,
              batch_size=32, optimizer="AdamW", optimizer_args={}, scheduler="ReduceLROnPlateau",
              scheduler_args={"mode": "min", "factor": 0.8, "patience": 10}, batch_norm="False")
    job = dict(job_name="g", seed=5, save_model="False", write_output="False")
    quiet = lambda *a: None
    gpu = train_regular("cuda", 1, _pt10().to("cuda"), job, training, mp, log=quiet)
    cpu = train_regular("cpu", 1, _pt10().to("cpu"), job, training, mp, log=quiet,
                        model_factory=lambda n: omodels.REGISTRY[n], rbf=lambda d: oops.rbf_expand(d))
    # same seed -> same init, same split, same batch order; fp32 HIP vs fp32 CPU drift stays tiny over 3 epochs
    for a, b in zip(gpu["history"], cpu["history"]):
        assert abs(a["train"] - b["train"]) < 2e-3 * max(1.0, abs(b["train"])), (a, b)
    assert abs(gpu["val_error"] - cpu["val_error"]) < 2e-3 * max(1.0, abs(cpu["val_error"]))
    assert gpu["history"][0]["edges"] == cpu["history"][0]["edges"] > 0


def test_bf16_models_train_finite():
    from matdeeplearn_amd import models
    from matdeeplearn_amd.process import synthetic_bulk
    ds = synthetic_bulk(64, seed=2).to("cuda")
    b = ds.collate(np.arange(48), edge_dtype=torch.bfloat16, x_dtype=torch.bfloat16)
    for name, kw in [("CGCNN", {}), ("SchNet", dict(dim3=64)), ("GCN", {})]:
        torch.manual_seed(0)
        m = getattr(models, name)(ds, dim1=64, dim2=64, gc_count=2, post_fc_count=1, compute_dtype="bf16", **kw).to("cuda")
        out = m(b)
        assert out.dtype == torch.float32 and out.shape == (48,)
        torch.nn.functional.l1_loss(out, b.y).backward()
        assert all(p.grad is None or torch.isfinite(p.grad).all() for p in m.parameters()), name


_RCCL_SNIPPET = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from matdeeplearn_amd.training import FlatDataParallel
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(0)
m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1)).to(dev)
dp = FlatDataParallel(m)
dp.broadcast_state()                      # one flat broadcast per dtype over RCCL
x = torch.randn(32, 8, device=dev)
dp.zero_grad()
m(x).sum().backward()
ref = [p.grad.clone() for p in m.parameters()]
dp.reduce_grads(force=True)               # pack + all_reduce(SUM) on the side stream + average: identity at world size 1
torch.cuda.synchronize()
for p, r in zip(m.parameters(), ref):
    assert p.grad.data_ptr() != r.data_ptr() and torch.equal(p.grad, r), "all-reduce at world size 1 must be the identity"
assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(dp.params, dp.views))
t = torch.ones(4, device=dev); dist.all_reduce(t); assert float(t.sum()) == 4.0
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK")
"""


def test_rccl_backend_executes_on_one_gpu():
    """The `nccl` (= RCCL) branch of the data-parallel engine on real hardware at world size 1: process-group init bound
    to the device, flat broadcast, the side-stream pack + all-reduce, and bench.py's distributed path (MDL_FORCE_DIST=1)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_SNIPPET], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    env["MDL_FORCE_DIST"] = "1"
    env["MASTER_PORT"] = "29534"
    r = subprocess.run([sys.executable, "bench.py", "--graphs", "640", "--batch", "256", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline", "--no-extras"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert line, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(line[-1])
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["roofline"]["launches"] == 3 * 4


def test_graph_replayed_step_matches_the_eager_step():
    """training.GraphedStep (batch assembly + forward + loss + backward + fused AdamW captured once on padded static
    buffers, replayed per step) against the same steps run eagerly from the same initial weights: the padding must be
    invisible — same losses, same parameters (BatchNorm statistics over the rows that exist), same running stats and
    step counters — fp32 to rounding, bf16 to its own noise."""
    import copy
    from matdeeplearn_amd import models, ops
    from matdeeplearn_amd.process import synthetic_bulk
    from matdeeplearn_amd.training import GraphedStep, make_optimizer
    dev = torch.device("cuda:0")
    ds = synthetic_bulk(640, seed=11).to(dev)
    B = 96
    rng = np.random.default_rng(0)
    batches = [rng.choice(len(ds), size=B, replace=False) for _ in range(6)]
    for cd, dt, tol in (("fp32", torch.float32, 2e-4), ("bf16", torch.bfloat16, 3e-2)):
        torch.manual_seed(3)
        m_e = models.CGCNN(ds, dim1=64, dim2=64, gc_count=3, post_fc_count=2, compute_dtype=cd).to(dev)
        m_g = copy.deepcopy(m_e)
        o_e = make_optimizer(m_e.parameters(), "AdamW", lr=0.002)
        o_g = make_optimizer(m_g.parameters(), "AdamW", lr=0.002, capturable=True)
        gs = GraphedStep(ds, m_g, o_g, B, compute_dtype=dt)
        assert gs.sb.n_cap > max(int((ds.node_ptr[b + 1] - ds.node_ptr[b]).sum()) for b in batches)
        m_e.train()
        losses_e, losses_g = [], []
        for ids in batches:
            batch = ds.collate(ids, edge_dtype=dt, x_dtype=dt)
            o_e.zero_grad(set_to_none=True)
            with ops.zero_arena(dev):
                loss = torch.nn.functional.l1_loss(m_e(batch), batch.y)
                loss.backward()
            o_e.step()
            losses_e.append(float(loss))
            e, n = gs.step(ids)
            assert (e, n) == (batch.num_edges, batch.num_nodes)
            losses_g.append(float(gs.loss_value))
        assert gs.replays == len(batches) and gs.eager_steps == 0
        assert np.allclose(losses_g, losses_e, rtol=tol, atol=tol), (cd, losses_g, losses_e)
        sd_e, sd_g = m_e.state_dict(), m_g.state_dict()
        for k in sd_e:
            a, b = sd_g[k].float(), sd_e[k].float()
            # fp32: rounding only.  bf16: Adam divides by sqrt(v), so bf16 noise in a small gradient moves a parameter
            # by a sizeable fraction of lr (0.002) per step: six steps -> an absolute bound of a few lr
            bound = tol * (float(b.abs().max()) + 1e-3) * 5 if cd == "fp32" else 6 * 0.002
            assert float((a - b).abs().max()) <= bound, (cd, k)
        assert int(sd_g["bn_list.0.num_batches_tracked"]) == int(sd_e["bn_list.0.num_batches_tracked"])
        # a batch that does not fit the static capacity takes the eager path with the same optimizer state
        big = np.argsort(-(ds.node_ptr[1:] - ds.node_ptr[:-1]))[:B]
        if not gs.sb.fits(big):
            gs.step(big)
            assert gs.eager_steps == 1
