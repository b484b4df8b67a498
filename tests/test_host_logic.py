"""CPU suite for the host side: C-ABI surface, dataset/loader logic, training loops (with the oracle
model — the product has no CPU compute path), and the data-parallel engine over gloo (world_size 2)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------------------------------------
# C ABI
# ---------------------------------------------------------------------------------------------
def test_library_loads_and_exports_every_declared_symbol():
    from matdeeplearn_amd import _lib
    handle = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "mdl_hip.h")).read()
    declared = set(re.findall(r"\b(mdl_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found in include/mdl_hip.h"
    for name in declared:
        assert hasattr(handle, name), "declared in mdl_hip.h but not exported: " + name
    assert declared == set(_lib.PROTOTYPES), "ctypes prototype table out of sync with mdl_hip.h"
    assert handle.mdl_version() == 100
    assert handle.mdl_cgconv_wpack_bytes(64, 50, _lib.MDL_BF16) == 128 * 200 * 2
    assert handle.mdl_cgconv_wpack_bytes(64, 50, 7) == 0


def test_library_reads_no_environment_and_ships_no_experiment_kernels():
    """include/mdl_hip.h promises a stateless library: libmdl_hip.so must not import getenv (variants are explicit flag bits in
    `dtype`), and the measured-negative kernels live in the experiments build only (experiments/), not in the product."""
    import shutil
    import subprocess
    from matdeeplearn_amd import _lib
    nm = shutil.which("nm")
    if nm is None:
        pytest.skip("nm not available")
    undefined = subprocess.run([nm, "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined, "libmdl_hip.so reads the environment"
    exported = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for gone in ("mdl_cgconv_fwd_save", "mdl_cgconv_bwd_saved", "mdl_cgconv_fwd_p", "mdl_cgconv_bwd_p", "mdl_mlp2", "cb10fwd_kernel",
                 "cgconv_bwd_ab_kernel"):
        assert gone not in exported, gone
    assert _lib.MDL_DETERMINISTIC == 0x100 and "#define MDL_DETERMINISTIC 0x100" in open(os.path.join(ROOT, "include", "mdl_hip.h")).read()


def test_ops_fail_loudly_without_a_hip_device():
    from matdeeplearn_amd import nn as mnn, ops
    x = torch.randn(4, 64)
    ei = torch.tensor([[0, 1, 2], [1, 2, 3]])
    with pytest.raises(ops.MdlError):
        ops.scatter(x, torch.tensor([0, 0, 1, 1]), 0, 2, "mean")
    with pytest.raises(ops.MdlError):
        ops.rbf_expand(torch.rand(5))
    with pytest.raises(ops.MdlError):
        mnn.CGConv(64, 50, aggr="mean")(x, ei, torch.randn(3, 50))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "matdeeplearn_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dirpath, f)


# ---------------------------------------------------------------------------------------------
# graph builder / dataset / loader
# ---------------------------------------------------------------------------------------------
def test_product_graph_builder_matches_reference_goldens():
    from matdeeplearn_amd.process import graph as pg
    z = np.load(os.path.join(G, "threshold_sort.npz"))
    for t in "abcde":
        assert np.array_equal(pg.threshold_sort(z["D_" + t], float(z["r_" + t]), int(z["k_" + t])), z["out_" + t])
    ds, gg = np.load(os.path.join(G, "pt10_dataset.npz")), np.load(os.path.join(G, "pt10_graphs.npz"))
    counts = []
    for s in range(len(ds["ids"])):
        r = pg.build_graph(ds["positions"][s], ds["numbers"][s], ds["cell"][s], ds["pbc"][s])
        counts.append(r["edge_index"].shape[1])
        if s < 8:
            assert np.array_equal(r["edge_index"], gg["edge_index_%d" % s])
            assert np.array_equal(r["edge_weight"], gg["edge_weight_%d" % s])
            assert r["x"].shape == (10, 114)
    assert np.array_equal(np.array(counts), gg["edges_per_graph"]) and sum(counts) == 99672
    oh = np.load(os.path.join(G, "onehot_degree.npz"))
    assert np.array_equal(pg.one_hot_degree(oh["edge_index"], 10, 13), oh["x"][:, 1:])


def test_minimum_image_distances_against_bruteforce():
    from matdeeplearn_amd.process import graph as pg
    from oracle import graph as og
    rng = np.random.default_rng(0)
    cell = np.diag([7.0, 8.0, 9.0])
    pos = rng.uniform(0, 9, size=(12, 3))
    a = pg.distance_matrix(pos, cell, [True, True, False])
    b = og.mic_distances(pos, cell, [True, True, False])
    assert np.allclose(a, b, atol=1e-12)


@pytest.fixture(scope="module")
def small_ds():
    from matdeeplearn_amd.process import synthetic_bulk
    return synthetic_bulk(64, seed=3)


def test_synthetic_bulk_graph_invariants(small_ds):
    ds = small_ds
    assert ds.x.shape[1] == 114 and len(ds) == 64
    for g in range(len(ds)):
        n0, n1, e0, e1 = ds.node_ptr[g], ds.node_ptr[g + 1], ds.edge_ptr[g], ds.edge_ptr[g + 1]
        n = n1 - n0
        src, tgt, d = ds.src[e0:e1], ds.tgt[e0:e1], ds.dist[e0:e1]
        assert np.all(np.diff(tgt) >= 0), "edges must be sorted by target inside every graph"
        assert src.min() >= 0 and src.max() < n and tgt.max() < n
        loops = src == tgt
        assert loops.sum() == n and np.all(d[loops] == 0)                 # exactly one self loop / node, weight 0
        out_deg = np.bincount(src[~loops], minlength=n)
        assert out_deg.max() <= 12                                           # <= 12 non-self out-edges per node
        assert np.all(d[~loops] > 0) and d.max() <= 8.0
        assert len(set(zip(src.tolist(), tgt.tolist()))) == e1 - e0         # no duplicate (i, j)
        assert np.array_equal(ds.x[n0:n1, 100:].argmax(1), out_deg + 1)     # one-hot OUT-degree incl. the loop
    assert ds.dist_norm.min() == 0.0 and abs(ds.dist_norm.max() - 1.0) < 1e-6


def test_batch_assembly_matches_per_graph_concatenation(small_ds):
    ds = small_ds
    ds.to("cpu")
    ids = np.array([5, 1, 7, 60, 5])
    b, dn = ds.assemble(ids)
    off = 0
    for k, g in enumerate(ids):
        n0, n1, e0, e1 = ds.node_ptr[g], ds.node_ptr[g + 1], ds.edge_ptr[g], ds.edge_ptr[g + 1]
        sl = b.batch == k
        assert torch.equal(b.x[sl], torch.from_numpy(ds.x[n0:n1]))
        m = (b.csr.tgt >= off) & (b.csr.tgt < off + (n1 - n0))
        assert torch.equal(b.csr.src[m] - off, torch.from_numpy(ds.src[e0:e1]))
        assert torch.equal(b.csr.tgt[m] - off, torch.from_numpy(ds.tgt[e0:e1]))
        assert torch.equal(dn[m], torch.from_numpy(ds.dist_norm[e0:e1]))
        off += n1 - n0
    assert (b.csr.tgt[1:] >= b.csr.tgt[:-1]).all()
    cnt = torch.bincount(b.csr.tgt.long(), minlength=b.num_nodes)
    assert torch.equal(torch.cumsum(cnt, 0).int(), b.csr.rowptr[1:]) and int(b.csr.rowptr[0]) == 0
    assert torch.equal(b.y, torch.from_numpy(ds.y[ids, 0])) and b.u.shape == (5, 3) and b.num_graphs == 5
    # lazily materialised PyG-style edge_index
    assert b.edge_index.shape == (2, b.num_edges) and b.edge_index.dtype == torch.int64


def test_dataset_by_source_order_is_the_stable_per_graph_sort(small_ds):
    """GraphDataset.by_source(): eperm_s = graph-local edge ids in stable by-source order, lrowptr_s = exclusive
    out-degree prefix inside the graph — what mdl_assemble_transposed turns into the batch's by-source CSR."""
    ds = small_ds
    ds.to("cpu")
    eperm_s, lrowptr_s = (a.numpy() for a in ds.by_source())
    assert eperm_s.dtype == np.int32 and len(eperm_s) == len(ds.src) and len(lrowptr_s) == len(ds.z)
    for g in range(len(ds)):
        n0, n1, e0, e1 = ds.node_ptr[g], ds.node_ptr[g + 1], ds.edge_ptr[g], ds.edge_ptr[g + 1]
        src = ds.src[e0:e1]
        assert np.array_equal(eperm_s[e0:e1], np.argsort(src, kind="stable"))
        out_deg = np.bincount(src, minlength=n1 - n0)
        assert np.array_equal(lrowptr_s[n0:n1], np.concatenate([[0], np.cumsum(out_deg)[:-1]]))


def test_device_loader_partitions_like_distributed_sampler(small_ds):
    from matdeeplearn_amd.process import DeviceLoader
    ds = small_ds
    ds.to("cpu")
    idx = np.arange(50)
    seen = []
    for r in range(4):
        ld = DeviceLoader(ds, idx, batch_size=5, shuffle=True, seed=7, rank=r, world_size=4, rbf=lambda d: torch.zeros(len(d), 50))
        ld.set_epoch(3)
        order = ld._order()
        assert len(order) == 13 and len(ld) == 3                        # ceil(50/4) padded, ceil(13/5) batches
        seen.append(order)
    flat = np.concatenate(seen)
    assert set(flat.tolist()) == set(range(50)) and len(flat) == 52        # every index, 2 padded repeats
    ld0 = DeviceLoader(ds, idx, batch_size=5, shuffle=True, seed=7, rbf=lambda d: torch.zeros(len(d), 50))
    ld0.set_epoch(3)
    a = ld0._order()
    ld0.set_epoch(4)
    assert not np.array_equal(a, ld0._order())


# ---------------------------------------------------------------------------------------------
# training loops on the reference's own test dataset (config 1: Pt10 CGCNN, CPU plumbing)
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def pt10():
    from matdeeplearn_amd.process import from_structures
    z = np.load(os.path.join(G, "pt10_dataset.npz"))
    n = 200
    structs = [dict(positions=z["positions"][s], numbers=z["numbers"][s], cell=z["cell"][s], pbc=z["pbc"][s])
               for s in range(n)]
    ds = from_structures(structs, z["y"][:n], [str(v) for v in z["ids"][:n]])
    return ds.to("cpu")


def test_cfg1_pt10_cgcnn_trains_on_cpu_through_the_harness(pt10):
    from matdeeplearn_amd.process import DeviceLoader, split_data
    from matdeeplearn_amd.training import make_optimizer, make_scheduler, trainer, evaluate
    from oracle import models as omodels, ops as oops
    torch.manual_seed(42)
    rbf = lambda d: oops.rbf_expand(d)
    tr, va, te = split_data(len(pt10), 0.8, 0.05, 0.15, seed=42)
    assert (len(tr), len(va), len(te)) == (160, 10, 30)
    mk = lambda idx, sh: DeviceLoader(pt10, idx, batch_size=40, shuffle=sh, seed=1, rbf=rbf)
    model = omodels.CGCNN(pt10, dim1=100, dim2=150, pre_fc_count=1, gc_count=4, post_fc_count=3)   # config.yml CGCNN_demo
    opt = make_optimizer(model.parameters(), "AdamW", lr=0.002)
    sch = make_scheduler(opt, "ReduceLROnPlateau", mode="min", factor=0.8, patience=10, min_lr=1e-5, threshold=2e-4)
    model, hist = trainer("cpu", 1, model, opt, sch, "l1_loss", mk(tr, True), mk(va, False), epochs=4, verbosity=0)
    assert len(hist) == 4 and all(np.isfinite(h["train"]) and np.isfinite(h["val"]) for h in hist)
    assert hist[-1]["train"] < hist[0]["train"], hist                       # it learns
    assert hist[0]["edges"] == int(sum(np.diff(pt10.edge_ptr)[tr]))         # edges/s bookkeeping
    loss, rows = evaluate(mk(te, False), model, "l1_loss", out=True)
    assert rows.shape == (30, 3) and np.isfinite(float(loss))


def test_train_loss_is_sample_weighted_mean(pt10):
    """training.py:45,51-53 — sum(batch-mean x batch-size) / sum(batch-size) == dataset MAE."""
    from matdeeplearn_amd.process import DeviceLoader
    from matdeeplearn_amd.training import evaluate
    from oracle import models as omodels, ops as oops
    torch.manual_seed(0)
    rbf = lambda d: oops.rbf_expand(d)
    model = omodels.CGCNN(pt10, dim1=32, dim2=32, gc_count=1, post_fc_count=1)
    idx = np.arange(50)
    a = float(evaluate(DeviceLoader(pt10, idx, batch_size=16, rbf=rbf), model, "l1_loss"))
    b = float(evaluate(DeviceLoader(pt10, idx, batch_size=50, rbf=rbf), model, "l1_loss"))
    assert abs(a - b) < 1e-5 * max(1.0, abs(b))


# ---------------------------------------------------------------------------------------------
# data parallel engine: world_size 2 over gloo
# ---------------------------------------------------------------------------------------------
def _dp_worker(rank, world, port, tmp):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from matdeeplearn_amd.process import DeviceLoader, synthetic_bulk
    from matdeeplearn_amd.training import FlatDataParallel, ddp_setup, ddp_cleanup, make_optimizer
    from oracle import models as omodels, ops as oops
    assert ddp_setup(rank, world, backend="gloo", master_port=port)
    torch.manual_seed(100 + rank)                     # different init per rank: broadcast must fix it
    ds = synthetic_bulk(32, seed=5).to("cpu")
    rbf = lambda d: oops.rbf_expand(d)
    model = omodels.CGCNN(ds, dim1=16, dim2=16, gc_count=2, post_fc_count=1, batch_norm="False")
    dp = FlatDataParallel(model)
    p0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(p0) for _ in range(world)]
    dist.all_gather(gathered, p0)
    assert torch.equal(gathered[0], gathered[1]), "parameter broadcast failed"
    # each rank: its shard of one global batch of 8 graphs
    ids = np.arange(8)
    batch = ds.collate(ids[rank::world], rbf=rbf)
    dp.zero_grad()
    loss = torch.nn.functional.l1_loss(model(batch), batch.y, reduction="sum") / 8.0
    loss.backward()
    dp.reduce_grads()
    # reference: the full batch on one process; DDP averages, so compare with grad / world
    ref = omodels.CGCNN(ds, dim1=16, dim2=16, gc_count=2, post_fc_count=1, batch_norm="False")
    ref.load_state_dict(model.state_dict())
    full = ds.collate(ids, rbf=rbf)
    (torch.nn.functional.l1_loss(ref(full), full.y, reduction="sum") / 8.0).backward()
    for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p.grad, q.grad / world, rtol=1e-4, atol=1e-6), k
        assert p.grad.data_ptr() >= dp.flat_grad.data_ptr()          # after the reduce, grads are views of the flat buffer
    opt = make_optimizer(model.parameters(), "AdamW", lr=0.01)
    opt.step()
    p1 = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    dist.all_gather(gathered, p1)
    assert torch.equal(gathered[0], gathered[1]), "ranks diverged after the optimizer step"
    # two-chunk exchange (payloads above chunk_bytes; opt-in): the first step runs ONE collective while hooks observe the order
    # in which the gradients become ready; from the second step on the first-ready half leaves from a hook while the backward
    # still runs, the rest afterwards — same gradients, two collectives, the early chunk (offset 0 of the re-laid-out buffer) first
    m2 = omodels.CGCNN(ds, dim1=16, dim2=16, gc_count=2, post_fc_count=1, batch_norm="False")
    m2.load_state_dict(ref.state_dict())
    dp2 = FlatDataParallel(m2, chunk_bytes=1024)
    assert dp2.split == "observe"
    calls = []
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        calls.append((t.data_ptr() - dp2.flat_grad.data_ptr()) // 4)
        return real_all_reduce(t, *a, **k)
    dist.all_reduce = counting_all_reduce
    try:
        for step in range(3):                             # the hook countdown re-arms in zero_grad()
            calls.clear()
            dp2.zero_grad()
            (torch.nn.functional.l1_loss(m2(batch), batch.y, reduction="sum") / 8.0).backward()
            if step == 0:
                assert calls == [], "the observed step exchanges nothing before reduce_grads()"
                dp2.reduce_grads()
                assert calls == [0] and isinstance(dp2.split, tuple) and 0 < dp2.split[0] < len(dp2.params)
                # backward order, not registration order: the output layer's gradients are ready first
                names = {id(p): k for k, p in m2.named_parameters()}
                assert names[id(dp2.params[0])].startswith("lin_out") and names[id(dp2.params[-1])].startswith("pre_lin_list.0")
            else:
                assert calls == [0], "the first-ready chunk must be on its way when the backward returns"
                dp2.reduce_grads()
                assert calls == [0, dp2.split[1]]
            for (k, p), (_, q) in zip(m2.named_parameters(), ref.named_parameters()):
                assert torch.allclose(p.grad, q.grad / world, rtol=1e-4, atol=1e-6), (step, k)
        dp2.single_collective()                           # what training.GraphedStep asks for: one collective per step
        calls.clear()
        dp2.zero_grad()
        (torch.nn.functional.l1_loss(m2(batch), batch.y, reduction="sum") / 8.0).backward()
        dp2.reduce_grads()
        assert calls == [0]
    finally:
        dist.all_reduce = real_all_reduce
    if rank == 0:
        open(tmp, "w").write("ok %d" % dp.grad_bytes())
    ddp_cleanup()


def test_flat_data_parallel_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    out = str(tmp_path / "dp.txt")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read().startswith("ok")


def _replica_worker(rank, world, port, out_path):
    """Seed agreement + replica-per-rank Repeat / Ensemble drivers on a world_size-2 gloo group."""
    import torch.distributed as dist
    from matdeeplearn_amd.process import from_structures
    from matdeeplearn_amd.training import (ddp_cleanup, ddp_setup, resolve_seed, train_ensemble_replicas,
                                           train_repeat_replicas)
    from oracle import models as omodels, ops as oops
    os.environ["MASTER_PORT"] = str(port)
    ddp_setup(rank, world, backend="gloo", master_port=port)
    np.random.seed(100 + rank)                                    # ranks would draw DIFFERENT seeds on their own
    seeds = [resolve_seed(0), resolve_seed(0), resolve_seed(7)]
    got = [None, None]
    dist.all_gather_object(got, seeds)
    assert got[0] == got[1] and got[0][2] == 7 and got[0][0] != got[0][1]
    z = np.load(os.path.join(G, "pt10_dataset.npz"))
    n = 60
    structs = [dict(positions=z["positions"][s], numbers=z["numbers"][s], cell=z["cell"][s], pbc=z["pbc"][s]) for s in range(n)]
    ds = from_structures(structs, z["y"][:n], [str(v) for v in z["ids"][:n]]).to("cpu")
    training = dict(target_index=0, loss="l1_loss", train_ratio=0.7, val_ratio=0.1, test_ratio=0.2, verbosity=0)
    mp_ = dict(model="CGCNN", dim1=8, dim2=8, gc_count=1, post_fc_count=1, epochs=1, lr=0.005, batch_size=20,
               optimizer="AdamW", optimizer_args={}, scheduler="ReduceLROnPlateau", scheduler_args={"mode": "min"})
    kw = dict(model_factory=lambda name: omodels.REGISTRY[name], rbf=lambda d: oops.rbf_expand(d), log=lambda *a: None)
    job = dict(job_name="r", seed=0, save_model="False", write_output="False", repeat_trials=3)
    rep = train_repeat_replicas(rank, world, ds, job, training, mp_, **kw)
    assert rep["errors"].shape == (3, 3) and np.isfinite(rep["errors"]).all()
    ens = train_ensemble_replicas(rank, world, ds, dict(job, seed=5), training, [mp_, dict(mp_, model="GCN"), dict(mp_, dim1=12)], **kw)
    assert ens["model_errors"].shape == (3,) and np.isfinite(ens["ensemble_error"])
    both = [None, None]
    dist.all_gather_object(both, (rep["errors"].tolist(), ens["ensemble_error"]))
    assert both[0] == both[1]                                     # every rank holds the same gathered table
    if rank == 0:
        open(out_path, "w").write("ok")
    ddp_cleanup()


def test_seed_agreement_and_replica_drivers_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    out = str(tmp_path / "rep.txt")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    mp.spawn(_replica_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _train_regular_worker(rank, world, port, out_path):
    """The reference's DDP training job (training.py:377-539: ddp_setup, DistributedSampler partition, lr x world size, one
    gradient exchange per step, per-epoch error reduction) through train_regular on two gloo ranks, with the validation
    sharded over the ranks and with the reference's rank-0 validation: same validation errors, ranks in lock step."""
    import torch.distributed as dist
    from matdeeplearn_amd.process import from_structures
    from matdeeplearn_amd.training import ddp_cleanup, ddp_setup, train_regular
    from oracle import models as omodels, ops as oops
    os.environ["MASTER_PORT"] = str(port)
    ddp_setup(rank, world, backend="gloo", master_port=port)
    z = np.load(os.path.join(G, "pt10_dataset.npz"))
    n = 90
    structs = [dict(positions=z["positions"][s], numbers=z["numbers"][s], cell=z["cell"][s], pbc=z["pbc"][s]) for s in range(n)]
    ds = from_structures(structs, z["y"][:n], [str(v) for v in z["ids"][:n]]).to("cpu")
    job = dict(job_name="d", seed=11, save_model="False", write_output="False")
    mp_ = dict(model="CGCNN", dim1=8, dim2=8, gc_count=2, post_fc_count=1, epochs=3, lr=0.005, batch_size=16,
               optimizer="AdamW", optimizer_args={}, scheduler="ReduceLROnPlateau", scheduler_args={"mode": "min"})
    kw = dict(model_factory=lambda name: omodels.REGISTRY[name], rbf=lambda d: oops.rbf_expand(d), log=lambda *a: None)
    res = {}
    for tag, shard in (("sharded", "True"), ("rank0", "False")):
        training = dict(target_index=0, loss="l1_loss", train_ratio=0.6, val_ratio=0.25, test_ratio=0.15, verbosity=0,
                        shard_validation=shard)
        r = train_regular(rank, world, ds, job, training, mp_, **kw)
        res[tag] = ([h["val"] for h in r["history"]], [h["train"] for h in r["history"]],
                    torch.cat([p.detach().reshape(-1) for p in r["model"].parameters()]),
                    torch.cat([b.detach().double().reshape(-1) for b in r["model"].buffers()]))
    both = [None, None]
    dist.all_gather_object(both, {k: (v[0], v[1], v[2].tolist(), v[3].tolist()) for k, v in res.items()})
    a, b = both
    # sharded: every rank knows every epoch's validation error, keeps the same best weights and (after the broadcast in front of
    # the validation) the same buffers
    assert a["sharded"][0] == b["sharded"][0] and all(v is not None for v in a["sharded"][0])
    assert a["sharded"][2] == b["sharded"][2] and a["sharded"][3] == b["sharded"][3]
    # the reference's form: rank 0 validates alone, rank 1 never learns the number
    assert all(v is not None for v in a["rank0"][0]) and all(v is None for v in b["rank0"][0])
    # same job, same seeds: the training errors agree across the two forms and the validation errors are the same numbers
    assert a["sharded"][1] == a["rank0"][1]
    assert np.allclose(a["sharded"][0], a["rank0"][0], rtol=1e-5, atol=1e-7), (a["sharded"][0], a["rank0"][0])
    # an EMPTY validation split (val_ratio = 0): "no validation" in both forms — every epoch's val is None and the LATEST weights
    # are kept (training.py:130-170 without a val loader), not the epoch-1 weights a best-of-zero-errors rule would reload
    nov = {}
    for tag, shard in (("sharded", "True"), ("rank0", "False")):
        training = dict(target_index=0, loss="l1_loss", train_ratio=0.8, val_ratio=0.0, test_ratio=0.2, verbosity=0,
                        shard_validation=shard)
        r = train_regular(rank, world, ds, job, training, mp_, **kw)
        assert [h["val"] for h in r["history"]] == [None] * 3, (tag, [h["val"] for h in r["history"]])
        nov[tag] = torch.cat([p.detach().reshape(-1) for p in r["model"].parameters()])
    assert torch.equal(nov["sharded"], nov["rank0"])
    if rank == 0:
        open(out_path, "w").write("ok")
    ddp_cleanup()


def test_distributed_training_job_with_sharded_validation_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = 33500 + (os.getpid() % 2000)
    out = str(tmp_path / "tr.txt")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    mp.spawn(_train_regular_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_flat_on_disk_dataset_round_trip(tmp_path):
    """GraphDataset.save_flat / load_flat (the own on-disk format): every array bit-identical, memory-mapped or
    read; the reloaded dataset assembles the same batches (same x, CSR, distances, targets) as the original."""
    from matdeeplearn_amd.process import GraphDataset, synthetic_bulk
    ds = synthetic_bulk(40, seed=9)
    ds.target_index = 0
    path = str(tmp_path / "bulk.mdlflat")
    ds.save_flat(path)
    assert open(path, "rb").read(8) == GraphDataset.FLAT_MAGIC
    for mm in (True, False):
        d2 = GraphDataset.load_flat(path, mmap=mm)
        for k in GraphDataset._FLAT_FIELDS + ("dist_norm", "in_deg", "lrowptr"):
            a, b = np.asarray(getattr(ds, k)), np.asarray(getattr(d2, k))
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), k
        assert d2.ids == ds.ids and d2.dist_range == ds.dist_range and d2.num_edge_features == ds.num_edge_features
        ids = np.array([3, 17, 5, 22])
        b1, n1 = ds.to("cpu").assemble(ids)
        b2, n2 = d2.to("cpu").assemble(ids)
        assert torch.equal(b1.x, b2.x) and torch.equal(b1.csr.rowptr, b2.csr.rowptr) and torch.equal(b1.csr.src, b2.csr.src)
        assert torch.equal(n1, n2) and torch.equal(b1.y, b2.y) and torch.equal(b1.edge_weight, b2.edge_weight)


def test_minimum_image_distances_in_skewed_cells():
    """distance_matrix == brute force over a wide image range for thin / strongly skewed triclinic cells (where the true
    minimum image lies outside the +-1 images of the raw cell), mixed pbc, and is unchanged for orthorhombic cells."""
    from matdeeplearn_amd.process import graph as pg
    rng = np.random.default_rng(5)
    cells = [np.array([[4.0, 0, 0], [3.6, 1.2, 0], [0.3, 0.2, 9.0]]),            # thin, strongly skewed in the ab plane
             np.array([[5.0, 0, 0], [9.0, 2.0, 0], [7.0, 5.0, 3.0]]),            # every pair far from orthogonal
             np.diag([6.0, 7.0, 8.0])]
    for cell in cells:
        for pbc in ([True, True, True], [True, True, False]):
            frac = rng.uniform(0, 1, size=(9, 3))
            pos = frac @ cell
            got = pg.distance_matrix(pos, cell, pbc)
            d = pos[None] - pos[:, None]
            best = np.full(d.shape[:2], np.inf)
            R = range(-6, 7)
            for a in (R if pbc[0] else [0]):
                for b in (R if pbc[1] else [0]):
                    for c in (R if pbc[2] else [0]):
                        v = d + a * cell[0] + b * cell[1] + c * cell[2]
                        best = np.minimum(best, np.sqrt((v * v).sum(-1)))
            assert np.allclose(got, best, atol=1e-9), (cell, pbc, np.abs(got - best).max())


# ---------------------------------------------------------------------------------------------
# job drivers (Training / Repeat / CV / Ensemble / Predict) on CPU with the oracle models
# ---------------------------------------------------------------------------------------------
def _oracle_factory(name):
    from oracle import models as omodels
    return omodels.REGISTRY[name]


def test_drivers_regular_cv_ensemble_predict(pt10, tmp_path, monkeypatch):
    from matdeeplearn_amd.training import train_regular, train_CV, train_ensemble, train_repeat, predict
    from oracle import ops as oops
    monkeypatch.chdir(tmp_path)
    rbf = lambda d: oops.rbf_expand(d)
    training = dict(target_index=0, loss="l1_loss", train_ratio=0.8, val_ratio=0.05, test_ratio=0.15, verbosity=0)
    mp = dict(model="CGCNN", dim1=16, dim2=16, gc_count=1, post_fc_count=1, epochs=2, lr=0.005, batch_size=50,
              optimizer="AdamW", optimizer_args={}, scheduler="ReduceLROnPlateau",
              scheduler_args={"mode": "min", "factor": 0.8, "patience": 10})
    job = dict(job_name="t", seed=11, save_model="True", model_path="m.pth", write_output="True")
    kw = dict(model_factory=_oracle_factory, rbf=rbf, log=lambda *a: None)
    r = train_regular("cpu", 1, pt10, job, training, mp, **kw)
    assert np.isfinite([r["train_error"], r["val_error"], r["test_error"]]).all()
    assert os.path.exists("m.pth") and os.path.exists("t_test_outputs.csv")
    rows = open("t_test_outputs.csv").read().strip().splitlines()
    assert rows[0] == "ids,target,prediction" and len(rows) == 1 + 30
    err, prows = predict(pt10, "CGCNN", mp, "m.pth", model_factory=_oracle_factory, rbf=rbf)
    assert prows.shape == (200, 3) and np.isfinite(err)
    cv = train_CV("cpu", 1, pt10, dict(job, cv_folds=4, save_model="False", write_output="False"), training, mp, **kw)
    assert cv["fold_errors"].shape == (4,) and cv["rows"].shape[0] == 200
    ens = train_ensemble("cpu", 1, pt10, dict(job, save_model="False", write_output="False"), training,
                         [mp, dict(mp, model="GCN"), dict(mp, model="SchNet", dim3=8)], **kw)
    assert ens["model_errors"].shape == (3,) and np.isfinite(ens["ensemble_error"])
    rep = train_repeat("cpu", 1, pt10, dict(job, repeat_trials=2, save_model="False", write_output="False"), training, mp, **kw)
    assert rep["errors"].shape == (2, 3)


def test_graph_replay_key_off_the_device_and_loader_batch_ids(pt10):
    """Training.graph_replay (driver.graph_replay_wanted): a dataset that is not resident on a HIP device never replays in auto
    mode — the CPU jobs of this suite stay what they were — and "True" says why it cannot instead of falling back silently;
    DeviceLoader.batch_ids() hands out the id batches in the order __iter__ assembles them (shuffled per epoch, rank-sharded)."""
    from matdeeplearn_amd.process import DeviceLoader
    from matdeeplearn_amd.training import train_regular
    from matdeeplearn_amd.training.driver import graph_replay_wanted
    from oracle import ops as oops
    ld = DeviceLoader(pt10, np.arange(150), 40, shuffle=True, seed=3, rbf=lambda d: oops.rbf_expand(d))
    model = _oracle_factory("CGCNN")(data=pt10, dim1=8, dim2=8, gc_count=1, post_fc_count=1)
    for mode in ("auto", "False"):
        assert graph_replay_wanted(mode, pt10, model, ld, None, False, "AdamW", "l1_loss") is False
    with pytest.raises(ValueError):
        graph_replay_wanted("True", pt10, model, ld, None, False, "AdamW", "l1_loss")
    training = dict(target_index=0, loss="l1_loss", train_ratio=0.8, val_ratio=0.05, test_ratio=0.15, verbosity=0, graph_replay="True")
    mp = dict(model="CGCNN", dim1=8, dim2=8, gc_count=1, post_fc_count=1, epochs=1, lr=0.005, batch_size=50, optimizer="AdamW",
              optimizer_args={}, scheduler="ReduceLROnPlateau", scheduler_args={"mode": "min", "factor": 0.8, "patience": 10})
    with pytest.raises(ValueError):
        train_regular("cpu", 1, pt10, dict(job_name="t", seed=1, save_model="False", write_output="False"), training, mp,
                      model_factory=_oracle_factory, rbf=lambda d: oops.rbf_expand(d), log=lambda *a: None)
    for epoch in (0, 1):
        ld.set_epoch(epoch)
        got = [ids.tolist() for ids in ld.batch_ids()]
        assert [len(g) for g in got] == [40, 40, 40, 30] and sorted(sum(got, [])) == list(range(150))
        ys = np.asarray(pt10.y)[:, 0]
        for ids, b in zip(got, ld):                                  # the assembled batches hold exactly these graphs, in this order
            assert np.array_equal(np.asarray(b.y).reshape(-1), ys[np.asarray(ids)])
    ld2 = [DeviceLoader(pt10, np.arange(150), 40, shuffle=True, seed=3, rank=r, world_size=2) for r in (0, 1)]
    parts = [sum((ids.tolist() for ids in l.batch_ids()), []) for l in ld2]
    assert len(parts[0]) == len(parts[1]) == 75 and sorted(parts[0] + parts[1]) == list(range(150))


def test_load_reference_style_config(tmp_path):
    from matdeeplearn_amd.training import load_config
    cfg = tmp_path / "config.yml"
    cfg.write_text("""
Job:
    Training:
        job_name: "my_train_job"
        model: CGCNN_demo
        seed: 0
        save_model: "True"
Processing:
    graph_max_radius: 8.0
    graph_max_neighbors: 12
Training:
    target_index: 0
    loss: "l1_loss"
    train_ratio: 0.8
    val_ratio: 0.05
    test_ratio: 0.15
    verbosity: 5
Models:
    CGCNN_demo:
        model: CGCNN
        dim1: 100
        gc_count: 4
        batch_norm: "True"
        lr: 0.002
        batch_size: 100
""")
    job, proc, tr, mp = load_config(str(cfg), "Training")
    assert job["model"] == "CGCNN_demo" and mp["model"] == "CGCNN" and mp["batch_norm"] == "True"
    assert proc["graph_max_neighbors"] == 12 and tr["loss"] == "l1_loss"


# ---------------------------------------------------------------------------------------------
# static check of the compiled conv kernels (no GPU needed: hipcc cross-compiles gfx950)
# ---------------------------------------------------------------------------------------------
def test_conv_kernels_register_budget():
    """Spills are a performance bug in these kernels, not a detail: a scratch reload is a VMEM load whose wait drains
    the whole prefetch queue.  None of the conv kernels may spill."""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    text = ""
    with tempfile.TemporaryDirectory() as td:
        for name in ("cgconv.hip", "cgconv_node.hip"):
            src = os.path.join(ROOT, "matdeeplearn_amd", "csrc", name)
            out = os.path.join(td, name + ".s")
            subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
                            "-Wno-unused-result", "-DMDL_CG_FAST_ONLY", "-S", "--cuda-device-only", "-o", out, src],
                           check=True, capture_output=True, timeout=600)
            text += open(out).read()
    stats = {}
    for m in re.finditer(r"^(_ZN3mdl[0-9A-Za-z_]+):.*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", text, re.S | re.M):
        stats[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    def find(frag):
        hits = [v for k, v in stats.items() if frag in k]
        assert hits, "kernel %s not found in the assembly" % frag
        return hits[0]
    scratch, occ = find("cgconv_bwd_kernelItLi64ELi50ELi9ELi2ELi1ELi0E")          # per-wave backward, bf16 C=64 G=50
    assert scratch == 0 and occ == 1
    scratch, occ = find("cgconv_fwd_kernelItLi64ELi50ELi9ELi2ELi1ELb0ELi0ELb0E")  # all-slices forward
    assert scratch == 0 and occ == 2
    # ... with the BatchNorm statistics in its epilogue (an instantiation of its own so that the plain one keeps its allocation):
    # a few 64-bit base pointers may spill, stored in the prologue and reloaded in the group epilogue — never in the tile loop
    scratch, occ = find("cgconv_fwd_kernelItLi64ELi50ELi9ELi2ELi1ELb0ELi0ELb1E")
    assert scratch <= 32 and occ == 2
    scratch, occ = find("cgconv_fwd_kernelItLi128ELi50ELi9ELi2ELi1ELb0ELi0ELb0E") # 128-channel forward (padded C = 100): one
    assert scratch == 0 and occ == 2                                              # slice per wave
    scratch, occ = find("cgconv_bwd_kernelItLi128ELi50ELi9ELi2ELi1ELi0E")         # 128-channel backward: a few loop-invariant
    assert scratch <= 64 and occ == 1                                             # dwords spill
    for frag in ("cgconv_node_stream_kernelILi64", "cgconv_node_stream_kernelILi32"):    # node-level dense half
        scratch, occ = find(frag)
        assert scratch == 0 and occ >= 2
    # kernel 2 of K3 and K4 (the fused CFConv forward): their translation units are built with the VGPR form of the MFMA results
    # (matdeeplearn_amd/_build.py FILE_FLAGS).  K4 is the kernel that paid for the lesson in round 5: 28 spilled registers, each
    # reloaded behind a vmcnt(0) in its tile loop, were a third of its time ("Round 5" item 8)
    text = ""
    with tempfile.TemporaryDirectory() as td:
        for name in ("cgconv_ep.hip", "cfconv.hip"):
            src = os.path.join(ROOT, "matdeeplearn_amd", "csrc", name)
            out = os.path.join(td, name + ".s")
            subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result",
                            "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only", "-o", out, src],
                           check=True, capture_output=True, timeout=600)
            text += open(out).read()
    stats.clear()
    for m in re.finditer(r"^(_ZN3mdl[0-9A-Za-z_]+):.*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", text, re.S | re.M):
        stats[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    for frag in ("2ep11bwd2_kernelILi64E", "2cf17cfconv_fwd_kernel"):
        scratch, occ = find(frag)
        assert scratch == 0 and occ == 2, (frag, scratch, occ)


# ---------------------------------------------------------------------------------------------
# host-side launch diet: per-step zero arena, lazily counted BatchNorm steps, low-precision weight copies
# ---------------------------------------------------------------------------------------------
def test_zero_arena_hands_out_zeroed_aligned_views_and_falls_back():
    from matdeeplearn_amd import ops
    cpu = torch.device("cpu")
    assert ops._ARENA is None
    t = ops._zeros_step((3, 5), cpu)                      # outside a step: plain zeros
    assert t.shape == (3, 5) and float(t.abs().sum()) == 0.0
    with ops.zero_arena(cpu, nbytes=4096) as arena:       # 1024 floats
        a = ops._zeros_step((2, 50), cpu)
        b = ops._zeros_step((7,), cpu)
        assert a.data_ptr() != b.data_ptr() and (b.data_ptr() - a.data_ptr()) % 256 == 0
        assert a.untyped_storage().data_ptr() == arena.buf.untyped_storage().data_ptr()
        a.fill_(3.0)
        b.fill_(4.0)
        big = ops._zeros_step((2000,), cpu)               # does not fit: falls back to a fresh tensor
        assert big.untyped_storage().data_ptr() != arena.buf.untyped_storage().data_ptr() and float(big.sum()) == 0.0
    assert ops._ARENA is None
    with ops.zero_arena(cpu, nbytes=4096):                # next step: same buffer, zero again
        c = ops._zeros_step((2, 50), cpu)
        assert c.data_ptr() == a.data_ptr() and float(c.abs().sum()) == 0.0


def test_batchnorm_step_counter_is_folded_in_when_the_state_is_read():
    from matdeeplearn_amd import nn as mnn
    bn = mnn.BatchNorm1d(8)
    bn.train()
    bn(torch.randn(16, 8))                                # CPU: library path, counts on the tensor
    assert int(bn.num_batches_tracked) == 1
    bn._nbt_pending = 3                                   # what three HIP-path steps leave behind
    assert int(bn.state_dict()["num_batches_tracked"]) == 4 and bn._nbt_pending == 0
    other = mnn.BatchNorm1d(8)
    other._nbt_pending = 5
    other.load_state_dict(bn.state_dict())
    assert other._nbt_pending == 0 and int(other.state_dict()["num_batches_tracked"]) == 4


def test_low_precision_weight_copies_are_dropped_when_the_parameters_move_on():
    from matdeeplearn_amd.models import _base
    lin = torch.nn.Linear(6, 4)
    assert _base._lowp(lin) is None
    w16, b16 = lin.weight.detach().to(torch.bfloat16), lin.bias.detach().to(torch.bfloat16)
    lin._mdl_lowp = (w16, b16, lin.weight._version, lin.bias._version)
    assert _base._lowp(lin)[0] is w16
    with torch.no_grad():
        lin.weight.add_(1.0)                              # an optimizer step
    assert _base._lowp(lin) is None


def test_linear_act_composes_the_library_ops_off_the_fast_path():
    from matdeeplearn_amd import ops
    g = torch.Generator().manual_seed(0)
    x, w, b = torch.randn(9, 6, generator=g), torch.randn(4, 6, generator=g), torch.randn(4, generator=g)
    assert torch.equal(ops.linear_act(x, w, b, "relu"), torch.relu(torch.nn.functional.linear(x, w, b)))
    assert torch.equal(ops.linear_act(x, w, None, None), torch.nn.functional.linear(x, w))
    assert torch.equal(ops.linear_act(x, w, b, "softplus"), torch.nn.functional.softplus(torch.nn.functional.linear(x, w, b)))


# ---------------------------------------------------------------------------------------------
# product graph builder vs the loop oracle on random inputs (the goldens pin five clouds; this covers ties,
# duplicate distances, tiny graphs, every pbc combination)
# ---------------------------------------------------------------------------------------------
def test_product_graph_builder_matches_loop_oracle_on_random_clouds():
    from hypothesis import given, settings, strategies as st
    from matdeeplearn_amd.process import graph as pg
    from oracle import graph as og

    @settings(max_examples=40, deadline=None, derandomize=True)
    @given(n=st.integers(1, 24), seed=st.integers(0, 10 ** 6), radius=st.sampled_from([2.5, 4.0, 8.0]),
           k=st.sampled_from([1, 4, 12]), grid=st.booleans(), pbc=st.tuples(st.booleans(), st.booleans(), st.booleans()))
    def check(n, seed, radius, k, grid, pbc):
        rng = np.random.default_rng(seed)
        cell = np.diag(rng.uniform(5.0, 9.0, size=3))
        pos = rng.uniform(0.0, 9.0, size=(n, 3))
        if grid:                                             # lattice points: many exactly equal distances (rank ties)
            pos = np.round(pos / 1.5) * 1.5
        d_p = pg.distance_matrix(pos, cell, list(pbc))
        d_o = og.mic_distances(pos, cell, list(pbc))
        assert np.allclose(d_p, d_o, atol=1e-12)
        t_p, t_o = pg.threshold_sort(d_o, radius, k), og.threshold_sort(d_o, radius, k)
        assert np.array_equal(t_p, t_o)
        ei_p, ew_p = pg.edges_from_trimmed(t_o)
        ei_o, ew_o = og.dense_to_edges(t_o)
        assert np.array_equal(ei_p, ei_o) and np.array_equal(ew_p, ew_o)
        assert np.array_equal(pg.one_hot_degree(ei_o, n, k + 1), og.one_hot_degree(ei_o, n, k + 1))

    check()


def test_written_out_gru_step_is_torch_gru():
    """models.mpnn.gru_step (the one-step GRU the product MPNN runs instead of the library's RNN path, mpnn.py:160-161)
    against torch.nn.GRU on the same parameters: output and every gradient, fp32 on the CPU."""
    from matdeeplearn_amd.models.mpnn import gru_step
    torch.manual_seed(0)
    c, n = 12, 37
    gru = torch.nn.GRU(c, c)
    x = torch.randn(n, c, requires_grad=True)
    h = torch.randn(n, c, requires_grad=True)
    out_ref, h_ref = gru(x.unsqueeze(0), h.unsqueeze(0))
    assert torch.equal(out_ref.squeeze(0), h_ref.squeeze(0))
    w = torch.randn(n, c)
    g_ref = torch.autograd.grad((h_ref.squeeze(0) * w).sum(), [x, h] + list(gru.parameters()))
    out, out_cd = gru_step(gru, x, h)                  # (the state, and the state in the compute dtype: the same values in fp32)
    assert torch.equal(out, out_cd)
    g = torch.autograd.grad((out * w).sum(), [x, h] + list(gru.parameters()))
    assert torch.allclose(out, h_ref.squeeze(0), rtol=1e-5, atol=1e-6)
    for a, b in zip(g, g_ref):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


def test_static_capacity_covers_random_batches(small_ds):
    """process.static_capacity: (n_cap, e_cap) of the padded static batch = mean + 6 sigma of a random batch, rounded up to
    the quantum — every one of 200 random batches of the dataset fits."""
    from matdeeplearn_amd.process import static_capacity
    ds, B = small_ds, 16
    n_cap, e_cap = static_capacity(ds, B, quantum=64)
    assert n_cap % 64 == 0 and e_cap % 64 == 0
    rng = np.random.default_rng(0)
    nn, ne = np.diff(ds.node_ptr), np.diff(ds.edge_ptr)
    for _ in range(200):
        ids = rng.choice(len(ds), size=B, replace=False)
        assert nn[ids].sum() < n_cap and ne[ids].sum() <= e_cap
    assert n_cap < 4 * B * nn.mean() and e_cap < 4 * B * ne.mean()          # ... without being absurdly large


def test_loss_and_shape_predicates_off_the_hip_path():
    """ops.loss on CPU tensors is torch's loss; the shape predicates that route dense layers to the HIP kernels."""
    from matdeeplearn_amd import ops
    p, y = torch.randn(9, requires_grad=True), torch.randn(9)
    assert torch.equal(ops.loss("l1_loss", p, y), torch.nn.functional.l1_loss(p, y))
    assert torch.equal(ops.loss("smooth_l1_loss", p, y), torch.nn.functional.smooth_l1_loss(p, y))
    assert ops._hip_shape_ok(150, 150) and ops._hip_shape_ok(100, 256) and not ops._hip_shape_ok(150, 256)
    assert not ops._hip_shape_ok(161, 100) and not ops._hip_shape_ok(100, 51)
    x = torch.zeros(4, 300)
    assert ops._tn_split_ok(x, torch.zeros(100, 300)) and not ops._tn_split_ok(x, torch.zeros(130, 300))
    assert not ops._tn_split_ok(torch.zeros(4, 600), torch.zeros(100, 600))


def test_gcnconv_consumes_two_glorot_draws_like_pyg():
    """PyG 2.0.1's GCNConv builds its weight with PyG's own Linear (allocated without a draw, glorot once in the ctor) and ends
    its constructor with reset_parameters() (glorot again): TWO draws.  One more (torch's nn.Linear default init) would shift
    the seeded initial weights of every layer created after a GCNConv (gcn.py:77-87 creates conv_i, bn_i, ... in a loop)."""
    from matdeeplearn_amd import nn as pnn
    from oracle import ops as oops
    for cls in (pnn.GCNConv, oops.GCNConv):
        torch.manual_seed(5)
        conv = cls(7, 5, improved=True, add_self_loops=False)
        after = torch.rand(3)
        torch.manual_seed(5)
        w = torch.empty(5, 7)
        torch.nn.init.xavier_uniform_(w)
        torch.nn.init.xavier_uniform_(w)
        assert torch.equal(conv.lin.weight.detach(), w) and torch.equal(after, torch.rand(3)), cls


def test_trial_checkpoints_carry_the_reference_names():
    """training.py:744 (repeat: "<i>_<model_path>") and :1086-1088 (ensemble: "<i>_<model name>_<model_path>")"""
    from matdeeplearn_amd.training import driver
    assert driver._trial_path("my_model.pth", 3) == "3_my_model.pth"
    assert driver._trial_path("my_model.pth", 0, "CGCNN_demo") == "0_CGCNN_demo_my_model.pth"
    assert driver._trial_path("out/m.pth", 2, "SchNet") == "out/2_SchNet_m.pth"
    assert driver._member_name({"ensemble_list": ["CGCNN_demo", "MPNN_demo"]}, [{}, {}], 1) == "MPNN_demo"
    assert driver._member_name({}, [{"model": "GCN"}], 0) == "GCN"


def test_sequential_chains_fuse_each_dense_layer_with_its_activation(monkeypatch):
    """nn._seq: every Linear of a Sequential goes to ops.linear_act together with the activation module that follows it (one
    fused dense layer per pair); other modules are applied as they are.  Between two layers that both run as fused dense
    layers the activation derivative is handed down the chain: the earlier layer is told `out_pre` (its gradient arrives as a
    pre-activation gradient), the later one `in_act` = the earlier layer's activation — and only then."""
    from matdeeplearn_amd import nn as mnn
    calls = []

    def fake_linear_act(h, weight, bias, act, lowp=None, in_act=None, out_pre=False, pre=None):
        calls.append((tuple(weight.shape), act, in_act, out_pre))
        return torch.zeros(h.shape[0], weight.shape[0], dtype=h.dtype)
    monkeypatch.setattr(mnn.ops, "linear_act", fake_linear_act)
    h = torch.zeros(4, 6, dtype=torch.bfloat16)                       # bf16 rows, fp32 master weights: the ops path
    seq = torch.nn.Sequential(torch.nn.Linear(6, 10), mnn.ShiftedSoftplus(), torch.nn.Linear(10, 8))
    mnn._seq(seq, h)                                                  # (CPU rows: no layer is fused, nothing is handed over)
    assert calls == [((10, 6), "ssp", None, False), ((8, 10), None, None, False)]
    calls.clear()
    seq5 = torch.nn.Sequential(torch.nn.Linear(6, 10), torch.nn.ReLU(), torch.nn.Linear(10, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    mnn._seq(seq5, h)
    assert [c[:2] for c in calls] == [((10, 6), "relu"), ((8, 10), "relu"), ((3, 8), None)]
    calls.clear()
    mnn._seq(torch.nn.Sequential(torch.nn.Linear(6, 10), torch.nn.Linear(10, 8)), h)
    assert [c[:2] for c in calls] == [((10, 6), None), ((8, 10), None)]
    # every layer fused (as on bf16 device rows): ssp -> relu -> none chain of even widths, 2048 rows
    calls.clear()
    monkeypatch.setattr(mnn.ops, "linear_act_fused_ok", lambda x, w, act: True)
    big = torch.zeros(2048, 50, dtype=torch.bfloat16)
    chain = torch.nn.Sequential(torch.nn.Linear(50, 150), mnn.ShiftedSoftplus(), torch.nn.Linear(150, 150), torch.nn.ReLU(),
                                torch.nn.Linear(150, 64))
    mnn._seq(chain, big)
    assert calls == [((150, 50), "ssp", None, True), ((150, 150), "relu", "ssp", True), ((64, 150), None, "relu", False)]
    # a module between two Linears ends the hand-over; a last activated layer keeps its own derivative
    calls.clear()
    broken = torch.nn.Sequential(torch.nn.Linear(50, 150), torch.nn.ReLU(), torch.nn.Dropout(0.0), torch.nn.Linear(150, 64), torch.nn.ReLU())
    mnn._seq(broken, big)
    assert calls == [((150, 50), "relu", None, False), ((64, 150), "relu", None, False)]
    # without autograd nothing is handed over
    calls.clear()
    with torch.no_grad():
        mnn._seq(chain, big)
    assert [c[2:] for c in calls] == [(None, False)] * 3
    # outputs already formed by a fused forward (K4, ops.cfconv_fused) travel to the layers in order — the hand-over is unchanged
    seen = []

    def fake_with_pre(h, weight, bias, act, lowp=None, in_act=None, out_pre=False, pre=None):
        seen.append((tuple(weight.shape), act, in_act, out_pre, None if pre is None else tuple(pre.shape)))
        return torch.zeros(h.shape[0], weight.shape[0], dtype=h.dtype) if pre is None else pre
    monkeypatch.setattr(mnn.ops, "linear_act", fake_with_pre)
    filt = torch.nn.Sequential(torch.nn.Linear(50, 150), mnn.ShiftedSoftplus(), torch.nn.Linear(150, 150))
    a1, w = torch.ones(2048, 150, dtype=torch.bfloat16), torch.full((2048, 150), 2.0, dtype=torch.bfloat16)
    out = mnn._seq(filt, big, pre=[a1, w])
    assert seen == [((150, 50), "ssp", None, True, (2048, 150)), ((150, 150), None, "ssp", False, (2048, 150))] and out is w


def test_k4_unit_order_of_the_packed_weight_rows():
    """csrc/cfconv.hip packs row rho of every 32-row weight block with unit pi(rho) = rho with bits 2 and 3 exchanged.  What the
    kernel relies on (restated here in integers): accumulator register r of lane half h sits in MFMA row (r & 3) + 8 (r >> 2) + 4 h
    and therefore holds unit 16 (r >> 3) + 8 h + (r & 7) of the block — a lane half owns two runs of eight consecutive units
    (its 16-byte pieces of an h row), the packed registers of a block are k-slots 8 h .. 8 h + 7 of two fragments in NATURAL unit
    order (W2p needs no column permutation), 8-byte chunk q (registers 4 q .. 4 q + 3) is row 4 (q >> 1) + 2 h + (q & 1) of the
    chunk buffer = units 4 row .. 4 row + 3, and the layer-2 bias slot (unit 159) is register 15 of lane half 1 of the last block."""
    pi = lambda r: (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)
    d_row = lambda r, h: (r & 3) + 8 * (r >> 2) + 4 * h
    assert sorted(pi(r) for r in range(160)) == list(range(160)) and all(pi(pi(r)) == r for r in range(160))
    for b in range(5):
        for h in (0, 1):
            units = [pi(32 * b + d_row(r, h)) for r in range(16)]
            assert units == [32 * b + 16 * (r >> 3) + 8 * h + (r & 7) for r in range(16)]
            for t in (0, 1):                      # fragment 2 b + t, k-slot s of this lane half = position 16 (2 b + t) + 8 h + s
                assert [units[8 * t + s] for s in range(8)] == [16 * (2 * b + t) + 8 * h + s for s in range(8)]
            for q in range(4):                    # chunk q -> chunk-buffer row (in unit order)
                row = 4 * (q >> 1) + 2 * h + (q & 1)
                assert units[4 * q:4 * q + 4] == [32 * b + 4 * row + j for j in range(4)]
    assert pi(32 * 4 + d_row(15, 1)) == 159


def test_dense_layer_dispatch_predicates():
    """The shape / dtype predicates that route a dense layer to the fused HIP forms (pure host logic): the one-pass backward
    takes even widths in [34, 160]; the Linear -> ReLU -> BatchNorm node additionally wants bf16 device rows with a gradient;
    the transposed TN path takes many-output / few-input Linears (the GRU gate matrices); nothing of this applies to CPU rows."""
    from matdeeplearn_amd import ops

    class Ctx:
        def __init__(self, M, K, bias=True, need=True):
            self.shape, self.has_bias, self.needs_input_grad = (M, K), bias, (need,)
    g = torch.zeros(2048, 100, dtype=torch.bfloat16)
    x = torch.zeros(2048, 100, dtype=torch.bfloat16)
    w = torch.zeros(100, 100, dtype=torch.bfloat16)
    assert not ops._dense_bwd_ok(Ctx(100, 100), g, x, w)                          # CPU tensors: never
    w300 = torch.zeros(300, 100)
    assert ops._tn_wide_out_ok(x, w300) and not ops._tn_wide_out_ok(x, torch.zeros(100, 100))
    assert not ops._tn_wide_out_ok(x, torch.zeros(300, 200)) and not ops._tn_wide_out_ok(x, torch.zeros(301, 100))
    xr = torch.zeros(2048, 100, dtype=torch.bfloat16, requires_grad=True)
    wp = torch.zeros(100, 100, requires_grad=True)
    assert not ops.linear_relu_bn_ok(xr, wp, True)                                # CPU rows
    assert not ops.linear_act_fused_ok(xr, wp, "relu")
    assert ops._hip_shape_ok(150, 150) and ops._hip_shape_ok(128, 256) and not ops._hip_shape_ok(150, 200) and not ops._hip_shape_ok(64, 51)


def test_wide_matmul_falls_back_to_the_library_off_device():
    """ops.matmul_wide only takes the streaming kernel for bf16 rows on a HIP device; anything else is `x @ w`."""
    from matdeeplearn_amd import ops
    x, w = torch.randn(5, 6), torch.randn(6, 700)
    assert torch.equal(ops.matmul_wide(x, w), x @ w)


def test_two_chunk_exchange_split_and_hook_rearm_in_process(tmp_path):
    """FlatDataParallel at world size 1 (gloo, forced; chunk_bytes is opt-in): the first step observes the order in which the
    gradients become ready, the flat buffer is re-laid out in that order and split where it reaches half its size; from then on
    the first-ready chunk's all-reduce is started by the gradient hook of its LAST member, zero_grad() re-arms it, a member
    without a gradient delays its chunk to reduce_grads() (still two collectives, zeros packed), the exchange at world size 1
    is the identity BIT FOR BIT, and every .grad ends as a view of the flat buffer."""
    import torch.distributed as dist
    from matdeeplearn_amd.training import FlatDataParallel
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this interpreter")
    dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "pg"), rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Linear(8, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))
        assert FlatDataParallel(m, force=True).split is None                  # off by default
        dp = FlatDataParallel(m, force=True, chunk_bytes=256)
        assert dp.split == "observe"
        x = torch.randn(16, 8)
        dp.zero_grad()
        m(x).sum().backward()
        assert dp._early_work is None
        ref = {id(p): p.grad.clone() for p in m.parameters()}
        dp.reduce_grads()
        sizes = [p.numel() for p in dp.params]
        k, off = dp.split
        assert off == sum(sizes[:k]) and off <= sum(sizes) // 2 < off + sizes[k]
        # ready order = backward order: the last layer first, its bias before / after its weight as autograd delivers them
        assert {id(p) for p in dp.params[:2]} == {id(m[4].weight), id(m[4].bias)}
        assert {id(p) for p in dp.params[-2:]} == {id(m[0].weight), id(m[0].bias)}
        for p, v in zip(dp.params, dp.views):
            assert torch.equal(p.grad, ref[id(p)]) and p.grad.data_ptr() == v.data_ptr()
        calls = []
        real = dist.all_reduce

        def counting(t, *a, **kw):
            calls.append((t.data_ptr() - dp.flat_grad.data_ptr()) // 4)
            return real(t, *a, **kw)
        dist.all_reduce = counting
        try:
            for _ in range(2):
                calls.clear()
                dp.zero_grad()
                assert dp._early_left == k and dp._early_work is None
                m(x).sum().backward()
                assert dp._early_work is not None and calls == [0]     # started from the hook, before reduce_grads()
                ref = {id(p): p.grad.clone() for p in m.parameters()}
                dp.reduce_grads()
                assert dp._early_work is None and calls == [0, off]
                for p, v in zip(dp.params, dp.views):
                    assert torch.equal(p.grad, ref[id(p)]) and p.grad.data_ptr() == v.data_ptr()
            # a member of the first-ready chunk without a gradient: its chunk leaves from reduce_grads() — still two collectives
            calls.clear()
            dp.zero_grad()
            h = m[2](torch.relu(m[0](x)))
            h.sum().backward()                                          # m[4] unused
            assert dp._early_work is None and calls == []
            dp.reduce_grads()
            assert calls == [0, off]
            assert all(float(v.abs().sum()) == 0.0 for p, v in zip(dp.params, dp.views) if p is m[4].weight or p is m[4].bias)
        finally:
            dist.all_reduce = real
    finally:
        dist.destroy_process_group()


def test_two_chunk_exchange_survives_accumulation_and_foreign_zero_grad(tmp_path):
    """(advisor, round 4) Two backward passes before the first reduce_grads() — gradient accumulation, a warm-up backward without
    dp.zero_grad() — fire every observe hook twice: the ready order must still be a permutation of the parameters.  And in
    steady state the early chunk's countdown re-arms when the exchange finishes, so a loop that clears gradients with the
    optimizer's zero_grad() (not dp.zero_grad()) keeps the hook-started first collective."""
    import torch.distributed as dist
    from matdeeplearn_amd.training import FlatDataParallel
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this interpreter")
    dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "pg"), rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Linear(8, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))
        dp = FlatDataParallel(m, force=True, chunk_bytes=256)
        x = torch.randn(16, 8)
        dp.zero_grad()
        m(x).sum().backward()
        m(x).sum().backward()                                   # accumulation: every hook has fired twice
        assert len(dp._ready) == 2 * len(dp.params)
        ref = {id(p): p.grad.clone() for p in m.parameters()}
        dp.reduce_grads()
        assert isinstance(dp.split, tuple) and sorted(map(id, dp.params)) == sorted(map(id, m.parameters()))
        assert sum(v.numel() for v in dp.views) == dp.flat_grad.numel()
        for p in m.parameters():
            assert torch.equal(p.grad, ref[id(p)])
        opt = torch.optim.SGD(m.parameters(), lr=0.0)
        calls = []
        real = dist.all_reduce

        def counting(t, *a, **kw):
            calls.append((t.data_ptr() - dp.flat_grad.data_ptr()) // 4)
            return real(t, *a, **kw)
        dist.all_reduce = counting
        try:
            for _ in range(3):
                calls.clear()
                opt.zero_grad(set_to_none=True)                 # NOT dp.zero_grad()
                m(x).sum().backward()
                assert calls == [0], "the first-ready chunk leaves from its hook without dp.zero_grad()"
                dp.reduce_grads()
                assert calls == [0, dp.split[1]]
            # (advisor, round 5) accumulation in the two-chunk steady state: the early chunk leaves after the FIRST backward with
            # that micro-batch's gradients only; the exchange must re-pack and re-reduce it from the accumulated gradients
            calls.clear()
            opt.zero_grad(set_to_none=True)
            m(x).sum().backward()
            m(2 * x).sum().backward()
            want = {id(p): p.grad.clone() for p in m.parameters()}
            dp.reduce_grads()
            assert calls == [0, 0, dp.split[1]], calls
            for p in m.parameters():
                assert torch.equal(p.grad, want[id(p)]), "second micro-batch's gradient of an early parameter was lost"
            calls.clear()                                       # ... and the step after it is an ordinary two-collective step
            opt.zero_grad(set_to_none=True)
            m(x).sum().backward()
            dp.reduce_grads()
            assert calls == [0, dp.split[1]]
        finally:
            dist.all_reduce = real
    finally:
        dist.destroy_process_group()


def test_checkpointed_optimizer_state_holds_float_learning_rates():
    """A tensor learning rate (what make_optimizer(capturable=True) builds on a device) must not reach the checkpoint: the
    reference format holds floats, and loading a tensor lr into a non-capturable optimizer raises.  Loading back into an
    optimizer that keeps a tensor lr leaves ITS tensor in place (a captured graph holds its address) with the loaded value."""
    from matdeeplearn_amd.training import load_optimizer_state, optimizer_state_for_checkpoint
    m = torch.nn.Linear(3, 2)
    lr_t = torch.tensor(0.004)
    opt = torch.optim.AdamW(m.parameters(), lr=lr_t, capturable=False, foreach=False)
    m(torch.randn(5, 3)).sum().backward()
    opt.step()
    sd = optimizer_state_for_checkpoint(opt)
    assert isinstance(sd["param_groups"][0]["lr"], float) and abs(sd["param_groups"][0]["lr"] - 0.004) < 1e-9
    assert opt.param_groups[0]["lr"] is lr_t                       # the live optimizer is untouched
    plain = torch.optim.AdamW(m.parameters(), lr=0.1)
    plain.load_state_dict(sd)                                      # what failed with a tensor lr in the checkpoint
    assert abs(plain.param_groups[0]["lr"] - 0.004) < 1e-9
    sd["param_groups"][0]["lr"] = 0.001
    load_optimizer_state(opt, sd)
    assert opt.param_groups[0]["lr"] is lr_t and abs(float(lr_t) - 0.001) < 1e-9


def test_struct_entry_points_have_the_layout_the_header_declares(tmp_path):
    """MdlCgConv / MdlCgNode: the ctypes mirrors in _lib.py against sizeof / offsetof of include/mdl_hip.h as gcc lays them out
    (a field added on one side only would shift every pointer behind it)."""
    import ctypes
    import shutil
    import subprocess
    from matdeeplearn_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    lines = []
    for cls in (_lib.MdlCgConv, _lib.MdlCgNode):
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cls.__name__, cls.__name__))
        for name, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cls.__name__, name, cls.__name__, name))
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mdl_hip.h"\nint main(void) {\n%s\nreturn 0; }\n' % "\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True, capture_output=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cls in (_lib.MdlCgConv, _lib.MdlCgNode):
        assert int(got[cls.__name__]) == ctypes.sizeof(cls)
        for name, _ in cls._fields_:
            assert int(got["%s.%s" % (cls.__name__, name)]) == getattr(cls, name).offset, (cls.__name__, name)
    hdr = open(os.path.join(ROOT, "include", "mdl_hip.h")).read()
    for gone in ("mdl_cgconv_bwd_h", "mdl_cgconv_bwd_hb", "mdl_cgconv_bwd_node_z", "mdl_cgconv_bwd_node_h"):
        assert not re.search(r"\b%s\s*\(" % gone, hdr), gone


def _bench_plumbing_worker(rank, world, port, out_path):
    """bench.py's own N > 1 branch — settle phase (MIN reduce), timed region (MAX of the times, SUM of the edges), strong-scaling
    leg — on a world_size-2 gloo group with CPU tensors behind a stub step: the first 8-GPU run of the driver cannot die in the
    bench's reductions (training.py:227-237, 291-294 is the reference's multi-process launch these stand beside)."""
    import importlib.util
    import time
    import torch.distributed as dist
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    barrier = bench.make_barrier(dev, True)
    calls = []

    def step(ids, timed, next_ids=None):                       # rank 1 is the slow rank; every batch "has" 10 edges per graph
        calls.append((int(ids[0]), bool(timed), None if next_ids is None else int(next_ids[0])))
        time.sleep(0.002 * (1 + rank))
        return 10 * len(ids) + rank, len(ids)

    def stream():
        k = 0
        while True:
            yield np.arange(k, k + 4)
            k += 4

    n, groups = bench.settle_phase(step, stream(), dev, True, 0.05, 0.5)
    assert n >= 8 and n % 8 == 0 and len(groups) == n // 8
    ns = [None, None]
    dist.all_gather_object(ns, n)
    assert ns[0] == ns[1], "ranks must leave the settle phase after the same number of steps (MIN-reduced clock and flag)"
    assert all(c[2] is not None for c in calls), "settle steps assemble the next batch ahead, like the timed ones"
    calls.clear()

    W, K = 2, 9
    ids = [np.arange(100 * i, 100 * i + 4) for i in range(W + K)]
    tr = bench.timed_region(step, ids, W, K, dev, True, 4, barrier)
    assert len(calls) == W + K and [c[0] for c in calls] == [100 * i for i in range(W + K)]
    assert [c[1] for c in calls[:W]] == [False] * W
    assert [c[1] for c in calls[W:]] == [k % 4 == 1 for k in range(K)], "event steps: every 4th, never the first timed step"
    assert calls[-1][2] == 0 and calls[W - 1][2] == 100 * W    # the K timed steps contain K assemblies
    assert tr["edges"] == K * (40 + rank) and tr["edges_all"] == K * (40 + 40 + 1)
    assert tr["ev_steps"] == 2 and len(tr["by4"]) == 3 and len(tr["host_enqueue_ms"]) == K
    both = [None, None]
    dist.all_gather_object(both, (tr["elapsed"], tr["elapsed_max"]))
    assert both[0][1] == both[1][1] == max(both[0][0], both[1][0]) and both[1][0] >= K * 0.004
    calls.clear()
    tr2 = bench.timed_region(step, ids, W, K, dev, True, 1, barrier, run_in=1)
    assert tr2["run_in"] == 1 and len(calls) == W + K and all(c[1] for c in calls[W:])

    s_ids = [np.arange(7 * i, 7 * i + 2) for i in range(W + 3)]
    st = bench.strong_leg(step, s_ids, W, 3, 2, world, dev, True, barrier)
    assert st["scaling"] == "strong" and st["global_batch_graphs"] == 4 and st["steps"] == 3
    assert abs(st["value"] * st["ms_per_step"] * 1e-3 * 3 - 3 * (20 + 21)) < 0.5     # SUM of edges / MAX of times
    assert st["ms_per_step"] >= 4.0                                                   # the slow rank's 4 ms per step
    r0, r1, r2 = bench.all_reduce_scalars([rank + 1.0, 5.0], "sum", dev, True), bench.all_reduce_scalars([rank], "max", dev, True), \
        bench.all_reduce_scalars([rank], "min", dev, True)
    assert r0 == [3.0, 10.0] and r1 == [1.0] and r2 == [0.0]
    if rank == 0:
        open(out_path, "w").write("ok")
    dist.destroy_process_group()


def test_bench_multi_gpu_plumbing_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = 33500 + (os.getpid() % 2000)
    out = str(tmp_path / "bench.txt")
    mp.spawn(_bench_plumbing_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_package_reads_no_mode_environment():
    """No import-time (or any) A/B switch behind an environment variable in the product: the only names the package may read are
    the rendezvous variables of torch.distributed and the build's HIPCC; dispatch options go through ops.configure()."""
    from matdeeplearn_amd import ops
    allowed = {"MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK", "HIPCC"}
    pkg = os.path.join(ROOT, "matdeeplearn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                for name in re.findall(r"os\.environ(?:\.get|\.setdefault)?\(\s*[\"']([A-Za-z0-9_]+)[\"']", src) + \
                        re.findall(r"os\.environ\[\s*[\"']([A-Za-z0-9_]+)[\"']", src) + re.findall(r"getenv\(\s*[\"']([A-Za-z0-9_]+)", src):
                    assert name in allowed, "%s reads the environment variable %s" % (f, name)
    prev = ops.configure(rsrc16=False, balance=False)
    try:
        assert prev == {"rsrc16": True, "balance": True} and ops.options()["rsrc16"] is False and ops._BALANCE is False
    finally:
        ops.configure(**prev)
    assert ops.options()["rsrc16"] is True
    with pytest.raises(ops.MdlError):
        ops.configure(no_such_option=True)
