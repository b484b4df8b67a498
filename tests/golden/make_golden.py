#!/usr/bin/env python3
"""Regenerate the golden vectors under tests/golden/ (BUILD CONTAINER ONLY).

This script IMPORTS the reference's own Python files from /root/reference (never copies
them) behind `sys.modules` stubs for the third-party packages that are absent in this
image (ase, torch_geometric, torch_scatter), feeds them seeded inputs and stores
inputs + outputs as small .npz fixtures.  The fixtures are data; the reference source
does not travel.  Nothing under tests/ reads /root/reference at test time.

Reference entry points exercised (file:line, relative to /root/reference):
  matdeeplearn/process/process.py:580-590  GaussianSmearing        -> rbf.npz
  matdeeplearn/process/process.py:540-576  threshold_sort          -> threshold_sort.npz, pt10_graphs.npz
  matdeeplearn/process/process.py:27-79    split_data / _CV        -> splits.npz
  matdeeplearn/process/process.py:594-605  OneHotDegree            -> onehot_degree.npz
  matdeeplearn/process/process.py:626-653  GetRanges/NormalizeEdge -> normalize_edge.npz
  matdeeplearn/models/megnet.py:16-371     MEGNet + blocks         -> megnet.npz
  matdeeplearn/models/cgcnn.py:17-174, schnet.py:16-172, mpnn.py:17-188, gcn.py:17-173
                                           the four wrappers       -> wrappers.npz
      (their conv operators live in torch_geometric 2.0.1, which is absent: the oracle's restatements
       oracle.ops.{CGConv, InteractionBlock, NNConv, GCNConv, Set2Set, global_*_pool} are injected as
       the `torch_geometric.nn` stub, so these vectors pin everything the REFERENCE files own — layer
       creation order / seeded initialisation, state_dict key order, pre-FC -> conv -> BN -> act ->
       dropout ordering, pooling order, set2set sizes, residuals, the GRU wiring — not the conv
       arithmetic itself, which stays "restated from the published semantics".)
  data/test_data/test_data.tar.gz                                   -> pt10_dataset.npz (positions/targets: data)

Third-party semantics needed by the stubs (torch_scatter.scatter / scatter_mean,
torch_geometric.nn.MetaLayer, torch_geometric.utils.degree) follow SURVEY.md Appendix A.

Usage:  python tests/golden/make_golden.py
"""
import importlib.util
import io
import json
import os
import sys
import tarfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Dummy:  # placeholder base classes for the PyG dataset types
    pass


def _degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros(n, dtype=dtype or torch.float)
    return out.scatter_add_(0, index, torch.ones(index.numel(), dtype=out.dtype))


def _scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    assert dim == 0
    n = int(index.max()) + 1 if dim_size is None else dim_size
    if reduce in ("sum", "add", "mean"):
        res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype).index_add_(0, index, src)
        if reduce == "mean":
            cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
            res = res / cnt.clamp(min=1).view(-1, *([1] * (src.dim() - 1)))
        return res
    if reduce == "max":
        res = torch.full((n,) + tuple(src.shape[1:]), float("-inf"), dtype=src.dtype)
        res = res.scatter_reduce(0, index.view(-1, *([1] * (src.dim() - 1))).expand_as(src), src, "amax")
        return torch.where(torch.isinf(res), torch.zeros_like(res), res)
    raise ValueError(reduce)


def _scatter_mean(src, index, dim=0, out=None, dim_size=None):
    return _scatter(src, index, dim, out, dim_size, "mean")


class _MetaLayer(torch.nn.Module):
    """PyG 2.0.1 MetaLayer semantics."""

    def __init__(self, edge_model=None, node_model=None, global_model=None):
        super().__init__()
        self.edge_model, self.node_model, self.global_model = edge_model, node_model, global_model

    def forward(self, x, edge_index, edge_attr=None, u=None, batch=None):
        row, col = edge_index[0], edge_index[1]
        edge_attr = self.edge_model(x[row], x[col], edge_attr, u, batch if batch is None else batch[row])
        x = self.node_model(x, edge_index, edge_attr, u, batch)
        u = self.global_model(x, edge_index, edge_attr, u, batch)
        return x, edge_attr, u


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ref_process():
    _stub("ase")
    _stub("ase.io")
    sys.modules["ase"].io = sys.modules["ase.io"]
    _stub("torch_geometric")
    _stub("torch_geometric.transforms")
    _stub("torch_geometric.data", DataLoader=_Dummy, Dataset=_Dummy, Data=_Dummy, InMemoryDataset=_Dummy)
    _stub("torch_geometric.utils", dense_to_sparse=None, degree=_degree, add_self_loops=None)
    return _load(os.path.join(REF, "matdeeplearn/process/process.py"), "ref_process")


def load_ref_megnet():
    _stub("torch_scatter", scatter=_scatter, scatter_mean=_scatter_mean, scatter_add=None, scatter_max=None)
    tg = sys.modules.get("torch_geometric") or _stub("torch_geometric")
    nn = _stub(
        "torch_geometric.nn", MetaLayer=_MetaLayer, Set2Set=None,
        global_mean_pool=lambda x, b: _scatter(x, b, reduce="mean"),
        global_add_pool=lambda x, b: _scatter(x, b, reduce="sum"),
        global_max_pool=lambda x, b: _scatter(x, b, reduce="max"),
    )
    tg.nn = nn
    return _load(os.path.join(REF, "matdeeplearn/models/megnet.py"), "ref_megnet")


# --------------------------------------------------------------------------------------


def load_ref_wrapper(fname):
    """Import /root/reference/matdeeplearn/models/<fname>.py with the oracle operators standing in for
    torch_geometric.nn (PyG 2.0.1 is not installable here)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import ops as oops
    _stub("torch_scatter", scatter=_scatter, scatter_mean=_scatter_mean, scatter_add=None, scatter_max=None)
    tg = _stub("torch_geometric")
    nn = _stub("torch_geometric.nn", Set2Set=oops.Set2Set, CGConv=oops.CGConv, NNConv=oops.NNConv,
               GCNConv=oops.GCNConv, MetaLayer=_MetaLayer, global_mean_pool=oops.global_mean_pool,
               global_add_pool=oops.global_add_pool, global_max_pool=oops.global_max_pool)
    tg.nn = nn
    nn.models = _stub("torch_geometric.nn.models")
    nn.models.schnet = _stub("torch_geometric.nn.models.schnet", InteractionBlock=oops.InteractionBlock)
    return _load(os.path.join(REF, "matdeeplearn/models/%s.py" % fname), "ref_" + fname)


# wrapper configurations stored in wrappers.npz (shared with tests/test_oracle_golden.py via the npz itself)
WRAPPER_DIMS = dict(dim1=16, dim2=12, dim3=8, gc_count=2)
WRAPPER_CONFIGS = [
    ("bn", dict(pre_fc_count=1, post_fc_count=2, batch_norm="True")),
    ("nobn_max", dict(pre_fc_count=1, post_fc_count=1, batch_norm="False", pool="global_max_pool")),
    ("late_add", dict(pre_fc_count=2, post_fc_count=2, batch_norm="True", pool="global_add_pool", pool_order="late")),
    ("s2s", dict(pre_fc_count=1, post_fc_count=1, batch_norm="True", pool="set2set", batch_track_stats="False")),
    ("s2s_late", dict(pre_fc_count=1, post_fc_count=0, batch_norm="False", pool="set2set", pool_order="late")),
    ("pre0", dict(pre_fc_count=0, post_fc_count=0, batch_norm="True", act="softplus")),
]


def wrapper_goldens(bt, y, DS):
    out = {}
    meta = {}
    for fname, cls in [("cgcnn", "CGCNN"), ("schnet", "SchNet"), ("mpnn", "MPNN"), ("gcn", "GCN")]:
        mod = load_ref_wrapper(fname)
        for tag, kw in WRAPPER_CONFIGS:
            if cls == "MPNN" and kw.get("pre_fc_count") == 0:
                continue                      # gc_dim = 114 -> a 114*114-wide edge network: not a useful fixture
            key = "%s/%s" % (cls, tag)
            torch.manual_seed(4321)
            model = getattr(mod, cls)(DS(), **WRAPPER_DIMS, **kw)
            meta[key] = dict(kw=kw, sd_keys=list(model.state_dict().keys()))
            for k, v in model.state_dict().items():
                out["%s/init/%s" % (key, k)] = v.detach().numpy().copy()
            model.train()
            pred = model(bt)
            loss = torch.nn.functional.l1_loss(pred, y)
            loss.backward()
            for k, p in model.named_parameters():
                out["%s/grad/%s" % (key, k)] = p.grad.numpy() if p.grad is not None else np.zeros(0)
            for k, v in model.state_dict().items():          # buffers after one training forward
                if "running_" in k or "num_batches" in k:
                    out["%s/post/%s" % (key, k)] = v.detach().numpy().copy()
            out["%s/pred_train" % key] = pred.detach().numpy()
            model.eval()
            out["%s/pred_eval" % key] = model(bt).detach().numpy()
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    return out


def read_pt10():
    """Parse the ASE-json structures of the reference's test tarball (data, not code)."""
    tf = tarfile.open(os.path.join(REF, "data/test_data/test_data.tar.gz"))
    members = {m.name: m for m in tf.getmembers()}
    targets = io.TextIOWrapper(tf.extractfile(members["test_data/targets.csv"])).read().strip().splitlines()
    ids, ys, pos, num, cell, pbc = [], [], [], [], [], []
    for line in targets:
        sid, y = line.split(",")[0], [float(v) for v in line.split(",")[1:]]
        rec = json.load(tf.extractfile(members["test_data/%s.json" % sid]))["1"]
        ids.append(int(sid))
        ys.append(y)
        pos.append(np.array(rec["positions"]["__ndarray__"][2], dtype=np.float64).reshape(-1, 3))
        num.append(np.array(rec["numbers"]["__ndarray__"][2], dtype=np.int64))
        cell.append(np.array(rec["cell"]["array"]["__ndarray__"][2], dtype=np.float64).reshape(3, 3))
        pbc.append(np.array(rec["pbc"]["__ndarray__"][2], dtype=bool))
    return (np.array(ids), np.array(ys, dtype=np.float64), np.stack(pos), np.stack(num),
            np.stack(cell), np.stack(pbc))


def dense_to_graph(trimmed):
    """dense_to_sparse + add_self_loops(fill 0) per SURVEY T6 (row-major nonzeros, loops appended)."""
    t = torch.as_tensor(trimmed, dtype=torch.float32)
    idx = t.nonzero(as_tuple=False).t().contiguous()
    w = t[idx[0], idx[1]]
    n = t.shape[0]
    loop = torch.arange(n)
    idx = torch.cat([idx, torch.stack([loop, loop])], dim=1)
    w = torch.cat([w, torch.zeros(n)])
    return idx.numpy().astype(np.int64), w.numpy()


def main():
    torch.manual_seed(0)
    ref = load_ref_process()

    # (1) RBF ---------------------------------------------------------------------------
    g = torch.Generator().manual_seed(0)
    d = torch.cat([torch.tensor([0.0, 1e-3, 0.25, 0.5, 1.0]), torch.rand(4096, generator=g)])
    sm = ref.GaussianSmearing(0, 1, 50, 0.2)
    np.savez_compressed(os.path.join(OUT, "rbf.npz"), d=d.numpy(), out=sm(d).numpy(),
                        offset=sm.offset.numpy(), coeff=np.float64(sm.coeff))

    # (2) threshold_sort ----------------------------------------------------------------
    rng = np.random.default_rng(7)
    ts = {}
    for tag, n, box, r, k in [("a", 20, 9.0, 8.0, 12), ("b", 40, 14.0, 8.0, 12), ("c", 40, 14.0, 4.0, 12),
                              ("d", 13, 5.0, 8.0, 12), ("e", 30, 6.0, 8.0, 4)]:
        p = rng.uniform(0, box, size=(n, 3))
        D = np.linalg.norm(p[:, None, :] - p[None, :, :], axis=-1)
        ts["D_" + tag] = D
        ts["r_" + tag], ts["k_" + tag] = np.float64(r), np.int64(k)
        ts["out_" + tag] = ref.threshold_sort(D, r, k, adj=False)
    np.savez_compressed(os.path.join(OUT, "threshold_sort.npz"), **ts)

    # (3) splits ------------------------------------------------------------------------
    sp = {}
    for n, seed in [(1000, 42), (1000, 7), (46744, 42), (37, 3)]:
        tr, va, te = ref.split_data(list(range(n)), 0.8, 0.05, 0.15, seed=seed)
        sp["train_%d_%d" % (n, seed)] = np.array(tr.indices)
        sp["val_%d_%d" % (n, seed)] = np.array(va.indices)
        sp["test_%d_%d" % (n, seed)] = np.array(te.indices)
    folds = ref.split_data_CV(list(range(1000)), num_folds=5, seed=42)
    for i, f in enumerate(folds):
        sp["cv5_1000_42_fold%d" % i] = np.array(f.indices)
    np.savez_compressed(os.path.join(OUT, "splits.npz"), **sp)

    # (4) Pt10 dataset + graphs ---------------------------------------------------------
    ids, ys, pos, num, cell, pbc = read_pt10()
    np.savez_compressed(os.path.join(OUT, "pt10_dataset.npz"), ids=ids, y=ys, positions=pos,
                        numbers=num, cell=cell, pbc=pbc)
    ecount, dmin, dmax = [], np.inf, 0.0
    gg = {}
    for s in range(len(ids)):
        D = np.linalg.norm(pos[s][:, None, :] - pos[s][None, :, :], axis=-1)  # pbc all False
        trimmed = ref.threshold_sort(D, 8.0, 12, adj=False)
        ei, ew = dense_to_graph(trimmed)
        ecount.append(ei.shape[1])
        nz = ew[ew > 0]
        dmin, dmax = min(dmin, nz.min()), max(dmax, nz.max())
        if s < 8:
            gg["edge_index_%d" % s], gg["edge_weight_%d" % s] = ei, ew
    gg["edges_per_graph"] = np.array(ecount)
    gg["dist_min"], gg["dist_max"] = np.float64(dmin), np.float64(dmax)
    np.savez_compressed(os.path.join(OUT, "pt10_graphs.npz"), **gg)
    print("pt10: E/graph min %d mean %.2f max %d total %d; d %.4f..%.4f" % (
        min(ecount), np.mean(ecount), max(ecount), sum(ecount), dmin, dmax))

    # (5) OneHotDegree + NormalizeEdge --------------------------------------------------
    ns = types.SimpleNamespace
    ei = torch.as_tensor(gg["edge_index_0"])
    data = ns(edge_index=ei, x=torch.arange(10, dtype=torch.float).view(-1, 1), num_nodes=10)
    data = ref.OneHotDegree(data, 13)
    np.savez_compressed(os.path.join(OUT, "onehot_degree.npz"), edge_index=ei.numpy(), x=data.x.numpy())
    lst = [ns(edge_descriptor={"distance": torch.as_tensor(gg["edge_weight_%d" % i])}) for i in range(4)]
    ref.NormalizeEdge(lst, "distance")
    np.savez_compressed(os.path.join(OUT, "normalize_edge.npz"),
                        **{"in_%d" % i: gg["edge_weight_%d" % i] for i in range(4)},
                        **{"out_%d" % i: lst[i].edge_descriptor["distance"].numpy() for i in range(4)})

    # (6) MEGNet ------------------------------------------------------------------------
    mg = load_ref_megnet()
    torch.manual_seed(1234)
    B, F_in, G = 3, 114, 50
    sizes = [10, 10, 10]
    eis, ews = [], []
    off = 0
    for b in range(B):
        eis.append(torch.as_tensor(gg["edge_index_%d" % b]) + off)
        ews.append(torch.as_tensor(gg["edge_weight_%d" % b]))
        off += sizes[b]
    edge_index = torch.cat(eis, dim=1)
    ew = torch.cat(ews)
    edge_attr = sm(ew / 8.0)
    N = off
    x = torch.rand(N, F_in)
    batch = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes))
    u = torch.zeros(B, 3)
    y = torch.randn(B)

    class DS:
        num_features, num_edge_features = F_in, G

        def __getitem__(self, i):
            return ns(y=torch.tensor(0.0), u=torch.zeros(1, 3))

    out = {}
    for tag, kw in [("bn", dict(batch_norm="True")), ("nobn", dict(batch_norm="False")),
                    ("max", dict(batch_norm="False", pool="global_max_pool")),
                    ("late", dict(batch_norm="True", pool_order="late"))]:
        torch.manual_seed(99)
        model = mg.MEGNet(DS(), dim1=32, dim2=24, dim3=16, pre_fc_count=1, gc_count=2, gc_fc_count=1,
                          post_fc_count=2, **kw)
        model.train()
        bt = ns(x=x, edge_index=edge_index, edge_attr=edge_attr, u=u, batch=batch)
        pred = model(bt)
        loss = torch.nn.functional.l1_loss(pred, y)
        loss.backward()
        for k, v in model.state_dict().items():
            out["%s/sd/%s" % (tag, k)] = v.detach().numpy()
        for k, p in model.named_parameters():
            out["%s/grad/%s" % (tag, k)] = p.grad.numpy() if p.grad is not None else np.zeros(0)
        out["%s/pred_train" % tag] = pred.detach().numpy()
        model.eval()
        out["%s/pred_eval" % tag] = model(bt).detach().numpy()
    out.update(x=x.numpy(), edge_index=edge_index.numpy(), edge_attr=edge_attr.numpy(), u=u.numpy(),
               batch=batch.numpy(), y=y.numpy())
    np.savez_compressed(os.path.join(OUT, "megnet.npz"), **out)

    # (7) CGCNN / SchNet / MPNN / GCN wrappers (reference files, oracle operators injected) ----------
    torch.manual_seed(777)
    bt = ns(x=x, edge_index=edge_index, edge_attr=edge_attr, edge_weight=ew, u=u, batch=batch)
    wr = wrapper_goldens(bt, y, DS)
    wr.update(x=x.numpy(), edge_index=edge_index.numpy(), edge_attr=edge_attr.numpy(), edge_weight=ew.numpy(),
              batch=batch.numpy(), y=y.numpy())
    np.savez_compressed(os.path.join(OUT, "wrappers.npz"), **wr)
    print("goldens written to", OUT)


if __name__ == "__main__":
    main()
