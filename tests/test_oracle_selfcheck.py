"""Self-consistency pins for the oracle operators whose reference arithmetic lives in PyG 2.0.1
(absent here, 'parity unpinned'): independent fp64 re-derivations with python loops / dense
adjacency, permutation invariance, and autograd gradcheck."""
import math

import pytest
import torch

from oracle import ops as oops

torch.set_default_dtype(torch.float32)


def _graph(n=9, seed=0, loops=True):
    g = torch.Generator().manual_seed(seed)
    src, tgt = [], []
    for i in range(n):
        k = int(torch.randint(0, 4, (1,), generator=g))
        for j in torch.randperm(n, generator=g)[:k].tolist():
            if j != i:
                src.append(j); tgt.append(i)
        if loops:
            src.append(i); tgt.append(i)
    return torch.tensor([src, tgt])


def test_cgconv_matches_dense_loop_fp64_and_is_permutation_invariant():
    g = torch.Generator().manual_seed(1)
    n, C, G = 9, 5, 4
    ei = _graph(n, 1)
    E = ei.shape[1]
    x = torch.randn(n, C, generator=g, dtype=torch.float64)
    ea = torch.rand(E, G, generator=g, dtype=torch.float64)
    wf, ws = torch.randn(C, 2 * C + G, generator=g, dtype=torch.float64), torch.randn(C, 2 * C + G, generator=g, dtype=torch.float64)
    bf, bs = torch.randn(C, generator=g, dtype=torch.float64), torch.randn(C, generator=g, dtype=torch.float64)
    a = oops.cgconv(x, ei, ea, wf, bf, ws, bs, "mean")
    b = oops.cgconv_dense(x, ei, ea, wf, bf, ws, bs)
    assert torch.allclose(a, b, rtol=1e-12, atol=1e-12)
    perm = torch.randperm(E, generator=g)
    assert torch.allclose(oops.cgconv(x, ei[:, perm], ea[perm], wf, bf, ws, bs, "mean"), a, rtol=1e-12, atol=1e-12)
    # node 0 with no incoming edge keeps x (mean over nothing = 0)
    ei2 = ei[:, ei[1] != 0]
    assert torch.allclose(oops.cgconv(x, ei2, ea[ei[1] != 0], wf, bf, ws, bs, "mean")[0], x[0])


def test_cgconv_gradcheck():
    g = torch.Generator().manual_seed(2)
    n, C, G = 6, 3, 2
    ei = _graph(n, 2)
    x = torch.randn(n, C, generator=g, dtype=torch.float64, requires_grad=True)
    ea = torch.rand(ei.shape[1], G, generator=g, dtype=torch.float64)
    wf = torch.randn(C, 2 * C + G, generator=g, dtype=torch.float64, requires_grad=True)
    ws = torch.randn(C, 2 * C + G, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(C, generator=g, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda x_, wf_, ws_, b_: oops.cgconv(x_, ei, ea, wf_, b_, ws_, b_, "mean"), (x, wf, ws, b))


def test_scatter_semantics():
    src = torch.tensor([[1.0, -2.0], [3.0, 4.0], [5.0, 6.0], [-1.0, 0.5]])
    idx = torch.tensor([2, 0, 2, 0])
    assert torch.equal(oops.scatter(src, idx, 0, 4, "sum"), torch.tensor([[2.0, 4.5], [0, 0], [6.0, 4.0], [0, 0]]))
    assert torch.equal(oops.scatter(src, idx, 0, 4, "mean"), torch.tensor([[1.0, 2.25], [0, 0], [3.0, 2.0], [0, 0]]))
    assert torch.equal(oops.scatter(src, idx, 0, None, "max"), torch.tensor([[3.0, 4.0], [0, 0], [5.0, 6.0]]))
    assert oops.scatter_mean(src, idx, 0).shape[0] == 3      # dim_size defaults to index.max()+1 (megnet.py:86)


def test_interaction_block_matches_per_edge_loop():
    torch.manual_seed(3)
    n, H, G, Fn, cutoff = 7, 6, 5, 8, 8.0
    ei = _graph(n, 3)
    blk = oops.InteractionBlock(H, G, Fn, cutoff).double()
    x = torch.randn(n, H, dtype=torch.float64)
    ew = torch.rand(ei.shape[1], dtype=torch.float64) * 8
    ea = torch.rand(ei.shape[1], G, dtype=torch.float64)
    got = blk(x, ei, ew, ea)
    ssp = lambda t: torch.nn.functional.softplus(t) - math.log(2.0)
    h = x @ blk.conv.lin1.weight.T
    agg = torch.zeros(n, Fn, dtype=torch.float64)
    for e in range(ei.shape[1]):
        j, i = int(ei[0, e]), int(ei[1, e])
        w = blk.mlp[2](ssp(blk.mlp[0](ea[e]))) * 0.5 * (math.cos(float(ew[e]) * math.pi / cutoff) + 1.0)
        agg[i] += h[j] * w
    ref = blk.lin(ssp(blk.conv.lin2(agg)))
    assert torch.allclose(got, ref, rtol=1e-10, atol=1e-12)
    assert blk.conv.nn is blk.mlp                      # `conv.nn.*` aliases `mlp.*` in the state_dict (A.3)


def test_nnconv_and_gcnconv_match_dense_forms():
    torch.manual_seed(4)
    n, C, G = 6, 4, 3
    ei = _graph(n, 4)
    x = torch.randn(n, C, dtype=torch.float64)
    ea = torch.rand(ei.shape[1], G, dtype=torch.float64)
    net = torch.nn.Sequential(torch.nn.Linear(G, 5), torch.nn.ReLU(), torch.nn.Linear(5, C * C)).double()
    conv = oops.NNConv(C, C, net, aggr="mean").double()
    got = conv(x, ei, ea)
    ref = x @ conv.lin.weight.T + conv.bias
    for i in range(n):
        msgs = [x[int(ei[0, e])] @ net(ea[e]).view(C, C) for e in range(ei.shape[1]) if int(ei[1, e]) == i]
        if msgs:
            ref[i] = ref[i] + torch.stack(msgs).mean(0)
    assert torch.allclose(got, ref, rtol=1e-10, atol=1e-12)
    # GCNConv: D^-1/2 A_w D^-1/2 X W with weighted in-degree, zero-weight self loops contribute nothing
    gcn = oops.GCNConv(C, C, improved=True, add_self_loops=False).double()
    w = torch.rand(ei.shape[1], dtype=torch.float64) * 5
    w[ei[0] == ei[1]] = 0.0
    A = torch.zeros(n, n, dtype=torch.float64)
    for e in range(ei.shape[1]):
        A[int(ei[1, e]), int(ei[0, e])] += w[e]
    deg = A.sum(1)
    dis = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
    ref = (dis[:, None] * A * dis[None, :]) @ (x @ gcn.lin.weight.T) + gcn.bias
    assert torch.allclose(gcn(x, ei, w), ref, rtol=1e-10, atol=1e-12)


def test_set2set_softmax_is_per_graph_and_shape():
    torch.manual_seed(5)
    s2s = oops.Set2Set(4, processing_steps=3)
    x = torch.randn(7, 4)
    batch = torch.tensor([0, 0, 0, 1, 1, 2, 2])
    out = s2s(x, batch)
    assert out.shape == (3, 8)
    # graph 1 alone gives the same row (no cross-graph leakage)
    alone = s2s(x[3:5], torch.tensor([0, 0]))
    assert torch.allclose(out[1], alone[0], atol=1e-6)


@pytest.mark.parametrize("name", ["CGCNN", "SchNet", "MEGNet", "MPNN", "GCN"])
def test_oracle_models_run_and_have_reference_state_dict_keys(name):
    import types
    from oracle import models as om
    torch.manual_seed(0)

    class DS:
        num_features, num_edge_features = 11, 6

        def __getitem__(self, i):
            return types.SimpleNamespace(y=torch.tensor(0.0), u=torch.zeros(1, 3))

    ei = torch.cat([_graph(5, 7), _graph(4, 8) + 5], 1)
    data = types.SimpleNamespace(x=torch.randn(9, 11), edge_index=ei, edge_attr=torch.rand(ei.shape[1], 6),
                                 edge_weight=torch.rand(ei.shape[1]) * 8, batch=torch.tensor([0] * 5 + [1] * 4),
                                 u=torch.zeros(2, 3))
    model = om.REGISTRY[name](DS(), dim1=8, dim2=8, dim3=8, gc_count=2, post_fc_count=1)
    out = model(data)
    assert out.shape == (2,)
    out.sum().backward()
    keys = set(model.state_dict())
    assert {"pre_lin_list.0.weight", "lin_out.weight", "post_lin_list.0.bias"} <= keys
    expect = {"CGCNN": "conv_list.0.lin_f.weight", "SchNet": "conv_list.0.conv.lin1.weight",
              "MEGNet": "conv_list.0.edge_model.edge_mlp.0.weight", "MPNN": "gru_list.0.weight_ih_l0",
              "GCN": "conv_list.0.lin.weight"}[name]
    assert expect in keys
