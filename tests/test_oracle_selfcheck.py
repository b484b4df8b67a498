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


# ------------------------------------------------------------------------------------------------
# Second, independent derivations (oracle/ops.py: *_loop / *_dense, fp64 python loops written from SURVEY Appendix A, sharing
# no code with the operator classes): value AND gradient of every PyG operator the oracle restates.  Together with the
# reading of Appendix A against the published 2.0.1 sources this is the whole pin of the PyG arithmetic.
# ------------------------------------------------------------------------------------------------
def _grads(fn, params):
    for p in params:
        p.grad = None
    fn().square().sum().backward()
    return [p.grad.clone() for p in params]


def test_cfconv_second_derivation_value_and_gradients():
    torch.manual_seed(13)
    n, H, G, Fn, cutoff = 8, 5, 4, 7, 8.0
    ei = _graph(n, 13)
    blk = oops.InteractionBlock(H, G, Fn, cutoff).double()
    for p in blk.parameters():                       # (biases are initialised to zero: make every term visible)
        p.data = p.data + 0.1 * torch.randn_like(p)
    x = torch.randn(n, H, dtype=torch.float64, requires_grad=True)
    ew = torch.rand(ei.shape[1], dtype=torch.float64) * 8
    ew[ei[0] == ei[1]] = 0.0                        # self loops carry distance 0 (cutoff factor 1), process.py:301-305
    ea = torch.rand(ei.shape[1], G, dtype=torch.float64)
    P = [x] + list(blk.parameters())
    a = lambda: blk(x, ei, ew, ea)
    b = lambda: oops.cfconv_loop(x, ei, ew, ea, blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias,
                                 blk.conv.lin1.weight, blk.conv.lin2.weight, blk.conv.lin2.bias, blk.lin.weight, blk.lin.bias, cutoff)
    assert torch.allclose(a(), b(), rtol=1e-11, atol=1e-12)
    for ga, gb in zip(_grads(a, P), _grads(b, P)):
        assert torch.allclose(ga, gb, rtol=1e-9, atol=1e-11)
    perm = torch.randperm(ei.shape[1])
    assert torch.allclose(blk(x, ei[:, perm], ew[perm], ea[perm]), a(), rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("aggr", ["mean", "add"])
def test_nnconv_second_derivation_value_and_gradients(aggr):
    torch.manual_seed(14)
    n, C, G, d3 = 7, 4, 3, 5
    ei = _graph(n, 14)
    ei = ei[:, ei[1] != 2]                            # a node without incoming edges: its mean is 0, root + bias remain
    net = torch.nn.Sequential(torch.nn.Linear(G, d3), torch.nn.ReLU(), torch.nn.Linear(d3, C * C))
    conv = oops.NNConv(C, C, net, aggr=aggr).double()
    conv.bias.data = torch.randn(C, dtype=torch.float64)
    x = torch.randn(n, C, dtype=torch.float64, requires_grad=True)
    ea = torch.rand(ei.shape[1], G, dtype=torch.float64)
    P = [x] + list(conv.parameters())
    a = lambda: conv(x, ei, ea)
    b = lambda: oops.nnconv_loop(x, ei, ea, net[0].weight, net[0].bias, net[2].weight, net[2].bias, conv.lin.weight, conv.bias,
                                 "mean" if aggr == "mean" else "sum")
    assert torch.allclose(a(), b(), rtol=1e-11, atol=1e-12)
    for ga, gb in zip(_grads(a, P), _grads(b, P)):
        assert torch.allclose(ga, gb, rtol=1e-9, atol=1e-11)


def test_gcnconv_second_derivation_value_and_gradients():
    torch.manual_seed(15)
    n, C, Co = 8, 5, 6
    ei = _graph(n, 15)
    ei = ei[:, ~((ei[1] == 3) & (ei[0] != 3))]        # node 3 keeps only its zero-weight self loop: weighted in-degree 0 -> 0
    gcn = oops.GCNConv(C, Co, improved=True, add_self_loops=False).double()
    gcn.bias.data = torch.randn(Co, dtype=torch.float64)
    x = torch.randn(n, C, dtype=torch.float64, requires_grad=True)
    w = torch.rand(ei.shape[1], dtype=torch.float64) * 5 + 0.1
    w[ei[0] == ei[1]] = 0.0
    P = [x] + list(gcn.parameters())
    a = lambda: gcn(x, ei, w)
    b = lambda: oops.gcnconv_dense(x, ei, w, gcn.lin.weight, gcn.bias)
    assert torch.allclose(a(), b(), rtol=1e-11, atol=1e-12)
    assert torch.allclose(a()[3], gcn.bias)           # nothing arrives at a node of weighted in-degree 0
    for ga, gb in zip(_grads(a, P), _grads(b, P)):
        assert torch.allclose(ga, gb, rtol=1e-9, atol=1e-11)


def test_set2set_second_derivation_value_and_gradients():
    torch.manual_seed(16)
    C, steps = 4, 3
    s2s = oops.Set2Set(C, processing_steps=steps).double()
    x = torch.randn(9, C, dtype=torch.float64, requires_grad=True)
    batch = torch.tensor([0, 0, 0, 0, 1, 2, 2, 2, 2])          # a one-node graph in the middle
    L = s2s.lstm
    P = [x] + list(s2s.parameters())
    a = lambda: s2s(x, batch)
    b = lambda: oops.set2set_loop(x, batch, L.weight_ih_l0, L.weight_hh_l0, L.bias_ih_l0, L.bias_hh_l0, steps)
    assert a().shape == (3, 2 * C)
    assert torch.allclose(a(), b(), rtol=1e-10, atol=1e-12)    # (the operator adds 1e-16 to the softmax denominator)
    for ga, gb in zip(_grads(a, P), _grads(b, P)):
        assert torch.allclose(ga, gb, rtol=1e-8, atol=1e-11)


def test_pyg_operator_gradchecks():
    """torch.autograd.gradcheck of the restated operators (the loops above share autograd with them: this checks the
    analytic gradients against finite differences)."""
    torch.manual_seed(17)
    n, C, G = 5, 3, 2
    ei = _graph(n, 17)
    E = ei.shape[1]
    x = torch.randn(n, C, dtype=torch.float64, requires_grad=True)
    ea = torch.rand(E, G, dtype=torch.float64)
    ew = torch.rand(E, dtype=torch.float64) * 6 + 0.2
    blk = oops.InteractionBlock(C, G, 4, 8.0).double()
    assert torch.autograd.gradcheck(lambda x_: blk(x_, ei, ew, ea), (x,))
    net = torch.nn.Sequential(torch.nn.Linear(G, 3), torch.nn.Softplus(), torch.nn.Linear(3, C * C))   # (smooth: no ReLU kink)
    conv = oops.NNConv(C, C, net).double()
    assert torch.autograd.gradcheck(lambda x_: conv(x_, ei, ea), (x,))
    gcn = oops.GCNConv(C, C, improved=True, add_self_loops=False).double()
    assert torch.autograd.gradcheck(lambda x_: gcn(x_, ei, ew), (x,))
    s2s = oops.Set2Set(C, processing_steps=2).double()
    assert torch.autograd.gradcheck(lambda x_: s2s(x_, torch.tensor([0, 0, 1, 1, 1])), (x,))
