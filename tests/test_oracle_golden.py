"""Pin the oracle against vectors produced by the reference's own code (tests/golden/make_golden.py)."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import graph as ograph
from oracle import models as omodels
from oracle import ops as oops

G = os.path.join(os.path.dirname(__file__), "golden")


def _npz(name):
    return np.load(os.path.join(G, name))


def test_rbf_matches_reference_gaussian_smearing():
    z = _npz("rbf.npz")
    d = torch.from_numpy(z["d"])
    out = oops.rbf_expand(d, 0.0, 1.0, 50, 0.2)
    assert float(z["coeff"]) == oops.rbf_coeff(0.0, 1.0, 0.2)  # python float -12.499999999999998
    assert np.float32(z["coeff"]) == np.float32(-12.5)  # what the fp32 multiply sees
    assert np.array_equal(z["offset"], oops.rbf_offsets(0.0, 1.0, 50).numpy())
    assert np.array_equal(out.numpy(), z["out"])  # same torch build, same ops: bit-exact
    # the probe values recorded in SURVEY.md Appendix C
    assert np.allclose(out[2, :4].numpy(), [0.45783, 0.51742, 0.57870, 0.64053], atol=1e-5)


@pytest.mark.parametrize("tag", list("abcde"))
def test_threshold_sort(tag):
    z = _npz("threshold_sort.npz")
    got = ograph.threshold_sort(z["D_" + tag], float(z["r_" + tag]), int(z["k_" + tag]))
    assert np.array_equal(got, z["out_" + tag])


def test_pt10_graphs_and_stats():
    ds, gg = _npz("pt10_dataset.npz"), _npz("pt10_graphs.npz")
    counts = []
    for s in range(len(ds["ids"])):
        ei, ew = ograph.build_graph(ds["positions"][s], ds["cell"][s], ds["pbc"][s], 8.0, 12)
        counts.append(ei.shape[1])
        if s < 8:
            assert np.array_equal(ei, gg["edge_index_%d" % s])
            assert np.array_equal(ew, gg["edge_weight_%d" % s])
        if s >= 40:  # the python-loop oracle is slow; stats over the first 40 + goldens for all
            break
    assert counts == list(gg["edges_per_graph"][: len(counts)])
    assert int(gg["edges_per_graph"].sum()) == 99672 and gg["edges_per_graph"].min() == 92


def test_one_hot_degree():
    z = _npz("onehot_degree.npz")
    got = ograph.one_hot_degree(z["edge_index"], 10, 13)
    assert np.array_equal(got, z["x"][:, 1:])


def test_normalize_edge():
    z = _npz("normalize_edge.npz")
    outs = oops.normalize_edges([torch.from_numpy(z["in_%d" % i]) for i in range(4)])
    for i in range(4):
        assert np.array_equal(outs[i].numpy(), z["out_%d" % i])


def _megnet_inputs(z):
    ns = types.SimpleNamespace
    return ns(x=torch.from_numpy(z["x"]), edge_index=torch.from_numpy(z["edge_index"]),
              edge_attr=torch.from_numpy(z["edge_attr"]), u=torch.from_numpy(z["u"]),
              batch=torch.from_numpy(z["batch"])), torch.from_numpy(z["y"])


class _DS:
    num_features, num_edge_features = 114, 50

    def __getitem__(self, i):
        return types.SimpleNamespace(y=torch.tensor(0.0), u=torch.zeros(1, 3))


@pytest.mark.parametrize("tag,kw", [("bn", dict(batch_norm="True")), ("nobn", dict(batch_norm="False")),
                                    ("max", dict(batch_norm="False", pool="global_max_pool")),
                                    ("late", dict(batch_norm="True", pool_order="late"))])
def test_megnet_matches_reference(tag, kw):
    z = _npz("megnet.npz")
    data, y = _megnet_inputs(z)
    model = omodels.MEGNet(_DS(), dim1=32, dim2=24, dim3=16, pre_fc_count=1, gc_count=2, gc_fc_count=1,
                           post_fc_count=2, **kw)
    sd = {k[len(tag) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + "/sd/")}
    assert set(sd) == set(model.state_dict()), "state_dict key skeleton differs from the reference"
    # training-mode prediction must be computed from the PRE-update BN buffers: reset them
    pre = {k: v.clone() for k, v in sd.items()}
    for k in pre:
        if k.endswith("running_mean"):
            pre[k].zero_()
        elif k.endswith("running_var"):
            pre[k].fill_(1.0)
        elif k.endswith("num_batches_tracked"):
            pre[k].zero_()
    model.load_state_dict(pre)
    model.train()
    pred = model(data)
    assert torch.allclose(pred, torch.from_numpy(z[tag + "/pred_train"]), rtol=1e-5, atol=1e-6)
    torch.nn.functional.l1_loss(pred, y).backward()
    for k, p in model.named_parameters():
        ref = z["%s/grad/%s" % (tag, k)]
        if ref.size == 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
        else:
            assert torch.allclose(p.grad, torch.from_numpy(ref), rtol=1e-4, atol=1e-6), k
    # buffers after one training forward == the reference's saved (post-forward) buffers
    for k, v in model.state_dict().items():
        assert torch.allclose(v.float(), sd[k].float(), rtol=1e-5, atol=1e-6), k
    model.eval()
    assert torch.allclose(model(data), torch.from_numpy(z[tag + "/pred_eval"]), rtol=1e-5, atol=1e-6)


def test_splits_match_reference():
    from matdeeplearn_amd.process import split_data, split_data_CV  # host logic under test

    z = _npz("splits.npz")
    for n, seed in [(1000, 42), (1000, 7), (46744, 42), (37, 3)]:
        tr, va, te = split_data(n, 0.8, 0.05, 0.15, seed=seed)
        assert np.array_equal(tr, z["train_%d_%d" % (n, seed)])
        assert np.array_equal(va, z["val_%d_%d" % (n, seed)])
        assert np.array_equal(te, z["test_%d_%d" % (n, seed)])
    assert list(z["train_1000_42"][:5]) == [542, 618, 816, 68, 94]  # SURVEY Appendix C probe
    folds = split_data_CV(1000, num_folds=5, seed=42)
    for i, f in enumerate(folds):
        assert np.array_equal(f, z["cv5_1000_42_fold%d" % i])


# ------------------------------------------------------------------------------------------------
# The four wrapper files of the reference (cgcnn / schnet / mpnn / gcn), run by make_golden.py with the oracle
# operators injected for the absent torch_geometric: layer creation order (seeded initialisation), state_dict key
# ORDER, pre-FC -> conv -> BN -> act -> dropout ordering, pooling order, set2set sizes, GRU wiring.
# ------------------------------------------------------------------------------------------------
import json  # noqa: E402

WRAPPER_DIMS = dict(dim1=16, dim2=12, dim3=8, gc_count=2)


def _wrappers():
    z = _npz("wrappers.npz")
    return z, json.loads(bytes(z["meta"]).decode())


def _wrapper_cases():
    return sorted(_wrappers()[1])


def _wrapper_batch(z):
    ns = types.SimpleNamespace
    return ns(x=torch.from_numpy(z["x"]), edge_index=torch.from_numpy(z["edge_index"]),
              edge_attr=torch.from_numpy(z["edge_attr"]), edge_weight=torch.from_numpy(z["edge_weight"]),
              batch=torch.from_numpy(z["batch"]), u=torch.zeros(3, 3)), torch.from_numpy(z["y"])


@pytest.mark.parametrize("case", _wrapper_cases())
def test_wrapper_matches_reference_file(case):
    z, meta = _wrappers()
    cls, _ = case.split("/")
    data, y = _wrapper_batch(z)
    torch.manual_seed(4321)
    model = omodels.REGISTRY[cls](_DS(), **WRAPPER_DIMS, **meta[case]["kw"])
    sd = model.state_dict()
    assert list(sd) == meta[case]["sd_keys"], "state_dict key ORDER differs from the reference file"
    for k, v in sd.items():                                     # same seed, same creation order: bit-identical
        assert np.array_equal(v.numpy(), z["%s/init/%s" % (case, k)]), k
    model.train()
    pred = model(data)
    assert torch.allclose(pred, torch.from_numpy(z[case + "/pred_train"]), rtol=1e-5, atol=1e-6)
    torch.nn.functional.l1_loss(pred, y).backward()
    for k, p in model.named_parameters():
        ref = z["%s/grad/%s" % (case, k)]
        if ref.size == 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert torch.allclose(p.grad, torch.from_numpy(ref), rtol=1e-4, atol=1e-6), k
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            assert torch.allclose(v.float(), torch.from_numpy(z["%s/post/%s" % (case, k)]).float(), rtol=1e-5, atol=1e-6), k
    model.eval()
    assert torch.allclose(model(data), torch.from_numpy(z[case + "/pred_eval"]), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", _wrapper_cases())
def test_product_wrapper_seeded_init_and_key_order(case):
    """The product models (constructed on the CPU; their forward needs a HIP device) draw the reference's initial
    weights from the same seed and list the reference's state_dict keys in the reference's order."""
    from matdeeplearn_amd import models

    z, meta = _wrappers()
    cls, _ = case.split("/")
    torch.manual_seed(4321)
    model = getattr(models, cls)(_DS(), **WRAPPER_DIMS, **meta[case]["kw"])
    sd = model.state_dict()
    assert list(sd) == meta[case]["sd_keys"]
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), z["%s/init/%s" % (case, k)]), k


@pytest.mark.parametrize("tag,kw", [("bn", dict(batch_norm="True")), ("late", dict(batch_norm="True", pool_order="late"))])
def test_megnet_seeded_init_and_key_order(tag, kw):
    from matdeeplearn_amd import models

    z = _npz("megnet.npz")
    keys = [k[len(tag) + 4:] for k in z.files if k.startswith(tag + "/sd/")]        # npz keeps insertion order
    for factory in (omodels.MEGNet, models.MEGNet):
        torch.manual_seed(99)
        model = factory(_DS(), dim1=32, dim2=24, dim3=16, pre_fc_count=1, gc_count=2, gc_fc_count=1, post_fc_count=2, **kw)
        assert list(model.state_dict()) == keys
        for k, p in model.named_parameters():
            assert np.array_equal(p.detach().numpy(), z["%s/sd/%s" % (tag, k)]), k
