"""CPU oracle for the MatDeepLearn message-passing hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32/fp64) restatement of the
arithmetic the reference executes on the hot path named by BASELINE.json:north_star.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it;
the product package (`matdeeplearn_amd/`) never does and fails loudly without its HIP library.

Pinning status (see DESIGN.md "Oracle"):
  * PINNED by goldens generated from the reference's own code (tests/golden/make_golden.py,
    tests/test_oracle_golden.py): Gaussian RBF expansion, min/max edge normalisation,
    threshold_sort graph rule, one-hot degree, split_data/split_data_CV, and the complete MEGNet
    model (edge/node/global blocks, forward, gradients) incl. scatter / scatter_mean semantics.
  * PARITY UNPINNED (arithmetic lives in torch_geometric 2.0.1, which is neither vendored under
    /root/reference nor installable here; the reference has no tests or golden vectors):
    CGConv, CFConv/InteractionBlock, NNConv, GCNConv, Set2Set.  These follow the published
    PyG 2.0.1 semantics at the reference's call sites and are
    cross-checked by independent dense-adjacency fp64 derivations (tests/test_oracle_selfcheck.py).
"""
from . import ops, models, graph  # noqa: F401
