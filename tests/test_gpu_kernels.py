"""GPU parity: every HIP kernel (through the C ABI / ctypes binding) against the CPU oracle on the
same seeded inputs.  Tolerances: fp32 mode rtol 2e-5 / atol 2e-5 relative to the tensor scale
(different summation order than the CPU GEMM, device exp/log within ~2 ulp);  bf16 mode is a
performance mode — inputs are rounded to bf16 for BOTH sides and the result must agree to 3e-2 of
the tensor scale (bf16 has 8 bits of mantissa; accumulation is fp32)."""
import os

import numpy as np
import pytest
import torch

from oracle import ops as oops

pytestmark = pytest.mark.gpu
G_DIR = os.path.join(os.path.dirname(__file__), "golden")


def dev():
    return torch.device("cuda:0")


def rand_graph(n, seed, max_in=20, window=40, self_loops=True, empty_frac=0.1, sort=False):
    g = torch.Generator().manual_seed(seed)
    src, tgt = [], []
    for i in range(n):
        if torch.rand(1, generator=g).item() < empty_frac:
            continue
        k = int(torch.randint(1, max_in + 1, (1,), generator=g))
        lo = max(0, i - window)
        hi = min(n, i + window)
        s = torch.randint(lo, hi, (k,), generator=g).tolist()
        if self_loops:
            s.append(i)
        src += s
        tgt += [i] * len(s)
    ei = torch.tensor([src, tgt], dtype=torch.int64)
    if not sort:
        perm = torch.randperm(ei.shape[1], generator=g)
        ei = ei[:, perm]
    return ei


def close(a, b, rtol, atol_scale):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    scale = float(b.abs().max()) + 1e-30
    err = float((a - b).abs().max())
    assert torch.allclose(a, b, rtol=rtol, atol=atol_scale * scale), "max abs err %.3e (scale %.3e)" % (err, scale)


# ---------------------------------------------------------------------------------------------
def test_library_loads_and_reports_version():
    from matdeeplearn_amd import _lib
    assert _lib.lib().mdl_version() == 100


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rbf_matches_reference_golden(dtype):
    from matdeeplearn_amd import ops
    z = np.load(os.path.join(G_DIR, "rbf.npz"))
    d = torch.from_numpy(z["d"]).to(dev())
    out = ops.rbf_expand(d, 0.0, 1.0, 50, 0.2, out_dtype=dtype)
    ref = torch.from_numpy(z["out"])
    if dtype == torch.float32:
        assert torch.allclose(out.cpu(), ref, rtol=1e-6, atol=1e-7)   # exp within ~2 ulp
    else:
        assert torch.allclose(out.float().cpu(), ref.to(torch.bfloat16).float(), rtol=1e-2, atol=1e-6)
    # strided output (padded rows) and empty input
    buf = torch.zeros(d.numel(), 64, dtype=dtype, device=dev())
    ops.rbf_expand(d, out=buf[:, :50])
    assert torch.equal(buf[:, :50], out) and float(buf[:, 50:].abs().max()) == 0.0
    assert ops.rbf_expand(d[:0]).shape == (0, 50)


@pytest.mark.parametrize("reduce", ["sum", "mean", "max"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("sorted_idx", [True, False])
@pytest.mark.parametrize("C", [48, 50, 64])      # 48 / 64: 16-byte vector paths (one / four row lanes), 50: scalar path
def test_scatter_matches_oracle(reduce, dtype, sorted_idx, C):
    from matdeeplearn_amd import ops
    g = torch.Generator().manual_seed(3)
    n_seg, E = 37, 500
    idx = torch.randint(0, n_seg, (E,), generator=g)
    idx[idx == 5] = 6          # an empty segment
    if sorted_idx:
        idx, _ = torch.sort(idx)
    src = torch.randn(E, C, generator=g).to(dtype).float()
    src_o = src.clone().requires_grad_(True)
    ref = oops.scatter(src_o, idx, 0, n_seg, reduce)
    w = torch.randn(n_seg, C, generator=g).to(dtype).float()
    (ref * w).sum().backward()
    src_d = src.to(dev()).to(dtype).requires_grad_(True)
    out = ops.scatter(src_d, idx.to(dev()), 0, n_seg, reduce, assume_sorted=sorted_idx)
    (out.float() * w.to(dev())).sum().backward()
    tol = (1e-5, 1e-6) if dtype == torch.float32 else (2e-2, 1e-2)
    close(out, ref, *tol)
    close(src_d.grad, src_o.grad, *tol)


def _cgconv_case(n, C, G, dtype, sort, seed, aggr="mean", empty_frac=0.1, window=40, split=False, tol=None):
    from matdeeplearn_amd import ops
    g = torch.Generator().manual_seed(seed)
    ei = rand_graph(n, seed, sort=sort, empty_frac=empty_frac, window=window)
    E = ei.shape[1]
    rnd = lambda *s: torch.randn(*s, generator=g)
    x = rnd(n, C).to(dtype).float()
    ea = torch.rand(E, G, generator=g).to(dtype).float()
    k = 1.0 / (2 * C + G) ** 0.5
    wf, ws = (rnd(C, 2 * C + G) * k * 3).to(dtype).float(), (rnd(C, 2 * C + G) * k * 3).to(dtype).float()
    bf, bs = rnd(C) * 0.1, rnd(C) * 0.1
    gout = rnd(n, C).to(dtype).float()

    # oracle (fp32 CPU on the same, already-rounded inputs)
    xo, wfo, wso, bfo, bso = [t.clone().requires_grad_(True) for t in (x, wf, ws, bf, bs)]
    ref = oops.cgconv(xo, ei, ea, wfo, bfo, wso, bso, aggr)
    (ref * gout).sum().backward()

    d = dev()
    xd = x.to(d).to(dtype).requires_grad_(True)
    wfd, wsd, bfd, bsd = [t.to(d).clone().requires_grad_(True) for t in (wf, ws, bf, bs)]
    csr = ops.build_csr(ei.to(d), n, assume_sorted=sort)
    out = ops.cgconv(xd, ei.to(d), ea.to(d).to(dtype), wfd, bfd, wsd, bsd, aggr, csr=csr, split=split)
    (out.float() * gout.to(d)).sum().backward()
    if tol is None:
        tol = (2e-5, 2e-5) if dtype == torch.float32 else (3e-2, 3e-2)
    close(out, ref, *tol)
    close(xd.grad, xo.grad, *tol)
    close(wfd.grad, wfo.grad, *tol)
    close(wsd.grad, wso.grad, *tol)
    close(bfd.grad, bfo.grad, *tol)
    close(bsd.grad, bso.grad, *tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,C,G,sort", [(200, 64, 50, True), (200, 64, 50, False), (77, 32, 50, True),
                                         (130, 100, 50, False), (65, 128, 50, True), (900, 100, 50, True), (50, 64, 41, True),
                                         (33, 20, 7, False), (1, 64, 50, True)])
def test_cgconv_matches_oracle(dtype, n, C, G, sort):
    _cgconv_case(n, C, G, dtype, sort, seed=n + C + G)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cgconv_sum_aggr_and_isolated_nodes(dtype):
    _cgconv_case(150, 64, 50, dtype, True, seed=5, aggr="add", empty_frac=0.5)


def test_cgconv_split_bf16_products_match_oracle():
    """MDL_SPLIT_BF16 ("bf16x3": fp32 storage, the K = 2C + G product as three bf16 MFMAs on (hi, lo)-split operands) against
    the fp32 oracle on UNROUNDED fp32 inputs: forward and every gradient to 2e-4 of the tensor scale — operands carry 16
    significant bits (2^-17 each), i.e. ~1e-5 per product, two orders below the bf16 mode's 3e-2 — on several workgroups, partial
    tiles, isolated nodes, far sources, sum and mean aggregation; and the flag falls back to the exact form off its shape."""
    from matdeeplearn_amd import ops
    t = (2e-4, 2e-4)
    _cgconv_case(700, 64, 50, torch.float32, True, seed=31, empty_frac=0.05, split=True, tol=t)
    _cgconv_case(2500, 64, 50, torch.float32, True, seed=33, empty_frac=0.3, split=True, tol=t)
    _cgconv_case(90, 64, 50, torch.float32, True, seed=32, aggr="add", split=True, tol=t)
    _cgconv_case(1500, 64, 50, torch.float32, True, seed=35, empty_frac=0.0, window=400, split=True, tol=t)
    _cgconv_case(200, 64, 50, torch.float32, False, seed=34, split=True, tol=t)          # unsorted edge list
    _cgconv_case(77, 32, 50, torch.float32, True, seed=36, split=True)                   # no split form at C = 32: exact, exact bound
    # the reference's default width (config.yml:123 dim1 = 100) and 128: the static 128-channel split kernels on zero-padded rows
    _cgconv_case(900, 100, 50, torch.float32, True, seed=37, split=True, tol=t)
    _cgconv_case(130, 100, 50, torch.float32, False, seed=38, split=True, tol=t)
    _cgconv_case(65, 128, 50, torch.float32, True, seed=39, aggr="add", split=True, tol=t)
    # the split form really ran where it exists: its result differs from the exact form's by more than fp32 rounding
    g = torch.Generator().manual_seed(3)
    ei = rand_graph(300, 3, sort=True)
    d = dev()
    x = torch.randn(300, 64, generator=g).to(d)
    ea = torch.rand(ei.shape[1], 50, generator=g).to(d)
    w = [(torch.randn(64, 178, generator=g) * 0.2).to(d) for _ in range(2)]
    b = [torch.zeros(64, device=d) for _ in range(2)]
    csr = ops.build_csr(ei.to(d), 300, assume_sorted=True)
    y0 = ops.cgconv(x, None, ea, w[0], b[0], w[1], b[1], "mean", csr=csr)
    y1 = ops.cgconv(x, None, ea, w[0], b[0], w[1], b[1], "mean", csr=csr, split=True)
    rel = float((y0 - y1).abs().max() / y0.abs().max())
    assert 1e-7 < rel < 1e-4, rel


def test_split_bf16_operands_and_the_split_weight_gradient():
    """mdl_split_bf16 (raw C ABI): hi + lo reproduces an fp32 value to 2^-16 of its magnitude, hi is the round-to-nearest bf16 of
    it; ops._LinearSplitTN (the bf16x3 mode's tall dense layer): forward and dX exact fp32, dW = g^T x over 5e4 rows on split
    operands within 1e-4 of an fp64 product, db exact."""
    from matdeeplearn_amd import _lib, ops
    d = dev()
    g = torch.Generator().manual_seed(11)
    v = (torch.randn(4096, generator=g) * torch.exp(torch.randn(4096, generator=g) * 4)).to(d)
    hi, lo = torch.empty_like(v, dtype=torch.bfloat16), torch.empty_like(v, dtype=torch.bfloat16)
    _lib.check(_lib.lib().mdl_split_bf16(_lib.ptr(v), _lib.ptr(hi), _lib.ptr(lo), v.numel(), _lib.stream()), "mdl_split_bf16")
    assert torch.equal(hi, v.to(torch.bfloat16))
    rec = hi.double() + lo.double()
    assert float(((rec - v.double()).abs() / v.double().abs()).max()) < 2.0 ** -16
    N, K, M = 50000, 114, 64
    x = torch.randn(N, K, generator=g).to(d).requires_grad_(True)
    w = (torch.randn(M, K, generator=g) * 0.1).to(d).requires_grad_(True)
    b = torch.randn(M, generator=g).to(d).requires_grad_(True)
    go = torch.randn(N, M, generator=g).to(d)
    assert ops.linear_split_ok(x, w)
    y = ops._LinearSplitTN.apply(x, w, b)
    y.backward(go)
    ref = torch.nn.functional.linear(x.detach(), w.detach(), b.detach())
    assert torch.equal(y.detach(), ref)
    dw64 = go.double().t() @ x.detach().double()
    err = float((w.grad.double() - dw64).abs().max() / dw64.abs().max())
    assert err < 1e-4, err
    assert torch.allclose(x.grad, go @ w.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(b.grad, go.sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("variant", ["per_wave", "edge_lane"])
def test_cgconv_backward_edge_pass_variants_match_oracle(variant):
    """Both backward edge passes for bf16, C = 64, G = 50 against the oracle, chosen per launch with the explicit flag
    (ops.K3_VARIANT -> MDL_K3_PER_WAVE / MDL_K3_EDGE_LANE in `dtype`; no environment): the per-wave kernel with bf16 by-source
    sums (what small batches run) and the edge-per-lane kernel 2 (producer / reducer waves; the default from 4e5 edges) —
    several workgroups and rounds, partial tiles, isolated nodes (groups without edges), sources outside the by-source window
    (the last case spreads them over +-400 nodes), sum and mean aggregation."""
    from matdeeplearn_amd import ops
    prev, ops.K3_VARIANT = ops.K3_VARIANT, variant
    try:
        _cgconv_case(700, 64, 50, torch.bfloat16, True, seed=21, empty_frac=0.05)
        assert ops.last_k3() == (2 if variant == "edge_lane" else 1)
        _cgconv_case(2500, 64, 50, torch.bfloat16, True, seed=23, empty_frac=0.3)
        _cgconv_case(90, 64, 50, torch.bfloat16, True, seed=22, aggr="add")
        _cgconv_case(1500, 64, 50, torch.bfloat16, True, seed=25, empty_frac=0.0, window=400)
        assert ops.last_k3() == (2 if variant == "edge_lane" else 1)
        _cgconv_case(200, 64, 50, torch.bfloat16, False, seed=24)          # unsorted edge list: sorted copy of the features
    finally:
        ops.K3_VARIANT = prev


def _bulk_like_case(n_graphs, seed, C=64, G=50):
    """a batch shaped like the bench batch: graphs of 4..200 atoms (a tenth wider than the 160-row by-source window), 12
    neighbours + self loop per atom, sources anywhere inside the graph, target-sorted"""
    g = torch.Generator().manual_seed(seed)
    sizes = torch.randint(4, 60, (n_graphs,), generator=g)
    big = torch.rand(n_graphs, generator=g) < 0.1
    sizes = torch.where(big, torch.randint(100, 201, (n_graphs,), generator=g), sizes)
    start = torch.cumsum(sizes, 0) - sizes
    n = int(sizes.sum())
    gid = torch.repeat_interleave(torch.arange(n_graphs), sizes)
    tgt = torch.arange(n).repeat_interleave(13)
    lo, sz = start[gid].repeat_interleave(13), sizes[gid].repeat_interleave(13)
    src = lo + (torch.rand(tgt.numel(), generator=g) * sz).long().clamp_(max=sz.max() - 1) % sz
    src[12::13] = torch.arange(n)                                         # the self loop
    return n, torch.stack([src, tgt])


def test_cgconv_default_dispatch_on_a_large_batch_matches_oracle():
    """What the headline step runs, with NO override of any kind: bf16, C = 64, E >= 4e5 through ops.cgconv — the library's own
    heuristic must pick the edge-per-lane kernel 2 (mdl_debug_last_k3() == 2) on cost-balanced node ranges (the CSR's balance
    prefix is built and handed over) — against the oracle op (cgcnn.py:136-145 via PyG CGConv): output and all five gradients."""
    from matdeeplearn_amd import ops
    assert ops.K3_VARIANT is None and not ops._DET and ops._BALANCE and ops._RSRC16
    n, ei = _bulk_like_case(1400, seed=3)
    E, C, G = ei.shape[1], 64, 50
    assert E >= 400000
    g = torch.Generator().manual_seed(4)
    dtype = torch.bfloat16
    x = torch.randn(n, C, generator=g).to(dtype).float()
    ea = torch.rand(E, G, generator=g).to(dtype).float()
    k = 3.0 / (2 * C + G) ** 0.5
    wf, ws = (torch.randn(C, 2 * C + G, generator=g) * k).to(dtype).float(), (torch.randn(C, 2 * C + G, generator=g) * k).to(dtype).float()
    bf, bs = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    gout = torch.randn(n, C, generator=g).to(dtype).float()
    xo, wfo, wso, bfo, bso = [t.clone().requires_grad_(True) for t in (x, wf, ws, bf, bs)]
    ref = oops.cgconv(xo, ei, ea, wfo, bfo, wso, bso, "mean")
    (ref * gout).sum().backward()
    d = dev()
    xd = x.to(d).to(dtype).requires_grad_(True)
    wfd, wsd, bfd, bsd = [t.to(d).clone().requires_grad_(True) for t in (wf, ws, bf, bs)]
    csr = ops.build_csr(ei.to(d), n, assume_sorted=True)
    assert csr._bal is None
    out = ops.cgconv(xd, ei.to(d), ea.to(d).to(dtype), wfd, bfd, wsd, bsd, "mean", csr=csr)
    (out.float() * gout.to(d)).sum().backward()
    assert ops.last_k3() == 2, "the default dispatch at E = %d must be the edge-per-lane kernel 2" % E
    assert csr._bal is not None and int(csr._bal[-1]) >= 4 * (E + n), "cost-balanced ranges were not used"
    for a, b in ((out, ref), (xd.grad, xo.grad), (wfd.grad, wfo.grad), (wsd.grad, wso.grad), (bfd.grad, bfo.grad), (bsd.grad, bso.grad)):
        close(a, b, 3e-2, 3e-2)


def test_cgconv_deterministic_mode_is_bit_reproducible_and_matches_the_default():
    """MDL_DETERMINISTIC (ops.deterministic()): the backward's atomically accumulated results (r_src -> dx, dwe / db / dWn -> the
    weight and bias gradients) come out with the same BITS on every run, and agree with the default (atomic) launch shape to
    the usual tolerance; the default shape itself is allowed to differ in the last bits from run to run."""
    from matdeeplearn_amd import ops
    d = dev()
    for dtype, C in ((torch.bfloat16, 64), (torch.float32, 64), (torch.bfloat16, 100), (torch.bfloat16, 32)):
        n, ei = _bulk_like_case(60, seed=8)
        E, G = ei.shape[1], 50
        g = torch.Generator().manual_seed(9)
        x0 = torch.randn(n, C, generator=g).to(d).to(dtype)
        ea = torch.rand(E, G, generator=g).to(d).to(dtype)
        wf0, ws0 = (torch.randn(C, 2 * C + G, generator=g) * 0.1).to(d), (torch.randn(C, 2 * C + G, generator=g) * 0.1).to(d)
        bf0, bs0 = (torch.randn(C, generator=g) * 0.1).to(d), (torch.randn(C, generator=g) * 0.1).to(d)
        gout = torch.randn(n, C, generator=g).to(d).to(dtype)
        csr = ops.build_csr(ei.to(d), n, assume_sorted=True)

        def run():
            leaves = [t.clone().requires_grad_(True) for t in (x0, wf0, bf0, ws0, bs0)]
            out = ops.cgconv(leaves[0], ei.to(d), ea, leaves[1], leaves[2], leaves[3], leaves[4], "mean", csr=csr)
            (out.float() * gout.float()).sum().backward()
            return [out.detach()] + [t.grad for t in leaves]
        with ops.deterministic():
            a, b = run(), run()
            assert ops.last_k3() == 3
        for u, v in zip(a, b):
            assert torch.equal(u, v), (dtype, C)
        tol = (2e-5, 2e-5) if dtype == torch.float32 else (3e-2, 3e-2)
        for u, v in zip(a, run()):
            close(u, v, *tol)


def test_cgconv_balanced_ranges_through_the_c_abi():
    """mdl_cgconv_balance against numpy (integer work: exact) and mdl_cgconv_bwd_ex with the prefix — kernel 2 (MDL_K3_EDGE_LANE in `flags`)
    with node ranges of equal cost — against the same kernel on its default partition: the partition changes which workgroup
    sums what, not the sums (bf16 by-source sums: atomic order and window cuts differ).  The oracle check of the balanced
    kernel is test_cgconv_default_dispatch_on_a_large_batch_matches_oracle."""
    import numpy as np
    from matdeeplearn_amd import _lib, ops
    d = dev()
    L, P, st = _lib.lib(), _lib.ptr, _lib.stream
    C, G, dt = 64, 50, _lib.MDL_BF16
    g = torch.Generator().manual_seed(5)
    n = 30000
    tgt = torch.arange(n).repeat_interleave(9)
    spread = torch.where(torch.arange(n) % 7 == 0, 150, 20).repeat_interleave(9)          # every seventh node: far sources
    src = (tgt + (torch.rand(tgt.numel(), generator=g) * 2 - 1) * spread).long().clamp_(0, n - 1)
    n_pad = n + 700                                                                        # edge-less rows at the end
    E = tgt.numel()
    csr = ops.build_csr(torch.stack([src, tgt]).to(d), n_pad, assume_sorted=True)
    cost = torch.empty(n_pad + 1, dtype=torch.int32, device=d)
    _lib.check(L.mdl_cgconv_balance(P(csr.rowptr), P(csr.src), n_pad, P(cost), st()), "balance")
    rp, s = csr.rowptr.cpu().numpy(), csr.src.cpu().numpy()
    deg = np.diff(rp)
    is_far = np.append((np.abs(s - np.repeat(np.arange(n_pad), deg)) >= 48).astype(np.int64), 0)    # (+ one slot: rp may equal E)
    far = np.add.reduceat(is_far, rp[:-1]) * (deg > 0)
    ref = np.zeros(n_pad + 1, np.int64)
    ref[1:] = 4 * (deg + 1) + 5 * far + 4 * (deg == 0)
    assert np.array_equal(cost.cpu().numpy(), ref), "mdl_cgconv_balance"
    assert torch.equal(csr.balance().cpu(), torch.from_numpy(np.cumsum(ref)).to(torch.int32))
    x = torch.randn(n_pad, C, generator=g).to(d).to(torch.bfloat16)
    ea = torch.rand(E, G, generator=g).to(d).to(torch.bfloat16)
    gout = torch.randn(n_pad, C, generator=g).to(d).to(torch.bfloat16)
    wf, ws = (torch.randn(C, 2 * C + G, generator=g) * 0.1).to(d), (torch.randn(C, 2 * C + G, generator=g) * 0.1).to(d)
    bf, bs = torch.zeros(C, device=d), torch.zeros(C, device=d)
    wpack = torch.empty(L.mdl_cgconv_wpack_bytes(C, G, dt), dtype=torch.uint8, device=d)
    bpack = torch.empty(2 * C, dtype=torch.float32, device=d)
    _lib.check(L.mdl_cgconv_pack_weights(P(wf), P(bf), P(ws), P(bs), C, G, P(wpack), P(bpack), dt, st()), "pack")
    res = []
    for bal in (None, csr.balance()):
        r_tgt = torch.empty(n_pad, 2 * C, device=d, dtype=torch.bfloat16)
        r_src = torch.zeros(n_pad, 2 * C, device=d, dtype=torch.bfloat16)
        dwe, db = torch.zeros(2 * C, 64, device=d), torch.zeros(2 * C, device=d)
        a = _lib.cg_args(dtype=dt, flags=_lib.MDL_K3_EDGE_LANE, aggr=1, N=n_pad, E=E, C=C, G=G, x=x, edge_attr=ea, rowptr=csr.rowptr,
                         src=csr.src, tgt=csr.tgt, wpack=wpack, bpack=bpack, grad_out=gout, r_tgt=r_tgt, r_src=r_src,
                         r_src_dtype=_lib.MDL_BF16, dwe=dwe, db=db, balance=bal)
        _lib.check(L.mdl_cgconv_bwd_ex(a, st()), "bwd_ex")
        assert L.mdl_debug_last_k3() == 2
        res.append((r_tgt, r_src, dwe, db))
    (rt0, rs0, dwe0, db0), (rt1, rs1, dwe1, db1) = res
    assert float(rt0[n:].abs().max()) == 0.0 and float(rt1[n:].abs().max()) == 0.0       # rows of the edge-less tail
    close(rt0, rt1, 8e-3, 1e-3)
    close(dwe0, dwe1, 2e-3, 1e-4)
    close(db0, db1, 2e-3, 1e-4)
    close(rs0, rs1, 2e-2, 2e-2)


def test_cgconv_c_abi_eperm_and_workspace_paths():
    """Straight through the C ABI: (a) unsorted edge features addressed through eperm (generic kernels) give the same
    forward as target-sorted features (static kernels); (b) the backward with and without the optional workspace
    (dynamic vs fixed work distribution) agrees to fp32 atomic-order noise."""
    from matdeeplearn_amd import _lib, ops
    d = dev()
    L, P, st = _lib.lib(), _lib.ptr, _lib.stream
    n, C, G = 5000, 64, 50
    ei = rand_graph(n, 31, sort=False, empty_frac=0.0)
    E = ei.shape[1]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, C, generator=g).to(d).to(torch.bfloat16)
    ea = torch.rand(E, G, generator=g).to(d).to(torch.bfloat16)
    wf, ws = (torch.randn(C, 2 * C + G, generator=g) * 0.1).to(d), (torch.randn(C, 2 * C + G, generator=g) * 0.1).to(d)
    bf, bs = torch.zeros(C, device=d), torch.zeros(C, device=d)
    csr = ops.build_csr(ei.to(d), n, assume_sorted=False)
    assert csr.eperm is not None
    dt = _lib.MDL_BF16
    wpack = torch.empty(L.mdl_cgconv_wpack_bytes(C, G, dt), dtype=torch.uint8, device=d)
    bpack = torch.empty(2 * C, dtype=torch.float32, device=d)
    _lib.check(L.mdl_cgconv_pack_weights(P(wf), P(bf), P(ws), P(bs), C, G, P(wpack), P(bpack), dt, st()), "pack")
    ea_sorted = ea.index_select(0, csr.eperm.long()).contiguous()
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    _lib.check(L.mdl_cgconv_fwd(P(x), P(ea), P(csr.rowptr), P(csr.src), P(csr.tgt), P(csr.eperm), P(wpack), P(bpack), P(o1),
                                n, E, C, G, 1, dt, st()), "fwd eperm")
    _lib.check(L.mdl_cgconv_fwd(P(x), P(ea_sorted), P(csr.rowptr), P(csr.src), P(csr.tgt), None, P(wpack), P(bpack), P(o2),
                                n, E, C, G, 1, dt, st()), "fwd sorted")
    close(o1, o2, 2e-2, 1e-2)

    # (b) a graph large enough for the dynamic schedule to engage (>= 4 groups per wave): 8 in-window sources per node
    n = 70000
    tgt = torch.arange(n).repeat_interleave(8)
    src = (tgt + torch.randint(-20, 21, (tgt.numel(),), generator=g)).clamp_(0, n - 1)
    E = tgt.numel()
    csr = ops.build_csr(torch.stack([src, tgt]).to(d), n, assume_sorted=True)
    x = torch.randn(n, C, generator=g).to(d).to(torch.bfloat16)
    ea_sorted = torch.rand(E, G, generator=g).to(d).to(torch.bfloat16)
    gout = torch.randn(n, C, generator=g).to(d).to(torch.bfloat16)
    res = []
    ws_bytes = L.mdl_cgconv_workspace_bytes(n, E, C, G, dt)
    assert ws_bytes > 0
    for use_ws in (False, True):
        r_tgt = torch.empty(n, 2 * C, device=d, dtype=torch.bfloat16)        # compute dtype
        r_src = torch.zeros(n, 2 * C, device=d)
        dwe = torch.zeros(2 * C, 64, device=d)
        db = torch.zeros(2 * C, device=d)
        wsb = torch.full((ws_bytes,), 0xAB, dtype=torch.uint8, device=d) if use_ws else None      # garbage: the library zeroes it
        a = _lib.cg_args(dtype=dt, aggr=1, N=n, E=E, C=C, G=G, x=x, edge_attr=ea_sorted, rowptr=csr.rowptr, src=csr.src, tgt=csr.tgt,
                         wpack=wpack, bpack=bpack, grad_out=gout, r_tgt=r_tgt, r_src=r_src, r_src_dtype=_lib.MDL_F32, dwe=dwe, db=db,
                         workspace=wsb, workspace_bytes=ws_bytes if use_ws else 0)
        _lib.check(L.mdl_cgconv_bwd_ex(a, st()), "bwd_ex")
        res.append((r_tgt, r_src, dwe, db))
    (rt0, rs0, dwe0, db0), (rt1, rs1, dwe1, db1) = res
    close(rt0, rt1, 8e-3, 1e-3)          # by-target sums: fp32 accumulation, stored in bf16 (one ulp of slack)
    close(dwe0, dwe1, 1e-4, 1e-5)        # fp32 partial sums per wave, atomics in a different order
    close(db0, db1, 1e-4, 1e-5)
    # by-source sums: an edge whose source falls outside its group's 64-node window is added in fp32 instead of as a
    # bf16-rounded MFMA operand, and the two schedules cut the groups (hence the windows) differently
    close(rs0, rs1, 1e-2, 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,Ci,Co,D3", [(300, 100, 100, 100), (57, 24, 24, 16), (40, 20, 36, 50), (64, 32, 128, 128), (33, 16, 30, 34),
                                        (40, 8, 4, 2)])      # (D3 = 2: the flat Y_j staging must not be chosen, w2 = 1)
def test_nnconv_contraction_matches_oracle(dtype, n, Ci, Co, D3):
    """K7 through nn.NNConv (Y = x W2r per node, per-edge mat-vec by source) against the oracle's NNConv, which builds the
    per-edge Ci x Co matrices like the reference does: output and every gradient (x, edge network, root weight, bias)."""
    from matdeeplearn_amd import nn as pnn
    g = torch.Generator().manual_seed(n + Ci)
    ei = rand_graph(n, n + Co, sort=False, empty_frac=0.15)
    # a hub with 75 out-edges: the MFMA kernels take a node's out-edges 32 at a time (two full chunks + a ragged one)
    ei = torch.cat([ei, torch.stack([torch.full((75,), 3, dtype=ei.dtype), torch.randint(0, n, (75,), generator=g).to(ei.dtype)])], dim=1)
    E = ei.shape[1]
    torch.manual_seed(n)
    mk = lambda: torch.nn.Sequential(torch.nn.Linear(50, D3), torch.nn.ReLU(), torch.nn.Linear(D3, Ci * Co))
    oc = oops.NNConv(Ci, Co, mk(), aggr="mean")
    with torch.no_grad():
        for p in oc.parameters():
            p.copy_((p * (1.0 if p.dim() > 1 else 1.0) + (0.05 * torch.randn(p.shape, generator=g) if p.dim() == 1 else 0)).to(dtype).float())
    pc = pnn.NNConv(Ci, Co, mk(), aggr="mean")
    pc.load_state_dict(oc.state_dict())
    pc.to(dev())
    x = torch.randn(n, Ci, generator=g).to(dtype).float()
    ea = torch.rand(E, 50, generator=g).to(dtype).float()
    gout = torch.randn(n, Co, generator=g)
    xo = x.clone().requires_grad_(True)
    ref = oc(xo, ei, ea)
    (ref * gout).sum().backward()
    xd = x.to(dev()).to(dtype).requires_grad_(True)
    out = pc(xd, ei.to(dev()), ea.to(dev()).to(dtype))
    (out.float() * gout.to(dev())).sum().backward()
    tol = (1e-4, 1e-4) if dtype == torch.float32 else (4e-2, 4e-2)
    close(out, ref, *tol)
    close(xd.grad, xo.grad, *tol)
    og = dict(oc.named_parameters())
    for k, p in pc.named_parameters():
        close(p.grad, og[k].grad, *tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C", [(3001, 100), (257, 64), (5, 7), (1, 4), (1200, 30)])
def test_fused_gru_gates_match_the_written_out_step_and_torch_gru(dtype, N, C):
    """csrc/gru.hip (ops.gru_gates: the gate arithmetic of one GRU step as one launch per direction, mpnn.py:160-161) against the
    written-out tensor form on the same (rounded) gi / gh in fp32 — both outputs and, with gradients arriving at BOTH outputs,
    d gi, d gh, d h — and, in fp32, the whole step against torch.nn.GRU; vector (C % 4 == 0) and scalar widths."""
    from matdeeplearn_amd import ops
    from matdeeplearn_amd.models.mpnn import gru_step
    d = dev()
    g = torch.Generator().manual_seed(N * 7 + C)
    gi0 = (torch.randn(N, 3 * C, generator=g) * 1.5).to(d).to(dtype)
    gh0 = (torch.randn(N, 3 * C, generator=g) * 1.5).to(d).to(dtype)
    h0 = torch.randn(N, C, generator=g).to(d)
    w1 = torch.randn(N, C, generator=g).to(d)
    w2 = torch.randn(N, C, generator=g).to(d).to(dtype)
    assert ops.gru_gates_ok(gi0, gh0, h0)
    gi, gh, h = (t.clone().requires_grad_(True) for t in (gi0, gh0, h0))
    hn, out = ops.gru_gates(gi, gh, h)
    assert hn.dtype == torch.float32 and out.dtype == dtype and out.shape == hn.shape
    ((hn * w1).sum() + (out.float() * w2.float()).sum()).backward()
    gir, ghr, hr = (t.clone().float().requires_grad_(True) for t in (gi0, gh0, h0))
    i_r, i_z, i_n = gir.chunk(3, dim=1)
    h_r, h_z, h_n = ghr.chunk(3, dim=1)
    r = torch.sigmoid(i_r + h_r); z = torch.sigmoid(i_z + h_z); n = torch.tanh(i_n + r * h_n)
    ref = n + z * (hr - n)
    ref_out = ref.to(dtype)
    ((ref * w1).sum() + (ref_out.float() * w2.float()).sum()).backward()
    close(hn, ref, 2e-6, 2e-6)
    assert torch.equal(out, hn.to(dtype))
    tol = (2e-5, 2e-5) if dtype == torch.float32 else (1e-2, 1e-2)          # (bf16: d gi / d gh are rounded once on the way out)
    close(gi.grad, gir.grad, *tol)
    close(gh.grad, ghr.grad, *tol)
    close(h.grad, hr.grad, 2e-5, 2e-5)
    if dtype == torch.float32:
        torch.manual_seed(C)
        gru = torch.nn.GRU(C, C).to(d)
        x = torch.randn(N, C, generator=g).to(d).requires_grad_(True)
        hs = h0.clone().requires_grad_(True)
        o_ref = gru(x.unsqueeze(0), hs.unsqueeze(0))[1].squeeze(0)
        g_ref = torch.autograd.grad((o_ref * w1).sum(), [x, hs] + list(gru.parameters()))
        o, o2 = gru_step(gru, x, hs)
        assert o2 is o
        g_hip = torch.autograd.grad((o * w1).sum(), [x, hs] + list(gru.parameters()))
        close(o, o_ref, 1e-5, 1e-5)
        for a, b in zip(g_hip, g_ref):
            close(a, b, 1e-4, 1e-4)


@pytest.mark.parametrize("N,K,M", [(1500, 100, 10000), (1024, 64, 641), (2077, 160, 1000), (4099, 24, 2400), (1100, 100, 192)])
def test_wide_output_dense_layer_matches_the_library_product(N, K, M):
    """mdl_linear_wide (ops.matmul_wide: NNConv's Y = x W2r — thousands of output columns in 192-column blocks per workgroup)
    against the library's x @ w on the same bf16 operands: ragged row counts, a last column block of 1 .. 191 columns, K below
    and at the kernel's maximum; and its autograd (the two library products) against torch's."""
    from matdeeplearn_amd import ops
    g = torch.Generator().manual_seed(N + M)
    x = torch.randn(N, K, generator=g).to(dev()).to(torch.bfloat16)
    w = (torch.randn(K, M, generator=g) * (1.0 / K ** 0.5)).to(dev()).to(torch.bfloat16)
    xa = x.clone().requires_grad_(True)
    wa = w.clone().requires_grad_(True)
    out = ops._LinearWide.apply(xa, wa)
    ref = x.float() @ w.float()
    assert out.shape == (N, M) and out.dtype == torch.bfloat16
    close(out, ref, 1e-2, 1e-2)                                              # one bf16 rounding of an fp32-accumulated sum
    gy = torch.randn(N, M, generator=g).to(dev()).to(torch.bfloat16)
    out.backward(gy)
    close(xa.grad, gy.float() @ w.float().t(), 2e-2, 2e-2)
    close(wa.grad, x.float().t() @ gy.float(), 2e-2, 2e-2)


@pytest.mark.parametrize("E,d", [(5003, 100), (1500, 64), (70, 24)])
def test_megnet_edge_block_first_layer_without_concatenation(E, d):
    """K6 (ops.linear_gather_act) against the reference's formulation relu(Linear(cat[x[row], x[col], e, u[b]])) in fp32 on
    bf16-rounded operands: output and the gradients w.r.t. the edge state, the node state, u, weight and bias."""
    from matdeeplearn_amd import ops
    d_ = dev()
    g = torch.Generator().manual_seed(E + d)
    N, B = 301, 17
    row = torch.randint(0, N, (E,), generator=g)
    col = torch.sort(torch.randint(0, N, (E,), generator=g)).values
    be = torch.sort(torch.randint(0, B, (E,), generator=g)).values
    bf = lambda t: t.to(torch.bfloat16).float()
    x, e, u = bf(torch.randn(N, d, generator=g)), bf(torch.randn(E, d, generator=g)), bf(torch.randn(B, d, generator=g))
    W, b = bf(torch.randn(d, 4 * d, generator=g) / (4 * d) ** 0.5), bf(torch.randn(d, generator=g) * 0.1)
    gout = torch.randn(E, d, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x, e, u, W, b)]
    xo, eo, uo, Wo, bo = leaves
    pre = torch.nn.functional.linear(torch.cat([xo[row], xo[col], eo, uo[be]], 1), Wo, bo)
    # relu' jumps at 0: a pre-activation within bf16 rounding of the kink flips its mask and moves a gradient by a whole
    # |gout| — not a kernel property.  The upstream gradient is therefore switched off in a band around the kink.
    gout = gout * (pre.detach().abs() > 0.05)
    ref = torch.relu(pre)
    (ref * gout).sum().backward()
    xd, ed, ud = [t.to(d_).to(torch.bfloat16).requires_grad_(True) for t in (x, e, u)]
    Wd, bd = W.to(d_).requires_grad_(True), b.to(d_).requires_grad_(True)
    cd = torch.bfloat16
    wa, wb, wc, wdd = (Wd[:, k * d:(k + 1) * d] for k in range(4))
    p1 = torch.nn.functional.linear(xd, wa.to(cd))
    p2 = torch.nn.functional.linear(xd, wb.to(cd))
    p3 = torch.nn.functional.linear(ud, wdd.to(cd), bd.to(cd))
    out = ops.linear_gather_act(ed, wc, None, "relu", [(p1, row.to(d_).int()), (p2, col.to(d_).int()), (p3, be.to(d_).int())])
    assert out.dtype == torch.bfloat16
    (out.float() * gout.to(d_)).sum().backward()
    close(out, ref, 3e-2, 3e-2)
    for a, r in ((xd, xo), (ed, eo), (ud, uo), (Wd, Wo), (bd, bo)):
        close(a.grad, r.grad, 4e-2, 4e-2)


def test_buffer_stores_past_the_last_row_are_dropped():
    """Kernels that end a row range with range-checked buffer stores (dense-layer forward, node-level backward, the gate
    factors of the training forward) must not touch memory behind their outputs when the row count is not a multiple of
    the tile: outputs are carved out of a sentinel-filled arena and the bytes behind them are checked."""
    from matdeeplearn_amd import _lib, ops
    d = dev()
    L, P, st = _lib.lib(), _lib.ptr, _lib.stream
    g = torch.Generator().manual_seed(23)
    SENT = -17.5

    def carve(rows, cols, dtype=torch.bfloat16, slack=64):
        arena = torch.full(((rows + slack) * cols,), SENT, dtype=dtype, device=d)
        return arena, arena[:rows * cols].view(rows, cols)

    def guard_ok(arena, rows, cols):
        return bool((arena[rows * cols:].float() == SENT).all())

    # dense layer forward: N = 1000 rows (tile 64), K = 114, M = 64
    N, K, M = 1000, 114, 64
    x = torch.randn(N, K, generator=g).to(d).to(torch.bfloat16)
    w = (torch.randn(M, K, generator=g) * 0.1).to(d).to(torch.bfloat16)
    b = torch.randn(M, generator=g).to(d).to(torch.bfloat16)
    arena, out = carve(N, M)
    _lib.check(L.mdl_linear_act(P(x), P(w), P(b), P(out), N, K, M, 1, _lib.MDL_BF16, st()), "linear_act")
    torch.cuda.synchronize()
    assert guard_ok(arena, N, M)
    close(out, torch.relu(x.float() @ w.float().t() + b.float()), 2e-2, 2e-2)

    # conv: training forward (gate rows) and node-level backward (dx rows) on a graph whose sizes are odd
    n, C, G = 1003, 64, 50
    tgt = torch.arange(n).repeat_interleave(7)
    src = (tgt + torch.randint(-9, 10, (tgt.numel(),), generator=g)).clamp_(0, n - 1)
    E = tgt.numel()
    csr = ops.build_csr(torch.stack([src, tgt]).to(d), n, assume_sorted=True)
    xx = torch.randn(n, C, generator=g).to(d).to(torch.bfloat16)
    ea = torch.rand(E, G, generator=g).to(d).to(torch.bfloat16)
    wf, ws = (torch.randn(C, 2 * C + G, generator=g) * 0.1).to(d), (torch.randn(C, 2 * C + G, generator=g) * 0.1).to(d)
    dt = _lib.MDL_BF16
    wpack = torch.empty(L.mdl_cgconv_wpack_bytes(C, G, dt), dtype=torch.uint8, device=d)
    bpack = torch.empty(2 * C, dtype=torch.float32, device=d)
    _lib.check(L.mdl_cgconv_pack_weights(P(wf), None, P(ws), None, C, G, P(wpack), P(bpack), dt, st()), "pack")
    oa, o = carve(n, C)
    _lib.check(L.mdl_cgconv_fwd(P(xx), P(ea), P(csr.rowptr), P(csr.src), P(csr.tgt), None, P(wpack), P(bpack), P(o),
                                n, E, C, G, 1, dt, st()), "fwd")
    torch.cuda.synchronize()
    assert guard_ok(oa, n, C)
    r_tgt = torch.randn(n, 2 * C, generator=g).to(d).to(torch.bfloat16)
    r_src = torch.randn(n, 2 * C, generator=g).to(d)
    gout = torch.randn(n, C, generator=g).to(d).to(torch.bfloat16)
    wn_t = torch.empty(C, 4 * C, dtype=torch.bfloat16, device=d)
    _lib.check(L.mdl_cgconv_pack_node_weights(P(wf), P(ws), C, G, P(wn_t), dt, st()), "pack_node")
    da, dx = carve(n, C)
    dwn = torch.zeros(4 * C, C, device=d)
    _lib.check(L.mdl_cgconv_bwd_node_ex(_lib.cg_node_args(dtype=dt, N=n, C=C, r_src_dtype=_lib.MDL_F32, x=xx, grad_out=gout, r_tgt=r_tgt,
                                                          r_src=r_src, wn_t=wn_t, dx=dx, dwn=dwn), st()), "bwd_node_ex")
    torch.cuda.synchronize()
    assert guard_ok(da, n, C)
    R = torch.cat([r_tgt.float(), r_src.to(torch.bfloat16).float()], dim=1)
    close(dx, gout.float() + R @ wn_t.float().t(), 3e-2, 3e-2)


def test_cgconv_rejects_cpu_tensors_and_bad_shapes():
    from matdeeplearn_amd import ops
    x = torch.randn(4, 64)
    ei = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(ops.MdlError):
        ops.cgconv(x, ei, torch.randn(2, 50), torch.randn(64, 178), None, torch.randn(64, 178), None)
    d = dev()
    with pytest.raises(ops.MdlError):   # G > 64 unsupported in this round
        ops.cgconv(x.to(d), ei.to(d), torch.randn(2, 80, device=d), torch.randn(64, 208, device=d), None,
                   torch.randn(64, 208, device=d), None)


def test_cgconv_large_graph_permutation_invariance():
    """Full-size property check (E ~ 1.3e5): shuffling the edge list must not change the result
    beyond fp32 summation-order noise, and sum-aggregation is linear in the in-degree."""
    from matdeeplearn_amd import ops
    d = dev()
    n, C, G = 10000, 64, 50
    ei = rand_graph(n, 11, sort=True, empty_frac=0.0)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, C, generator=g).to(d)
    ea = torch.rand(ei.shape[1], G, generator=g).to(d)
    wf, ws = torch.randn(C, 178, generator=g).to(d) * 0.1, torch.randn(C, 178, generator=g).to(d) * 0.1
    a = ops.cgconv(x, ei.to(d), ea, wf, None, ws, None, "mean", csr=ops.build_csr(ei.to(d), n, True))
    perm = torch.randperm(ei.shape[1], generator=g)
    eip = ei[:, perm].to(d)
    b = ops.cgconv(x, eip, ea[perm.to(d)], wf, None, ws, None, "mean")
    close(a, b, 1e-5, 1e-6)


# ---------------------------------------------------------------------------------------------
# generic gather / gather-mul-reduce (SchNet / GCN / MEGNet building blocks)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("sort", [True, False])
@pytest.mark.parametrize("reduce,use_w,use_scale", [("sum", True, True), ("mean", False, False), ("sum", False, True)])
def test_gather_mul_reduce_matches_oracle(dtype, sort, reduce, use_w, use_scale):
    from matdeeplearn_amd import ops
    g = torch.Generator().manual_seed(9)
    n, F_ = 90, 40
    ei = rand_graph(n, 9, sort=sort)
    E = ei.shape[1]
    h = torch.randn(n, F_, generator=g).to(dtype).float()
    w = torch.randn(E, F_, generator=g).to(dtype).float() if use_w else None
    sc = torch.rand(E, generator=g) if use_scale else None
    ho = h.clone().requires_grad_(True)
    wo = w.clone().requires_grad_(True) if use_w else None
    msg = ho.index_select(0, ei[0])
    if use_w:
        msg = msg * wo
    if use_scale:
        msg = msg * sc.view(-1, 1)
    ref = oops.scatter(msg, ei[1], 0, n, reduce)
    gout = torch.randn(n, F_, generator=g).to(dtype).float()
    (ref * gout).sum().backward()
    d = dev()
    hd = h.to(d).to(dtype).requires_grad_(True)
    wd = w.to(d).to(dtype).requires_grad_(True) if use_w else None
    csr = ops.build_csr(ei.to(d), n, assume_sorted=sort)
    out = ops.gather_mul_reduce(hd, csr, w=wd, scale=None if sc is None else sc.to(d), reduce=reduce)
    (out.float() * gout.to(d)).sum().backward()
    tol = (1e-5, 1e-6) if dtype == torch.float32 else (3e-2, 2e-2)
    close(out, ref, *tol)
    close(hd.grad, ho.grad, *tol)
    if use_w:
        close(wd.grad, wo.grad, *tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gather_rows_and_its_gradient(dtype):
    from matdeeplearn_amd import ops
    g = torch.Generator().manual_seed(4)
    src = torch.randn(50, 24, generator=g).to(dtype)
    idx = torch.randint(0, 50, (300,), generator=g)
    d = dev()
    sd = src.to(d).requires_grad_(True)
    out = ops.gather(sd, idx.to(d))
    assert torch.equal(out.cpu(), src.index_select(0, idx))
    wgt = torch.randn(300, 24, generator=g).to(dtype)
    (out * wgt.to(d)).sum().backward()
    ref = torch.zeros(50, 24).index_add_(0, idx, wgt.float())
    close(sd.grad, ref, 1e-5 if dtype == torch.float32 else 3e-2, 1e-6 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("N,M,K", [(1000, 64, 114), (129, 32, 50), (5000, 100, 200), (77, 128, 256), (1, 7, 3), (8192, 1, 64),
                                   (8192, 64, 64), (3000, 2, 64), (700, 64, 126), (2100, 150, 150), (3001, 150, 50),
                                   (900, 64, 158), (1500, 100, 100), (640, 160, 32), (333, 32, 160)])
def test_gemm_tn_matches_torch(N, M, K):
    """Tall-skinny weight-gradient GEMM (bf16 in, fp32 out) vs an fp32 matmul of the same rounded inputs."""
    from matdeeplearn_amd import _lib
    g = torch.Generator().manual_seed(N + M)
    a = torch.randn(N, M, generator=g).to(torch.bfloat16).to(dev())
    b = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev())
    c = torch.zeros(M, K, device=dev())
    _lib.check(_lib.lib().mdl_gemm_tn(_lib.ptr(a), a.stride(0), M, _lib.ptr(b), b.stride(0), K, _lib.ptr(c), N,
                                      _lib.MDL_BF16, _lib.stream()), "mdl_gemm_tn")
    ref = a.float().t() @ b.float()
    close(c, ref, 1e-4, 1e-5)
    if M % 2 == 0 and K % 2 == 0 and K <= 158:          # same pass + column sums of a (the Linear's bias gradient)
        c2, cs = torch.zeros(M, K, device=dev()), torch.zeros(M, device=dev())
        _lib.check(_lib.lib().mdl_gemm_tn_colsum(_lib.ptr(a), a.stride(0), M, _lib.ptr(b), b.stride(0), K, _lib.ptr(c2),
                                                 _lib.ptr(cs), N, _lib.MDL_BF16, _lib.stream()), "mdl_gemm_tn_colsum")
        close(c2, ref, 1e-4, 1e-5)
        close(cs, a.float().sum(0), 1e-4, 1e-5)


@pytest.mark.parametrize("N,K,M,act,use_bias", [(1000, 114, 64, "relu", True), (1001, 114, 64, "relu", True),
                                                (8192, 64, 64, "relu", True), (2051, 50, 20, None, False),
                                                (1500, 256, 128, "relu", True), (1027, 4, 1, None, True), (67, 114, 64, "relu", True)])
def test_linear_act_matches_torch(N, K, M, act, use_bias):
    """Fused dense layer forward (GEMM + bias + activation, one streaming kernel) through the C ABI, and the autograd
    wrapper's gradients, vs fp32 torch on the same bf16-rounded inputs.  N = 1001 / 1027 / 67: the last 16-byte chunk of
    the input straddles the end of the array; 67: a single partial tile."""
    from matdeeplearn_amd import _lib, ops
    g = torch.Generator().manual_seed(N + K + M)
    x = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev())
    w = (torch.randn(M, K, generator=g) / K ** 0.5).to(dev())
    b = (torch.randn(M, generator=g) * 0.1).to(dev()) if use_bias else None
    wl, bl = w.to(torch.bfloat16), None if b is None else b.to(torch.bfloat16)
    out = torch.empty(N, M, dtype=torch.bfloat16, device=dev())
    _lib.check(_lib.lib().mdl_linear_act(_lib.ptr(x), _lib.ptr(wl), _lib.ptr(bl), _lib.ptr(out), N, K, M,
                                         1 if act == "relu" else 0, _lib.MDL_BF16, _lib.stream()), "mdl_linear_act")
    ref = torch.nn.functional.linear(x.float(), wl.float(), None if bl is None else bl.float())
    if act == "relu":
        ref = torch.relu(ref)
    close(out, ref, 1e-2, 4e-3)
    if N >= 1024:                                            # the autograd path (forward fused, dW / db on the TN GEMM)
        wp = w.clone().requires_grad_(True)
        bp = None if b is None else b.clone().requires_grad_(True)
        xg = x.clone().requires_grad_(True)
        y = ops.linear_act(xg, wp, bp, act)
        go = torch.randn(N, M, generator=g).to(dev())
        (y.float() * go).sum().backward()
        xr, wr = x.float().clone().requires_grad_(True), wl.float().clone().requires_grad_(True)
        br = None if bl is None else bl.float().clone().requires_grad_(True)
        yr = torch.nn.functional.linear(xr, wr, br)
        yr = torch.relu(yr) if act == "relu" else yr
        (yr * go.to(torch.bfloat16).float()).sum().backward()
        close(y, yr, 1e-2, 4e-3)
        close(wp.grad, wr.grad, 3e-2, 2e-2)
        close(xg.grad, xr.grad, 3e-2, 2e-2)
        if bp is not None:
            close(bp.grad, br.grad, 3e-2, 2e-2)


@pytest.mark.parametrize("N,M,K,act", [(3001, 150, 50, 2), (2100, 150, 150, 2), (1000, 64, 114, 1), (777, 20, 50, 2),
                                       (5000, 128, 64, 1)])
def test_gemm_tn_act_matches_torch(N, M, K, act):
    """TN GEMM with the activation derivative applied in its staging (mdl_gemm_tn_act) vs the two-step form: masked /
    scaled gradient rounded to bf16 (what mdl_ssp_bwd / threshold_backward would have written), then an fp32 product."""
    from matdeeplearn_amd import _lib
    g = torch.Generator().manual_seed(N + M + act)
    a = torch.randn(N, M, generator=g).to(torch.bfloat16).to(dev())
    pre = torch.randn(N, M, generator=g) * 2
    y = (torch.relu(pre) if act == 1 else torch.nn.functional.softplus(pre) - 0.6931471805599453).to(torch.bfloat16).to(dev())
    b = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev())
    c, cs = torch.zeros(M, K, device=dev()), torch.zeros(M, device=dev())
    _lib.check(_lib.lib().mdl_gemm_tn_act(_lib.ptr(a), a.stride(0), M, _lib.ptr(y), y.stride(0), act, _lib.ptr(b), b.stride(0), K,
                                          _lib.ptr(c), _lib.ptr(cs), N, _lib.MDL_BF16, _lib.stream()), "mdl_gemm_tn_act")
    fac = (y.float() > 0).float() if act == 1 else 1.0 - torch.exp(-(y.float() + 0.6931471805599453))
    am = (a.float() * fac).to(torch.bfloat16).float()
    close(c, am.t() @ b.float(), 1e-4, 2e-5 * N ** 0.5)
    close(cs, am.sum(0), 1e-4, 2e-5 * N ** 0.5)


@pytest.mark.parametrize("N,M,K,act,xout,bias", [(3001, 100, 100, 1, 0, True), (2100, 150, 150, 0, 2, True), (1000, 64, 114, 1, 1, True),
                                                 (777, 150, 50, 2, 0, True), (5000, 128, 64, 0, 0, False), (64, 100, 100, 1, 1, True),
                                                 (4099, 160, 160, 2, 2, False), (1500, 34, 158, 1, 0, True)])
def test_dense_bwd_matches_torch(N, M, K, act, xout, bias):
    """The one-pass backward of a tall dense layer (mdl_dense_bwd: dW, db and dX from one read of g, y, x) vs the three
    separate fp32 products on the same bf16-rounded operands; act = derivative of the layer's own activation from its saved
    output, xout = derivative of the activation in FRONT of the layer from the layer's input (pre-activation hand-over)."""
    from matdeeplearn_amd import _lib
    gen = torch.Generator().manual_seed(N + M + K + act)
    d = dev()
    g = torch.randn(N, M, generator=gen).to(torch.bfloat16).to(d)
    pre = torch.randn(N, M, generator=gen) * 2
    y = (torch.relu(pre) if act == 1 else torch.nn.functional.softplus(pre) - 0.6931471805599453).to(torch.bfloat16).to(d)
    xin = torch.randn(N, K, generator=gen) * 2
    x = (torch.relu(xin) if xout == 1 else (torch.nn.functional.softplus(xin) - 0.6931471805599453 if xout == 2 else xin)).to(torch.bfloat16).to(d)
    w = (torch.randn(M, K, generator=gen) / M ** 0.5).to(torch.bfloat16).to(d)
    dx = torch.full((N, K), float("nan"), dtype=torch.bfloat16, device=d)
    dw, db = torch.zeros(M, K, device=d), torch.zeros(M, device=d)
    gmo = torch.full((N, M), float("nan"), dtype=torch.bfloat16, device=d) if N % 2 else None     # (every other case also asks for g')
    _lib.check(_lib.lib().mdl_dense_bwd(_lib.ptr(g), g.stride(0), M, _lib.ptr(y) if act else None, y.stride(0), act, _lib.ptr(x),
                                        x.stride(0), K, _lib.ptr(w), _lib.ptr(dx), dx.stride(0), xout, _lib.ptr(gmo), _lib.ptr(dw),
                                        _lib.ptr(db) if bias else None, N, _lib.MDL_BF16, _lib.stream()), "mdl_dense_bwd")

    def dact(code, out):
        if code == 1:
            return (out.float() > 0).float()
        if code == 2:
            return 1.0 - torch.exp(-(out.float() + 0.6931471805599453))
        return torch.ones_like(out, dtype=torch.float32)

    gm = (g.float() * dact(act, y)).to(torch.bfloat16).float()          # what threshold_backward / mdl_ssp_bwd would have written
    close(dw, gm.t() @ x.float(), 1e-4, 2e-5 * N ** 0.5)
    if bias:
        close(db, gm.sum(0), 1e-4, 2e-5 * N ** 0.5)
    close(dx, (gm @ w.float()) * dact(xout, x), 1e-2, 1e-2)
    if gmo is not None:
        assert torch.equal(gmo.float(), gm) or act == 2
        close(gmo, gm, 1e-2, 1e-3)


@pytest.mark.parametrize("tables", [0, 2])
def test_linear_relu_batchnorm_node_matches_torch_and_the_separate_layers(tables):
    """BatchNorm1d(relu(Linear(x) [+ gathered rows])) as one autograd node (ops.linear_relu_bn: the BatchNorm backward writes
    the pre-activation gradient, the one-pass dense backward takes it without an activation staging) (a) against fp32 torch on
    the same bf16-rounded operands, (b) against the same chain built from the separate product layers — the masked
    BatchNorm backward must reproduce `threshold_backward` of the unmasked one exactly, so dX agrees bit for bit and the
    reductions to accumulation order."""
    from matdeeplearn_amd import nn as mnn, ops
    d = dev()
    N, K, M, R = 6001, 100, 100, 257
    g = torch.Generator().manual_seed(40 + tables)
    r16 = lambda t: t.to(torch.bfloat16).float()
    x = r16(torch.randn(N, K, generator=g))
    W, b = r16(torch.randn(M, K, generator=g) / K ** 0.5), r16(torch.randn(M, generator=g) * 0.1)
    tabs = [r16(torch.randn(R, M, generator=g) * 0.5) for _ in range(tables)]
    ids = [torch.randint(0, R, (N,), generator=g) for _ in range(tables)]
    gamma, beta = torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g) * 0.2
    gout = torch.randn(N, M, generator=g)

    # (a) fp32 torch
    leaves = [t.clone().requires_grad_(True) for t in [x, W, b, gamma, beta] + tabs]
    xo, Wo, bo, go_, beo = leaves[:5]
    pre = torch.nn.functional.linear(xo, Wo, bo)
    for t, ix in zip(leaves[5:], ids):
        pre = pre + t[ix]
    ref = torch.nn.functional.batch_norm(torch.relu(pre), None, None, go_, beo, True, 0.1, 1e-5)
    (ref * gout).sum().backward()

    def run(fused):
        bn = mnn.BatchNorm1d(M).to(d)
        with torch.no_grad():
            bn.weight.copy_(gamma); bn.bias.copy_(beta)
        xd = x.to(d).to(torch.bfloat16).requires_grad_(True)
        Wd, bd = W.to(d).requires_grad_(True), b.to(d).requires_grad_(True)
        td = [t.to(d).to(torch.bfloat16).requires_grad_(True) for t in tabs]
        gath = [(t, ix.to(d).int()) for t, ix in zip(td, ids)] or None
        if fused:
            z = bn.after_linear_relu(xd, Wd, bd, None, gath)
            assert z is not None
        elif gath:
            z = bn(ops.linear_gather_act(xd, Wd, bd, "relu", gath))
        else:
            z = bn(ops.linear_act(xd, Wd, bd, "relu"))
        (z.float() * gout.to(d)).sum().backward()
        return [z.detach(), xd.grad, Wd.grad, bd.grad, bn.weight.grad, bn.bias.grad] + [t.grad for t in td] + [bn.running_mean.clone(), bn.running_var.clone()]

    with ops.deterministic():
        fu, se = run(True), run(False)
    fd = run(True)            # default mode: the BatchNorm statistics come out of the dense layer's epilogue (mdl_linear_act_stats)
    names = ["out", "dx", "dW", "db", "dgamma", "dbeta"] + ["dtab%d" % k for k in range(tables)] + ["running_mean", "running_var"]
    refs = [ref.detach(), xo.grad, Wo.grad, bo.grad, go_.grad, beo.grad] + [t.grad for t in leaves[5:]]
    for res in (fu, fd):
        for nm, a, r_ in zip(names, res, refs):                                     # (a)
            scale = float(r_.abs().max())
            err = float((a.float().cpu() - r_).abs().max())
            assert err <= 5e-2 * max(scale, 0.1 * float(xo.grad.abs().max())), (nm, err, scale)
    close(fd[-2], fu[-2], 1e-4, 1e-5)                                               # running statistics: epilogue sums vs statistics kernel
    close(fd[-1], fu[-1], 1e-4, 1e-5)
    for nm, a, c in zip(names, fu, se):                                             # (b)
        if nm in ("out", "dx", "running_mean", "running_var"):
            assert torch.equal(a, c), nm
        else:
            close(a, c, 2e-3, 2e-3 * float(c.float().abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("reduce", ["mean", "sum"])
def test_residual_scatter_node_matches_the_two_ops(dtype, reduce):
    """ops.residual_scatter = (src + res, scatter(src, index)) with the two gradients of `src` summed inside the reduction's
    backward (mdl_segment_reduce_bwd_add): values and gradients against index_add_ in fp32, and against the separate ops
    (fp32: bit-equal — same arithmetic, one pass less)."""
    from matdeeplearn_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(77)
    E, C, N = 5003, 100, 311
    index = torch.randint(0, N, (E,), generator=g)                       # unsorted: the by-source index of a target-sorted edge list
    src0, res0 = torch.randn(E, C, generator=g).to(dtype), torch.randn(E, C, generator=g).to(dtype)
    g1, g2 = torch.randn(E, C, generator=g).to(dtype), torch.randn(N, C, generator=g).to(dtype)

    def run(fused):
        src, res = src0.to(d).requires_grad_(True), res0.to(d).requires_grad_(True)
        idx = index.to(d)
        if fused:
            s, v = ops.residual_scatter(src, res, idx, N, reduce)
        else:
            s, v = src + res, ops.scatter(src, idx, 0, N, reduce)
        ((s * g1.to(d)).sum() + (v * g2.to(d)).sum()).backward()
        return s.detach(), v.detach(), src.grad, res.grad

    fu, se = run(True), run(False)
    for a, b in zip(fu, se):
        if dtype == torch.float32:
            assert torch.equal(a, b)
        else:                     # (bf16: the separate ops round the scattered gradient before the sum, the node rounds once)
            close(a, b, 1e-2, 1e-2)
    sf, rf = src0.float().requires_grad_(True), res0.float().requires_grad_(True)
    v = torch.zeros(N, C).index_add_(0, index, sf)
    if reduce == "mean":
        v = v / torch.bincount(index, minlength=N).clamp(min=1).unsqueeze(1)
    (((sf + rf) * g1.float()).sum() + (v * g2.float()).sum()).backward()
    tol = (1e-5, 1e-5) if dtype == torch.float32 else (2e-2, 2e-2)
    close(fu[0], sf.detach() + rf.detach(), *tol)
    close(fu[1], v.detach(), *tol)
    close(fu[2], sf.grad, *tol)
    close(fu[3], rf.grad, *tol)


@pytest.mark.parametrize("mid", [150, 160])
@pytest.mark.parametrize("x_grad", [False, True])
def test_dense_chain_hands_the_activation_derivative_down(x_grad, mid):
    """nn._seq on Linear -> ShiftedSoftplus -> Linear -> ReLU -> Linear in bf16: between fused layers the later layer's backward
    returns the gradient w.r.t. the earlier layer's pre-activation (mdl_dense_bwd's xout) and the earlier one applies no
    derivative; output and all gradients against fp32 torch on bf16-rounded weights, with and without an input gradient
    (SchNet's filter network has none).  mid = 160: with its bias column the second layer's input is one column wider than the one-pass backward takes, so
    the hand-over runs on the fallback path (streaming dX kernel + TN GEMM, the derivative applied to dX afterwards)."""
    from matdeeplearn_amd import nn as mnn, ops
    d = dev()
    torch.manual_seed(12)
    N = 3001
    seq = torch.nn.Sequential(torch.nn.Linear(50, mid), mnn.ShiftedSoftplus(), torch.nn.Linear(mid, 150 if mid == 150 else 128), torch.nn.ReLU(),
                              torch.nn.Linear(150 if mid == 150 else 128, 64))
    with torch.no_grad():
        for p_ in seq.parameters():
            p_.copy_(p_.to(torch.bfloat16).float())
    ref = [p_.detach().clone().requires_grad_(True) for p_ in seq.parameters()]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, 50, generator=g).to(torch.bfloat16).float()
    gout = torch.randn(N, 64, generator=g)
    xr = x.clone().requires_grad_(x_grad)
    h = torch.nn.functional.softplus(torch.nn.functional.linear(xr, ref[0], ref[1])) - 0.6931471805599453
    pre2 = torch.nn.functional.linear(h, ref[2], ref[3])
    yr = torch.nn.functional.linear(torch.relu(pre2), ref[4], ref[5])
    (yr * gout).sum().backward()
    seq.to(d)

    def run(handed):
        for p_ in seq.parameters():
            p_.grad = None
        xd = x.to(d).to(torch.bfloat16).requires_grad_(x_grad)
        if handed:
            y = mnn._seq(seq, xd)
        else:                                             # the same fused layers, every one applying its own derivative
            h1 = ops.linear_act(xd, seq[0].weight, seq[0].bias, "ssp")
            h2 = ops.linear_act(h1, seq[2].weight, seq[2].bias, "relu")
            y = ops.linear_act(h2, seq[4].weight, seq[4].bias, None)
        (y.float() * gout.to(d)).sum().backward()
        return [y.detach()] + [p_.grad.clone() for p_ in seq.parameters()] + ([xd.grad] if x_grad else [])

    ha, se = run(True), run(False)
    refs = [yr.detach()] + [r.grad for r in ref] + ([xr.grad] if x_grad else [])
    fro = lambda a, r: float((a.float().cpu() - r.float().cpu()).norm() / r.float().norm().clamp(min=1e-6))
    assert ha[0].dtype == torch.bfloat16 and torch.equal(ha[0], se[0])
    # against fp32 torch: relative Frobenius error — a pre-activation within bf16 rounding of the ReLU kink flips its mask (a
    # fraction f of the elements, error ~ sqrt(f): a few per cent here), not a kernel property; against the chain without the
    # hand-over (same forward, same masks) only the rounding of the intermediate gradients differs
    for k, (a, r) in enumerate(zip(ha, refs)):
        assert fro(a, r) <= 8e-2, (k, fro(a, r))
    for k, (a, c) in enumerate(zip(ha, se)):
        assert fro(a, c) <= 1e-2, (k, fro(a, c))


@pytest.mark.parametrize("K", [300, 420])
def test_wide_input_linear_takes_its_weight_gradient_from_two_tn_gemms(K):
    """ops.linear with 256 < in <= 512 (MEGNet's node block: 3 x 100 concatenated columns): library forward, dW as two
    TN-GEMM column halves (+ db from the first); vs fp32 torch on the same bf16-rounded operands."""
    from matdeeplearn_amd import ops
    N, M = 5000, 100
    g = torch.Generator().manual_seed(K)
    x = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev()).requires_grad_(True)
    w = (torch.randn(M, K, generator=g) / K ** 0.5).to(dev()).requires_grad_(True)
    b = (torch.randn(M, generator=g) * 0.1).to(dev()).requires_grad_(True)
    go = torch.randn(N, M, generator=g).to(dev())
    y = ops.linear(x, w, b)
    (y.float() * go).sum().backward()
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = b.detach().to(torch.bfloat16).float().requires_grad_(True)
    yr = torch.nn.functional.linear(xr, wr, br)
    (yr * go.to(torch.bfloat16).float()).sum().backward()
    close(y, yr, 1e-2, 1e-2)
    close(w.grad, wr.grad, 3e-2, 3e-2)
    close(b.grad, br.grad, 3e-2, 3e-2)
    close(x.grad, xr.grad, 3e-2, 2e-2)


@pytest.mark.parametrize("M,K", [(300, 100), (192, 64), (450, 150)])
def test_wide_output_linear_takes_its_weight_gradient_from_transposed_tn_gemms(M, K):
    """ops.linear with many outputs and few inputs (the GRU gate matrices of MPNN, [3C, C]): library forward and dX, dW as
    TN GEMMs of the transposed product x^T g in column chunks; vs fp32 torch on the same bf16-rounded operands."""
    from matdeeplearn_amd import ops
    N = 5003
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev()).requires_grad_(True)
    w = (torch.randn(M, K, generator=g) / K ** 0.5).to(dev()).requires_grad_(True)
    b = (torch.randn(M, generator=g) * 0.1).to(dev()).requires_grad_(True)
    go = torch.randn(N, M, generator=g).to(dev())
    y = ops.linear(x, w, b)
    (y.float() * go).sum().backward()
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = b.detach().to(torch.bfloat16).float().requires_grad_(True)
    yr = torch.nn.functional.linear(xr, wr, br)
    (yr * go.to(torch.bfloat16).float()).sum().backward()
    close(y, yr, 1e-2, 1e-2)
    close(w.grad, wr.grad, 1e-3, 1e-3 * float(wr.grad.abs().max()))
    close(b.grad, br.grad, 3e-2, 3e-2)
    close(x.grad, xr.grad, 3e-2, 2e-2)


@pytest.mark.parametrize("act", ["relu", "ssp"])
def test_linear_act_without_input_grad(act):
    """A fused dense layer whose input needs no gradient (SchNet's filter network on the edge features): the backward
    takes the activation derivative into the TN GEMM; weights / bias gradients vs fp32 torch."""
    from matdeeplearn_amd import ops
    N, K, M = 4000, 50, 150
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev())
    w = (torch.randn(M, K, generator=g) / K ** 0.5).to(dev()).requires_grad_(True)
    b = (torch.randn(M, generator=g) * 0.1).to(dev()).requires_grad_(True)
    go = torch.randn(N, M, generator=g).to(dev())
    y = ops.linear_act(x, w, b, act)
    (y.float() * go).sum().backward()
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = b.detach().to(torch.bfloat16).float().requires_grad_(True)
    yr = torch.nn.functional.linear(x.float(), wr, br)
    yr = torch.relu(yr) if act == "relu" else torch.nn.functional.softplus(yr) - 0.6931471805599453
    (yr * go.to(torch.bfloat16).float()).sum().backward()
    close(y, yr, 1e-2, 4e-3)
    close(w.grad, wr.grad, 3e-2, 3e-2)
    close(b.grad, br.grad, 3e-2, 3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C", [(1000, 64), (37, 64), (5000, 128), (1, 32), (999, 16), (6144, 100),
                                 (20000, 64), (6145, 100), (9000, 16)])
def test_batchnorm_train_matches_torch(dtype, N, C):
    """HIP BatchNorm1d (training mode) vs torch.nn.BatchNorm1d on CPU: output, running stats, all gradients."""
    from matdeeplearn_amd import nn as mnn
    g = torch.Generator().manual_seed(N + C)
    x = (torch.randn(N, C, generator=g) * 2 + 3).to(dtype).float()
    go = torch.randn(N, C, generator=g).to(dtype).float()
    ref = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        ref.weight.copy_(torch.rand(C, generator=g) + 0.5)
        ref.bias.copy_(torch.randn(C, generator=g))
    mine = mnn.BatchNorm1d(C)
    mine.load_state_dict(ref.state_dict())
    mine.to(dev())
    if N == 1:
        return   # torch refuses batch statistics of a single row in training mode
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    (yr * go).sum().backward()
    xd = x.to(dev()).to(dtype).requires_grad_(True)
    yd = mine(xd)
    assert yd.dtype == dtype
    (yd.float() * go.to(dev())).sum().backward()
    tol = (2e-5, 2e-5) if dtype == torch.float32 else (3e-2, 2e-2)
    close(yd, yr, *tol)
    close(xd.grad, xr.grad, *tol)
    close(mine.weight.grad, ref.weight.grad, *tol)
    close(mine.bias.grad, ref.bias.grad, *tol)
    close(mine.running_mean, ref.running_mean, 1e-4 if dtype == torch.float32 else 1e-2, 1e-5 if dtype == torch.float32 else 1e-2)
    close(mine.running_var, ref.running_var, 1e-4 if dtype == torch.float32 else 2e-2, 1e-5 if dtype == torch.float32 else 1e-2)
    assert int(mine.state_dict()["num_batches_tracked"]) == 1      # counted on the host, folded in when the state is read


@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
def test_hip_batch_assembly_matches_tensor_op_assembly(x_dtype):
    """K8 (one launch per batch) vs the index-arithmetic assembly checked on CPU against per-graph concatenation."""
    from matdeeplearn_amd.process import synthetic_bulk
    ds = synthetic_bulk(200, seed=11).to(dev())
    ids = np.random.default_rng(0).choice(200, size=57, replace=False)
    a, dna = ds.assemble(ids)
    b, dnb = ds.assemble_hip(ids, x_dtype)
    assert torch.equal(a.x.to(x_dtype), b.x) and torch.equal(a.batch, b.batch) and torch.equal(a.y, b.y)
    assert torch.equal(a.csr.rowptr, b.csr.rowptr) and torch.equal(a.csr.src, b.csr.src) and torch.equal(a.csr.tgt, b.csr.tgt)
    assert torch.equal(a.edge_weight, b.edge_weight) and torch.equal(dna, dnb)
    assert (a.num_nodes, a.num_edges) == (b.num_nodes, b.num_edges)
    full = ds.collate(ids, edge_dtype=torch.float32)
    assert full.edge_attr.shape == (b.num_edges, 50)


@pytest.mark.parametrize("name", ["l1_loss", "mse_loss"])
@pytest.mark.parametrize("n", [1, 100, 8192, 20000])
def test_fused_loss_matches_torch(name, n):
    """ops.loss (value + gradient in one launch) vs F.l1_loss / F.mse_loss and their autograd, incl. exact ties (gradient 0)."""
    from matdeeplearn_amd import ops
    g = torch.Generator().manual_seed(n)
    p = torch.randn(n, generator=g)
    y = torch.randn(n, generator=g)
    if n > 4:
        y[3] = p[3]
    pd = p.to(dev()).requires_grad_(True)
    out = ops.loss(name, pd, y.to(dev()))
    (out * 3.0).backward()
    pr = p.clone().requires_grad_(True)
    ref = getattr(torch.nn.functional, name)(pr, y)
    (ref * 3.0).backward()
    close(out, ref, 1e-5, 1e-6)
    close(pd.grad, pr.grad, 1e-6, 1e-9)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("E", [1, 127, 128, 129, 100003])
def test_rbf_block_kernel_matches_oracle_on_ragged_sizes(dtype, E):
    """K1's 128-edges-per-block kernel (G = 50) vs the oracle expansion: block tails, a single edge, many blocks."""
    from matdeeplearn_amd import ops
    g = torch.Generator().manual_seed(E)
    d = torch.rand(E, generator=g)
    ref = oops.rbf_expand(d, 0.0, 1.0, 50, 0.2)
    out = ops.rbf_expand(d.to(dev()), 0.0, 1.0, 50, 0.2, out_dtype=dtype)
    assert out.shape == (E, 50)
    if dtype == torch.float32:
        assert torch.allclose(out.cpu(), ref, rtol=1e-6, atol=1e-7)
    else:
        assert torch.allclose(out.float().cpu(), ref.to(torch.bfloat16).float(), rtol=1e-2, atol=1e-6)


def test_cfconv_backward_in_one_walk_matches_the_pair():
    """mdl_gather_mul_reduce_dw (gradient w.r.t. the gathered rows AND the filter gradient from one walk over the by-source
    CSR) against mdl_gather_mul_reduce on the transposed CSR + mdl_edge_mul: same arithmetic per element."""
    from matdeeplearn_amd import _lib, ops
    d = dev()
    L, P, st = _lib.lib(), _lib.ptr, _lib.stream
    g = torch.Generator().manual_seed(11)
    n, F, dt = 3000, 150, _lib.MDL_BF16
    ei = rand_graph(n, 13, sort=True, empty_frac=0.1)
    E = ei.shape[1]
    csr = ops.build_csr(ei.to(d), n, assume_sorted=True)
    h = torch.randn(n, F, generator=g).to(d).to(torch.bfloat16)
    w = torch.randn(E, F, generator=g).to(d).to(torch.bfloat16)
    go = torch.randn(n, F, generator=g).to(d).to(torch.bfloat16)
    c = torch.rand(E, generator=g).to(d)
    rowptr_s, col_s, eid_s, _ = csr.transposed()
    dh0, dw0 = torch.empty_like(h), torch.empty_like(w)
    _lib.check(L.mdl_gather_mul_reduce(P(go), P(w), P(c), P(rowptr_s), P(col_s), P(eid_s), P(dh0), n, F, _lib.MDL_SUM, dt, st()), "gmr T")
    _lib.check(L.mdl_edge_mul(P(h), P(csr.row), P(go), P(csr.col), P(c), P(dw0), E, F, dt, st()), "edge_mul")
    dh1, dw1 = torch.empty_like(h), torch.full_like(w, float("nan"))
    _lib.check(L.mdl_gather_mul_reduce_dw(P(go), P(w), P(c), P(rowptr_s), P(col_s), P(eid_s), P(dh1), P(h), P(dw1), n, F, dt, st()),
               "gmr dw")
    assert torch.equal(dh0, dh1)
    close(dw1, dw0, 1e-2, 1e-3)          # (a * b) * c against (a * c) * b before the bf16 rounding


@pytest.mark.parametrize("shape", [(3000 + 17, 64, (64, 64, 64, 1)), (700, 50, (32, 2)), (1000, 64, (40,)), (257, 16, (64, 64, 8)),
                                   (101, 64, (64, 64, 64, 1)), (37, 64, (64, 1)), (1, 64, (64, 64, 1))])     # the reference's batch size and below
def test_fused_post_fc_head_matches_the_layer_by_layer_path(shape):
    """csrc/mlp.hip (post_lin_list + lin_out as one launch per direction) against the same chain on the streaming dense layers
    and against an fp32 torch reference: outputs, input gradient, every weight / bias gradient; ragged row counts, a
    1-column output, widths below 64."""
    from matdeeplearn_amd import ops
    from matdeeplearn_amd.models._base import dense, dense_act
    d = dev()
    N, K0, widths = shape
    g = torch.Generator().manual_seed(N + K0)
    lins, k = [], K0
    for m in widths:
        lin = torch.nn.Linear(k, m)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(m, k, generator=g) * (1.5 / k ** 0.5))
            lin.bias.copy_(torch.randn(m, generator=g) * 0.2)
        lins.append(lin.to(d))
        k = m
    x0 = torch.randn(N, K0, generator=g).to(d).to(torch.bfloat16)
    gy = torch.randn(N, widths[-1], generator=g).to(d).to(torch.bfloat16)
    assert ops.mlp_head_ok(x0, lins, "relu")
    res = []
    for mode in ("fused", "layers", "fp32"):
        for lin in lins:
            lin.zero_grad(set_to_none=True)
        x = (x0.float() if mode == "fp32" else x0).clone().requires_grad_(True)
        if mode == "fused":
            y = ops.mlp_head(x, lins)
        elif mode == "layers":
            h = x
            for lin in lins[:-1]:
                h = dense_act(lin, h, "relu")
            y = dense(lins[-1], h)
        else:
            h = x
            for lin in lins[:-1]:
                h = torch.relu(lin(h))
            y = lins[-1](h)
        (y.float() * gy.float()).sum().backward()
        res.append((y.detach().float(), x.grad.float(), [lin.weight.grad.float().clone() for lin in lins],
                    [lin.bias.grad.float().clone() for lin in lins]))
    (yf, dxf, dwf, dbf), (yl, dxl, dwl, dbl), (yr, dxr, dwr, dbr) = res
    def frob(a, c):
        return float((a - c).norm() / (c.norm() + 1e-30))
    close(yf, yl, 2e-2, 2e-2)
    close(yf, yr, 3e-2, 3e-2)
    close(dxf, dxl, 3e-2, 3e-2)
    # against fp32 a ReLU whose pre-activation rounds across zero in bf16 flips whole gradient contributions (7 % of the norm
    # after three layers, in the layer-by-layer path just the same): the fused path must be no farther from fp32 than that one
    assert frob(dxf, dxr) <= max(3e-2, 1.2 * frob(dxl, dxr))
    for a, b, c in zip(dwf + dbf, dwl + dbl, dwr + dbr):
        close(a, b, 3e-2, 3e-2)
        assert frob(a, c) <= max(3e-2, 1.2 * frob(b, c))


@pytest.mark.parametrize("shape", [(101, 64, (64, 64, 64, 1)), (8193, 64, (64, 64, 64, 1)), (257, 16, (64, 8)), (1, 64, (64, 1))])
def test_fused_head_with_fp32_prediction_equals_the_bf16_head_and_its_casts(shape):
    """MDL_MLP_F32_IO (round 6): the head's last output as fp32 rows and its gradient read as fp32 rows — bit-identical to the
    bf16 head with `.float()` behind it and the bf16 cast of the gradient in front of its backward (the two launches it removes)."""
    from matdeeplearn_amd import ops
    d = dev()
    N, K0, widths = shape
    g = torch.Generator().manual_seed(7 * N + K0)
    lins, k = [], K0
    for m in widths:
        lin = torch.nn.Linear(k, m)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(m, k, generator=g) * (1.5 / k ** 0.5))
            lin.bias.copy_(torch.randn(m, generator=g) * 0.2)
        lins.append(lin.to(d))
        k = m
    x0 = torch.randn(N, K0, generator=g).to(d).to(torch.bfloat16)
    gy = torch.randn(N, widths[-1], generator=g).to(d)                        # an fp32 gradient (what the fp32 loss hands back)
    res = []
    with ops.deterministic():                                                 # one workgroup: the weight-gradient sums in one order
        for f32 in (False, True):
            for lin in lins:
                lin.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            y = ops.mlp_head(x, lins, f32_out=f32)
            assert y.dtype == (torch.float32 if f32 else torch.bfloat16)
            y.float().backward(gy)
            res.append((y.detach().float(), x.grad.clone(), [lin.weight.grad.clone() for lin in lins], [lin.bias.grad.clone() for lin in lins]))
    (y0, dx0, dw0, db0), (y1, dx1, dw1, db1) = res
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    for a, b in zip(dw0 + db0, dw1 + db1):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["l1_loss", "mse_loss"])
@pytest.mark.parametrize("n,total", [(100, 101), (1, 2), (8192, 8193), (300, 300), (5, 4000)])
def test_fused_loss_over_the_first_rows_matches_the_sliced_loss(name, n, total):
    """ops.loss(..., rows=n) — mdl_loss_fwd_bwd_rows: the loss of pred[:n] with exact zeros as the gradient of the rows behind
    (the padded static batch's dummy graph) — against torch's loss of the slice; with the constant unit root gradient
    (ops.backward) and with a scaled one."""
    from matdeeplearn_amd import ops
    g = torch.Generator().manual_seed(n + total)
    p = torch.randn(total, generator=g)
    y = torch.randn(n, generator=g)
    if n > 4:
        y[3] = p[3]
    for scale in (None, 3.0):
        pd = p.to(dev()).requires_grad_(True)
        out = ops.loss(name, pd, y.to(dev()), rows=n)
        if scale is None:
            ops.backward(out)
        else:
            (out * scale).backward()
        pr = p.clone().requires_grad_(True)
        ref = getattr(torch.nn.functional, name)(pr[:n], y)
        (ref * (scale or 1.0)).backward()
        close(out, ref, 1e-5, 1e-6)
        close(pd.grad, pr.grad, 1e-6, 1e-9)
        assert not pd.grad[n:].any()
    # the unit root gradient is recognised by its address only: an equal-valued tensor takes the multiply and gives the same values
    pd = p.to(dev()).requires_grad_(True)
    ops.loss(name, pd, y.to(dev()), rows=n).backward(gradient=torch.ones((), device=dev()))
    close(pd.grad, pr.grad / (scale or 1.0), 1e-6, 1e-9)
    assert float(ops.unit_grad(dev())) == 1.0


@pytest.mark.parametrize("by_source", [False, True])
@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
def test_padded_batch_assembly_in_one_launch_matches_the_separate_launches(by_source, x_dtype):
    """mdl_assemble_batch_padded (K8 + both tail paddings + the pooling index's int32 segment ids as ONE launch: what
    StaticBatch.assemble runs inside the captured step) against mdl_assemble_batch, mdl_pad_batch_tail, mdl_pad_edge_tail and the
    int64 -> int32 copy it replaces — every buffer bit-equal, stale contents of an earlier, larger batch overwritten."""
    from matdeeplearn_amd import _lib
    from matdeeplearn_amd.process import StaticBatch, synthetic_bulk
    d = dev()
    ds = synthetic_bulk(300, seed=5).to(d)
    B = 37
    order = np.argsort(ds.node_ptr[1:] - ds.node_ptr[:-1])
    big = order[-B:]
    n_cap = int((ds.node_ptr[big + 1] - ds.node_ptr[big]).sum()) + 77
    e_cap = int((ds.edge_ptr[1:] - ds.edge_ptr[:-1]).max()) * B + 301
    sb = StaticBatch(ds, B, n_cap, e_cap, x_dtype=x_dtype, edge_dtype=x_dtype, by_source=by_source)
    rng = np.random.default_rng(3)
    p = _lib.ptr
    for ids in (order[-B:], order[:B], rng.choice(300, size=B, replace=False)):      # a large batch first: its tail must not survive
        assert sb.fits(ids)
        sb.load(ids)
        batch = sb.assemble()
        torch.cuda.synchronize()
        dd = ds._dev
        x = torch.zeros_like(sb.x); bi = torch.full_like(sb.batch_idx, -7); rp = torch.full_like(sb.rowptr, -7)
        src = torch.full_like(sb.src, -7); tgt = torch.full_like(sb.tgt, -7); ew = torch.zeros_like(sb.ew); dn = torch.zeros_like(sb.dn)
        y = torch.zeros_like(sb.y)
        cs, es, ss = (torch.full_like(sb.col_s, -7) for _ in range(3))
        ids_d, noff_d, eoff_d = sb.pack[:B], sb.pack[B:2 * B + 1], sb.pack[2 * B + 1:]
        _lib.check(_lib.lib().mdl_assemble_batch(
            p(ids_d), p(noff_d), p(eoff_d), p(dd["node_ptr"]), p(dd["edge_ptr"]), p(dd["x"]), p(dd["src"]), p(dd["tgt"]),
            p(dd["dist"]), p(dd["dist_norm"]), p(dd["lrowptr"]), p(dd["y"]), p(x), p(bi), p(rp), p(src), p(tgt), p(ew), p(dn), p(y),
            B, ds.num_features, ds.y.shape[1], int(ds.target_index), _lib.dtype_code(x), _lib.stream()), "mdl_assemble_batch")
        _lib.check(_lib.lib().mdl_pad_batch_tail(p(noff_d), p(eoff_d), B, n_cap, p(rp), p(bi), _lib.stream()), "mdl_pad_batch_tail")
        _lib.check(_lib.lib().mdl_pad_edge_tail(p(noff_d), p(eoff_d), B, n_cap, e_cap, p(src), p(tgt), p(cs) if by_source else None,
                                                p(es) if by_source else None, p(ss) if by_source else None, _lib.stream()),
                   "mdl_pad_edge_tail")
        N, E = sb.true_nodes, sb.true_edges
        assert torch.equal(sb.x[:N], x[:N]) and torch.equal(sb.batch_idx, bi) and torch.equal(sb.rowptr, rp) and torch.equal(sb.y, y)
        assert torch.equal(sb.src, src) and torch.equal(sb.tgt, tgt) and torch.equal(sb.ew[:E], ew[:E]) and torch.equal(sb.dn[:E], dn[:E])
        assert torch.equal(sb.pool_seg, bi.to(torch.int32))
        assert int(sb.rowptr[N]) == E and int(sb.rowptr[-1]) == E and int(sb.batch_idx[N]) == B and int(sb.src[-1]) == min(N, n_cap - 1)
        if by_source:
            assert torch.equal(sb.col_s[E:], cs[E:]) and torch.equal(sb.eid_s[E:], es[E:]) and torch.equal(sb.src_s[E:], ss[E:])
            assert int(sb.rowptr_s[N]) == E and int(sb.rowptr_s[-1]) == E
        assert batch.pool_index.seg.data_ptr() == sb.pool_seg.data_ptr()


# ---------------------------------------------------------------------------------------------
# BatchNorm statistics formed by the PRODUCER of the rows (round 5): the CGConv forward's epilogue (mdl_cgconv_fwd_ex) and
# the dense layer's (mdl_linear_act_stats), about a per-column shift, normalised by mdl_bn_apply_n(MDL_BN_SHIFT_ROW)
# ---------------------------------------------------------------------------------------------
def _bn_fp64(y, gamma, beta, eps=1e-5):
    y = y.double()
    m, v = y.mean(0), y.var(0, unbiased=False)
    return ((y - m) / torch.sqrt(v + eps) * gamma.double() + beta.double()), m, y.var(0, unbiased=True)


@pytest.mark.parametrize("C,n,shift_kind", [(64, 3000, "beta"), (64, 517, "none"), (32, 1200, "beta"), (64, 2000, "far")])
def test_cgconv_forward_forms_the_batchnorm_statistics_in_its_epilogue(C, n, shift_kind, monkeypatch):
    """nn.CGConv(x, ..., bn=BatchNorm1d) (cgcnn.py:136-145: conv -> bn) with the statistics out of the conv kernel's epilogue,
    against (a) the same layer followed by the separate statistics kernel and (b) an fp64 BatchNorm of the conv kernel's own
    (bf16) output: normalised rows, running statistics, every gradient.  Isolated nodes (their rows are copied, not computed)
    are part of the statistics; `far`: columns whose mean sits 300 standard deviations from zero — the cancellation case the
    shifted sums exist for (with the previous layer's beta as the shift the sums stay well conditioned)."""
    from matdeeplearn_amd import nn as mnn, ops
    monkeypatch.setattr(ops, "_CG_BN_STATS", True)          # (opt-in: measured time-neutral on the bench batch)
    d = dev()
    G = 50
    g = torch.Generator().manual_seed(n + C)
    ei = rand_graph(n, n, sort=True, empty_frac=0.1)
    E = ei.shape[1]
    off = 30.0 if shift_kind == "far" else 0.0
    x = (torch.randn(n, C, generator=g) * 0.1 + off).to(torch.bfloat16)
    ea = torch.rand(E, G, generator=g).to(torch.bfloat16)
    gout = torch.randn(n, C, generator=g)
    conv = mnn.CGConv(C, G, aggr="mean").to(d)
    csr = ops.build_csr(ei.to(d), n, assume_sorted=True)
    shift = None if shift_kind == "none" else torch.full((C,), off, device=d) + (0.0 if shift_kind == "far" else 0.05)

    def run(fused):
        bn = mnn.BatchNorm1d(C).to(d)
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.2, 0.2, C))
        conv.zero_grad(set_to_none=True)
        xd = x.to(d).requires_grad_(True)
        if fused:
            assert ops.cgconv_bn_stats_ok(xd, ea.to(d), csr)
            z = conv(xd, None, ea.to(d), csr=csr, bn=bn, bn_shift=shift)
        else:
            z = bn(conv(xd, None, ea.to(d), csr=csr))
        (z.float() * gout.to(d)).sum().backward()
        return z.detach(), xd.grad, conv.lin_f.weight.grad.clone(), conv.lin_s.weight.grad.clone(), bn.weight.grad, bn.bias.grad, \
            bn.running_mean.clone(), bn.running_var.clone(), bn

    fu, se = run(True), run(False)
    with torch.no_grad():
        y = conv(x.to(d), None, ea.to(d), csr=csr)                      # the conv kernel's own bf16 output
    ref, m64, v64 = _bn_fp64(y.float().cpu(), fu[8].weight.detach().cpu(), fu[8].bias.detach().cpu())
    for res, what in ((fu, "epilogue sums"), (se, "statistics kernel")):
        close(res[0], ref.float(), 2e-2, 1e-2)                                        # bf16 output of an fp32 normalisation
        rm = res[6].cpu().double() / 0.1                                              # momentum 0.1 from running_mean 0
        assert float((rm - m64).abs().max()) <= 1e-5 * (1.0 + float(m64.abs().max())), what
        rv = (res[7].cpu().double() - 0.9) / 0.1
        assert float((rv - v64).abs().max()) <= 2e-3 * float(v64.abs().max()) + 1e-7, (what, float((rv - v64).abs().max()), float(v64.abs().max()))
    for k, nm in enumerate(["out", "dx", "dW_f", "dW_s", "dgamma", "dbeta"]):
        close(fu[k], se[k], 3e-2, 3e-2)


def test_producer_side_batchnorm_sums_survive_a_large_mean_small_variance_column():
    """(advisor, round 4) a post-ReLU column whose mean is far above its spread over 2e5 rows: plain sum x / sum x^2 in fp32
    cancel (E[x^2] - mean^2), the sums about the producer's shift row do not.  Linear -> ReLU -> BatchNorm with the statistics in
    the dense layer's epilogue against an fp64 BatchNorm of the same (bf16) activations."""
    from matdeeplearn_amd import nn as mnn, ops
    d = dev()
    N, K, M = 200000, 64, 64
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(N, K, generator=g) * 0.25).to(torch.bfloat16)
    W = (torch.randn(M, K, generator=g) * 0.05)
    b = torch.full((M,), 40.0)                            # relu(x W^T + b) = 40 +- 0.1, stored in bf16 steps of 0.25: mostly 40.0,
    b[::2] = 0.1                                          # some 39.75 / 40.25 — mean^2 / var ~ 1e5; ... next to ordinary columns
    bn = mnn.BatchNorm1d(M).to(d)
    xd = x.to(d).requires_grad_(True)
    Wd, bd = W.to(d).requires_grad_(True), b.to(d).requires_grad_(True)
    z = bn.after_linear_relu(xd, Wd, bd, None, None)
    assert z is not None
    with torch.no_grad():
        y = ops.linear_act(x.to(d), W.to(d), b.to(d), "relu")          # the same activations, as stored (bf16)
    ref, m64, v64 = _bn_fp64(y.float().cpu(), torch.ones(M), torch.zeros(M))
    rm = bn.running_mean.cpu().double() / 0.1
    rv = (bn.running_var.cpu().double() - 0.9) / 0.1
    assert float((rm - m64).abs().max()) <= 1e-5 * (1.0 + float(m64.abs().max()))
    # the variance of the large-mean columns is ~1e-3 (bf16 steps of 0.25 at 40): it must come out to a few per cent, not as
    # rounding noise of a 1600-sized square
    assert float(v64[1]) > 1e-4 and float(m64[1]) ** 2 / float(v64[1]) > 3e4           # (the case is what it claims to be)
    err = (rv - v64).abs() - 5e-2 * v64
    assert float(err.max()) <= 1e-6, (float(err.max()), v64[:4].tolist(), rv[:4].tolist())
    # (normalised rows: the ordinary columns; in the large-mean columns one bf16 step of the activation — 0.25 at 40 — is two
    # standard deviations, so two launches that round one pre-activation differently are not comparable element by element)
    close(z[:, ::2], ref[:, ::2].float(), 5e-2, 2e-2)


def test_weight_gradients_written_in_place_match_the_assembled_ones(monkeypatch):
    """MdlCgConv.ld_dwe / MdlCgNode.ld_dwn (opt-in, ops._DIRECT_GRADS): K3 and K3c add their partial sums straight into the two
    Linears' stacked weight gradient [2C, 2C + G] instead of into staging buffers that mdl_cgconv_assemble_grads re-lays out —
    same gradients up to the order of the fp32 atomic adds."""
    from matdeeplearn_amd import ops
    d = dev()
    n, C, G = 1500, 64, 50
    g = torch.Generator().manual_seed(77)
    ei = rand_graph(n, 77, sort=True, empty_frac=0.05)
    E = ei.shape[1]
    x = torch.randn(n, C, generator=g).to(torch.bfloat16)
    ea = torch.rand(E, G, generator=g).to(torch.bfloat16)
    wf, ws = torch.randn(C, 2 * C + G, generator=g) * 0.1, torch.randn(C, 2 * C + G, generator=g) * 0.1
    bf, bs = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    gout = torch.randn(n, C, generator=g)
    csr = ops.build_csr(ei.to(d), n, assume_sorted=True)
    res = []
    for direct in (False, True):
        monkeypatch.setattr(ops, "_DIRECT_GRADS", direct)
        xd = x.to(d).requires_grad_(True)
        ps = [t.to(d).clone().requires_grad_(True) for t in (wf, bf, ws, bs)]
        out = ops.cgconv(xd, None, ea.to(d), ps[0], ps[1], ps[2], ps[3], "mean", csr=csr)
        (out.float() * gout.to(d)).sum().backward()
        res.append([xd.grad] + [p.grad for p in ps])
    close(res[1][0], res[0][0], 2e-2, 2e-2)                    # dx: the same kernels either way (by-source sums: bf16 atomics, order-dependent)
    # the weight gradients: the x_j columns are r_src^T x with r_src summed by bf16 atomics, so two runs of the SAME mode differ
    # by ~1e-3 of the gradient scale (observed 1.0e-3 on a 282-pass box); a wrong row/column offset would be an O(1) error
    for a, b in zip(res[0][1:], res[1][1:]):
        close(b, a, 1e-3, 5e-3)


def _cfconv_case(n, F, seed, empty_frac=0.1, max_in=20):
    from matdeeplearn_amd import nn as mnn, ops
    d = dev()
    g = torch.Generator().manual_seed(seed)
    ei = rand_graph(n, seed, sort=True, empty_frac=empty_frac, max_in=max_in)
    E = ei.shape[1]
    csr = ops.build_csr(ei.to(d), n, assume_sorted=True)
    dist = torch.rand(E, generator=g) * 7.5
    rbf = torch.exp(-((dist.view(-1, 1) / 8.0 - torch.linspace(0, 1, 50).view(1, -1)) ** 2) / 0.2 ** 2)
    conv = mnn.InteractionBlock(100, 50, F, 8.0)
    with torch.no_grad():
        for m in (conv.mlp[0], conv.mlp[2]):
            m.bias.copy_(torch.randn(F, generator=g) * 0.2)
    conv = conv.to(d)
    x = (torch.randn(n, 100, generator=g) * 0.5).to(torch.bfloat16).to(d)
    return conv, x, ei.to(d), dist.to(d), rbf.to(torch.bfloat16).to(d), csr


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3000, 150, 0.1, 20), (1500, 150, 0.5, 3), (2500, 130, 0.0, 40), (2000, 158, 0.1, 12),
                                   (2000, 64, 0.1, 20), (2000, 100, 0.2, 35), (2000, 128, 0.1, 12), (1500, 94, 0.1, 12), (1500, 96, 0.1, 12)])
def test_recomputing_cfconv_backward_matches_the_three_pass_sequence(shape, monkeypatch):
    """K4 + K4b (ops.cfconv_recompute: the fused forward storing nothing per edge; backward = the same kernel on the by-source
    CSR for dh + mdl_cfconv_bwd_w for the filter network's parameter gradients, filter recomputed in both) against the unfused
    sequence (mdl_linear_act x 2 -> mdl_gather_mul_reduce and its autograd nodes) and against the fused forward with stored
    activations: the block's output, x.grad and the gradient of EVERY parameter of the InteractionBlock; many isolated nodes,
    in-degrees above one tile, filter widths at both ends of the supported range, with and without the by-source cache object.
    Tolerance 3e-2 of each gradient's scale (bf16 activations in all three forms)."""
    from matdeeplearn_amd import ops
    n, F, empty, max_in = shape
    conv, x, ei, dist, rbf, csr = _cfconv_case(n, F, 100 + n, empty, max_in)
    gy = torch.randn(n, 100, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).to(dev())
    res = {}
    for mode in ("unfused", "stored", "recompute", "recompute_cached"):
        monkeypatch.setattr(ops, "_CFCONV_FUSED", mode != "unfused")
        monkeypatch.setattr(ops, "_CFCONV_RECOMPUTE", mode.startswith("recompute"))
        monkeypatch.setattr(ops, "_CFCONV_RECOMPUTE_MIN_F", 0)            # (the default dispatch keeps the stored form below 96 units)
        ev = {"cfconv_fwd": [], "gmr_fwd": [], "cfconv_bwd_w": [], "cfconv_bwd_h": []}
        ops.KERNEL_EVENTS = ev
        xr = x.clone().requires_grad_(True)
        conv.zero_grad(set_to_none=True)
        kw = {"by_source": ops.BySourceAttrs()} if mode == "recompute_cached" else {}
        y = conv(xr, ei, dist, rbf, csr=csr, **kw)
        (y.float() * gy.float()).sum().backward()
        ops.KERNEL_EVENTS = None
        assert bool(ev["cfconv_bwd_w"]) == bool(ev["cfconv_bwd_h"]) == mode.startswith("recompute"), (mode, {k: len(v) for k, v in ev.items()})
        assert bool(ev["gmr_fwd"]) == (mode == "unfused")
        res[mode] = [y, xr.grad] + [p.grad for p in conv.parameters()]
    names = ["y", "x.grad"] + [k for k, _ in conv.named_parameters()]
    for mode in ("stored", "recompute", "recompute_cached"):
        for nm, a, b in zip(names, res[mode], res["unfused"]):
            assert a is not None and b is not None, (mode, nm)
            try:
                close(a, b, 3e-2, 3e-2)
            except AssertionError as e:
                raise AssertionError("%s / %s: %s" % (mode, nm, e))
    assert torch.equal(res["recompute"][0], res["stored"][0])          # the same forward kernel, with and without the stores
    assert torch.equal(res["recompute"][1], res["recompute_cached"][1])


@pytest.mark.gpu
@pytest.mark.parametrize("F", [150, 64, 100, 128])
def test_cfconv_weight_gradient_kernel_matches_fp64_and_ignores_padded_edges(F):
    """mdl_cfconv_bwd_w through the raw C-ABI against fp64 arithmetic on the operands the kernel multiplies (bf16 inputs and
    weights; dw, a1 and da rounded to bf16 where the kernel rounds them), on an edge array LONGER than the CSR covers (padded
    static batch: the tail rows hold NaN features and must not be read into any sum); then dh = mdl_cfconv_fwd on the by-source
    CSR against the per-edge loop; then the deterministic launch shape twice, bit-equal.  2e-2 of each gradient's scale."""
    from matdeeplearn_amd import _lib, ops
    d = dev()
    L, P, st = _lib.lib(), _lib.ptr, _lib.stream
    n, G, pad = 700, 50, 77
    g = torch.Generator().manual_seed(11)
    ei = rand_graph(n, 23, sort=True, empty_frac=0.2, max_in=40)
    E = ei.shape[1]
    csr = ops.build_csr(ei.to(d), n, assume_sorted=True)
    rbf = torch.rand(E + pad, G, generator=g).to(torch.bfloat16)
    rbf[E:] = float("nan")
    cut = torch.rand(E + pad, generator=g)
    cut[E:] = float("nan")
    h = torch.randn(n, F, generator=g).to(torch.bfloat16)
    gout = torch.randn(n, F, generator=g).to(torch.bfloat16)
    w1, b1 = torch.randn(F, G, generator=g) * 0.3, torch.randn(F, generator=g) * 0.2
    w2, b2 = torch.randn(F, F, generator=g) * 0.1, torch.randn(F, generator=g) * 0.2
    src = torch.cat([csr.src, torch.zeros(pad, dtype=torch.int32, device=d)])
    tgt = torch.cat([csr.tgt, torch.zeros(pad, dtype=torch.int32, device=d)])
    wpack = torch.empty(L.mdl_cfconv_wpack_bytes(), dtype=torch.uint8, device=d)
    dv = [t.to(d).contiguous() for t in (rbf, cut, h, gout, w1, b1, w2, b2)]
    _lib.check(L.mdl_cfconv_pack_weights(P(dv[4]), P(dv[5]), P(dv[6]), P(dv[7]), F, G, P(wpack), st()), "pack")

    scratch = torch.full((L.mdl_cfconv_bwd_w_scratch_bytes() // 4,), float("nan"), device=d)

    def run(flags, scr=None):
        outs = [torch.zeros(s_, dtype=torch.float32, device=d) for s_ in ((F, G), (F,), (F, F), (F,))]
        _lib.check(L.mdl_cfconv_bwd_w(P(dv[0]), P(dv[1]), P(dv[2]), P(dv[3]), P(csr.rowptr), P(src), P(tgt), P(wpack), P(outs[0]), P(outs[1]),
                                      P(outs[2]), P(outs[3]), P(scr), n, E + pad, F, G, _lib.MDL_BF16 | flags, st()), "cfconv_bwd_w")
        return outs
    dw1, db1, dw2, db2 = run(0)                                # partial sums added with atomics
    bfr = lambda t: t.to(torch.bfloat16).double()
    s_cpu, t_cpu = csr.src.cpu().long(), csr.tgt.cpu().long()
    a1 = bfr(torch.nn.functional.softplus(bfr(rbf[:E]) @ bfr(w1).t() + bfr(b1)) - np.log(2.0))
    dw = bfr(gout.double()[t_cpu] * h.double()[s_cpu] * cut[:E].double().view(-1, 1))
    da = bfr((dw @ bfr(w2)) * (1.0 - torch.exp(-(a1 + np.log(2.0)))))
    close(dw2, dw.t() @ a1, 2e-2, 2e-2)
    close(db2, dw.sum(0), 2e-2, 2e-2)
    close(dw1, da.t() @ bfr(rbf[:E]), 2e-2, 2e-2)
    close(db1, da.sum(0), 2e-2, 2e-2)
    for a, c in zip(run(0, scratch), (dw1, db1, dw2, db2)):    # the two-launch form (partial sums through the scratch buffer): the same sums
        close(a, c, 1e-3, 1e-3)
    # without bias outputs: the weight gradients alone
    outs = [torch.zeros(F, G, device=d), torch.zeros(F, F, device=d)]
    _lib.check(L.mdl_cfconv_bwd_w(P(dv[0]), P(dv[1]), P(dv[2]), P(dv[3]), P(csr.rowptr), P(src), P(tgt), P(wpack), P(outs[0]), None,
                                  P(outs[1]), None, P(scratch), n, E + pad, F, G, _lib.MDL_BF16, st()), "cfconv_bwd_w")
    close(outs[0], dw1, 1e-3, 1e-3)
    close(outs[1], dw2, 1e-3, 1e-3)
    # dh: the forward kernel on the by-source CSR (g in the place of h), rbf / cut rows in by-source order
    rowptr_s, col_s, eid_s, src_sorted = csr.transposed()
    idx = eid_s.long()
    rbf_s, cut_s = dv[0][:E].index_select(0, idx).contiguous(), dv[1][:E].index_select(0, idx).contiguous()
    dh = torch.full((n, F), float("nan"), dtype=torch.bfloat16, device=d)
    _lib.check(L.mdl_cfconv_fwd(P(rbf_s), P(cut_s), P(dv[3]), P(rowptr_s), P(col_s), P(src_sorted), P(wpack), P(dh), None, None, n, E,
                                F, G, _lib.MDL_BF16, st()), "cfconv(T)")
    w_ref = bfr(a1 @ bfr(w2).t() + bfr(b2))
    dh_ref = torch.zeros(n, F, dtype=torch.float64)
    dh_ref.index_add_(0, s_cpu, gout.double()[t_cpu] * w_ref * cut[:E].double().view(-1, 1))
    close(dh, dh_ref, 2e-2, 2e-2)
    # deterministic launch shape: bit-equal run to run, and equal to the parallel one within summation order
    r1, r2 = run(_lib.MDL_DETERMINISTIC), run(_lib.MDL_DETERMINISTIC)
    for a, b, c in zip(r1, r2, (dw1, db1, dw2, db2)):
        assert torch.equal(a, b)
        close(a, c, 1e-3, 1e-3)
    # no edges at all: outputs untouched
    z = torch.zeros(n + 1, dtype=torch.int32, device=d)
    o = [torch.full((F, G), 7.0, device=d), torch.full((F, F), 7.0, device=d)]
    _lib.check(L.mdl_cfconv_bwd_w(P(dv[0]), P(dv[1]), P(dv[2]), P(dv[3]), P(z), P(src), P(tgt), P(wpack), P(o[0]), None, P(o[1]), None,
                                  P(scratch), n, E + pad, F, G, _lib.MDL_BF16, st()), "cfconv_bwd_w")
    assert float(o[0].min()) == 7.0 and float(o[1].max()) == 7.0


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3000, 150, 0.1, 20), (1500, 150, 0.5, 3), (2500, 130, 0.0, 40), (2000, 158, 0.1, 12),
                                   (2000, 64, 0.1, 20), (2000, 100, 0.2, 35), (2000, 128, 0.1, 12)])
def test_fused_cfconv_forward_matches_the_three_pass_sequence_and_trains_through_it(shape, monkeypatch):
    """K4 (mdl_cfconv_fwd: filter network -> cutoff -> h[src] * W -> segmented sum in one pass, csrc/cfconv.hip) against the
    sequence it replaces (mdl_linear_act x 2 -> mdl_gather_mul_reduce) on the same bf16 operands: the InteractionBlock's output,
    the two stored activations through the gradients they produce (x and every parameter of the block), in train and in
    no-grad mode; many isolated nodes, in-degrees above one tile, the filter widths at both ends of the supported range.
    Tolerances: the fused pass rounds each MESSAGE to bf16 before the one-hot product (the three-pass kernel sums fp32
    products): 2e-2 of the output scale; gradients 3e-2 of their scale (bf16 activations either way)."""
    from matdeeplearn_amd import ops
    n, F, empty, max_in = shape
    conv, x, ei, dist, rbf, csr = _cfconv_case(n, F, 100 + n, empty, max_in)
    assert csr.E >= 1024
    gy = torch.randn(n, 100, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).to(dev())
    res = {}
    monkeypatch.setattr(ops, "_CFCONV_RECOMPUTE", False)              # the stored-activation form of the fused forward (kept as an option)
    for fused in (False, True):
        monkeypatch.setattr(ops, "_CFCONV_FUSED", fused)
        ev = {"cfconv_fwd": [], "gmr_fwd": []}
        ops.KERNEL_EVENTS = ev
        xr = x.clone().requires_grad_(True)
        conv.zero_grad(set_to_none=True)
        y = conv(xr, ei, dist, rbf, csr=csr)
        ops.KERNEL_EVENTS = None
        assert bool(ev["cfconv_fwd"]) == fused and bool(ev["gmr_fwd"]) != fused, ev      # the path under test is the one that ran
        (y.float() * gy.float()).sum().backward()
        with torch.no_grad():
            y_ng = conv(x, ei, dist, rbf, csr=csr)
        res[fused] = [y, y_ng, xr.grad] + [p.grad for p in conv.parameters()]
    close(res[True][0], res[False][0], 2e-2, 2e-2)
    close(res[True][1], res[False][1], 2e-2, 2e-2)
    assert torch.equal(res[True][0], res[True][1])                    # with and without the stored activations: the same sums
    for a, b in zip(res[True][2:], res[False][2:]):
        close(a, b, 3e-2, 3e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("F", [150, 64, 100, 128, 94, 96])
def test_fused_cfconv_matches_the_fp64_loop_oracle_and_zeroes_padded_rows(F):
    """mdl_cfconv_fwd through the raw C-ABI against the literal per-edge loop of the oracle (fp64, on the bf16-rounded operands)
    — out, and the two activations it stores — on an edge array LONGER than the CSR covers (a padded static batch): the rows
    past rowptr[N] of both activations are written as zeros.  Tolerance 2e-2 of the output scale (bf16 activations between
    the two layers, bf16 messages)."""
    from matdeeplearn_amd import _lib, ops
    d = dev()
    L, P, st = _lib.lib(), _lib.ptr, _lib.stream
    n, G, pad = 400, 50, 77
    g = torch.Generator().manual_seed(3)
    ei = rand_graph(n, 21, sort=True, empty_frac=0.2, max_in=40)
    E = ei.shape[1]
    csr = ops.build_csr(ei.to(d), n, assume_sorted=True)
    rbf = torch.rand(E + pad, G, generator=g).to(torch.bfloat16)
    cut = torch.rand(E + pad, generator=g)
    h = torch.randn(n, F, generator=g).to(torch.bfloat16)
    w1, b1 = torch.randn(F, G, generator=g) * 0.3, torch.randn(F, generator=g) * 0.2
    w2, b2 = torch.randn(F, F, generator=g) * 0.1, torch.randn(F, generator=g) * 0.2
    src = torch.cat([csr.src, torch.zeros(pad, dtype=torch.int32, device=d)])
    tgt = torch.cat([csr.tgt, torch.zeros(pad, dtype=torch.int32, device=d)])
    wpack = torch.empty(L.mdl_cfconv_wpack_bytes(), dtype=torch.uint8, device=d)
    dv = [t.to(d).contiguous() for t in (rbf, cut, h, w1, b1, w2, b2)]
    _lib.check(L.mdl_cfconv_pack_weights(P(dv[3]), P(dv[4]), P(dv[5]), P(dv[6]), F, G, P(wpack), st()), "pack")
    out = torch.full((n, F), float("nan"), dtype=torch.bfloat16, device=d)
    a1 = torch.full((E + pad, F), float("nan"), dtype=torch.bfloat16, device=d)
    w = torch.full((E + pad, F), float("nan"), dtype=torch.bfloat16, device=d)
    _lib.check(L.mdl_cfconv_fwd(P(dv[0]), P(dv[1]), P(dv[2]), P(csr.rowptr), P(src), P(tgt), P(wpack), P(out), P(a1), P(w), n, E + pad,
                                F, G, _lib.MDL_BF16, st()), "cfconv")
    # the oracle on the operands the kernel multiplies: bf16 inputs and weights, a1 rounded before layer 2, the filter before the product
    bfr = lambda t: t.to(torch.bfloat16).double()
    a1_ref = torch.nn.functional.softplus(bfr(rbf[:E]) @ bfr(w1).t() + bfr(b1)) - np.log(2.0)
    w_ref = bfr(a1_ref) @ bfr(w2).t() + bfr(b2)
    out_ref = torch.zeros(n, F, dtype=torch.float64)
    s_cpu, t_cpu = csr.src.cpu().long(), csr.tgt.cpu().long()
    for e in range(E):
        out_ref[t_cpu[e]] += h[s_cpu[e]].double() * bfr(w_ref[e]) * cut[e].double()
    close(a1[:E], a1_ref, 1e-2, 1e-2)
    close(w[:E], w_ref, 2e-2, 2e-2)
    close(out, out_ref, 2e-2, 2e-2)
    assert float(a1[E:].float().abs().max()) == 0.0 and float(w[E:].float().abs().max()) == 0.0
    # inference form: no activations, the same sums
    out2 = torch.full_like(out, float("nan"))
    _lib.check(L.mdl_cfconv_fwd(P(dv[0]), P(dv[1]), P(dv[2]), P(csr.rowptr), P(src), P(tgt), P(wpack), P(out2), None, None, n, E + pad,
                                F, G, _lib.MDL_BF16, st()), "cfconv")
    assert torch.equal(out, out2)
    assert L.mdl_cfconv_supported(150, 50, _lib.MDL_F32) == 0 and L.mdl_cfconv_supported(150, 64, _lib.MDL_BF16) == 0
    assert [L.mdl_cfconv_supported(f, 50, _lib.MDL_BF16) for f in (62, 64, 100, 101, 128, 158, 160)] == [0, 1, 1, 0, 1, 1, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(70000, 100, 100, 1), (70001, 150, 150, 2), (3000, 64, 114, 0), (40000, 150, 50, 0), (2100, 100, 100, 1)])
def test_tn_products_with_a_scratch_buffer_match_the_atomic_flush(shape):
    """mdl_gemm_tn_ex / mdl_dense_bwd_ex with `scratch` (the workgroups' accumulator blocks leave as plain stores and
    tn_reduce_kernel adds them) against the same entry points with scratch = NULL (atomics from every workgroup): dW, db, the
    column sums — and dX, which does not depend on the flush — on row counts that are not multiples of the tile, with the scratch
    buffer full of NaN on entry.  Same sums in another order: 1e-4 of the scale (fp32 accumulation of ~1e3 bf16 products per term)."""
    from matdeeplearn_amd import _lib
    d = dev()
    L, P, st = _lib.lib(), _lib.ptr, _lib.stream
    N, M, K, act = shape
    g_ = torch.Generator().manual_seed(N + M)
    g = torch.randn(N, M, generator=g_).to(torch.bfloat16).to(d)
    y = torch.randn(N, M, generator=g_).to(torch.bfloat16).to(d)
    x = torch.randn(N, K, generator=g_).to(torch.bfloat16).to(d)
    w = (torch.randn(M, K, generator=g_) * 0.1).to(torch.bfloat16).to(d)
    scratch = torch.full((L.mdl_tn_scratch_bytes() // 4,), float("nan"), device=d)
    res = []
    for scr in (None, scratch):
        c, cs = torch.zeros(M, K, device=d), torch.zeros(M, device=d)
        _lib.check(L.mdl_gemm_tn_ex(P(g), M, M, P(y) if act else None, M, act, P(x), K, K, P(c), P(cs) if K <= 158 else None, P(scr), N,
                                    _lib.MDL_BF16, st()), "gemm_tn_ex")
        out = [c, cs]
        if 34 <= M <= 160 and 34 <= K <= 158:
            dx = torch.full((N, K), float("nan"), dtype=torch.bfloat16, device=d)
            dw, db = torch.zeros(M, K, device=d), torch.zeros(M, device=d)
            _lib.check(L.mdl_dense_bwd_ex(P(g), M, M, P(y) if act else None, M, act, P(x), K, K, P(w), P(dx), K, 0, None, P(dw), P(db), P(scr),
                                          N, _lib.MDL_BF16, st()), "dense_bwd_ex")
            out += [dw, db, dx]
        res.append(out)
    for a, b in zip(res[1], res[0]):
        assert torch.isfinite(a.float()).all()
        close(a, b, 1e-4, 1e-4)
    assert torch.equal(res[1][-1], res[0][-1]) or len(res[0]) == 2          # dX: the same arithmetic either way
