"""Host-side operators over libmdl_hip.so: CSR graph index, RBF expansion, scatter / pooling, CGConv.

These mirror the third-party operator API the reference calls (level b2):
  scatter / scatter_mean / scatter_add  <- torch_scatter      (matdeeplearn/models/megnet.py:13,86,130-132,342-348)
  global_{mean,add,max}_pool            <- torch_geometric.nn (matdeeplearn/models/cgcnn.py:154)
  cgconv                                <- torch_geometric.nn.CGConv.forward (matdeeplearn/models/cgcnn.py:136-145)
  rbf_expand                            <- GaussianSmearing.forward (matdeeplearn/process/process.py:588-590)
Same argument meaning, differentiable through torch.autograd.Function, errors raised as Python
exceptions.  Every op runs a hand-written HIP kernel; there is no eager/CPU fallback.
"""
import collections
import os

import torch

from . import _lib
from ._lib import MdlError, check, dtype_code, lib, ptr, require_hip, stream


# ------------------------------------------------------------------------------------------------
# CSR-by-target graph index
# ------------------------------------------------------------------------------------------------
class EdgeCSR:
    """Edges sorted by target.  rowptr [N+1], src/tgt [E] int32, eperm [E] int32 or None when the
    caller's per-edge tensors are already in CSR order (the product loader guarantees that)."""

    __slots__ = ("rowptr", "src", "tgt", "eperm", "N", "E", "_row", "_col", "_t", "_attr", "_tb", "_bal", "partial", "__weakref__")

    def __init__(self, rowptr, src, tgt, eperm, N, E, row=None, col=None):
        self.rowptr, self.src, self.tgt, self.eperm, self.N, self.E = rowptr, src, tgt, eperm, int(N), int(E)
        self._attr = None
        self._bal = None
        self._row, self._col, self._t, self._tb = row, col, None, None
        self.partial = False      # padded static batch: the arrays hold more rows than the rowptr ranges cover

    def set_transposed(self, t):
        """(rowptr_s, col_s, eid_s, src_sorted) supplied by the loader (static buffers of the HIP-graph path)."""
        self._t = tuple(t)

    def set_transposed_builder(self, fn):
        """fn() -> (rowptr_s, col_s, eid_s, src_sorted): the loader's sort-free construction, run on first use."""
        self._tb = fn

    def balance(self):
        """[N + 1] int32 cost prefix for the work distribution of the edge-per-lane CGConv backward (mdl_cgconv_balance + one
        cumsum): topology only, so it is built once per batch and serves every layer."""
        if self._bal is None:
            cost = torch.empty(self.N + 1, dtype=torch.int32, device=self.rowptr.device)
            check(lib().mdl_cgconv_balance(ptr(self.rowptr), ptr(self.src), self.N, ptr(cost), stream()), "mdl_cgconv_balance")
            self._bal = torch.cumsum(cost, 0, dtype=torch.int32)
        return self._bal

    def refresh_balance(self):
        """A CSR over static buffers (HIP-graph path) is rewritten in place with every batch: its prefix is rebuilt into the
        SAME tensor by the batch assembly, inside the captured graph."""
        if not (_BALANCE and self.E >= 400000 and self.eperm is None):
            return
        if self._bal is None:
            self._bal = torch.zeros(self.N + 1, dtype=torch.int32, device=self.rowptr.device)
        cost = torch.empty(self.N + 1, dtype=torch.int32, device=self.rowptr.device)
        check(lib().mdl_cgconv_balance(ptr(self.rowptr), ptr(self.src), self.N, ptr(cost), stream()), "mdl_cgconv_balance")
        torch.cumsum(cost, 0, dtype=torch.int32, out=self._bal)

    def seg_tgt(self):
        """segment index of `col` (target per edge): what scatter(..., index=edge_index[1]) needs, without a sort"""
        if self.eperm is not None:
            return None
        return make_seg_index(self.rowptr, self.tgt, partial=self.partial)

    def seg_src(self):
        """segment index of `row` (source per edge) from the by-source CSR"""
        if self.eperm is not None:
            return None
        rowptr_s, _, eid_s, src_sorted = self.transposed()
        si = make_seg_index(rowptr_s, src_sorted, partial=self.partial)
        si.perm = eid_s
        return si

    def sorted_attr(self, edge_attr):
        """edge_attr rows in CSR (target-sorted) order.  The static CGConv kernels stream the edge features
        sequentially, so an unsorted edge list is permuted ONCE per (edge_attr, CSR) here (cached; edge_attr
        is constant across the layers of a model) instead of being gathered through eperm in every launch."""
        if self.eperm is None:
            return edge_attr
        k = (edge_attr.data_ptr(), edge_attr._version, edge_attr.dtype, tuple(edge_attr.shape))
        if self._attr is None or self._attr[0] != k:
            self._attr = (k, edge_attr.index_select(0, self.eperm.long()).contiguous(), edge_attr)
        return self._attr[1]

    @property
    def row(self):
        """source of every edge in the CALLER's edge order (int32)"""
        if self._row is None:
            if self.eperm is None:
                self._row = self.src
            else:
                self._row = torch.empty_like(self.src)
                self._row[self.eperm.long()] = self.src
        return self._row

    @property
    def col(self):
        """target of every edge in the caller's edge order (int32)"""
        if self._col is None:
            if self.eperm is None:
                self._col = self.tgt
            else:
                self._col = torch.empty_like(self.tgt)
                self._col[self.eperm.long()] = self.tgt
        return self._col

    def transposed(self):
        """CSR by SOURCE: (rowptr_s [N+1], col_s = target per slot, eid_s = caller's edge id per slot)."""
        if self._t is None and self._tb is not None:
            self._t = tuple(self._tb())
        if self._t is None:
            perm = torch.argsort(self.src, stable=True)
            src_sorted = self.src.index_select(0, perm)
            col_s = self.tgt.index_select(0, perm)
            eid_s = (perm if self.eperm is None else self.eperm.long().index_select(0, perm)).to(torch.int32)
            self._t = (csr_rowptr(src_sorted.contiguous(), self.N), col_s.contiguous(), eid_s.contiguous(),
                       src_sorted.contiguous())
        return self._t


def csr_rowptr(sorted_index_i32, num_segments):
    require_hip(sorted_index_i32)
    rowptr = torch.empty(num_segments + 1, dtype=torch.int32, device=sorted_index_i32.device)
    check(lib().mdl_csr_rowptr(ptr(sorted_index_i32), sorted_index_i32.numel(), num_segments, ptr(rowptr), stream()),
          "mdl_csr_rowptr")
    return rowptr


def build_csr(edge_index, num_nodes, assume_sorted=False):
    """edge_index: [2, E] int64/int32 (row 0 = source j, row 1 = target i).  No host sync."""
    require_hip(edge_index)
    row, col = edge_index[0], edge_index[1]
    if assume_sorted:
        tgt = col.to(torch.int32).contiguous()
        src = row.to(torch.int32).contiguous()
        eperm = None
    else:
        perm = torch.argsort(col, stable=True)
        tgt = col.index_select(0, perm).to(torch.int32)
        src = row.index_select(0, perm).to(torch.int32)
        eperm = perm.to(torch.int32)
    return EdgeCSR(csr_rowptr(tgt, num_nodes), src, tgt, eperm, num_nodes, col.numel(),
                   row=None if assume_sorted else row.to(torch.int32).contiguous(),
                   col=None if assume_sorted else col.to(torch.int32).contiguous())


_CSR_CACHE = collections.OrderedDict()
_CSR_CACHE_MAX = 64                      # entries
_CSR_CACHE_MAX_BYTES = 256 << 20         # and bytes pinned (an entry keeps ~40 B/edge of index tensors alive)
_csr_cache_bytes = 0


def _key(edge_index, n):
    return (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(n), edge_index.device.index)


def _csr_entry_bytes(edge_index, csr):
    return edge_index.numel() * edge_index.element_size() + 4 * (3 * csr.E + csr.N + 1) + (4 * csr.E if csr.eperm is not None else 0)


def register_csr(edge_index, csr):
    """Attach a pre-built CSR to an edge_index tensor (used by the product loader).  The cache is bounded by entries AND
    by the device bytes it pins; the newest entry always stays."""
    global _csr_cache_bytes
    k = _key(edge_index, csr.N)
    old = _CSR_CACHE.pop(k, None)
    if old is not None:
        _csr_cache_bytes -= old[2]
    nb = _csr_entry_bytes(edge_index, csr)
    _CSR_CACHE[k] = (csr, edge_index, nb)  # keep the tensor alive so data_ptr cannot be recycled
    _csr_cache_bytes += nb
    while len(_CSR_CACHE) > 1 and (len(_CSR_CACHE) > _CSR_CACHE_MAX or _csr_cache_bytes > _CSR_CACHE_MAX_BYTES):
        _csr_cache_bytes -= _CSR_CACHE.popitem(last=False)[1][2]


def csr_for(edge_index, num_nodes):
    """CSR of a PyG-style edge_index; built on first use (device sort) and cached per tensor."""
    if NO_INDEX_CACHE:
        return build_csr(edge_index, num_nodes)
    k = _key(edge_index, num_nodes)
    hit = _CSR_CACHE.get(k)
    if hit is not None:
        _CSR_CACHE.move_to_end(k)
        return hit[0]
    csr = build_csr(edge_index, num_nodes)
    register_csr(edge_index, csr)
    return csr


# ------------------------------------------------------------------------------------------------
# K1 — Gaussian RBF expansion
# ------------------------------------------------------------------------------------------------
def rbf_offsets(start=0.0, stop=1.0, resolution=50, device=None):
    """process.py:583 — the fp32 torch.linspace centre buffer (computed on the host so the grid is
    bit-identical to the reference's, then uploaded)."""
    return torch.linspace(start, stop, resolution).to(device)


def rbf_coeff(start=0.0, stop=1.0, width=0.2):
    return -0.5 / ((stop - start) * width) ** 2  # process.py:585


def rbf_expand(dist, start=0.0, stop=1.0, resolution=50, width=0.2, out_dtype=torch.float32, offsets=None,
               out=None):
    """GaussianSmearing(start, stop, resolution, width)(dist): [E] fp32 -> [E, resolution]."""
    require_hip(dist)
    if dist.dtype != torch.float32:
        raise MdlError("rbf_expand: distances must be float32")
    dist = dist.contiguous()
    if offsets is None:
        offsets = rbf_offsets(start, stop, resolution, dist.device)
    E, G = dist.numel(), offsets.numel()
    if out is None:
        out = torch.empty((E, G), dtype=out_dtype, device=dist.device)
    check(lib().mdl_rbf_expand(ptr(dist), ptr(offsets), float(rbf_coeff(start, stop, width)), ptr(out), E, G,
                               out.stride(0) if E else G, dtype_code(out), stream()), "mdl_rbf_expand")
    return out


# ------------------------------------------------------------------------------------------------
# K5 — scatter / pooling over a sorted (or sortable) index
# ------------------------------------------------------------------------------------------------
class _SegIndex:
    __slots__ = ("rowptr", "seg", "perm", "N", "E", "partial")


_SEG_CACHE = collections.OrderedDict()
# index tensors whose segment index the loader already knows (edge_index rows of a product batch: no sort, and the only
# correct index for the static buffers of the HIP-graph path, whose unused tail must stay outside every segment)
_SEG_KNOWN = collections.OrderedDict()


def register_seg_index(index, si, owner=None):
    """`scatter(src, index, ...)` calls with this very memory (same address and length) use `si` — a segment index or a
    callable that builds it on first use — for as long as `owner` (the tensor that owns the storage; default `index`)
    is alive.  Only a weak reference is kept: when the owner dies its address may be recycled and the entry is dropped."""
    import weakref
    _SEG_KNOWN[(index.data_ptr(), index.numel(), index.device.index)] = [si, weakref.ref(index if owner is None else owner)]
    while len(_SEG_KNOWN) > 64:
        _SEG_KNOWN.popitem(last=False)


def _known_seg_index(index, dim_size):
    key = (index.data_ptr(), index.numel(), index.device.index)
    ent = _SEG_KNOWN.get(key)
    if ent is None:
        return None
    if ent[1]() is None:
        del _SEG_KNOWN[key]
        return None
    si = ent[0]() if callable(ent[0]) else ent[0]     # (a callable is resolved per use: the registry must not pin a batch)
    return si if (si is not None and si.N == int(dim_size)) else None


# Static buffers (HIP-graph path) are rewritten in place by kernels the version counter does not see: with NO_INDEX_CACHE
# every index structure is rebuilt (and the building kernels become part of the captured graph).
NO_INDEX_CACHE = False


def _seg_index(index, dim_size, assume_sorted):
    known = _known_seg_index(index, dim_size)
    if known is not None:
        return known
    if NO_INDEX_CACHE:
        return _seg_index_build(index, dim_size, assume_sorted)
    key = (index.data_ptr(), index._version, index.numel(), int(dim_size), bool(assume_sorted), index.device.index)
    hit = _SEG_CACHE.get(key)
    if hit is not None:
        return hit[0]
    si = _seg_index_build(index, dim_size, assume_sorted)
    _SEG_CACHE[key] = (si, index)
    while len(_SEG_CACHE) > 64:
        _SEG_CACHE.popitem(last=False)
    return si


def _seg_index_build(index, dim_size, assume_sorted):
    si = _SegIndex()
    si.N, si.E, si.partial = int(dim_size), index.numel(), False
    if assume_sorted:
        si.seg, si.perm = index.to(torch.int32).contiguous(), None
    else:
        perm = torch.argsort(index, stable=True)
        si.seg, si.perm = index.index_select(0, perm).to(torch.int32), perm.to(torch.int32)
    si.rowptr = csr_rowptr(si.seg, si.N)
    return si


class _SegmentReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, si, reduce):
        require_hip(src)
        src = src.contiguous()
        C = src.numel() // max(src.shape[0], 1) if src.dim() > 1 else 1
        out = torch.empty((si.N,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        argmax = torch.empty((si.N, C), dtype=torch.int32, device=src.device) if reduce == _lib.MDL_MAX else None
        check(lib().mdl_segment_reduce_fwd(ptr(src), ptr(si.rowptr), ptr(si.perm), ptr(out), ptr(argmax), si.N, C,
                                           reduce, dtype_code(src), stream()), "mdl_segment_reduce_fwd")
        ctx.si, ctx.reduce, ctx.C, ctx.shape = si, reduce, C, tuple(src.shape)
        ctx.save_for_backward(argmax) if argmax is not None else ctx.save_for_backward()
        return out

    @staticmethod
    def backward(ctx, g):
        si, C = ctx.si, ctx.C
        g = g.contiguous()
        argmax = ctx.saved_tensors[0] if ctx.reduce == _lib.MDL_MAX else None
        # rows outside every segment (the unused tail of a padded static batch) must get exact zeros: they feed dense
        # per-row kernels (weight-gradient GEMMs) that run over all rows.  The sum / mean kernels write EVERY row from g[seg[row]],
        # and a padded row's segment id is the dummy graph / the first padding node, whose gradient row is exactly zero by the
        # static batch's invariant — so only the max form (which writes the argmax rows alone) needs the fill launch
        alloc = torch.zeros if ctx.reduce == _lib.MDL_MAX else torch.empty
        gs = alloc(ctx.shape, dtype=g.dtype, device=g.device)
        check(lib().mdl_segment_reduce_bwd(ptr(g), ptr(si.rowptr), ptr(si.seg), ptr(si.perm), ptr(argmax), ptr(gs),
                                           si.N, si.E, C, ctx.reduce, dtype_code(g), stream()),
              "mdl_segment_reduce_bwd")
        return gs, None, None


class _ResidualSegmentReduce(torch.autograd.Function):
    """(src + res, reduce(src by segment)) as one autograd node: `src` feeds a residual sum AND a segmented reduction (the edge
    state of a MEGNet block: e' + e and scatter_mean(e', row), megnet.py:86 with :321-336), so its gradient is the sum of two
    [E, C] tensors — formed here inside the reduction's backward (mdl_segment_reduce_bwd_add) instead of by autograd's
    separate accumulation pass."""

    @staticmethod
    def forward(ctx, src, res, si, reduce):
        C = src.shape[1]
        out = torch.empty((si.N, C), dtype=src.dtype, device=src.device)
        check(lib().mdl_segment_reduce_fwd(ptr(src), ptr(si.rowptr), ptr(si.perm), ptr(out), None, si.N, C, reduce,
                                           dtype_code(src), stream()), "mdl_segment_reduce_fwd")
        ctx.si, ctx.reduce = si, reduce
        return src + res, out

    @staticmethod
    def backward(ctx, g_sum, g_red):
        si = ctx.si
        g_sum, g_red = g_sum.contiguous(), g_red.contiguous()
        gs = torch.empty_like(g_sum)
        check(lib().mdl_segment_reduce_bwd_add(ptr(g_red), ptr(si.rowptr), ptr(si.seg), ptr(si.perm), ptr(g_sum), ptr(gs),
                                               si.N, si.E, g_sum.shape[1], ctx.reduce, dtype_code(g_sum), stream()),
              "mdl_segment_reduce_bwd_add")
        return gs, g_sum, None, None


def residual_scatter(src, res, index, dim_size, reduce="mean", assume_sorted=False):
    """(src + res, scatter(src, index, 0, dim_size, reduce)) — one node when the shapes allow (2-D contiguous rows of a
    multiple of 4 elements, sum / mean, an index whose segments cover every row), the two separate ops otherwise."""
    if (reduce in ("sum", "mean") and src.dim() == 2 and src.is_cuda and src.is_contiguous() and res.is_contiguous()
            and res.shape == src.shape and res.dtype == src.dtype and src.shape[1] % 4 == 0 and src.data_ptr() % 16 == 0
            and src.shape[0] > 0 and torch.is_grad_enabled() and src.requires_grad):
        require_hip(src, index)
        si = _seg_index(index, dim_size, assume_sorted)
        if not si.partial and si.E == src.shape[0]:
            return _ResidualSegmentReduce.apply(src, res, si, _lib.REDUCE[reduce])
    return src + res, scatter(src, index, 0, dim_size, reduce, assume_sorted)


def scatter(src, index, dim=0, dim_size=None, reduce="sum", assume_sorted=False, seg_index=None):
    """torch_scatter.scatter(src, index, dim=0, dim_size, reduce) semantics.  When
    dim_size is None it is index.max()+1, which costs a host sync — pass dim_size on hot paths.
    `seg_index`: a prebuilt segment index (make_seg_index) for `index`, e.g. the loader's node -> graph map."""
    if dim != 0:
        raise MdlError("scatter: only dim=0 is on the hot path")
    require_hip(src, index)
    if reduce not in _lib.REDUCE:
        raise MdlError("scatter: unsupported reduce %r" % (reduce,))
    if seg_index is None:
        if dim_size is None:
            dim_size = int(index.max()) + 1 if index.numel() else 0
        seg_index = _seg_index(index, dim_size, assume_sorted)
    return _SegmentReduce.apply(src, seg_index, _lib.REDUCE[reduce])


def make_seg_index(rowptr_i32, seg_i32, partial=False):
    """Segment index over a SORTED segment id vector from its row pointers (rowptr [S+1] int32, seg [E] int32): what the
    loaders know anyway (graph -> first node), so pooling needs neither a sort nor a search.  partial: the rowptr ranges
    do not cover all E rows (padded static batch)."""
    si = _SegIndex()
    si.N, si.E, si.partial = rowptr_i32.numel() - 1, seg_i32.numel(), bool(partial)
    si.rowptr, si.seg, si.perm = rowptr_i32, seg_i32, None
    return si


def scatter_mean(src, index, dim=0, dim_size=None, assume_sorted=False):
    return scatter(src, index, dim, dim_size, "mean", assume_sorted)


def scatter_add(src, index, dim=0, dim_size=None, assume_sorted=False):
    return scatter(src, index, dim, dim_size, "sum", assume_sorted)


def global_mean_pool(x, batch, size=None, seg_index=None):
    """`batch` is non-decreasing by construction (PyG collate / the product loader)."""
    return scatter(x, batch, 0, size, "mean", assume_sorted=True, seg_index=seg_index)


def global_add_pool(x, batch, size=None, seg_index=None):
    return scatter(x, batch, 0, size, "sum", assume_sorted=True, seg_index=seg_index)


def global_max_pool(x, batch, size=None, seg_index=None):
    return scatter(x, batch, 0, size, "max", assume_sorted=True, seg_index=seg_index)


POOLS = {"global_mean_pool": global_mean_pool, "global_add_pool": global_add_pool,
         "global_max_pool": global_max_pool}


# ------------------------------------------------------------------------------------------------
# K2/K3 — fused CGConv
# ------------------------------------------------------------------------------------------------
def _rup(a, b):
    return (a + b - 1) // b * b


# bench.py sets this to {"fwd": [], "bwd": [], "bwd_node": [], "bwd_grads": []} to collect (start, end) HIP events recorded
# on the launch stream around the conv kernels (K2; K3 = edge pass + node kernel + gradient assembly); None = no events.
KERNEL_EVENTS = None


def _launch_timed(key, fn):
    ev = KERNEL_EVENTS
    if ev is None or key not in ev:
        return fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    rc = fn()
    b.record()
    ev[key].append((a, b))
    return rc


class ZeroArena:
    """One zero-filled scratch per training step for the kernels' small accumulators (BatchNorm sums, the conv
    backward's dW partials): a model step needs a dozen of them, and a dozen 5-us fill launches cost more than the
    kernels they serve.  `with ops.zero_arena(device):` around forward + backward zero-fills ONE buffer and hands out
    views; outside of it (or when it is full) zeros() falls back to torch.zeros.  Only for tensors that die with the
    step — nothing that autograd may hand over as a parameter gradient."""

    def __init__(self, device, nbytes=4 << 20):
        self.buf = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        self.off = 0

    def reset(self):
        self.buf.zero_()
        self.off = 0

    def take(self, shape):
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 63) & ~63                       # 256-byte granules keep every view 16-byte aligned
        if self.off + n_al > self.buf.numel():
            return None
        t = self.buf[self.off:self.off + n].view(*shape)
        self.off += n_al
        return t


_ARENA = None

# Weight-gradient accumulators of the dense layers (the TN GEMM adds into zero-filled [out, in] + [out] buffers): a model
# with dozens of Linear layers would launch dozens of 3-us fills per step.  The model's forward opens ONE freshly
# allocated, zero-filled buffer per step (new_grad_arena) and the backward passes carve their accumulators out of it.
# Unlike the ZeroArena this memory is never reused: the slices become the parameters' .grad tensors and live as long as
# autograd / the optimizer keeps them.
_GRAD_ARENA = None


def new_grad_arena(device, nfloats):
    global _GRAD_ARENA
    _GRAD_ARENA = [torch.zeros(int(nfloats), dtype=torch.float32, device=device), 0]


def _zeros_grad(n, device):
    a = _GRAD_ARENA
    n_al = (int(n) + 63) & ~63
    if a is not None and a[0].device == device and a[1] + n_al <= a[0].numel():
        t = a[0][a[1]:a[1] + int(n)]
        a[1] += n_al
        return t
    return torch.zeros(int(n), dtype=torch.float32, device=device)


class zero_arena:
    def __init__(self, device, nbytes=4 << 20):
        self.key = (torch.device(device), nbytes)

    _cache = {}

    def __enter__(self):
        global _ARENA
        a = zero_arena._cache.get(self.key)
        if a is None:
            a = zero_arena._cache[self.key] = ZeroArena(*self.key)
        a.reset()
        self.prev, _ARENA = _ARENA, a
        return a

    def __exit__(self, *exc):
        global _ARENA
        _ARENA = self.prev
        return False


def _zeros_step(shape, device):
    """fp32 zeros that live no longer than the current step (see ZeroArena)."""
    a = _ARENA
    if a is not None and a.buf.device == device:
        t = a.take(shape)
        if t is not None:
            return t
    return torch.zeros(shape, dtype=torch.float32, device=device)


_WS = {}


def _workspace(device, nbytes):
    """Per-device scratch for the kernels' work counters (the library zeroes what it uses on the stream; launches on one
    stream are ordered, so the layers of a model share it)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        t = _WS[key] = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    return t


_TN_SCRATCH = True     # partial sums of the streaming TN products through a scratch buffer + reduce launch instead of atomics
_TNS = {}


_TN_SCRATCH_MIN_ROWS = 4096      # (measured, tools/bench_dense.py: atomics win at 2.6 k rows — 8.2 vs 10.8 us —, the scratch form from 8 k — 13.8 vs 17.8)
# rows from which a bf16 dense layer (forward, dX, the one-pass backward, the TN weight gradient) takes the streaming HIP kernels
# instead of the library's GEMMs
_DENSE_MIN_ROWS = 64        # (1024 until round 6: at ~100 graph rows — the reference's batch size — the library's GEMMs cost 11-12 us each against 5-6)


def _tn_scratch(device, rows=None):
    """Per device and stream: mdl_tn_scratch_bytes() bytes for the _ex forms of the TN products (launches on one stream are
    ordered, so every layer shares it; allocated once — before any HIP-graph capture, by the warm-up steps).  None (= atomics
    from the few workgroups there are) for a product over few rows: at the reference's batch size the reduce launch costs more
    than the handful of atomics it replaces."""
    if not _TN_SCRATCH or (rows is not None and rows < _TN_SCRATCH_MIN_ROWS):
        return None
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _TNS.get(key)
    if t is None:
        t = _TNS[key] = torch.empty(lib().mdl_tn_scratch_bytes(), dtype=torch.uint8, device=device)
    return t


def _gemm_tn(a, lda, M, y, ldy, act, b, ldb, K, c, colsum, N, flags, device):
    """mdl_gemm_tn / _colsum / _act in their scratch form (c [M, K] += (a .* act'(y))^T b, colsum += column sums)"""
    return lib().mdl_gemm_tn_ex(ptr(a), lda, M, ptr(y), ldy, act, ptr(b), ldb, K, ptr(c), ptr(colsum), ptr(_tn_scratch(device, N)), N, flags,
                                stream())


# Deterministic mode (include/mdl_hip.h, MDL_DETERMINISTIC): the kernels that combine per-workgroup partial sums with
# floating-point atomics — the CGConv backward edge pass and node kernel, the TN GEMM, the BatchNorm sums, the fused head —
# are launched in a shape in which every sum gets its terms from ONE wave in program order: bit-reproducible from run to run,
# a few hundred times slower.  For HIP-vs-HIP regression checks (graph replay vs eager, padded rows, data-parallel exchange);
# `with ops.deterministic():` or `ops.configure(deterministic=True)` (neither the library nor the package reads the environment).
_DET = False


def set_deterministic(on=True):
    global _DET
    prev, _DET = _DET, bool(on)
    return prev


class deterministic:
    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        self.prev = set_deterministic(self.on)
        return self

    def __exit__(self, *exc):
        set_deterministic(self.prev)
        return False


# Dispatch options.  Defaults are the measured-best paths; there is NO environment variable behind any of them (round 6): a
# caller that wants another path says so with ops.configure(name=value, ...) — bench.py's `--ops name=value` for A/B runs.
# name -> (module attribute, what it selects)
OPTIONS = {
    "deterministic": ("_DET", "bit-reproducible launch shapes of every atomically accumulating kernel (slow; HIP-vs-HIP tests)"),
    "gmr_dw": ("_GMR_DW", "CFConv backward: dh and dw from one walk over the by-source CSR"),
    "balance": ("_BALANCE", "cost-balanced node ranges for the edge-per-lane CGConv backward (E >= 4e5)"),
    "pad128": ("_PAD128", "C in (96, 128): static 128-channel CGConv kernels on zero-padded rows"),
    "rsrc16": ("_RSRC16", "by-source sums of the CGConv backward in bf16 (packed atomics)"),
    "cg_bn_stats": ("_CG_BN_STATS", "BatchNorm statistics in the CGConv forward's epilogue (measured time-neutral: off)"),
    "direct_grads": ("_DIRECT_GRADS", "K3 / K3c add into the final dW layout (measured slower at small batches: off)"),
    "cfconv_fused": ("_CFCONV_FUSED", "K4: the fused CFConv forward"),
    "cfconv_recompute": ("_CFCONV_RECOMPUTE", "K4b: CFConv backward with the filter recomputed (no per-edge activations)"),
    "tn_colsum": ("_TN_COLSUM", "bias gradients out of the TN GEMM"),
    "dense_bwd": ("_DENSE_BWD", "dX + dW + db of a tall dense layer in one pass"),
    "dense_bwd_wide": ("_DENSE_BWD_WIDE", "... also when both widths exceed 128"),
    "mlp_head": ("_MLP_HEAD", "post-FC head as one launch per direction"),
    "linear_wide": ("_LINEAR_WIDE", "NNConv's Y = x W2r on the streaming kernel"),
    "tn_scratch": ("_TN_SCRATCH", "weight-gradient blocks of the TN products as plain stores + a reduce launch (off: atomics from every workgroup)"),
}


def configure(**kw):
    """Set dispatch options (see OPTIONS); returns the previous values of the ones given.  Unknown names raise."""
    prev = {}
    g = globals()
    for k, v in kw.items():
        if k not in OPTIONS:
            raise MdlError("ops.configure: unknown option %r (known: %s)" % (k, ", ".join(sorted(OPTIONS))))
        attr = OPTIONS[k][0]
        prev[k] = g[attr]
        g[attr] = bool(v)
    return prev


def options():
    """current value of every dispatch option"""
    return {k: globals()[a] for k, (a, _) in OPTIONS.items()}


def _dflag():
    return _lib.MDL_DETERMINISTIC if _DET else 0


# Which backward edge pass ops.cgconv asks for where both exist (bf16, C = 64, bf16 by-source sums): None = the library's
# edge-count heuristic (kernel 2 from 4e5 edges), "per_wave" / "edge_lane" force one (tests, A/B).
K3_VARIANT = None


def _k3flag():
    return {None: 0, "per_wave": _lib.MDL_K3_PER_WAVE, "edge_lane": _lib.MDL_K3_EDGE_LANE}[K3_VARIANT]


def last_k3():
    """1 per-wave kernel, 2 edge-per-lane kernel 2, 3 per-wave kernel in its deterministic shape (mdl_debug_last_k3)"""
    return int(lib().mdl_debug_last_k3())


_GMR_DW = True          # CFConv backward: dh and dw from one walk over the by-source CSR
_BALANCE = True     # cost-balanced node ranges for the edge-per-lane backward
_PAD128 = True      # C in (96, 128): static 128-channel kernels on zero-padded rows
# By-source sums of the CGConv backward in bf16, accumulated with packed bf16 atomics (mdl_cgconv_bwd_h / mdl_cgconv_bwd_node_h;
# bf16 mode, C in {32, 64, 128}, G = 50): half the atomic operations and bytes of the fp32 buffer; a source row is rounded to bf16
# once per window flush (1-3 partial sums per node).  ops.configure(rsrc16=False) restores the fp32 buffer.
_RSRC16 = True


# r_src (by-source sums of the CGConv backward: fp32 [N, 2*Cp], accumulated with atomics) must start at zero: 107 MB per
# layer at the bench batch, i.e. a 15-22 us fill launch in front of every edge pass.  Its only reader, the node kernel,
# can hand it back zeroed (mdl_cgconv_bwd_node_z), so eager steps keep ONE buffer per device and stream for all layers and
# steps.  `dirty` covers an exception between the two launches; captured graphs keep their own zero-filled tensors.
_RSRC = {}


_RSRC_CAP = {}


def _take_rsrc(nfloats, device):
    if torch.cuda.is_current_stream_capturing():
        # A captured step adopts the buffer its warm-up iterations left behind (training.GraphedStep runs the step eagerly on a
        # side stream, synchronises, then captures): allocated outside the capture, zero on entry, handed back zeroed by the
        # node kernel at the end of every layer — so every replay finds it zero, and the four zero fills per step
        # (53 MB each at the bench batch) that a fresh tensor per layer cost are gone.  The buffer leaves the eager table for
        # good (the graph owns its address); without a clean candidate the caller falls back to a fresh zero-filled tensor.
        # Every adopted buffer stays referenced for the life of the process (a LIST per device): its address is baked into the
        # graph that adopted it, so a later, larger capture must not drop the last reference to it (the allocator would hand the
        # memory out again and a replay of the earlier graph would zero-fill and add into somebody else's tensor).  A buffer
        # whose dirty flag is set (an exception between an edge pass and its node kernel) is never handed to a capture.
        owned = _RSRC_CAP.setdefault(device.index, [])
        for ent in owned:
            if ent[0].numel() >= nfloats and not ent[1]:
                return ent
        cand = [k for k, v in _RSRC.items() if k[0] == device.index and not v[1] and v[0].numel() >= nfloats]
        if not cand:
            return None
        ent = _RSRC.pop(max(cand, key=lambda k: _RSRC[k][0].numel()))
        owned.append(ent)
        return ent
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ent = _RSRC.get(key)
    if ent is None or ent[0].numel() < nfloats or ent[1]:
        ent = _RSRC[key] = [torch.zeros(int(nfloats), dtype=torch.float32, device=device), False]
    return ent


class _CGConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, edge_attr, w_f, b_f, w_s, b_s, csr, aggr, bn=None, packed=None, split=False):
        require_hip(x, edge_attr, w_f, w_s)
        if edge_attr.requires_grad:
            raise MdlError("cgconv: gradients w.r.t. edge_attr are not implemented (the reference's edge features are "
                           "constants, cgcnn.py:136-145); detach() them or use a differentiable composition")
        if edge_attr.dtype != x.dtype:
            raise MdlError("cgconv: x (%s) and edge_attr (%s) must share a dtype" % (x.dtype, edge_attr.dtype))
        x, edge_attr = x.contiguous(), edge_attr.contiguous()
        N, C = x.shape
        E, G = edge_attr.shape
        if csr.N != N or csr.E != E:
            raise MdlError("cgconv: CSR (%d nodes, %d edges) does not match x/edge_attr (%d, %d)" % (csr.N, csr.E, N, E))
        dt = dtype_code(x)
        L = lib()
        # split-bf16 products on fp32 storage (MDL_SPLIT_BF16, the "bf16x3" parity mode): where the kernels have the shape
        # for it (C = 64, G = 50, CSR-ordered edge features); everything else of an fp32 tensor runs the exact form
        # (C = 64; C in (96, 128]: the static 128-channel kernels on zero-padded rows — the reference's default width 100)
        sp = _lib.MDL_SPLIT_BF16 if (split and dt == _lib.MDL_F32 and G == 50 and E > 0 and not _DET
                                     and ((C == 64 and x.data_ptr() % 16 == 0) or (_PAD128 and 96 < C <= 128))) else 0
        ctx.split = sp
        wf32, ws32 = w_f.detach().float().contiguous(), w_s.detach().float().contiguous()
        bf32 = None if b_f is None else b_f.detach().float().contiguous()
        bs32 = None if b_s is None else b_s.detach().float().contiguous()
        nbytes = L.mdl_cgconv_wpack_bytes(C, G, dt | sp)
        if nbytes == 0:
            raise MdlError("cgconv: unsupported C=%d G=%d" % (C, G))
        # widths between 97 and 127 (the reference's default dim1 = 100, config.yml:123) run the static 128-channel kernels on
        # zero-padded rows: the packed weights already have that layout (rows / K columns past C are zeros), so the padded
        # output columns hold the constant sigmoid(0) * softplus(0) and are cut off again; their gradients are zeros
        Ck = C
        if _PAD128 and (dt == _lib.MDL_BF16 or sp) and G == 50 and 96 < C < 128 and (csr.eperm is None or sp) and E > 0:
            Ck = 128
            x = torch.nn.functional.pad(x, (0, Ck - C))
        wn_t = None
        if packed is not None and packed[3] == (C, G, dt):
            wpack, bpack, wn_t = packed[:3]               # packed with the model's other layers (cgconv_prepack): no launch here
        else:
            wpack = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            bpack = torch.empty(2 * _rup(C, 32), dtype=torch.float32, device=x.device)
        # a training step on the K3c shapes packs the backward node kernel's operand in the same launch
        if packed is not None and packed[3] == (C, G, dt):
            pass
        elif dt == _lib.MDL_BF16 and (C in (32, 64) or Ck == 128) and any(ctx.needs_input_grad):
            wn_t = torch.empty((C, 4 * _rup(C, 32)), dtype=torch.bfloat16, device=x.device)
            check(L.mdl_cgconv_pack_weights_node(ptr(wf32), ptr(bf32), ptr(ws32), ptr(bs32), C, G, ptr(wpack), ptr(bpack),
                                                 ptr(wn_t), dt, stream()), "mdl_cgconv_pack_weights_node")
        else:
            check(L.mdl_cgconv_pack_weights(ptr(wf32), ptr(bf32), ptr(ws32), ptr(bs32), C, G, ptr(wpack), ptr(bpack), dt | sp,
                                            stream()), "mdl_cgconv_pack_weights")
        ctx.wn_t = wn_t
        out = torch.empty_like(x)
        edge_attr = csr.sorted_attr(edge_attr)          # CSR order: the kernels never go through eperm
        # statistics of the output for the BatchNorm behind the layer, in the kernel's epilogue (bn = (sums buffer, shift)): the
        # static bf16 kernels only — the caller (cgconv_bn_stats_ok) has checked
        bn_sums, bn_shift = bn if bn is not None else (None, None)
        args = _lib.cg_args(dtype=dt, flags=sp, aggr=aggr, N=N, E=E, C=Ck, G=G, x=x, edge_attr=edge_attr, rowptr=csr.rowptr, src=csr.src,
                            tgt=csr.tgt, wpack=wpack, bpack=bpack, out=out, bn_sums=bn_sums, bn_shift=bn_shift,
                            bn_rows=_true_rows_for(N) if bn_sums is not None else None)
        check(_launch_timed("fwd", lambda: L.mdl_cgconv_fwd_ex(args, stream())), "mdl_cgconv_fwd_ex")
        ctx.save_for_backward(x, edge_attr, wf32, ws32, wpack, bpack)
        ctx.csr, ctx.aggr, ctx.has_bias = csr, aggr, (b_f is not None, b_s is not None)
        ctx.wdtypes = (w_f.dtype, w_s.dtype)
        ctx.C = C
        return out if Ck == C else out[:, :C].contiguous()

    @staticmethod
    def backward(ctx, g):
        x, edge_attr, wf32, ws32, wpack, bpack = ctx.saved_tensors
        csr = ctx.csr
        N, Ck = x.shape                                   # Ck: the width the kernels ran at (128 for a padded layer)
        C = getattr(ctx, "C", Ck)
        E, G = edge_attr.shape
        Cp, GP = _rup(C, 32), _rup(G, 64)
        g = g.contiguous()
        dt = dtype_code(x)
        xk, gk = x, g
        if Ck != C:
            gk = torch.nn.functional.pad(g, (0, Ck - C))
            x = x[:, :C]                                  # (the node-level part below works on the true width)
        r_tgt = torch.empty((N, 2 * Cp), dtype=x.dtype, device=x.device)          # by-target sums, compute dtype
        sp = getattr(ctx, "split", 0)
        node_x3 = bool(sp) and dt == _lib.MDL_F32 and C == 64                  # split-product node kernel (fp32 storage)
        node_hip = (dt == _lib.MDL_BF16 and C == Cp and C in (32, 64)) or node_x3   # K3c consumes r_tgt / r_src
        rs16 = (_RSRC16 and dt == _lib.MDL_BF16 and (node_hip or Ck == 128) and G == 50 and E > 0
                and x.data_ptr() % 16 == 0 and edge_attr.data_ptr() % 4 == 0)
        nrs = N * 2 * Cp // 2 if rs16 else N * 2 * Cp                         # fp32 words of the by-source buffer
        keep = _take_rsrc(nrs, x.device) if node_hip else None
        if keep is not None:
            r_src, keep[1] = keep[0][:nrs], True
            r_src = r_src.view(torch.bfloat16).view(N, 2 * Cp) if rs16 else r_src.view(N, 2 * Cp)
        else:
            r_src = torch.zeros((N, 2 * Cp), dtype=torch.bfloat16 if rs16 else torch.float32, device=x.device)
        ldw = 2 * C + G
        direct = node_hip and _DIRECT_GRADS
        if direct:
            # K3 and K3c add their partial sums straight into the two Linears' STACKED weight gradient dW [2C, 2C + G] (rows f | s,
            # columns target | source | edge: MdlCgConv.ld_dwe, MdlCgNode.ld_dwn) and into db [2C]: no assembly kernel, no staging
            # buffers.  One zero-filled slice of the step's gradient arena; dW_f / dW_s / db_f / db_s are views of it.
            gbuf = _zeros_grad(2 * C * ldw + 2 * C, x.device)
            dW = gbuf[:2 * C * ldw].view(2 * C, ldw)
            db = gbuf[2 * C * ldw:]
            dwe, dwn, ld_dwe = dW[:, 2 * C:], dW, ldw
        else:
            small = _zeros_step((2 * Cp * GP + 2 * Cp + 4 * Cp * C,), x.device)
            dwe = small[:2 * Cp * GP].view(2 * Cp, GP)
            db = small[2 * Cp * GP:2 * Cp * GP + 2 * Cp]
            dwn = small[2 * Cp * GP + 2 * Cp:].view(4 * Cp, C)
            ld_dwe = 0
        ws = _workspace(x.device, lib().mdl_cgconv_workspace_bytes(N, E, C, G, dt))
        fl = _dflag()
        # node ranges of equal COST (far sources make a tile dearer): one prefix per batch, shared by all layers
        bal = csr.balance() if (rs16 and _BALANCE and E >= 400000 and not fl) else None
        eargs = _lib.cg_args(dtype=dt, flags=fl | (_k3flag() if rs16 else 0) | getattr(ctx, "split", 0), aggr=ctx.aggr, N=N, E=E, C=Ck, G=G, x=xk, edge_attr=edge_attr,
                             rowptr=csr.rowptr, src=csr.src, tgt=csr.tgt, wpack=wpack, bpack=bpack, grad_out=gk, r_tgt=r_tgt, r_src=r_src,
                             r_src_dtype=_lib.MDL_BF16 if rs16 else _lib.MDL_F32, dwe=dwe, ld_dwe=ld_dwe, db=db, workspace=ws,
                             workspace_bytes=ws.numel(), balance=bal)
        check(_launch_timed("bwd", lambda: lib().mdl_cgconv_bwd_ex(eargs, stream())), "mdl_cgconv_bwd_ex")
        # node-level dense part: rows of Wn / dWn = (f_tgt, s_tgt, f_src, s_src)
        if node_hip:
            wn_t, ctx.wn_t = getattr(ctx, "wn_t", None), None                                      # Wn^T (packed by the forward)
            if wn_t is None:
                wn_t = torch.empty((C, 4 * Cp), dtype=torch.float32 if node_x3 else torch.bfloat16, device=x.device)
                check(lib().mdl_cgconv_pack_node_weights(ptr(wf32), ptr(ws32), C, G, ptr(wn_t), dt, stream()),
                      "mdl_cgconv_pack_node_weights")
            dx = torch.empty_like(x)
            nargs = _lib.cg_node_args(dtype=dt, flags=fl | (sp if node_x3 else 0), zero_src=1 if keep is not None else 0, N=N, C=C,
                                      r_src_dtype=_lib.MDL_BF16 if rs16 else _lib.MDL_F32, ld_dwn=ldw if direct else 0, x=x, grad_out=g,
                                      r_tgt=r_tgt, r_src=r_src, wn_t=wn_t, dx=dx, dwn=dwn)
            check(_launch_timed("bwd_node", lambda: lib().mdl_cgconv_bwd_node_ex(nargs, stream())), "mdl_cgconv_bwd_node_ex")
            if keep is not None:
                keep[1] = False                                                                     # handed back zeroed
            if direct:
                return (dx, None, dW[:C].to(ctx.wdtypes[0]), db[:C].to(ctx.wdtypes[0]) if ctx.has_bias[0] else None,
                        dW[C:].to(ctx.wdtypes[1]), db[C:].to(ctx.wdtypes[1]) if ctx.has_bias[1] else None, None, None, None, None, None)
            dW_f = torch.empty((C, 2 * C + G), dtype=torch.float32, device=x.device)
            dW_s = torch.empty_like(dW_f)
            db_f = torch.empty(C, dtype=torch.float32, device=x.device) if ctx.has_bias[0] else None
            db_s = torch.empty(C, dtype=torch.float32, device=x.device) if ctx.has_bias[1] else None
            check(_launch_timed("bwd_grads", lambda: lib().mdl_cgconv_assemble_grads(
                ptr(dwn), ptr(dwe), ptr(db), C, G, ptr(dW_f), ptr(dW_s), ptr(db_f), ptr(db_s), stream())), "mdl_cgconv_assemble_grads")
            return dx, None, dW_f.to(ctx.wdtypes[0]), db_f, dW_s.to(ctx.wdtypes[1]), db_s, None, None, None, None, None
        if dt == _lib.MDL_BF16 and Cp == 128 and C % 2 == 0 and N > 0:
            # wide layers (C = 100 / 128): the same products on the streaming kernels.  r_tgt / r_src keep their padded
            # [N, 2 Cp] layout (padded columns are exact zeros), so  dx = g + r_tgt Wn_t + r_src Wn_s  is two library GEMMs on
            # zero-padded weights (no gather / cat of the four column blocks) and  dWn = [r_tgt | r_src]^T x  four TN-GEMM
            # launches (128 rows each) into the [4 Cp, C] layout mdl_cgconv_assemble_grads reads — the library's
            # (4C x N)(N x C) form of that contraction ran 531 us per layer on a 64x64x256 macro tile
            rs_b = r_src.to(torch.bfloat16)
            # Wn^T [C, 4 Cp] (columns f_tgt | s_tgt | f_src | s_src, zero-padded to Cp each) comes packed with the forward's weights
            # (one launch for all layers: cgconv_prepack) — round 6; before, the two zero-padded operands were built here with
            # eight fill / slice-copy / cast launches per layer
            wn_t, ctx.wn_t = getattr(ctx, "wn_t", None), None
            if wn_t is None:
                wn_t = torch.empty((C, 4 * Cp), dtype=torch.bfloat16, device=x.device)
                check(lib().mdl_cgconv_pack_node_weights(ptr(wf32), ptr(ws32), C, G, ptr(wn_t), dt, stream()),
                      "mdl_cgconv_pack_node_weights")
            dx = torch.addmm(g, r_tgt, wn_t[:, :2 * Cp].t())
            dx.addmm_(rs_b, wn_t[:, 2 * Cp:].t())
            xc = x.contiguous()
            for blk, r in enumerate((r_tgt[:, :Cp], r_tgt[:, Cp:], rs_b[:, :Cp], rs_b[:, Cp:])):
                check(_gemm_tn(r, r.stride(0), Cp, None, 0, 0, xc, xc.stride(0), C, dwn[blk * Cp:(blk + 1) * Cp], None, N, dt | fl,
                               x.device), "mdl_gemm_tn")
            dW_f = torch.empty((C, 2 * C + G), dtype=torch.float32, device=x.device)
            dW_s = torch.empty_like(dW_f)
            db_f = torch.empty(C, dtype=torch.float32, device=x.device) if ctx.has_bias[0] else None
            db_s = torch.empty(C, dtype=torch.float32, device=x.device) if ctx.has_bias[1] else None
            check(lib().mdl_cgconv_assemble_grads(ptr(dwn), ptr(dwe), ptr(db), C, G, ptr(dW_f), ptr(dW_s), ptr(db_f), ptr(db_s),
                                                  stream()), "mdl_cgconv_assemble_grads")
            return dx, None, dW_f.to(ctx.wdtypes[0]), db_f, dW_s.to(ctx.wdtypes[1]), db_s, None, None, None, None, None
        Wn = torch.cat([wf32[:, :C], ws32[:, :C], wf32[:, C:2 * C], ws32[:, C:2 * C]], dim=0)      # [4C, C]
        rt = r_tgt.view(N, 2, Cp)[:, :, :C]                                                        # library GEMMs
        rs = r_src.view(N, 2, Cp)[:, :, :C]
        cd = torch.float32 if dt == _lib.MDL_F32 else torch.bfloat16
        R = torch.cat([rt[:, 0], rt[:, 1], rs[:, 0], rs[:, 1]], dim=1).to(cd)                      # [N, 4C]
        dx = torch.addmm(g, R, Wn.to(cd))
        dWn = torch.mm(R.t(), x).float()                                                           # [4C, C]
        dwe_f, dwe_s = dwe[:C, :G], dwe[Cp:Cp + C, :G]
        dW_f = torch.cat([dWn[0:C], dWn[2 * C:3 * C], dwe_f], dim=1).to(ctx.wdtypes[0])
        dW_s = torch.cat([dWn[C:2 * C], dWn[3 * C:4 * C], dwe_s], dim=1).to(ctx.wdtypes[1])
        db_f = db[:C].clone() if ctx.has_bias[0] else None
        db_s = db[Cp:Cp + C].clone() if ctx.has_bias[1] else None
        return dx, None, dW_f, db_f, dW_s, db_s, None, None, None, None, None


def cgconv_prepack(convs, x_dtype, device, want_node=True):
    """The packed weights of several CGConv layers of equal shape in ONE launch (mdl_cgconv_pack_weights_multi): a list of
    (wpack, bpack, wn_t, key) to hand to cgconv(..., packed=...), or None when the layers do not share a shape / the fast
    path does not apply.  A model calls it once per forward: the layers' weights are all known before the first one runs."""
    import ctypes
    if len(convs) < 2 or len(convs) > 16 or device.type != "cuda":
        return None
    C, G = convs[0].channels, convs[0].dim
    dt = _lib.MDL_BF16 if x_dtype == torch.bfloat16 else _lib.MDL_F32
    if any(c.channels != C or c.dim != G or c.lin_f.weight.dtype != torch.float32 or not c.lin_f.weight.is_contiguous()
           or not c.lin_s.weight.is_contiguous() or (c.lin_f.bias is None) != (convs[0].lin_f.bias is None) for c in convs):
        return None
    if not (dt == _lib.MDL_BF16 and (C in (32, 64) or (_PAD128 and G == 50 and 96 < C < 128))):
        return None
    L = lib()
    nbytes = L.mdl_cgconv_wpack_bytes(C, G, dt)
    if nbytes == 0:
        return None
    n = len(convs)
    wbuf = torch.empty((n, nbytes), dtype=torch.uint8, device=device)
    bbuf = torch.empty((n, 2 * _rup(C, 32)), dtype=torch.float32, device=device)
    nbuf = torch.empty((n, C, 4 * _rup(C, 32)), dtype=torch.bfloat16, device=device) if want_node else None
    tab = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
    has_b = convs[0].lin_f.bias is not None
    check(L.mdl_cgconv_pack_weights_multi(
        n, tab([c.lin_f.weight.detach() for c in convs]), tab([c.lin_f.bias.detach() for c in convs]) if has_b else None,
        tab([c.lin_s.weight.detach() for c in convs]), tab([c.lin_s.bias.detach() for c in convs]) if has_b else None, C, G,
        tab(list(wbuf)), tab(list(bbuf)), tab(list(nbuf)) if want_node else None, dt, stream()), "mdl_cgconv_pack_weights_multi")
    return [(wbuf[k], bbuf[k], nbuf[k] if want_node else None, (C, G, dt)) for k in range(n)]


def cgconv(x, edge_index, edge_attr, w_f, b_f, w_s, b_s, aggr="mean", csr=None, bn_stats=None, packed=None, split=False):
    """CGConv forward.  x [N,C], edge_index [2,E], edge_attr [E,G]; returns [N,C].
    bn_stats = (sums [2 R + 3, C] fp32 zero-filled, shift [C] fp32 or None): the kernel's epilogue also forms the statistics
    of its output for the BatchNorm behind the layer (callers check cgconv_bn_stats_ok first)."""
    if aggr not in ("mean", "add", "sum"):
        raise MdlError("cgconv: aggr must be mean or add")
    if csr is None:
        csr = csr_for(edge_index, x.shape[0])
    return _CGConvFn.apply(x, edge_attr, w_f, b_f, w_s, b_s, csr, _lib.REDUCE[aggr], bn_stats, packed, split)


# BatchNorm statistics in the CGConv forward's epilogue (mdl_cgconv_fwd_ex, bn_sums): OPT-IN.  Measured on the bench batch
# (profiles/r05_k2_bn_stats_ab.txt, kernel traces of alternating runs): the statistics kernel it removes costs 7.7 us per layer,
# the forward instantiation that carries the epilogue is 12.5 us per layer slower than the plain one (188.0 vs 175.8 us: it
# sits at 256 VGPRs with 28 bytes of scratch, the plain kernel at 252 and none) — time-neutral at 8192 graphs and at 100.
_CG_BN_STATS = False
# K3 / K3c can add their weight-gradient partial sums straight into the stacked dW [2C, 2C + G] (MdlCgConv.ld_dwe, MdlCgNode.ld_dwn:
# no mdl_cgconv_assemble_grads launch).  OPT-IN: measured in one box session (profiles/r05_direct_grads_ab.txt) the node kernel's
# flush into 712-byte rows (every 128-byte atomic instruction straddles two cache lines) costs it +8.5 us at 8192 graphs — what the
# 7-us assembly kernel cost — and +10 us per layer at the reference's batch size, where that flush IS the kernel: 0.59 vs 0.55 ms/step.
_DIRECT_GRADS = False


def cgconv_bn_stats_ok(x, edge_attr, csr):
    """The CGConv forward can form the statistics of its output for the BatchNorm1d behind it: the static bf16 kernels
    (C in {32, 64}, G = 50, edge features in CSR order, 16-byte aligned rows), training with batch statistics, not in
    deterministic mode (there the one-workgroup statistics kernel gives the run-to-run reproducible sums)."""
    return (_CG_BN_STATS and not _DET and x.is_cuda and x.dtype == torch.bfloat16 and edge_attr.dtype == torch.bfloat16 and x.dim() == 2
            and x.shape[1] in (32, 64) and edge_attr.shape[1] == 50 and csr.eperm is None and csr.E > 0 and x.shape[0] >= 2
            and x.is_contiguous() and x.data_ptr() % 16 == 0 and edge_attr.data_ptr() % 4 == 0)


def cgconv_bn(x, edge_index, edge_attr, w_f, b_f, w_s, b_s, aggr, csr, bn_weight, bn_bias, running_mean, running_var, eps, momentum,
              shift=None, packed=None):
    """BatchNorm1d(train)(cgconv(...)) — cgcnn.py:136-145 — with the statistics pass over [N, C] folded into the conv kernel's
    epilogue: the sums are formed about `shift` ([C] fp32 near the column means: the beta of the BatchNorm in FRONT of the layer,
    None = 0) and normalised by mdl_bn_apply_n(MDL_BN_SHIFT_ROW)."""
    C = x.shape[1]
    R = lib().mdl_bn_sums_rows()
    buf = _zeros_step((R + 3, C), x.device)                  # sums (copies + totals) | shift row | save (mean, invstd)
    y = cgconv(x, edge_index, edge_attr, w_f, b_f, w_s, b_s, aggr, csr=csr,
               bn_stats=(buf, None if shift is None else shift.detach().float().contiguous()), packed=packed)
    return _BatchNormTrain.apply(y, bn_weight, bn_bias, running_mean, running_var, eps, momentum, buf)


# ------------------------------------------------------------------------------------------------
# generic gather / edge-weighted gather-reduce (SchNet CFConv, GCNConv, MEGNet, NNConv building blocks)
# ------------------------------------------------------------------------------------------------
class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index):
        require_hip(src, index)
        src = src.contiguous()
        idx = index.to(torch.int32).contiguous()
        C = src.numel() // max(src.shape[0], 1)
        out = torch.empty((idx.numel(),) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        check(lib().mdl_gather_rows(ptr(src), ptr(idx), ptr(out), idx.numel(), C, dtype_code(src), stream()),
              "mdl_gather_rows")
        ctx.index, ctx.n = index, src.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        return scatter(g, ctx.index, 0, ctx.n, "sum"), None


def gather(src, index):
    """src.index_select(0, index) with a HIP forward and a segmented-reduce backward."""
    return _Gather.apply(src, index)


class _GatherMulReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, w, scale, csr, reduce, pre=None):
        # pre: the result, already formed by the fused CFConv forward (cfconv_fused) — this node then only records the graph
        require_hip(h)
        if scale is not None and scale.requires_grad:
            raise MdlError("gather_mul_reduce: gradients w.r.t. the per-edge scale are not implemented (cutoff / GCN "
                           "norm are functions of constant distances on the reference path)")
        h = h.contiguous()
        N, F = csr.N, h.shape[1]
        if w is not None:
            w = w.contiguous()
            if w.dtype != h.dtype:
                raise MdlError("gather_mul_reduce: h and w must share a dtype")
        if scale is not None:
            scale = scale.float().contiguous()
        if pre is not None:
            out = pre.view_as(pre)
        else:
            out = torch.empty((N, F), dtype=h.dtype, device=h.device)
            check(_launch_timed("gmr_fwd", lambda: lib().mdl_gather_mul_reduce(
                ptr(h), ptr(w), ptr(scale), ptr(csr.rowptr), ptr(csr.src), ptr(csr.eperm), ptr(out), N, F, reduce, dtype_code(h),
                stream())), "mdl_gather_mul_reduce")
        ctx.csr, ctx.reduce = csr, reduce
        ctx.save_for_backward(h, w, scale)
        return out

    @staticmethod
    def backward(ctx, g):
        h, w, scale = ctx.saved_tensors
        csr = ctx.csr
        g = g.contiguous()
        if ctx.reduce == _lib.MDL_MEAN:
            deg = (csr.rowptr[1:] - csr.rowptr[:-1]).clamp(min=1).to(g.dtype)
            g = g / deg.unsqueeze(1)
        dh = dw = None
        if (w is not None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and _GMR_DW and h.dtype == torch.bfloat16
                and h.shape[1] % 2 == 0 and h.shape[1] <= 512 and scale is not None):
            # one walk over the by-source CSR gives both gradients (no mdl_edge_mul pass)
            rowptr_s, col_s, eid_s, _ = csr.transposed()
            dh = torch.empty_like(h)
            # (unused edge slots of a padded static batch belong to no by-source segment: their dw rows must read as zeros)
            dw = (torch.zeros_like if _true_rows_for(w.shape[0]) is not None else torch.empty_like)(w)
            check(lib().mdl_gather_mul_reduce_dw(ptr(g), ptr(w), ptr(scale), ptr(rowptr_s), ptr(col_s), ptr(eid_s), ptr(dh), ptr(h),
                                                 ptr(dw), csr.N, h.shape[1], dtype_code(h), stream()), "mdl_gather_mul_reduce_dw")
            return dh, dw, None, None, None, None
        if ctx.needs_input_grad[0]:
            rowptr_s, col_s, eid_s, _ = csr.transposed()
            dh = torch.empty_like(h)
            check(lib().mdl_gather_mul_reduce(ptr(g), ptr(w), ptr(scale), ptr(rowptr_s), ptr(col_s), ptr(eid_s),
                                              ptr(dh), csr.N, h.shape[1], _lib.MDL_SUM, dtype_code(h), stream()),
                  "mdl_gather_mul_reduce(T)")
        if w is not None and ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            check(lib().mdl_edge_mul(ptr(h), ptr(csr.row), ptr(g), ptr(csr.col), ptr(scale), ptr(dw), csr.E,
                                     h.shape[1], dtype_code(h), stream()), "mdl_edge_mul")
        return dh, dw, None, None, None, None


def gather_mul_reduce(h, csr, w=None, scale=None, reduce="sum", pre=None):
    """out[i] = reduce_{edges k -> i} h[src_k] * w[k] * scale[k]  (w: [E,F], scale: [E], caller's edge order)."""
    return _GatherMulReduce.apply(h, w, scale, csr, _lib.REDUCE[reduce], pre)


# ------------------------------------------------------------------------------------------------
# K4 — the fused CFConv forward (csrc/cfconv.hip): filter network + cutoff + h[src] * W -> segmented sum in one pass
# ------------------------------------------------------------------------------------------------
_CFCONV_FUSED = True
_CFCONV_RECOMPUTE = True
# Forward + backward of the block at E = 1.46 M (profiles/r06d_k4_k4b_forms.txt; K4 and K4b both static at the padded width 96 /
# 128 / 160): F = 64: 864 us recomputing / 696 stored / 748 unfused; 80: 899 / 886 / 952; 100: 1048 / 1076 / 1126; 112: 1055 / 1115 /
# 1175; 128: 1215 / 1313 / 1343; 150: 1254 / 1686 / 1794 — the recomputing backward from the 128-wide instantiation on (it also keeps
# 2 E F s bytes of activations per layer out of memory), stored activations below
_CFCONV_RECOMPUTE_MIN_F = 96


def cfconv_fused_ok(rbf, h, csr, lin_a, lin_b):
    """the fused forward takes (edge features [E, 50] bf16 in CSR order, h [N, F] bf16, the two Linears of the filter network):
    shapes mdl_cfconv_fwd supports AND for which both dense layers would take the fused dense path (the backward is theirs)."""
    if not (_CFCONV_FUSED and rbf.is_cuda and rbf.dtype == torch.bfloat16 and h.dtype == torch.bfloat16 and rbf.dim() == 2
            and h.dim() == 2 and rbf.is_contiguous() and csr.eperm is None and rbf.shape[0] == csr.E):
        return False
    F, G = h.shape[1], rbf.shape[1]
    if tuple(lin_a.weight.shape) != (F, G) or tuple(lin_b.weight.shape) != (F, F):
        return False
    # mdl_cfconv_pack_weights reads both weights and both biases as dense fp32 rows
    for t in (lin_a.weight, lin_b.weight, lin_a.bias, lin_b.bias):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda):
            return False
    if not lib().mdl_cfconv_supported(F, G, dtype_code(h)):
        return False
    if torch.is_grad_enabled() and (lin_a.weight.requires_grad or lin_b.weight.requires_grad or h.requires_grad):
        # the autograd nodes this forward stands in for: _LinearActTN (ssp, derivative handed down) -> _LinearActTN -> K4a
        return (linear_act_fused_ok(rbf, lin_a.weight, "ssp") and _hip_shape_ok(F, F) and rbf.shape[0] >= _DENSE_MIN_ROWS
                and lin_a.weight.requires_grad and lin_b.weight.requires_grad)
    return True


def cfconv_fused(rbf, cut, h, csr, lin_a, lin_b, want_acts):
    """One launch (plus the weight packing): out [N, F] and, with want_acts, the layer-1 output a1 [E, F] and the filter w [E, F]."""
    require_hip(rbf, h)
    F, G = h.shape[1], rbf.shape[1]
    E, N = csr.E, csr.N
    dev = h.device
    wpack = torch.empty(lib().mdl_cfconv_wpack_bytes(), dtype=torch.uint8, device=dev)
    check(lib().mdl_cfconv_pack_weights(ptr(lin_a.weight), ptr(lin_a.bias), ptr(lin_b.weight), ptr(lin_b.bias), F, G, ptr(wpack),
                                        stream()), "mdl_cfconv_pack_weights")
    out = torch.empty((N, F), dtype=h.dtype, device=dev)
    a1 = torch.empty((E, F), dtype=h.dtype, device=dev) if want_acts else None
    w = torch.empty((E, F), dtype=h.dtype, device=dev) if want_acts else None
    h = h.contiguous()
    cut = cut.float().contiguous()
    check(_launch_timed("cfconv_fwd", lambda: lib().mdl_cfconv_fwd(
        ptr(rbf), ptr(cut), ptr(h), ptr(csr.rowptr), ptr(csr.src), ptr(csr.tgt), ptr(wpack), ptr(out), ptr(a1), ptr(w), N, E, F, G,
        dtype_code(h), stream())), "mdl_cfconv_fwd")
    return out, a1, w


class BySourceAttrs:
    """(rbf, cut) rows in by-source order for the backward of the recomputing CFConv (K4b): the same for every interaction block
    of a model, so a model makes ONE of these per forward call and hands it to its blocks; the first backward to run gathers
    (one mdl_gather_rows over [E, G] + one over [E]), the others reuse.  Lives exactly as long as the autograd graph of that
    forward — nothing is keyed on tensor identity, so the in-place rewritten buffers of a HIP-graph replay are safe."""

    def __init__(self):
        self._v = None

    def get(self, rbf, cut, eid_s):
        if self._v is None:
            rbf_s = torch.empty_like(rbf)
            check(lib().mdl_gather_rows(ptr(rbf), ptr(eid_s), ptr(rbf_s), eid_s.numel(), rbf.shape[1], dtype_code(rbf), stream()),
                  "mdl_gather_rows")                 # (torch's index_select on [E, 50] bf16 rows: 177 us against 60)
            self._v = (rbf_s, cut.index_select(0, eid_s.long()))
        return self._v


class _CFConvRecompute(torch.autograd.Function):
    """out = CFConv aggregation (mdl_cfconv_fwd, nothing stored per edge); backward: dh by the same kernel on the by-source CSR,
    the filter network's parameter gradients by mdl_cfconv_bwd_w (csrc/cfconv_bwd.hip) — the filter is recomputed in both."""

    @staticmethod
    def forward(ctx, rbf, cut, h, w1, b1, w2, b2, csr, cache):
        require_hip(rbf, h)
        F, G = h.shape[1], rbf.shape[1]
        dev = h.device
        wpack = torch.empty(lib().mdl_cfconv_wpack_bytes(), dtype=torch.uint8, device=dev)
        check(lib().mdl_cfconv_pack_weights(ptr(w1), ptr(b1), ptr(w2), ptr(b2), F, G, ptr(wpack), stream()), "mdl_cfconv_pack_weights")
        out = torch.empty((csr.N, F), dtype=h.dtype, device=dev)
        h = h.contiguous()
        cut = cut.float().contiguous()
        check(_launch_timed("cfconv_fwd", lambda: lib().mdl_cfconv_fwd(
            ptr(rbf), ptr(cut), ptr(h), ptr(csr.rowptr), ptr(csr.src), ptr(csr.tgt), ptr(wpack), ptr(out), None, None, csr.N, csr.E,
            F, G, dtype_code(h), stream())), "mdl_cfconv_fwd")
        ctx.csr, ctx.cache, ctx.has_b = csr, cache, (b1 is not None, b2 is not None)
        ctx.save_for_backward(rbf, cut, h, wpack)
        return out

    @staticmethod
    def backward(ctx, g):
        rbf, cut, h, wpack = ctx.saved_tensors
        csr = ctx.csr
        N, E, F, G = csr.N, csr.E, h.shape[1], rbf.shape[1]
        g = g.contiguous()
        dh = dw1 = db1 = dw2 = db2 = None
        if ctx.needs_input_grad[2]:
            rowptr_s, col_s, eid_s, src_sorted = csr.transposed()
            rbf_s, cut_s = (ctx.cache if ctx.cache is not None else BySourceAttrs()).get(rbf, cut, eid_s)
            dh = torch.empty_like(h)
            check(_launch_timed("cfconv_bwd_h", lambda: lib().mdl_cfconv_fwd(
                ptr(rbf_s), ptr(cut_s), ptr(g), ptr(rowptr_s), ptr(col_s), ptr(src_sorted), ptr(wpack), ptr(dh), None, None, N, E, F, G,
                dtype_code(h), stream())), "mdl_cfconv_fwd(T)")
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[5]:
            buf = torch.zeros(F * G + F * F + 2 * F, dtype=torch.float32, device=h.device)    # (not the step arena: autograd may adopt these views as .grad)
            dw1, dw2 = buf[:F * G].view(F, G), buf[F * G:F * G + F * F].view(F, F)
            db1 = buf[F * G + F * F:F * G + F * F + F] if ctx.has_b[0] else None
            db2 = buf[F * G + F * F + F:] if ctx.has_b[1] else None
            scratch = torch.empty(lib().mdl_cfconv_bwd_w_scratch_bytes(), dtype=torch.uint8, device=h.device)
            check(_launch_timed("cfconv_bwd_w", lambda: lib().mdl_cfconv_bwd_w(
                ptr(rbf), ptr(cut), ptr(h), ptr(g), ptr(csr.rowptr), ptr(csr.src), ptr(csr.tgt), ptr(wpack), ptr(dw1), ptr(db1),
                ptr(dw2), ptr(db2), ptr(scratch), N, E, F, G, dtype_code(h) | _dflag(), stream())), "mdl_cfconv_bwd_w")
        return None, None, dh, dw1, db1, dw2, db2, None, None


def cfconv_recompute(rbf, cut, h, csr, lin_a, lin_b, cache=None):
    """K4 + K4b: the CFConv aggregation as ONE autograd node that stores nothing per edge (see cfconv_fused_ok for the shapes)."""
    return _CFConvRecompute.apply(rbf, cut, h, lin_a.weight, lin_a.bias, lin_b.weight, lin_b.bias, csr, cache)


# ------------------------------------------------------------------------------------------------
# K7 — NNConv edge contraction  m_e = Y[src_e] (Co x D3) . h_e   (csrc/nnconv.hip)
# ------------------------------------------------------------------------------------------------
class _NNConvMsg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Y, h, csr, Co, D3):
        require_hip(Y, h)
        if Y.dtype != h.dtype:
            raise MdlError("nnconv_msg: Y (%s) and h (%s) must share a dtype" % (Y.dtype, h.dtype))
        Y, h = Y.contiguous(), h.contiguous()
        rowptr_s, _, eid_s, _ = csr.transposed()
        # padded static batch (HIP-graph replay): the unused edge slots belong to no by-source segment, so neither kernel
        # writes their rows — they must read as zeros (finite forward values, exactly zero gradients on every padded row)
        ctx.padded = _true_rows_for(h.shape[0]) is not None
        m = (torch.zeros if ctx.padded else torch.empty)((csr.E, Co), dtype=Y.dtype, device=Y.device)
        check(_launch_timed("nnconv_fwd", lambda: lib().mdl_nnconv_msg_fwd(
            ptr(Y), ptr(h), ptr(rowptr_s), ptr(eid_s), ptr(m), csr.N, Co, D3, dtype_code(Y), stream())), "mdl_nnconv_msg_fwd")
        ctx.csr, ctx.dims = csr, (Co, D3)
        ctx.save_for_backward(Y, h)
        return m

    @staticmethod
    def backward(ctx, dm):
        Y, h = ctx.saved_tensors
        Co, D3 = ctx.dims
        rowptr_s, _, eid_s, _ = ctx.csr.transposed()
        dm = dm.contiguous()
        dh, dY = (torch.zeros_like(h) if ctx.padded else torch.empty_like(h)), torch.empty_like(Y)
        check(lib().mdl_nnconv_msg_bwd(ptr(Y), ptr(h), ptr(dm), ptr(rowptr_s), ptr(eid_s), ptr(dh), ptr(dY), ctx.csr.N, Co,
                                       D3, dtype_code(Y), stream()), "mdl_nnconv_msg_bwd")
        return dY, dh, None, None, None


def nnconv_msg(Y, h, csr, out_channels):
    """m[e] = Y[src_e].view(Co, D3) @ h[e]  (caller's edge order): Y [N, Co*D3], h [E, D3] -> [E, Co]."""
    D3 = h.shape[1]
    if Y.shape[1] != out_channels * D3:
        raise MdlError("nnconv_msg: Y has %d columns, expected %d x %d" % (Y.shape[1], out_channels, D3))
    return _NNConvMsg.apply(Y, h, csr, int(out_channels), int(D3))


# ------------------------------------------------------------------------------------------------
# dense Linear on many rows (nodes / graphs): library GEMMs for the forward and dX, the tall-skinny TN HIP
# GEMM for the weight gradient — dW = g^T x contracts over the ROWS (2e5 nodes, 8192 graphs), a shape the
# library handles badly (47 us for a 64 x 64 x 8192 product, 0.7 ms for 64 x 114 x 2e5)
# ------------------------------------------------------------------------------------------------
_TN_COLSUM = True


class _LinearTN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, w_lp=None, b_lp=None):
        # w_lp / b_lp: the caller's copies of weight / bias already in x.dtype (one multi-tensor cast per step instead of
        # two launches per layer); gradients go to the fp32 masters
        w = weight.to(x.dtype) if w_lp is None else w_lp
        b = None if bias is None else (bias.to(x.dtype) if b_lp is None else b_lp)
        out = torch.nn.functional.linear(x, w, b)
        ctx.save_for_backward(x, w)
        ctx.wdtype, ctx.has_bias, ctx.shape = weight.dtype, bias is not None, tuple(weight.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        if _dense_bwd_ok(ctx, g, x, w):
            return _dense_bwd(ctx, g, x, w) + (None, None)
        dx, dw, db = _linear_tn_grads(ctx, g.contiguous(), x, w)
        return dx, dw, db, None, None


def _tn_split_ok(x, weight):
    """dW of a Linear with 256 < in <= 512 as two TN-GEMM halves (_linear_tn_grads): even widths, dword-aligned halves"""
    M, K = weight.shape
    kh = (K // 2 + 1) & ~1
    return (K <= 512 and K % 2 == 0 and M % 2 == 0 and M <= 128 and x.stride(0) % 2 == 0 and x.data_ptr() % 4 == 0
            and kh <= 256 and K - kh <= 256)


def _tn_wide_out_ok(x, weight):
    """dW of a Linear with MANY outputs and few inputs (the GRU's 3C x C gate matrices, mpnn.py:160-161) as TN GEMMs of the
    TRANSPOSED product: dW^T[K, M] = x^T g in column chunks of <= 150 outputs (_linear_tn_grads)"""
    M, K = weight.shape
    return (160 < M <= 600 and M % 2 == 0 and 4 <= K <= 160 and K % 2 == 0 and x.stride(0) % 2 == 0 and x.data_ptr() % 4 == 0)


def _hip_shape_ok(M, K):
    """(out, in) features the streaming dense kernels take: linear.hip forward and gemm_tn.hip weight gradient."""
    return 1 <= M <= 160 and 4 <= K <= 256 and K % 2 == 0 and (M <= 128 or (K <= 160 and M % 2 == 0))


def _dx_hip_ok(g, w):
    M, K = w.shape
    return (g.dtype == torch.bfloat16 and g.is_cuda and g.shape[0] >= _DENSE_MIN_ROWS and g.is_contiguous() and g.data_ptr() % 16 == 0
            and _hip_shape_ok(K, M) and M % 2 == 0)


def _dx_hip(g, w, act_y=None):
    """dX = g W for a tall bf16 g through the streaming dense kernel (the 'weight' of that product is W^T): the library
    picks a 64x64x256 macro tile for [1.5e6, 150] x [150, 150] and runs it at 35 TFLOP/s (1.9 ms); as a stream it is
    E*(M+K)*2 bytes.  act_y = (code, y): g is the gradient w.r.t. the activated output y; the activation derivative is
    applied while the kernel stages its input (mdl_linear_act_in) — callers check _dx_hip_ok first."""
    M, K = w.shape
    if _dx_hip_ok(g, w):
        wt = w.t().contiguous()
        dx = torch.empty((g.shape[0], K), dtype=g.dtype, device=g.device)
        if act_y is not None:
            check(lib().mdl_linear_act_in(ptr(g), ptr(act_y[1]), act_y[0], ptr(wt), None, ptr(dx), g.shape[0], M, K, 0,
                                          dtype_code(g), stream()), "mdl_linear_act_in(dX)")
        else:
            check(lib().mdl_linear_act(ptr(g), ptr(wt), None, ptr(dx), g.shape[0], M, K, 0, dtype_code(g), stream()),
                  "mdl_linear_act(dX)")
        return dx
    assert act_y is None
    return g @ w


_DENSE_BWD_WIDE = True   # ... also when both widths exceed 128 (SchNet's 150 x 150 filter layer)
_DENSE_BWD = True     # dX + dW + db of a tall dense layer in one pass (csrc/dense_bwd.hip)


def _dense_bwd_ok(ctx, g, x, w, y=None):
    """mdl_dense_bwd takes this layer's backward: the input gradient is wanted, even widths in [34, 160], dword-addressable
    rows."""
    M, K = ctx.shape
    kb = K + (1 if ctx.has_bias else 0)
    return (_DENSE_BWD and ctx.needs_input_grad[0] and g.dtype == torch.bfloat16 and g.is_cuda and g.shape[0] >= _DENSE_MIN_ROWS
            and 34 <= M <= 160 and 34 <= K and kb <= 160 and M % 2 == 0 and K % 2 == 0 and (_DENSE_BWD_WIDE or not (M > 128 and kb > 128))
            and g.stride(1) == 1 and x.stride(1) == 1 and g.stride(0) % 2 == 0 and x.stride(0) % 2 == 0
            and g.data_ptr() % 4 == 0 and x.data_ptr() % 4 == 0 and w.is_contiguous() and w.dtype == torch.bfloat16
            and x.dtype == torch.bfloat16
            and (y is None or (y.dtype == torch.bfloat16 and y.stride(1) == 1 and y.stride(0) % 2 == 0 and y.data_ptr() % 4 == 0)))


def _dense_bwd(ctx, g, x, w, act_y=None, xout=0, want_gm=False):
    """(dx, dW, db[, g']) of y = act(x W^T + b) from ONE pass over g, y, x (callers check _dense_bwd_ok); act_y = (code, y):
    g is the gradient w.r.t. the activated output; xout: x is the relu (1) / shifted-softplus (2) output of a private layer in
    front and dx is returned w.r.t. that layer's pre-activation; want_gm: also return g' = g .* act'(y) (bf16, dense)."""
    M, K = ctx.shape
    N = g.shape[0]
    buf = _zeros_grad(M * K + M, g.device)
    dw, dbv = buf[:M * K].view(M, K), buf[M * K:]
    dx = torch.empty((N, K), dtype=g.dtype, device=g.device)
    gm = torch.empty((N, M), dtype=g.dtype, device=g.device) if want_gm else None
    code, y = act_y if act_y is not None else (0, None)
    check(lib().mdl_dense_bwd_ex(ptr(g), g.stride(0), M, ptr(y), 0 if y is None else y.stride(0), code, ptr(x), x.stride(0), K,
                                 ptr(w), ptr(dx), K, xout, ptr(gm), ptr(dw), ptr(dbv) if ctx.has_bias else None,
                                 ptr(_tn_scratch(g.device, N)), N, dtype_code(g) | _dflag(), stream()), "mdl_dense_bwd")
    out = (dx, dw.to(ctx.wdtype), dbv.to(ctx.wdtype) if ctx.has_bias else None)
    return out + (gm,) if want_gm else out


def _tn_act_ok(ctx, g, x, y, w=None):
    """The activation derivative can go into the staging of the backward products instead of a pass of its own: the TN
    GEMM (dW, db) always, the dX product when it runs on the streaming kernel (or is not wanted)."""
    M, K = ctx.shape
    if ctx.needs_input_grad[0] and not (w is not None and g.is_contiguous() and y.is_contiguous() and _dx_hip_ok(g, w)):
        return False
    return (_TN_COLSUM and M % 2 == 0 and K % 2 == 0 and K <= 158 and (M <= 128 or K <= 160)
            and g.dtype == torch.bfloat16 and x.stride(0) % 2 == 0 and g.stride(0) % 2 == 0 and y.stride(0) % 2 == 0
            and x.data_ptr() % 4 == 0 and g.data_ptr() % 4 == 0 and y.data_ptr() % 4 == 0 and g.stride(1) == 1
            and y.stride(1) == 1)


def _linear_tn_grads(ctx, g, x, w, act_y=None):
    """(dx, dW, db) of y = x W^T + b for a tall x: streaming HIP product (or library GEMM) for dx, one TN-GEMM launch for dW
    and db.  act_y = (code, y): g is the gradient w.r.t. the ACTIVATED output y, dx is not wanted (_tn_act_ok), and the
    activation derivative is applied inside the TN GEMM's staging (mdl_gemm_tn_act)."""
    M, K = ctx.shape
    if act_y is not None:
        buf = _zeros_grad(M * K + M, g.device)
        dw, dbv = buf[:M * K].view(M, K), buf[M * K:]
        check(_gemm_tn(g, g.stride(0), M, act_y[1], act_y[1].stride(0), act_y[0], x, x.stride(0), K, dw,
                       dbv if ctx.has_bias else None, g.shape[0], dtype_code(g) | _dflag(), g.device), "mdl_gemm_tn_act")
        dx = _dx_hip(g, w, act_y) if ctx.needs_input_grad[0] else None
        return dx, dw.to(ctx.wdtype), (dbv.to(ctx.wdtype) if ctx.has_bias else None)
    dx = _dx_hip(g, w) if ctx.needs_input_grad[0] else None
    if M > 160:
        # many outputs, few inputs: the transposed product dW^T = x^T g, one TN GEMM per chunk of <= 150 gradient columns
        # (the library runs the (M x N)(N x K) form of this N ~ 6e4 contraction at 234 us for 300 x 100; this is 2 x ~15 us)
        nch = -(-M // 150)
        mc = ((M + nch - 1) // nch + 1) & ~1
        buf = _zeros_grad(K * M, g.device)
        parts, off = [], 0
        for m0 in range(0, M, mc):
            m1 = min(m0 + mc, M)
            c = buf[off:off + K * (m1 - m0)].view(K, m1 - m0)
            off += K * (m1 - m0)
            gs = g[:, m0:m1]
            check(_gemm_tn(x, x.stride(0), K, None, 0, 0, gs, g.stride(0), m1 - m0, c, None, g.shape[0], dtype_code(g) | _dflag(), g.device),
                  "mdl_gemm_tn_colsum")
            parts.append(c)
        dw = torch.cat(parts, dim=1).t()
        db = g.sum(dim=0, dtype=torch.float32).to(ctx.wdtype) if ctx.has_bias else None
        return dx, dw.to(ctx.wdtype), db
    if K > 256:
        # wide inputs (MEGNet's node block: [x | v_e | u[batch]] = 3d columns): the TN GEMM takes <= 256 input columns, so dW
        # is two products over column halves of x (the library's (M x N)(N x K) form runs this K = N ~ 1e5 contraction on 9
        # workgroups: 335 us for 100 x 300 x 1e5)
        kh = (K // 2 + 1) & ~1
        buf = _zeros_grad(M * K + M, g.device)
        d1, d2, dbv = buf[:M * kh].view(M, kh), buf[M * kh:M * K].view(M, K - kh), buf[M * K:]
        fused_db = ctx.has_bias and kh <= 158
        check(_gemm_tn(g, g.stride(0), M, None, 0, 0, x, x.stride(0), kh, d1, dbv if fused_db else None, g.shape[0],
                       dtype_code(g) | _dflag(), g.device), "mdl_gemm_tn_colsum")
        x2 = x[:, kh:]
        check(_gemm_tn(g, g.stride(0), M, None, 0, 0, x2, x.stride(0), K - kh, d2, None, g.shape[0], dtype_code(g) | _dflag(), g.device),
              "mdl_gemm_tn_colsum")
        dw = torch.cat([d1, d2], dim=1)
        db = None
        if ctx.has_bias:
            db = dbv.to(ctx.wdtype) if fused_db else g.sum(dim=0, dtype=torch.float32).to(ctx.wdtype)
        return dx, dw.to(ctx.wdtype), db
    ga, Ma = g, M
    if M % 2:                               # the streaming kernel stages rows as dwords: pad an odd width (the model's
        ga, Ma = torch.nn.functional.pad(g, (0, 1)), M + 1             # 1-column output layer) with a zero column
    # db out of the same pass (mdl_gemm_tn_colsum: the column sums of g ride in a padding column of the B tile and are
    # flushed with one gathered atomic instruction per block); ops.configure(tn_colsum=False) falls back to the library reduction
    fused_db = (_TN_COLSUM and ctx.has_bias and K % 2 == 0 and K <= 158 and x.stride(0) % 2 == 0
                and ga.stride(0) % 2 == 0 and x.data_ptr() % 4 == 0 and ga.data_ptr() % 4 == 0)
    buf = _zeros_grad(Ma * K + Ma, g.device)                                       # dW | db: zero-filled (one fill per step)
    dw, dbv = buf[:Ma * K].view(Ma, K), buf[Ma * K:]
    check(_gemm_tn(ga, ga.stride(0), Ma, None, 0, 0, x, x.stride(0), K, dw, dbv if fused_db else None, g.shape[0],
                   dtype_code(g) | _dflag(), g.device), "mdl_gemm_tn_colsum")
    dw = dw[:M]
    db = None
    if ctx.has_bias:                          # bias gradient = column sums of g: out of the same pass when the shape allows
        db = dbv[:M].to(ctx.wdtype) if fused_db else g.sum(dim=0, dtype=torch.float32).to(ctx.wdtype)
    return dx, dw.to(ctx.wdtype), db


_MLP_HEAD = True        # post-FC head (post_lin_list + lin_out) as one launch per direction
_LINEAR_WIDE = True   # NNConv's Y = x W2r on the streaming kernel


class _LinearActTN(torch.autograd.Function):
    """act(x W^T + b) with the forward as ONE streaming HIP kernel (GEMM + bias + activation) and the backward of
    _LinearTN (one-pass dense backward, or library dX + TN GEMM for dW and db); the ReLU mask comes from the saved output.

    Chains of such layers whose intermediate outputs nobody else sees (nn._seq) hand the activation derivative DOWN the chain:
    in_act = code of the activation that produced x (x is that layer's output): the input gradient is returned w.r.t. that
    layer's PRE-activation (dx .* act'(x), from the x tile the backward stages anyway); out_pre: the gradient arriving here
    already is w.r.t. this layer's pre-activation (the layer behind applied in_act), so no derivative, no read of the output."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_lp, b_lp, act, in_act=0, out_pre=False, pre=None):
        # pre: this layer's output, already formed by a fused forward (cfconv_fused) — the node then only records the graph
        w = weight.to(x.dtype) if w_lp is None else w_lp
        b = None if bias is None else (bias.to(x.dtype) if b_lp is None else b_lp)
        N, K = x.shape
        M = weight.shape[0]
        if pre is not None:
            out = pre.view_as(pre)
        else:
            out = torch.empty((N, M), dtype=x.dtype, device=x.device)
            check(lib().mdl_linear_act(ptr(x), ptr(w), ptr(b), ptr(out), N, K, M, {"relu": 1, "ssp": 2}.get(act, 0), dtype_code(x),
                                       stream()), "mdl_linear_act")
        ctx.save_for_backward(x, w, out if act in ("relu", "ssp") and not out_pre else None)
        ctx.wdtype, ctx.has_bias, ctx.shape, ctx.act = weight.dtype, bias is not None, tuple(weight.shape), act
        ctx.in_act, ctx.out_pre = int(in_act), bool(out_pre)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, out = ctx.saved_tensors
        code = 0 if ctx.out_pre else {"relu": 1, "ssp": 2}.get(ctx.act, 0)
        tail = (None, None, None, None, None, None)
        if _dense_bwd_ok(ctx, g, x, w, out if code else None):
            return _dense_bwd(ctx, g, x, w, (code, out) if code else None, xout=ctx.in_act) + tail
        if code and _tn_act_ok(ctx, g, x, out, w):
            dx, dw, db = _linear_tn_grads(ctx, g, x, w, act_y=(code, out))
        else:
            if code == 1:
                g = torch.ops.aten.threshold_backward(g, out, 0)
            elif code == 2:                            # d/dv (softplus(v) - ln2) = sigmoid(v) = 1 - exp(-(out + ln2))
                g = g.contiguous()
                dpre = torch.empty_like(g)
                check(lib().mdl_ssp_bwd(ptr(g), ptr(out), ptr(dpre), g.numel(), dtype_code(g), stream()), "mdl_ssp_bwd")
                g = dpre
            dx, dw, db = _linear_tn_grads(ctx, g.contiguous(), x, w)
        if dx is not None and ctx.in_act:              # the hand-over on the paths without the fused epilogue
            if ctx.in_act == 1:
                dx = torch.ops.aten.threshold_backward(dx, x, 0)
            else:
                dx = (dx.float() * (1.0 - torch.exp(-(x.float() + 0.6931471805599453)))).to(dx.dtype)
        return (dx, dw, db) + tail


class _LinearGatherAct(torch.autograd.Function):
    """K6: act(x W^T + bias + sum_i p_i[idx_i]) in one streaming HIP kernel; backward = masked gradient -> library dX,
    TN GEMM for dW (+ db), segmented sums of the gradient rows for the gathered tables."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, idx, *tables):
        w = weight.to(x.dtype)
        b = None if bias is None else bias.to(x.dtype)
        N, K = x.shape
        M = weight.shape[0]
        tabs = [None if t is None else t.contiguous() for t in tables]
        tabs += [None] * (3 - len(tabs))
        ids = list(idx) + [None] * (3 - len(idx))
        out = torch.empty((N, M), dtype=x.dtype, device=x.device)
        check(_launch_timed("edge_linear", lambda: lib().mdl_linear_gather_act(
            ptr(x), ptr(w), ptr(b), ptr(tabs[0]), ptr(ids[0]), ptr(tabs[1]), ptr(ids[1]), ptr(tabs[2]), ptr(ids[2]), ptr(out),
            N, K, M, 1 if act == "relu" else 0, dtype_code(x), stream())), "mdl_linear_gather_act")
        ctx.save_for_backward(x, w, out if act == "relu" else None)
        ctx.idx, ctx.rows = ids, [None if t is None else t.shape[0] for t in tabs]
        ctx.wdtype, ctx.has_bias, ctx.shape, ctx.act, ctx.ntab = weight.dtype, bias is not None, tuple(weight.shape), act, len(tables)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, out = ctx.saved_tensors
        if _dense_bwd_ok(ctx, g, x, w, out if ctx.act == "relu" else None):
            # one pass: dX, dW, db and the masked gradient rows the table gradients are segment sums of
            need_gm = any(ctx.needs_input_grad[5 + t] and ctx.rows[t] is not None for t in range(ctx.ntab))
            res = _dense_bwd(ctx, g, x, w, (1, out) if ctx.act == "relu" else None, want_gm=need_gm and ctx.act == "relu")
            dx, dw, db = res[:3]
            g = res[3] if len(res) > 3 else g.contiguous()
        else:
            if ctx.act == "relu":
                g = torch.ops.aten.threshold_backward(g, out, 0)
            g = g.contiguous()
            dx, dw, db = _linear_tn_grads(ctx, g, x, w)
        dts = []
        for t in range(ctx.ntab):
            need = ctx.needs_input_grad[5 + t] and ctx.rows[t] is not None
            dts.append(scatter(g, ctx.idx[t], 0, ctx.rows[t], "sum") if need else None)
        return (dx, dw, db, None, None) + tuple(dts)


def linear_gather_act(x, weight, bias, act, gathered):
    """act(x W^T + bias + sum_i table_i[index_i]) for bf16 rows; `gathered` = [(table [rows, M], index [N] int), ...] (<= 3)."""
    ok = (act in ("relu", None) and x.dtype == torch.bfloat16 and x.is_cuda and x.dim() == 2 and x.is_contiguous()
          and weight.shape[0] <= 128 and 4 <= weight.shape[1] <= 256 and weight.shape[1] % 2 == 0 and x.data_ptr() % 16 == 0
          and len(gathered) <= 3 and x.shape[0] > 0)
    if not ok:
        y = linear(x, weight, bias)
        for tab, ix in gathered:
            y = y + gather(tab, ix)
        return y if act is None else getattr(torch.nn.functional, act)(y)
    idx = [ix if ix.dtype == torch.int32 else ix.to(torch.int32) for _, ix in gathered]
    return _LinearGatherAct.apply(x, weight, bias, act, idx, *[t.to(x.dtype) for t, _ in gathered])


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class _MlpHead(torch.autograd.Function):
    """relu(...relu(x W_0^T + b_0)...) W_last^T + b_last as one launch forward and one backward (csrc/mlp.hip): the post-FC
    head on the pooled graph rows.  params = (W_0, b_0, ..., W_last, b_last) fp32 masters; lowp = their bf16 copies or None."""

    @staticmethod
    def forward(ctx, x, lowp, f32_out, *params):
        import ctypes
        NL = len(params) // 2
        ws = [(params[2 * l].to(x.dtype) if lowp is None or lowp[l] is None else lowp[l][0]).contiguous() for l in range(NL)]
        bs = [None if params[2 * l + 1] is None else
              (params[2 * l + 1].to(x.dtype) if lowp is None or lowp[l] is None else lowp[l][1]) for l in range(NL)]
        N, K0 = x.shape
        M = [int(w.shape[0]) for w in ws]
        # f32_out: the prediction as fp32 rows (the values the bf16 output holds) — the model's `out.float()` behind the head and
        # the cast of the loss gradient in front of its backward are then no launches (MDL_MLP_F32_IO)
        hs = [torch.empty((N, m), dtype=torch.float32 if (f32_out and l == NL - 1) else x.dtype, device=x.device)
              for l, m in enumerate(M)]
        Ma = (ctypes.c_int * NL)(*M)
        io = _lib.MDL_MLP_F32_IO if f32_out else 0
        check(lib().mdl_mlp_head_fwd(ptr(x), _ptr_array(ws), _ptr_array(bs), _ptr_array(hs), N, K0, NL, Ma, dtype_code(x) | io,
                                     stream()), "mdl_mlp_head_fwd")
        ctx.save_for_backward(x, *ws, *hs[:-1])
        ctx.NL, ctx.M, ctx.K0, ctx.io = NL, M, K0, io
        ctx.has_bias = [params[2 * l + 1] is not None for l in range(NL)]
        ctx.wdtypes = [params[2 * l].dtype for l in range(NL)]
        return hs[-1]

    @staticmethod
    def backward(ctx, gy):
        import ctypes
        saved = ctx.saved_tensors
        NL, M, K0 = ctx.NL, ctx.M, ctx.K0
        x, ws, hs = saved[0], list(saved[1:1 + NL]), list(saved[1 + NL:])
        N = x.shape[0]
        gy = gy.contiguous()
        if gy.dtype != (torch.float32 if ctx.io else x.dtype):
            gy = gy.to(torch.float32 if ctx.io else x.dtype)
        Ks = [K0] + M[:-1]
        dws, dbs = [], []
        for l in range(NL):
            buf = _zeros_grad(M[l] * Ks[l] + M[l], x.device)
            dws.append(buf[:M[l] * Ks[l]].view(M[l], Ks[l]))
            dbs.append(buf[M[l] * Ks[l]:] if ctx.has_bias[l] else None)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        Ma = (ctypes.c_int * NL)(*M)
        check(lib().mdl_mlp_head_bwd(ptr(x), _ptr_array(ws), _ptr_array(hs + [None]), ptr(gy), ptr(dx), _ptr_array(dws), _ptr_array(dbs),
                                     N, K0, NL, Ma, dtype_code(x) | _dflag() | ctx.io, stream()), "mdl_mlp_head_bwd")
        grads = []
        for l in range(NL):
            grads.append(dws[l].to(ctx.wdtypes[l]))
            grads.append(dbs[l].to(ctx.wdtypes[l]) if ctx.has_bias[l] else None)
        return (dx, None, None) + tuple(grads)


def mlp_head_ok(x, lins, act):
    """the fused head takes bf16 rows on the device, 1..4 dense layers of width <= 64 (hidden widths even) with ReLU between.
    Any row count: at the reference's batch size (100 graphs, config.yml:136) the layer-by-layer fallback is ~35 launches of
    4-8 us each (library GEMMs, casts, ReLU masks, bias sums) against two."""
    if not (_MLP_HEAD and act == "relu" and x.dtype == torch.bfloat16 and x.is_cuda and x.dim() == 2 and x.is_contiguous()
            and 1 <= len(lins) <= 4 and x.shape[0] >= 1 and x.shape[1] <= 64 and x.shape[1] % 2 == 0 and x.data_ptr() % 4 == 0):
        return False
    k = x.shape[1]
    for j, lin in enumerate(lins):
        if lin.in_features != k or lin.out_features > 64 or (j + 1 < len(lins) and lin.out_features % 2) or not lin.weight.requires_grad:
            return False
        k = lin.out_features
    return True


def mlp_head(x, lins, lowp=None, f32_out=False):
    """lins: the nn.Linear modules of the chain (ReLU after every one but the last); f32_out: the last layer's output as fp32 rows
    (the bf16-rounded values; for a head whose output is the model's fp32 prediction)"""
    params = []
    for lin in lins:
        params += [lin.weight, lin.bias]
    return _MlpHead.apply(x, lowp, bool(f32_out), *params)


class _LinearWide(torch.autograd.Function):
    """y = x @ w for a wide w [K, M] (M in the thousands): forward as a streaming HIP kernel over 192-column blocks
    (mdl_linear_wide), backward = the two library products autograd would form."""

    @staticmethod
    def forward(ctx, x, w):
        N, K = x.shape
        M = w.shape[1]
        wt = w.t().contiguous()                                        # [M, K] rows for the kernel (2 MB for NNConv's W2r)
        out = torch.empty((N, M), dtype=x.dtype, device=x.device)
        check(lib().mdl_linear_wide(ptr(x), ptr(wt), ptr(out), N, K, M, dtype_code(x), stream()), "mdl_linear_wide")
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx = g @ w.t() if ctx.needs_input_grad[0] else None
        dw = x.t() @ g if ctx.needs_input_grad[1] else None
        return dx, dw


def matmul_wide(x, w):
    """x @ w; bf16 tall x (even K <= 160) with a wide w take the streaming kernel for the forward product."""
    if (_LINEAR_WIDE and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.is_cuda and x.dim() == 2 and x.is_contiguous()
            and x.shape[0] >= 1024 and x.shape[1] % 2 == 0 and 4 <= x.shape[1] <= 160 and w.shape[1] >= 640
            and x.data_ptr() % 16 == 0):
        return _LinearWide.apply(x, w)
    return x @ w


def linear_act_fused_ok(x, weight, act):
    """the fused dense layer (_LinearActTN) takes (x, weight, act)"""
    # (ssp: the one-pass softplus backward works on element PAIRS — an odd width would reach it with an odd element count)
    return (act in ("relu", "ssp", None) and x.dtype == torch.bfloat16 and x.is_cuda and x.dim() == 2 and x.is_contiguous()
            and x.shape[0] >= _DENSE_MIN_ROWS and _hip_shape_ok(weight.shape[0], weight.shape[1])
            and (act != "ssp" or weight.shape[0] % 2 == 0)
            and x.data_ptr() % 16 == 0 and weight.requires_grad)


def linear_act(x, weight, bias, act, lowp=None, in_act=None, out_pre=False, pre=None):
    """getattr(F, act)(F.linear(x, weight, bias)) — fused forward for bf16 inputs with dense rows, even in <= 256,
    out <= 128 and act in (relu, ssp, none); anything else composes `linear` with the library activation.
    in_act / out_pre: the activation hand-over of a private chain (see _LinearActTN; callers check linear_act_fused_ok for
    both layers first)."""
    if linear_act_fused_ok(x, weight, act):
        w_lp, b_lp = (lowp if lowp is not None and lowp[0].dtype == x.dtype else (None, None))
        return _LinearActTN.apply(x, weight, bias, w_lp, b_lp, act, {"relu": 1, "ssp": 2}.get(in_act, 0), out_pre, pre)
    assert not in_act and not out_pre and pre is None, "linear_act: activation hand-over on a layer that is not fused"
    y = linear(x, weight, bias, lowp)
    if act is None:
        return y
    if act == "ssp":
        return torch.nn.functional.softplus(y) - 0.6931471805599453
    return getattr(torch.nn.functional, act)(y)


class _LinearSplitTN(torch.autograd.Function):
    """F.linear on fp32 tensors whose WEIGHT GRADIENT dW = g^T x (contraction over the rows: N ~ 2e5 for a node-level layer) runs
    as three bf16 TN-GEMM launches on (hi, lo)-split operands (mdl_split_bf16 + mdl_gemm_tn, fp32 accumulation) instead of the
    library's fp32 product (0.6 ms for the pre-FC layer of the bench batch against 3 x 0.03): the "bf16x3" parity mode.  Forward
    and dX stay the library's exact fp32 products."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        M, K = weight.shape
        N = x.shape[0]
        dx = g @ weight if ctx.needs_input_grad[0] else None

        def split(t):
            t = t.contiguous()
            hi, lo = torch.empty_like(t, dtype=torch.bfloat16), torch.empty_like(t, dtype=torch.bfloat16)
            check(lib().mdl_split_bf16(ptr(t), ptr(hi), ptr(lo), t.numel(), stream()), "mdl_split_bf16")
            return hi, lo
        xh, xl = split(x)
        gh, gl = split(g)
        dw = torch.zeros((M, K), dtype=torch.float32, device=x.device)
        fl = _dflag()
        for a, b in ((gl, xh), (gh, xl), (gh, xh)):
            check(_gemm_tn(a, M, M, None, 0, 0, b, K, K, dw, None, N, _lib.MDL_BF16 | fl, x.device), "mdl_gemm_tn")
        db = g.sum(0) if ctx.has_bias else None
        return dx, dw.to(weight.dtype), db


def linear_split_ok(x, weight):
    """shapes of _LinearSplitTN: fp32 rows on a HIP device, enough of them, even widths inside mdl_gemm_tn's limits"""
    M, K = weight.shape
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 1024
            and M % 2 == 0 and K % 2 == 0 and ((M <= 128 and K <= 256) or (M <= 160 and K <= 160))
            and weight.requires_grad and torch.is_grad_enabled() and not _DET)


def linear(x, weight, bias, lowp=None):
    """F.linear in the dtype of x (fp32 master weights); bf16 inputs with many rows, out <= 128, in <= 256 take the
    HIP TN GEMM for dW, anything else the library autograd path.  `lowp` = (weight, bias) already cast to x.dtype."""
    if (x.dtype == torch.bfloat16 and x.is_cuda and x.dim() == 2 and x.stride(1) == 1 and x.shape[0] >= _DENSE_MIN_ROWS
            and (weight.shape[0] <= 128 or _hip_shape_ok(weight.shape[0], weight.shape[1]) or _tn_wide_out_ok(x, weight))
            and (weight.shape[1] <= 256 or _tn_split_ok(x, weight)) and weight.requires_grad):
        if lowp is not None and lowp[0].dtype == x.dtype:
            return _LinearTN.apply(x, weight, bias, lowp[0], lowp[1])
        return _LinearTN.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight.to(x.dtype), None if bias is None else bias.to(x.dtype))


# ------------------------------------------------------------------------------------------------
# GRU gates (MPNN): one launch per direction instead of ~12 + ~25 elementwise / chunk / cat / cast launches
# ------------------------------------------------------------------------------------------------
class _GRUGates(torch.autograd.Function):
    """(h_new fp32, out) from gi, gh [N, 3C] (gate order r | z | n) and the state h [N, C] fp32 — mpnn.py:160-161; `out` is
    h_new in the dtype of gi (the tensor the next layer reads).  Backward recomputes the gates (csrc/gru.hip)."""

    @staticmethod
    def forward(ctx, gi, gh, h):
        N, C3 = gi.shape
        C = C3 // 3
        gi, gh, h = gi.contiguous(), gh.contiguous(), h.contiguous()
        h_new = torch.empty((N, C), dtype=torch.float32, device=gi.device)
        lp = gi.dtype != torch.float32
        out = torch.empty((N, C), dtype=gi.dtype, device=gi.device) if lp else None
        check(lib().mdl_gru_gates_fwd(ptr(gi), ptr(gh), ptr(h), ptr(h_new), ptr(out), N, C, dtype_code(gi), stream()), "mdl_gru_gates_fwd")
        ctx.save_for_backward(gi, gh, h)
        ctx.lp = lp
        if lp:
            return h_new, out
        dummy = h_new.new_empty(0)            # (fp32: callers use the first output for both roles — gru_gates below)
        ctx.mark_non_differentiable(dummy)
        return h_new, dummy

    @staticmethod
    def backward(ctx, g_h, g_out):
        gi, gh, h = ctx.saved_tensors
        N, C3 = gi.shape
        C = C3 // 3
        if not ctx.lp:
            g_out = None
        g_h = None if g_h is None else g_h.contiguous().float()
        g_out = None if g_out is None else g_out.contiguous().to(gi.dtype)
        if g_h is None and g_out is None:
            return None, None, None
        dgi, dgh = torch.empty_like(gi), torch.empty_like(gh)
        dh = torch.empty_like(h)
        check(lib().mdl_gru_gates_bwd(ptr(gi), ptr(gh), ptr(h), ptr(g_h), ptr(g_out), ptr(dgi), ptr(dgh), ptr(dh), N, C,
                                      dtype_code(gi), stream()), "mdl_gru_gates_bwd")
        return dgi, dgh, dh


def gru_gates_ok(gi, gh, h):
    return (gi.is_cuda and gi.dim() == 2 and gi.shape == gh.shape and gi.dtype == gh.dtype and gi.dtype in (torch.float32, torch.bfloat16)
            and gi.shape[1] % 3 == 0 and h.dtype == torch.float32 and h.shape == (gi.shape[0], gi.shape[1] // 3) and gi.shape[0] > 0)


def gru_gates(gi, gh, h):
    """One GRU step's gates: returns (h_new [N, C] fp32, out [N, C] in gi's dtype — the same tensor as h_new for fp32)."""
    h_new, out = _GRUGates.apply(gi, gh, h)
    return (h_new, out) if gi.dtype != torch.float32 else (h_new, h_new)


# ------------------------------------------------------------------------------------------------
# training loss: value and gradient in one launch
# ------------------------------------------------------------------------------------------------
_UNIT = {}


def unit_grad(device):
    """The constant fp32 scalar 1.0 on `device`, for `loss.backward(gradient=ops.unit_grad(dev))`: autograd's own root gradient is a
    fresh ones_like (a fill launch) that the fused loss then multiplies its stored gradient with (a second one); the loss node
    recognises THIS tensor by its address and hands its gradient on as it is.  Never written to."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    t = _UNIT.get(key)
    if t is None:
        t = _UNIT[key] = torch.ones((), dtype=torch.float32, device=device)
    return t


def backward(loss):
    """loss.backward() with the constant unit root gradient (see unit_grad) when the loss is an fp32 device scalar"""
    if loss.is_cuda and loss.dtype == torch.float32 and loss.dim() == 0:
        loss.backward(gradient=unit_grad(loss.device))
    else:
        loss.backward()


def _is_unit(g):
    t = _UNIT.get((g.device.type, g.device.index))
    return t is not None and g.data_ptr() == t.data_ptr() and g.dim() == 0


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, kind, rows, buf):
        p = pred.contiguous()
        y = target.contiguous()
        # loss | gradient; buf: the caller's persistent fp32 buffer of >= 1 + numel elements (GraphedStep reads the loss value of
        # a replayed step from its first element — no copy launch at the end of the step)
        out = torch.empty(1 + p.numel(), dtype=torch.float32, device=p.device) if buf is None else buf[:1 + p.numel()]
        if rows is None:
            check(lib().mdl_loss_fwd_bwd(ptr(p), ptr(y), p.numel(), kind, ptr(out), ptr(out[1:]), stream()), "mdl_loss_fwd_bwd")
        else:
            check(lib().mdl_loss_fwd_bwd_rows(ptr(p), ptr(y), int(rows), p.numel(), kind, ptr(out), ptr(out[1:]), stream()),
                  "mdl_loss_fwd_bwd_rows")
        ctx.save_for_backward(out)
        ctx.shape = tuple(pred.shape)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        grad = out[1:] if _is_unit(g) else out[1:] * g
        return grad.view(ctx.shape), None, None, None, None


def loss(name, pred, target, rows=None, buf=None):
    """getattr(F, name)(pred, target) as the reference's train() evaluates it (training.py:44-47).  l1_loss / mse_loss on
    fp32 HIP tensors of equal shape compute the value and d loss / d pred in one launch; anything else is torch's.
    rows: the loss of pred[:rows] against target (`rows` elements) with the gradient of the remaining predictions written as
    zeros by the same launch (the padded static batch's dummy graph) — no slice node between the model and the loss.
    buf: optional persistent fp32 device buffer (>= 1 + pred.numel() elements, contiguous) that receives [loss | gradient] when the
    fused kernel runs; the returned loss is then a view of buf[0]."""
    if buf is not None and not (buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous() and buf.dim() == 1
                                and buf.numel() >= 1 + pred.numel() and buf.device == pred.device):
        buf = None
    kind = {"l1_loss": 0, "mse_loss": 1}.get(name)
    if rows is not None:
        rows = int(rows)
        if (kind is not None and pred.is_cuda and pred.dtype == torch.float32 and target.dtype == torch.float32 and pred.dim() == 1
                and target.dim() == 1 and target.numel() == rows and 1 <= rows <= pred.numel() and not target.requires_grad):
            return _FusedLoss.apply(pred, target, kind, rows, buf)
        pred = pred[:rows]
    if (kind is not None and pred.is_cuda and pred.dtype == torch.float32 and target.dtype == torch.float32
            and pred.shape == target.shape and pred.numel() >= 1 and not target.requires_grad):
        return _FusedLoss.apply(pred, target, kind, None, buf)
    return getattr(torch.nn.functional, name)(pred, target)


# ------------------------------------------------------------------------------------------------
# training-mode BatchNorm1d over rows (HIP streams instead of four slow library passes)
# ------------------------------------------------------------------------------------------------
def bn_supported(x):
    if not (x.is_cuda and x.dim() == 2 and x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    c = x.shape[1]
    w = 8 if (x.dtype == torch.bfloat16 and c % 8 == 0) else 4          # 16-byte vectors, or 4 bf16 (C = 100, 150)
    # (N == 1 in training mode goes to the library path, which raises like torch does)
    return x.shape[0] >= 2 and c % 4 == 0 and 4 <= c <= 256 and x.data_ptr() % (w * x.element_size()) == 0


# Static (padded) batches of the HIP-graph path: the tensors hold `capacity` rows, the first *n_rows (a device scalar)
# exist.  Row-count-dependent kernels (BatchNorm) read it on the device.  None = every row exists.
_TRUE_ROWS = None


def _true_rows_for(nrows):
    """device row count for a tensor with `nrows` rows (None = all rows exist)"""
    tr = _TRUE_ROWS
    if tr is None:
        return None
    if isinstance(tr, dict):
        return tr.get(int(nrows))
    return tr


class true_rows:
    """`with ops.true_rows(n_dev):` — n_dev: int64 device tensor with one element, the number of rows that exist; or a dict
    {rows of the padded tensor: device scalar} when node-, edge- and graph-level tensors are padded to different capacities
    (a tensor whose row count is not in the dict is taken as complete)."""

    def __init__(self, n_dev):
        self.n_dev = n_dev

    def __enter__(self):
        global _TRUE_ROWS
        self.prev, _TRUE_ROWS = _TRUE_ROWS, self.n_dev
        return self

    def __exit__(self, *exc):
        global _TRUE_ROWS
        _TRUE_ROWS = self.prev
        return False


def bn_sums_rows():
    """rows of the BatchNorm kernels' sums buffer (replicas + totals)"""
    return int(lib().mdl_bn_sums_rows())


class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, pre=None):
        N, C = x.shape
        dt = dtype_code(x)
        R = lib().mdl_bn_sums_rows()
        # pre: [R + 3, C] buffer whose sums the PRODUCER of x has already formed about the shift in row R (cgconv_bn)
        buf = _zeros_step((R + 2, C), x.device) if pre is None else pre       # rows 0..R-1: sums (copies + totals) [| shift] | save
        sums, save = buf[:R], buf[-2:]
        gw = None if weight is None else weight.detach().float().contiguous()
        gb = None if bias is None else bias.detach().float().contiguous()
        y = torch.empty_like(x)
        nd = _true_rows_for(N)
        # (statistics + apply as ONE launch for few rows was built and measured slower at the reference's batch size — 0.54 vs
        # 0.49 ms/step: C / 8 workgroups walking all rows twice are a longer dependent chain than two wide launches;
        # experiments/patches/bn_one_launch.patch)
        if pre is None:
            check(lib().mdl_bn_stats_n(ptr(x), ptr(sums), N, C, ptr(nd), dt | _dflag(), stream()), "mdl_bn_stats")
        check(lib().mdl_bn_apply_n(ptr(x), ptr(sums), ptr(gw), ptr(gb), ptr(save), ptr(running_mean), ptr(running_var),
                                   ptr(y), N, C, float(eps), float(momentum), ptr(nd),
                                   dt | (_lib.MDL_BN_SHIFT_ROW if pre is not None else 0), stream()), "mdl_bn_apply")
        ctx.n_dev = nd
        ctx.save_for_backward(x, save, gw)
        ctx.has = (weight is not None, bias is not None)
        ctx.wdt = None if weight is None else weight.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, save, gw = ctx.saved_tensors
        N, C = x.shape
        dy = dy.contiguous()
        dt = dtype_code(x)
        R = lib().mdl_bn_sums_rows()
        sums = _zeros_grad(R * C, x.device).view(R, C)                       # (the step's GRADIENT arena, never reused: the totals
        dx = torch.empty_like(x)                                              # rows are returned as parameter gradients)
        nd = ctx.n_dev
        check(lib().mdl_bn_bwd_stats_n(ptr(dy), ptr(x), ptr(save), ptr(sums), N, C, ptr(nd), dt | _dflag(), stream()), "mdl_bn_bwd_stats")
        check(lib().mdl_bn_bwd_apply_n(ptr(dy), ptr(x), ptr(save), ptr(sums), ptr(gw), ptr(dx), N, C, ptr(nd), dt, stream()),
              "mdl_bn_bwd_apply")
        dgamma = sums[R - 1].to(ctx.wdt) if ctx.has[0] else None          # totals row pair published by bwd_apply
        dbeta = sums[R - 2].to(ctx.wdt) if ctx.has[1] else None
        return dx, dgamma, dbeta, None, None, None, None, None


class _LinearReluBN(torch.autograd.Function):
    """BatchNorm1d(relu(x W^T + b [+ sum_i table_i[idx_i]])) in training mode as ONE autograd node (the layer order of the
    reference's MEGNet blocks, megnet.py:41-56).  Forward: the streaming dense layer (K6 when tables are gathered), then the
    BatchNorm statistics and apply kernels.  Backward: the BatchNorm backward's apply pass also takes the ReLU mask (it reads the
    ReLU output anyway: mdl_bn_bwd_apply_relu_n), so what it writes is the gradient w.r.t. the pre-activation — the dense
    backward behind it runs without an activation staging (no second read of the output rows, the faster 8-wave form) and the
    gathered tables' gradients are segment sums of the same rows (no separate masked copy).  (Folding the BatchNorm apply
    into the dense backward's staging instead was measured no faster: experiments/patches/linear_relu_bn_fold.patch.)"""

    @staticmethod
    def forward(ctx, x, weight, bias, w_lp, b_lp, bn_w, bn_b, rm, rv, eps, momentum, idx, *tables):
        w = weight.to(x.dtype) if w_lp is None else w_lp
        b = None if bias is None else (bias.to(x.dtype) if b_lp is None else b_lp)
        N, K = x.shape
        M = weight.shape[0]
        dt = dtype_code(x)
        y = torch.empty((N, M), dtype=x.dtype, device=x.device)
        tabs = [None if t is None else t.contiguous() for t in tables]
        tb = tabs + [None] * (3 - len(tabs))
        ids = list(idx) + [None] * (3 - len(idx))
        R = lib().mdl_bn_sums_rows()
        buf = _zeros_step((R + 3, M), x.device)                       # sums | the epilogue's shift row | save
        sums, save = buf[:R], buf[R + 1:]
        nd = _true_rows_for(N)
        # the statistics of the BatchNorm ride in the dense layer's epilogue (per-column sums of the rounded outputs about output
        # row 0, which the kernel evaluates for itself and leaves in the shift row) —
        # except in deterministic mode, where the one-workgroup statistics kernel gives the run-to-run reproducible sums
        in_epilogue = not _DET and len(tabs) in (0, 2) and K <= 160
        if in_epilogue:
            launch = lambda: lib().mdl_linear_act_stats(
                ptr(x), ptr(w), ptr(b), ptr(tb[0]), ptr(ids[0]), ptr(tb[1]), ptr(ids[1]), ptr(tb[2]), ptr(ids[2]), ptr(y),
                N, K, M, 1, ptr(sums), ptr(nd), dt, stream())
            # (bench.py's K6 events: the edge block's first layer only — the launches with gathered tables)
            check(_launch_timed("edge_linear", launch) if tabs else launch(), "mdl_linear_act_stats")
        elif tabs:
            check(_launch_timed("edge_linear", lambda: lib().mdl_linear_gather_act(
                ptr(x), ptr(w), ptr(b), ptr(tb[0]), ptr(ids[0]), ptr(tb[1]), ptr(ids[1]), ptr(tb[2]), ptr(ids[2]), ptr(y),
                N, K, M, 1, dt, stream())), "mdl_linear_gather_act")
        else:
            check(lib().mdl_linear_act(ptr(x), ptr(w), ptr(b), ptr(y), N, K, M, 1, dt, stream()), "mdl_linear_act")
        gw = None if bn_w is None else bn_w.detach().float().contiguous()
        gb = None if bn_b is None else bn_b.detach().float().contiguous()
        z = torch.empty_like(y)
        if not in_epilogue:
            check(lib().mdl_bn_stats_n(ptr(y), ptr(sums), N, M, ptr(nd), dt | _dflag(), stream()), "mdl_bn_stats")
        check(lib().mdl_bn_apply_n(ptr(y), ptr(sums), ptr(gw), ptr(gb), ptr(save), ptr(rm), ptr(rv), ptr(z), N, M, float(eps),
                                   float(momentum), ptr(nd), dt | (_lib.MDL_BN_SHIFT_ROW if in_epilogue else 0), stream()),
              "mdl_bn_apply")
        ctx.save_for_backward(x, w, y, save, gw)
        ctx.n_dev, ctx.idx, ctx.rows, ctx.ntab = nd, list(idx), [None if t is None else t.shape[0] for t in tabs], len(tabs)
        ctx.wdtype, ctx.has_bias, ctx.shape = weight.dtype, bias is not None, tuple(weight.shape)
        ctx.bn_has = (bn_w is not None, bn_b is not None)
        ctx.bn_wdt = None if bn_w is None else bn_w.dtype
        return z

    @staticmethod
    def backward(ctx, gz):
        x, w, y, save, gw = ctx.saved_tensors
        M, K = ctx.shape
        N = x.shape[0]
        gz = gz.contiguous()
        dt = dtype_code(x)
        R = lib().mdl_bn_sums_rows()
        sums = _zeros_grad(R * M, x.device).view(R, M)
        gp = torch.empty_like(y)                                                 # gradient w.r.t. the Linear's output (pre-activation)
        check(lib().mdl_bn_bwd_stats_n(ptr(gz), ptr(y), ptr(save), ptr(sums), N, M, ptr(ctx.n_dev), dt | _dflag(), stream()),
              "mdl_bn_bwd_stats")
        check(lib().mdl_bn_bwd_apply_relu_n(ptr(gz), ptr(y), ptr(save), ptr(sums), ptr(gw), ptr(gp), N, M, ptr(ctx.n_dev), dt,
                                            stream()), "mdl_bn_bwd_apply_relu")
        dgamma = sums[R - 1].to(ctx.bn_wdt) if ctx.bn_has[0] else None
        dbeta = sums[R - 2].to(ctx.bn_wdt) if ctx.bn_has[1] else None
        buf = _zeros_grad(M * K + M, x.device)
        dw, dbv = buf[:M * K].view(M, K), buf[M * K:]
        dx = torch.empty((N, K), dtype=x.dtype, device=x.device)
        check(lib().mdl_dense_bwd_ex(ptr(gp), M, M, None, 0, 0, ptr(x), x.stride(0), K, ptr(w), ptr(dx), K, 0, None, ptr(dw),
                                     ptr(dbv) if ctx.has_bias else None, ptr(_tn_scratch(x.device, N)), N, dt | _dflag(), stream()),
              "mdl_dense_bwd")
        dts = []
        for t in range(ctx.ntab):
            need = ctx.needs_input_grad[12 + t] and ctx.rows[t] is not None
            dts.append(scatter(gp, ctx.idx[t], 0, ctx.rows[t], "sum") if need else None)
        return (dx if ctx.needs_input_grad[0] else None, dw.to(ctx.wdtype), dbv.to(ctx.wdtype) if ctx.has_bias else None,
                None, None, dgamma, dbeta, None, None, None, None, None) + tuple(dts)


def linear_relu_bn_ok(x, weight, has_bias, gathered=None):
    """The fused Linear -> ReLU -> BatchNorm1d(train) node takes this layer: bf16 rows with a gradient, the dense-backward
    shapes (even widths in [34, 160], not both above 128), at most three gathered tables."""
    M, K = weight.shape
    kb = K + (1 if has_bias else 0)
    return (_DENSE_BWD and x.dtype == torch.bfloat16 and x.is_cuda and x.dim() == 2 and x.is_contiguous() and x.shape[0] >= _DENSE_MIN_ROWS
            and x.requires_grad and torch.is_grad_enabled() and weight.requires_grad
            and 34 <= M <= (128 if gathered else 160) and 34 <= K and kb <= 160 and M % 2 == 0 and K % 2 == 0
            and not (M > 128 and kb > 128) and M % 4 == 0 and x.data_ptr() % 16 == 0 and (gathered is None or len(gathered) <= 3))


def linear_relu_bn(x, weight, bias, lowp, bn_weight, bn_bias, running_mean, running_var, eps, momentum, gathered=None):
    """BatchNorm1d(relu(F.linear(x, weight, bias) + sum_i table_i[index_i])) with batch statistics (callers check
    linear_relu_bn_ok); `gathered` = [(table [rows, M], index [N] int), ...]."""
    w_lp, b_lp = (lowp if lowp is not None and lowp[0].dtype == x.dtype else (None, None))
    gathered = gathered or []
    idx = [ix if ix.dtype == torch.int32 else ix.to(torch.int32) for _, ix in gathered]
    return _LinearReluBN.apply(x, weight, bias, w_lp, b_lp, bn_weight, bn_bias, running_mean, running_var, eps, momentum, idx,
                               *[t.to(x.dtype) for t, _ in gathered])


def batch_norm_train(x, weight, bias, running_mean, running_var, eps=1e-5, momentum=0.1):
    return _BatchNormTrain.apply(x, weight, bias, running_mean, running_var, eps, momentum)
